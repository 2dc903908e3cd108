"""Does the 256 MiB Infinity Cache keep a buffer that one kernel writes and the next one reads?  In-place scaling of a bf16 buffer of S MB
(read S + write S per pass) and a two-buffer ping-pong (a -> b, b -> a): effective GB/s per pass against S.  Far above the ~5-6 TB/s an
HBM stream reaches = the working set stays on the die between launches.     usage: python tools/abl/mall_probe.py"""
import torch

dev = torch.device("cuda:0")
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096):
    n = mb * (1 << 20) // 2
    a = torch.randn(n, device=dev).bfloat16()
    b = torch.empty_like(a)
    for name, fn, traffic in (("in place", lambda: a.mul_(1.0), 2 * mb), ("ping-pong", lambda: (torch.mul(a, 1.0, out=b), torch.mul(b, 1.0, out=a)), 4 * mb)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(10, 4096 // mb)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{mb:5d} MB  {name:9s}  {ms * 1e3:9.1f} us per pass  {traffic / 1024 / (ms / 1e3) / 1e3:6.2f} TB/s effective", flush=True)
