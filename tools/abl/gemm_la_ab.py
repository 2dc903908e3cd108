"""t256la against t256w (IA_GEMM_LA = 1 / 0 in one process) on the plain NT shapes of the bench step: HIP-event time per launch, interleaved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
shapes = [("text out-proj / dgrad K=1024", 130560, 1024, 1024), ("text qkv dgrad K=3072", 130560, 1024, 3072), ("text ffn2 / ffn1 dgrad K=4096", 130560, 1024, 4096),
          ("vit proj K=768", 295424, 768, 768), ("vit qkv dgrad K=2304", 295424, 768, 2304), ("vit fc1 dgrad K=3072", 295424, 768, 3072), ("K=512", 130560, 1024, 512)]
for name, M, N, K in shapes:
    a = torch.randn((M, K), device=dev).bfloat16(); b = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    res = {0: [], 1: []}
    for rep in range(3):
        for la in (1, 0):
            os.environ["IA_GEMM_LA"] = str(la)
            res[la].append(timeit(lambda: ops.gemm(a, b, out=out)))
    t1, t0 = min(res[1]), min(res[0])
    print(f"{name:32s} M={M} N={N} K={K}: t256w {t0*1e6:8.1f} us ({2*M*N*K/t0/1e12:6.1f} TF/s)  t256la {t1*1e6:8.1f} us ({2*M*N*K/t1/1e12:6.1f} TF/s)  x{t0/t1:.3f}", flush=True)
