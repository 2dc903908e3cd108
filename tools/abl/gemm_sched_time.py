"""Times large GEMM shapes with the one-wave-per-SIMD kernel (IA_GEMM_WIDE=1).  usage: IA_GEMM_WIDE=1 python tools/abl/gemm_sched_time.py tag"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else ""


def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


shapes = [("qkv  NT", 65280, 3072, 1024, 0, 0), ("ffn2 NT", 65280, 1024, 4096, 0, 0), ("dX   NN", 65280, 1024, 4096, 0, 1),
          ("dX2  NN", 65280, 4096, 1024, 0, 1), ("dW   TN", 4096, 1024, 65280, 1, 1), ("vit fc1", 147712, 3072, 768, 0, 0),
          ("8k^3 NT", 8192, 8192, 8192, 0, 0), ("8k^3 NN", 8192, 8192, 8192, 0, 1), ("8k^3 TN", 8192, 8192, 8192, 1, 1)]
res = {}
for rep in range(3):
    for name, M, N, K, aks, bks in shapes:
        a = torch.randn((K, M) if aks else (M, K), device=dev).bfloat16()
        b = torch.randn((K, N) if bks else (N, K), device=dev).bfloat16()
        f32 = bool(aks)
        out = torch.empty((M, N), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, b, a_kstrided=bool(aks), b_kstrided=bool(bks), out=out, out_f32=f32))
        res.setdefault(name, []).append(2 * M * N * K / t / 1e12)
        if os.environ.get("IA_TORCH_REF") and not aks:          # hipBLASLt through torch on the same operands (NT / NN)
            tt = timeit(lambda: torch.matmul(a, b.t() if not bks else b))
            res.setdefault(name + " [torch]", []).append(2 * M * N * K / tt / 1e12)
print(tag, " | ".join(f"{n} {min(v):.0f}-{max(v):.0f}" for n, v in res.items()), flush=True)
