"""Times the bench-step GEMM shapes under the current IA_GEMM_DBG (read once per process) and checks one result against torch.
usage: IA_GEMM_DBG=n python tools/abl/gemm_dbg_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


tag = os.environ.get("IA_GEMM_DBG", "0")
torch.manual_seed(0)
a = torch.randn((2048, 4096 + 64), device=dev).bfloat16(); w = (torch.randn((1024, 4096 + 64), device=dev) * 0.05).bfloat16()
ref = a.float() @ w.float().t()
err = ((ops.gemm(a, w).float() - ref).abs().max() / ref.abs().max()).item()
at, wt = a.t().contiguous(), w.t().contiguous()
err2 = ((ops.gemm(at, wt, a_kstrided=True, b_kstrided=True, out_f32=True) - ref).abs().max() / ref.abs().max()).item()
print(f"dbg={tag} rel err NT {err:.1e} TN {err2:.1e}", flush=True)
shapes = [("qkv  NT", 65280, 3072, 1024, 0, 0), ("ffn2 NT", 65280, 1024, 4096, 0, 0), ("dX   NN", 65280, 1024, 4096, 0, 1),
          ("dX2  NN", 65280, 4096, 1024, 0, 1), ("dW   TN", 4096, 1024, 65280, 1, 1), ("vit fc1", 147712, 3072, 768, 0, 0),
          ("4k^3 NT", 4096, 4096, 4096, 0, 0), ("8k^3 NT", 8192, 8192, 8192, 0, 0)]
for rep in range(2):
    for name, M, N, K, aks, bks in shapes:
        a = torch.randn((K, M) if aks else (M, K), device=dev).bfloat16()
        b = torch.randn((K, N) if bks else (N, K), device=dev).bfloat16()
        f32 = bool(aks)
        out = torch.empty((M, N), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, b, a_kstrided=bool(aks), b_kstrided=bool(bks), out=out, out_f32=f32))
        print(f"dbg={tag} {name} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
