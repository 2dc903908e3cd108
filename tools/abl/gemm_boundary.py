"""Tile-boundary cost of the plain NT GEMM at K = 1024 under the current IA_GEMM_DBG (0: as is, 64: no stores, 32: no epilogue), with the
vendor library's time for the same shapes beside it.  usage: IA_GEMM_DBG=n python tools/abl/gemm_boundary.py"""
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
for name, M, N, K in [("ffn1", 65280, 4096, 1024), ("qkv", 65280, 3072, 1024), ("k512", 65280, 4096, 512), ("k2048", 65280, 4096, 2048), ("ffn2", 65280, 1024, 4096)]:
    a = torch.randn((M, K), device=dev).bfloat16(); b = torch.randn((N, K), device=dev).bfloat16()
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, b, out=out))
    line = f"dbg={tag} {name} M={M} N={N} K={K}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s"
    if tag == "0":
        tv = timeit(lambda: torch.matmul(a, b.t(), out=out))
        line += f" | vendor {tv*1e6:8.1f} us {2*M*N*K/tv/1e12:7.1f} TF/s"
    print(line, flush=True)
