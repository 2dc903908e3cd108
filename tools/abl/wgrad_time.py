"""weight-gradient GEMMs of the bench step at full K: time and rate per shape (HIP events, 10 launches)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
for name, M, N, K in [("fc1/fc2 text", 4096, 1024, 130560), ("qkv text", 3072, 1024, 130560), ("wo text", 1024, 1024, 130560),
                      ("fc1 vit", 3072, 768, 295424), ("qkv vit", 2304, 768, 295424), ("proj vit", 768, 768, 295424)]:
    a = torch.randn((K, M), device=dev).bfloat16(); b = torch.randn((K, N), device=dev).bfloat16()
    out = torch.zeros((M, N), device=dev, dtype=torch.float32)
    f = lambda: ops.gemm(a, b, a_kstrided=True, b_kstrided=True, out=out, out_f32=True, accumulate=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e-3
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"{name:14s} M={M} N={N} K={K}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s  ({tiles} tiles)", flush=True)
    del a, b
