"""Achieved HBM GB/s of the GEMM on the 1x1-convolution shapes of eca_nfnet_l0 at 32 images of 800x800 (memory-bound: bytes = M(K+N)2)."""
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


shapes = [("s0 conv1", 1280000, 64, 128), ("s0 conv3", 1280000, 256, 64), ("s0 down ", 1280000, 256, 128), ("s1 conv1", 320000, 128, 256),
          ("s1 conv3", 320000, 512, 128), ("s2 conv1", 80000, 384, 1536), ("s2 conv3", 80000, 1536, 384), ("s3 conv3", 20000, 1536, 384)]
for name, M, N, K in shapes:
    x = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.05).bfloat16(); b = torch.zeros(N, device=dev)
    y = torch.empty((M, N), device=dev, dtype=torch.bfloat16); dy = torch.randn((M, N), device=dev).bfloat16(); dx = torch.empty_like(x)
    t = timeit(lambda: ops.gemm(x, w, epilogue=ops.EPI_BIAS, bias=b, out=y))
    print(f"{name} fwd  M={M} N={N} K={K}: {t*1e6:7.1f} us  {M*(K+N)*2/t/1e9:6.0f} GB/s", flush=True)
    t = timeit(lambda: ops.gemm(dy, w, b_kstrided=True, out=dx))          # dx[M,K] = dy[M,N] * w[N,K]
    print(f"{name} dgrad                      : {t*1e6:7.1f} us  {M*(K+N)*2/t/1e9:6.0f} GB/s", flush=True)
    dw = torch.empty((N, K), device=dev, dtype=torch.float32)
    t = timeit(lambda: ops.gemm(dy, x, a_kstrided=True, b_kstrided=True, out=dw, out_f32=True))
    print(f"{name} wgrad                      : {t*1e6:7.1f} us  {M*(K+N)*2/t/1e9:6.0f} GB/s", flush=True)
