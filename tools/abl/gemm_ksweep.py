"""time of the plain NT GEMM against K at fixed M x N: slope = time per k-tile, intercept = per-tile overhead (prologue + epilogue).
usage: gemm_ksweep.py [M N]     (IA_GEMM_DBG bits as in gemm.hip: 64 = skip the output stores, 2 = no DMA, 16 = no k-loop barriers)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (65280, 1024)
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


tiles = ((M + 255) // 256) * ((N + 255) // 256)
rounds = (tiles + 255) // 256
pts = []
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    a = torch.randn((M, K), device=dev).bfloat16(); b = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, b, out=out))
    pts.append((K // 64, t * 1e6))
    print(f"M={M} N={N} K={K:5d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s   per tile-round {t*1e6/rounds:7.2f} us")
# least squares through the last four points
xs = [p[0] for p in pts[-4:]]; ys = [p[1] / rounds for p in pts[-4:]]
n = len(xs); mx, my = sum(xs) / n, sum(ys) / n
slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
print(f"tiles {tiles} = {rounds} rounds of 256 CUs; per k-tile {slope:.3f} us ({2*256*256*64/slope/1e6*256/1e6:.0f} TF/s k-loop rate), per-tile overhead {my - slope * mx:.2f} us")
