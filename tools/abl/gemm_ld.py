"""Ablation: does the row stride (leading dimension) of A / B / C change the 256x256 GEMM rate?  (power-of-two strides vs padded)
usage: python tools/abl/gemm_ld.py  (IA_GEMM_WIDE=0/1 picks the kernel)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import _lib
from item_alignment_amd.ops import stream_ptr

lib = _lib.load()
dev = torch.device("cuda:0")


def run(M, N, K, lda, ldb, ldc, bks=0):
    a = torch.randn((M, lda), device=dev).bfloat16()
    b = torch.randn((K, ldb) if bks else (N, ldb), device=dev).bfloat16()
    c = torch.empty((M, ldc), device=dev, dtype=torch.bfloat16)
    def f():
        r = lib.ia_gemm_bf16(a.data_ptr(), 0, lda, b.data_ptr(), bks, ldb, c.data_ptr(), 0, ldc, M, N, K, 0, None, None, 0, None, 0, None, 0, stream_ptr())
        assert r == 0
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e-3
    print(f"WIDE={os.environ.get('IA_GEMM_WIDE','0')} M={M} N={N} K={K} lda={lda} ldb={ldb} ldc={ldc} bks={bks}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)


for M, N, K in [(8192, 8192, 8192), (65280, 4096, 1024), (65280, 1024, 4096)]:
    run(M, N, K, K, K, N)
    run(M, N, K, K + 64, K + 64, N)
    run(M, N, K, K + 64, K + 64, N + 64)
    run(M, N, K, K, N, N, bks=1)
    run(M, N, K, K + 64, N + 64, N + 64, bks=1)
