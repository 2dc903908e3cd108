"""Does a power-of-two row stride cost the GEMM (L2 / HBM channel conflicts on the operand DMA and the output stores)?
NT GEMM M x N x K with leading dimensions lda = K + pa, ldb = K + pb, ldc = N + pc (elements)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import _lib
from item_alignment_amd.ops import stream_ptr, check
lib = _lib.load()
dev = torch.device("cuda:0")


def run(M, N, K, pa, pb, pc):
    a = torch.randn((M, K + pa), device=dev).bfloat16(); b = (torch.randn((N, K + pb), device=dev) * 0.05).bfloat16()
    c = torch.empty((M, N + pc), device=dev, dtype=torch.bfloat16)
    def f():
        check(lib.ia_gemm_bf16(a.data_ptr(), 0, K + pa, b.data_ptr(), 0, K + pb, c.data_ptr(), 0, N + pc, M, N, K, 0, None, None, 0, None, 0, None, 0,
                               stream_ptr()), "gemm")
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    ref = a[:, :K].float() @ b[:, :K].float().t() if M * N <= 2 ** 24 else None
    err = "" if ref is None else f" rel err {((c[:, :N].float() - ref).abs().max() / ref.abs().max()).item():.1e}"
    print(f"M={M} N={N} K={K} lda=K+{pa:3d} ldb=K+{pb:3d} ldc=N+{pc:3d}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s{err}")


for (M, N, K) in ((65280, 4096, 1024), (65280, 1024, 4096), (65280, 1024, 1024)):
    for (pa, pb, pc) in ((0, 0, 0), (0, 0, 64), (0, 0, 128), (64, 64, 0), (64, 64, 64), (128, 128, 128), (0, 0, 0)):
        run(M, N, K, pa, pb, pc)
run(2048, 1024, 1024, 64, 64, 64)
