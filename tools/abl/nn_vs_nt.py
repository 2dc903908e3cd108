"""Data-gradient GEMMs dx = dy W: today W[K_out, N_in] is read k-strided (B[k][n], transpose reads in LDS); with a transposed bf16
shadow W^T[N_in, K_out] the same product is the k-contiguous NT form.  What would the transposed shadow buy, per bench shape?"""
import os, sys
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
    return s.elapsed_time(e) / iters * 1e3

tot_nn = tot_nt = 0.0
for name, M, N, K, count, epi in [("text ffn1 dgrad", 65280, 1024, 4096, 24, 0), ("text out-proj dgrad", 65280, 1024, 1024, 24, 0), ("text qkv dgrad", 65280, 1024, 3072, 24, 0),
                                  ("text ffn2 dgrad x gelu'", 65280, 4096, 1024, 24, 1), ("vit fc1 dgrad", 295424, 768, 3072, 12, 0), ("vit proj dgrad", 295424, 768, 768, 12, 0),
                                  ("vit qkv dgrad", 295424, 768, 2304, 12, 0), ("vit fc2 dgrad x gelu'", 295424, 3072, 768, 12, 1)]:
    dy = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((K, N), device=dev) * 0.03).to(torch.bfloat16)          # weight [out = K, in = N]: B[k][n]
    wt = w.t().contiguous()                                                   # [N, K]: k-contiguous
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    aux = torch.randn((M, N), device=dev).to(torch.bfloat16) if epi else None
    kw = dict(epilogue=ops.EPI_DGELU, aux=aux) if epi else {}
    t_nn = timeit(lambda: ops.gemm(dy, w, b_kstrided=True, out=out, **kw))
    kw_nt = dict(epilogue=ops.EPI_ADD, aux=aux) if epi else {}               # (the NT dispatch has no x aux epilogue yet: + aux costs the same)
    t_nt = timeit(lambda: ops.gemm(dy, wt, out=out, **kw_nt))
    fl = 2.0 * M * N * K
    print(f"{name:26s} M={M:6d} N={N:4d} K={K:4d}: NN {t_nn:8.1f} us {fl/t_nn/1e6:7.1f} TF/s | NT {t_nt:8.1f} us {fl/t_nt/1e6:7.1f} TF/s | x{t_nn/t_nt:.3f}  ({count}/step)")
    tot_nn += t_nn * count; tot_nt += t_nt * count
print(f"per step: NN {tot_nn/1e3:.1f} ms, NT {tot_nt/1e3:.1f} ms, saving {(tot_nn-tot_nt)/1e3:.1f} ms")
