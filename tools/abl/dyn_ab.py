"""Same-box A/B of the static and the dynamic tile order on the bench's GEMM shapes (no contention): in-process toggle, alternating."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

M = 65280
for name, m, n, k, bks, epi in [("qkv NT bias", M, 3072, 1024, 0, ops.EPI_BIAS), ("ffn1 NT bias+gelu", M, 4096, 1024, 0, ops.EPI_BIAS_GELU), ("ffn2 NT", M, 1024, 4096, 0, ops.EPI_NONE),
                                ("out-proj NT", M, 1024, 1024, 0, ops.EPI_NONE), ("dgrad NN", M, 4096, 1024, 1, ops.EPI_NONE), ("vit fc1 NT bias+gelu", 295424, 3072, 768, 0, ops.EPI_BIAS_GELU),
                                ("vit qkv NT bias", 295424, 2304, 768, 0, ops.EPI_BIAS)]:
    a = torch.randn((m, k), device=dev).to(torch.bfloat16)
    b = (torch.randn((k, n) if bks else (n, k), device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty((m, n), device=dev, dtype=torch.bfloat16)
    kw = {}
    if epi in (ops.EPI_BIAS, ops.EPI_BIAS_GELU): kw["bias"] = torch.zeros(n, device=dev)
    if epi == ops.EPI_BIAS_GELU: kw["pre_out"] = torch.empty_like(out)
    run = lambda: ops.gemm(a, b, b_kstrided=bool(bks), out=out, epilogue=epi, **kw)
    ts = {0: [], 1: []}
    for rep in range(4):
        for mode in (0, 1):
            lib.ia_debug_gemm_dynamic(mode)
            ts[mode].append(timeit(run))
    s0, s1 = sorted(ts[0])[1], sorted(ts[1])[1]
    print(f"{name:22s} M={m:6d} N={n:4d} K={k:4d}: static {s0:8.1f} us | dynamic {s1:8.1f} us | dynamic/static {s1 / s0:.4f}")
