"""Time every epilogue variant of the NT GEMM at the FFN1 shape (use with IA_GEMM_DBG=32 / 64 to look at the main loop alone)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
M, H, I = 32640, 1024, 4096
x = torch.randn((M, H), device=dev).bfloat16(); w1 = (torch.randn((I, H), device=dev) * 0.03).bfloat16()
b1 = torch.zeros(I, device=dev); aux = torch.randn((M, I), device=dev).bfloat16()
pre = torch.empty((M, I), device=dev, dtype=torch.bfloat16); act = torch.empty_like(pre)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for rep in range(2):
    for name, fn in [("none", lambda: ops.gemm(x, w1, out=act)), ("bias", lambda: ops.gemm(x, w1, epilogue=ops.EPI_BIAS, bias=b1, out=act)),
                     ("bias_gelu", lambda: ops.gemm(x, w1, epilogue=ops.EPI_BIAS_GELU, bias=b1, out=act, pre_out=pre)),
                     ("bias_add", lambda: ops.gemm(x, w1, epilogue=ops.EPI_BIAS_ADD, bias=b1, aux=aux, out=act)),
                     ("add", lambda: ops.gemm(x, w1, epilogue=ops.EPI_ADD, aux=aux, out=act))]:
        print(f"DBG={os.environ.get('IA_GEMM_DBG', '0'):>3} {name:10s} {timeit(fn):8.1f} us", flush=True)
