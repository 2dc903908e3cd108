"""Edge shapes of the 256x256 GEMM kernels (one / two k-tiles, ragged K, ragged M / N, split-K weight gradients) against torch, for
IA_GEMM_WIDE = 0 / 1 / unset.  usage: python tools/abl/gemm_edge.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(1)
worst = 0.0
for (M, N, K) in [(1024, 1024, 64), (1024, 1024, 128), (1024, 768, 8), (1000, 520, 72), (2048, 2048, 192), (4096, 1024, 1024), (264, 272, 320),
                  (3000, 4096, 64), (512, 512, 4096 + 8)]:
    a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.1).bfloat16()
    ref = a.float() @ w.float().t()
    bias = torch.randn(N, device=dev); aux = torch.randn((M, N), device=dev).bfloat16()
    sc = ref.abs().max().item() + 1e-6
    errs = [((ops.gemm(a, w).float() - ref).abs().max() / sc).item(),
            ((ops.gemm(a, w, epilogue=ops.EPI_BIAS, bias=bias).float() - (ref + bias)).abs().max() / sc).item(),
            ((ops.gemm(a, w, epilogue=ops.EPI_BIAS_ADD, bias=bias, aux=aux).float() - (ref + bias + aux.float())).abs().max() / sc).item(),
            ((ops.gemm(a, w, out_f32=True) - ref).abs().max() / sc).item()]
    wt = w.t().contiguous(); at = a.t().contiguous()
    errs.append(((ops.gemm(a, wt, b_kstrided=True).float() - ref).abs().max() / sc).item())
    errs.append(((ops.gemm(at, wt, a_kstrided=True, b_kstrided=True, out_f32=True) - ref).abs().max() / sc).item())
    worst = max(worst, max(errs))
    print(f"WIDE={os.environ.get('IA_GEMM_WIDE', 'policy')} M={M} N={N} K={K}: " + " ".join(f"{e:.1e}" for e in errs), flush=True)
print("worst", worst)
assert worst < 1.2e-2
