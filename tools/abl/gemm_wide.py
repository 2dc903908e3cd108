"""A/B of the two 256x256 GEMM kernels (IA_GEMM_WIDE=0/1 is read once per process: run twice): correctness vs torch + rates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12)).item()


print("IA_GEMM_WIDE =", os.environ.get("IA_GEMM_WIDE", "0"))
torch.manual_seed(0)
for (M, N, K) in [(512, 512, 256), (1000, 776, 200), (2048, 1024, 4096 + 64)]:
    a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    ref = a.float() @ w.float().t()
    bias = torch.randn(N, device=dev); aux = torch.randn((M, N), device=dev).bfloat16()
    e = [rel(ops.gemm(a, w), ref), rel(ops.gemm(a, w, epilogue=ops.EPI_BIAS, bias=bias), ref + bias),
         rel(ops.gemm(a, w, epilogue=ops.EPI_BIAS_ADD, bias=bias, aux=aux), ref + bias + aux.float()),
         rel(ops.gemm(a, w, out_f32=True), ref)]
    act, der = ops.gemm(a, w, epilogue=ops.EPI_BIAS_GELU, bias=bias)
    e.append(rel(act, torch.nn.functional.gelu(ref + bias)))
    wt = w.t().contiguous()                    # [K, N]
    e.append(rel(ops.gemm(a, wt, b_kstrided=True), ref))
    e.append(rel(ops.gemm(a, wt, b_kstrided=True, epilogue=ops.EPI_DGELU, aux=aux), ref * aux.float()))
    cs = torch.zeros(N, device=dev)
    e.append(rel(ops.gemm(a, wt, b_kstrided=True, epilogue=ops.EPI_DGELU_COLSUM, aux=aux, colsum_out=cs), ref * aux.float()))
    e.append(rel(cs, (ref * aux.float()).sum(0)))
    at = a.t().contiguous()                    # [K, M]
    e.append(rel(ops.gemm(at, wt, a_kstrided=True, b_kstrided=True, out_f32=True), ref))
    print(f"M={M} N={N} K={K} rel errors:", " ".join(f"{x:.1e}" for x in e), flush=True)

shapes = [("qkv  NT", 16320, 3072, 1024, 0, 0), ("ffn1 NT", 16320, 4096, 1024, 0, 0), ("ffn2 NT", 16320, 1024, 4096, 0, 0),
          ("dX   NN", 16320, 1024, 4096, 0, 1), ("dX2  NN", 16320, 4096, 1024, 0, 1), ("dW   TN", 4096, 1024, 16320, 1, 1),
          ("dW2  TN", 1024, 4096, 16320, 1, 1), ("vit fc1", 36928, 3072, 768, 0, 0), ("big NT", 65280, 4096, 1024, 0, 0),
          ("4k^3 NT", 4096, 4096, 4096, 0, 0), ("8k^3 NT", 8192, 8192, 8192, 0, 0), ("8k^3 NN", 8192, 8192, 8192, 0, 1), ("8k^3 TN", 8192, 8192, 8192, 1, 1)]
for name, M, N, K, aks, bks in shapes:
    a = torch.randn((K, M) if aks else (M, K), device=dev).bfloat16()
    b = torch.randn((K, N) if bks else (N, K), device=dev).bfloat16()
    f32 = bool(aks)
    out = torch.empty((M, N), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, b, a_kstrided=bool(aks), b_kstrided=bool(bks), out=out, out_f32=f32))
    print(f"gemm {name} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
