"""Micro-benchmarks of the hot kernels on one MI355X (random data, HIP-event timing).
usage: python tools/kbench.py [gemm] [gemm_epi] [attn] [ln] [img]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from item_alignment_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def bench_gemm():
    shapes = [("qkv  NT", 16320, 3072, 1024, 0, 0), ("ffn1 NT", 16320, 4096, 1024, 0, 0), ("ffn2 NT", 16320, 1024, 4096, 0, 0),
              ("dX   NN", 16320, 1024, 4096, 0, 1), ("dX2  NN", 16320, 4096, 1024, 0, 1), ("dW   TN", 4096, 1024, 16320, 1, 1),
              ("dW2  TN", 1024, 4096, 16320, 1, 1), ("vit qkv", 36928, 2304, 768, 0, 0), ("vit fc1", 36928, 3072, 768, 0, 0),
              ("4k^3 NT", 4096, 4096, 4096, 0, 0), ("8k^3 NT", 8192, 8192, 8192, 0, 0)]
    for name, M, N, K, aks, bks in shapes:
        a = torch.randn((K, M) if aks else (M, K), device=dev).to(torch.bfloat16)
        b = torch.randn((K, N) if bks else (N, K), device=dev).to(torch.bfloat16)
        f32 = bool(aks)
        out = torch.empty((M, N), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, b, a_kstrided=bool(aks), b_kstrided=bool(bks), out=out, out_f32=f32))
        print(f"gemm {name} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s")
        if not aks and not bks:
            w = b
            t2 = timeit(lambda: torch.matmul(a, w.t()))
            print(f"     torch/hipBLASLt same shape:    {t2*1e6:8.1f} us  {2*M*N*K/t2/1e12:7.1f} TF/s")


def bench_gemm_epi():
    """the epilogue-heavy GEMMs of the RoBERTa-large layer at the bench size (128 sequences x 255 tokens)"""
    M, H, I = 32640, 1024, 4096
    x = torch.randn((M, H), device=dev).bfloat16(); w1 = (torch.randn((I, H), device=dev) * 0.03).bfloat16()
    b1 = torch.zeros(I, device=dev)
    pre = torch.empty((M, I), device=dev, dtype=torch.bfloat16); act = torch.empty_like(pre)
    t = timeit(lambda: ops.gemm(x, w1, epilogue=ops.EPI_BIAS_GELU, bias=b1, out=act, pre_out=pre))
    print(f"ffn1 bias+gelu  M={M} N={I} K={H}: {t*1e6:8.1f} us {2*M*I*H/t/1e12:7.1f} TF/s")
    t = timeit(lambda: ops.gemm(x, w1, epilogue=ops.EPI_BIAS, bias=b1, out=act))
    print(f"ffn1 bias       M={M} N={I} K={H}: {t*1e6:8.1f} us {2*M*I*H/t/1e12:7.1f} TF/s")
    dy = torch.randn((M, H), device=dev).bfloat16(); w2 = (torch.randn((H, I), device=dev) * 0.03).bfloat16()     # fc2 weight [H, I]
    dpre = torch.empty_like(pre)
    t = timeit(lambda: ops.gemm(dy, w2, b_kstrided=True, epilogue=ops.EPI_DGELU, aux=pre, out=dpre))
    print(f"dgrad x gelu'   M={M} N={I} K={H}: {t*1e6:8.1f} us {2*M*I*H/t/1e12:7.1f} TF/s")
    cs = torch.zeros(I, device=dev)
    t = timeit(lambda: ops.gemm(dy, w2, b_kstrided=True, epilogue=ops.EPI_DGELU_COLSUM, aux=pre, out=dpre, colsum_out=cs))
    print(f"dgrad x gelu' + column sums      : {t*1e6:8.1f} us {2*M*I*H/t/1e12:7.1f} TF/s")
    t = timeit(lambda: ops.colsum(dpre, cs, accumulate=True))
    print(f"separate column sums of [M,{I}]  : {t*1e6:8.1f} us {M*I*2/t/1e9:7.0f} GB/s")
    t = timeit(lambda: ops.gemm(dy, w2, b_kstrided=True, out=dpre))
    print(f"dgrad plain     M={M} N={I} K={H}: {t*1e6:8.1f} us {2*M*I*H/t/1e12:7.1f} TF/s")


def bench_attn():
    for B, L, nh in [(64, 255, 16), (32, 510, 16), (64, 577, 12)]:
        H = nh * 64
        qkv = torch.randn((B * L, 3 * H), device=dev).to(torch.bfloat16)
        mask = torch.ones((B, L), device=dev, dtype=torch.uint8)
        ctx, lse = ops.attn_fwd(qkv, B, L, nh, key_mask=mask)
        d = torch.randn_like(ctx)
        t = timeit(lambda: ops.attn_fwd(qkv, B, L, nh, key_mask=mask))
        fl = 4 * L * L * 64 * B * nh
        print(f"attn fwd B={B} L={L} nh={nh}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
        t = timeit(lambda: ops.attn_bwd(qkv, ctx, d, lse, B, L, nh, key_mask=mask))
        print(f"attn bwd B={B} L={L} nh={nh}: {t*1e6:8.1f} us {2.5*fl/t/1e12:7.1f} TF/s (algorithmic 10 L^2 d)")
        td = timeit(lambda: ops.attn_fwd(qkv, B, L, nh, key_mask=mask, drop_p=0.1, seed=1))
        print(f"attn fwd+dropout:              {td*1e6:8.1f} us")
        td = timeit(lambda: ops.attn_bwd(qkv, ctx, d, lse, B, L, nh, key_mask=mask, drop_p=0.1, seed=1))
        print(f"attn bwd+dropout:              {td*1e6:8.1f} us")


def bench_ln():
    for M, H in [(32640, 1024), (73856, 768)]:
        x = torch.randn((M, H), device=dev).to(torch.bfloat16); r = torch.randn_like(x)
        g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev)
        t = timeit(lambda: ops.ln_fwd(x, g, b, 1e-12, bias=b, residual=r))
        print(f"ln fwd M={M} H={H}: {t*1e6:7.1f} us  {M*H*2*4/t/1e9:7.0f} GB/s (x,res read; z,y write)")
        y, z, mean, rstd = ops.ln_fwd(x, g, b, 1e-12, bias=b, residual=r)
        dg = torch.zeros(H, device=dev)
        t = timeit(lambda: ops.ln_bwd(x, z, mean, rstd, g, dgamma=dg, dbeta=dg, dbias=dg))
        print(f"ln bwd M={M} H={H}: {t*1e6:7.1f} us  {M*H*2*3/t/1e9:7.0f} GB/s (dy,z read; dz write)")


def bench_img():
    from item_alignment_amd.data.gpu_preproc import GpuImagePipeline
    for B, H, S in [(128, 800, 384), (16, 800, 800)]:
        frames = torch.randint(0, 256, (B, H, H, 3), dtype=torch.uint8, device=dev)
        pipe = GpuImagePipeline(S, dev)
        t = timeit(lambda: pipe(frames))
        print(f"image pipeline {B} x {H}x{H} uint8 -> {S}x{S} fp32 normalised: {t*1e6:8.1f} us  {B/t:9.0f} images/s  "
              f"{(B*H*H*3 + B*S*S*3*4)/t/1e9:6.0f} GB/s (in + out)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "ln"]
    if "gemm" in which: bench_gemm()
    if "gemm_epi" in which: bench_gemm_epi()
    if "attn" in which: bench_attn()
    if "ln" in which: bench_ln()
    if "img" in which: bench_img()
