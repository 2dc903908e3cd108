"""Which library kernels torch (hipBLASLt / rocBLAS) picks for the step's bf16 GEMM shapes -- run under rocprofv3 --kernel-trace --stats;
used only to learn the tile geometry the vendor library runs these shapes with (tools/abl: measurements, not product code)."""
import torch

shapes = {"qkv": (16320, 3072, 1024), "ffn1": (16320, 4096, 1024), "ffn2": (16320, 1024, 4096), "4k": (4096, 4096, 4096), "8k": (8192, 8192, 8192),
          "ffn2_full": (130560, 1024, 4096)}
for name, (M, N, K) in shapes.items():
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"{name} M={M} N={N} K={K}: {us:.1f} us {2 * M * N * K / us / 1e6:.1f} TF/s", flush=True)
