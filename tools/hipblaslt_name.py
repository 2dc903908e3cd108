"""Which hipBLASLt (Tensile) kernels serve the bench GEMM shapes: run under `rocprofv3 --kernel-trace --stats` and read the
kernel names (they spell the macro-tile / wave-group / prefetch configuration).  python tools/hipblaslt_name.py"""
import torch
dev = torch.device("cuda:0")
for M, N, K in [(16320, 3072, 1024), (16320, 1024, 4096), (8192, 8192, 8192)]:
    a = torch.randn((M, K), device=dev).bfloat16(); w = torch.randn((N, K), device=dev).bfloat16()
    for _ in range(5):
        c = torch.matmul(a, w.t())
    torch.cuda.synchronize()
