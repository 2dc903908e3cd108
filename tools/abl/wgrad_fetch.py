"""weight-gradient GEMM of the bench step (fc1: [130560 x 4096]^T [130560 x 1024]) a few launches: run under rocprofv3 --pmc FETCH_SIZE"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
for M, N, K in [(4096, 1024, 130560), (1024, 1024, 130560), (3072, 768, 295424)]:
    a = torch.randn((K, M), device=dev).bfloat16(); b = torch.randn((K, N), device=dev).bfloat16()
    out = torch.zeros((M, N), device=dev, dtype=torch.float32)
    for _ in range(4):
        ops.gemm(a, b, a_kstrided=True, b_kstrided=True, out=out, out_f32=True, accumulate=True)
    torch.cuda.synchronize()
