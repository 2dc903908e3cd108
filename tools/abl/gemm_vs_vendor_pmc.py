"""One shape, our plain NT GEMM and the vendor library's, a few launches each: run under rocprofv3 --pmc to compare where the cycles go."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (65280, 4096, 2048)))
a = torch.randn((M, K), device=dev).bfloat16(); b = torch.randn((N, K), device=dev).bfloat16()
out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
for _ in range(6):
    ops.gemm(a, b, out=out)
torch.cuda.synchronize()
for _ in range(6):
    torch.matmul(a, b.t(), out=out)
torch.cuda.synchronize()
