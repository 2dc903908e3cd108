"""Runs one GEMM shape a few times (for rocprofv3 PMC collection).  usage: gemm_one.py M N K [aks bks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from item_alignment_amd import ops
M, N, K = map(int, sys.argv[1:4])
aks, bks = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (0, 0)
dev = torch.device("cuda:0")
a = torch.randn((K, M) if aks else (M, K), device=dev).to(torch.bfloat16)
b = torch.randn((K, N) if bks else (N, K), device=dev).to(torch.bfloat16)
out = torch.empty((M, N), device=dev, dtype=torch.float32 if aks else torch.bfloat16)
for _ in range(5):
    ops.gemm(a, b, a_kstrided=bool(aks), b_kstrided=bool(bks), out=out, out_f32=bool(aks))
torch.cuda.synchronize()
