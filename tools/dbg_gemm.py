import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in [(4096, 1024, 4096), (1024, 1024, 4096), (2048, 1024, 8192), (512, 512, 2048)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    ref = a.float() @ w.float().t()
    out = ops.gemm(a, w, out_f32=True)
    err = (out - ref).abs()
    bad = (err > 1e-2 * ref.abs().max()).nonzero()
    print(M, N, K, "max err", err.max().item(), "bad", bad.shape[0], bad[:5].tolist(), bad[-3:].tolist() if bad.shape[0] else "")
    if bad.shape[0]:
        rows = torch.unique(bad[:, 0]); cols = torch.unique(bad[:, 1])
        print("  rows", rows[:10].tolist(), len(rows), "cols", cols[:10].tolist(), len(cols))
