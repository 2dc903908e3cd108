"""Runs the attention kernels on one shape a few times (for rocprofv3 PMC collection). usage: attn_one.py B L nh [drop]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from item_alignment_amd import ops
B, L, nh = map(int, sys.argv[1:4]); drop = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
dev = torch.device("cuda:0"); H = nh * 64
qkv = torch.randn((B * L, 3 * H), device=dev).to(torch.bfloat16)
mask = torch.ones((B, L), device=dev, dtype=torch.uint8)
for _ in range(3):
    ctx, lse = ops.attn_fwd(qkv, B, L, nh, key_mask=mask, drop_p=drop, seed=1)
    d = torch.randn_like(ctx)
    ops.attn_bwd(qkv, ctx, d, lse, B, L, nh, key_mask=mask, drop_p=drop, seed=1)
torch.cuda.synchronize()
