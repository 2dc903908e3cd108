"""Runs the streaming and the LDS-resident attention forward (and backward when built) on one shape a few times (rocprofv3 PMC runs).
usage: attn_one_res.py B L nh [drop] [fwd|all]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from item_alignment_amd import ops
import attn_dev as A
B, L, nh = map(int, sys.argv[1:4]); drop = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
what = sys.argv[5] if len(sys.argv) > 5 else "fwd"
H = nh * 64
qkv = torch.randn((B * L, 3 * H), device=A.dev).to(torch.bfloat16)
qs = A.prescale(qkv, H)
for _ in range(3):
    ctx, lse = ops.attn_fwd(qkv, B, L, nh, drop_p=drop, seed=1)
    ctx2, nlse = A.fwd_res(qs, B, L, nh, None, drop_p=drop, seed=1)
    if what == "all":
        d = torch.randn_like(ctx)
        ops.attn_bwd(qkv, ctx, d, lse, B, L, nh, drop_p=drop, seed=1)
        A.bwd_res(qs, ctx2, d, nlse, B, L, nh, None, drop_p=drop, seed=1)
torch.cuda.synchronize()
