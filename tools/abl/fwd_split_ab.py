#!/usr/bin/env python3
"""ViT attention forward (L = 577): 128-query workgroups (IA_ATTN_FWD=3), 256-query workgroups (=4) and -- in the build that had it -- the
split launch (two 256-query blocks + one 128-query launch for the last 65 queries; measured slower, removed: DESIGN.md 9a,
profiles/r05_ab_attention_fwd_split.txt), one process per setting (the switch is read once), same box.
    python tools/abl/fwd_split_ab.py            -> prints the three settings for B = 512 and 256 images"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from item_alignment_amd import ops
dev = torch.device("cuda:0")
for B, L, nh in [(512, 577, 12), (256, 577, 12), (128, 300, 16)]:
    H = nh * 64
    torch.manual_seed(1)
    qkv = torch.randn((B * L, 3 * H), device=dev).to(torch.bfloat16)
    for _ in range(3):
        ops.attn_fwd(qkv, B, L, nh)
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.attn_fwd(qkv, B, L, nh)
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 10 * 1e3)
    t = sorted(best)[1]
    print(f"  B={B} L={L} nh={nh}: {t:8.1f} us  {4 * L * L * 64 * B * nh / t / 1e6:7.1f} TFLOP/s", flush=True)
''' % ROOT

for name, env in (("128-query workgroups (IA_ATTN_FWD=3)", {"IA_ATTN_FWD": "3"}), ("256-query workgroups (IA_ATTN_FWD=4)", {"IA_ATTN_FWD": "4"}),
                  ("split: 256-query blocks + one 128-query launch (default)", {}), ("by shape without the split (IA_ATTN_FWD_SPLIT=0)", {"IA_ATTN_FWD_SPLIT": "0"})):
    print(name, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), check=False)
