"""time of ia_ln_bwd's second stage and ia_colsum at the bench shapes (HIP events around single launches are too coarse: 200 launches)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
x = torch.randn((130560, 1024), device=dev).bfloat16()
out = torch.zeros(1024, device=dev)
for name, fn in [("colsum 130560 x 1024", lambda: ops.colsum(x, out=out, accumulate=True))]:
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name}: {s.elapsed_time(e) / 200 * 1e3:.1f} us per call")
