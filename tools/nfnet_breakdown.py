"""Where the eca_nfnet_l0 tower's forward+backward time goes, module by module (16 images @800x800, the C3 shape):
each stem conv / stage block is run alone on an input of its own shape, forward + backward, timed with events.
usage: python tools/nfnet_breakdown.py [images]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import item_alignment_amd.models as M
from item_alignment_amd.models.nfnet import FeatureMap, ScaledStdConv2d, SiluFn

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = M.create_model("eca_nfnet_l0").cuda().train()
net.ensure_arena()
net.param_arena.zero_grad()


def timed(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fwd_bwd(mod, shape):
    B, H, W, C = shape
    x = torch.randn((B * H * W, C), device=dev).bfloat16().requires_grad_(True)

    def go():
        y = mod(FeatureMap(x, B, H, W))
        y.t.backward(torch.ones_like(y.t))
    out = mod(FeatureMap(x, B, H, W))
    return timed(go), (out.B, out.H, out.W, out.t.shape[1])


total = 0.0
shape = (N, 800, 800, 8)
for name, m in net.stem.named_children():
    if isinstance(m, ScaledStdConv2d):
        t, shape2 = fwd_bwd(m, shape)
        print(f"stem.{name:6s} {shape} -> {shape2}: {t:.3f} ms", flush=True)
        shape = shape2
    else:
        act = lambda f: FeatureMap(SiluFn.apply(f.t, 1.0, False), f.B, f.H, f.W)
        t, _ = fwd_bwd(act, shape)
        print(f"stem.{name:6s} {shape}: {t:.3f} ms", flush=True)
    total += t
for si, stage in enumerate(net.stages):
    for bi, blk in enumerate(stage):
        t, shape2 = fwd_bwd(blk, shape)
        print(f"stages.{si}.{bi} {shape} -> {shape2}: {t:.3f} ms", flush=True)
        shape = shape2
        total += t
t, shape2 = fwd_bwd(net.final_conv, shape)
print(f"final_conv {shape} -> {shape2}: {t:.3f} ms")
print(f"sum {total + t:.2f} ms")
