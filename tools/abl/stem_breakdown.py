"""eca_nfnet_l0 stem convs and blocks, forward + backward WITHOUT the input gradient of conv1 (the image needs none), 32 images @800"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import item_alignment_amd.models as M
from item_alignment_amd.models.nfnet import FeatureMap
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = M.create_model("eca_nfnet_l0").cuda().train()
net.ensure_arena()

def timed(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

x = torch.randn((N * 800 * 800, 8), device=dev).bfloat16()
m = net.stem.conv1
def fwd():
    return m(FeatureMap(x, N, 800, 800))
def fb():
    y = fwd(); y.t.backward(torch.ones_like(y.t))
print(f"conv1 fwd only {timed(fwd):.3f} ms; fwd + weight gradient {timed(fb):.3f} ms")
imgs = torch.randn((N, 3, 800, 800), device=dev)
def whole():
    f = net.forward_features(imgs); f.t.backward(torch.ones_like(f.t))
print(f"whole tower fwd+bwd, {N} images: {timed(whole, 3):.2f} ms")
