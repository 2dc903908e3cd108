import sys, os
sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
from types import SimpleNamespace
from oracle.weights import seeded_state_dict
from item_alignment_amd.models.resnetv2 import PreActBottleneck
from item_alignment_amd.models.nfnet import FeatureMap
from item_alignment_amd.models.base import HipModule

def rms(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

class Wrap(HipModule):
    def __init__(self, blk):
        super().__init__(); self.blk = blk
cin = cout = 256; B = 4; H = 4; mid = 64
spec = [("b.norm1.weight", (cin,)), ("b.norm1.bias", (cin,)), ("b.conv1.weight", (mid, cin, 1, 1)), ("b.norm2.weight", (mid,)),
        ("b.norm2.bias", (mid,)), ("b.conv2.weight", (mid, mid, 3, 3)), ("b.norm3.weight", (mid,)), ("b.norm3.bias", (mid,)),
        ("b.conv3.weight", (cout, mid, 1, 1))]
sd = seeded_state_dict(spec, 5, scale=0.08)
g = torch.Generator().manual_seed(1)
x = (torch.randn((B, cin, H, H), generator=g) * 2 + 0.5).bfloat16().float()
dy = torch.randn((B, cout, H, H), generator=g).bfloat16().float()
rsd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
xr = x.clone().requires_grad_(True)
keep = {}
def bn(t, p):
    return F.relu(F.batch_norm(t, None, None, rsd[p + ".weight"], rsd[p + ".bias"], True, 0.1, 1e-5))
def tag(t, name):
    t.retain_grad(); keep[name] = t; return t
pre = tag(bn(xr, "b.norm1"), "pre")
c1 = tag(F.conv2d(pre, rsd["b.conv1.weight"]), "c1")
n2 = tag(bn(c1, "b.norm2"), "n2")
c2 = tag(F.conv2d(n2, rsd["b.conv2.weight"], padding=1), "c2")
n3 = tag(bn(c2, "b.norm3"), "n3")
c3 = tag(F.conv2d(n3, rsd["b.conv3.weight"]), "c3")
(c3 + xr).backward(dy)

m = Wrap(PreActBottleneck(cin, cout, 0.25, 1, False))
m.blk.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=False)
m = m.cuda().train(); m.ensure_arena(); m.param_arena.zero_grad()
blk = m.blk
xh = x.permute(0, 2, 3, 1).reshape(-1, cin).cuda().bfloat16().requires_grad_(True)
hk = {}
def htag(f, name):
    f.t.retain_grad(); hk[name] = f.t; return f
f = FeatureMap(xh, B, H, H)
hpre, sc = blk.norm1(f, 1, passthrough=True)
htag(hpre, "pre")
h = htag(blk.conv1(hpre), "c1")
h = htag(blk.norm2(h, 1), "n2")
h = htag(blk.conv2(h), "c2")
h = htag(blk.norm3(h, 1), "n3")
h = blk.conv3(h, residual=sc)
h.t.backward(dy.permute(0, 2, 3, 1).reshape(-1, cout).cuda().bfloat16())
def nchw(t, C): return t.view(B, H, H, C).permute(0, 3, 1, 2)
for name, C in [("n3", mid), ("c2", mid), ("n2", mid), ("c1", mid), ("pre", cin)]:
    print(f"{name}: value rms {rms(nchw(hk[name], C), keep[name]):.4f}  grad rms {rms(nchw(hk[name].grad, C), keep[name].grad):.4f}")
# mask agreement at n3
print("n3 mask mismatch frac", ((nchw(hk['n3'], mid).float().cpu() > 0) != (keep['n3'] > 0)).float().mean().item())
print("n2 mask mismatch frac", ((nchw(hk['n2'], mid).float().cpu() > 0) != (keep['n2'] > 0)).float().mean().item())
