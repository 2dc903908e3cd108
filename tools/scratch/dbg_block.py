import sys, os
sys.path.insert(0, os.getcwd())
import torch
from types import SimpleNamespace
from oracle import ref_models as O
from oracle.weights import seeded_state_dict
from item_alignment_amd.models.resnetv2 import PreActBottleneck, BatchNormAct2d
from item_alignment_amd.models.nfnet import FeatureMap
from item_alignment_amd.models.base import HipModule

import torch.nn.functional as TF
class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return x.bfloat16().float()
    @staticmethod
    def backward(ctx, g): return g
class FR:
    """torch.nn.functional with conv outputs rounded to bf16 (what the HIP tower stores), straight-through gradient"""
    def __getattr__(self, k): return getattr(TF, k)
    def conv2d(self, *a, **kw): return _RoundSTE.apply(TF.conv2d(*a, **kw))
if os.environ.get("ROUND"): O.F = FR()

def rms(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

class Wrap(HipModule):
    def __init__(self, blk):
        super().__init__(); self.blk = blk
cfg = SimpleNamespace(eps=1e-5, momentum=0.1)
for (cin, cout, stride, ds, B, H) in [(256, 256, 1, False, 4, 4), (128, 256, 2, True, 4, 8), (64, 64, 1, True, 4, 32), (256, 256, 1, False, 4, 16)]:
    blkd = dict(in_chs=cin, out_chs=cout, mid_chs=cout // 4, stride=stride, downsample=ds)
    spec = []
    if ds: spec.append(("b.downsample.conv.weight", (cout, cin, 1, 1)))
    spec += [("b.norm1.weight", (cin,)), ("b.norm1.bias", (cin,)), ("b.conv1.weight", (cout // 4, cin, 1, 1)), ("b.norm2.weight", (cout // 4,)),
             ("b.norm2.bias", (cout // 4,)), ("b.conv2.weight", (cout // 4, cout // 4, 3, 3)), ("b.norm3.weight", (cout // 4,)), ("b.norm3.bias", (cout // 4,)),
             ("b.conv3.weight", (cout, cout // 4, 1, 1))]
    sd = seeded_state_dict(spec, 5, scale=0.08)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn((B, cin, H, H), generator=g) * 2 + 0.5).bfloat16().float()
    Ho = (H - 1) // stride + 1
    dy = torch.randn((B, cout, Ho, Ho), generator=g).bfloat16().float()
    rsd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref = O.preact_bottleneck(xr, rsd, "b", blkd, cfg, True, None)
    ref.backward(dy)
    m = Wrap(PreActBottleneck(cin, cout, 0.25, stride, ds))
    m.blk.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=False)
    m = m.cuda().train(); m.ensure_arena(); m.param_arena.zero_grad()
    xh = x.permute(0, 2, 3, 1).reshape(-1, cin).cuda().bfloat16().requires_grad_(True)
    out = m.blk(FeatureMap(xh, B, H, H))
    out.t.backward(dy.permute(0, 2, 3, 1).reshape(-1, cout).cuda().bfloat16())
    print(f"block {cin}->{cout} s{stride} ds={ds} H={H}: fwd rms {rms(out.t.view(B, Ho, Ho, cout).permute(0, 3, 1, 2), ref):.4f}  dx rms "
          f"{rms(xh.grad.view(B, H, H, cin).permute(0, 3, 1, 2), xr.grad):.4f}")
    for k, p in m.blk.named_parameters():
        print(f"   {k:28s} rms {rms(p.grad, rsd['b.' + k].grad):.4f}")
