import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from types import SimpleNamespace
from oracle import ref_models as O
from oracle.weights import seeded_state_dict
from item_alignment_amd.models.resnetv2 import ResNetV2
rcfg = SimpleNamespace(layers=(1, 2, 1, 1), channels=(64, 128, 256, 256), stem_chs=32, bottle_ratio=0.25, eps=1e-5, momentum=0.1, num_features=256)
sd = seeded_state_dict(O.resnetv2_state_spec(rcfg, prefix="e"), 31, scale=0.08)
g = torch.Generator().manual_seed(6)
images = torch.randn((4, 3, 128, 128), generator=g)
wts = torch.randn((4, 256), generator=g)
ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
stats = O.resnetv2_running_stats(rcfg, "e")
ref = O.resnetv2_forward_features(ref_sd, "e", rcfg, images, True, stats).mean((2, 3))
(ref * wts).sum().backward()
net = ResNetV2(rcfg.layers, rcfg.channels, stem_chs=rcfg.stem_chs)
net.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=False)
net = net.cuda().train()
out = net(images.cuda())
net.param_arena.zero_grad()
(out * wts.cuda()).sum().backward()
torch.cuda.synchronize()
for k, p in net.named_parameters():
    if "head" in k: continue
    got, want = p.grad.float().cpu().flatten(), ref_sd["e." + k].grad.flatten()
    c = (torch.dot(got, want) / (got.norm() * want.norm() + 1e-30)).item()
    print(f"{k:50s} cos {c:.4f}  norm ratio {got.norm() / want.norm():.3f}")
