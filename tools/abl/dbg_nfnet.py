"""eca_nfnet_l0 stage by stage: HIP tower vs the fp32 oracle at a few image sizes (which op drifts at 800 x 800?)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch.nn.functional as F
import item_alignment_amd.models as M
from item_alignment_amd.models import nfnet as NF
from oracle import ref_models as O
from types import SimpleNamespace
sizes = [int(a) for a in sys.argv[1:]] or [256, 800]
cfg = SimpleNamespace(num_labels=2, hidden_dropout_prob=0.0, loss_type="ce", loss_margin=0.0, classification_method="cls", hidden_size=2304)
torch.manual_seed(5)
model = M.NFNetTwoTower(cfg, M.create_model("eca_nfnet_l0"))
with torch.no_grad():
    for k, v in model.named_parameters():
        if k.endswith("conv3.gain"): v.fill_(1.0)
sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
model = model.cuda().eval()
enc = model.img_encoder
ncfg = O.nfnet_cfg("eca_nfnet_l0")
for S in sizes:
    g = torch.Generator().manual_seed(3)
    im = torch.randn((1, 3, S, S), generator=g)
    caps = []
    hooks = []
    def cap(name):
        def h(mod, inp, out):
            if isinstance(out, NF.FeatureMap):
                caps.append((name, out.t.detach().float().cpu().reshape(out.B, out.H, out.W, -1).permute(0, 3, 1, 2)))
        return h
    for si, st in enumerate(enc.stages):
        for bi, blk in enumerate(st):
            hooks.append(blk.register_forward_hook(cap(f"stage{si}.{bi}")))
    with torch.no_grad():
        f = enc.forward_features(im.cuda())
        caps.append(("final", f.t.float().cpu().reshape(f.B, f.H, f.W, -1).permute(0, 3, 1, 2)))
    for h in hooks: h.remove()
    # oracle, same capture points
    ref = []
    with torch.no_grad():
        x = im
        for i, s in enumerate((2, 1, 1, 2)):
            x = O.scaled_std_conv(x, sd, f"img_encoder.stem.conv{i + 1}", stride=s, eps=ncfg.eps)
            if i != 3: x = F.silu(x)
        ref.append(("stem", x))
        for si, blocks in enumerate(O.nfnet_plan(ncfg)):
            for bi, blk in enumerate(blocks):
                x = O.nf_block(x, sd, f"img_encoder.stages.{si}.{bi}", blk, ncfg)
                ref.append((f"stage{si}.{bi}", x))
        x = F.silu(O.scaled_std_conv(x, sd, "img_encoder.final_conv", eps=ncfg.eps))
        ref.append(("final", x))
    refd = dict(ref)
    print(f"== S={S}")
    for name, got in caps:
        want = refd[name]
        if got.shape != want.shape:
            print(f"   {name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"); continue
        err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-9)
        rms_g, rms_w = got.pow(2).mean().sqrt().item(), want.pow(2).mean().sqrt().item()
        mg, mw = got.mean((2, 3)), want.mean((2, 3))
        merr = (mg - mw).abs().max().item() / (mw.abs().max().item() + 1e-9)
        # border vs interior
        b_err = (got - want)[..., 0, :].abs().max().item() / (want.abs().max().item() + 1e-9)
        print(f"   {name:10s} {tuple(got.shape)}: max err {err:.3e}  rms {rms_g:.4f} / {rms_w:.4f}  channel-mean err {merr:.3e}  top-row err {b_err:.3e}")
