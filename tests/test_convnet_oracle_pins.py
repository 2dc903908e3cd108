"""Pins for the conv-tower part of the CPU oracle.  timm 0.6.5 is the reference's dependency for these
towers and is not installed, so the oracle's NFNet / ResNetV2 functions (oracle/ref_models.py) are restatements; here they are checked
by something other than themselves:
  * tests/golden/nfnet_reference_assembly.npz — the REFERENCE's in-tree NormFreeNet assembly (src/models/image.py:40-199) run in the
    build container over an independently written nn.Module restatement of the timm blocks (oracle/timm_blocks.py);
  * closed forms on torch primitives: ScaledStdConv2d = F.conv2d with (w - mu)/sqrt(var + eps) * gain * gamma * fan_in^-0.5 via
    F.batch_norm, ECA = GAP -> nn.Conv1d -> sigmoid, BatchNormAct2d = nn.BatchNorm2d + ReLU (batch statistics, running buffers,
    eval mode), pre-activation bottleneck / whole ResNetV2 as nn.Modules — outputs AND gradients."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_models as O
from oracle import timm_blocks as TB
from oracle.weights import seeded_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nfnet_reference_assembly.npz")


def close(a, b, tol=2e-5):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item() < tol


@pytest.mark.parametrize("name", ["eca_nfnet_l0", "eca_nfnet_l1"])
def test_nfnet_oracle_matches_the_reference_assembly(name):
    z = np.load(GOLD)
    meta = json.loads(bytes(z["meta"]).decode())["cases"][name]
    cfg = O.nfnet_cfg(name)
    spec = [(k, tuple(s)) for k, s in meta["spec"]]
    # the reference module's state_dict keys / shapes are exactly what the oracle (and the HIP model) expect
    assert sorted(spec) == sorted((k, tuple(s)) for k, s in O.nfnet_state_spec(cfg))
    plan = O.nfnet_plan(cfg)
    assert [[b["stride"] for b in st] for st in plan] == meta["strides"]
    assert [[b["groups"] for b in st] for st in plan] == meta["groups"]
    for got, want in zip(plan, meta["betas"]):
        assert np.allclose([b["beta"] for b in got], want, rtol=1e-12)
    sd = seeded_state_dict(spec, meta["seed"])
    x = torch.from_numpy(z[f"{name}_in"])
    with torch.no_grad():
        feats = O.nfnet_forward_features(sd, "img_encoder", cfg, x)
    assert feats.shape == tuple(z[f"{name}_features"].shape)
    assert close(feats, torch.from_numpy(z[f"{name}_features"]), 1e-4)
    # stem alone (the four strided / unstrided ScaledStdConv2d + SiLU)
    with torch.no_grad():
        s = x
        for i, st in enumerate((2, 1, 1, 2)):
            s = O.scaled_std_conv(s, sd, f"img_encoder.stem.conv{i + 1}", stride=st, eps=cfg.eps)
            if i != 3:
                s = F.silu(s)
    assert close(s, torch.from_numpy(z[f"{name}_stem"]), 1e-5)


@pytest.mark.parametrize("cin,cout,k,stride,groups", [(8, 16, 3, 1, 1), (64, 64, 3, 2, 1), (128, 128, 3, 1, 2), (24, 40, 1, 1, 1), (3, 16, 3, 2, 1)])
def test_scaled_std_conv_against_the_batch_norm_form(cin, cout, k, stride, groups):
    g = torch.Generator().manual_seed(cin * 7 + cout)
    m = TB.ScaledStdConv2d(cin, cout, k, stride=stride, groups=groups, gamma=O.NONLIN_GAMMA_SILU, eps=1e-5)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.3 + 0.05)
        m.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        m.gain.copy_(1 + 0.2 * torch.randn(m.gain.shape, generator=g))
    sd = {"c.weight": m.weight.detach().clone().requires_grad_(True), "c.bias": m.bias.detach().clone().requires_grad_(True),
          "c.gain": m.gain.detach().clone().requires_grad_(True)}
    x = torch.randn((2, cin, 13, 11), generator=g)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y_ref = m(x1)
    y = O.scaled_std_conv(x2, sd, "c", stride=stride, groups=groups, eps=1e-5)
    assert y.shape == y_ref.shape and close(y, y_ref)
    w = torch.randn(y.shape, generator=g)
    (y_ref * w).sum().backward()
    (y * w).sum().backward()
    assert close(x2.grad, x1.grad) and close(sd["c.weight"].grad, m.weight.grad, 1e-4)
    assert close(sd["c.gain"].grad, m.gain.grad, 1e-4) and close(sd["c.bias"].grad, m.bias.grad)


@pytest.mark.parametrize("channels", [256, 512, 1536])
def test_eca_against_conv1d_module(channels):
    g = torch.Generator().manual_seed(channels)
    m = TB.EcaModule(channels)
    assert m.conv.kernel_size[0] == O.eca_kernel_size(channels)
    x = torch.randn((3, channels, 5, 4), generator=g)
    sd = {"a.conv.weight": m.conv.weight.detach().clone()}
    assert close(O.eca(x, sd, "a"), m(x))


def test_nf_block_against_module_with_gradients():
    cfg = O.nfnet_cfg("eca_nfnet_l0")
    plan = O.nfnet_plan(cfg)
    for si, bi in ((0, 0), (1, 0), (1, 1)):
        blk = plan[si][bi]
        m = TB.NormFreeBlock(in_chs=blk["in_chs"], out_chs=blk["out_chs"], stride=blk["stride"], alpha=cfg.alpha, beta=blk["beta"],
                             bottle_ratio=cfg.bottle_ratio, group_size=cfg.group_size, ch_div=cfg.ch_div, reg=False, extra_conv=True,
                             attn_layer=TB.EcaModule, attn_gain=cfg.attn_gain, act_layer=lambda inplace=False: torch.nn.SiLU(),
                             conv_layer=lambda *a, **k: TB.ScaledStdConv2d(*a, gamma=O.NONLIN_GAMMA_SILU, eps=cfg.eps, **k))
        spec = [("b." + k, tuple(v.shape)) for k, v in m.state_dict().items()]
        sd = seeded_state_dict(spec, 100 + si * 10 + bi)
        m.load_state_dict({k[2:]: v for k, v in sd.items()})
        for v in sd.values():
            v.requires_grad_(True)
        x = torch.randn((2, blk["in_chs"], 9, 9), generator=torch.Generator().manual_seed(3))
        x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y_ref, y = m(x1), O.nf_block(x2, sd, "b", blk, cfg)
        assert close(y, y_ref, 1e-5)
        y_ref.square().sum().backward()
        y.square().sum().backward()
        assert close(x2.grad, x1.grad, 1e-4)
        for k, p in m.named_parameters():
            assert close(sd["b." + k].grad, p.grad, 2e-4), k


def test_bn_act_against_batchnorm2d_module():
    cfg = SimpleNamespace(momentum=0.1, eps=1e-5)
    g = torch.Generator().manual_seed(9)
    m = TB.BatchNormAct2d(24)
    with torch.no_grad():
        m.bn.weight.copy_(1 + 0.3 * torch.randn(24, generator=g))
        m.bn.bias.copy_(0.2 * torch.randn(24, generator=g))
    sd = {"n.weight": m.bn.weight.detach().clone(), "n.bias": m.bn.bias.detach().clone()}
    stats = {"n.running_mean": torch.zeros(24), "n.running_var": torch.ones(24)}
    for step in range(3):                                   # training: batch statistics + running-buffer updates (unbiased variance)
        x = torch.randn((4, 24, 6, 5), generator=g) * (1 + step) + 0.3 * step
        assert close(O.bn_act(x, sd, "n", cfg, True, stats), m(x))
        assert close(stats["n.running_mean"], m.bn.running_mean) and close(stats["n.running_var"], m.bn.running_var)
    m.eval()
    x = torch.randn((2, 24, 3, 3), generator=g)
    assert close(O.bn_act(x, sd, "n", cfg, False, stats), m(x))


def test_resnetv2_oracle_against_module_restatement():
    cfg = SimpleNamespace(layers=(1, 2, 1, 1), channels=(64, 128, 256, 256), stem_chs=32, bottle_ratio=0.25, eps=1e-5, momentum=0.1, num_features=256)
    sd = seeded_state_dict(O.resnetv2_state_spec(cfg), 17)
    m = TB.ResNetV2(cfg.layers, cfg.channels, cfg.stem_chs)
    m.load_timm_state(sd)
    m.train()
    x = torch.randn((3, 3, 64, 64), generator=torch.Generator().manual_seed(2))
    stats = O.resnetv2_running_stats(cfg)
    keys = ["img_encoder.stem.conv.weight", "img_encoder.stages.1.blocks.0.downsample.conv.weight", "img_encoder.stages.3.blocks.0.conv2.weight",
            "img_encoder.stages.2.blocks.0.norm2.weight"]
    for k in keys:
        sd[k].requires_grad_(True)
    y = O.resnetv2_forward_features(sd, "img_encoder", cfg, x, True, stats)
    y_ref = m.forward_features(x)
    assert close(y, y_ref, 1e-4)
    y.mean((2, 3)).square().sum().backward()
    y_ref.mean((2, 3)).square().sum().backward()
    mods = {keys[0]: m.stem_conv.weight, keys[1]: m.stages[1][0].downsample.weight, keys[2]: m.stages[3][0].conv2.weight,
            keys[3]: m.stages[2][0].norm2.bn.weight}
    for k in keys:
        assert close(sd[k].grad, mods[k].grad, 1e-3), k
    # running statistics of the last norm and of a mid-net norm after one training pass
    assert close(stats["img_encoder.norm.running_var"], m.norm.bn.running_var, 1e-4)
    assert close(stats["img_encoder.stages.1.blocks.1.norm1.running_mean"], m.stages[1][1].norm1.bn.running_mean, 1e-4)
    # and the real resnetv2_50 plan: parameter count as timm lists it (25.55 M incl. the 1000-class fc of 2.049 M -> 23.50 M here)
    full = O.resnetv2_cfg("resnetv2_50")
    n = sum(int(np.prod(s)) for _, s in O.resnetv2_state_spec(full))
    m50 = TB.ResNetV2(full.layers, full.channels, full.stem_chs)
    assert n == sum(p.numel() for p in m50.parameters())
