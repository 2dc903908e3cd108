"""TEST INFRASTRUCTURE — a SECOND, independently written restatement of the timm 0.6.5 building blocks the reference's image towers
are assembled from (timm is pinned by the reference's requirements.txt and absent offline), as torch.nn MODULES built on different
torch primitives than oracle/ref_models.py's functional forms, so that the two can check each other:

  ScaledStdConv2d   weight standardisation through F.batch_norm on the [1, Cout, fan_in] view (how timm's layers/std_conv.py does it),
                    where ref_models.scaled_std_conv uses explicit mean / var arithmetic
  EcaModule         nn.Conv1d on the pooled [B, 1, C] view + nn.Sigmoid (layers/eca.py)
  DownsampleAvg / NormFreeBlock / create_stem / NfCfg   (models/nfnet.py)
  BatchNormAct2d    nn.BatchNorm2d + nn.ReLU modules with their own running buffers (layers/norm_act.py)
  PreActBottleneck / ResNetV2   nn.Conv2d / nn.MaxPool2d modules (models/resnetv2.py, `resnetv2_50` family)

oracle/gen_golden_convnets.py plugs the NFNet blocks into the REFERENCE's own in-tree NormFreeNet assembly (src/models/image.py:40-199:
stage / stride / beta / expected-variance bookkeeping) to capture tests/golden/nfnet_reference_assembly.npz.  Only tests/ and that
generator import this file; nothing in the product does."""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

_nonlin_gamma = dict(identity=1.0, celu=1.270926833152771, elu=1.2716004848480225, gelu=1.7015043497085571, leaky_relu=1.70590341091156,
                     log_sigmoid=1.9193484783172607, log_softmax=1.0002083778381348, relu=1.7139588594436646, relu6=1.7131484746932983,
                     selu=1.0008515119552612, sigmoid=4.803835391998291, silu=1.7881293296813965, softsign=2.338853120803833,
                     softplus=1.9203323125839233, tanh=1.5939117670059204)


@dataclass
class NfCfg:
    depths: Tuple[int, int, int, int]
    channels: Tuple[int, int, int, int]
    alpha: float = 0.2
    stem_type: str = "3x3"
    stem_chs: Optional[int] = None
    group_size: Optional[int] = None
    attn_layer: Optional[str] = None
    attn_kwargs: dict = None
    attn_gain: float = 2.0
    width_factor: float = 1.0
    bottle_ratio: float = 0.5
    num_features: int = 0
    ch_div: int = 8
    reg: bool = False
    extra_conv: bool = False
    gamma_in_act: bool = False
    same_padding: bool = False
    std_conv_eps: float = 1e-5
    skipinit: bool = False
    zero_init_fc: bool = False
    act_layer: str = "silu"


def eca_nfnet_cfg(name):
    """models/nfnet.py `_nfnet_cfg(...)` rows for the eca_nfnet_l* family"""
    depths, feat_mult = {"eca_nfnet_l0": ((1, 2, 6, 3), 1.5), "eca_nfnet_l1": ((2, 4, 12, 6), 2.0), "eca_nfnet_l2": ((3, 6, 18, 9), 2.0)}[name]
    channels = (256, 512, 1536, 1536)
    return NfCfg(depths=depths, channels=channels, stem_type="deep_quad", stem_chs=128, group_size=64, bottle_ratio=0.25, extra_conv=True,
                 num_features=int(channels[-1] * feat_mult), act_layer="silu", attn_layer="eca", attn_kwargs=dict())


def make_divisible(v, divisor=8, min_value=None, round_limit=.9):
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def get_padding(kernel_size, stride=1, dilation=1):
    return ((stride - 1) + dilation * (kernel_size - 1)) // 2


class ScaledStdConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=None, dilation=1, groups=1, bias=True, gamma=1.0,
                 eps=1e-6, gain_init=1.0):
        if padding is None:
            padding = get_padding(kernel_size, stride, dilation)
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.gain = nn.Parameter(torch.full((self.out_channels, 1, 1, 1), gain_init))
        self.scale = gamma * self.weight[0].numel() ** -0.5
        self.eps = eps

    def forward(self, x):
        weight = F.batch_norm(self.weight.reshape(1, self.out_channels, -1), None, None, weight=(self.gain * self.scale).view(-1),
                              training=True, momentum=0., eps=self.eps).reshape_as(self.weight)
        return F.conv2d(x, weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class ScaledStdConv2dSame(ScaledStdConv2d):          # only named by the reference's import line; cfg.same_padding is False for eca_nfnet
    pass


class EcaModule(nn.Module):
    def __init__(self, channels=None, kernel_size=3, gamma=2, beta=1):
        super().__init__()
        if channels is not None:
            t = int(abs(math.log(channels, 2) + beta) / gamma)
            kernel_size = max(t if t % 2 else t + 1, 3)
        assert kernel_size % 2 == 1
        self.conv = nn.Conv1d(1, 1, kernel_size=kernel_size, padding=(kernel_size - 1) // 2, bias=False)
        self.gate = nn.Sigmoid()

    def forward(self, x):
        y = x.mean((2, 3)).view(x.shape[0], 1, -1)
        y = self.gate(self.conv(y)).view(x.shape[0], -1, 1, 1)
        return x * y.expand_as(x)


def get_attn(name):
    return {"eca": EcaModule}[name]


def get_act_layer(name):
    return {"silu": nn.SiLU, "relu": nn.ReLU, "gelu": nn.GELU}[name]


def act_with_gamma(act_type, gamma=1.):
    raise NotImplementedError("gamma_in_act is off for the eca_nfnet family")


class DownsampleAvg(nn.Module):
    def __init__(self, in_chs, out_chs, stride=1, dilation=1, first_dilation=None, conv_layer=ScaledStdConv2d):
        super().__init__()
        assert dilation == 1
        self.pool = nn.AvgPool2d(2, stride, ceil_mode=True, count_include_pad=False) if stride > 1 else nn.Identity()
        self.conv = conv_layer(in_chs, out_chs, 1, stride=1)

    def forward(self, x):
        return self.conv(self.pool(x))


class NormFreeBlock(nn.Module):
    def __init__(self, in_chs, out_chs=None, stride=1, dilation=1, first_dilation=None, alpha=1.0, beta=1.0, bottle_ratio=0.25,
                 group_size=None, ch_div=1, reg=True, extra_conv=False, skipinit=False, attn_layer=None, attn_gain=2.0, act_layer=None,
                 conv_layer=None, drop_path_rate=0.):
        super().__init__()
        first_dilation = first_dilation or dilation
        out_chs = out_chs or in_chs
        mid_chs = make_divisible(in_chs * bottle_ratio if reg else out_chs * bottle_ratio, ch_div)
        groups = 1 if not group_size else mid_chs // group_size
        if group_size and group_size % ch_div == 0:
            mid_chs = group_size * groups
        self.alpha, self.beta, self.attn_gain = alpha, beta, attn_gain
        if in_chs != out_chs or stride != 1 or dilation != first_dilation:
            self.downsample = DownsampleAvg(in_chs, out_chs, stride=stride, dilation=dilation, first_dilation=first_dilation, conv_layer=conv_layer)
        else:
            self.downsample = None
        self.act1 = act_layer()
        self.conv1 = conv_layer(in_chs, mid_chs, 1)
        self.act2 = act_layer(inplace=True)
        self.conv2 = conv_layer(mid_chs, mid_chs, 3, stride=stride, dilation=first_dilation, groups=groups)
        if extra_conv:
            self.act2b = act_layer(inplace=True)
            self.conv2b = conv_layer(mid_chs, mid_chs, 3, stride=1, dilation=dilation, groups=groups)
        else:
            self.act2b, self.conv2b = None, None
        self.attn = attn_layer(mid_chs) if reg and attn_layer is not None else None
        self.act3 = act_layer()
        self.conv3 = conv_layer(mid_chs, out_chs, 1, gain_init=1. if skipinit else 0.)
        self.attn_last = attn_layer(out_chs) if not reg and attn_layer is not None else None
        assert drop_path_rate == 0 and not skipinit

    def forward(self, x):
        out = self.act1(x) * self.beta
        shortcut = x
        if self.downsample is not None:
            shortcut = self.downsample(out)
        out = self.conv1(out)
        out = self.conv2(self.act2(out))
        if self.conv2b is not None:
            out = self.conv2b(self.act2b(out))
        if self.attn is not None:
            out = self.attn_gain * self.attn(out)
        out = self.conv3(self.act3(out))
        if self.attn_last is not None:
            out = self.attn_gain * self.attn_last(out)
        return out * self.alpha + shortcut


def create_stem(in_chs, out_chs, stem_type="", conv_layer=None, act_layer=None, preact_feature=True):
    assert stem_type == "deep_quad"
    stem = OrderedDict()
    chs, strides = (out_chs // 8, out_chs // 4, out_chs // 2, out_chs), (2, 1, 1, 2)
    for i, (c, s) in enumerate(zip(chs, strides)):
        stem[f"conv{i + 1}"] = conv_layer(in_chs, c, kernel_size=3, stride=s)
        if i != len(chs) - 1:
            stem[f"act{i + 2}"] = act_layer(inplace=True)
        in_chs = c
    return nn.Sequential(stem), 4, dict(num_chs=out_chs // 2, reduction=2, module="stem.conv3")


class _AvgPoolFlatten(nn.Module):
    def forward(self, x):
        return x.mean((2, 3))


def _create_pool(num_features, num_classes, pool_type="avg", use_conv=False):
    assert pool_type == "avg" and not use_conv
    return _AvgPoolFlatten(), num_features


def _create_fc(num_features, num_classes, use_conv=False):
    return nn.Linear(num_features, num_classes, bias=True) if num_classes > 0 else nn.Identity()


# ------------------------------------------------------------------------------------------------ resnetv2_50 family (timm resnetv2.py)
class BatchNormAct2d(nn.Module):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm2d(c, eps=eps, momentum=momentum)
        self.act = nn.ReLU()

    def forward(self, x):
        return self.act(self.bn(x))


class PreActBottleneck(nn.Module):
    def __init__(self, in_chs, out_chs, stride, downsample, bottle_ratio=0.25):
        super().__init__()
        mid = make_divisible(out_chs * bottle_ratio)
        self.downsample = nn.Conv2d(in_chs, out_chs, 1, stride=stride, bias=False) if downsample else None
        self.norm1 = BatchNormAct2d(in_chs)
        self.conv1 = nn.Conv2d(in_chs, mid, 1, bias=False)
        self.norm2 = BatchNormAct2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride=stride, padding=1, bias=False)
        self.norm3 = BatchNormAct2d(mid)
        self.conv3 = nn.Conv2d(mid, out_chs, 1, bias=False)

    def forward(self, x):
        x_preact = self.norm1(x)
        shortcut = x
        if self.downsample is not None:
            shortcut = self.downsample(x_preact)
        x = self.conv1(x_preact)
        x = self.conv2(self.norm2(x))
        x = self.conv3(self.norm3(x))
        return x + shortcut


class ResNetV2(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), channels=(256, 512, 1024, 2048), stem_chs=64):
        super().__init__()
        self.stem_conv = nn.Conv2d(3, stem_chs, 7, stride=2, padding=3, bias=False)
        self.stem_pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        blocks, prev = [], stem_chs
        for si, (d, c) in enumerate(zip(layers, channels)):
            stage = []
            for bi in range(d):
                stage.append(PreActBottleneck(prev, c, (1 if si == 0 else 2) if bi == 0 else 1, bi == 0))
                prev = c
            blocks.append(nn.ModuleList(stage))
        self.stages = nn.ModuleList(blocks)
        self.norm = BatchNormAct2d(prev)

    def forward_features(self, x):
        x = self.stem_pool(self.stem_conv(x))
        for stage in self.stages:
            for b in stage:
                x = b(x)
        return self.norm(x)

    def load_timm_state(self, sd, prefix="img_encoder"):
        """timm key names (stem.conv.weight, stages.S.blocks.B.{downsample.conv,norm1,conv1,...}.weight, norm.weight/bias)"""
        with torch.no_grad():
            self.stem_conv.weight.copy_(sd[f"{prefix}.stem.conv.weight"])
            for si, stage in enumerate(self.stages):
                for bi, b in enumerate(stage):
                    q = f"{prefix}.stages.{si}.blocks.{bi}"
                    if b.downsample is not None:
                        b.downsample.weight.copy_(sd[q + ".downsample.conv.weight"])
                    for n in ("norm1", "norm2", "norm3"):
                        getattr(b, n).bn.weight.copy_(sd[f"{q}.{n}.weight"])
                        getattr(b, n).bn.bias.copy_(sd[f"{q}.{n}.bias"])
                    for n in ("conv1", "conv2", "conv3"):
                        getattr(b, n).weight.copy_(sd[f"{q}.{n}.weight"])
            self.norm.bn.weight.copy_(sd[prefix + ".norm.weight"])
            self.norm.bn.bias.copy_(sd[prefix + ".norm.bias"])
