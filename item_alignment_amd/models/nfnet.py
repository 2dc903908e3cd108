"""ECA-NFNet image tower on the HIP engine (reference src/models/image.py:40-199 NormFreeNet, built there from timm 0.6.5's
NormFreeBlock / create_stem / ScaledStdConv2d / EcaModule / DownsampleAvg; timm itself is absent offline, so the block
definitions follow timm's published source and only the in-tree bookkeeping, image.py:98-137, could be read directly).

Activations are NHWC bf16 ([B*H*W, C] rows): a 1x1 convolution is the bf16 MFMA GEMM, a (grouped) 3x3 convolution is a patch
gather + one GEMM per group (csrc/conv.hip).  Module / parameter names are timm's, so `image_encoder.bin` state_dicts load:
`stem.conv{1..4}.{weight,bias,gain}`, `stages.{s}.{b}.{downsample.conv,conv1,conv2,conv2b,conv3}.{weight,bias,gain}`,
`stages.{s}.{b}.attn_last.conv.weight`, `final_conv.*`, `head.fc.*`.
"""
import math
import os

import torch
from torch import nn

from .. import _lib
from .._lib import check, ptr, stream_ptr
from . import functional as Fn
from .base import HipModule

BF16, F32 = torch.bfloat16, torch.float32
NONLIN_GAMMA_SILU = 1.7881293296813965          # timm nfnet.py _nonlin_gamma['silu']
# stride-1 grouped 3x3 convolutions run as shifted-view GEMMs over a zero-bordered tensor; IA_CONV_PATCH_MATRIX=1 keeps the
# gathered patch matrix for them too (A/B measurement switch, tools/config_bench.py)
PADDED_CONV = os.environ.get("IA_CONV_PATCH_MATRIX", "0") != "1"
FUSE_TAIL_ACT = os.environ.get("IA_NFNET_FUSE_TAIL", "1") != "0"   # a block's tail pass also writes the next block's opening activation (EcaResidualFn)
FUSE_TAIL_BWD = os.environ.get("IA_NFNET_FUSE_TAIL", "1") == "1"   # ... and its backward folds SiLU' and the gate gradient's spatial sums into one pass
ECA_LINEAR = os.environ.get("IA_ECA_LINEAR", "1") != "0"      # ECA pooling from conv3's input (EcaResidualFn); 0: the reduction over conv3's output

NFNET_CONFIGS = {   # timm nfnet.py model_cfgs (_nfnet_cfg): depths, channels, feat_mult
    "eca_nfnet_l0": ((1, 2, 6, 3), (256, 512, 1536, 1536), 1.5),
    "eca_nfnet_l1": ((2, 4, 12, 6), (256, 512, 1536, 1536), 2.0),
    "eca_nfnet_l2": ((3, 6, 18, 9), (256, 512, 1536, 1536), 2.0),
}


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def _ws(dev, nbytes):
    return torch.empty(max(int(nbytes), 16), device=dev, dtype=torch.uint8)


# ----------------------------------------------------------------------------------------- autograd glue
class StdConvFn(torch.autograd.Function):
    """ScaledStdConv2d on NHWC rows: standardise the weight (ia_ws_conv_weight_fwd), convolve (ia_conv_nhwc_fwd); backward
    = data gradient, weight gradient of the standardised weight, then back through the standardisation into the arena."""

    @staticmethod
    def forward(ctx, x, weight, conv, B, H, W):
        lib = _lib.load()
        Fn._need_gpu(x, "feature map")
        x = x.contiguous()
        C, Cout, k, s, g = conv.in_pad, conv.out_channels, conv.kernel_size, conv.stride, conv.groups
        Cg, Cgp, kk = conv.weight.shape[1], C // g, k * k
        dev = x.device
        what = torch.empty((Cout, kk * Cgp), device=dev, dtype=BF16)
        mean = torch.empty(Cout, device=dev, dtype=F32)
        rstd = torch.empty(Cout, device=dev, dtype=F32)
        check(lib.ia_ws_conv_weight_fwd(conv.weight.data_ptr(), conv.gain.data_ptr(), what.data_ptr(), mean.data_ptr(), rstd.data_ptr(), Cout, Cg,
                                        kk, Cgp, conv.scale, conv.eps, stream_ptr()), "ia_ws_conv_weight_fwd")
        Ho, Wo = (H, W) if k == 1 else ((H - 1) // s + 1, (W - 1) // s + 1)
        y = torch.empty((B * Ho * Wo, Cout), device=dev, dtype=BF16)
        wsb = lib.ia_conv_nhwc_workspace_bytes(B, H, W, C, Cout, k, s, g)
        ws = _ws(dev, wsb)
        check(lib.ia_conv_nhwc_fwd(x.data_ptr(), what.data_ptr(), ptr(conv.bias), y.data_ptr(), B, H, W, C, Cout, k, s, g, ws.data_ptr(), wsb,
                                   stream_ptr()), "ia_conv_nhwc_fwd")
        ctx.conv, ctx.saved, ctx.dims = conv, (x, what, mean, rstd), (B, H, W, C, Cout, k, s, g, Cg, Cgp, kk)
        ctx.need_dx = ctx.needs_input_grad[0]
        conv.__dict__["_last_what"] = what      # the standardised bf16 weight of this call (the block tail takes the ECA pooling from it)
        # keep the 3x3 patch matrix for the weight gradient (9x the input, a few GB per step on 288 GB of HBM) instead of
        # gathering it again in backward
        ctx.cols_ws = ws if (k == 3 and conv.weight.requires_grad and conv.keep_cols) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        conv = ctx.conv
        x, what, mean, rstd = ctx.saved
        B, H, W, C, Cout, k, s, g, Cg, Cgp, kk = ctx.dims
        dy = dy.contiguous()
        dev = dy.device
        wsb = lib.ia_conv_nhwc_workspace_bytes(B, H, W, C, Cout, k, s, g)
        ws = _ws(dev, wsb)                 # scratch of the data gradient (it overwrites the patch area)
        dx = None
        if ctx.need_dx:
            dx = torch.empty_like(x)
            check(lib.ia_conv_nhwc_bwd_data(dy.data_ptr(), what.data_ptr(), dx.data_ptr(), B, H, W, C, Cout, k, s, g, ws.data_ptr(), wsb,
                                            stream_ptr()), "ia_conv_nhwc_bwd_data")
        if conv.weight.requires_grad:
            dwhat = torch.empty((Cout, kk * Cgp), device=dev, dtype=F32)
            bg = conv.bias.grad.data_ptr() if conv.bias is not None and conv.bias.requires_grad else None
            wws = ctx.cols_ws if ctx.cols_ws is not None else ws
            check(lib.ia_conv_nhwc_bwd_weight(x.data_ptr(), dy.data_ptr(), dwhat.data_ptr(), bg, B, H, W, C, Cout, k, s, g,
                                              int(ctx.cols_ws is not None), wws.data_ptr(), wsb, stream_ptr()), "ia_conv_nhwc_bwd_weight")
            ctx.cols_ws = None
            check(lib.ia_ws_conv_weight_bwd(dwhat.data_ptr(), conv.weight.data_ptr(), conv.gain.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            conv.weight.grad.data_ptr(), conv.gain.grad.data_ptr(), Cout, Cg, kk, Cgp, conv.scale, stream_ptr()),
                  "ia_ws_conv_weight_bwd")
            Fn._notify([p for p in (conv.weight, conv.bias, conv.gain) if p is not None])
        ctx.saved = None
        return dx, None, None, None, None, None


class PaddedStdConvFn(torch.autograd.Function):
    """ScaledStdConv2d, 3x3 / stride 1 (power-of-two channels per group), on the zero-bordered domain [B*(H+2)*(W+2), C]: no
    patch matrix — every tap is read as the input shifted by whole rows inside the GEMM, all groups in one launch
    (ia_conv3x3_padded_*).  The input must carry a zero border (SiluPadFn writes it); the output's border rows are garbage
    and are only ever read by SiluPadFn, which ignores them.  The incoming gradient needs a zero border too:
    SiluPadFn.backward provides it."""

    @staticmethod
    def forward(ctx, xp, weight, conv, B, H, W):
        lib = _lib.load()
        Fn._need_gpu(xp, "feature map")
        Cin, Cout, g = conv.in_channels, conv.out_channels, conv.groups
        ci = Cin // g
        dev = xp.device
        what = torch.empty((Cout, 9 * ci), device=dev, dtype=BF16)
        mean = torch.empty(Cout, device=dev, dtype=F32)
        rstd = torch.empty(Cout, device=dev, dtype=F32)
        check(lib.ia_ws_conv_weight_fwd(conv.weight.data_ptr(), conv.gain.data_ptr(), what.data_ptr(), mean.data_ptr(), rstd.data_ptr(), Cout, ci, 9,
                                        ci, conv.scale, conv.eps, stream_ptr()), "ia_ws_conv_weight_fwd")
        yp = torch.empty((xp.shape[0], Cout), device=dev, dtype=BF16)
        check(lib.ia_conv3x3_padded_fwd(xp.data_ptr(), what.data_ptr(), ptr(conv.bias), yp.data_ptr(), B, H, W, Cin, Cout, g, stream_ptr()),
              "ia_conv3x3_padded_fwd")
        ctx.conv, ctx.saved, ctx.dims = conv, (xp, what, mean, rstd), (B, H, W, Cin, Cout, g, ci)
        ctx.need_dx = ctx.needs_input_grad[0]
        return yp

    @staticmethod
    def backward(ctx, dyp):
        lib = _lib.load()
        conv = ctx.conv
        xp, what, mean, rstd = ctx.saved
        B, H, W, Cin, Cout, g, ci = ctx.dims
        dyp = dyp.contiguous()
        dev = dyp.device
        dxp = None
        if ctx.need_dx:
            dxp = torch.empty_like(xp)
            if lib.ia_conv3x3_direct_supported(Cin, Cout, g):
                # the direct kernel on dy with the tap-flipped, transposed filter bank [Cin][9 * Cout / groups] (as many elements as `what`)
                what_t = torch.empty((Cin, 9 * (Cout // g)), device=dev, dtype=BF16)
                check(lib.ia_conv3x3_flip_weights(what.data_ptr(), what_t.data_ptr(), Cin, Cout, g, stream_ptr()), "ia_conv3x3_flip_weights")
                check(lib.ia_conv3x3_padded_bwd_data_t(dyp.data_ptr(), what_t.data_ptr(), dxp.data_ptr(), B, H, W, Cin, Cout, g, stream_ptr()),
                      "ia_conv3x3_padded_bwd_data_t")
            else:
                check(lib.ia_conv3x3_padded_bwd_data(dyp.data_ptr(), what.data_ptr(), dxp.data_ptr(), B, H, W, Cin, Cout, g, stream_ptr()),
                      "ia_conv3x3_padded_bwd_data")
        if conv.weight.requires_grad:
            dwhat = torch.empty((Cout, 9 * ci), device=dev, dtype=F32)
            bg = conv.bias.grad.data_ptr() if conv.bias is not None and conv.bias.requires_grad else None
            wsb = lib.ia_conv3x3_padded_workspace_bytes(B, H, W, Cin, Cout, g)
            ws = _ws(dev, wsb)
            check(lib.ia_conv3x3_padded_bwd_weight(xp.data_ptr(), dyp.data_ptr(), dwhat.data_ptr(), bg, B, H, W, Cin, Cout, g, ws.data_ptr(), wsb,
                                                   stream_ptr()), "ia_conv3x3_padded_bwd_weight")
            check(lib.ia_ws_conv_weight_bwd(dwhat.data_ptr(), conv.weight.data_ptr(), conv.gain.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            conv.weight.grad.data_ptr(), conv.gain.grad.data_ptr(), Cout, ci, 9, ci, conv.scale, stream_ptr()),
                  "ia_ws_conv_weight_bwd")
            Fn._notify([p for p in (conv.weight, conv.bias, conv.gain) if p is not None])
        ctx.saved = None
        return dxp, None, None, None, None, None


class PaddedS2ConvFn(torch.autograd.Function):
    """ScaledStdConv2d, 3x3 / stride 2 (64 channels per group, or the stem's 64 -> 128), from the zero-bordered domain at H x W to the
    bordered -- or, `y_compact`, the compact -- domain at ceil(H/2) x ceil(W/2) without a patch matrix (ia_conv3x3_s2_padded_*: forward over
    the four parity views of x, weight gradient = the direct kernel on those views, data gradient = GEMM + gather).  Borders as for
    PaddedStdConvFn: the input's must be zero, the output's is garbage, the incoming gradient's is not read."""

    @staticmethod
    def forward(ctx, xp, weight, conv, B, H, W, y_compact):
        lib = _lib.load()
        Fn._need_gpu(xp, "feature map")
        Cin, Cout, g = conv.in_channels, conv.out_channels, conv.groups
        ci = Cin // g
        dev = xp.device
        what = torch.empty((Cout, 9 * ci), device=dev, dtype=BF16)
        mean = torch.empty(Cout, device=dev, dtype=F32)
        rstd = torch.empty(Cout, device=dev, dtype=F32)
        check(lib.ia_ws_conv_weight_fwd(conv.weight.data_ptr(), conv.gain.data_ptr(), what.data_ptr(), mean.data_ptr(), rstd.data_ptr(), Cout, ci, 9,
                                        ci, conv.scale, conv.eps, stream_ptr()), "ia_ws_conv_weight_fwd")
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        rows = B * Ho * Wo if y_compact else B * (Ho + 2) * (Wo + 2)
        yp = torch.empty((rows, Cout), device=dev, dtype=BF16)
        check(lib.ia_conv3x3_s2_padded_fwd(xp.data_ptr(), what.data_ptr(), ptr(conv.bias), yp.data_ptr(), B, H, W, Cin, Cout, g, int(y_compact),
                                           stream_ptr()), "ia_conv3x3_s2_padded_fwd")
        ctx.conv, ctx.saved, ctx.dims = conv, (xp, what, mean, rstd), (B, H, W, Cin, Cout, g, ci, int(y_compact))
        ctx.need_dx = ctx.needs_input_grad[0]
        return yp

    @staticmethod
    def backward(ctx, dyp):
        lib = _lib.load()
        conv = ctx.conv
        xp, what, mean, rstd = ctx.saved
        B, H, W, Cin, Cout, g, ci, yc = ctx.dims
        dyp = dyp.contiguous()
        dev = dyp.device
        wsb = lib.ia_conv3x3_s2_padded_workspace_bytes(B, H, W, Cin, Cout, g)
        ws = _ws(dev, wsb)
        dxp = None
        if ctx.need_dx:
            dxp = torch.empty_like(xp)
            if lib.ia_conv3x3_s2_dgrad_supported(Cin, Cout, g):
                # one kernel over the four parity classes of dx, on the tap-flipped transposed bank (the incoming gradient's border is zero:
                # the SiLU backward behind this convolution writes it)
                what_t = torch.empty((Cout, 9 * ci), device=dev, dtype=BF16)
                if Cout == Cin:
                    check(lib.ia_conv3x3_flip_weights(what.data_ptr(), what_t.data_ptr(), Cin, Cout, g, stream_ptr()), "ia_conv3x3_flip_weights")
                else:       # the stem's 64 -> 128: one bank per 64-channel slice of dy (the flip sees them as groups of a 128 -> 128 layout)
                    check(lib.ia_conv3x3_flip_weights(what.data_ptr(), what_t.data_ptr(), Cout, Cout, Cout // 64, stream_ptr()), "ia_conv3x3_flip_weights")
                check(lib.ia_conv3x3_s2_padded_bwd_data_t(dyp.data_ptr(), what_t.data_ptr(), dxp.data_ptr(), B, H, W, Cin, Cout, g, yc, stream_ptr()),
                      "ia_conv3x3_s2_padded_bwd_data_t")
            else:
                check(lib.ia_conv3x3_s2_padded_bwd_data(dyp.data_ptr(), what.data_ptr(), dxp.data_ptr(), B, H, W, Cin, Cout, g, yc, ws.data_ptr(), wsb,
                                                        stream_ptr()), "ia_conv3x3_s2_padded_bwd_data")
        if conv.weight.requires_grad:
            dwhat = torch.empty((Cout, 9 * ci), device=dev, dtype=F32)
            bg = conv.bias.grad.data_ptr() if conv.bias is not None and conv.bias.requires_grad else None
            check(lib.ia_conv3x3_s2_padded_bwd_weight(xp.data_ptr(), dyp.data_ptr(), dwhat.data_ptr(), bg, B, H, W, Cin, Cout, g, yc, ws.data_ptr(), wsb,
                                                      stream_ptr()), "ia_conv3x3_s2_padded_bwd_weight")
            check(lib.ia_ws_conv_weight_bwd(dwhat.data_ptr(), conv.weight.data_ptr(), conv.gain.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            conv.weight.grad.data_ptr(), conv.gain.grad.data_ptr(), Cout, ci, 9, ci, conv.scale, stream_ptr()),
                  "ia_ws_conv_weight_bwd")
            Fn._notify([p for p in (conv.weight, conv.bias, conv.gain) if p is not None])
        ctx.saved = None
        return dxp, None, None, None, None, None, None


class SiluPadFn(torch.autograd.Function):
    """y = silu(x) * scale while moving between the compact [B*H*W, C] rows and the zero-bordered [B*(H+2)*(W+2), C] rows
    that PaddedStdConvFn works on (one flag per side; ia_silu_pad_*).  A padded output gets its border written as zero, a
    padded input's border is never read; the same holds for the gradients in backward."""

    @staticmethod
    def forward(ctx, x, scale, B, H, W, in_padded, out_padded):
        lib = _lib.load()
        x = x.contiguous()
        C = x.shape[1]
        rows = B * (H + 2) * (W + 2) if out_padded else B * H * W
        y = torch.empty((rows, C), device=x.device, dtype=BF16)
        check(lib.ia_silu_pad_fwd(x.data_ptr(), y.data_ptr(), B, H, W, C, scale, int(in_padded), int(out_padded), stream_ptr()), "ia_silu_pad_fwd")
        ctx.x, ctx.args = x, (B, H, W, C, scale, int(in_padded), int(out_padded))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x = ctx.x
        B, H, W, C, scale, in_p, out_p = ctx.args
        dx = torch.empty_like(x)
        check(lib.ia_silu_pad_bwd(dy.contiguous().data_ptr(), x.data_ptr(), dx.data_ptr(), B, H, W, C, scale, in_p, out_p, stream_ptr()),
              "ia_silu_pad_bwd")
        ctx.x = None
        return dx, None, None, None, None, None, None


class SiluFn(torch.autograd.Function):
    """y = silu(x) * scale.  passthrough = True: also hands x back as a second output (the identity shortcut of a NormFreeBlock);
    passthrough = 2: hands y out twice (a downsampling block's activation feeds conv1 and the projected shortcut).  Either way the
    two gradients are combined inside the SiLU backward kernel instead of by a separate torch add over the feature map."""

    @staticmethod
    def forward(ctx, x, scale, passthrough):
        lib = _lib.load()
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.ia_silu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), scale, stream_ptr()), "ia_silu_fwd")
        ctx.x, ctx.scale, ctx.mode = x, scale, passthrough
        ctx.set_materialize_grads(False)
        if passthrough == 2:
            return y, y.view_as(y)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dother=None):
        lib = _lib.load()
        x = ctx.x
        if ctx.mode == 2:
            if dy is None or dother is None:
                dy = dy if dother is None else dother
                if dy is None:
                    return None, None, None
                dother = None
            dx = torch.empty_like(x)
            if dother is None:
                check(lib.ia_silu_bwd(dy.contiguous().data_ptr(), x.data_ptr(), None, dx.data_ptr(), x.numel(), ctx.scale, stream_ptr()), "ia_silu_bwd")
            else:
                check(lib.ia_silu_bwd_sum(dy.contiguous().data_ptr(), dother.contiguous().data_ptr(), x.data_ptr(), None, dx.data_ptr(), x.numel(),
                                          ctx.scale, stream_ptr()), "ia_silu_bwd_sum")
            return dx, None, None
        dpass = dother
        if dy is None:
            return dpass, None, None
        dx = torch.empty_like(x)
        dp = None if dpass is None else dpass.contiguous()
        check(lib.ia_silu_bwd(dy.contiguous().data_ptr(), x.data_ptr(), ptr(dp), dx.data_ptr(), x.numel(), ctx.scale, stream_ptr()), "ia_silu_bwd")
        return dx, None, None


class AvgPool2Fn(torch.autograd.Function):
    """AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False) (timm DownsampleAvg)."""

    @staticmethod
    def forward(ctx, x, B, H, W):
        lib = _lib.load()
        x = x.contiguous()
        C = x.shape[1]
        y = torch.empty((B * ((H + 1) // 2) * ((W + 1) // 2), C), device=x.device, dtype=BF16)
        check(lib.ia_avgpool2_fwd(x.data_ptr(), y.data_ptr(), B, H, W, C, stream_ptr()), "ia_avgpool2_fwd")
        ctx.dims = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        B, H, W, C = ctx.dims
        dx = torch.empty((B * H * W, C), device=dy.device, dtype=BF16)
        check(lib.ia_avgpool2_bwd(dy.contiguous().data_ptr(), dx.data_ptr(), B, H, W, C, stream_ptr()), "ia_avgpool2_bwd")
        return dx, None, None, None


class EcaResidualFn(torch.autograd.Function):
    """out = x * sigmoid(conv1d(mean_HW x)) * (attn_gain * alpha) + shortcut  (timm EcaModule + NormFreeBlock.forward tail).

    pre = (a, what, bias): x = a what^T + bias is the output of a 1x1 convolution (conv3): the ECA pooling mean_HW(x) is then taken as
    (mean_HW a) what^T + bias -- exact, the mean commutes with the per-pixel linear map -- from the 4 x narrower tensor a
    (ia_eca_fwd_linear, round 6; IA_ECA_LINEAR=0 keeps the reduction over x for A/B runs).
    act_mode 1 / 2 (with pre): the same pass also writes act = silu(out) * act_scale, the activation the NEXT block opens with
    (act1(x) * beta; mode 2 hands it out twice: a downsampling block feeds it to conv1 and to the projected shortcut), so `out` is not
    read back by a SiLU pass of its own; the gradient that arrives through act is folded into out's by ia_silu_bwd(_sum) here -- the
    same kernels SiluFn.backward ran, the same arithmetic."""

    @staticmethod
    def forward(ctx, x, shortcut, conv_w, eca, B, HW, coef, pre=None, act_scale=1.0, act_mode=0):
        lib = _lib.load()
        x, shortcut = x.contiguous(), shortcut.contiguous()
        C, k = x.shape[1], conv_w.shape[-1]
        dev = x.device
        out = torch.empty_like(x)
        pooled = torch.empty((B, C), device=dev, dtype=F32)
        gate = torch.empty((B, C), device=dev, dtype=F32)
        act = None
        if pre is not None and ECA_LINEAR:
            a, what, bias = pre
            Cmid = a.shape[1]
            wsb = lib.ia_eca_fwd_linear_workspace_bytes(B, HW, Cmid)
            ws = _ws(dev, wsb)
            if act_mode:
                act = torch.empty_like(x)
            check(lib.ia_eca_fwd_linear(x.data_ptr(), a.data_ptr(), what.data_ptr(), ptr(bias), Cmid, conv_w.data_ptr(), k, shortcut.data_ptr(),
                                        out.data_ptr(), ptr(act), act_scale, pooled.data_ptr(), gate.data_ptr(), B, HW, C, coef, ws.data_ptr(), wsb,
                                        stream_ptr()), "ia_eca_fwd_linear")
        else:
            wsb = lib.ia_gap_workspace_bytes(B, HW, C)
            ws = _ws(dev, wsb)
            check(lib.ia_eca_fwd(x.data_ptr(), conv_w.data_ptr(), k, shortcut.data_ptr(), out.data_ptr(), pooled.data_ptr(), gate.data_ptr(), B, HW, C,
                                 coef, ws.data_ptr(), wsb, stream_ptr()), "ia_eca_fwd")
        ctx.eca, ctx.saved, ctx.dims = eca, (x, pooled, gate), (B, HW, C, k, coef)
        ctx.act_scale, ctx.out = act_scale, (out if act is not None else None)
        if act is None:
            return out
        ctx.set_materialize_grads(False)
        return (out, act, act.view_as(act)) if act_mode == 2 else (out, act)

    @staticmethod
    def backward(ctx, dout, dact=None, dact2=None):
        lib = _lib.load()
        x, pooled, gate = ctx.saved
        B, HW, C, k, coef = ctx.dims
        w = ctx.eca.conv.weight
        if ctx.out is not None:
            # out's whole gradient = what arrives at `out` itself (the next block's identity shortcut; nothing for a downsampling block)
            # + (dact [+ dact2]) * act_scale * silu'(out): formed by ia_eca_silu_bwd in the pass that also takes the gate gradient's spatial
            # sums (IA_NFNET_FUSE_TAIL=2: by ia_silu_bwd(_sum) first, then the plain ia_eca_bwd -- the two-kernel form, for A/B runs)
            g1, g2 = (dact, dact2) if dact is not None else (dact2, None)
            out, ctx.out = ctx.out, None
            if g1 is not None:
                dtot = torch.empty_like(out)
                dadd = None if dout is None else dout.contiguous()
                g1 = g1.contiguous()
                g2 = None if g2 is None else g2.contiguous()
                if FUSE_TAIL_BWD:
                    dx = torch.empty_like(x)
                    wsb = lib.ia_eca_bwd_workspace_bytes(B, HW, C)
                    ws = _ws(dtot.device, wsb)
                    check(lib.ia_eca_silu_bwd(g1.data_ptr(), ptr(g2), out.data_ptr(), ptr(dadd), ctx.act_scale, x.data_ptr(), w.data_ptr(), k,
                                              pooled.data_ptr(), gate.data_ptr(), dtot.data_ptr(), dx.data_ptr(),
                                              w.grad.data_ptr() if w.requires_grad else None, B, HW, C, coef, ws.data_ptr(), wsb, stream_ptr()),
                          "ia_eca_silu_bwd")
                    Fn._notify([w])
                    ctx.saved = None
                    return dx, dtot, None, None, None, None, None, None, None, None
                if g2 is None:
                    check(lib.ia_silu_bwd(g1.data_ptr(), out.data_ptr(), ptr(dadd), dtot.data_ptr(), out.numel(), ctx.act_scale, stream_ptr()),
                          "ia_silu_bwd")
                else:
                    check(lib.ia_silu_bwd_sum(g1.data_ptr(), g2.data_ptr(), out.data_ptr(), ptr(dadd), dtot.data_ptr(), out.numel(), ctx.act_scale,
                                              stream_ptr()), "ia_silu_bwd_sum")
                dout = dtot
        if dout is None:
            ctx.saved = None
            return (None,) * 10
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        wsb = lib.ia_eca_bwd_workspace_bytes(B, HW, C)
        ws = _ws(dout.device, wsb)
        check(lib.ia_eca_bwd(dout.data_ptr(), x.data_ptr(), w.data_ptr(), k, pooled.data_ptr(), gate.data_ptr(), dx.data_ptr(),
                             w.grad.data_ptr() if w.requires_grad else None, B, HW, C, coef, ws.data_ptr(), wsb, stream_ptr()), "ia_eca_bwd")
        Fn._notify([w])
        ctx.saved = None
        return dx, dout, None, None, None, None, None, None, None, None


class GapFn(torch.autograd.Function):
    """[B*HW, C] bf16 -> mean over HW -> [B, C] fp32 (SelectAdaptivePool2d('avg', flatten=True))."""

    @staticmethod
    def forward(ctx, x, B, HW):
        lib = _lib.load()
        x = x.contiguous()
        C = x.shape[1]
        pooled = torch.empty((B, C), device=x.device, dtype=F32)
        wsb = lib.ia_gap_workspace_bytes(B, HW, C)
        ws = _ws(x.device, wsb)
        check(lib.ia_gap_fwd(x.data_ptr(), pooled.data_ptr(), B, HW, C, ws.data_ptr(), wsb, stream_ptr()), "ia_gap_fwd")
        ctx.dims = (B, HW, C)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        lib = _lib.load()
        B, HW, C = ctx.dims
        dx = torch.empty((B * HW, C), device=dp.device, dtype=BF16)
        check(lib.ia_gap_bwd(dp.contiguous().to(F32).data_ptr(), dx.data_ptr(), B, HW, C, stream_ptr()), "ia_gap_bwd")
        return dx, None, None


# ---------------------------------------------------------------------------------------------- modules
class FeatureMap:
    """NHWC feature map handed between the tower's modules: rows [B*H*W, C] bf16 + its geometry."""
    __slots__ = ("t", "B", "H", "W", "act")

    def __init__(self, t, B, H, W, act=None):
        self.t, self.B, self.H, self.W = t, B, H, W
        self.act = act          # (tensors) the next block's opening activation silu(t) * beta, when the producing block tail wrote it already

    @property
    def shape(self):
        return (self.B, self.H, self.W, self.t.shape[1])


class ScaledStdConv2d(nn.Module):
    """timm layers/std_conv.py ScaledStdConv2d (gamma folded into the weight scale, eps 1e-5, bias on)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, groups=1, gamma=NONLIN_GAMMA_SILU, eps=1e-5, gain_init=1.0):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.groups, self.eps = in_channels, out_channels, kernel_size, stride, groups, eps
        self.in_pad = max(8, in_channels) if groups == 1 else in_channels      # the 3-channel image is zero-padded to 8
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.gain = nn.Parameter(torch.full((out_channels, 1, 1, 1), gain_init))
        self.scale = gamma * self.weight[0].numel() ** -0.5
        self.keep_cols = True              # stash the 3x3 patch matrix from forward for the weight gradient
        nn.init.kaiming_normal_(self.weight, mode="fan_in", nonlinearity="linear")       # reference image.py:156

    @property
    def shifted_views(self):
        """3x3 / stride 1 / power-of-two channels per group: runs without a patch matrix on the zero-bordered domain"""
        ci, co = self.in_channels // self.groups, self.out_channels // self.groups
        return (PADDED_CONV and self.kernel_size == 3 and self.stride == 1 and ci >= 8 and co >= 8 and not (ci & (ci - 1)) and not (co & (co - 1)))

    @property
    def strided_direct(self):
        """3x3 / stride 2 with a shape the patch-matrix-free strided kernels take (64 channels per group; the stem's 64 -> 128)"""
        return (PADDED_CONV and self.kernel_size == 3 and self.stride == 2
                and bool(_lib.load().ia_conv3x3_s2_supported(self.in_channels, self.out_channels, self.groups)))

    def fits_padded(self, B, H, W):
        """row indices of the padded tensor are 32-bit ints (its byte size is not limited)"""
        return B * (H + 2) * (W + 2) < 0x7FFFFFFF

    def forward(self, f):
        y = StdConvFn.apply(f.t, self.weight, self, f.B, f.H, f.W)
        k, s = self.kernel_size, self.stride
        Ho, Wo = (f.H, f.W) if k == 1 else ((f.H - 1) // s + 1, (f.W - 1) // s + 1)
        return FeatureMap(y, f.B, Ho, Wo)


class EcaModule(nn.Module):
    """timm layers/eca.py EcaModule: kernel size from the channel count (gamma 2, beta 1)."""

    def __init__(self, channels, gamma=2, beta=1):
        super().__init__()
        t = int(abs(math.log(channels, 2) + beta) / gamma)
        k = max(t if t % 2 else t + 1, 3)
        self.conv = nn.Conv1d(1, 1, kernel_size=k, padding=(k - 1) // 2, bias=False)


class DownsampleAvg(nn.Module):
    """timm nfnet.py DownsampleAvg: 2x2 average pool (stride > 1 only) then a 1x1 ScaledStdConv2d."""

    def __init__(self, in_chs, out_chs, stride=1):
        super().__init__()
        self.stride = stride
        self.pool = nn.AvgPool2d(2, stride, ceil_mode=True, count_include_pad=False) if stride > 1 else nn.Identity()
        self.conv = ScaledStdConv2d(in_chs, out_chs, 1)

    def forward(self, f):
        if self.stride > 1:
            f = FeatureMap(AvgPool2Fn.apply(f.t, f.B, f.H, f.W), f.B, (f.H + 1) // 2, (f.W + 1) // 2)
        return self.conv(f)


class NormFreeBlock(nn.Module):
    """timm nfnet.py NormFreeBlock (reg=False, extra_conv=True, skipinit off, eca as attn_last, no drop path)."""

    def __init__(self, in_chs, out_chs, stride=1, alpha=0.2, beta=1.0, bottle_ratio=0.25, group_size=64, ch_div=8, attn_gain=2.0):
        super().__init__()
        mid_chs = make_divisible(out_chs * bottle_ratio, ch_div)
        groups = mid_chs // group_size
        mid_chs = group_size * groups
        self.alpha, self.beta, self.attn_gain = alpha, beta, attn_gain
        self.downsample = DownsampleAvg(in_chs, out_chs, stride=stride) if (in_chs != out_chs or stride != 1) else None
        self.conv1 = ScaledStdConv2d(in_chs, mid_chs, 1)
        self.conv2 = ScaledStdConv2d(mid_chs, mid_chs, 3, stride=stride, groups=groups)
        self.conv2b = ScaledStdConv2d(mid_chs, mid_chs, 3, stride=1, groups=groups)
        self.conv3 = ScaledStdConv2d(mid_chs, out_chs, 1, gain_init=0.0)       # timm: gain_init = 1 if skipinit else 0
        self.attn_last = EcaModule(out_chs)

    def forward(self, f, next_block=None):
        """next_block: the NormFreeBlock that consumes this block's output (None: something else does, e.g. final_conv) -- its opening
        activation silu(out) * beta' is then written by this block's tail pass (EcaResidualFn) and travels as FeatureMap.act"""
        act = lambda g: FeatureMap(SiluFn.apply(g.t, 1.0, False), g.B, g.H, g.W)
        if self.downsample is not None:
            ya, yb = f.act if f.act is not None else SiluFn.apply(f.t, self.beta, 2)
            out = FeatureMap(ya, f.B, f.H, f.W)
            shortcut = self.downsample(FeatureMap(yb, f.B, f.H, f.W)).t
        else:
            o, shortcut = (f.act[0], f.t) if f.act is not None else SiluFn.apply(f.t, self.beta, True)
            out = FeatureMap(o, f.B, f.H, f.W)
        out = self.conv1(out)
        B, H, W = out.B, out.H, out.W
        fits = self.conv2b.fits_padded(B, H, W)
        if self.conv2.shifted_views and self.conv2b.shifted_views and fits:
            # conv1 -> [silu -> padded] conv2 [silu, padded -> padded] conv2b [silu -> compact] -> conv3
            t = SiluPadFn.apply(out.t, 1.0, B, H, W, False, True)
            t = PaddedStdConvFn.apply(t, self.conv2.weight, self.conv2, B, H, W)
            t = SiluPadFn.apply(t, 1.0, B, H, W, True, True)
            t = PaddedStdConvFn.apply(t, self.conv2b.weight, self.conv2b, B, H, W)
            out = FeatureMap(SiluPadFn.apply(t, 1.0, B, H, W, True, False), B, H, W)
        elif self.conv2.strided_direct and self.conv2b.shifted_views and fits:
            # the same chain through a stage transition: conv2 reads the bordered map at H x W and writes the bordered map at half size
            t = SiluPadFn.apply(out.t, 1.0, B, H, W, False, True)
            t = PaddedS2ConvFn.apply(t, self.conv2.weight, self.conv2, B, H, W, False)
            H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            t = SiluPadFn.apply(t, 1.0, B, H, W, True, True)
            t = PaddedStdConvFn.apply(t, self.conv2b.weight, self.conv2b, B, H, W)
            out = FeatureMap(SiluPadFn.apply(t, 1.0, B, H, W, True, False), B, H, W)
        elif self.conv2b.shifted_views and fits:
            out = self.conv2(act(out))                       # the strided 3x3 keeps the patch-matrix path
            B, H, W = out.B, out.H, out.W
            t = SiluPadFn.apply(out.t, 1.0, B, H, W, False, True)
            t = PaddedStdConvFn.apply(t, self.conv2b.weight, self.conv2b, B, H, W)
            out = FeatureMap(SiluPadFn.apply(t, 1.0, B, H, W, True, False), B, H, W)
        else:
            out = self.conv2(act(out))
            out = act(self.conv2b(act(out)))
        a = out.t                                                 # conv3's input [B*H*W, mid] (contiguous: every producer above allocates it)
        out = self.conv3(out)
        pre = (a, self.conv3.__dict__.pop("_last_what"), self.conv3.bias) if a.is_contiguous() else None
        fuse = next_block is not None and pre is not None and ECA_LINEAR and FUSE_TAIL_ACT
        mode = 0 if not fuse else (2 if next_block.downsample is not None else 1)
        y = EcaResidualFn.apply(out.t, shortcut, self.attn_last.conv.weight, self.attn_last, out.B, out.H * out.W, self.attn_gain * self.alpha, pre,
                                next_block.beta if fuse else 1.0, mode)
        if mode:
            return FeatureMap(y[0], out.B, out.H, out.W, act=tuple(y[1:]))
        return FeatureMap(y, out.B, out.H, out.W)


class _GlobalPool(nn.Module):
    def forward(self, f):
        return GapFn.apply(f.t, f.B, f.H * f.W)


class _Head(nn.Module):
    """timm ClassifierHead: only `global_pool` is on the item-alignment path (reference image.py:255); `fc` is kept for
    state_dict parity with timm checkpoints."""

    def __init__(self, num_features, num_classes=1000):
        super().__init__()
        self.global_pool = _GlobalPool()
        self.fc = nn.Linear(num_features, num_classes)


class NormFreeNet(HipModule):
    """reference image.py:40-199.  forward_features(images [B,3,S,S] fp32) -> FeatureMap ([B, S/32, S/32, num_features] NHWC
    bf16); head.global_pool(map) -> [B, num_features] fp32."""

    def __init__(self, depths, channels, feat_mult, stem_chs=128, group_size=64, bottle_ratio=0.25, alpha=0.2, attn_gain=2.0, num_classes=1000):
        super().__init__()
        chs = (stem_chs // 8, stem_chs // 4, stem_chs // 2, stem_chs)
        stem, cin = [], 3
        for i, (c, s) in enumerate(zip(chs, (2, 1, 1, 2))):                       # timm create_stem('deep_quad')
            stem.append((f"conv{i + 1}", ScaledStdConv2d(cin, c, 3, stride=s)))
            if i != 3:
                stem.append((f"act{i + 2}", nn.SiLU()))
            cin = c
        from collections import OrderedDict
        self.stem = nn.Sequential(OrderedDict(stem))
        prev, expected_var, stages = stem_chs, 1.0, []
        for si, depth in enumerate(depths):                                        # reference image.py:98-137
            stride = 1 if si == 0 else 2
            blocks = []
            for bi in range(depth):
                out_chs = make_divisible(channels[si], 8)
                blocks.append(NormFreeBlock(prev, out_chs, stride=stride if bi == 0 else 1, alpha=alpha, beta=1.0 / expected_var ** 0.5,
                                            bottle_ratio=bottle_ratio, group_size=group_size, attn_gain=attn_gain))
                if bi == 0:
                    expected_var = 1.0
                expected_var += alpha ** 2
                prev = out_chs
            stages.append(nn.Sequential(*blocks))
        self.stages = nn.Sequential(*stages)
        self.num_features = make_divisible(int(channels[-1] * feat_mult), 8)
        self.final_conv = ScaledStdConv2d(prev, self.num_features, 1)
        self.final_act = nn.SiLU()
        self.head = _Head(self.num_features, num_classes)
        nn.init.normal_(self.head.fc.weight, 0.0, 0.01)
        nn.init.zeros_(self.head.fc.bias)

    def forward_features(self, images):
        (self._root if "_root" in self.__dict__ else self).ensure_arena()
        lib = _lib.load()
        Fn._need_gpu(images, "images")
        images = images.contiguous().to(F32)
        B, C, H, W = images.shape
        x = torch.empty((B * H * W, 8), device=images.device, dtype=BF16)
        check(lib.ia_nchw_to_nhwc_bf16(images.data_ptr(), x.data_ptr(), B, C, H, W, 8, stream_ptr()), "ia_nchw_to_nhwc_bf16")
        f = FeatureMap(x, B, H, W)
        # deep_quad stem: conv1 (s2) act conv2 act conv3 act conv4 (s2).  The stride-1 convs run on the zero-bordered domain,
        # the SiLU in front of each one also converts the layout (compact -> padded -> padded -> compact).
        mods = list(self.stem.children())
        padded, i = False, 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, ScaledStdConv2d):
                if padded and m.stride == 2:                  # conv4: bordered in, compact out (the blocks read compact maps)
                    y = PaddedS2ConvFn.apply(f.t, m.weight, m, f.B, f.H, f.W, True)
                    f, padded = FeatureMap(y, f.B, (f.H - 1) // 2 + 1, (f.W - 1) // 2 + 1), False
                elif padded:
                    f = FeatureMap(PaddedStdConvFn.apply(f.t, m.weight, m, f.B, f.H, f.W), f.B, f.H, f.W)
                else:
                    f = m(f)
            else:
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                to_padded = (isinstance(nxt, ScaledStdConv2d) and (nxt.shifted_views or nxt.strided_direct) and nxt.fits_padded(f.B, f.H, f.W))
                if padded or to_padded:
                    f = FeatureMap(SiluPadFn.apply(f.t, 1.0, f.B, f.H, f.W, padded, to_padded), f.B, f.H, f.W)
                else:
                    f = FeatureMap(SiluFn.apply(f.t, 1.0, False), f.B, f.H, f.W)
                padded = to_padded
            i += 1
        blocks = [blk for stage in self.stages for blk in stage]
        for i, blk in enumerate(blocks):
            f = blk(f, blocks[i + 1] if i + 1 < len(blocks) else None)
        f = self.final_conv(f)
        return FeatureMap(SiluFn.apply(f.t, 1.0, False), f.B, f.H, f.W)

    def forward(self, images):
        return self.head.global_pool(self.forward_features(images))


def create_nfnet(model_name, **kwargs):
    depths, channels, feat_mult = NFNET_CONFIGS[model_name]
    return NormFreeNet(depths, channels, feat_mult)
