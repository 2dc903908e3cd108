"""Pre-activation ResNetV2 image tower on the HIP engine (reference src/models/image.py:298-378 ResNetTwoTower; the encoder
is timm 0.6.5's `resnetv2_50` = ResNetV2(layers=[3,4,6,3], conv_layer=create_conv2d, norm_layer=BatchNormAct2d), created at
finetune_image.py:191 — timm is absent offline, so the definitions follow timm's published resnetv2.py).

Activations are NHWC bf16 rows [B*H*W, C] (models/nfnet.py FeatureMap): 1x1 convolutions are the bf16 MFMA GEMM (the residual
add of conv3 rides in its epilogue), stride-1 3x3 convolutions run patch-matrix-free on the zero-bordered domain
(ia_conv3x3_padded_*), strided ones through the patch gather; BatchNorm + ReLU, the 7x7 stem gather, MaxPool and the strided
row subsampling are csrc/resnet.hip.  Module / parameter / buffer names are timm's so `image_encoder.bin` loads:
`stem.conv.weight`, `stages.{s}.blocks.{b}.{downsample.conv,conv1,conv2,conv3}.weight`,
`stages.{s}.blocks.{b}.norm{1,2,3}.{weight,bias,running_mean,running_var,num_batches_tracked}`, `norm.*`, `head.fc.*`.

BatchNorm statistics are per forward call in the reference, and ResNetTwoTower calls the encoder once per tower
(image.py:337-341).  The HIP two-tower wrapper runs both towers as one 2B batch, so the encoder normalises in
`bn_segments` = 2 runs of B images and updates the running statistics run by run — the same numbers as two calls.

The BiT variants (`resnetv2_50x3_bitm_in21k` is one of the two names the help text of finetune_image.py:23 gives; timm resnetv2.py
`_create_resnetv2_bit`: stem_type='fixed', conv_layer=StdConv2d(eps=1e-8), norm_layer=GroupNormAct(num_groups=32), channels and
stem scaled by `width_factor`) are the same graph with three pieces exchanged: the weights are standardised per output channel
before every convolution (ia_ws_conv_weight_*, the kernel of the NF-Net tower with gain 1 and scale 1), the norm is per (image,
group) (ia_gn_act_*: no batch statistics, so nothing to segment), and the stem pads its 7x7/2 output with a ring of ZEROS in front
of an unpadded 3x3/2 MaxPool (ia_maxpool3s2_fwd_ex).
"""
import torch
from torch import nn

from .. import _lib, ops
from .._lib import check, ptr, stream_ptr
from . import functional as Fn
from .base import HipModule
from .nfnet import PADDED_CONV, FeatureMap, GapFn, _ws, make_divisible

BF16, F32 = torch.bfloat16, torch.float32

RESNETV2_CONFIGS = {   # timm resnetv2.py: the BatchNorm (non-BiT) variants
    "resnetv2_50": (3, 4, 6, 3),
    "resnetv2_101": (3, 4, 23, 3),
    "resnetv2_152": (3, 8, 36, 3),
}
_BIT_GEOMETRY = {"50x1": ((3, 4, 6, 3), 1), "50x3": ((3, 4, 6, 3), 3), "101x1": ((3, 4, 23, 3), 1), "101x3": ((3, 4, 23, 3), 3),
                 "152x2": ((3, 8, 36, 3), 2), "152x4": ((3, 8, 36, 3), 4)}
BIT_CONFIGS = {}        # timm resnetv2.py: name -> (layers, width_factor, num_classes of the (unused) head)
for _g, (_layers, _wf) in _BIT_GEOMETRY.items():
    BIT_CONFIGS[f"resnetv2_{_g}_bitm"] = (_layers, _wf, 1000)
    BIT_CONFIGS[f"resnetv2_{_g}_bitm_in21k"] = (_layers, _wf, 21843)
BIT_CONFIGS["resnetv2_50x1_bit_distilled"] = ((3, 4, 6, 3), 1, 1000)
BIT_CONFIGS["resnetv2_152x2_bit_teacher"] = ((3, 8, 36, 3), 2, 1000)
BIT_CONFIGS["resnetv2_152x2_bit_teacher_384"] = ((3, 8, 36, 3), 2, 1000)
BIT_STD_EPS, GN_GROUPS = 1e-8, 32


# ---------------------------------------------------------------------------------------------- autograd functions
class BnActFn(torch.autograd.Function):
    """BatchNormAct2d.  With `passthrough` the input is handed back as a second output (the identity shortcut of a
    pre-activation block) so that its gradient is added inside ia_bn_act_bwd instead of by a torch add."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, segments, passthrough):
        lib = _lib.load()
        Fn._need_gpu(x, "feature map")
        x = x.contiguous()
        rows, C = x.shape
        dev = x.device
        training = bn.training
        if rows % segments:
            raise ValueError(f"BatchNorm segments: {rows} rows do not split into {segments} runs")
        y = torch.empty_like(x)
        mean = torch.empty((segments, C), device=dev, dtype=F32)
        rstd = torch.empty((segments, C), device=dev, dtype=F32)
        wsb = lib.ia_bn_act_workspace_bytes(rows, C, segments)
        ws = _ws(dev, wsb)
        check(lib.ia_bn_act_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C, segments, bn.eps, bn.momentum, int(training), 1,
                                ws.data_ptr(), wsb, stream_ptr()), "ia_bn_act_fwd")
        if training:
            bn.num_batches_tracked += segments
        ctx.bn, ctx.saved, ctx.args = bn, (x, mean, rstd), (rows, C, segments, int(training))
        ctx.set_materialize_grads(False)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        lib = _lib.load()
        bn = ctx.bn
        x, mean, rstd = ctx.saved
        rows, C, segments, training = ctx.args
        if dy is None:
            return dpass, None, None, None, None, None
        dy = dy.contiguous()
        dp = None if dpass is None else dpass.contiguous()
        dx = torch.empty_like(x)
        wsb = lib.ia_bn_act_workspace_bytes(rows, C, segments)
        ws = _ws(dy.device, wsb)
        wg = bn.weight.grad.data_ptr() if bn.weight.requires_grad else None
        bg = bn.bias.grad.data_ptr() if bn.bias.requires_grad else None
        check(lib.ia_bn_act_bwd(dy.data_ptr(), x.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ptr(dp),
                                dx.data_ptr(), wg, bg, rows, C, segments, training, 1, ws.data_ptr(), wsb, stream_ptr()), "ia_bn_act_bwd")
        Fn._notify([bn.weight, bn.bias])
        ctx.saved = None
        return dx, None, None, None, None, None


class GnActFn(torch.autograd.Function):
    """GroupNormAct (timm layers/norm_act.py: nn.GroupNorm + ReLU) on NHWC rows of `images` images; `passthrough` as in BnActFn."""

    @staticmethod
    def forward(ctx, x, weight, bias, gn, images, passthrough):
        lib = _lib.load()
        Fn._need_gpu(x, "feature map")
        x = x.contiguous()
        rows, C = x.shape
        dev = x.device
        if rows % images:
            raise ValueError(f"GroupNorm: {rows} rows are not {images} images of equal size")
        y = torch.empty_like(x)
        mean = torch.empty((images, C), device=dev, dtype=F32)
        rstd = torch.empty((images, C), device=dev, dtype=F32)
        wsb = lib.ia_gn_act_workspace_bytes(rows, C, images)
        ws = _ws(dev, wsb)
        check(lib.ia_gn_act_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C, images,
                                gn.num_groups, gn.eps, 1, ws.data_ptr(), wsb, stream_ptr()), "ia_gn_act_fwd")
        ctx.gn, ctx.saved, ctx.args = gn, (x, mean, rstd), (rows, C, images)
        ctx.set_materialize_grads(False)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        lib = _lib.load()
        gn = ctx.gn
        x, mean, rstd = ctx.saved
        rows, C, images = ctx.args
        if dy is None:
            return dpass, None, None, None, None, None
        dy = dy.contiguous()
        dp = None if dpass is None else dpass.contiguous()
        dx = torch.empty_like(x)
        wsb = lib.ia_gn_act_workspace_bytes(rows, C, images)
        ws = _ws(dy.device, wsb)
        wg = gn.weight.grad.data_ptr() if gn.weight.requires_grad else None
        bg = gn.bias.grad.data_ptr() if gn.bias.requires_grad else None
        check(lib.ia_gn_act_bwd(dy.data_ptr(), x.data_ptr(), gn.weight.data_ptr(), gn.bias.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ptr(dp),
                                dx.data_ptr(), wg, bg, rows, C, images, gn.num_groups, 1, ws.data_ptr(), wsb, stream_ptr()), "ia_gn_act_bwd")
        Fn._notify([gn.weight, gn.bias])
        ctx.saved = None
        return dx, None, None, None, None, None


_ONES = {}


def _ones(dev, n):
    """the gain of a plain StdConv2d (ia_ws_conv_weight_* are the ScaledStdConv2d kernels of the NF-Net tower: gain[o] * scale)"""
    key = (str(dev), n)
    if key not in _ONES:
        _ONES[key] = torch.ones(n, device=dev, dtype=F32)
    return _ONES[key]


def _packed_weight(conv, Cgp, ldw):
    """the bf16 GEMM operand [Cout, ldw] (tap-major) of the convolution's weight -- standardised first when the module is a StdConv2d"""
    lib = _lib.load()
    w = conv.weight
    Cout, Cg, k, _ = w.shape
    std_eps = getattr(conv, "std_eps", None)
    if std_eps is None:
        what = torch.empty((Cout, ldw), device=w.device, dtype=BF16)
        check(lib.ia_conv_weight_pack(w.data_ptr(), what.data_ptr(), Cout, Cg, k * k, Cgp, ldw, stream_ptr()), "ia_conv_weight_pack")
        return what
    what = torch.empty((Cout, k * k * Cgp), device=w.device, dtype=BF16)
    mean = torch.empty(Cout, device=w.device, dtype=F32)
    rstd = torch.empty(Cout, device=w.device, dtype=F32)
    check(lib.ia_ws_conv_weight_fwd(w.data_ptr(), _ones(w.device, Cout).data_ptr(), what.data_ptr(), mean.data_ptr(), rstd.data_ptr(), Cout, Cg, k * k,
                                    Cgp, 1.0, std_eps, stream_ptr()), "ia_ws_conv_weight_fwd")
    conv.__dict__["_ws_stats"] = (mean, rstd)      # of the weights as they are now: the same for every call until the optimiser steps
    if ldw > k * k * Cgp:                           # (the stem's 147 columns padded to the GEMM's 8-column granule)
        what = torch.nn.functional.pad(what, (0, ldw - k * k * Cgp))
    return what


def _weight_grad(conv, dwhat, Cgp, ldw):
    lib = _lib.load()
    w = conv.weight
    Cout, Cg, k, _ = w.shape
    if getattr(conv, "std_eps", None) is None:
        check(lib.ia_conv_weight_unpack_grad(dwhat.data_ptr(), w.grad.data_ptr(), Cout, Cg, k * k, Cgp, ldw, stream_ptr()), "ia_conv_weight_unpack_grad")
    else:
        mean, rstd = conv.__dict__["_ws_stats"]
        if ldw > k * k * Cgp:
            dwhat = dwhat[:, :k * k * Cgp].contiguous()
        check(lib.ia_ws_conv_weight_bwd(dwhat.data_ptr(), w.data_ptr(), _ones(w.device, Cout).data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        w.grad.data_ptr(), None, Cout, Cg, k * k, Cgp, 1.0, stream_ptr()), "ia_ws_conv_weight_bwd")
    Fn._notify([w])


class Conv1x1Fn(torch.autograd.Function):
    """1x1 convolution (no bias) on NHWC rows = one GEMM; stride > 1 first picks the rows on the stride grid; `residual` is
    added in the GEMM epilogue (the `x + shortcut` of PreActBottleneck.forward)."""

    @staticmethod
    def forward(ctx, x, weight, conv, B, H, W, residual):
        lib = _lib.load()
        Fn._need_gpu(x, "feature map")
        x = x.contiguous()
        Cin, Cout, s = conv.in_channels, conv.out_channels, conv.stride
        what = _packed_weight(conv, Cin, Cin)
        rows = x
        if s > 1:
            Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
            rows = torch.empty((B * Ho * Wo, Cin), device=x.device, dtype=BF16)
            check(lib.ia_rows_subsample_fwd(x.data_ptr(), rows.data_ptr(), B, H, W, Cin, s, stream_ptr()), "ia_rows_subsample_fwd")
        if residual is not None:
            y = ops.gemm(rows, what, epilogue=ops.EPI_ADD, aux=residual.contiguous())
        else:
            y = ops.gemm(rows, what)
        ctx.conv, ctx.saved, ctx.dims = conv, (rows, what), (B, H, W, Cin, Cout, s)
        ctx.has_res = residual is not None
        ctx.need_dx = ctx.needs_input_grad[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        conv = ctx.conv
        rows, what = ctx.saved
        B, H, W, Cin, Cout, s = ctx.dims
        dy = dy.contiguous()
        dx = None
        if ctx.need_dx:
            d = ops.gemm(dy, what, b_kstrided=True)
            if s > 1:
                dx = torch.empty((B * H * W, Cin), device=dy.device, dtype=BF16)
                check(lib.ia_rows_subsample_bwd(d.data_ptr(), None, dx.data_ptr(), B, H, W, Cin, s, stream_ptr()), "ia_rows_subsample_bwd")
            else:
                dx = d
        if conv.weight.requires_grad:
            dwhat = ops.gemm(dy, rows, a_kstrided=True, b_kstrided=True, out_f32=True)
            _weight_grad(conv, dwhat, Cin, Cin)
        ctx.saved = None
        return dx, None, None, None, None, None, (dy if ctx.has_res else None)


class Conv3x3Fn(torch.autograd.Function):
    """3x3 convolution (no bias, padding 1).  Stride 1 with power-of-two channels: patch-matrix-free on the zero-bordered
    domain (ia_silu_pad-style layout moves are done by ia_pad_rows); otherwise patch gather + GEMM (ia_conv_nhwc_*)."""

    @staticmethod
    def forward(ctx, x, weight, conv, B, H, W):
        lib = _lib.load()
        Fn._need_gpu(x, "feature map")
        x = x.contiguous()
        C, Cout, s = conv.in_channels, conv.out_channels, conv.stride
        dev = x.device
        what = _packed_weight(conv, C, 9 * C)
        padded = (PADDED_CONV and s == 1 and not (C & (C - 1)) and not (Cout & (Cout - 1)) and C >= 8 and Cout >= 8
                  and B * (H + 2) * (W + 2) < 0x7FFFFFFF)
        ctx.padded = padded
        if padded:
            Mp = B * (H + 2) * (W + 2)
            xp = torch.empty((Mp, C), device=dev, dtype=BF16)
            check(lib.ia_pad_rows(x.data_ptr(), xp.data_ptr(), B, H, W, C, 0, 1, stream_ptr()), "ia_pad_rows")
            yp = torch.empty((Mp, Cout), device=dev, dtype=BF16)
            check(lib.ia_conv3x3_padded_fwd(xp.data_ptr(), what.data_ptr(), None, yp.data_ptr(), B, H, W, C, Cout, 1, stream_ptr()),
                  "ia_conv3x3_padded_fwd")
            y = torch.empty((B * H * W, Cout), device=dev, dtype=BF16)
            check(lib.ia_pad_rows(yp.data_ptr(), y.data_ptr(), B, H, W, Cout, 1, 0, stream_ptr()), "ia_pad_rows")
            ctx.saved = (xp, what)
        else:
            Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
            y = torch.empty((B * Ho * Wo, Cout), device=dev, dtype=BF16)
            wsb = lib.ia_conv_nhwc_workspace_bytes(B, H, W, C, Cout, 3, s, 1)
            ws = _ws(dev, wsb)
            check(lib.ia_conv_nhwc_fwd(x.data_ptr(), what.data_ptr(), None, y.data_ptr(), B, H, W, C, Cout, 3, s, 1, ws.data_ptr(), wsb, stream_ptr()),
                  "ia_conv_nhwc_fwd")
            ctx.saved = (x, what)
        ctx.conv, ctx.dims = conv, (B, H, W, C, Cout, s)
        ctx.need_dx = ctx.needs_input_grad[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        conv = ctx.conv
        xs, what = ctx.saved
        B, H, W, C, Cout, s = ctx.dims
        dy = dy.contiguous()
        dev = dy.device
        dx = None
        dwhat = torch.empty((Cout, 9 * C), device=dev, dtype=F32) if conv.weight.requires_grad else None
        if ctx.padded:
            Mp = B * (H + 2) * (W + 2)
            dyp = torch.empty((Mp, Cout), device=dev, dtype=BF16)
            check(lib.ia_pad_rows(dy.data_ptr(), dyp.data_ptr(), B, H, W, Cout, 0, 1, stream_ptr()), "ia_pad_rows")
            if ctx.need_dx:
                dxp = torch.empty((Mp, C), device=dev, dtype=BF16)
                check(lib.ia_conv3x3_padded_bwd_data(dyp.data_ptr(), what.data_ptr(), dxp.data_ptr(), B, H, W, C, Cout, 1, stream_ptr()),
                      "ia_conv3x3_padded_bwd_data")
                dx = torch.empty((B * H * W, C), device=dev, dtype=BF16)
                check(lib.ia_pad_rows(dxp.data_ptr(), dx.data_ptr(), B, H, W, C, 1, 0, stream_ptr()), "ia_pad_rows")
            if dwhat is not None:
                wsb = lib.ia_conv3x3_padded_workspace_bytes(B, H, W, C, Cout, 1)
                ws = _ws(dev, wsb)
                check(lib.ia_conv3x3_padded_bwd_weight(xs.data_ptr(), dyp.data_ptr(), dwhat.data_ptr(), None, B, H, W, C, Cout, 1, ws.data_ptr(), wsb,
                                                       stream_ptr()), "ia_conv3x3_padded_bwd_weight")
        else:
            wsb = lib.ia_conv_nhwc_workspace_bytes(B, H, W, C, Cout, 3, s, 1)
            ws = _ws(dev, wsb)
            if ctx.need_dx:
                dx = torch.empty_like(xs)
                check(lib.ia_conv_nhwc_bwd_data(dy.data_ptr(), what.data_ptr(), dx.data_ptr(), B, H, W, C, Cout, 3, s, 1, ws.data_ptr(), wsb,
                                                stream_ptr()), "ia_conv_nhwc_bwd_data")
            if dwhat is not None:
                check(lib.ia_conv_nhwc_bwd_weight(xs.data_ptr(), dy.data_ptr(), dwhat.data_ptr(), None, B, H, W, C, Cout, 3, s, 1, 0, ws.data_ptr(),
                                                  wsb, stream_ptr()), "ia_conv_nhwc_bwd_weight")
        if dwhat is not None:
            _weight_grad(conv, dwhat, C, 9 * C)
        ctx.saved = None
        return dx, None, None, None, None, None


class StemConvFn(torch.autograd.Function):
    """k x k / stride / symmetric-pad convolution of the NCHW fp32 images (the 7x7/2 stem): patch gather straight from the
    images (ia_patches_nchw) + GEMM.  The images need no gradient."""

    @staticmethod
    def forward(ctx, images, weight, conv):
        lib = _lib.load()
        Fn._need_gpu(images, "images")
        images = images.contiguous().to(F32)
        B, C, H, W = images.shape
        k, s = conv.kernel_size, conv.stride
        pad = ((s - 1) + (k - 1)) // 2                       # timm padding.py get_padding (dilation 1)
        Kp = (k * k * C + 7) // 8 * 8
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        cols = torch.empty((B * Ho * Wo, Kp), device=images.device, dtype=BF16)
        check(lib.ia_patches_nchw(images.data_ptr(), cols.data_ptr(), B, C, H, W, k, s, pad, Kp, stream_ptr()), "ia_patches_nchw")
        what = _packed_weight(conv, C, Kp)
        y = ops.gemm(cols, what)
        ctx.conv, ctx.saved, ctx.Kp = conv, (cols,), Kp
        return y

    @staticmethod
    def backward(ctx, dy):
        conv = ctx.conv
        (cols,) = ctx.saved
        if conv.weight.requires_grad:
            dwhat = ops.gemm(dy.contiguous(), cols, a_kstrided=True, b_kstrided=True, out_f32=True)
            _weight_grad(conv, dwhat, conv.in_channels, ctx.Kp)
        ctx.saved = None
        return None, None, None


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on NHWC rows; pad_zero: ConstantPad2d(1, 0.) + MaxPool2d(3, 2, padding 0)."""

    @staticmethod
    def forward(ctx, x, B, H, W, pad_zero=False):
        lib = _lib.load()
        x = x.contiguous()
        C = x.shape[1]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B * Ho * Wo, C), device=x.device, dtype=BF16)
        arg = torch.empty((B * Ho * Wo, C), device=x.device, dtype=torch.uint8)
        check(lib.ia_maxpool3s2_fwd_ex(x.data_ptr(), y.data_ptr(), arg.data_ptr(), B, H, W, C, int(pad_zero), stream_ptr()), "ia_maxpool3s2_fwd_ex")
        ctx.arg, ctx.dims = arg, (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        B, H, W, C = ctx.dims
        dx = torch.empty((B * H * W, C), device=dy.device, dtype=BF16)
        check(lib.ia_maxpool3s2_bwd(dy.contiguous().data_ptr(), ctx.arg.data_ptr(), dx.data_ptr(), B, H, W, C, stream_ptr()), "ia_maxpool3s2_bwd")
        ctx.arg = None
        return dx, None, None, None, None


# ---------------------------------------------------------------------------------------------- modules
class Conv2d(nn.Module):
    """timm create_conv2d(in, out, k, stride=..) = nn.Conv2d(bias=False, symmetric padding) holding only the weight"""

    std_eps = None

    def __init__(self, in_channels, out_channels, kernel_size, stride=1):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_channels, out_channels, kernel_size, stride
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")           # timm resnetv2.py _init_weights

    def forward(self, f, residual=None):
        s = self.stride
        if self.kernel_size == 1:
            y = Conv1x1Fn.apply(f.t, self.weight, self, f.B, f.H, f.W, residual)
        else:
            y = Conv3x3Fn.apply(f.t, self.weight, self, f.B, f.H, f.W)
        return FeatureMap(y, f.B, (f.H - 1) // s + 1, (f.W - 1) // s + 1)


class StdConv2d(Conv2d):
    """timm layers/std_conv.py StdConv2d(eps=1e-8) as the BiT towers configure it: the weight is standardised per output channel
    (F.batch_norm over its fan-in: biased variance, eps inside the root) in front of every convolution"""
    std_eps = BIT_STD_EPS


class GroupNormAct(nn.GroupNorm):
    """timm layers/norm_act.py GroupNormAct (GroupNorm(32, C, eps 1e-5) + ReLU); statistics are per image: `segments` is unused"""

    def __init__(self, num_channels, num_groups=GN_GROUPS):
        super().__init__(num_groups, num_channels, eps=1e-5, affine=True)

    def forward(self, f, segments=1, passthrough=False):
        out = GnActFn.apply(f.t, self.weight, self.bias, self, f.B, passthrough)
        if passthrough:
            return FeatureMap(out[0], f.B, f.H, f.W), out[1]
        return FeatureMap(out, f.B, f.H, f.W)


class BatchNormAct2d(nn.BatchNorm2d):
    """timm layers/norm_act.py BatchNormAct2d (BatchNorm2d + ReLU, eps 1e-5, momentum 0.1); `segments` as in the module doc"""

    def forward(self, f, segments=1, passthrough=False):
        out = BnActFn.apply(f.t, self.weight, self.bias, self, segments, passthrough)
        if passthrough:
            return FeatureMap(out[0], f.B, f.H, f.W), out[1]
        return FeatureMap(out, f.B, f.H, f.W)


class DownsampleConv(nn.Module):
    """timm resnetv2.py DownsampleConv with preact=True: a strided 1x1 convolution, no norm"""

    def __init__(self, in_chs, out_chs, stride=1, conv_layer=Conv2d):
        super().__init__()
        self.conv = conv_layer(in_chs, out_chs, 1, stride=stride)
        self.norm = nn.Identity()

    def forward(self, f):
        return self.conv(f)


class PreActBottleneck(nn.Module):
    """timm resnetv2.py PreActBottleneck: norm1 -> (shortcut = downsample(preact) | x) -> conv1 -> norm2 -> conv2 (3x3, stride)
    -> norm3 -> conv3 -> + shortcut"""

    def __init__(self, in_chs, out_chs, bottle_ratio=0.25, stride=1, downsample=False, conv_layer=Conv2d, norm_layer=BatchNormAct2d):
        super().__init__()
        mid_chs = make_divisible(out_chs * bottle_ratio)
        self.downsample = DownsampleConv(in_chs, out_chs, stride=stride, conv_layer=conv_layer) if downsample else None
        self.norm1 = norm_layer(in_chs)
        self.conv1 = conv_layer(in_chs, mid_chs, 1)
        self.norm2 = norm_layer(mid_chs)
        self.conv2 = conv_layer(mid_chs, mid_chs, 3, stride=stride)
        self.norm3 = norm_layer(mid_chs)
        self.conv3 = conv_layer(mid_chs, out_chs, 1)

    def forward(self, f, segments=1):
        if self.downsample is not None:
            pre = self.norm1(f, segments)
            shortcut = self.downsample(pre).t
        else:
            pre, shortcut = self.norm1(f, segments, passthrough=True)
        out = self.conv1(pre)
        out = self.conv2(self.norm2(out, segments))
        return self.conv3(self.norm3(out, segments), residual=shortcut)


class ResNetStage(nn.Module):
    def __init__(self, in_chs, out_chs, stride, depth, bottle_ratio=0.25, conv_layer=Conv2d, norm_layer=BatchNormAct2d):
        super().__init__()
        blocks, prev = [], in_chs
        for bi in range(depth):
            blocks.append(PreActBottleneck(prev, out_chs, bottle_ratio, stride=stride if bi == 0 else 1, downsample=bi == 0,
                                           conv_layer=conv_layer, norm_layer=norm_layer))
            prev = out_chs
        self.blocks = nn.Sequential(*blocks)


class _Stem(nn.Module):
    """timm resnetv2.py create_resnetv2_stem(preact=True): stem_type '' = conv 7x7/2 + MaxPool2d(3, 2, padding 1); 'fixed' (BiT) =
    conv 7x7/2 + ConstantPad2d(1, 0.) + MaxPool2d(3, 2, padding 0)"""

    def __init__(self, in_chans, stem_chs, conv_layer=Conv2d, fixed=False):
        super().__init__()
        self.conv = conv_layer(in_chans, stem_chs, 7, stride=2)
        if fixed:
            self.pad = nn.ConstantPad2d(1, 0.0)
        self.pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=0 if fixed else 1)
        self.fixed = fixed


class _GlobalPool(nn.Module):
    def forward(self, f):
        return GapFn.apply(f.t, f.B, f.H * f.W)


class _Head(nn.Module):
    """timm ClassifierHead(use_conv=True): `global_pool` is on the item-alignment path (reference image.py:339); `fc` (a 1x1
    conv in timm) is kept for state_dict parity with timm checkpoints."""

    def __init__(self, num_features, num_classes=1000):
        super().__init__()
        self.global_pool = _GlobalPool()
        self.fc = nn.Conv2d(num_features, num_classes, 1, bias=True)


class ResNetV2(HipModule):
    """forward_features(images [B,3,S,S] fp32) -> FeatureMap [B, S/32, S/32, 2048] NHWC bf16; head.global_pool(map) -> [B, 2048]"""

    def __init__(self, layers, channels=(256, 512, 1024, 2048), num_classes=1000, in_chans=3, stem_chs=64, bottle_ratio=0.25, width_factor=1,
                 bit=False):
        super().__init__()
        conv_layer, norm_layer = (StdConv2d, GroupNormAct) if bit else (Conv2d, BatchNormAct2d)
        stem_chs = make_divisible(stem_chs * width_factor)
        self.stem = _Stem(in_chans, stem_chs, conv_layer, fixed=bit)
        stages, prev = [], stem_chs
        for si, (d, c) in enumerate(zip(layers, channels)):
            c = make_divisible(c * width_factor)
            stages.append(ResNetStage(prev, c, 1 if si == 0 else 2, d, bottle_ratio, conv_layer, norm_layer))
            prev = c
        self.stages = nn.Sequential(*stages)
        self.num_features = prev
        self.norm = norm_layer(prev)
        self.head = _Head(prev, num_classes)
        for p in self.head.parameters():
            p.requires_grad = False            # never reached by the pair step
        nn.init.normal_(self.head.fc.weight, 0.0, 0.01)
        nn.init.zeros_(self.head.fc.bias)
        self.bn_segments = 1                  # see the module docstring; the two-tower wrapper sets 2 around its call

    def forward_features(self, images):
        (self._root if "_root" in self.__dict__ else self).ensure_arena()
        B, _, H, W = images.shape
        seg = self.bn_segments if self.training else 1
        conv = self.stem.conv
        y = StemConvFn.apply(images, conv.weight, conv)
        k, s = conv.kernel_size, conv.stride
        pad = ((s - 1) + (k - 1)) // 2
        H, W = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        f = FeatureMap(MaxPoolFn.apply(y, B, H, W, self.stem.fixed), B, (H - 1) // 2 + 1, (W - 1) // 2 + 1)
        for stage in self.stages:
            for blk in stage.blocks:
                f = blk(f, seg)
        return self.norm(f, seg)

    def forward(self, images):
        return self.head.global_pool(self.forward_features(images))


def create_resnetv2(model_name, **kwargs):
    if model_name in BIT_CONFIGS:
        layers, wf, num_classes = BIT_CONFIGS[model_name]
        return ResNetV2(layers, num_classes=num_classes, width_factor=wf, bit=True)
    return ResNetV2(RESNETV2_CONFIGS[model_name])
