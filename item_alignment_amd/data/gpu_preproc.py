"""GPU side of the image input pipeline.

The reference decodes, resizes and normalises every image inside `__getitem__` on one CPU thread (data.py:838-866: timm
`create_transform(is_training=False)` = PIL bicubic resize -> ToTensor -> Normalize).  Here the host only decodes to uint8
RGB; the resize (Pillow's own 8-bit separable resampler, bit-identical: csrc/image.hip), the /255, mean/std and the optional
horizontal flip run on the GPU over the whole batch.  The coefficient tables are built here exactly as Pillow's
`precompute_coeffs` / `normalize_coeffs_8bpc` (src/libImaging/Resample.c) build them, in double precision, once per
(input length, output length) pair.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from .._lib import check, stream_ptr

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Pillow Resample.c precompute_coeffs (box = the whole axis, bicubic, support 2) + normalize_coeffs_8bpc.
    Returns (bounds int32 [out,2], coeffs int32 [out,ksize], ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


class GpuImagePipeline:
    """uint8 RGB frames [B, H, W, 3] (device or host tensor) -> normalised fp32 [B, 3, S, S] on the GPU."""

    def __init__(self, size, device, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.size, self.device = int(size), torch.device(device)
        self._tables = {}
        self._mean = (C.c_float * 3)(*mean)
        self._std = (C.c_float * 3)(*std)

    def _table(self, n_in):
        t = self._tables.get(n_in)
        if t is None:
            b, k, ks = precompute_coeffs(n_in, self.size)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), ks)
            self._tables[n_in] = t
        return t

    def resize(self, frames):
        """Bit-identical to PIL Image.resize((S, S), Image.BICUBIC) per frame; returns uint8 [B, S, S, 3] on the device."""
        lib = _lib.load()
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError("frames must be uint8 [B, H, W, 3]")
        x = frames.to(self.device, non_blocking=True).contiguous()
        B, H, W, _ = x.shape
        S = self.size
        if W != S:                                        # Pillow: horizontal pass first, over every input row
            b, k, ks = self._table(W)
            y = torch.empty((B, H, S, 3), device=self.device, dtype=torch.uint8)
            check(lib.ia_resize_pass_u8(x.data_ptr(), y.data_ptr(), b.data_ptr(), k.data_ptr(), ks, B, W, S, H, 1, stream_ptr()), "ia_resize_pass_u8[h]")
            x = y
        if H != S:
            b, k, ks = self._table(H)
            y = torch.empty((B, S, S, 3), device=self.device, dtype=torch.uint8)
            check(lib.ia_resize_pass_u8(x.data_ptr(), y.data_ptr(), b.data_ptr(), k.data_ptr(), ks, B, H, S, S, 0, stream_ptr()), "ia_resize_pass_u8[v]")
            x = y
        return x

    def __call__(self, frames, flip=None):
        """resize -> /255 -> (x - mean)/std (-> horizontal flip where flip[b] != 0): fp32 [B, 3, S, S]."""
        lib = _lib.load()
        x = self.resize(frames)
        B, S = x.shape[0], self.size
        out = torch.empty((B, 3, S, S), device=self.device, dtype=torch.float32)
        f = None if flip is None else torch.as_tensor(flip, dtype=torch.uint8).to(self.device).contiguous()
        check(lib.ia_u8_to_nchw_normalized(x.data_ptr(), None if f is None else f.data_ptr(), out.data_ptr(), B, S, S, self._mean, self._std,
                                           stream_ptr()), "ia_u8_to_nchw_normalized")
        return out
