"""GPU side of the image input pipeline.

The reference decodes, transforms and normalises every image inside `__getitem__` on one CPU thread (data.py:838-866: timm
`create_transform`, see data/transforms.py).  Here the host only decodes to uint8 RGB and draws the transform's random parameters;
the resampling (Pillow's own 8-bit separable resampler, bit-identical: csrc/image.hip), the colour jitter (Pillow's ImageEnhance
blends), the /255, mean/std and the horizontal flip run on the GPU.  The coefficient tables are built here exactly as Pillow's
`precompute_coeffs` / `normalize_coeffs_8bpc` (src/libImaging/Resample.c) build them, in double precision, once per
(filter, input length, output length) triple.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from .._lib import check, stream_ptr
from .transforms import IMAGENET_MEAN, IMAGENET_STD, center_crop_geometry

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def _bilinear(x):
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


def precompute_coeffs(in_size, out_size, filt="bicubic"):
    """Pillow Resample.c precompute_coeffs (box = the whole axis) + normalize_coeffs_8bpc, vectorised over the output positions; the
    weight sum runs tap by tap in Pillow's order so the doubles are the same.  Returns (bounds int32 [out, 2] = (first tap, tap
    count), coeffs int32 [out, ksize] in 8.22 fixed point, ksize)."""
    fn, fsupport = FILTERS[filt]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.trunc(center - support + 0.5).astype(np.int64)          # C (int) cast; negative values clamp to 0 anyway
    xmin = np.maximum(xmin, 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    w = fn((x + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):                                             # sequential sum, as Pillow accumulates it
        ww = ww + w[:, t]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = k * float(1 << PRECISION_BITS)
    kk = np.where(k < 0, np.trunc(-0.5 + fixed), np.trunc(0.5 + fixed)).astype(np.int32)
    kk = np.where(x < xmax[:, None], kk, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk, ksize


class GpuImagePipeline:
    """Decoded uint8 RGB frames -> normalised fp32 [B, 3, S, S] on the GPU.

    __call__(frames [B, H, W, 3], flip)   plain resize of equally sized frames to S x S (filter = `filt`) + normalisation
    process(items)                         RawImage items (data/datasets.py) with the transform's parameters: evaluation items go
                                           through Resize(floor(S / 0.875), bilinear) + CenterCrop(S) batched per frame size, training
                                           items through their own RandomResizedCrop window, then flip / colour jitter / normalisation
    """

    def __init__(self, size, device, mean=IMAGENET_MEAN, std=IMAGENET_STD, filt="bilinear"):
        self.size, self.device, self.filt = int(size), torch.device(device), filt
        self._tables = {}
        self._mean = (C.c_float * 3)(*mean)
        self._std = (C.c_float * 3)(*std)

    def _table(self, n_in, n_out=None, lo=0, count=None):
        """device tables for resizing n_in -> n_out, restricted to the output positions [lo, lo + count)"""
        n_out = self.size if n_out is None else n_out
        count = n_out if count is None else count
        key = (n_in, n_out, lo, count)
        t = self._tables.get(key)
        if t is None:
            b, k, ks = precompute_coeffs(n_in, n_out, self.filt)
            t = (torch.from_numpy(np.ascontiguousarray(b[lo:lo + count])).to(self.device),
                 torch.from_numpy(np.ascontiguousarray(k[lo:lo + count])).to(self.device), ks)
            if len(self._tables) > 4096:
                self._tables.clear()
            self._tables[key] = t
        return t

    def _pass(self, src_ptr, pitch, frame, dst, table, B, in_len, out_len, other_len, horizontal):
        b, k, ks = table
        check(_lib.load().ia_resize_pass_u8_ex(src_ptr, pitch, frame, dst.data_ptr(), b.data_ptr(), k.data_ptr(), ks, B, in_len, out_len, other_len,
                                               int(horizontal), stream_ptr()), "ia_resize_pass_u8_ex")

    def resize(self, frames):
        """Bit-identical to PIL Image.resize((S, S), filter) per frame; returns uint8 [B, S, S, 3] on the device."""
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError("frames must be uint8 [B, H, W, 3]")
        x = frames.to(self.device, non_blocking=True).contiguous()
        B, H, W, _ = x.shape
        S = self.size
        if W != S:                                        # Pillow: horizontal pass first, over every input row
            y = torch.empty((B, H, S, 3), device=self.device, dtype=torch.uint8)
            self._pass(x.data_ptr(), W * 3, H * W * 3, y, self._table(W), B, W, S, H, True)
            x = y
        if H != S:
            y = torch.empty((B, S, S, 3), device=self.device, dtype=torch.uint8)
            self._pass(x.data_ptr(), 0, 0, y, self._table(H), B, H, S, S, False)
            x = y
        return x

    def _normalize(self, x, flip):
        lib = _lib.load()
        B, S = x.shape[0], self.size
        out = torch.empty((B, 3, S, S), device=self.device, dtype=torch.float32)
        f = None if flip is None else torch.as_tensor(flip, dtype=torch.uint8).to(self.device).contiguous()
        check(lib.ia_u8_to_nchw_normalized(x.data_ptr(), None if f is None else f.data_ptr(), out.data_ptr(), B, S, S, self._mean, self._std,
                                           stream_ptr()), "ia_u8_to_nchw_normalized")
        return out

    def __call__(self, frames, flip=None):
        """resize -> /255 -> (x - mean)/std (-> horizontal flip where flip[b] != 0): fp32 [B, 3, S, S]."""
        return self._normalize(self.resize(frames), flip)

    # ---------------------------------------------------------------- the reference's transform (data/transforms.py) on the GPU
    def _eval_group(self, frames, out):
        """Resize(floor(S / 0.875)) + CenterCrop(S) of equally sized frames [n, H, W, 3]: only the cropped window is computed (every
        output sample depends on its own taps only, so the window of the resize IS the crop of the resize)."""
        n, H, W, _ = frames.shape
        S = self.size
        new_w, new_h, top, left = center_crop_geometry(W, H, S)
        x = frames.to(self.device, non_blocking=True).contiguous()
        mid = torch.empty((n, H, S, 3), device=self.device, dtype=torch.uint8)
        self._pass(x.data_ptr(), W * 3, H * W * 3, mid, self._table(W, new_w, left, S), n, W, S, H, True)
        self._pass(mid.data_ptr(), 0, 0, out, self._table(H, new_h, top, S), n, H, S, S, False)

    def _train_one(self, frame, box, out):
        """RandomResizedCrop window `box` (top, left, h, w) of one frame [H, W, 3] -> out [S, S, 3]"""
        H, W, _ = frame.shape
        top, left, h, w = box
        S = self.size
        x = frame.to(self.device, non_blocking=True).contiguous()
        src = x.data_ptr() + (top * W + left) * 3
        mid = torch.empty((h, S, 3), device=self.device, dtype=torch.uint8)
        self._pass(src, W * 3, 0, mid, self._table(w, S), 1, w, S, h, True)
        self._pass(mid.data_ptr(), 0, 0, out, self._table(h, S), 1, h, S, S, False)
        return x                      # keeps the source alive until the stream has consumed it (caller holds the reference)

    def process(self, items):
        """RawImage items -> fp32 [N, 3, S, S]"""
        lib = _lib.load()
        N, S = len(items), self.size
        u8 = torch.empty((N, S, S, 3), device=self.device, dtype=torch.uint8)
        keep = []
        groups = {}
        for i, it in enumerate(items):
            if it.params.train:
                keep.append(self._train_one(it.u8, it.params.box, u8[i]))
            else:
                groups.setdefault(tuple(it.u8.shape), []).append(i)
        for idxs in groups.values():
            frames = torch.stack([items[i].u8 for i in idxs])
            if len(idxs) == N:
                self._eval_group(frames, u8)
            else:
                tmp = torch.empty((len(idxs), S, S, 3), device=self.device, dtype=torch.uint8)
                self._eval_group(frames, tmp)
                u8[torch.as_tensor(idxs, device=self.device)] = tmp
        # colour jitter: one launch per position of the random op order, every image applying its own op there
        if any(it.params.jitter is not None for it in items):
            scratch = torch.empty(N, device=self.device, dtype=torch.int64)
            for pos in range(4):
                ops, factors = np.zeros(N, dtype=np.int32), np.ones(N, dtype=np.float32)
                for i, it in enumerate(items):
                    j = it.params.jitter
                    if j is not None and j[0][pos] < 3:
                        ops[i] = j[0][pos] + 1
                        factors[i] = np.float32(j[1 + j[0][pos]])
                if not ops.any():
                    continue
                o, f = torch.from_numpy(ops).to(self.device), torch.from_numpy(factors).to(self.device)
                check(lib.ia_color_jitter_step_u8(u8.data_ptr(), o.data_ptr(), f.data_ptr(), scratch.data_ptr(), N, S, S, int((ops == 2).any()),
                                                  stream_ptr()), "ia_color_jitter_step_u8")
                keep += [o, f]
        out = self._normalize(u8, [int(it.params.flip) for it in items])
        del keep
        return out
