"""The image transform of the reference's datasets (src/data/data.py:838-841, :922-925):

    create_transform(input_size=S, is_training=..., hflip=..., color_jitter=...)          # timm 0.6.5

which is torchvision's Resize(floor(S / 0.875), bilinear) + CenterCrop(S) for evaluation and RandomResizedCrop(S, scale
(0.08, 1), ratio (3/4, 4/3), bilinear) + RandomHorizontalFlip(hflip) + ColorJitter(cj, cj, cj) for training, each followed by
ToTensor + Normalize(ImageNet mean / std).  finetune_image.py:246 trains every model except ViT with is_training=True;
finetune_multimodal.py:288 / :333 use True for the training set and False for validation / test.

Two executions of the same arithmetic:
  * host:  Pillow, inside __getitem__ (what the reference does);
  * GPU (--gpu_preproc):  the host only decodes and draws the random parameters (ImageParams travel with the frame), the
    resampling / jitter / normalisation run in csrc/image.hip, bit-identical to Pillow (tests/test_kernels_gpu.py).
Random numbers come from the global `random` module, like the reference's crop (its flip / jitter use torch's global generator:
same distributions, not the same stream): DataLoader reseeds the global generators in every worker process and every epoch
(base_seed + worker_id), so workers draw different augmentations and no epoch repeats another; train.py mixes the rank in.  A
private random.Random(seed) is used only when a seed is passed (tests hand the host path, the GPU path and the restatement the
same draw) -- a private generator inside the dataset object would be pickled to every worker in the same state.
"""
import math
import random
from collections import namedtuple

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
CROP_PCT = 0.875

# train: box = (top, left, height, width) crop, flip, jitter = None | (order of ops 0..3, brightness, contrast, saturation)
# eval:  box = None
ImageParams = namedtuple("ImageParams", "train box flip jitter")


def center_crop_geometry(width, height, size):
    """(new_w, new_h, top, left): shorter side resized to floor(size / 0.875) keeping the aspect ratio (the longer side truncated),
    then a centred size x size window (offsets rounded half to even, like torchvision's int(round(...)))."""
    target = int(math.floor(size / CROP_PCT))
    if width <= height:
        new_w, new_h = target, int(target * height / width)
    else:
        new_w, new_h = int(target * width / height), target
    return new_w, new_h, int(round((new_h - size) / 2.0)), int(round((new_w - size) / 2.0))


class ImageTransform:
    def __init__(self, size, is_training, hflip=0.5, color_jitter=None, seed=None):
        self.size, self.train, self.hflip = int(size), bool(is_training), float(hflip or 0.0)
        self.cj = None if color_jitter is None else float(color_jitter)
        self._seeded = None if seed is None else random.Random(seed)

    @property
    def rng(self):
        return self._seeded if self._seeded is not None else random

    # ---- random parameters
    def _crop_box(self, width, height):
        """RandomResizedCrop: ten attempts at an area in [8 %, 100 %] with log-uniform aspect in [3/4, 4/3], else the central crop"""
        rng, area = self.rng, width * height
        lo, hi = math.log(3.0 / 4.0), math.log(4.0 / 3.0)
        for _ in range(10):
            target = rng.uniform(0.08, 1.0) * area
            aspect = math.exp(rng.uniform(lo, hi))
            w, h = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if w <= width and h <= height:
                top = rng.randint(0, height - h)
                left = rng.randint(0, width - w)
                return top, left, h, w
        ratio = width / height
        if ratio < 3.0 / 4.0:
            w, h = width, int(round(width / (3.0 / 4.0)))
        elif ratio > 4.0 / 3.0:
            w, h = int(round(height * (4.0 / 3.0))), height
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def draw(self, width, height):
        if not self.train:
            return ImageParams(False, None, False, None)
        box = self._crop_box(width, height)
        flip = self.hflip > 0.0 and self.rng.random() < self.hflip
        jitter = None
        if self.cj is not None:
            order = [0, 1, 2, 3]
            self.rng.shuffle(order)
            lo = max(0.0, 1.0 - self.cj)
            jitter = (tuple(order), self.rng.uniform(lo, 1.0 + self.cj), self.rng.uniform(lo, 1.0 + self.cj), self.rng.uniform(lo, 1.0 + self.cj))
        return ImageParams(True, box, flip, jitter)

    # ---- host execution (Pillow)
    def apply(self, img, params):
        from PIL import Image, ImageEnhance
        S = self.size
        if not params.train:
            new_w, new_h, top, left = center_crop_geometry(img.width, img.height, S)
            img = img.resize((new_w, new_h), Image.BILINEAR).crop((left, top, left + S, top + S))
        else:
            top, left, h, w = params.box
            img = img.crop((left, top, left + w, top + h)).resize((S, S), Image.BILINEAR)
            if params.flip:
                img = img.transpose(Image.FLIP_LEFT_RIGHT)
            if params.jitter is not None:
                order, b, c, s = params.jitter
                enhancers = {0: (ImageEnhance.Brightness, b), 1: (ImageEnhance.Contrast, c), 2: (ImageEnhance.Color, s)}
                for op in order:
                    if op in enhancers:
                        cls, factor = enhancers[op]
                        img = cls(img).enhance(factor)
        a = torch.from_numpy(np.ascontiguousarray(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1))).to(torch.float32).div(255)
        return (a - torch.tensor(IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(IMAGENET_STD).view(3, 1, 1)

    def __call__(self, img):
        return self.apply(img, self.draw(img.width, img.height))
