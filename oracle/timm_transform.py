"""TEST INFRASTRUCTURE — restatement of the image transform the reference builds for every image it loads:

    self.transform = create_transform(input_size=input_size, is_training=is_training, hflip=hflip, color_jitter=color_jitter)
                                                         (reference src/data/data.py:838-841 and :922-925; finetune_image.py:246 passes
                                                          is_training=True for every model except ViT, finetune_multimodal.py:288 True / :333 False)

`create_transform` is timm 0.6.5 (timm/data/transforms_factory.py) [THIRD PARTY, absent offline -> PARITY UNPINNED: restated from the
published source], which composes torchvision transforms on PIL images; those in turn call Pillow, which IS installed, so this
restatement calls the same Pillow primitives in the same order:

  is_training=False   transforms_imagenet_eval(img_size, interpolation='bilinear', crop_pct=None -> 0.875):
                        Resize(int(floor(S / 0.875)), bilinear)   [shorter side, torchvision _compute_resized_output_size]
                        CenterCrop(S) -> ToTensor -> Normalize(IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)
  is_training=True    transforms_imagenet_train(img_size, scale=(0.08, 1.0), ratio=(3/4, 4/3), hflip, color_jitter, interpolation='bilinear'):
                        RandomResizedCropAndInterpolation(S) -> RandomHorizontalFlip(hflip) [if hflip > 0]
                        -> ColorJitter(cj, cj, cj) [if color_jitter is not None; no hue] -> ToTensor -> Normalize

Random draws are made explicit (`TrainParams`) so that a test can hand the same draw to this restatement and to the product.
Only tests/ may import this module.
"""
import math
from collections import namedtuple

import numpy as np
import torch
from PIL import Image, ImageEnhance

DEFAULT_CROP_PCT = 0.875
IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)

# box = (top, left, height, width) of the crop; jitter = None or (order, brightness, contrast, saturation) where order is the
# permutation of (0 brightness, 1 contrast, 2 saturation, 3 hue) torchvision applies them in
TrainParams = namedtuple("TrainParams", "box flip jitter")


def random_resized_crop_params(rng, width, height, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """timm RandomResizedCropAndInterpolation.get_params (rng: random.Random; the reference uses the module-level `random`)."""
    area = width * height
    for _ in range(10):
        target_area = rng.uniform(*scale) * area
        log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
        aspect_ratio = math.exp(rng.uniform(*log_ratio))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if w <= width and h <= height:
            i = rng.randint(0, height - h)
            j = rng.randint(0, width - w)
            return i, j, h, w
    in_ratio = width / height                    # fallback: central crop
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def eval_geometry(width, height, size):
    """torchvision Resize(int) output size (shorter side -> floor(size / 0.875), the longer one truncated) and the CenterCrop
    offsets (Python round, i.e. banker's rounding, as torchvision does).  Returns (new_w, new_h, top, left)."""
    scale_size = int(math.floor(size / DEFAULT_CROP_PCT))
    short, long_ = (width, height) if width <= height else (height, width)
    new_short, new_long = scale_size, int(scale_size * long_ / short)
    new_w, new_h = (new_short, new_long) if width <= height else (new_long, new_short)
    top = int(round((new_h - size) / 2.0))
    left = int(round((new_w - size) / 2.0))
    return new_w, new_h, top, left


def to_tensor_normalized(img):
    a = np.asarray(img, dtype=np.uint8)
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)      # ToTensor
    mean = torch.tensor(IMAGENET_DEFAULT_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_DEFAULT_STD).view(3, 1, 1)
    return (t - mean) / std                                                                            # Normalize


def eval_transform(img, size):
    new_w, new_h, top, left = eval_geometry(img.width, img.height, size)
    img = img.resize((new_w, new_h), Image.BILINEAR)
    img = img.crop((left, top, left + size, top + size))
    return to_tensor_normalized(img)


def train_transform(img, size, params):
    i, j, h, w = params.box
    img = img.crop((j, i, j + w, i + h)).resize((size, size), Image.BILINEAR)       # F.resized_crop
    if params.flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    if params.jitter is not None:
        order, b, c, s = params.jitter
        for fn in order:                                                             # torchvision ColorJitter.forward
            if fn == 0:
                img = ImageEnhance.Brightness(img).enhance(b)
            elif fn == 1:
                img = ImageEnhance.Contrast(img).enhance(c)
            elif fn == 2:
                img = ImageEnhance.Color(img).enhance(s)
    return to_tensor_normalized(img)
