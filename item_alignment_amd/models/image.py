"""Host-side mirror of the reference's src/models/image.py plus the ViT encoder the reference takes from
timm (`timm.create_model("vit_base_patch16_384")`, finetune_multimodal.py:223, finetune_image.py:191).

VisionTransformer keeps timm==0.6.5's attribute / state_dict names (cls_token, pos_embed, patch_embed.proj,
blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}, norm, head) and its public methods used by the
reference wrappers: forward_features(x), forward_head(x, pre_logits=True), num_features (image.py:459-462,
multimodal.py:811-812).  The blocks run on the same HIP layer engine as the text tower (pre-LN variant).
"""
import torch
from torch import nn

from . import functional as Fn
from .base import HipModule, SequenceClassifierOutput, TwoTowerClassificationHead, _EngineStack, cls_rows
from .loss import apply_loss, make_loss

VIT_CONFIGS = {
    # timm 0.6.5 model table [third party]: name -> (img_size, patch, embed_dim, depth, heads)
    "vit_base_patch16_384": (384, 16, 768, 12, 12),
    "vit_base_patch16_224": (224, 16, 768, 12, 12),
    "vit_large_patch16_384": (384, 16, 1024, 24, 16),
    "vit_large_patch16_224": (224, 16, 1024, 24, 16),
    "vit_small_patch16_384": (384, 16, 384, 12, 6),
}


class _PatchEmbed(nn.Module):
    def __init__(self, in_chans, embed_dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch, stride=patch)


class _VitAttention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _VitMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _VitBlock(nn.Module):
    def __init__(self, dim, mlp_ratio=4.0, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _VitAttention(dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _VitMlp(dim, int(dim * mlp_ratio))


class VisionTransformer(HipModule, _EngineStack):
    """timm VisionTransformer (global_pool='token', no dropout, qkv_bias) on the HIP engine."""

    def __init__(self, img_size=384, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0):
        super().__init__()
        if embed_dim != 64 * num_heads:
            raise ValueError("head_dim must be 64")
        self.img_size, self.patch_size, self.embed_dim = img_size, patch_size, embed_dim
        self.num_features = embed_dim
        self.num_patches = (img_size // patch_size) ** 2
        self.patch_embed = _PatchEmbed(in_chans, embed_dim, patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.num_patches + 1, embed_dim) * 0.02)
        self.blocks = nn.ModuleList([_VitBlock(embed_dim, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        for p in self.head.parameters():
            p.requires_grad = False            # never reached by the pair step (receives no gradient in the reference either)
        self.hidden_size, self.intermediate_size, self.num_heads = embed_dim, int(embed_dim * mlp_ratio), num_heads
        self.eps, self.hidden_drop, self.attn_drop, self.pre_ln, self.layer_id_base = 1e-6, 0.0, 0.0, True, 500
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    @property
    def layers(self):
        return self.blocks

    def layer_tensors(self, i):
        b = self.blocks[i]
        return dict(qkv_w=b.attn.qkv.weight, qkv_b=b.attn.qkv.bias, w_o=b.attn.proj.weight, b_o=b.attn.proj.bias,
                    ln1_g=b.norm1.weight, ln1_b=b.norm1.bias, w_fc1=b.mlp.fc1.weight, b_fc1=b.mlp.fc1.bias,
                    w_fc2=b.mlp.fc2.weight, b_fc2=b.mlp.fc2.bias, ln2_g=b.norm2.weight, ln2_b=b.norm2.bias)

    def forward_features(self, x):
        """x: [B, 3, S, S] images, or a tuple / list of such batches that are to run as ONE batch in that order (the two items of a pair:
        no concatenated copy of the images is made, Fn.PatchEmbedFn)."""
        (self._root if "_root" in self.__dict__ else self).ensure_arena()
        xs = tuple(x) if isinstance(x, (tuple, list)) else (x,)
        B = sum(t.shape[0] for t in xs)
        for t in xs:
            if t.shape[-1] != self.img_size or t.shape[-2] != self.img_size:
                raise ValueError(f"Input image size ({t.shape[-2]}*{t.shape[-1]}) doesn't match model ({self.img_size}*{self.img_size}).")
        N, H = self.num_patches + 1, self.embed_dim
        tok = Fn.PatchEmbedFn.apply(xs[0], self.anchor, self, *xs[1:])
        outs = Fn.EncoderStackFn.apply(tok, self.anchor, self, None, B, N, torch.is_grad_enabled(), None)
        y = Fn.LayerNormFn.apply(outs[-1], self.anchor, self.norm, 1e-6)
        return y.view(B, N, H)

    def forward_head(self, x, pre_logits=False):
        if not pre_logits:
            raise NotImplementedError("the reference only calls forward_head(x, pre_logits=True) (image.py:460, multimodal.py:812)")
        B, N, H = x.shape
        return Fn.GatherRowsFn.apply(x.reshape(B * N, H), self.anchor, cls_rows(B, N, 0, x.device), 0.0, 0)

    def forward(self, x):
        return self.forward_head(self.forward_features(x), pre_logits=True)


def create_model(model_name, pretrained=False, **kwargs):
    """Stand-in for timm.create_model for the encoders the reference scripts name (no network: pretrained
    weights are loaded later from image_encoder.bin when present)."""
    if model_name in VIT_CONFIGS:
        s, p, d, depth, h = VIT_CONFIGS[model_name]
        return VisionTransformer(img_size=kwargs.get("img_size", s), patch_size=p, embed_dim=d, depth=depth, num_heads=h)
    from .nfnet import NFNET_CONFIGS, create_nfnet
    if model_name in NFNET_CONFIGS:
        return create_nfnet(model_name, **kwargs)
    from .resnetv2 import BIT_CONFIGS, RESNETV2_CONFIGS, create_resnetv2
    if model_name in RESNETV2_CONFIGS or model_name in BIT_CONFIGS:
        return create_resnetv2(model_name, **kwargs)
    raise ValueError(unsupported_encoder_message(model_name))


def supported_image_encoders():
    """every name `create_model` builds on the HIP engine (the reference hands any name to timm, finetune_image.py:191)"""
    from .nfnet import NFNET_CONFIGS
    from .resnetv2 import BIT_CONFIGS, RESNETV2_CONFIGS
    return sorted(VIT_CONFIGS) + sorted(NFNET_CONFIGS) + sorted(RESNETV2_CONFIGS) + sorted(BIT_CONFIGS)


def unsupported_encoder_message(model_name):
    extra = (" (of timm's ResNetV2 family the BatchNorm resnetv2_50 / 101 / 152 and the BiT resnetv2_*_bit* towers are built; the"
             " 'd' / 't' stems and the EvoNorm / FilterResponseNorm variants are not)" if "resnetv2" in model_name else "")
    return f"image encoder {model_name!r} has no HIP tower{extra}; supported: {', '.join(supported_image_encoders())}"


def check_image_encoder_name(parser, model_name):
    """argparse-time check of --model_name / --image_model_name: an unsupported tower ends the run with a usage error that names the
    supported set, before any data is read or any model is built"""
    if model_name not in supported_image_encoders():
        parser.error(unsupported_encoder_message(model_name))


class _ImageTwoTower(HipModule):
    """reference image.py:214-294 (NFNetTwoTower) / :415-499 (VitTwoTower) / :298-378 (ResNetTwoTower)."""
    feature_attr = "num_features"

    def __init__(self, config, image_encoder):
        super().__init__()
        self.config = config
        self.num_labels = config.num_labels
        self.img_encoder = image_encoder
        self.classifier = TwoTowerClassificationHead(self._feat_dim(config, image_encoder), dropout=config.hidden_dropout_prob,
                                                     num_labels=config.num_labels)
        self.loss_fct = make_loss(config)
        from .text import adopt
        adopt(self, image_encoder)

    def _feat_dim(self, config, enc):
        return enc.num_features

    def _embed(self, images):
        raise NotImplementedError

    def forward(self, images_1, images_2, labels=None):
        self.ensure_arena()
        B = images_1.shape[0]
        f = self._embed((images_1, images_2))                            # both towers share weights: one 2B batch (no image concat)
        f1, f2 = f[:B], f[B:]
        training = self.training and torch.is_grad_enabled()
        p = self.classifier.drop_p if training else 0.0
        if p > 0:
            f1, f2 = nn.functional.dropout(f1, p, True), nn.functional.dropout(f2, p, True)
        ce = self.config.loss_type == "ce"
        src, tgt, logits, probs2, loss = self.classifier(f1.contiguous(), f2.contiguous(), labels if ce else None,
                                                         differentiable_logits=(labels is not None and not ce))
        src, tgt, probs = probs2[:, 0], probs2[:, 1], probs2[:, 1]
        if labels is not None and not ce:
            loss = apply_loss(self.loss_fct, self.config, logits, labels, src, tgt)
        return SequenceClassifierOutput(loss=loss, logits=logits, probs=probs, src_embeds=src, tgt_embeds=tgt)


class VitTwoTower(_ImageTwoTower):
    def _feat_dim(self, config, enc):
        return config.hidden_size

    def _embed(self, images):
        return self.img_encoder.forward_head(self.img_encoder.forward_features(images), pre_logits=True)


class NFNetTwoTower(_ImageTwoTower):
    # Operand size is not a limit (the GEMM re-bases its 32-bit buffer windows per workgroup), activation memory is: one 800x800
    # image keeps ~0.8 GiB of activations for backward, so a 288 GB MI355X holds ~300 of them in one pass.  The tower is
    # image-independent (no BatchNorm), so a caller short of memory can run it in chunks of `max_images` 800x800-equivalents
    # (None = one pass; 0 = one image per chunk).
    max_images = None

    def _embed(self, images):
        enc = self.img_encoder
        if isinstance(images, (tuple, list)):          # the conv towers take one tensor (their layout kernel reads it once anyway)
            images = torch.cat(tuple(images), dim=0)
        n = images.shape[0]
        if self.max_images is None:
            return enc.head.global_pool(enc.forward_features(images))
        per = self.max_images * 800 * 800 // max(1, images.shape[-1] * images.shape[-2])
        per = max(1, per)
        if n <= per:
            return enc.head.global_pool(enc.forward_features(images))
        return torch.cat([enc.head.global_pool(enc.forward_features(images[i:i + per])) for i in range(0, n, per)], dim=0)


class ResNetTwoTower(NFNetTwoTower):
    """reference image.py:298-378.  The reference calls the encoder once per tower, so its BatchNorm layers see two batches of
    B images; here both towers are one 2B batch normalised in two segments (models/resnetv2.py)."""

    def _embed(self, images):
        enc = self.img_encoder
        if isinstance(images, (tuple, list)):
            images = torch.cat(tuple(images), dim=0)
        had = getattr(enc, "bn_segments", None)
        if had is not None:
            enc.bn_segments = 2
        try:
            return enc.head.global_pool(enc.forward_features(images)).flatten(1)
        finally:
            if had is not None:
                enc.bn_segments = had
