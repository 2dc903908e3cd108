"""TEST INFRASTRUCTURE — second set of golden vectors (round 2), kept apart from gen_golden.py so that the first set's random
draws (and therefore its committed fixtures) stay bit-identical.  Build container only (needs /root/reference, read-only).

    python oracle/gen_golden_r2.py [wrappers] [vit_hf] [deep] [bit_hf] [bit_wrapper]      (default: the first three)

wrappers  the reference's OWN image two-tower classes — NFNetTwoTower (src/models/image.py:212-294), ResNetTwoTower (:298-378),
          VitTwoTower (:418-499) — instantiated with an encoder module that evaluates the oracle's restatement of the timm tower
          (timm 0.6.5 is absent offline).  This pins everything the reference owns on that path: which encoder methods are called
          (forward_features / head.global_pool / .flatten(1) / forward_head(pre_logits=True)), the pair head, the probs[:, 0] /
          probs[:, 1] reuse as src/tgt embeds, the loss.  The tower arithmetic itself stays "parity unpinned by the reference".
vit_hf    cross-check of the oracle's ViT restatement against an independent third-party implementation that IS installed:
          transformers.ViTModel (eager attention, layer_norm_eps 1e-6) with the same seeded weights under HF's key names.
          A cross-check between two restatements of the same public architecture, not a pin by the reference.
bit_hf / bit_wrapper   (round 6) the same two kinds of evidence for the BiT ResNetV2 towers (GroupNorm + StdConv2d + 'fixed' stem):
          transformers.BitModel as the independent implementation, the reference's ResNetTwoTower as the wrapper.
deep      roberta_large.json geometry with all 24 layers through the reference's RobertaModel, B = 2, L = 510 (config C2
          shapes): strided subsamples of the hidden states after layers 1, 6, 12, 18, 24, so the bf16 engine's drift over depth
          is bounded against the fp32 reference.
"""
import os
import sys
import warnings
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_harness import load_reference, reference_config  # noqa: E402
from oracle.weights import seeded_state_dict  # noqa: E402
from oracle import ref_models as O  # noqa: E402
from oracle.gen_golden import save, load_weights, t, text_batch  # noqa: E402

warnings.filterwarnings("ignore")
GOLDEN = os.path.join(ROOT, "tests", "golden")


class OracleEncoder(torch.nn.Module):
    """A timm-shaped image encoder whose arithmetic is the oracle's functional restatement (parameters under timm's key names)."""

    def __init__(self, kind, tcfg, spec, num_features, stats=None):
        super().__init__()
        self.kind, self.tcfg, self.num_features, self.stats = kind, tcfg, num_features, stats
        self.params = torch.nn.ParameterDict()
        self._keys = {}
        for k, shp in spec:
            assert k.startswith("e."), k
            pk = k[2:].replace(".", "__")
            self.params[pk] = torch.nn.Parameter(torch.zeros(shp))
            self._keys[k] = pk
        outer = self

        class Head(torch.nn.Module):
            def global_pool(self, x):
                # timm SelectAdaptivePool2d('avg'): NormFreeNet's ClassifierHead flattens, ResNetV2's does not (the reference
                # adds .flatten(1) itself, image.py:339)
                return x.mean((2, 3)) if outer.kind == "nfnet" else x.mean((2, 3), keepdim=True)
        self.head = Head()

    def _sd(self):
        return {k: self.params[pk] for k, pk in self._keys.items()}

    def forward_features(self, x):
        if self.kind == "nfnet":
            return O.nfnet_forward_features(self._sd(), "e", self.tcfg, x)
        if self.kind == "resnet":
            return O.resnetv2_forward_features(self._sd(), "e", self.tcfg, x, self.training, self.stats)
        return O.vit_forward_features(self._sd(), "e", self.tcfg, x)

    def forward_head(self, x, pre_logits=False):
        assert self.kind == "vit" and pre_logits
        return O.vit_forward_head(x)


def vit_spec(prefix, D, depth, P, npatch, mlp=4):
    spec = [("cls_token", (1, 1, D)), ("pos_embed", (1, npatch + 1, D)), ("patch_embed.proj.weight", (D, 3, P, P)), ("patch_embed.proj.bias", (D,))]
    for i in range(depth):
        b = f"blocks.{i}"
        spec += [(b + ".norm1.weight", (D,)), (b + ".norm1.bias", (D,)), (b + ".attn.qkv.weight", (3 * D, D)), (b + ".attn.qkv.bias", (3 * D,)),
                 (b + ".attn.proj.weight", (D, D)), (b + ".attn.proj.bias", (D,)), (b + ".norm2.weight", (D,)), (b + ".norm2.bias", (D,)),
                 (b + ".mlp.fc1.weight", (mlp * D, D)), (b + ".mlp.fc1.bias", (mlp * D,)), (b + ".mlp.fc2.weight", (D, mlp * D)),
                 (b + ".mlp.fc2.bias", (D,))]
    spec += [("norm.weight", (D,)), ("norm.bias", (D,))]
    return [(prefix + "." + k, s) for k, s in spec]


def wrappers(M):
    rs = np.random.RandomState(4242)
    B = 3
    ncfg = SimpleNamespace(depths=(1, 2, 1, 1), channels=(256, 512, 512, 512), stem_chs=128, group_size=64, bottle_ratio=0.25,
                           num_features=512, alpha=0.2, attn_gain=2.0, eps=1e-5, ch_div=8)
    rcfg = SimpleNamespace(layers=(1, 2, 1, 1), channels=(64, 128, 256, 256), stem_chs=32, bottle_ratio=0.25, eps=1e-5, momentum=0.1,
                           num_features=256)
    vcfg = SimpleNamespace(embed_dim=128, depth=2, num_heads=2, patch_size=16, eps=1e-6, image_size=64)
    towers = [
        ("nfnet_two_tower", M.NFNetTwoTower, "nfnet", ncfg, O.nfnet_state_spec(ncfg, prefix="e"), 512, None, 64,
         ["stem.conv1.weight", "stages.1.1.conv2b.weight", "final_conv.weight"]),
        ("resnet_two_tower", M.ResNetTwoTower, "resnet", rcfg, O.resnetv2_state_spec(rcfg, prefix="e"), 256, 0.08, 96,
         ["stem.conv.weight", "stages.3.blocks.0.conv3.weight", "norm.weight"]),
        ("vit_two_tower", M.VitTwoTower, "vit", vcfg, vit_spec("e", 128, 2, 16, 16), 128, None, 64,
         ["blocks.0.attn.qkv.weight", "patch_embed.proj.weight", "pos_embed"]),
    ]
    for name, cls, kind, tcfg, spec, nf, scale, size, gnames in towers:
        # only "ce" runs in the reference here: its cosine branch hands 1-D probs columns to CosineEmbeddingLoss (image.py:276 raises)
        for lt in ["ce"]:
            cfg = reference_config(hidden_size=nf, num_labels=2, hidden_dropout_prob=0.1, interaction_type="two_tower", loss_type=lt,
                                   loss_margin=0.3)
            stats = O.resnetv2_running_stats(tcfg, "e") if kind == "resnet" else None
            enc = OracleEncoder(kind, tcfg, spec, nf, stats)
            model = cls(cfg, enc).eval()
            seed = 51
            full_spec = [("img_encoder." + k[2:], s) for k, s in spec] + [("classifier.out_proj.weight", (2, 2 * nf)), ("classifier.out_proj.bias", (2,))]
            sd = seeded_state_dict(full_spec, seed) if scale is None else seeded_state_dict(full_spec, seed, scale=scale)
            with torch.no_grad():
                for k, v in sd.items():
                    if k.startswith("img_encoder."):
                        enc.params[k[len("img_encoder."):].replace(".", "__")].copy_(v)
                model.classifier.out_proj.weight.copy_(sd["classifier.out_proj.weight"])
                model.classifier.out_proj.bias.copy_(sd["classifier.out_proj.bias"])
            im1 = rs.standard_normal((B, 3, size, size)).astype(np.float32)
            im2 = (rs.standard_normal((B, 3, size, size)) * 1.3 + 0.2).astype(np.float32)
            labels = np.array([1, 0, 1], dtype=np.int64)
            out = model(t(im1), t(im2), t(labels))
            out.loss.backward()
            grads = {"classifier.out_proj.weight": model.classifier.out_proj.weight.grad}
            for g in gnames:
                grads["img_encoder." + g] = enc.params[g.replace(".", "__")].grad
            case = name if lt == "ce" else f"{name}_{lt}"
            save(case, cfg, seed, full_spec, dict(images_1=im1, images_2=im2, labels=labels), out, grads,
                 extra=dict(seed_scale=np.array(-1.0 if scale is None else scale, dtype=np.float32)))


def vit_hf():
    """the oracle's ViT restatement (timm layout) against transformers.ViTModel with the same weights"""
    from transformers import ViTConfig, ViTModel
    D, depth, nh, P, S = 128, 3, 2, 16, 64
    npatch = (S // P) ** 2
    spec = vit_spec("v", D, depth, P, npatch)
    sd = seeded_state_dict(spec, 61)
    hf_cfg = ViTConfig(hidden_size=D, num_hidden_layers=depth, num_attention_heads=nh, intermediate_size=4 * D, hidden_act="gelu",
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-6, image_size=S, patch_size=P,
                       num_channels=3, qkv_bias=True)
    hf_cfg._attn_implementation = "eager"
    hf = ViTModel(hf_cfg, add_pooling_layer=False).eval()
    m = {"embeddings.cls_token": sd["v.cls_token"], "embeddings.position_embeddings": sd["v.pos_embed"],
         "embeddings.patch_embeddings.projection.weight": sd["v.patch_embed.proj.weight"],
         "embeddings.patch_embeddings.projection.bias": sd["v.patch_embed.proj.bias"],
         "layernorm.weight": sd["v.norm.weight"], "layernorm.bias": sd["v.norm.bias"]}
    for i in range(depth):
        b, h = f"v.blocks.{i}", f"layers.{i}"          # key names of transformers 5.x ViTModel
        qw, kw, vw = sd[b + ".attn.qkv.weight"].chunk(3, 0)
        qb, kb, vb = sd[b + ".attn.qkv.bias"].chunk(3, 0)
        m.update({h + ".attention.q_proj.weight": qw, h + ".attention.q_proj.bias": qb, h + ".attention.k_proj.weight": kw,
                  h + ".attention.k_proj.bias": kb, h + ".attention.v_proj.weight": vw, h + ".attention.v_proj.bias": vb,
                  h + ".attention.o_proj.weight": sd[b + ".attn.proj.weight"], h + ".attention.o_proj.bias": sd[b + ".attn.proj.bias"],
                  h + ".layernorm_before.weight": sd[b + ".norm1.weight"], h + ".layernorm_before.bias": sd[b + ".norm1.bias"],
                  h + ".layernorm_after.weight": sd[b + ".norm2.weight"], h + ".layernorm_after.bias": sd[b + ".norm2.bias"],
                  h + ".mlp.fc1.weight": sd[b + ".mlp.fc1.weight"], h + ".mlp.fc1.bias": sd[b + ".mlp.fc1.bias"],
                  h + ".mlp.fc2.weight": sd[b + ".mlp.fc2.weight"], h + ".mlp.fc2.bias": sd[b + ".mlp.fc2.bias"]})
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    rs = np.random.RandomState(62)
    images = rs.standard_normal((2, 3, S, S)).astype(np.float32)
    with torch.no_grad():
        want = hf(pixel_values=torch.from_numpy(images)).last_hidden_state
    vcfg = SimpleNamespace(embed_dim=D, depth=depth, num_heads=nh, patch_size=P, eps=1e-6)
    got = O.vit_forward_features(sd, "v", vcfg, torch.from_numpy(images))
    err = (got - want).abs().max().item()
    print(f"vit_hf_crosscheck: oracle vs transformers.ViTModel max |diff| = {err:.2e}")
    assert err < 1e-4
    import json
    meta = dict(case="vit_hf_crosscheck", seed=61, config=dict(embed_dim=D, depth=depth, num_heads=nh, patch_size=P, image_size=S),
                spec=[[k, list(s)] for k, s in spec], transformers=__import__("transformers").__version__)
    np.savez_compressed(os.path.join(GOLDEN, "vit_hf_crosscheck.npz"), in_images=images, out_tokens=want.numpy(),
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print("wrote vit_hf_crosscheck.npz")


NARROW_BIT = SimpleNamespace(layers=(1, 2, 1, 1), channels=(128, 256, 256, 512), stem_chs=32, bottle_ratio=0.25, eps=1e-5, momentum=0.1,
                             num_features=512, bit=True, std_eps=1e-8, groups=32)


def bit_hf():
    """the oracle's BiT restatement (timm resnetv2.py `_create_resnetv2_bit` layout and key names) against transformers.BitModel -- an
    independent implementation of the same published architecture -- with the same seeded weights: features, pooled output and the
    gradients of a weighted sum of the pooled output"""
    from transformers import BitConfig, BitModel
    c = NARROW_BIT
    spec = O.resnetv2_state_spec(c, prefix="e")
    sd = seeded_state_dict(spec, 83, scale=0.08)
    hf_cfg = BitConfig(num_channels=3, embedding_size=c.stem_chs, hidden_sizes=list(c.channels), depths=list(c.layers), layer_type="preactivation",
                       hidden_act="relu", global_padding=None, num_groups=c.groups, drop_path_rate=0.0, embedding_dynamic_padding=False,
                       output_stride=32, width_factor=1)
    hf = BitModel(hf_cfg).eval()
    m = {"embedder.convolution.weight": sd["e.stem.conv.weight"], "norm.weight": sd["e.norm.weight"], "norm.bias": sd["e.norm.bias"]}
    for k, v in sd.items():
        if k.startswith("e.stages."):
            _, _, si, _, bi, rest = k.split(".", 5)
            m[f"encoder.stages.{si}.layers.{bi}.{rest}"] = v
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    rs = np.random.RandomState(84)
    images = rs.standard_normal((3, 3, 96, 96)).astype(np.float32)
    wts = rs.standard_normal((3, c.num_features)).astype(np.float32)
    gnames = ["stem.conv.weight", "stages.0.blocks.0.downsample.conv.weight", "stages.0.blocks.0.norm1.weight", "stages.1.blocks.0.conv2.weight",
              "stages.1.blocks.1.norm2.bias", "stages.2.blocks.0.conv1.weight", "stages.3.blocks.0.conv3.weight", "norm.weight", "norm.bias"]
    hf_params = dict(hf.named_parameters())
    out = hf(pixel_values=torch.from_numpy(images))
    (out.pooler_output.flatten(1) * torch.from_numpy(wts)).sum().backward()
    want_feat, want_pool = out.last_hidden_state.detach(), out.pooler_output.flatten(1).detach()
    hf_key = lambda k: ("embedder.convolution.weight" if k == "stem.conv.weight" else k if k.startswith("norm.") else
                        "encoder.stages.{}.layers.{}.{}".format(*(lambda q: (q[1], q[3], q[4]))(k.split(".", 4))))
    want_grads = {k: hf_params[hf_key(k)].grad.detach() for k in gnames}
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feat = O.resnetv2_forward_features(ref, "e", c, torch.from_numpy(images))
    (feat.mean((2, 3)) * torch.from_numpy(wts)).sum().backward()
    err = (feat.detach() - want_feat).abs().max().item() / want_feat.abs().max().item()
    print(f"bit_hf_crosscheck: oracle vs transformers.BitModel max |diff| / max |.| = {err:.2e}")
    assert err < 1e-4
    for k in gnames:
        g, w = ref["e." + k].grad, want_grads[k]
        e = (g - w).norm().item() / w.norm().item()
        print(f"  grad {k}: rel {e:.2e}")
        assert e < 1e-3, k
    import json
    meta = dict(case="bit_hf_crosscheck", seed=83, seed_scale=0.08, config=dict(vars(c)), spec=[[k, list(s)] for k, s in spec],
                transformers=__import__("transformers").__version__)
    arrays = dict(in_images=images, in_wts=wts, out_features=want_feat.numpy(), out_pooled=want_pool.numpy(),
                  extra_seed_scale=np.array(0.08, dtype=np.float32))
    arrays.update({"grad_" + k: v.numpy() for k, v in want_grads.items()})
    np.savez_compressed(os.path.join(GOLDEN, "bit_hf_crosscheck.npz"), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print("wrote bit_hf_crosscheck.npz")


def bit_wrapper(M):
    """the reference's OWN ResNetTwoTower (src/models/image.py:298-378) around the oracle's BiT restatement -- the `resnetv2_*_bitm` names
    take the same `"resnet" in args.model_name` branch of finetune_image.py:215-216 as resnetv2_50.  Own random stream: the fixtures of
    wrappers() stay bit-identical."""
    rs = np.random.RandomState(4343)
    B, size, nf, seed, scale = 3, 96, NARROW_BIT.num_features, 53, 0.08
    spec = O.resnetv2_state_spec(NARROW_BIT, prefix="e")
    cfg = reference_config(hidden_size=nf, num_labels=2, hidden_dropout_prob=0.1, interaction_type="two_tower", loss_type="ce", loss_margin=0.3)
    enc = OracleEncoder("resnet", NARROW_BIT, spec, nf, None)
    model = M.ResNetTwoTower(cfg, enc).eval()
    full_spec = [("img_encoder." + k[2:], s) for k, s in spec] + [("classifier.out_proj.weight", (2, 2 * nf)), ("classifier.out_proj.bias", (2,))]
    sd = seeded_state_dict(full_spec, seed, scale=scale)
    with torch.no_grad():
        for k, v in sd.items():
            if k.startswith("img_encoder."):
                enc.params[k[len("img_encoder."):].replace(".", "__")].copy_(v)
        model.classifier.out_proj.weight.copy_(sd["classifier.out_proj.weight"])
        model.classifier.out_proj.bias.copy_(sd["classifier.out_proj.bias"])
    im1 = rs.standard_normal((B, 3, size, size)).astype(np.float32)
    im2 = (rs.standard_normal((B, 3, size, size)) * 1.3 + 0.2).astype(np.float32)
    labels = np.array([0, 1, 1], dtype=np.int64)
    out = model(t(im1), t(im2), t(labels))
    out.loss.backward()
    grads = {"classifier.out_proj.weight": model.classifier.out_proj.weight.grad}
    for g in ["stem.conv.weight", "stages.1.blocks.1.conv2.weight", "stages.3.blocks.0.conv3.weight", "norm.weight"]:
        grads["img_encoder." + g] = enc.params[g.replace(".", "__")].grad
    save("resnet_bit_two_tower", cfg, seed, full_spec, dict(images_1=im1, images_2=im2, labels=labels), out, grads,
         extra=dict(seed_scale=np.array(scale, dtype=np.float32)))


def deep(M):
    """24 layers of roberta_large geometry through the reference's RobertaModel (C2 shapes, B = 2, L = 510)"""
    rs = np.random.RandomState(7171)
    cfg = reference_config(os.path.join("/root/reference/src/config/roberta_large.json"), interaction_type="one_tower", max_seq_len=50,
                           max_seq_len_pv=205)
    cfg.pad_token_id = 0
    cfg.vocab_size = 2000
    assert cfg.num_hidden_layers == 24 and cfg.hidden_size == 1024
    model = M.RobertaModel(cfg, add_pooling_layer=False).eval()
    seed = 71
    spec = load_weights(model, seed)
    ids, mask, tt = text_batch(rs, 2, 510, cfg.vocab_size, ragged=False)
    mask[0, 470:] = 0; ids[0, 470:] = 0
    mask[1, 333:] = 0; ids[1, 333:] = 0
    tt = tt * mask
    with torch.no_grad():
        hs = model(t(ids), attention_mask=t(mask), token_type_ids=t(tt), output_hidden_states=True).hidden_states
    extra = {}
    for layer in (0, 1, 6, 12, 18, 24):
        extra[f"h{layer}_sub"] = hs[layer][:, ::15, ::16]
        extra[f"h{layer}_norm"] = hs[layer].norm()
    extra["h24_rows"] = hs[24][:, :3, :]
    save("roberta_large_24_layers", cfg, seed, spec, dict(input_ids=ids, attention_mask=mask, token_type_ids=tt), SimpleNamespace(), None, extra=extra)


if __name__ == "__main__":
    which = sys.argv[1:] or ["wrappers", "vit_hf", "deep"]
    os.makedirs(GOLDEN, exist_ok=True)
    M = load_reference() if ("wrappers" in which or "deep" in which or "bit_wrapper" in which) else None
    if "wrappers" in which:
        wrappers(M)
    if "vit_hf" in which:
        vit_hf()
    if "bit_hf" in which:
        bit_hf()
    if "bit_wrapper" in which:
        bit_wrapper(M)
    if "deep" in which:
        deep(M)
