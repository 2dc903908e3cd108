"""Helpers shared by the parity tests: load a golden fixture (captured from the reference by
oracle/gen_golden.py), regenerate its seeded weights, and run the matching oracle function."""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    cfg = SimpleNamespace(**meta["config"])
    inputs = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    outs = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out_")}
    grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad_")}
    extra = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("extra_")}
    spec = [(k, tuple(s)) for k, s in meta["spec"]]
    return SimpleNamespace(name=name, cfg=cfg, seed=meta["seed"], spec=spec, inputs=inputs, outs=outs, grads=grads, extra=extra)


def case_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


# narrow instances of the timm towers used by the image two-tower fixtures (oracle/gen_golden_r2.py)
NARROW_NFNET = SimpleNamespace(depths=(1, 2, 1, 1), channels=(256, 512, 512, 512), stem_chs=128, group_size=64, bottle_ratio=0.25,
                               num_features=512, alpha=0.2, attn_gain=2.0, eps=1e-5, ch_div=8)
NARROW_RESNET = SimpleNamespace(layers=(1, 2, 1, 1), channels=(64, 128, 256, 256), stem_chs=32, bottle_ratio=0.25, eps=1e-5, momentum=0.1,
                                num_features=256)
NARROW_BIT = SimpleNamespace(layers=(1, 2, 1, 1), channels=(128, 256, 256, 512), stem_chs=32, bottle_ratio=0.25, eps=1e-5, momentum=0.1,
                             num_features=512, bit=True, std_eps=1e-8, groups=32)
TINY_VIT = SimpleNamespace(embed_dim=128, depth=2, num_heads=2, patch_size=16, eps=1e-6, image_size=64)


def weights(case, requires_grad=False):
    from oracle.weights import seeded_state_dict
    scale = float(case.extra["seed_scale"]) if "seed_scale" in case.extra else -1.0
    sd = seeded_state_dict(case.spec, case.seed, scale=scale) if scale > 0 else seeded_state_dict(case.spec, case.seed)
    if requires_grad:
        for v in sd.values():
            v.requires_grad_(True)
    return sd


def vit_cfg(case):
    v = case.extra["vit"].tolist()
    return SimpleNamespace(embed_dim=v[0], depth=v[1], num_heads=v[2], patch_size=v[3], image_size=v[4], eps=1e-6)


def pair_list(case):
    """the fixture stores the ragged `pair_indices` list padded with -1 rows: back to a list of [n_i, 5] tensors (n_i may be 0)"""
    p = case.inputs.get("pair_indices")
    if p is None:
        return None
    return [rows[(rows[:, 0] >= 0)] for rows in p]


def run_oracle(case, sd, training=False):
    """Dispatch a fixture to the oracle function that restates the reference class it was captured from."""
    from oracle import ref_models as O
    i, cfg, n = case.inputs, case.cfg, case.name
    g = i.get
    if n.startswith("roberta_one_tower"):
        lab = i["labels"].float() if cfg.loss_type == "bce" else i["labels"]
        return O.roberta_one_tower(sd, cfg, i["input_ids"], i["attention_mask"], i["token_type_ids"], None, lab, training,
                                   pair_indices=pair_list(case))
    if n.startswith("roberta_two_tower"):
        return O.roberta_two_tower(sd, cfg, i["input_ids_1"], i["attention_mask_1"], i["token_type_ids_1"], None, i["input_ids_2"],
                                   i["attention_mask_2"], i["token_type_ids_2"], None, i["labels"], training)
    if n.startswith("pkgm_one_tower"):
        return O.pkgm_one_tower(sd, cfg, i["input_ids"], i["attention_mask"], i["token_type_ids"], i["position_ids"], i["labels"], training)
    if n.startswith("pkgm_two_tower"):
        return O.pkgm_two_tower(sd, cfg, i["input_ids_1"], i["attention_mask"], i["token_type_ids"], i["position_ids"], i["input_ids_2"],
                                i["attention_mask"], i["token_type_ids"], i["position_ids"], i["labels"], training)
    if n.startswith("roberta_image_one_tower"):
        return O.roberta_image_one_tower(sd, cfg, i["input_ids"], i["attention_mask"], i["token_type_ids"], None, [i["img1"], i["img2"]],
                                         i["image_indices"], i["labels"], training)
    if n.startswith("roberta_image_two_tower"):
        return O.roberta_image_two_tower(sd, cfg, i["input_ids_1"], i["attention_mask_1"], i["token_type_ids_1"], None, i["img1"],
                                         i["input_ids_2"], i["attention_mask_2"], i["token_type_ids_2"], None, i["img2"], i["labels"], training)
    if n.startswith("textcnn"):
        return O.textcnn_two_tower(sd, cfg, i["input_ids_1"], i["input_ids_2"], i["labels"], training)
    if n.startswith("nfnet_two_tower"):
        return O.nfnet_two_tower(sd, cfg, NARROW_NFNET, i["images_1"], i["images_2"], i["labels"], training)
    if n.startswith("resnet_bit_two_tower"):
        return O.resnetv2_two_tower(sd, cfg, NARROW_BIT, i["images_1"], i["images_2"], i["labels"], training, None)
    if n.startswith("resnet_two_tower"):
        stats = O.resnetv2_running_stats(NARROW_RESNET, "img_encoder")
        return O.resnetv2_two_tower(sd, cfg, NARROW_RESNET, i["images_1"], i["images_2"], i["labels"], training, stats)
    if n.startswith("vit_two_tower"):
        f1 = O.vit_forward_head(O.vit_forward_features(sd, "img_encoder", TINY_VIT, i["images_1"]))
        f2 = O.vit_forward_head(O.vit_forward_features(sd, "img_encoder", TINY_VIT, i["images_2"]))
        return O.image_two_tower(sd, cfg, f1, f2, i["labels"], training)
    if n.startswith("coca"):
        return O.coca_item_alignment(sd, cfg, vit_cfg(case), i["input_ids_1"], i["attention_mask_1"], i["token_type_ids_1"], None, i["img1"],
                                     i["input_ids_2"], i["attention_mask_2"], i["token_type_ids_2"], None, i["img2"], i["labels"], training)
    raise KeyError(n)
