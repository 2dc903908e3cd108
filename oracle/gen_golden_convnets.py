"""TEST INFRASTRUCTURE — tests/golden/nfnet_reference_assembly.npz: the REFERENCE's own in-tree NormFreeNet assembly
(/root/reference/src/models/image.py:40-199 — stem, stage / stride / dilation, beta = 1/sqrt(expected_var), final conv) instantiated
for eca_nfnet_l0 in the build container, with the timm 0.6.5 building blocks it imports (absent offline) supplied by
oracle/timm_blocks.py — an nn.Module restatement written independently of oracle/ref_models.py's functional one.  The fixture pins
(a) ref_models.nfnet_plan / nfnet_forward_features against the reference's assembly code and (b) ref_models' block arithmetic against
a second implementation on other torch primitives.  What it can NOT pin is timm itself (not installed): DESIGN.md 6 says so.

    python oracle/gen_golden_convnets.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import timm_blocks as TB  # noqa: E402
from oracle.weights import seeded_state_dict  # noqa: E402


class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return type(n, (), {})


def install():
    import transformers  # noqa: F401
    for m in ["timm", "timm.models", "timm.models.nfnet", "timm.models.vision_transformer", "timm.models.layers", "timm.models.layers.classifier",
              "timm.data", "timm.data.transforms_factory", "torch_geometric", "torch_geometric.nn", "jieba"]:
        if m not in sys.modules:
            s = _Stub(m); s.__path__ = []; sys.modules[m] = s
    nf = sys.modules["timm.models.nfnet"]
    for n in ("NfCfg", "_nonlin_gamma", "act_with_gamma", "create_stem", "NormFreeBlock"):
        setattr(nf, n, getattr(TB, n))
    ly = sys.modules["timm.models.layers"]
    for n in ("ScaledStdConv2d", "ScaledStdConv2dSame", "get_act_layer", "get_attn", "make_divisible"):
        setattr(ly, n, getattr(TB, n))
    cl = sys.modules["timm.models.layers.classifier"]
    cl._create_pool, cl._create_fc = TB._create_pool, TB._create_fc


def main():
    install()
    from src.models.image import NormFreeNet
    out = {}
    meta = {"cases": {}}
    for name, size in (("eca_nfnet_l0", 96), ("eca_nfnet_l1", 64)):
        torch.manual_seed(0)
        net = NormFreeNet(TB.eca_nfnet_cfg(name), num_classes=2).eval()
        spec = [("img_encoder." + k, tuple(v.shape)) for k, v in net.state_dict().items() if not k.startswith("classifier.")]
        sd = seeded_state_dict(spec, 31)
        net.load_state_dict({k[len("img_encoder."):]: v for k, v in sd.items()}, strict=False)
        x = torch.from_numpy(np.random.RandomState(5).standard_normal((2, 3, size, size)).astype(np.float32))
        with torch.no_grad():
            stem = net.stem(x)
            s0 = net.stages[0](stem)
            feats = net.forward_features(x)
        out[f"{name}_in"], out[f"{name}_stem"], out[f"{name}_stage0_mean"], out[f"{name}_features"] = (
            x.numpy(), stem.numpy(), s0.mean((2, 3)).numpy(), feats.numpy())
        betas = [[float(b.beta) for b in stage] for stage in net.stages]
        meta["cases"][name] = {"spec": [[k, list(s)] for k, s in spec], "seed": 31, "betas": betas,
                               "strides": [[int(b.conv2.stride[0]) for b in stage] for stage in net.stages],
                               "groups": [[int(b.conv2.groups) for b in stage] for stage in net.stages]}
        print(name, "features", tuple(feats.shape), "abs max", feats.abs().max().item())
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "nfnet_reference_assembly.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
