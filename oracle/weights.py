"""TEST INFRASTRUCTURE — deterministic weights for parity work (no checkpoints exist offline).

`seeded_state_dict(spec, seed)` fills every floating tensor of a state_dict spec [(key, shape), ...] from
np.random.RandomState(seed) in spec order (bit-stable across numpy versions), so the reference classes
(build container), the CPU oracle and the HIP engine (GPU box) all see identical parameters without
shipping tensors.  Non-trivial LayerNorm gains / biases are used on purpose: all-ones / all-zeros would
hide errors in those code paths.
"""
import numpy as np
import torch


def _kind(key):
    k = key.lower()
    if k.endswith("position_ids") or k.endswith("token_type_ids") or "inv_freq" in k:
        return "buffer"
    if k.endswith("layernorm.weight") or k.endswith(".gamma") or k.endswith(".gain") or ".norm" in k and k.endswith(".weight") or k.endswith("norm.weight"):
        return "gain"
    if k.endswith(".beta") and "norm" in k:
        return "zero_buffer"      # CoCa LayerNorm.beta is a zero buffer (reference multimodal.py:479)
    if k.endswith(".bias") or k.endswith(".beta"):
        return "bias"
    if "cls_token" in k or "pos_embed" in k:
        return "small"
    return "weight"


def seeded_state_dict(spec, seed, scale=0.05):
    rs = np.random.RandomState(seed)
    sd = {}
    for key, shape in spec:
        kind = _kind(key)
        shape = tuple(int(s) for s in shape)
        if kind == "buffer":
            continue
        if kind == "zero_buffer":
            sd[key] = torch.zeros(shape)
            continue
        x = rs.standard_normal(size=shape).astype(np.float32)
        if kind == "gain":
            x = 1.0 + 0.1 * x
        elif kind == "bias":
            x = 0.02 * x
        elif kind == "small":
            x = 0.02 * x
        else:
            x = scale * x
        sd[key] = torch.from_numpy(x)
    return sd


def spec_of(state_dict):
    return [(k, tuple(v.shape)) for k, v in state_dict.items() if torch.is_floating_point(v)]
