#!/usr/bin/env python3
"""What holds the LAST text layer's query-projection gradient of the full-width C5 pair at rel 0.27 (tests/test_baseline_shapes_gpu.py)?
Four gradients of the same tensors on the same inputs: the HIP engine, the fp32 oracle, the oracle under bf16 storage rounding, and the
oracle under bf16 storage rounding with the flash-style delta = rowsum(dO o O_rounded).  Prints |a - b|_max / |fp32|_max for every pair.
Usage (GPU box): python tools/c5_delta_probe.py > gpurun_out/c5_delta_probe.txt"""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import item_alignment_amd.models as M
    from bench import roberta_large_config
    from item_alignment_amd.data.synthetic import SyntheticCocaPairs
    from item_alignment_amd.models.image import VIT_CONFIGS
    from oracle import ref_models as O
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    cfg = roberta_large_config(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(2345)
    model = M.CoCaForItemAlignment(cfg, M.create_model("vit_base_patch16_384"), M.RobertaModel(cfg))
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    data = SyntheticCocaPairs(2, image_size=384, seed=9)
    b = data.batch([0, 1], "cuda")
    model.param_arena.zero_grad()
    model(*b[:10], labels=b[10]).loss.backward()
    torch.cuda.synchronize()
    s_, p_, d_, depth, h_ = VIT_CONFIGS["vit_base_patch16_384"]
    vcfg = SimpleNamespace(embed_dim=d_, depth=depth, num_heads=h_, patch_size=p_, eps=1e-6)
    bc = data.batch([0, 1], "cpu")
    keys = [k for k in sd if ("layer.23." in k or "layer.0." in k or "layer.12." in k) and k.endswith(("self.query.weight", "self.key.weight", "self.value.weight"))]
    params = dict(model.named_parameters())
    got = {"hip": {k: params[k].grad.float().cpu().clone() for k in keys}}

    def oracle(mode):
        rsd = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
        if mode == "fp32":
            O.coca_item_alignment(rsd, cfg, vcfg, *bc[:10], labels=bc[10], training=False).loss.backward()
        else:
            with O.rounding(torch.bfloat16, flash_delta=(mode == "bf16+flash_delta")):
                O.coca_item_alignment(rsd, cfg, vcfg, *bc[:10], labels=bc[10], training=False).loss.backward()
        return {k: rsd[k].grad.clone() for k in keys}
    for mode in ("fp32", "bf16", "bf16+flash_delta"):
        got[mode] = oracle(mode)
    names = list(got)
    for k in keys:
        scale = got["fp32"][k].abs().max().item() + 1e-30
        print(k)
        for i, a in enumerate(names):
            for bname in names[i + 1:]:
                d = (got[a][k] - got[bname][k]).abs().max().item() / scale
                x, y = got[a][k].flatten(), got[bname][k].flatten()
                c = (torch.dot(x, y) / (x.norm() * y.norm() + 1e-30)).item()
                print(f"   {a:>18s} vs {bname:<18s} rel {d:.4f} cosine {c:.5f}")


if __name__ == "__main__":
    main()
