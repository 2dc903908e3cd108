"""Train-step throughput of the other BASELINE.json configurations on one MI355X (synthetic inputs, dropout on, fused
AdamW; SURVEY.md §8(d) shapes and FLOP counts).  bench.py stays the headline C5 measurement; this fills DESIGN.md §7.

usage: python tools/config_bench.py [--pmc] [c2] [c3] [c3r] [c3b] [c3b3] [c4] [c5x]
  --pmc  also re-run each configuration (1 warm-up + 2 steps) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes)
         and print the HBM bytes per step, the achieved HBM GB/s at the measured step time and its fraction of the 8 TB/s peak
  c2  roberta_large one_tower cls/ce, L = 510          c3  eca_nfnet_l0 two_tower, 800x800
  c4  pkgm_large one_tower, max_pvs 30 (L = 220)        c5x CoCa roberta_large + vit_large_patch16_384, --ensemble cross_attn
"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import item_alignment_amd.models as M
from bench import roberta_large_config
from item_alignment_amd.data.synthetic import SyntheticCocaPairs, one_tower_text

dev = torch.device("cuda:0")


HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PMC = False
CHILD_STEPS = os.environ.get("IA_CB_CHILD_STEPS")     # set in the profiled child: "warm,steps"


def hbm_bytes_per_step(which):
    """Sum of FETCH_SIZE x 2 + WRITE_SIZE (KB; the x2 is the gfx950 correction for wide reads, MI355X_MICROARCH.md) over every
    kernel of a 1 + 2 step child run of configuration `which`, divided by its 3 steps."""
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="ia_cb_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        env = dict(os.environ, IA_CB_CHILD_STEPS="1,2")
        try:
            subprocess.run([rocprof, "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--", sys.executable,
                            os.path.abspath(__file__), which], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900,
                           cwd=out, env=env)      # (exit code not checked: rocprofv3 has returned 1 with a complete counter file)
            f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
            kb[counter] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter) / 3.0
        except Exception as e:
            print(f"  ({counter} pass failed: {e!r})")
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0


def run(name, model, batch_fn, pairs, flops_per_pair, steps=8, warm=3, which=None):
    steps = int(os.environ.get("IA_CB_STEPS", steps))          # soak runs: IA_CB_STEPS=200 ... c3 (the final loss printed must be finite)
    if CHILD_STEPS:
        warm, steps = (int(v) for v in CHILD_STEPS.split(","))
    torch.cuda.reset_peak_memory_stats()          # (several configurations may run in one process: the peak printed below is this one's)
    model = model.cuda().train()
    arena = model.param_arena

    def step():
        arena.zero_grad()
        out = batch_fn(model)
        out.loss.backward()
        arena.adamw_step(1e-5)
        return out.loss
    for _ in range(warm):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    nparam = sum(p.numel() for p in model.parameters())
    print(f"{name}: {pairs} pairs/step, {dt*1e3:.1f} ms/step, {pairs/dt:.1f} pairs/s, {flops_per_pair*pairs/dt/1e12:.0f} TFLOP/s "
          f"({flops_per_pair*pairs/dt/2.5e15*100:.1f} % of bf16 MFMA peak), {nparam/1e6:.0f} M params, loss {loss.item():.4f}, "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del model
    torch.cuda.empty_cache()
    if PMC and which and not CHILD_STEPS:
        b = hbm_bytes_per_step(which)
        if b is not None:
            print(f"  HBM traffic {b/1e9:.2f} GB/step (rocprofv3 --pmc, all kernels) -> {b/dt/1e9:.0f} GB/s achieved = "
                  f"{b/dt/HBM_PEAK*100:.1f} % of the 8 TB/s HBM peak; {flops_per_pair*pairs/b:.0f} FLOP per HBM byte", flush=True)


def c2(pairs=128):
    cfg = roberta_large_config(interaction_type="one_tower")
    rs = np.random.RandomState(2345)
    t = {k: torch.from_numpy(v).to(dev) for k, v in one_tower_text(rs, pairs).items()}
    labels = torch.from_numpy(rs.randint(0, 2, size=pairs)).to(dev)
    torch.manual_seed(2345)
    run("C2 roberta_large one_tower L=510", M.RobertaOneTower(cfg),
        lambda m: m(input_ids=t["input_ids"], attention_mask=t["attention_mask"], token_type_ids=t["token_type_ids"], labels=labels),
        pairs, 1.001e12, which="c2")


def c3(pairs=64, S=800):      # 64 pairs/step since round 6 (profiles/r05_c3_batch_sweep.txt: 16 -> 431, 32 -> 470, 64 -> 484 pairs/s; 98.6 GiB)
    from types import SimpleNamespace
    cfg = SimpleNamespace(num_labels=2, hidden_dropout_prob=0.1, loss_type="ce", loss_margin=0.0, classification_method="cls", hidden_size=2304)
    g = torch.Generator().manual_seed(0)
    im1, im2 = torch.randn((pairs, 3, S, S), generator=g).to(dev), torch.randn((pairs, 3, S, S), generator=g).to(dev)
    labels = torch.randint(0, 2, (pairs,), generator=g).to(dev)
    torch.manual_seed(2345)
    run(f"C3 eca_nfnet_l0 two_tower {S}x{S}", M.NFNetTwoTower(cfg, M.create_model("eca_nfnet_l0")), lambda m: m(im1, im2, labels), pairs, 6.49e11, which="c3")


def c3r(pairs=32, S=800):
    """resnetv2_50 two_tower (reference README.md:187-197): 4.1 GMAC @224 (timm model table) -> 52.3 GMAC / image @800"""
    from types import SimpleNamespace
    cfg = SimpleNamespace(num_labels=2, hidden_dropout_prob=0.1, loss_type="ce", loss_margin=0.0, classification_method="cls", hidden_size=2048)
    g = torch.Generator().manual_seed(0)
    im1, im2 = torch.randn((pairs, 3, S, S), generator=g).to(dev), torch.randn((pairs, 3, S, S), generator=g).to(dev)
    labels = torch.randint(0, 2, (pairs,), generator=g).to(dev)
    torch.manual_seed(2345)
    run(f"C3r resnetv2_50 two_tower {S}x{S}", M.ResNetTwoTower(cfg, M.create_model("resnetv2_50")), lambda m: m(im1, im2, labels), pairs, 6.28e11, which="c3r")


def c3b(pairs=32, S=800):
    """resnetv2_50x1_bitm two_tower (a BiT name; finetune_image.py:23 lists resnetv2_50x3_bitm_in21k): the resnetv2_50 graph with StdConv2d and
    GroupNormAct -- 4.1 GMAC @224 -> 52.3 GMAC / image @800, like c3r"""
    from types import SimpleNamespace
    cfg = SimpleNamespace(num_labels=2, hidden_dropout_prob=0.1, loss_type="ce", loss_margin=0.0, classification_method="cls", hidden_size=2048)
    g = torch.Generator().manual_seed(0)
    im1, im2 = torch.randn((pairs, 3, S, S), generator=g).to(dev), torch.randn((pairs, 3, S, S), generator=g).to(dev)
    labels = torch.randint(0, 2, (pairs,), generator=g).to(dev)
    torch.manual_seed(2345)
    run(f"C3b resnetv2_50x1_bitm two_tower {S}x{S}", M.ResNetTwoTower(cfg, M.create_model("resnetv2_50x1_bitm")), lambda m: m(im1, im2, labels), pairs,
        6.28e11, which="c3b")


def c3b3(pairs=16, S=800):
    """resnetv2_50x3_bitm_in21k two_tower -- the BiT name finetune_image.py:23 gives: every width x 3 (6144 features; 37.1 GMAC @224 in timm's
    model table -> 473 GMAC / image @800).  Its 192 / 384 / 768 / 1536-channel 3x3 convolutions are not powers of two: patch-matrix path."""
    from types import SimpleNamespace
    cfg = SimpleNamespace(num_labels=2, hidden_dropout_prob=0.1, loss_type="ce", loss_margin=0.0, classification_method="cls", hidden_size=6144)
    g = torch.Generator().manual_seed(0)
    im1, im2 = torch.randn((pairs, 3, S, S), generator=g).to(dev), torch.randn((pairs, 3, S, S), generator=g).to(dev)
    labels = torch.randint(0, 2, (pairs,), generator=g).to(dev)
    torch.manual_seed(2345)
    run(f"C3b3 resnetv2_50x3_bitm_in21k two_tower {S}x{S}", M.ResNetTwoTower(cfg, M.create_model("resnetv2_50x3_bitm_in21k")),
        lambda m: m(im1, im2, labels), pairs, 5.68e12, which="c3b3")


def c4(pairs=256):
    S, P = 50, 30
    cfg = roberta_large_config(interaction_type="one_tower", max_seq_len=S, max_seq_len_pv=None, max_pvs=P, num_entities=258211,
                               num_relations=1379, kg_embedding_dim=1024, entity_projection_bias=False)
    rs = np.random.RandomState(2345)
    L_ids, L_emb = 2 * (S + P + 1), 2 * (S + 2 * P)
    ids = np.zeros((pairs, L_ids), dtype=np.int64)
    mask = np.zeros((pairs, L_emb), dtype=np.int64)
    tt = np.zeros((pairs, L_emb), dtype=np.int64)
    for i in range(pairs):
        for side in range(2):
            n = int(rs.randint(8, S - 1))
            o_ids, o_emb = side * (S + P + 1), side * (S + 2 * P)
            ids[i, o_ids] = 101 if side == 0 else 102
            ids[i, o_ids + 1:o_ids + 1 + n] = rs.randint(1000, 21128, size=n)
            ids[i, o_ids + 1 + n] = 102
            mask[i, o_emb:o_emb + n + 2] = 1
            nrel = int(rs.randint(5, P + 1))
            ids[i, o_ids + S] = rs.randint(1, 258211)
            ids[i, o_ids + S + 1:o_ids + S + 1 + nrel] = rs.randint(1, 1379, size=nrel)
            mask[i, o_emb + S:o_emb + S + 2 * nrel] = 1
            tt[i, o_emb:o_emb + S + 2 * P] = side
    pos = np.tile(np.arange(L_emb), (pairs, 1))
    t = [torch.from_numpy(a).to(dev) for a in (ids, mask, tt, pos)]
    labels = torch.from_numpy(rs.randint(0, 2, size=pairs)).to(dev)
    torch.manual_seed(2345)
    run("C4 pkgm_large one_tower L=220", M.PKGMOneTower(cfg),
        lambda m: m(input_ids=t[0], attention_mask=t[1], token_type_ids=t[2], position_ids=t[3], labels=labels), pairs, 4.129e11, which="c4")


def c5x(pairs=64):      # 64 pairs/step since round 6 (16 until then: profiles/r06_c5x_batch_sweep.txt -- 16 -> 112, 64 -> 184 pairs/s, 131 GiB)
    # coca_large.json: 24 multimodal layers, 16 heads, ff_mult 12 (reference src/config/coca_large.json) + ViT-L/16 @384 (1024-d tokens)
    cfg = roberta_large_config(ensemble="cross_attn", num_hidden_layers_multimodal=24, num_attention_heads_multimodal=16,
                               feedforward_multiplication_multimodal=12)
    torch.manual_seed(2345)
    model = M.CoCaForItemAlignment(cfg, M.create_model("vit_large_patch16_384"), M.RobertaModel(cfg))
    data = SyntheticCocaPairs(pairs)
    batch = data.batch(list(range(pairs)), dev)
    # per pair (only item 1 reaches the output, quirk A4): text 1.604e11 + ViT-L 3.9e11 (fwd) + 24 multimodal layers
    L, N, H, F = 255, 577, 1024, 12288
    mm = 24 * (2 * L * H * (H + 128 + 2 * F) + 2 * L * H * H + 2 * L * F * H + 4 * L * L * 16 * 64       # parallel block
               + 2 * L * H * H + 2 * N * H * 128 + 4 * L * N * 16 * 64 + 2 * L * H * H + 2 * L * H * 2 * F + 2 * L * F * H)   # cross attention
    fwd = 1.604e11 + 3.9e11 + mm
    run("C5x CoCa roberta_large + ViT-L/16 cross_attn (coca_large.json)", model, lambda m: m(*batch[:10], labels=batch[10]), pairs, 3 * fwd, which="c5x")


if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--pmc" in argv:
        PMC = True
        argv.remove("--pmc")
    which = argv or ["c2", "c3", "c4", "c5x"]
    pairs = os.environ.get("IA_CB_PAIRS")              # pairs per step override (batch-size sweeps: IA_CB_PAIRS=32 ... c3)
    for w in which:
        globals()[w](int(pairs)) if pairs else globals()[w]()
