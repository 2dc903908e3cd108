"""Parity at the BASELINE.json shapes themselves (round-2 verdict, weak #7): the golden fixtures cover every model at narrow widths;
these run the real widths against the CPU oracle (oracle/ref_models.py, fp32) on the GPU box -- the 258 211 x 1024 PKGM entity
table (a > 2^31-byte gather), eca_nfnet_l0 at 800 x 800, and one full-width CoCa pair (24-layer roberta_large at L = 255 + ViT-B/16
at 384).  Weights are the models' own seeded initialisation copied into the oracle's state dict; eval mode (dropout off)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu
TOL = 5e-2


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()


def state_of(model, head_gain=1.0):
    """the model's own (seeded) initialisation as the oracle's state dict; the pair head's weights are scaled up first so that the
    logits are O(1) instead of the O(0.02) a fresh head produces (a relative tolerance on them then means something)"""
    with torch.no_grad():
        for k, v in model.named_parameters():
            if k.startswith("classifier.") and k.endswith("weight"):
                v.mul_(head_gain)
    return {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}


def test_c4_pkgm_large_full_entity_table(gpu):
    """C4: PKGMOneTower, num_entities 258 211, kg_embedding_dim 1024, max_pvs 30 -> 220 embedded positions, H = 1024, all 24 layers, B = 2
    (reference src/models/base.py:347-392, text.py:720-783).  Loss, logits and the entity-table gradient rows against the oracle."""
    import item_alignment_amd.models as M
    from bench import roberta_large_config
    from oracle import ref_models as O
    S, P, B = 50, 30, 2
    cfg = roberta_large_config(interaction_type="one_tower", max_seq_len=S, max_seq_len_pv=None, max_pvs=P, num_entities=258211,
                               num_relations=1379, kg_embedding_dim=1024, entity_projection_bias=False, num_hidden_layers=24,
                               hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    rs = np.random.RandomState(7)
    L_ids, L_emb = 2 * (S + P + 1), 2 * (S + 2 * P)
    ids = np.zeros((B, L_ids), dtype=np.int64); mask = np.zeros((B, L_emb), dtype=np.int64); tt = np.zeros((B, L_emb), dtype=np.int64)
    ents = [[258210, 131072 + 7], [1, 200000]]            # the last row, a row past 2^31 bytes / 4, the first real row
    for i in range(B):
        for side in range(2):
            n = int(rs.randint(8, S - 1))
            o_ids, o_emb = side * (S + P + 1), side * (S + 2 * P)
            ids[i, o_ids] = 101 if side == 0 else 102
            ids[i, o_ids + 1:o_ids + 1 + n] = rs.randint(1000, 21128, size=n)
            ids[i, o_ids + 1 + n] = 102
            mask[i, o_emb:o_emb + n + 2] = 1
            nrel = int(rs.randint(5, P + 1))
            ids[i, o_ids + S] = ents[i][side]
            ids[i, o_ids + S + 1:o_ids + S + 1 + nrel] = rs.randint(1, 1379, size=nrel)
            mask[i, o_emb + S:o_emb + S + 2 * nrel] = 1
            tt[i, o_emb:o_emb + S + 2 * P] = side
    pos = np.tile(np.arange(L_emb), (B, 1))
    labels = torch.tensor([1, 0])
    torch.manual_seed(11)
    model = M.PKGMOneTower(cfg)
    sd = state_of(model, head_gain=20.0)
    model = model.cuda().eval()
    t = [torch.from_numpy(a) for a in (ids, mask, tt, pos)]
    out = model(input_ids=t[0].cuda(), attention_mask=t[1].cuda(), token_type_ids=t[2].cuda(), position_ids=t[3].cuda(), labels=labels.cuda())
    model.param_arena.zero_grad()
    out.loss.backward()
    torch.cuda.synchronize()
    ent_key = next(k for k in sd if k.endswith("ent_emb.weight"))
    rsd = {k: (v.requires_grad_(True) if k == ent_key or k.endswith("rel_emb.weight") or k.endswith("proj_mat.weight") else v) for k, v in sd.items()}
    ref = O.pkgm_one_tower(rsd, cfg, *t, labels=labels, training=False)
    ref.loss.backward()
    assert abs(out.loss.item() - ref.loss.item()) < TOL * max(1.0, abs(ref.loss.item())), (out.loss.item(), ref.loss.item())
    assert rel(out.logits.detach(), ref.logits.detach()) < TOL
    got = dict(model.named_parameters())[ent_key].grad.float().cpu()
    want = rsd[ent_key].grad
    rows = sorted({e for pair in ents for e in pair})
    rel_key = next(k for k in sd if k.endswith("rel_emb.weight"))
    # quirk A1 (F.normalize over a size-1 dim = sign(x)) has zero gradient: the HIP path returns exact zeros for the entity rows.  The
    # oracle differentiates x / max(|x|, eps) in fp32, whose two cancelling terms leave rounding noise of order eps_fp32 / |x| per
    # element (|x| goes down to 2e-6 in a fresh table: 2^-9 here) -- bounded against the scale of the relation-table gradient.
    untouched = torch.ones(got.shape[0], dtype=torch.bool); untouched[rows] = False
    assert got[untouched].abs().max().item() == 0.0 and want[untouched].abs().max().item() == 0.0
    if got[rows].abs().max().item() < 1e-6:
        assert want[rows].abs().max().item() < 0.02 * rsd[rel_key].grad.abs().max().item(), want[rows].abs().max().item()
    else:
        assert (got[rows] - want[rows]).abs().max().item() <= TOL * want[rows].abs().max().item() + 1e-6
    # 24 bf16 layers between these tables and the loss, a batch of 2.  The bar is set by a measurement that does not involve the engine
    # (advisor, round 4): the same oracle under bf16 storage rounding (oracle.ref_models.rounding) is one more draw of the rounding noise
    # this depth carries; the HIP gradient has to stay within 1.6 x of that draw's distance from fp32 (floor: the common 0.10 bar), and
    # inside the absolute caps that were in force before (0.19 Frobenius / 0.22 max-norm).
    table_keys = (rel_key, next(k for k in sd if k.endswith("proj_mat.weight")))
    bsd = {k: (v.detach().clone().requires_grad_(True) if k in table_keys else v.detach()) for k, v in sd.items()}
    with O.rounding(torch.bfloat16):
        O.pkgm_one_tower(bsd, cfg, *t, labels=labels, training=False).loss.backward()
    for k in table_keys:
        g_, w_, b_ = dict(model.named_parameters())[k].grad, rsd[k].grad, bsd[k].grad
        fro = ((g_.float().cpu() - w_.float()).norm() / w_.float().norm()).item()
        noise_fro, noise_rel = ((b_.float() - w_.float()).norm() / w_.float().norm()).item(), rel(b_, w_)
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/c4_table_gradients.txt", "a") as f:
            f.write(f"{k}: cosine {cosine(g_, w_):.5f} max-norm rel {rel(g_, w_):.4f} frobenius rel {fro:.4f} | bf16-rounding oracle vs fp32: "
                    f"max-norm rel {noise_rel:.4f} frobenius rel {noise_fro:.4f}\n")
        assert cosine(g_, w_) > 0.98, (k, cosine(g_, w_))
        assert fro < min(0.19, max(0.10, 1.6 * noise_fro)), (k, fro, noise_fro)
        assert rel(g_, w_) < min(0.22, max(0.10, 1.6 * noise_rel)), (k, rel(g_, w_), noise_rel)


def test_c3_eca_nfnet_l0_at_800(gpu):
    """C3: NFNetTwoTower(eca_nfnet_l0) at 800 x 800, one pair (reference src/models/image.py:253-294; the tower is timm's, restated in
    the oracle from its published definition -- parity unpinned by the reference).  Logits and the stem / last-stage
    weight gradients against the fp32 oracle."""
    import item_alignment_amd.models as M
    from oracle import ref_models as O
    cfg = SimpleNamespace(num_labels=2, hidden_dropout_prob=0.0, loss_type="ce", loss_margin=0.0, classification_method="cls", hidden_size=2304)
    g = torch.Generator().manual_seed(3)
    im1, im2 = torch.randn((1, 3, 800, 800), generator=g), torch.randn((1, 3, 800, 800), generator=g)
    labels = torch.tensor([1])
    torch.manual_seed(5)
    model = M.NFNetTwoTower(cfg, M.create_model("eca_nfnet_l0"))
    with torch.no_grad():                                  # timm starts conv3.gain at 0 (the residual branches are off in a fresh
        for k, v in model.named_parameters():              # net): switch them on so that all 12 blocks take part, as after training
            if k.endswith("conv3.gain"):
                v.fill_(1.0)
    sd = state_of(model, head_gain=30.0)
    model = model.cuda().eval()
    out = model(im1.cuda(), im2.cuda(), labels.cuda())
    model.param_arena.zero_grad()
    out.loss.backward()
    torch.cuda.synchronize()
    keys = ["img_encoder.stem.conv1.weight", "img_encoder.stages.3.2.conv3.weight", "img_encoder.final_conv.weight"]
    rsd = {k: (v.requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    ncfg = O.nfnet_cfg("eca_nfnet_l0")
    f1 = O.nfnet_global_pool(O.nfnet_forward_features(rsd, "img_encoder", ncfg, im1))
    f2 = O.nfnet_global_pool(O.nfnet_forward_features(rsd, "img_encoder", ncfg, im2))
    ref = O.image_two_tower(rsd, cfg, f1, f2, labels, False)          # = nfnet_two_tower, with the pooled features kept
    ref.loss.backward()
    # the tower: pooled 2304-d features of both images (measured at 800 x 800: every stage within 2 % of the oracle's maximum)
    with torch.no_grad():
        feats = model._embed(torch.cat((im1, im2)).cuda())
    want_f = torch.cat((f1, f2)).detach()
    assert rel(feats, want_f) < 0.03, rel(feats, want_f)
    assert cosine(feats, want_f) > 0.9995, cosine(feats, want_f)
    # the head: a fresh pair head's logits are a small difference of large sums over 3 x 2304 feature terms (0.05 and 0.22 here with
    # the x30 weights), so the features' bf16 error reaches them amplified by the head's gain: absolute tolerance, like the loss
    assert (out.logits.detach().float().cpu() - ref.logits.detach()).abs().max().item() < TOL, (out.logits, ref.logits)
    assert abs(out.loss.item() - ref.loss.item()) < TOL * max(1.0, abs(ref.loss.item()))
    params = dict(model.named_parameters())
    for k in keys:
        c = cosine(params[k].grad, rsd[k].grad)
        assert c > 0.97, (k, c)
        assert rel(params[k].grad, rsd[k].grad) < 0.25, (k, rel(params[k].grad, rsd[k].grad))


def test_c5_full_width_coca_pair(gpu):
    """C5: one full-width pair through CoCaForItemAlignment -- roberta_large (24 layers, H = 1024, L = 255) + ViT-B/16 at 384, ensemble
    sum (reference src/models/multimodal.py:983-1045).  Loss, probabilities AND the backward pass against the fp32 oracle: gradients of
    the pair head, the first and the last text layer's query projection, the ViT's first qkv projection and the patch embedding (the
    deepest points of both towers' backward chains -- a fault in any attention / GEMM / LayerNorm backward kernel on the way down
    shows up there)."""
    import item_alignment_amd.models as M
    from bench import roberta_large_config
    from item_alignment_amd.data.synthetic import SyntheticCocaPairs
    from item_alignment_amd.models.image import VIT_CONFIGS
    from oracle import ref_models as O
    cfg = roberta_large_config(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(2345)
    model = M.CoCaForItemAlignment(cfg, M.create_model("vit_base_patch16_384"), M.RobertaModel(cfg))
    sd = state_of(model)                                  # the summed CLS vectors are large: the fresh head already gives O(1) logits
    model = model.cuda().eval()
    data = SyntheticCocaPairs(2, image_size=384, seed=9)
    b = data.batch([0, 1], "cuda")
    model.param_arena.zero_grad()
    out = model(*b[:10], labels=b[10])
    out.loss.backward()
    torch.cuda.synchronize()
    s_, p_, d_, depth, h_ = VIT_CONFIGS["vit_base_patch16_384"]
    vcfg = SimpleNamespace(embed_dim=d_, depth=depth, num_heads=h_, patch_size=p_, eps=1e-6)
    bc = data.batch([0, 1], "cpu")

    def key(*parts):
        hits = [k for k in sd if all(p_ in k for p_ in parts)]
        assert len(hits) == 1, (parts, hits)
        return hits[0]
    keys = [key("classifier", "out_proj.weight"), key("layer.0.", "self.query.weight"), key("layer.23.", "self.query.weight"),
            key("layer.23.", "self.key.weight"), key("layer.23.", "self.value.weight"), key("blocks.0.", "attn.qkv.weight"), key("patch_embed.proj.weight")]
    rsd = {k: (v.requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    ref = O.coca_item_alignment(rsd, cfg, vcfg, *bc[:10], labels=bc[10], training=False)
    ref.loss.backward()
    assert abs(out.loss.item() - ref.loss.item()) < TOL * max(1.0, abs(ref.loss.item())), (out.loss.item(), ref.loss.item())
    assert rel(out.logits.detach(), ref.logits.detach()) < TOL, (out.logits, ref.logits)
    assert (out.probs.detach().float().cpu() - ref.probs.detach()).abs().max().item() < TOL
    params = dict(model.named_parameters())
    report = {}
    for k in keys:
        g_, w_ = params[k].grad, rsd[k].grad
        assert torch.isfinite(g_).all(), k
        report[k] = (cosine(g_, w_), rel(g_, w_))
    print("C5 full-width gradients (cosine, rel):", report)
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "c5_full_width_gradients.txt"), "w") as f:
        for k, (c, r) in report.items():
            f.write(f"{k}: cosine {c:.4f} rel {r:.4f}\n")
    # The LAST layer's query projection is the one tensor off the common bar, by construction and not by a kernel fault (measured
    # cosine 0.9858 / rel 0.268, identical for the round-3 kernel pair and the fused round-4 kernel): with ensemble = sum only the CLS
    # row of the last layer carries a gradient, so dW_q is ONE dQ row per sequence, dq_0 = sum_k P_0k (dP_0k - delta_0) K_k -- under a
    # fresh initialisation (P ~ uniform) a weak covariance between dP and K over the keys.  Flash-style backward takes delta =
    # rowsum(dO o O) from the bf16 context the forward stored, so delta carries 2^-9 |dO . O| of rounding, which enters as
    # eps * mean_k(K) -- small against |dq| of a trained model, comparable to that weak covariance here.  Every layer below averages
    # the same effect over its 255 query rows (layer 0: 0.017).  DESIGN.md 5 records it as a deviation of bf16 context storage.
    # Being noise, the figure moves with any change of a rounding point upstream: 0.9740 / 0.274 since the QKV projection rounds
    # q * scale * log2 e once (IA_Q_PRESCALE=1) instead of q and then q * sc inside the kernels (0.9858 / 0.268).
    # Round 5 -- the exception is MEASURED here, not only argued: the same oracle under bf16 storage rounding with delta taken the
    # flash way (rowsum(dO o O_rounded), oracle.ref_models.rounding(bf16, flash_delta=True)) moves this one tensor by the same order
    # (tools/c5_delta_probe.py, profiles/r05_c5_delta_probe.txt: fp32 vs bf16 0.029, fp32 vs bf16 + flash delta 0.212, HIP vs fp32
    # 0.275 -- two draws of one noise).  The HIP gradient has to stay within 1.6 x of what that restatement of the arithmetic does to
    # the tensor (and inside the absolute bound that has been in force since round 4); every other tensor keeps the common bar.
    # Round 6: the last layer's KEY projection sits in the same place for the same reason (dK of the last layer is the CLS row's dS
    # column, dS = P (dP - delta): the same delta; round 5 measured 0.249 and left it out of the list) -- it is in `keys` now, under
    # the same measured bound taken from ITS OWN flash-delta restatement.
    kq, kk = key("layer.23.", "self.query.weight"), key("layer.23.", "self.key.weight")
    fsd = {k: (v.detach().clone().requires_grad_(True) if k in (kq, kk) else v.detach()) for k, v in sd.items()}
    with O.rounding(torch.bfloat16, flash_delta=True):
        O.coca_item_alignment(fsd, cfg, vcfg, *bc[:10], labels=bc[10], training=False).loss.backward()
    loose = {}
    for kx in (kq, kk):
        flash_rel, flash_cos = rel(fsd[kx].grad, rsd[kx].grad), cosine(fsd[kx].grad, rsd[kx].grad)
        print(kx, "bf16 oracle with flash-style delta against fp32 (cosine, rel):", flash_cos, flash_rel)
        loose[kx] = (min(0.99, max(0.96, 1.0 - 3.0 * (1.0 - flash_cos))), max(0.10, min(0.32, 1.6 * flash_rel)))
    for k, (c, r) in report.items():
        cmin, rmax = loose.get(k, (0.99, 0.10))
        assert c >= cmin, (k, c, r, loose.get(k))
        assert r <= rmax, (k, c, r, loose.get(k))
    # every parameter of both towers received a finite gradient (a poisoned row anywhere in the backward would be non-finite here)
    for k, v in params.items():
        if v.grad is not None:
            assert torch.isfinite(v.grad).all(), k
    # Round 6 -- the opt-in that removes the deviation: IA_ATTN_EXACT_DELTA=1 (attn_bwd3_delta_kernel: delta = sum_k P dP in fp32) brings
    # both tensors back under the common bar; everything else stays there.
    os.environ["IA_ATTN_EXACT_DELTA"] = "1"
    try:
        model.param_arena.zero_grad()
        model(*b[:10], labels=b[10]).loss.backward()
        torch.cuda.synchronize()
    finally:
        os.environ.pop("IA_ATTN_EXACT_DELTA", None)
    exact = {k: (cosine(params[k].grad, rsd[k].grad), rel(params[k].grad, rsd[k].grad)) for k in keys}
    print("C5 full-width gradients with IA_ATTN_EXACT_DELTA=1 (cosine, rel):", exact)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "c5_full_width_gradients.txt"), "a") as f:
        for k, (c, r) in exact.items():
            f.write(f"IA_ATTN_EXACT_DELTA=1 {k}: cosine {c:.4f} rel {r:.4f}\n")
    for k, (c, r) in exact.items():
        assert c >= 0.99 and r <= 0.10, ("IA_ATTN_EXACT_DELTA=1", k, c, r)


def test_c2_full_width_one_tower(gpu):
    """C2: RobertaOneTower, roberta_large (24 layers, H = 1024, 16 heads), both items in ONE sequence of L = 2 x (50 + 205) = 510 tokens,
    cls / ce, B = 2 (reference src/models/text.py:1417-1492).  L = 510 is beyond the single-kernel attention backward (L <= 256): this
    is the configuration that runs the dQ / dK-dV kernel pair in every layer -- the shape class where round 3's rare NaN-dQ fault
    lived.  Loss, logits and the backward pass against the fp32 oracle, same six-point check as the C5 pair: the head, the first,
    a middle and the last layer's projections, and the embedding LayerNorm (the bottom of the backward chain)."""
    import item_alignment_amd.models as M
    from bench import roberta_large_config
    from oracle import ref_models as O
    L, B = 510, 2
    cfg = roberta_large_config(interaction_type="one_tower", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    rs = np.random.RandomState(21)
    ids = np.zeros((B, L), dtype=np.int64); mask = np.zeros((B, L), dtype=np.int64); tt = np.zeros((B, L), dtype=np.int64)
    for i, (n1, n2) in enumerate([(255, 255), (171, 98)]):            # one full-length pair, one ragged (241 padded positions)
        ids[i, 0] = 101
        ids[i, 1:n1 - 1] = rs.randint(1000, 21128, size=n1 - 2); ids[i, n1 - 1] = 102
        ids[i, n1:n1 + n2 - 1] = rs.randint(1000, 21128, size=n2 - 1); ids[i, n1 + n2 - 1] = 102
        mask[i, :n1 + n2] = 1
        tt[i, n1:n1 + n2] = 1
    labels = torch.tensor([1, 0])
    torch.manual_seed(77)
    model = M.RobertaOneTower(cfg)
    sd = state_of(model, head_gain=3.0)                   # dense and out_proj both: x 9 on the logits (x 400 gave |logits| ~ 10: the loss then
                                                          # amplifies the encoder's bf16 error by the head's gain)
    model = model.cuda().eval()
    t = [torch.from_numpy(a) for a in (ids, mask, tt)]
    model.param_arena.zero_grad()
    out = model(input_ids=t[0].cuda(), attention_mask=t[1].cuda(), token_type_ids=t[2].cuda(), position_ids=None, labels=labels.cuda(),
                output_hidden_states=True)
    out.loss.backward()
    torch.cuda.synchronize()

    def key(*parts):
        hits = [k for k in sd if all(p_ in k for p_ in parts)]
        assert len(hits) == 1, (parts, hits)
        return hits[0]
    keys = [key("classifier", "out_proj.weight"), key("layer.0.", "self.query.weight"), key("layer.11.", "self.key.weight"),
            key("layer.23.", "self.value.weight"), key("layer.23.", "intermediate.dense.weight"), key("embeddings.LayerNorm.weight")]
    rsd = {k: (v.requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    ref = O.roberta_one_tower(rsd, cfg, *t, None, labels, False)
    ref.loss.backward()
    params = dict(model.named_parameters())
    report = {k: (cosine(params[k].grad, rsd[k].grad), rel(params[k].grad, rsd[k].grad)) for k in keys}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "c2_full_width_gradients.txt"), "w") as f:
        for k, (c, r) in report.items():
            f.write(f"{k}: cosine {c:.4f} rel {r:.4f}\n")
        f.write(f"loss {out.loss.item():.4f} oracle {ref.loss.item():.4f} logits {out.logits.detach().float().cpu().tolist()} oracle {ref.logits.detach().tolist()}\n")
    print("C2 full-width gradients (cosine, rel):", report)
    assert abs(out.loss.item() - ref.loss.item()) < TOL * max(1.0, abs(ref.loss.item())), (out.loss.item(), ref.loss.item())
    # The encoder's output itself: the CLS rows of the last hidden state, 5e-2 of the tensor's maximum.
    h_got, h_ref = out.hidden_states[-1][:, 0].detach().float().cpu(), ref.hidden_states[-1][:, 0].detach()
    assert rel(h_got, h_ref) < TOL, rel(h_got, h_ref)
    # The head's logits are a small difference of large sums over those 1024 features (dense x 3, tanh, out_proj x 3).  The bar: 5e-2 of the
    # largest logit PLUS the distance the oracle itself moves when it stores bf16 where the engine does (oracle.rounding), measured here on
    # the same inputs -- an additive noise floor, not a multiplier.  Measured: HIP 0.050 from fp32, the bf16 oracle 0.028 from fp32 (and
    # 0.072 from HIP: three scattered samples of the same rounding noise on a logit of 0.39).
    scale = max(1.0, ref.logits.detach().abs().max().item())
    err32 = (out.logits.detach().float().cpu() - ref.logits.detach()).abs().max().item() / scale
    floor16 = 0.0
    if err32 >= TOL:
        with torch.no_grad(), O.rounding(torch.bfloat16):
            ref16 = O.roberta_one_tower({k: v.detach() for k, v in sd.items()}, cfg, *t, None, labels, False)
        floor16 = (ref16.logits - ref.logits.detach()).abs().max().item() / scale
    assert err32 < TOL + floor16, (err32, floor16, out.logits, ref.logits)
    assert (out.probs.detach().float().cpu() - ref.probs.detach()).abs().max().item() < TOL
    for k, (c, r) in report.items():
        assert c >= 0.99, (k, c, r)
        assert r <= 0.10, (k, c, r)
    for k, v in params.items():
        if v.grad is not None:
            assert torch.isfinite(v.grad).all(), k
