"""TEST INFRASTRUCTURE — CPU oracle: a plain PyTorch fp32 restatement of the reference's train-step
arithmetic (sunzeyeah/item-alignment, finetune_{text,image,multimodal}.py forward path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (item_alignment_amd/) never does.  Every function cites the reference lines it restates
(paths relative to the reference tree).  Functional style on purpose: models are a state_dict (same
key names as the reference) + a config namespace, so the same seeded weights
load into the reference classes (oracle/gen_golden.py), this oracle, and the HIP engine.

Pinning: tests/test_oracle_golden.py checks these functions against tests/golden/*.npz captured from
the reference's own classes in the build container (text towers, PKGM, image-embedding towers,
TextCNN, CoCa sum / cross_attn, heads and losses, and - oracle/gen_golden_r2.py - the image two-tower wrapper classes and a
24-layer roberta_large stack).  The ViT, NFNet and ResNetV2 encoders are third-party code (timm==0.6.5, absent offline): they
are restated from the published timm definitions and stay "parity unpinned by the reference"; the ViT restatement is additionally
cross-checked against transformers.ViTModel (tests/golden/vit_hf_crosscheck.npz, test_vit_restatement_against_transformers_vit).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------- basics


# Storage-rounding mode (tests only): inside `with rounding(torch.bfloat16):` the text-encoder functions round their tensors where
# the HIP engine keeps them in bf16 -- GEMM weights (the engine multiplies bf16 shadows of the fp32 masters), every linear / LayerNorm /
# GELU output, the attention probabilities fed to P V -- while accumulation, softmax, LayerNorm statistics and the loss stay fp32, as
# on the GPU.  The casts are differentiable (the gradient passes through, itself rounded), so the gradients of this mode carry the
# noise bf16 storage puts on an fp32 computation: the yardstick the GPU gradient tolerances are justified against
# (tests/test_models_gpu.py).  It does not reproduce the engine's roundings bit for bit (accumulation orders differ).
# `flash_delta=True` additionally differentiates the attention core the way a flash-style backward does: dS = P o (dP - delta) with
# delta = rowsum(dO o O) taken from the ROUNDED context the forward stored, instead of autograd's exact rowsum(P o dP) -- the one place
# where bf16 storage changes the backward's FORMULA and not just its operands (tools/c5_delta_probe.py measures what it
# does to the last layer's query projection).
_ROUND = None
_FLASH_DELTA = False


class rounding:
    def __init__(self, dtype, flash_delta=False):
        self.dtype = dtype
        self.flash_delta = flash_delta

    def __enter__(self):
        global _ROUND, _FLASH_DELTA
        self.prev, _ROUND = (_ROUND, _FLASH_DELTA), self.dtype
        _FLASH_DELTA = self.flash_delta

    def __exit__(self, *exc):
        global _ROUND, _FLASH_DELTA
        _ROUND, _FLASH_DELTA = self.prev


class _FlashStyleCore(torch.autograd.Function):
    """softmax(s) v with the probabilities and the context rounded for storage; backward from the stored (rounded) context."""

    @staticmethod
    def forward(ctx, s, v, dtype):
        p = torch.softmax(s, dim=-1)
        pr = p.to(dtype).to(s.dtype)
        o = torch.matmul(pr, v).to(dtype).to(s.dtype)
        ctx.save_for_backward(p, pr, v, o)
        return o

    @staticmethod
    def backward(ctx, do):
        p, pr, v, o = ctx.saved_tensors
        dv = torch.matmul(pr.transpose(-1, -2), do)
        dp = torch.matmul(do, v.transpose(-1, -2))
        delta = (do * o).sum(-1, keepdim=True)               # from the rounded o: the flash-style identity
        return p * (dp - delta), dv, None


def _r(x):
    return x if _ROUND is None else x.to(_ROUND).to(torch.float32)


def linear(x, sd, prefix, bias=True):
    return _r(F.linear(x, _r(sd[prefix + ".weight"]), sd[prefix + ".bias"] if bias and (prefix + ".bias") in sd else None))


def layer_norm(x, sd, prefix, eps):
    return _r(F.layer_norm(x, x.shape[-1:], sd[prefix + ".weight"], sd[prefix + ".bias"], eps))


def dropout(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0) else x


def embed(sd, prefix, ids, padding_idx=None):
    """nn.Embedding lookup; padding_idx rows receive no gradient (base.py:213,234-236)."""
    return F.embedding(ids, sd[prefix + ".weight"], padding_idx=padding_idx)


def create_position_ids_from_input_ids(input_ids, padding_idx):
    """base.py:189-202."""
    mask = input_ids.ne(padding_idx).int()
    incremental = torch.cumsum(mask, dim=1).type_as(mask) * mask
    return incremental.long() + padding_idx


def extended_attention_mask(attention_mask, dtype=torch.float32):
    """transformers get_extended_attention_mask (called at text.py:1213): (1-m) * finfo.min, [B,1,1,L]."""
    m = attention_mask[:, None, None, :].to(dtype)
    return (1.0 - m) * torch.finfo(dtype).min


# ------------------------------------------------------------------------------- RoBERTa/BERT encoder


def bert_self_attention(x, sd, p, cfg, ext_mask, training):
    """transformers RobertaSelfAttention, eager path (called through text.py:1241)."""
    B, L, H = x.shape
    nh = cfg.num_attention_heads
    dh = H // nh
    q = linear(x, sd, p + ".self.query").view(B, L, nh, dh).transpose(1, 2)
    k = linear(x, sd, p + ".self.key").view(B, L, nh, dh).transpose(1, 2)
    v = linear(x, sd, p + ".self.value").view(B, L, nh, dh).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    if ext_mask is not None:
        s = s + ext_mask
    if _FLASH_DELTA and _ROUND is not None and not (training and cfg.attention_probs_dropout_prob > 0):
        ctx = _FlashStyleCore.apply(s, v, _ROUND).transpose(1, 2).reshape(B, L, H)
    else:
        a = _r(dropout(torch.softmax(s, dim=-1), cfg.attention_probs_dropout_prob, training))
        ctx = _r(torch.matmul(a, v).transpose(1, 2).reshape(B, L, H))
    # RobertaSelfOutput: dense -> dropout -> LayerNorm(h + input)
    h = dropout(linear(ctx, sd, p + ".output.dense"), cfg.hidden_dropout_prob, training)
    return layer_norm(h + x, sd, p + ".output.LayerNorm", cfg.layer_norm_eps)


def bert_layer(x, sd, p, cfg, ext_mask, training):
    """transformers RobertaLayer = attention + RobertaIntermediate (erf GELU) + RobertaOutput."""
    a = bert_self_attention(x, sd, p + ".attention", cfg, ext_mask, training)
    h = _r(F.gelu(linear(a, sd, p + ".intermediate.dense")))
    h = dropout(linear(h, sd, p + ".output.dense"), cfg.hidden_dropout_prob, training)
    return layer_norm(h + a, sd, p + ".output.LayerNorm", cfg.layer_norm_eps)


def bert_encoder(x, sd, p, cfg, attention_mask, training=False):
    """RobertaEncoder: returns the tuple hidden_states = (embedding output, layer 1 .. layer N)."""
    ext = extended_attention_mask(attention_mask, x.dtype) if attention_mask is not None else None
    hs = [x]
    for i in range(cfg.num_hidden_layers):
        x = bert_layer(x, sd, f"{p}.layer.{i}", cfg, ext, training)
        hs.append(x)
    return hs


def roberta_embeddings(sd, p, cfg, input_ids, token_type_ids, position_ids, training=False, inputs_embeds=None):
    """base.py:238-279 RobertaEmbeddings.forward."""
    pad = cfg.pad_token_id
    if position_ids is None:
        position_ids = create_position_ids_from_input_ids(input_ids, pad)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    if inputs_embeds is None:
        inputs_embeds = embed(sd, p + ".word_embeddings", input_ids, pad)
    e = inputs_embeds + sd[p + ".token_type_embeddings.weight"][token_type_ids]
    e = e + embed(sd, p + ".position_embeddings", position_ids, pad)
    e = layer_norm(e, sd, p + ".LayerNorm", cfg.layer_norm_eps)
    return dropout(e, cfg.hidden_dropout_prob, training)


def roberta_model(sd, p, cfg, input_ids, attention_mask, token_type_ids, position_ids, training=False):
    """text.py:1137-1266 RobertaModel.forward (no pooler): hidden_states tuple."""
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    e = roberta_embeddings(sd, p + ".embeddings", cfg, input_ids, token_type_ids, position_ids, training)
    return bert_encoder(e, sd, p + ".encoder", cfg, attention_mask, training)


# ------------------------------------------------------------------------------------ heads and losses


def two_tower_head(sd, p, f1, f2, drop_p, training):
    """base.py:103-117 TwoTowerClassificationHead: dropout both, Linear(cat), Softmax() (implicit dim=1)."""
    x, y = dropout(f1, drop_p, training), dropout(f2, drop_p, training)
    logits = linear(torch.cat((x, y), dim=1), sd, p + ".out_proj")
    return x, y, logits, torch.softmax(logits, dim=1)


def roberta_cls_head(sd, p, cfg, features, training, inputs_embeds=None):
    """base.py:139-157 RobertaClassificationHead."""
    dp = cfg.classifier_dropout if getattr(cfg, "classifier_dropout", None) is not None else cfg.hidden_dropout_prob
    x = dropout(features[:, 0, :], dp, training)
    x = dropout(torch.tanh(linear(x, sd, p + ".dense")), dp, training)
    if getattr(cfg, "ensemble", None) == "end":
        y = dropout(torch.cat(inputs_embeds, dim=-1), dp, training)
        y = dropout(torch.tanh(linear(y, sd, p + ".dense_img")), dp, training)
        return linear(torch.cat((x, y), dim=-1), sd, p + ".out_proj")
    return linear(x, sd, p + ".out_proj")


def vec_sim_head(sd, p, cfg, f1, f2, training):
    """base.py:66-88 VecSimClassificationHead (+ InnerProduct base.py:29-34)."""
    dp = cfg.classifier_dropout if getattr(cfg, "classifier_dropout", None) is not None else cfg.hidden_dropout_prob
    x = dropout(torch.tanh(linear(dropout(f1, dp, training), sd, p + ".dense")), dp, training)
    y = dropout(torch.tanh(linear(dropout(f2, dp, training), sd, p + ".dense")), dp, training)
    sm = cfg.similarity_measure
    if sm == "cosine":
        sim = F.cosine_similarity(x, y)
        probs = (sim + 1) / 2
    elif sm in ("l1", "l2"):
        sim = F.pairwise_distance(x, y, p=1 if sm == "l1" else 2)
        probs = torch.exp(-sim)
    elif sm == "inner_product":
        sim = (x * y).sum(-1)
        probs = torch.sigmoid(sim)
    else:
        raise ValueError(f"Unsupported similarty measure: {sm}")
    return x, y, sim, probs


def pair_loss(cfg, logits, labels, src_embeds, tgt_embeds):
    """Loss dispatch shared by every tower (text.py:1283-1292 ctor, :1356-1364 use); loss.py:61-68,126-134."""
    lt = cfg.loss_type
    if lt == "cosine":
        return F.cosine_embedding_loss(src_embeds, tgt_embeds, (labels * 2 - 1).view(-1), margin=cfg.loss_margin)
    if lt == "ce":
        return F.cross_entropy(logits.view(-1, cfg.num_labels), labels.view(-1))
    if lt == "hinge":
        t = (labels * 2 - 1).view(-1)
        return torch.clamp(cfg.loss_margin - logits.view(-1) * t, min=0).mean()
    if lt == "euclidean":
        return torch.pow(logits.view(-1), (labels * 2 - 1).view(-1)).mean()   # quirk A5: pow(input, target)
    # "bce" and anything else falls to the reference's else branch: loss_fct(logits.view(-1), labels.view(-1))
    if lt == "bce":
        return F.binary_cross_entropy_with_logits(logits.view(-1), labels.view(-1).to(logits.dtype))
    return F.cross_entropy(logits.view(-1), labels.view(-1))


def _out(loss, logits, probs, src, tgt, **extra):
    return SimpleNamespace(loss=loss, logits=logits, probs=probs, src_embeds=src, tgt_embeds=tgt, **extra)


def _one_tower_tail(sd, cfg, hs, labels, max_seq_len, training, inputs_embeds=None):
    """text.py:1452-1477 (identical block in multimodal.py:283-308, text.py:755-775)."""
    cls_layers = [-int(i) for i in cfg.cls_layers.split(",")]
    seqs = [hs[i] for i in cls_layers]
    seq = torch.stack(seqs).mean(dim=0) if cfg.cls_pool == "avg" else torch.cat(seqs, dim=-1)
    if cfg.classification_method == "vec_sim":
        src, tgt, logits, probs = vec_sim_head(sd, "classifier", cfg, seq[:, 0, :], seq[:, max_seq_len, :], training)
    else:
        logits = roberta_cls_head(sd, "classifier", cfg, seq, training, inputs_embeds)
        probs = torch.softmax(logits, dim=1)
        src, tgt, probs = probs[:, 0], probs[:, 1], probs[:, 1]
    loss = pair_loss(cfg, logits, labels, src, tgt) if labels is not None else None
    return _out(loss, logits, probs, src, tgt, hidden_states=hs)


def _max_seq_len(cfg):
    if cfg.max_seq_len_pv is None:
        return cfg.max_seq_len
    if cfg.max_seq_len is None:
        return cfg.max_seq_len_pv
    return cfg.max_seq_len + cfg.max_seq_len_pv


def auxiliary_task_pair(sd, p, cfg, sequence_output, pair_indices, training=False):
    """text.py:79-102 AuxiliaryTaskPair.forward: span means of the (source, target) attribute of every pair row
    (src start, src end, tgt start, tgt end, label) -> dropout -> Linear(2H -> num_labels).  Returns (logits, labels)."""
    x, y, labels = [], [], []
    for i, rows in enumerate(pair_indices):
        for r in rows:
            a0, a1, b0, b1, lab = (int(v) for v in r)
            x.append(sequence_output[i, a0:a1, :].mean(dim=0))
            y.append(sequence_output[i, b0:b1, :].mean(dim=0))
            labels.append(lab)
    x, y = torch.stack(x), torch.stack(y)
    drop = getattr(cfg, "classifier_dropout", None)
    drop = drop if drop is not None else cfg.hidden_dropout_prob
    x, y = dropout(x, drop, training), dropout(y, drop, training)
    logits = F.linear(torch.cat((x, y), dim=1), sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])
    return logits, torch.tensor(labels, dtype=torch.long)


def roberta_one_tower(sd, cfg, input_ids, attention_mask, token_type_ids, position_ids=None, labels=None, training=False,
                      pair_indices=None):
    """text.py:1417-1492 RobertaOneTower.forward (with `auxiliary_task`, :1478-1480: + CE over the attribute pairs)."""
    hs = roberta_model(sd, "roberta", cfg, input_ids, attention_mask, token_type_ids, position_ids, training)
    out = _one_tower_tail(sd, cfg, hs, labels, _max_seq_len(cfg), training)
    if labels is not None and getattr(cfg, "auxiliary_task", False):
        cls_layers = [-int(i) for i in cfg.cls_layers.split(",")]
        seqs = [hs[i] for i in cls_layers]
        seq = torch.stack(seqs).mean(dim=0) if cfg.cls_pool == "avg" else torch.cat(seqs, dim=-1)
        logits2, labels2 = auxiliary_task_pair(sd, "auxiliary_task", cfg, seq, pair_indices, training)
        out.loss = out.loss + F.cross_entropy(logits2.view(-1, cfg.num_labels), labels2.view(-1))
    return out


def roberta_two_tower(sd, cfg, ids1, mask1, tt1, pos1, ids2, mask2, tt2, pos2, labels=None, training=False):
    """text.py:1298-1376 RobertaTwoTower.forward: same weights, two passes; probs stay [B,2] (quirk A3)."""
    h1 = roberta_model(sd, "roberta", cfg, ids1, mask1, tt1, pos1, training)[-1]
    h2 = roberta_model(sd, "roberta", cfg, ids2, mask2, tt2, pos2, training)[-1]
    src, tgt, logits, probs = two_tower_head(sd, "classifier", h1[:, 0, :], h2[:, 0, :], cfg.hidden_dropout_prob, training)
    loss = pair_loss(cfg, logits, labels, src, tgt) if labels is not None else None
    return _out(loss, logits, probs, src, tgt)


# ------------------------------------------------------------------------------------------------ PKGM


def pkgm_kg_embeddings(sd, p, cfg, input_ids):
    """base.py:347-392: entity/relation gathers; F.normalize over dim=1 of a [B,1,D] tensor == sign(x)
    (quirk A1); triple query h + r and relation query M h - r."""
    S, P = cfg.max_seq_len, cfg.max_pvs
    one = cfg.interaction_type == "one_tower"
    ent, rel = sd[p + ".ent_emb.weight"], sd[p + ".rel_emb.weight"]

    def side(ent_col, rel_lo, rel_hi):
        h = F.normalize(ent[input_ids[:, ent_col].unsqueeze(1)])          # dim=1 (size 1) -> sign
        r = rel[input_ids[:, rel_lo:rel_hi]]
        hp = linear(h, sd, p + ".proj_mat", bias=getattr(cfg, "entity_projection_bias", False))
        if (p + ".entity_embedding_projetor.weight") in sd:
            h = linear(h, sd, p + ".entity_embedding_projetor")
            r = linear(r, sd, p + ".relation_embedding_projetor")
            hp = linear(hp, sd, p + ".entity_projection_projetor")
        return torch.cat((h + r, hp - r), dim=1)

    src = side(S, S + 1, P + S + 1)
    tgt = side(2 * S + P + 1, 2 * S + P + 2, input_ids.shape[1]) if one else None
    return src, tgt


def pkgm_embeddings(sd, p, cfg, input_ids, token_type_ids, position_ids, training=False):
    """base.py:394-442 RobertaPKGMEmbeddings.forward."""
    S, P = cfg.max_seq_len, cfg.max_pvs
    pad = cfg.pad_token_id
    word = lambda ids: embed(sd, p + ".word_embeddings", ids, pad)
    src_kg, tgt_kg = pkgm_kg_embeddings(sd, p, cfg, input_ids)
    if cfg.interaction_type == "one_tower":
        e = torch.cat((word(input_ids[:, :S]), src_kg, word(input_ids[:, S + P + 1:2 * S + P + 1]), tgt_kg), dim=1)
    else:
        e = torch.cat((word(input_ids[:, :S]), src_kg), dim=1)
    e = e + sd[p + ".token_type_embeddings.weight"][token_type_ids]
    e = e + embed(sd, p + ".position_embeddings", position_ids, pad)
    e = layer_norm(e, sd, p + ".LayerNorm", cfg.layer_norm_eps)
    return dropout(e, cfg.hidden_dropout_prob, training)


def pkgm_model(sd, p, cfg, input_ids, attention_mask, token_type_ids, position_ids, training=False):
    """text.py:178-289 RobertaPKGMModel.forward."""
    e = pkgm_embeddings(sd, p + ".embeddings", cfg, input_ids, token_type_ids, position_ids, training)
    return bert_encoder(e, sd, p + ".encoder", cfg, attention_mask, training)


def pkgm_one_tower(sd, cfg, input_ids, attention_mask, token_type_ids, position_ids, labels=None, training=False):
    """text.py:720-783 PKGMOneTower.forward (cls head on the last layer; vec_sim uses token S + 2P)."""
    hs = pkgm_model(sd, "roberta", cfg, input_ids, attention_mask, token_type_ids, position_ids, training)
    seq = hs[-1]
    if cfg.classification_method == "vec_sim":
        src, tgt, logits, probs = vec_sim_head(sd, "classifier", cfg, seq[:, 0, :], seq[:, cfg.max_seq_len + 2 * cfg.max_pvs, :], training)
    else:
        logits = roberta_cls_head(sd, "classifier", cfg, seq, training)
        probs = torch.softmax(logits, dim=1)
        src, tgt, probs = probs[:, 0], probs[:, 1], probs[:, 1]
    loss = pair_loss(cfg, logits, labels, src, tgt) if labels is not None else None
    return _out(loss, logits, probs, src, tgt, hidden_states=hs)


def pkgm_two_tower(sd, cfg, ids1, mask1, tt1, pos1, ids2, mask2, tt2, pos2, labels=None, training=False):
    """text.py:321-391 PKGMTwoTower.forward."""
    h1 = pkgm_model(sd, "roberta", cfg, ids1, mask1, tt1, pos1, training)[-1]
    h2 = pkgm_model(sd, "roberta", cfg, ids2, mask2, tt2, pos2, training)[-1]
    src, tgt, logits, probs = two_tower_head(sd, "classifier", h1[:, 0, :], h2[:, 0, :], cfg.hidden_dropout_prob, training)
    loss = pair_loss(cfg, logits, labels, src, tgt) if labels is not None else None
    return _out(loss, logits, probs, src, tgt)


# ------------------------------------------------------------------- RoBERTa + pre-extracted image embeddings


def image_embeddings(sd, p, cfg, input_ids, token_type_ids, position_ids, inputs_embeds, attention_mask, image_indices,
                     training=False):
    """base.py:501-556 RobertaImageEmbeddings.forward (ensemble == "begin"): position ids come from the
    attention mask (:508); image rows from img2txt replace token 1 (and token image_index, one-tower)."""
    pad = cfg.pad_token_id
    if position_ids is None:
        position_ids = create_position_ids_from_input_ids(attention_mask, pad)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    txt = embed(sd, p + ".word_embeddings", input_ids, pad)
    if cfg.interaction_type == "one_tower":
        img = linear(torch.stack(inputs_embeds, dim=1), sd, p + ".img2txt")
        rows = []
        for i, idx in enumerate(image_indices):
            idx = int(idx)
            rows.append(torch.cat((txt[i][0:1], img[i][0:1], txt[i][2:idx], img[i][1:2], txt[i][idx + 1:]), dim=0))
        e = torch.stack(rows)
    else:
        img = linear(inputs_embeds, sd, p + ".img2txt")
        e = torch.cat([txt[:, 0:1, :], img.unsqueeze(1), txt[:, 2:, :]], dim=1)
    e = e + sd[p + ".token_type_embeddings.weight"][token_type_ids]
    e = e + embed(sd, p + ".position_embeddings", position_ids, pad)
    e = layer_norm(e, sd, p + ".LayerNorm", cfg.layer_norm_eps)
    return dropout(e, cfg.hidden_dropout_prob, training)


def roberta_image_model(sd, p, cfg, input_ids, attention_mask, token_type_ids, position_ids, inputs_embeds, image_indices,
                        training=False):
    """multimodal.py:72-210 RobertaImageModel.forward."""
    if cfg.ensemble == "begin":
        e = image_embeddings(sd, p + ".embeddings", cfg, input_ids, token_type_ids, position_ids, inputs_embeds, attention_mask,
                             image_indices, training)
    else:
        e = roberta_embeddings(sd, p + ".embeddings", cfg, input_ids, token_type_ids, position_ids, training)
    return bert_encoder(e, sd, p + ".encoder", cfg, attention_mask, training)


def roberta_image_one_tower(sd, cfg, input_ids, attention_mask, token_type_ids, position_ids, inputs_embeds, image_indices,
                            labels=None, training=False):
    """multimodal.py:249-320 RobertaImageOneTower.forward."""
    hs = roberta_image_model(sd, "roberta", cfg, input_ids, attention_mask, token_type_ids, position_ids, inputs_embeds,
                             image_indices, training)
    return _one_tower_tail(sd, cfg, hs, labels, _max_seq_len(cfg), training, inputs_embeds)


def roberta_image_two_tower(sd, cfg, ids1, mask1, tt1, pos1, img1, ids2, mask2, tt2, pos2, img2, labels=None, training=False):
    """multimodal.py:362-461 RobertaImageTwoTower.forward."""
    h1 = roberta_image_model(sd, "roberta", cfg, ids1, mask1, tt1, pos1, img1, None, training)[-1]
    h2 = roberta_image_model(sd, "roberta", cfg, ids2, mask2, tt2, pos2, img2, None, training)[-1]
    src, tgt, logits, probs = two_tower_head(sd, "classifier", h1[:, 0, :], h2[:, 0, :], cfg.hidden_dropout_prob, training)
    loss = pair_loss(cfg, logits, labels, src, tgt) if labels is not None else None
    return _out(loss, logits, probs, src, tgt)


# --------------------------------------------------------------------------------------------- TextCNN


def textcnn(sd, p, cfg, ids, training=False):
    """text.py:1516-1527 TextCNN.forward: two embedding pipelines stacked as 2 channels, 4 Conv2d (K x H),
    ReLU, max over time, concat, dropout."""
    x1 = roberta_embeddings(sd, p + ".embedding1", cfg, ids, None, None, training)
    x2 = roberta_embeddings(sd, p + ".embedding2", cfg, ids, None, None, training)
    x = torch.stack((x1, x2), dim=1)
    outs = []
    for i, _ in enumerate(cfg.filter_sizes.split(",")):
        c = F.relu(F.conv2d(x, sd[f"{p}.convs1.{i}.weight"], sd[f"{p}.convs1.{i}.bias"])).squeeze(3)
        outs.append(F.max_pool1d(c, c.size(2)).squeeze(2))
    return dropout(torch.cat(outs, 1), cfg.hidden_dropout_prob, training)


def textcnn_two_tower(sd, cfg, ids1, ids2, labels=None, training=False):
    """text.py:1555-1609 TextCNNTwoTower.forward."""
    o1, o2 = textcnn(sd, "textcnn", cfg, ids1, training), textcnn(sd, "textcnn", cfg, ids2, training)
    if cfg.classification_method == "vec_sim":
        src, tgt, logits, probs = vec_sim_head(sd, "classifier", cfg, o1, o2, training)
    else:
        src, tgt, logits, probs = two_tower_head(sd, "classifier", o1, o2, cfg.hidden_dropout_prob, training)
    src, tgt, probs = probs[:, 0], probs[:, 1], probs[:, 1]
    loss = pair_loss(cfg, logits, labels, src, tgt) if labels is not None else None
    return _out(loss, logits, probs, src, tgt)


# ------------------------------------------------------------------------------------------------- ViT


def vit_forward_features(sd, p, vcfg, images):
    """timm==0.6.5 VisionTransformer.forward_features [third party, restated from the public definition]:
    PatchEmbed conv(P, stride P) -> flatten -> cls concat -> + pos_embed -> pre-LN blocks -> final norm.
    vcfg: embed_dim, depth, num_heads, patch_size, eps (1e-6)."""
    B = images.shape[0]
    x = F.conv2d(images, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=vcfg.patch_size)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd[p + ".cls_token"].expand(B, -1, -1), x), dim=1) + sd[p + ".pos_embed"]
    nh = vcfg.num_heads
    for i in range(vcfg.depth):
        b = f"{p}.blocks.{i}"
        h = layer_norm(x, sd, b + ".norm1", vcfg.eps)
        Bq, N, C = h.shape
        qkv = linear(h, sd, b + ".attn.qkv").reshape(Bq, N, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        a = torch.softmax((q @ k.transpose(-2, -1)) * ((C // nh) ** -0.5), dim=-1)
        h = (a @ v).transpose(1, 2).reshape(Bq, N, C)
        x = x + linear(h, sd, b + ".attn.proj")
        h = layer_norm(x, sd, b + ".norm2", vcfg.eps)
        x = x + linear(F.gelu(linear(h, sd, b + ".mlp.fc1")), sd, b + ".mlp.fc2")
    return layer_norm(x, sd, p + ".norm", vcfg.eps)


def vit_forward_head(tokens):
    """timm forward_head(x, pre_logits=True) with global_pool='token', fc_norm=None: the cls token."""
    return tokens[:, 0]


# ------------------------------------------------------------------------------------------------ CoCa


def _coca_ln(x, sd, p):
    """multimodal.py:475-482 LayerNorm with learned gamma and a zero beta buffer."""
    return F.layer_norm(x, x.shape[-1:], sd[p + ".gamma"], sd[p + ".beta"])


def _rotate_half(x):
    """multimodal.py:509-512."""
    x = x.reshape(*x.shape[:-1], 2, x.shape[-1] // 2)
    x1, x2 = x.unbind(dim=-2)
    return torch.cat((-x2, x1), dim=-1)


def parallel_transformer_block(x, sd, p, heads, dim_head, ff_mult):
    """multimodal.py:572-626 ParallelTransformerBlock.forward (is_decoding False): one fused projection ->
    q (h heads), single k/v head (multi-query), SwiGLU feed-forward in parallel; rotary on q, k."""
    n, dim = x.shape[1], x.shape[2]
    h = _coca_ln(x, sd, p + ".norm")
    ff_inner = dim * ff_mult
    q, k, v, ff = F.linear(h, sd[p + ".fused_attn_ff_proj.weight"]).split((dim_head * heads, dim_head, dim_head, ff_inner * 2), dim=-1)
    q = q.reshape(q.shape[0], n, heads, dim_head).transpose(1, 2)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim_head, 2).float() / dim_head))
    freqs = torch.einsum("i,j->ij", torch.arange(n, dtype=inv_freq.dtype), inv_freq)
    pos = torch.cat((freqs, freqs), dim=-1)
    q = q * pos.cos() + _rotate_half(q) * pos.sin()
    k = k * pos.cos() + _rotate_half(k) * pos.sin()
    q = q * dim_head ** -0.5
    sim = torch.einsum("bhid,bjd->bhij", q, k)
    sim = sim - sim.amax(dim=-1, keepdim=True).detach()
    out = torch.einsum("bhij,bjd->bhid", sim.softmax(dim=-1), v)
    out = out.transpose(1, 2).reshape(out.shape[0], n, heads * dim_head)
    xg, gate = ff.chunk(2, dim=-1)
    return F.linear(out, sd[p + ".attn_out.weight"]) + F.linear(F.silu(gate) * xg, sd[p + ".ff_out.1.weight"])


def cross_attention(x, ctx, sd, p, heads, dim_head):
    """multimodal.py:665-706 CrossAttention.forward with parallel_ff=True, norm_context=False."""
    h = _coca_ln(x, sd, p + ".norm")
    q = F.linear(h, sd[p + ".to_q.weight"])
    q = q.reshape(q.shape[0], q.shape[1], heads, dim_head).transpose(1, 2) * dim_head ** -0.5
    k, v = F.linear(ctx, sd[p + ".to_kv.weight"]).chunk(2, dim=-1)
    sim = torch.einsum("bhid,bjd->bhij", q, k)
    sim = sim - sim.amax(dim=-1, keepdim=True)
    out = torch.einsum("bhij,bjd->bhid", sim.softmax(dim=-1), v)
    out = out.transpose(1, 2).reshape(out.shape[0], out.shape[2], heads * dim_head)
    out = F.linear(out, sd[p + ".to_out.weight"])
    xg, gate = F.linear(h, sd[p + ".ff.0.weight"]).chunk(2, dim=-1)
    return out + F.linear(F.silu(gate) * xg, sd[p + ".ff.2.weight"])


def coca_item_alignment(sd, cfg, vcfg, ids1, mask1, tt1, pos1, img1, ids2, mask2, tt2, pos2, img2, labels=None,
                        training=False, image_tower=None):
    """multimodal.py:983-1045 CoCaForItemAlignment.forward.  image_tower(images) -> tokens lets tests
    substitute the fake encoder used when capturing golden vectors (timm is absent offline)."""
    tower = image_tower or (lambda im: vit_forward_features(sd, "coca.img_encoder", vcfg, im))

    def side(ids, mask, tt, pos, img):
        hs = roberta_model(sd, "coca.text_encoder", cfg, ids, mask, tt, pos, training)[-1]
        tok = tower(img)
        ie = vit_forward_head(tok)
        if "img_proj.weight" in sd:      # product-side deviation N1: widths differ in the named C5 pairing
            ie = linear(ie, sd, "img_proj")
        return hs[:, 0], hs, ie, tok

    te1, tt_1, ie1, it1 = side(ids1, mask1, tt1, pos1, img1)
    te2, tt_2, ie2, it2 = side(ids2, mask2, tt2, pos2, img2)
    if cfg.ensemble == "cross_attn":
        heads = cfg.num_attention_heads_multimodal
        dh = cfg.hidden_size // heads
        for i in range(cfg.num_hidden_layers_multimodal):
            tt_1 = parallel_transformer_block(tt_1, sd, f"multimodal_layers.{i}.0.fn", heads, dh, cfg.feedforward_multiplication_multimodal) + tt_1
            tt_1 = cross_attention(tt_1, it1, sd, f"multimodal_layers.{i}.1.fn", heads, dh) + tt_1
        e1 = tt_1[:, 0]
        e2 = tt_1[:, 0]      # quirk A4 (multimodal.py:1013): the target embedding is the SOURCE tower's token
    else:
        e1, e2 = te1 + ie1, te2 + ie2
    if cfg.classification_method == "vec_sim":
        src, tgt, logits, probs = vec_sim_head(sd, "classifier", cfg, e1, e2, training)
    else:
        src, tgt, logits, probs = two_tower_head(sd, "classifier", e1, e2, cfg.hidden_dropout_prob, training)
    src, tgt, probs = probs[:, 0], probs[:, 1], probs[:, 1]
    loss = pair_loss(cfg, logits, labels, src, tgt) if labels is not None else None
    return _out(loss, logits, probs, src, tgt)


def image_two_tower(sd, cfg, feats1, feats2, labels=None, training=False):
    """image.py:253-294 / :454-499 NFNetTwoTower / VitTwoTower.forward after the encoder: pooled
    features -> TwoTowerClassificationHead -> probs[:,1] -> loss."""
    src, tgt, logits, probs = two_tower_head(sd, "classifier", feats1, feats2, cfg.hidden_dropout_prob, training)
    src, tgt, probs = probs[:, 0], probs[:, 1], probs[:, 1]
    loss = pair_loss(cfg, logits, labels, src, tgt) if labels is not None else None
    return _out(loss, logits, probs, src, tgt)


# ----------------------------------------------------------------------------------------------- NF-Net
# timm==0.6.5 `models/nfnet.py` + `layers/std_conv.py` + `layers/eca.py` restated from their published definitions
# (timm is pinned by requirements.txt:13 but absent offline: PARITY UNPINNED by the reference for this tower).
# The stage / stride / beta / expected-variance bookkeeping is the part the reference mirrors in-tree
# (src/models/image.py:81-142) and is followed literally.

NONLIN_GAMMA_SILU = 1.7881293296813965      # timm nfnet.py _nonlin_gamma['silu']


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def nfnet_cfg(name="eca_nfnet_l0"):
    """timm nfnet.py model_cfgs via _nfnet_cfg: deep_quad stem 128, group_size 64, bottle_ratio 0.25, extra_conv,
    num_features = channels[-1] * feat_mult, act silu, attn eca, alpha 0.2, attn_gain 2.0, std_conv_eps 1e-5."""
    from types import SimpleNamespace
    table = {
        "eca_nfnet_l0": dict(depths=(1, 2, 6, 3), channels=(256, 512, 1536, 1536), feat_mult=1.5),
        "eca_nfnet_l1": dict(depths=(2, 4, 12, 6), channels=(256, 512, 1536, 1536), feat_mult=2.0),
        "eca_nfnet_l2": dict(depths=(3, 6, 18, 9), channels=(256, 512, 1536, 1536), feat_mult=2.0),
    }
    t = table[name]
    return SimpleNamespace(depths=t["depths"], channels=t["channels"], stem_chs=128, group_size=64, bottle_ratio=0.25,
                           num_features=int(t["channels"][-1] * t["feat_mult"]), alpha=0.2, attn_gain=2.0, eps=1e-5, ch_div=8)


def nfnet_plan(cfg):
    """Per-block geometry exactly as reference image.py:98-137 derives it (output_stride 32, dilation 1 throughout):
    list of stages, each a list of dicts(in_chs, out_chs, mid_chs, groups, stride, beta, downsample)."""
    prev, expected_var, stages = cfg.stem_chs, 1.0, []
    for si, depth in enumerate(cfg.depths):
        stride = 1 if si == 0 else 2                       # stem stride 4 > 2 (image.py:99)
        blocks = []
        for bi in range(depth):
            out_chs = make_divisible(cfg.channels[si], cfg.ch_div)
            mid = make_divisible(out_chs * cfg.bottle_ratio, cfg.ch_div)        # reg=False: from out_chs (timm NormFreeBlock)
            groups = mid // cfg.group_size
            mid = cfg.group_size * groups
            s = stride if bi == 0 else 1
            blocks.append(dict(in_chs=prev, out_chs=out_chs, mid_chs=mid, groups=groups, stride=s, beta=1.0 / expected_var ** 0.5,
                               downsample=(prev != out_chs or s != 1)))
            if bi == 0:
                expected_var = 1.0
            expected_var += cfg.alpha ** 2
            prev = out_chs
        stages.append(blocks)
    return stages


def eca_kernel_size(channels, gamma=2, beta=1):
    """timm layers/eca.py EcaModule."""
    t = int(abs(math.log(channels, 2) + beta) / gamma)
    return max(t if t % 2 else t + 1, 3)


def scaled_std_conv(x, sd, p, stride=1, groups=1, eps=1e-5, gamma=NONLIN_GAMMA_SILU):
    """timm layers/std_conv.py ScaledStdConv2d.forward: weight standardised per output channel over its fan-in
    (biased variance, eps inside the sqrt) times gain * gamma * fan_in^-0.5; padding = get_padding(k, stride)."""
    w = sd[p + ".weight"]
    k = w.shape[-1]
    scale = gamma * w[0].numel() ** -0.5
    wf = w.reshape(w.shape[0], -1)
    mu = wf.mean(dim=1, keepdim=True)
    var = wf.var(dim=1, unbiased=False, keepdim=True)
    what = ((wf - mu) / torch.sqrt(var + eps) * (sd[p + ".gain"].reshape(-1, 1) * scale)).reshape_as(w)
    pad = ((stride - 1) + (k - 1)) // 2
    return F.conv2d(x, what, sd[p + ".bias"], stride, pad, 1, groups)


def eca(x, sd, p):
    """timm EcaModule.forward: GAP -> conv1d over channels -> sigmoid -> scale."""
    w = sd[p + ".conv.weight"]
    y = x.mean((2, 3)).unsqueeze(1)
    y = F.conv1d(y, w, padding=(w.shape[-1] - 1) // 2)
    return x * torch.sigmoid(y).reshape(x.shape[0], -1, 1, 1)


def nf_block(x, sd, p, blk, cfg):
    """timm NormFreeBlock.forward (reg=False, extra_conv=True, skipinit off, no drop path)."""
    out = F.silu(x) * blk["beta"]
    shortcut = x
    if blk["downsample"]:
        sc = out
        if blk["stride"] > 1:
            sc = F.avg_pool2d(sc, 2, blk["stride"], ceil_mode=True, count_include_pad=False)      # DownsampleAvg
        shortcut = scaled_std_conv(sc, sd, p + ".downsample.conv", eps=cfg.eps)
    out = scaled_std_conv(out, sd, p + ".conv1", eps=cfg.eps)
    out = scaled_std_conv(F.silu(out), sd, p + ".conv2", stride=blk["stride"], groups=blk["groups"], eps=cfg.eps)
    out = scaled_std_conv(F.silu(out), sd, p + ".conv2b", groups=blk["groups"], eps=cfg.eps)
    out = scaled_std_conv(F.silu(out), sd, p + ".conv3", eps=cfg.eps)
    out = cfg.attn_gain * eca(out, sd, p + ".attn_last")
    return out * cfg.alpha + shortcut


def nfnet_forward_features(sd, p, cfg, images):
    """reference image.py:191-199 NormFreeNet.forward_features: deep_quad stem (3x3 convs, strides 2,1,1,2, SiLU between),
    stages, final 1x1 conv, final SiLU.  images: [B, 3, S, S] fp32 -> [B, num_features, S/32, S/32]."""
    x = images
    chs = (cfg.stem_chs // 8, cfg.stem_chs // 4, cfg.stem_chs // 2, cfg.stem_chs)
    for i, s in enumerate((2, 1, 1, 2)):
        x = scaled_std_conv(x, sd, f"{p}.stem.conv{i + 1}", stride=s, eps=cfg.eps)
        if i != 3:
            x = F.silu(x)
    for si, blocks in enumerate(nfnet_plan(cfg)):
        for bi, blk in enumerate(blocks):
            x = nf_block(x, sd, f"{p}.stages.{si}.{bi}", blk, cfg)
    x = scaled_std_conv(x, sd, p + ".final_conv", eps=cfg.eps)
    return F.silu(x)


def nfnet_global_pool(x):
    """head.global_pool = SelectAdaptivePool2d('avg', flatten=True)."""
    return x.mean((2, 3))


def nfnet_state_spec(cfg, prefix="img_encoder"):
    """(key, shape) list with timm's NormFreeNet key names (conv weight / bias / gain; ECA conv.weight)."""
    spec = []

    def conv(name, cin, cout, k, groups=1):
        spec.extend([(f"{name}.weight", (cout, cin // groups, k, k)), (f"{name}.bias", (cout,)), (f"{name}.gain", (cout, 1, 1, 1))])
    chs = (cfg.stem_chs // 8, cfg.stem_chs // 4, cfg.stem_chs // 2, cfg.stem_chs)
    cin = 3
    for i, c in enumerate(chs):
        conv(f"{prefix}.stem.conv{i + 1}", cin, c, 3)
        cin = c
    for si, blocks in enumerate(nfnet_plan(cfg)):
        for bi, b in enumerate(blocks):
            q = f"{prefix}.stages.{si}.{bi}"
            if b["downsample"]:
                conv(q + ".downsample.conv", b["in_chs"], b["out_chs"], 1)
            conv(q + ".conv1", b["in_chs"], b["mid_chs"], 1)
            conv(q + ".conv2", b["mid_chs"], b["mid_chs"], 3, b["groups"])
            conv(q + ".conv2b", b["mid_chs"], b["mid_chs"], 3, b["groups"])
            conv(q + ".conv3", b["mid_chs"], b["out_chs"], 1)
            spec.append((q + ".attn_last.conv.weight", (1, 1, eca_kernel_size(b["out_chs"]))))
    conv(prefix + ".final_conv", cfg.channels[-1], cfg.num_features, 1)
    return spec


def nfnet_two_tower(sd, cfg, ncfg, images_1, images_2, labels=None, training=False):
    """reference image.py:253-294 NFNetTwoTower.forward."""
    f1 = nfnet_global_pool(nfnet_forward_features(sd, "img_encoder", ncfg, images_1))
    f2 = nfnet_global_pool(nfnet_forward_features(sd, "img_encoder", ncfg, images_2))
    return image_two_tower(sd, cfg, f1, f2, labels, training)


# ------------------------------------------------------------------------------------- ResNetV2 (pre-activation, BatchNorm)
# timm 0.6.5 resnetv2.py, `resnetv2_50` = ResNetV2(layers=[3, 4, 6, 3], conv_layer=create_conv2d, norm_layer=BatchNormAct2d):
# third-party, absent offline -> restated from the published definitions (parity unpinned, like the NFNet tower).  Called by
# the reference at finetune_image.py:191,215-216 and image.py:337-341.


# The BiT variants (`resnetv2_{50x1,50x3,101x1,101x3,152x2,152x4}_bitm[_in21k]`; finetune_image.py:23 names resnetv2_50x3_bitm_in21k) are
# timm resnetv2.py `_create_resnetv2_bit`: the same ResNetV2 class with stem_type='fixed', conv_layer=partial(StdConv2d, eps=1e-8),
# norm_layer=partial(GroupNormAct, num_groups=32) and every width (stem included) multiplied by width_factor.  cfg.bit selects them here.
# Cross-checked against an independent implementation of the same published architecture that IS installed (transformers.BitModel):
# oracle/gen_golden_r2.py bit_hf -> tests/golden/bit_hf_crosscheck.npz.


def resnetv2_cfg(name="resnetv2_50"):
    from types import SimpleNamespace
    depth = {"50": (3, 4, 6, 3), "101": (3, 4, 23, 3), "152": (3, 8, 36, 3)}
    if "_bit" in name:
        geo = name.split("_")[1]                      # "50x3"
        d, wf = geo.split("x")
        wf = int(wf)
        ch = tuple(make_divisible(c * wf) for c in (256, 512, 1024, 2048))
        return SimpleNamespace(layers=depth[d], channels=ch, stem_chs=make_divisible(64 * wf), bottle_ratio=0.25, eps=1e-5, momentum=0.1,
                               num_features=ch[-1], bit=True, std_eps=1e-8, groups=32, num_classes=21843 if name.endswith("in21k") else 1000)
    layers = depth[name.split("_")[1]]
    return SimpleNamespace(layers=layers, channels=(256, 512, 1024, 2048), stem_chs=64, bottle_ratio=0.25, eps=1e-5, momentum=0.1,
                           num_features=2048)


def resnetv2_plan(cfg):
    """ResNetV2.__init__ / ResNetStage: the first block of every stage projects (DownsampleConv, preact=True -> no norm) and
    carries the stage stride (1 for stage 0, 2 after)."""
    stages, prev = [], cfg.stem_chs
    for si, (depth, c) in enumerate(zip(cfg.layers, cfg.channels)):
        blocks = []
        for bi in range(depth):
            blocks.append(dict(in_chs=prev, out_chs=c, mid_chs=make_divisible(c * cfg.bottle_ratio), stride=(1 if si == 0 else 2) if bi == 0 else 1,
                               downsample=bi == 0))
            prev = c
        stages.append(blocks)
    return stages


def bn_act(x, sd, p, cfg, training, stats=None):
    """BatchNormAct2d = nn.BatchNorm2d + ReLU.  training: batch statistics (biased variance) and, when `stats` (a dict of
    running buffers) is given, the momentum update with the unbiased variance."""
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    if training:
        rm = stats[p + ".running_mean"] if stats is not None else None
        rv = stats[p + ".running_var"] if stats is not None else None
        return _r(F.relu(F.batch_norm(x, rm, rv, w, b, True, cfg.momentum, cfg.eps)))
    return _r(F.relu(F.batch_norm(x, stats[p + ".running_mean"], stats[p + ".running_var"], w, b, False, cfg.momentum, cfg.eps)))


def gn_act(x, sd, p, cfg):
    """timm layers/norm_act.py GroupNormAct = nn.GroupNorm(32, C, eps 1e-5) + ReLU (per image: no batch statistics, no buffers)"""
    return _r(F.relu(F.group_norm(x, cfg.groups, sd[p + ".weight"], sd[p + ".bias"], cfg.eps)))


def _norm_act(x, sd, p, cfg, training, stats):
    return gn_act(x, sd, p, cfg) if getattr(cfg, "bit", False) else bn_act(x, sd, p, cfg, training, stats)


def _conv_weight(sd, key, cfg):
    """timm layers/std_conv.py StdConv2d.forward for the BiT towers: F.batch_norm over the weight viewed as [1, Cout, fan_in] in training
    mode without affine terms = (w - mean) / sqrt(biased var + eps) per output channel; the BatchNorm variants use the weight as it is"""
    w = sd[key]
    if not getattr(cfg, "bit", False):
        return _r(w)
    return _r(F.batch_norm(w.reshape(1, w.shape[0], -1), None, None, training=True, momentum=0.0, eps=cfg.std_eps).reshape_as(w))


def preact_bottleneck(x, sd, p, blk, cfg, training, stats):
    """timm resnetv2.py PreActBottleneck.forward"""
    pre = _norm_act(x, sd, p + ".norm1", cfg, training, stats)
    shortcut = x
    if blk["downsample"]:
        shortcut = _r(F.conv2d(pre, _conv_weight(sd, p + ".downsample.conv.weight", cfg), None, stride=blk["stride"]))
    out = _r(F.conv2d(pre, _conv_weight(sd, p + ".conv1.weight", cfg)))
    out = _r(F.conv2d(_norm_act(out, sd, p + ".norm2", cfg, training, stats), _conv_weight(sd, p + ".conv2.weight", cfg), None, stride=blk["stride"],
                      padding=1))
    out = F.conv2d(_norm_act(out, sd, p + ".norm3", cfg, training, stats), _conv_weight(sd, p + ".conv3.weight", cfg))
    return _r(out + shortcut)           # (the engine adds the shortcut in the GEMM epilogue: one rounding)


def resnetv2_forward_features(sd, p, cfg, images, training=True, stats=None):
    """ResNetV2.forward_features: stem, stages, final norm.  Stem (timm create_resnetv2_stem, preact=True -> no norm): 7x7/2 conv then
    MaxPool2d(3, 2, padding 1) for stem_type '' / ConstantPad2d(1, 0.) + MaxPool2d(3, 2, padding 0) for the BiT towers' 'fixed'."""
    # (_r: the storage-rounding mode of the text towers, `with rounding(torch.bfloat16)`, covers this tower as well -- identity otherwise)
    x = _r(F.conv2d(_r(images), _conv_weight(sd, p + ".stem.conv.weight", cfg), None, stride=2, padding=3))
    if getattr(cfg, "bit", False):
        x = F.max_pool2d(F.pad(x, (1, 1, 1, 1), value=0.0), 3, 2, 0)
    else:
        x = F.max_pool2d(x, 3, 2, 1)
    for si, blocks in enumerate(resnetv2_plan(cfg)):
        for bi, blk in enumerate(blocks):
            x = preact_bottleneck(x, sd, f"{p}.stages.{si}.blocks.{bi}", blk, cfg, training, stats)
    return _norm_act(x, sd, p + ".norm", cfg, training, stats)


def resnetv2_state_spec(cfg, prefix="img_encoder"):
    spec = [(prefix + ".stem.conv.weight", (cfg.stem_chs, 3, 7, 7))]

    def norm(name, c):
        spec.extend([(name + ".weight", (c,)), (name + ".bias", (c,))])
    for si, blocks in enumerate(resnetv2_plan(cfg)):
        for bi, b in enumerate(blocks):
            q = f"{prefix}.stages.{si}.blocks.{bi}"
            if b["downsample"]:
                spec.append((q + ".downsample.conv.weight", (b["out_chs"], b["in_chs"], 1, 1)))
            norm(q + ".norm1", b["in_chs"])
            spec.append((q + ".conv1.weight", (b["mid_chs"], b["in_chs"], 1, 1)))
            norm(q + ".norm2", b["mid_chs"])
            spec.append((q + ".conv2.weight", (b["mid_chs"], b["mid_chs"], 3, 3)))
            norm(q + ".norm3", b["mid_chs"])
            spec.append((q + ".conv3.weight", (b["out_chs"], b["mid_chs"], 1, 1)))
    norm(prefix + ".norm", cfg.channels[-1])
    return spec


def resnetv2_running_stats(cfg, prefix="img_encoder"):
    """fresh running buffers (mean 0, var 1) keyed like the state_dict"""
    st = {}
    for k, shape in resnetv2_state_spec(cfg, prefix):
        if ".norm" in k and k.endswith(".weight"):
            st[k[:-7] + ".running_mean"] = torch.zeros(shape)
            st[k[:-7] + ".running_var"] = torch.ones(shape)
    return st


def resnetv2_two_tower(sd, cfg, rcfg, images_1, images_2, labels=None, training=False, stats=None):
    """reference image.py:337-378 ResNetTwoTower.forward: the two towers are two separate forward calls (their BatchNorm
    statistics are per call), global average pool, flatten, pair head."""
    f1 = resnetv2_forward_features(sd, "img_encoder", rcfg, images_1, training, stats).mean((2, 3))
    f2 = resnetv2_forward_features(sd, "img_encoder", rcfg, images_2, training, stats).mean((2, 3))
    return image_two_tower(sd, cfg, f1, f2, labels, training)


# ---------------------------------------------------------------------------------------- optimiser step


def adamw_step(params, grads, m, v, step, lr, beta1=0.9, beta2=0.98, eps=1e-8, wd=1e-5, decay_mask=None):
    """torch.optim.AdamW update as the reference configures it (finetune_multimodal.py:296-308): in-place on
    lists of tensors; decay_mask[i] False for names containing "bias" / "LayerNorm.weight"."""
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    for i, (p, g) in enumerate(zip(params, grads)):
        if decay_mask is None or decay_mask[i]:
            p.mul_(1 - lr * wd)
        m[i].mul_(beta1).add_(g, alpha=1 - beta1)
        v[i].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        p.addcdiv_(m[i], (v[i].sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
