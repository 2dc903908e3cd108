"""Host-side mirror of the reference's src/models/multimodal.py: RobertaImage{Model,OneTower,TwoTower}
(RoBERTa + pre-extracted image embeddings) and CoCaForItemAlignment (RoBERTa text tower + ViT image tower).
"""
import os

import torch
from torch import nn

from . import functional as Fn
from .base import (ACT_NONE, BaseModelOutput, HipModule, RobertaEmbeddings, RobertaEncoder, RobertaImageEmbeddings, RobertaPooler,
                   SequenceClassifierOutput, TwoTowerClassificationHead, VecSimClassificationHead, cls_rows, init_bert_weights)
from .loss import apply_loss, make_loss
from .text import PretrainedMixin, RobertaOneTower, RobertaTwoTower, adopt


TOWER_STREAMS = os.environ.get("IA_TOWER_STREAMS", "0") == "1"      # module switch (bench.py flips it for its variant line)


class RobertaImageModel(HipModule, PretrainedMixin):
    """reference multimodal.py:23-210."""

    def __init__(self, config, add_pooling_layer=True):
        super().__init__()
        self.config = config
        self.embeddings = RobertaImageEmbeddings(config) if config.ensemble == "begin" else RobertaEmbeddings(config)
        self.encoder = RobertaEncoder(config)
        self.pooler = RobertaPooler(config) if add_pooling_layer else None
        init_bert_weights(self, getattr(config, "initializer_range", 0.02))

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None, inputs_embeds=None,
                image_indices=None, **unused):
        (self._root if "_root" in self.__dict__ else self).ensure_arena()
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if self.config.ensemble == "begin":
            e = self.embeddings(input_ids=input_ids, position_ids=position_ids, token_type_ids=token_type_ids, inputs_embeds=inputs_embeds,
                                attention_mask=attention_mask, image_indices=image_indices)
        else:
            e = self.embeddings(input_ids=input_ids, position_ids=position_ids, token_type_ids=token_type_ids)
        hs = self.encoder(e, attention_mask)
        return BaseModelOutput(last_hidden_state=hs[-1], hidden_states=hs)


class RobertaImageOneTower(RobertaOneTower):
    """reference multimodal.py:213-320."""
    supports_auxiliary = False

    def _make_backbone(self, config):
        return RobertaImageModel(config, add_pooling_layer=False)

    def _backbone(self, input_ids, attention_mask, token_type_ids, position_ids, cate_ids, inputs_embeds, image_indices):
        return self.roberta(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids,
                            inputs_embeds=inputs_embeds, image_indices=image_indices)


class RobertaImageTwoTower(RobertaTwoTower):
    """reference multimodal.py:323-461 (keyword order: ..., position_ids_1, images_1, ..., images_2, head_mask, labels)."""

    def _make_backbone(self, config):
        return RobertaImageModel(config, add_pooling_layer=False)

    def _backbone(self, ids, mask, tts, pids, images_1=None, images_2=None):
        imgs = torch.cat((images_1, images_2), dim=0) if images_1 is not None else None
        return self.roberta(ids, attention_mask=mask, token_type_ids=tts, position_ids=pids, inputs_embeds=imgs).last_hidden_state

    def forward(self, input_ids_1=None, attention_mask_1=None, token_type_ids_1=None, position_ids_1=None, images_1=None,
                input_ids_2=None, attention_mask_2=None, token_type_ids_2=None, position_ids_2=None, images_2=None, head_mask=None,
                labels=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        return super().forward(input_ids_1=input_ids_1, attention_mask_1=attention_mask_1, token_type_ids_1=token_type_ids_1,
                               position_ids_1=position_ids_1, input_ids_2=input_ids_2, attention_mask_2=attention_mask_2,
                               token_type_ids_2=token_type_ids_2, position_ids_2=position_ids_2, labels=labels, images_1=images_1,
                               images_2=images_2)


class CoCaModel(nn.Module):
    """reference multimodal.py:709-840: holder of the two encoders (state_dict prefix coca.{img,text}_encoder)."""

    def __init__(self, config, image_encoder=None, text_encoder=None):
        super().__init__()
        self.config = config
        self.img_encoder = image_encoder
        self.text_encoder = text_encoder

    def embed_text(self, input_ids, attention_mask, token_type_ids, position_ids, padded_rows_matter=False):
        # padded_rows_matter: the caller reads the hidden states of PADDED positions (the cross_attn multimodal layers attend
        # over all L text tokens without a padding mask, reference multimodal.py:529-616), so the unpadded tower run (IA_UNPAD),
        # which leaves zeros there, must not be used
        out = self.text_encoder(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids,
                                allow_unpad=not padded_rows_matter)
        return out.last_hidden_state

    def embed_image(self, images):
        return self.img_encoder.forward_features(images)


class LayerNorm(nn.Module):
    """reference multimodal.py:475-482: LayerNorm with a learned gamma and a constant zero `beta` buffer."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))

    def forward(self, x2d):
        return Fn.GammaLayerNormFn.apply(x2d, self.gamma, self.beta, 1e-5)


class RotaryEmbedding(nn.Module):
    """reference multimodal.py:495-506.  Only the `inv_freq` buffer lives here (state_dict parity); the rotation itself
    is computed inside ia_rotary_split_fwd from the same closed form."""

    def __init__(self, dim):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim)))


class SwiGLU(nn.Module):
    """reference multimodal.py:521-524 (parameter-free; keeps the Sequential indices `ff_out.1` / `ff.0`, `ff.2`)."""

    def forward(self, x2d):
        return Fn.SwiGLUFn.apply(x2d)


class Residual(nn.Module):
    """reference multimodal.py:486-492.  The `+ x` is folded into the wrapped block's last GEMM epilogue."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x2d, *args, **kwargs):
        return self.fn(x2d, *args, residual=x2d, **kwargs)


def _check_head_dim(dim_head):
    if dim_head != 64:
        raise NotImplementedError(f"the fused attention kernels are built for head dim 64 (got {dim_head}); "
                                  "coca_base (768/12) and coca_large (1024/16) both give 64")


class ParallelTransformerBlock(nn.Module):
    """reference multimodal.py:529-626 (is_decoding False, no attention mask — CoCaForItemAlignment passes none, :1007):
    LN -> one fused projection -> multi-query attention with rotary q/k  ||  SwiGLU feed-forward -> attn_out + ff_out."""

    def __init__(self, dim, dim_head=64, heads=8, ff_mult=4, is_decoding=False):
        super().__init__()
        _check_head_dim(dim_head)
        if is_decoding:
            raise NotImplementedError("causal decoding blocks are not on the item-alignment path")
        self.norm = LayerNorm(dim)
        attn_inner_dim, ff_inner_dim = dim_head * heads, dim * ff_mult
        self.fused_dims = (attn_inner_dim, dim_head, dim_head, ff_inner_dim * 2)
        self.heads, self.scale, self.ff_inner_dim = heads, dim_head ** -0.5, ff_inner_dim
        self.rotary_emb = RotaryEmbedding(dim_head)
        self.fused_attn_ff_proj = nn.Linear(dim, sum(self.fused_dims), bias=False)
        self.attn_out = nn.Linear(attn_inner_dim, dim, bias=False)
        self.ff_out = nn.Sequential(SwiGLU(), nn.Linear(ff_inner_dim, dim, bias=False))

    def forward(self, x2d, B, n, residual=None):
        h = self.heads
        xn = self.norm(x2d)
        fused = Fn.LinearBf16Fn.apply(xn, None, self.fused_attn_ff_proj.weight, self)
        q, kv, s = Fn.FusedSplitFn.apply(fused, n, h, self.ff_inner_dim)
        # multi-query attention: every query head attends to the single k/v head -> fold the heads into query rows
        o = Fn.AttentionXFn.apply(q.view(B * n * h, 64), kv, B, 1, n * h, n, self.scale)
        y = Fn.LinearBf16Fn.apply(o.view(B * n, h * 64), residual, self.attn_out.weight, self)
        # (s is not kept for ff_out's weight gradient: backward recomputes it from the x | gate columns of `fused`, which FusedSplitFn holds)
        return Fn.LinearBf16Fn.apply(s, y, self.ff_out[1].weight, self, (fused, h * 64 + 128, self.ff_inner_dim))


class CrossAttention(nn.Module):
    """reference multimodal.py:630-706 (parallel_ff=True, norm_context=False as CoCaForItemAlignment builds it, :957-960):
    queries from the text tokens, one key/value head from the image tokens, plus a parallel SwiGLU feed-forward."""

    def __init__(self, dim, *, context_dim=None, dim_head=64, heads=8, parallel_ff=False, ff_mult=4, norm_context=False):
        super().__init__()
        _check_head_dim(dim_head)
        self.heads, self.scale = heads, dim_head ** -0.5
        inner_dim = heads * dim_head
        context_dim = dim if context_dim is None else context_dim
        self.context_dim = context_dim
        self.norm = LayerNorm(dim)
        self.context_norm = LayerNorm(context_dim) if norm_context else nn.Identity()
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(context_dim, dim_head * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)
        ff_inner_dim = ff_mult * dim
        self.ff = nn.Sequential(nn.Linear(dim, ff_inner_dim * 2, bias=False), SwiGLU(), nn.Linear(ff_inner_dim, dim, bias=False)) \
            if parallel_ff else None

    def forward(self, x2d, ctx2d, B, n, n_ctx, residual=None):
        h = self.heads
        if ctx2d.shape[-1] != self.context_dim:
            raise ValueError(f"CrossAttention: context width {ctx2d.shape[-1]} != to_kv in_features {self.context_dim} "
                             "(the reference raises the same shape error, SURVEY note N1)")
        xn = self.norm(x2d)
        ctx2d = self.context_norm(ctx2d)
        q = Fn.LinearBf16Fn.apply(xn, None, self.to_q.weight, self)
        kv = Fn.LinearBf16Fn.apply(ctx2d, None, self.to_kv.weight, self)
        o = Fn.AttentionXFn.apply(q.view(B * n * h, 64), kv, B, 1, n * h, n_ctx, self.scale)
        y = Fn.LinearBf16Fn.apply(o.view(B * n, h * 64), residual, self.to_out.weight, self)
        if self.ff is not None:
            f = Fn.LinearBf16Fn.apply(xn, None, self.ff[0].weight, self)
            y = Fn.LinearBf16Fn.apply(self.ff[1](f), y, self.ff[2].weight, self, (f, 0, f.shape[1] // 2))
        return y


class CoCaForItemAlignment(HipModule):
    """reference multimodal.py:936-1045.  ensemble == "sum": text CLS + image CLS -> two-tower head.

    Note N1: the headline pairing roberta_large (1024-d) + ViT-B/16 (768-d) raises a shape error
    in the reference (multimodal.py:1015 adds a 1024-d and a 768-d vector).  When the two widths differ this
    class inserts one Linear(image_dim -> hidden_size) on the image CLS (`img_proj`, a documented deviation
    needed to run the named benchmark config at all); with equal widths (coca_base + ViT-B, coca_large + ViT-L,
    the reference's working pairings) the module does not exist and the arithmetic is the reference's."""

    def __init__(self, config, image_encoder=None, text_encoder=None):
        super().__init__()
        self.config = config
        self.ensemble = config.ensemble
        self.num_labels = config.num_labels
        self.coca = CoCaModel(config, image_encoder, text_encoder)
        img_dim = getattr(image_encoder, "num_features", config.hidden_size)
        if config.ensemble == "cross_attn":
            # reference multimodal.py:946-961
            self.multimodal_layers = nn.ModuleList([])
            dh = config.hidden_size // config.num_attention_heads_multimodal
            for _ in range(config.num_hidden_layers_multimodal):
                self.multimodal_layers.append(nn.ModuleList([
                    Residual(ParallelTransformerBlock(dim=config.hidden_size, dim_head=dh, heads=config.num_attention_heads_multimodal,
                                                      ff_mult=config.feedforward_multiplication_multimodal, is_decoding=False)),
                    Residual(CrossAttention(dim=config.hidden_size, dim_head=dh, heads=config.num_attention_heads_multimodal,
                                            parallel_ff=True, ff_mult=config.feedforward_multiplication_multimodal))]))
        if img_dim != config.hidden_size and config.ensemble != "cross_attn":
            self.img_proj = nn.Linear(img_dim, config.hidden_size)
            nn.init.normal_(self.img_proj.weight, std=getattr(config, "initializer_range", 0.02))
            nn.init.zeros_(self.img_proj.bias)
        else:
            self.img_proj = None
        if config.classification_method == "vec_sim":
            self.classifier = VecSimClassificationHead(config)
        else:
            self.classifier = TwoTowerClassificationHead(config.hidden_size, dropout=config.hidden_dropout_prob, num_labels=config.num_labels)
            nn.init.normal_(self.classifier.out_proj.weight, std=getattr(config, "initializer_range", 0.02))
            nn.init.zeros_(self.classifier.out_proj.bias)
        self.loss_fct = make_loss(config)
        adopt(self, image_encoder, text_encoder)

    def forward(self, input_ids_1, attention_mask_1, token_type_ids_1, position_ids_1, images_1, input_ids_2, attention_mask_2,
                token_type_ids_2, position_ids_2, images_2, labels=None):
        self.ensure_arena()
        B, L = input_ids_1.shape
        if self.ensemble == "cross_attn":
            return self._forward_cross_attn(input_ids_1, attention_mask_1, token_type_ids_1, position_ids_1, images_1, labels)
        cat = lambda a, b: None if a is None else torch.cat((a, b), dim=0)
        # both items of a pair go through each shared-weight encoder as one 2B batch
        # The two towers are independent until the head.  IA_TOWER_STREAMS=1 runs the image tower on a second HIP stream so
        # that one tower's kernel tails (last partial round of tiles, epilogue drains, HBM-bound LayerNorm tails) fill with
        # the other's work: +3.5 % pairs/s on the bench step (autograd replays each tower's backward on the stream its
        # forward used).  Off by default: concurrent kernels stretch each other's wall time, so per-kernel timings (the
        # bench's roofline leg, rocprofv3 averages) stop describing the kernels themselves.
        two_streams = TOWER_STREAMS and images_1.is_cuda
        images = (images_1, images_2)               # run as one 2B batch; the patch gather reads the two tensors in turn (no fp32 concat)

        def image_tower():
            img_tok = self.coca.embed_image(images)                                                                  # [2B, N, Hi] bf16
            i_cls = self.coca.img_encoder.forward_head(img_tok, pre_logits=True)                                   # [2B, Hi] fp32
            if self.img_proj is not None:
                i_cls = Fn.LinearSmallFn.apply(i_cls, self.img_proj.weight, self.img_proj, ACT_NONE)
            return i_cls
        if two_streams:
            main = torch.cuda.current_stream()
            side = self.__dict__.get("_side_stream")
            if side is None:
                side = self.__dict__["_side_stream"] = torch.cuda.Stream()
                self.param_arena.side_streams.append(side)
            side.wait_stream(main)
            for im in images:
                im.record_stream(side)
            with torch.cuda.stream(side):
                i_cls = image_tower()
        else:
            i_cls = image_tower()
        txt = self.coca.embed_text(cat(input_ids_1, input_ids_2), cat(attention_mask_1, attention_mask_2),
                                   cat(token_type_ids_1, token_type_ids_2), cat(position_ids_1, position_ids_2))   # [2B, L, H]
        H = txt.shape[-1]
        dev = txt.device
        t_cls = Fn.GatherRowsFn.apply(txt.reshape(2 * B * L, H), self.anchor, cls_rows(2 * B, L, 0, dev), 0.0, 0)   # text_tokens[:, 0]
        if two_streams:
            main.wait_stream(side)
            i_cls.record_stream(main)
        emb = t_cls + i_cls                                                                                          # multimodal.py:1015
        e1, e2 = emb[:B].contiguous(), emb[B:].contiguous()
        return self._finish(e1, e2, labels)

    def _forward_cross_attn(self, input_ids, attention_mask, token_type_ids, position_ids, images, labels):
        """reference multimodal.py:1003-1013.  Quirk A4 (:1013): `embeds_2 = text_tokens_1[:, 0]` — the target embedding is
        the SOURCE item's token, so nothing computed from item 2 reaches the outputs or any gradient; item 2's encoders
        and multimodal layers are therefore not run at all (same outputs, same gradients, half the work)."""
        B, L = input_ids.shape
        img_tok = self.coca.embed_image(images)                                                     # [B, N, Hi] bf16
        txt = self.coca.embed_text(input_ids, attention_mask, token_type_ids, position_ids, padded_rows_matter=True)   # [B, L, H] bf16
        H, N = txt.shape[-1], img_tok.shape[1]
        x = txt.reshape(B * L, H)
        ctx = img_tok.reshape(B * N, img_tok.shape[-1])
        for attn_ff, cross_attn in self.multimodal_layers:
            x = attn_ff(x, B, L)
            x = cross_attn(x, ctx, B, L, N)
        e1 = Fn.GatherRowsFn.apply(x, self.anchor, cls_rows(B, L, 0, x.device), 0.0, 0)            # text_tokens_1[:, 0]
        return self._finish(e1, e1, labels)

    def _finish(self, e1, e2, labels):
        training = self.training and torch.is_grad_enabled()
        if self.config.classification_method == "vec_sim":
            p = self.classifier.drop_p if training else 0.0
            if p > 0:
                e1, e2 = nn.functional.dropout(e1, p, True), nn.functional.dropout(e2, p, True)
            src, tgt, logits, probs2 = self.classifier(e1, e2)
            loss = None
        else:
            p = self.classifier.drop_p if training else 0.0
            if p > 0:
                e1, e2 = nn.functional.dropout(e1, p, True), nn.functional.dropout(e2, p, True)
            ce = self.config.loss_type == "ce"
            src, tgt, logits, probs2, loss = self.classifier(e1, e2, labels if ce else None,
                                                             differentiable_logits=(labels is not None and not ce))
        src, tgt, probs = probs2[:, 0], probs2[:, 1], probs2[:, 1]
        if labels is not None and self.config.loss_type != "ce":
            loss = apply_loss(self.loss_fct, self.config, logits, labels, src, tgt)
        return SequenceClassifierOutput(loss=loss, probs=probs, logits=logits, src_embeds=src, tgt_embeds=tgt)
