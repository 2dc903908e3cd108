"""Host-side mirror of the reference's src/models/base.py: heads, output struct, embeddings — same class
names, constructor arguments, forward signatures and state_dict keys, with the
arithmetic done by HIP kernels through item_alignment_amd.models.functional.

Modules such as nn.Linear / nn.LayerNorm / nn.Embedding appear here only as *parameter holders* (they give
the reference's key names); their own forward() is never called on the hot path.
"""
import ctypes as C
from collections import OrderedDict

import torch
from torch import nn

from .. import _lib
from ..arena import ParamArena
from .._lib import LayerCfg, LayerGrads, LayerWeights
from . import functional as Fn

ACT_NONE, ACT_TANH = 0, 1
# transposed bf16 shadows of the encoder layers' 2-D weights (+2 B / parameter of HBM, one batched transpose per optimiser step): the
# backward's data-gradient GEMMs run in the k-contiguous form; IA_TRANSPOSED_SHADOWS=0 keeps the k-strided form (no extra copy)
import os as _os
TRANSPOSED_SHADOWS = _os.environ.get("IA_TRANSPOSED_SHADOWS", "1") != "0"
# padded text towers: the attention backward skips query blocks that hold only masked positions when the model reads none of them
# (RobertaEncoder.forward(masked_rows_dead=True), set by RobertaModel under the same condition that allows IA_UNPAD); 0 = compute every row
MASKED_ROWS_DEAD = _os.environ.get("IA_MASKED_ROWS_DEAD", "1") != "0"


class SequenceClassifierOutput(OrderedDict):
    """reference base.py:160-186 (a transformers ModelOutput): attribute-, key- and index-accessible;
    fields loss, logits, probs, src_embeds, tgt_embeds (`logits` is passed at every reference call site)."""

    def __init__(self, loss=None, logits=None, probs=None, src_embeds=None, tgt_embeds=None, hidden_states=None):
        super().__init__()
        for k, v in (("loss", loss), ("logits", logits), ("probs", probs), ("src_embeds", src_embeds), ("tgt_embeds", tgt_embeds),
                     ("hidden_states", hidden_states)):
            if v is not None:
                self[k] = v

    def __getattr__(self, name):
        if name in ("loss", "logits", "probs", "src_embeds", "tgt_embeds", "hidden_states"):
            return self.get(name)
        raise AttributeError(name)

    def __getitem__(self, k):
        if isinstance(k, (int, slice)):
            return tuple(self.values())[k]
        return super().__getitem__(k)


class BaseModelOutput(OrderedDict):
    """last_hidden_state / hidden_states container returned by the *Model classes."""

    def __init__(self, last_hidden_state=None, pooler_output=None, hidden_states=None):
        super().__init__()
        self["last_hidden_state"] = last_hidden_state
        if pooler_output is not None:
            self["pooler_output"] = pooler_output
        if hidden_states is not None:
            self["hidden_states"] = hidden_states

    def __getattr__(self, name):
        if name in ("last_hidden_state", "pooler_output", "hidden_states"):
            return self.get(name)
        raise AttributeError(name)

    def __getitem__(self, k):
        if isinstance(k, (int, slice)):
            return tuple(self.values())[k]
        return super().__getitem__(k)


# ------------------------------------------------------------------------------------- arena plumbing
class HipModule(nn.Module):
    """Top-level models derive from this: parameters are re-homed into a flat arena (arena.py) the first
    time the model is used on the GPU; `.cuda()` keeps working as in the reference scripts."""

    def ensure_arena(self):
        arena = self.__dict__.get("_arena")
        if arena is None:
            p = next(self.parameters())
            if not p.is_cuda:
                raise _lib.ItemAlignError("model parameters are on the CPU: call model.cuda() first (the HIP engine has no CPU path)")
            _lib.load()
            arena = ParamArena(self, p.device)
            self.__dict__["_arena"] = arena
            for m in self.modules():
                m.__dict__["arena"] = arena
            if not hasattr(self, "_anchor"):
                self.__dict__["_anchor"] = torch.zeros(1, device=p.device, requires_grad=True)
            for m in self.modules():
                m.__dict__["anchor"] = self.__dict__["_anchor"]
        else:
            arena.sync_shadow()
        return arena

    @property
    def param_arena(self):
        return self.ensure_arena()

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        arena = self.__dict__.get("_arena")
        if arena is not None:
            arena.refresh_shadow()
        return out

    def _apply(self, fn, *a, **k):
        if self.__dict__.get("_arena") is not None:
            raise RuntimeError("model already lives in a GPU parameter arena; moving it again is not supported")
        return super()._apply(fn, *a, **k)


def init_bert_weights(module, std=0.02):
    """transformers PreTrainedModel._init_weights as used by the reference's RobertaPreTrainedModel classes."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, mean=0.0, std=std)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, mean=0.0, std=std)
            if m.padding_idx is not None:
                with torch.no_grad():
                    m.weight[m.padding_idx].zero_()
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


def create_position_ids_from_input_ids(input_ids, padding_idx, past_key_values_length=0):
    """reference base.py:189-202 (integer index arithmetic on [B, L])."""
    mask = input_ids.ne(padding_idx).int()
    incremental_indices = (torch.cumsum(mask, dim=1).type_as(mask) + past_key_values_length) * mask
    return incremental_indices.long() + padding_idx


# ------------------------------------------------------------------------------------------ embeddings
class RobertaEmbeddings(nn.Module):
    """reference base.py:205-296."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.padding_idx = config.pad_token_id
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size, padding_idx=self.padding_idx)
        self.eps = config.layer_norm_eps
        self.drop_p = config.hidden_dropout_prob
        self.word_pad = -1 if config.pad_token_id is None else config.pad_token_id
        self.pos_pad = self.word_pad
        self.stream_id = 1000

    def _ids(self, input_ids, token_type_ids, position_ids, position_source=None):
        if position_ids is None:
            position_ids = create_position_ids_from_input_ids(position_source if position_source is not None else input_ids, self.padding_idx)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        return input_ids.contiguous(), token_type_ids.contiguous(), position_ids.contiguous()

    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, cate_ids=None, inputs_embeds=None,
                past_key_values_length=0):
        if cate_ids is not None:
            raise AttributeError("'RobertaEmbeddings' object has no attribute 'cate_embeddings'")  # reference quirk A8 (base.py:216,274)
        if inputs_embeds is not None:
            raise NotImplementedError("inputs_embeds is not used by the reference train scripts for RobertaEmbeddings")
        ids, tts, pids = self._ids(input_ids, token_type_ids, position_ids)
        B, L = ids.shape
        p = self.drop_p if (self.training and torch.is_grad_enabled()) else 0.0
        y = Fn.EmbedLNFn.apply(self.anchor, self, ids, tts, pids, None, None, p, self.stream_id, None)
        return y.view(B, L, -1)


class RobertaImageEmbeddings(RobertaEmbeddings):
    """reference base.py:462-573: RobertaEmbeddings + image rows (img2txt of pre-extracted image embeddings)
    spliced in at token 1 (and at image_index for the one-tower layout)."""

    def __init__(self, config):
        super().__init__(config)
        self.config = config
        if config.ensemble == "begin":
            self.img2txt = nn.Linear(config.image_hidden_size, config.hidden_size, bias=True)

    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None, attention_mask=None,
                image_indices=None, past_key_values_length=0):
        ids, tts, pids = self._ids(input_ids, token_type_ids, position_ids, position_source=attention_mask)
        B, L = ids.shape
        extra = extra_idx = None
        if self.config.ensemble == "begin":
            dev = ids.device
            extra_idx = torch.full((B, L), -1, device=dev, dtype=torch.int32)
            ar = torch.arange(B, device=dev, dtype=torch.int32)
            if self.config.interaction_type == "one_tower":
                stacked = torch.stack(inputs_embeds, dim=1).reshape(2 * B, -1).to(torch.float32)
                extra = Fn.LinearSmallFn.apply(stacked, self.img2txt.weight, self.img2txt, ACT_NONE)
                extra_idx[:, 1] = 2 * ar
                extra_idx[ar.long(), image_indices.long().view(-1)] = 2 * ar + 1
            else:
                extra = Fn.LinearSmallFn.apply(inputs_embeds.to(torch.float32), self.img2txt.weight, self.img2txt, ACT_NONE)
                extra_idx[:, 1] = ar
            extra_idx = extra_idx.contiguous()
        p = self.drop_p if (self.training and torch.is_grad_enabled()) else 0.0
        y = Fn.EmbedLNFn.apply(self.anchor, self, ids, tts, pids, extra_idx, extra, p, self.stream_id, None)
        return y.view(B, L, -1)


class RobertaPKGMEmbeddings(RobertaEmbeddings):
    """reference base.py:299-459: text ids and KG ids share one input row; entity / relation embeddings
    become the triple-query (h + r) and relation-query (M h - r) rows of the sequence."""

    def __init__(self, config):
        super().__init__(config)
        self.config = config
        self.ent_emb = nn.Embedding(config.num_entities, config.kg_embedding_dim)
        self.rel_emb = nn.Embedding(config.num_relations, config.kg_embedding_dim)
        self.proj_mat = nn.Linear(config.kg_embedding_dim, config.kg_embedding_dim, bias=config.entity_projection_bias)
        if config.kg_embedding_dim != config.hidden_size:
            self.entity_embedding_projetor = nn.Linear(config.kg_embedding_dim, config.hidden_size)
            self.relation_embedding_projetor = nn.Linear(config.kg_embedding_dim, config.hidden_size)
            self.entity_projection_projetor = nn.Linear(config.kg_embedding_dim, config.hidden_size)
        else:
            self.entity_embedding_projetor = None
            self.relation_embedding_projetor = None
            self.entity_projection_projetor = None

    def kg_rows(self, input_ids):
        """[B, n_sides * 2P, H] fp32 KG rows (base.py:347-392): ia_kg_gather_fwd (entity sign + relation gather),
        the projections through the HIP small-linear kernel, ia_kg_rows_fwd (h + r | M h - r)."""
        cfg = self.config
        S, P = cfg.max_seq_len, cfg.max_pvs
        sides = [(S, S + 1)]
        if cfg.interaction_type == "one_tower":
            sides.append((2 * S + P + 1, 2 * S + P + 2))
        B = input_ids.shape[0]
        parts = []
        for ent_col, lo in sides:
            h, r = Fn.KGGatherFn.apply(self.anchor, self, input_ids, ent_col, lo, P)     # sign (quirk A1) [B, Dk], [B*P, Dk]
            hp = Fn.LinearSmallFn.apply(h, self.proj_mat.weight, self.proj_mat, ACT_NONE)
            if self.entity_embedding_projetor is not None:
                h = Fn.LinearSmallFn.apply(h, self.entity_embedding_projetor.weight, self.entity_embedding_projetor, ACT_NONE)
                r = Fn.LinearSmallFn.apply(r, self.relation_embedding_projetor.weight, self.relation_embedding_projetor, ACT_NONE)
                hp = Fn.LinearSmallFn.apply(hp, self.entity_projection_projetor.weight, self.entity_projection_projetor, ACT_NONE)
            parts += [h, r, hp]
        return Fn.KGRowsFn.apply(P, *parts)

    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None, past_key_values_length=0):
        cfg = self.config
        S, P = cfg.max_seq_len, cfg.max_pvs
        B = input_ids.shape[0]
        dev = input_ids.device
        one = cfg.interaction_type == "one_tower"
        Lm = (2 if one else 1) * (S + 2 * P)
        kg = self.kg_rows(input_ids)                                              # [B, sides*2P, H]
        extra = kg.reshape(-1, kg.shape[-1]).contiguous()
        ids = torch.zeros((B, Lm), device=dev, dtype=torch.long)
        extra_idx = torch.full((B, Lm), -1, device=dev, dtype=torch.int32)
        base = (torch.arange(B, device=dev, dtype=torch.int32) * kg.shape[1]).unsqueeze(1)
        kcols = torch.arange(2 * P, device=dev, dtype=torch.int32).unsqueeze(0)
        ids[:, :S] = input_ids[:, :S]
        extra_idx[:, S:S + 2 * P] = base + kcols
        if one:
            ids[:, S + 2 * P:2 * S + 2 * P] = input_ids[:, S + P + 1:2 * S + P + 1]
            extra_idx[:, 2 * S + 2 * P:] = base + 2 * P + kcols
        if position_ids is None:
            position_ids = create_position_ids_from_input_ids(input_ids, self.padding_idx)
        if token_type_ids is None:
            token_type_ids = torch.zeros((B, Lm), device=dev, dtype=torch.long)
        p = self.drop_p if (self.training and torch.is_grad_enabled()) else 0.0
        y = Fn.EmbedLNFn.apply(self.anchor, self, ids.contiguous(), token_type_ids.contiguous(), position_ids.contiguous(),
                               extra_idx.contiguous(), extra, p, self.stream_id, None)
        return y.view(B, Lm, -1)


# --------------------------------------------------------------------------------------- encoder stack
class _SelfAttention(nn.Module):
    def __init__(self, H):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)


class _SelfOutput(nn.Module):
    def __init__(self, din, H, eps):
        super().__init__()
        self.dense = nn.Linear(din, H)
        self.LayerNorm = nn.LayerNorm(H, eps=eps)


class _Attention(nn.Module):
    def __init__(self, H, eps):
        super().__init__()
        self.self = _SelfAttention(H)
        self.output = _SelfOutput(H, H, eps)


class _Intermediate(nn.Module):
    def __init__(self, H, I):
        super().__init__()
        self.dense = nn.Linear(H, I)


class RobertaLayer(nn.Module):
    """Parameter layout of transformers RobertaLayer (state_dict keys of SURVEY.md Appendix D)."""

    def __init__(self, config):
        super().__init__()
        H, I, eps = config.hidden_size, config.intermediate_size, config.layer_norm_eps
        self.attention = _Attention(H, eps)
        self.intermediate = _Intermediate(H, I)
        self.output = _SelfOutput(I, H, eps)

    def arena_groups(self):
        s = self.attention.self
        return [(s.query.weight, s.key.weight, s.value.weight), (s.query.bias, s.key.bias, s.value.bias)]


class _EngineStack:
    """Shared by the RoBERTa encoder and the ViT blocks: ctypes structs for ia_layer_fwd/bwd."""

    def _build(self):
        if self.__dict__.get("_w") is not None:
            return
        A = self.arena
        ws, gs = [], []
        for i in range(len(self.layers)):
            d = self.layer_tensors(i)
            w, g = LayerWeights(), LayerGrads()
            qkv_w, qkv_b = d["qkv_w"], d["qkv_b"]
            w.w_qkv = (A.fused_shadow(qkv_w) if isinstance(qkv_w, (tuple, list)) else A.shadow_of(qkv_w)).data_ptr()
            g.w_qkv = (A.fused_master(qkv_w, grad=True) if isinstance(qkv_w, (tuple, list)) else qkv_w.grad).data_ptr()
            if isinstance(qkv_b, (tuple, list)):
                w.b_qkv, g.b_qkv = A.fused_master(qkv_b).data_ptr(), A.fused_master(qkv_b, grad=True).data_ptr()
            else:
                w.b_qkv, g.b_qkv = qkv_b.data_ptr(), qkv_b.grad.data_ptr()
            for name in ("w_o", "w_fc1", "w_fc2"):
                setattr(w, name, A.shadow_of(d[name]).data_ptr())
                setattr(g, name, d[name].grad.data_ptr())
            if TRANSPOSED_SHADOWS:          # W^T copies for the data-gradient GEMMs (arena.register_transposed)
                w.wt_qkv = A.register_transposed(qkv_w).data_ptr()
                for name in ("w_o", "w_fc1", "w_fc2"):
                    setattr(w, "wt" + name[1:], A.register_transposed(d[name]).data_ptr())
            for name in ("b_o", "ln1_g", "ln1_b", "b_fc1", "b_fc2", "ln2_g", "ln2_b"):
                setattr(w, name, d[name].data_ptr())
                setattr(g, name, d[name].grad.data_ptr())
            ws.append(w); gs.append(g)
        self.__dict__["_w"], self.__dict__["_g"] = ws, gs
        A.refresh_transposed()

    def weights(self, i):
        self._build()
        return self.__dict__["_w"][i]

    def grads(self, i):
        self._build()
        return self.__dict__["_g"][i]

    def layer_params(self, i):
        return list(self.layers[i].parameters())

    def layer_cfg(self, i, B, L, training, seed):
        c = LayerCfg()
        c.B, c.L, c.H, c.I, c.nh = B, L, self.hidden_size, self.intermediate_size, self.num_heads
        c.pre_ln = int(self.pre_ln)
        c.eps = self.eps
        c.hidden_drop = self.hidden_drop if training else 0.0
        c.attn_drop = self.attn_drop if training else 0.0
        c.seed = seed
        c.layer_id = self.layer_id_base + i
        c.masked_rows_dead = int(self.__dict__.get("_masked_rows_dead", False))
        return c


class RobertaEncoder(nn.Module, _EngineStack):
    """Stack of post-LN layers; replaces transformers RobertaEncoder at reference text.py:150,1108,
    multimodal.py:51.  forward returns the tuple of all hidden states (embedding output first)."""

    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([RobertaLayer(config) for _ in range(config.num_hidden_layers)])
        self.hidden_size, self.intermediate_size = config.hidden_size, config.intermediate_size
        self.num_heads = config.num_attention_heads
        if config.hidden_size != 64 * config.num_attention_heads:
            raise ValueError("the fused attention kernel is specialised for head_dim 64 (every config of the reference, SURVEY App. C)")
        if getattr(config, "hidden_act", "gelu") != "gelu":
            raise ValueError("only hidden_act='gelu' (erf) is implemented, as in every reference config")
        self.eps = config.layer_norm_eps
        self.hidden_drop, self.attn_drop = config.hidden_dropout_prob, config.attention_probs_dropout_prob
        self.pre_ln = False
        self.layer_id_base = 0

    @property
    def layers(self):
        return self.layer

    def layer_tensors(self, i):
        l = self.layer[i]
        s = l.attention.self
        return dict(qkv_w=(s.query.weight, s.key.weight, s.value.weight), qkv_b=(s.query.bias, s.key.bias, s.value.bias),
                    w_o=l.attention.output.dense.weight, b_o=l.attention.output.dense.bias,
                    ln1_g=l.attention.output.LayerNorm.weight, ln1_b=l.attention.output.LayerNorm.bias,
                    w_fc1=l.intermediate.dense.weight, b_fc1=l.intermediate.dense.bias,
                    w_fc2=l.output.dense.weight, b_fc2=l.output.dense.bias,
                    ln2_g=l.output.LayerNorm.weight, ln2_b=l.output.LayerNorm.bias)

    def forward(self, hidden_states, attention_mask=None, masked_rows_dead=False):
        """masked_rows_dead: the caller reads no hidden state of a masked position (heads on [CLS] / valid spans) -- the gradient arriving
        at such rows is then exactly zero in every layer and the attention backward may skip query blocks made of them (ia_layer_cfg)."""
        B, L, H = hidden_states.shape
        km = None
        if attention_mask is not None:
            km = (attention_mask != 0).to(torch.uint8).contiguous()
        self.__dict__["_masked_rows_dead"] = bool(masked_rows_dead and km is not None and MASKED_ROWS_DEAD)
        outs = Fn.EncoderStackFn.apply(hidden_states.reshape(B * L, H), self.anchor, self, km, B, L, torch.is_grad_enabled(), None)
        return (hidden_states,) + tuple(o.view(B, L, H) for o in outs)


class RobertaPooler(nn.Module):
    """Parameter holder for checkpoint interchange (coca.text_encoder.pooler.dense.*): the pair
    step never uses the pooled output, so these weights stay frozen exactly as they get no gradient in the reference."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        for p in self.parameters():
            p.requires_grad = False


# ---------------------------------------------------------------------------------------------- heads
def cls_rows(B, L, index, device):
    """row numbers of token `index` of every sequence in the flattened [B*L, H] hidden state."""
    return (torch.arange(B, device=device, dtype=torch.int32) * L + index).contiguous()


class TwoTowerClassificationHead(nn.Module):
    """reference base.py:91-117: dropout both vectors, Linear(cat) -> softmax.  `forward` takes fp32 [B, H]
    features (already dropped-out by the caller's GatherRowsFn when they come from hidden states)."""

    def __init__(self, hidden_size, dropout=0.0, num_labels=2):
        super().__init__()
        self.drop_p = dropout
        self.out_proj = nn.Linear(hidden_size * 2, num_labels)

    def forward(self, features_1, features_2, labels=None, differentiable_logits=False):
        if differentiable_logits:      # hinge / bce / euclidean losses act on the logits themselves
            logits = Fn.LinearSmallFn.apply(torch.cat((features_1, features_2), dim=1), self.out_proj.weight, self.out_proj, ACT_NONE)
            return features_1, features_2, logits, torch.softmax(logits, dim=1), None
        logits, probs, loss = Fn.PairHeadCEFn.apply(features_1, features_2, self.out_proj, labels)
        return features_1, features_2, logits, probs, (loss if labels is not None else None)


class RobertaClassificationHead(nn.Module):
    """reference base.py:120-157."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        cls_layers = [int(i) for i in config.cls_layers.split(",")]
        length = 1 if config.cls_pool == "avg" else len(cls_layers)
        self.dense = nn.Linear(config.hidden_size * length, config.hidden_size)
        self.drop_p = config.classifier_dropout if getattr(config, "classifier_dropout", None) is not None else config.hidden_dropout_prob
        if getattr(config, "ensemble", None) == "end":
            self.dense_img = nn.Linear(2 * config.image_hidden_size, config.hidden_size)
            self.out_proj = nn.Linear(2 * config.hidden_size, config.num_labels)
        else:
            self.out_proj = nn.Linear(config.hidden_size, config.num_labels)

    def forward(self, cls_features, labels=None, inputs_embeds=None, differentiable_logits=False):
        """cls_features: fp32 [B, k*H], dropout already applied (base.py:140-141)."""
        x = Fn.LinearSmallFn.apply(cls_features, self.dense.weight, self.dense, ACT_TANH)
        x = _dropout_small(x, self.drop_p, self.training)
        y = None
        if getattr(self.config, "ensemble", None) == "end":
            y = torch.cat(inputs_embeds, dim=-1).to(torch.float32)
            y = _dropout_small(y, self.drop_p, self.training)
            y = Fn.LinearSmallFn.apply(y, self.dense_img.weight, self.dense_img, ACT_TANH)
            y = _dropout_small(y, self.drop_p, self.training)
        if differentiable_logits:
            logits = Fn.LinearSmallFn.apply(x if y is None else torch.cat((x, y), dim=-1), self.out_proj.weight, self.out_proj, ACT_NONE)
            return logits, torch.softmax(logits, dim=1), None
        logits, probs, loss = Fn.PairHeadCEFn.apply(x, y, self.out_proj, labels)
        return logits, probs, (loss if labels is not None else None)


def _dropout_small(x, p, training):
    """dropout on a [B, H] fp32 head tensor (a few KB): plain torch elementwise."""
    if training and p > 0 and torch.is_grad_enabled():
        return torch.nn.functional.dropout(x, p, True)
    return x


class InnerProduct(nn.Module):
    """reference base.py:10-34."""

    def __init__(self, normalize=False):
        super().__init__()
        self.normalize = normalize

    def forward(self, x1, x2):
        if self.normalize:
            x1 = nn.functional.normalize(x1, p=2, dim=1)
            x2 = nn.functional.normalize(x2, p=2, dim=1)
        return (x1 * x2).sum(-1)


class VecSimClassificationHead(nn.Module):
    """reference base.py:37-88.  dense+tanh on the HIP small-linear kernel, similarity + probability on ia_pair_sim_*."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        cls_layers = [int(i) for i in config.cls_layers.split(",")]
        length = 1 if config.cls_pool == "avg" else len(cls_layers)
        self.dense = nn.Linear(config.hidden_size * length, config.hidden_size)
        self.drop_p = config.classifier_dropout if getattr(config, "classifier_dropout", None) is not None else config.hidden_dropout_prob
        if config.similarity_measure not in ("inner_product", "cosine", "l1", "l2"):
            raise ValueError(f"Unsupported similarty measure: {config.similarity_measure}")

    def forward(self, features_1, features_2):
        x = _dropout_small(Fn.LinearSmallFn.apply(features_1, self.dense.weight, self.dense, ACT_TANH), self.drop_p, self.training)
        y = _dropout_small(Fn.LinearSmallFn.apply(features_2, self.dense.weight, self.dense, ACT_TANH), self.drop_p, self.training)
        sim, probs = Fn.PairSimFn.apply(x, y, Fn.SIM_MEASURES[self.config.similarity_measure])
        return x, y, sim, probs
