"""Host-side mirror of the reference's src/models/text.py: RobertaModel, Roberta{One,Two}Tower,
PKGM{One,Two}Tower, TextCNNTwoTower — same constructor / forward signatures, outputs and state_dict keys,
encoder arithmetic on the HIP engine.  Two-tower models run both towers as ONE 2B batch (the reference runs
the same weights twice sequentially, text.py:1325-1351; the maths per sample is identical).
"""
import os

import torch
from torch import nn
import torch.nn.functional as F

from ..utils import logger, ROBERTA_WEIGHTS_NAME, KG_WEIGHTS_NAME
from . import functional as Fn
from .base import (BaseModelOutput, HipModule, RobertaClassificationHead, RobertaEmbeddings, RobertaEncoder, RobertaPKGMEmbeddings,
                   RobertaPooler, SequenceClassifierOutput, TwoTowerClassificationHead, VecSimClassificationHead, cls_rows,
                   init_bert_weights)
from .loss import make_loss, apply_loss


class PretrainedMixin:
    """`from_pretrained(path, config=cfg, ignore_mismatched_sizes=True)` of the reference classes
    (RobertaPreTrainedModel.from_pretrained; PKGM variant merges two files, reference text.py:620-654)."""

    weight_files = (ROBERTA_WEIGHTS_NAME,)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, ignore_mismatched_sizes=False, state_dict=None,
                        **kwargs):
        if config is None:
            raise ValueError("config= is required (the reference always passes a BertConfig)")
        model = cls(config, *model_args)
        merged = {}
        if state_dict is not None:
            merged.update(state_dict)
        elif pretrained_model_name_or_path is not None:
            for fn in cls.weight_files:
                f = os.path.join(str(pretrained_model_name_or_path), fn)
                if os.path.exists(f):
                    merged.update(torch.load(f, map_location="cpu"))
                else:
                    logger.warning(f"{f} not found: those weights keep their random initialisation")
        if merged:
            own = model.state_dict()
            loaded, skipped, unmatched = {}, [], []
            for k, v in merged.items():
                # checkpoint keys may carry a base-model prefix the target lacks (a bare RobertaModel as CoCa's text tower loading
                # a "roberta." / "bert." checkpoint: HF strips base_model_prefix) or lack one the target has
                bare = k.split(".", 1)[1] if k.startswith(("roberta.", "bert.")) else k
                cands = [k, "roberta." + k, k.replace("bert.", "roberta.", 1), "roberta.embeddings." + k, bare]
                tgt = next((c for c in cands if c in own), None)
                if tgt is None:
                    unmatched.append(k)
                    continue
                if own[tgt].shape != v.shape:
                    if not ignore_mismatched_sizes:
                        raise RuntimeError(f"size mismatch for {tgt}: {tuple(v.shape)} vs {tuple(own[tgt].shape)}")
                    skipped.append(tgt)
                    continue
                loaded[tgt] = v
            if not loaded:
                raise RuntimeError(f"no checkpoint key matches {cls.__name__} (first checkpoint keys: {list(merged)[:5]}, first model keys: "
                                   f"{list(own)[:5]}): the model would silently keep its random initialisation")
            model.load_state_dict(loaded, strict=False)
            if skipped:
                logger.warning(f"ignored mismatched sizes: {skipped}")
            if unmatched:
                logger.warning(f"{len(unmatched)} checkpoint keys have no counterpart in {cls.__name__} (first: {unmatched[:8]})")
            missing = [k for k in own if k not in loaded]
            if missing:
                logger.warning(f"{len(missing)} model tensors keep their random initialisation (first: {missing[:8]})")
        model.eval()
        return model


def _key_mask(attention_mask):
    return attention_mask


# IA_UNPAD=1: text towers run on the valid tokens only (RobertaModel._forward_unpadded); default off = the reference's dense padding
UNPAD = os.environ.get("IA_UNPAD", "0") == "1"


class _PaddedView:
    """hidden_states of an unpadded tower run: a tuple-like of [B, L, H] tensors built on demand from the packed [T, H] rows
    (zeros at the padding); only the layers somebody reads are scattered."""

    def __init__(self, packed, idx, B, L):
        self.packed, self.idx, self.B, self.L = packed, idx, B, L
        self._cache = {}

    def __len__(self):
        return len(self.packed)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[j] for j in range(*i.indices(len(self))))
        i = i % len(self.packed)
        if i not in self._cache:
            t = self.packed[i]
            dense = torch.zeros((self.B * self.L, t.shape[-1]), device=t.device, dtype=t.dtype)
            self._cache[i] = dense.index_copy(0, self.idx, t).view(self.B, self.L, -1)
        return self._cache[i]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class RobertaModel(HipModule, PretrainedMixin):
    """reference text.py:1084-1266."""

    def __init__(self, config, add_pooling_layer=True):
        super().__init__()
        self.config = config
        self.embeddings = RobertaEmbeddings(config)
        self.encoder = RobertaEncoder(config)
        self.pooler = RobertaPooler(config) if add_pooling_layer else None
        init_bert_weights(self, getattr(config, "initializer_range", 0.02))

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, cate_ids=None, head_mask=None,
                inputs_embeds=None, output_attentions=None, output_hidden_states=None, return_dict=None, allow_unpad=True, **unused):
        (self._root if "_root" in self.__dict__ else self).ensure_arena()
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if UNPAD and allow_unpad and cate_ids is None:
            hs = self._forward_unpadded(input_ids, attention_mask, token_type_ids, position_ids)
            if hs is not None:
                return BaseModelOutput(last_hidden_state=hs[-1], hidden_states=hs)
        e = self.embeddings(input_ids=input_ids, position_ids=position_ids, token_type_ids=token_type_ids, cate_ids=cate_ids)
        # allow_unpad = "nothing the caller reads depends on the padded rows" (every head of the reference reads [CLS] or valid spans; the
        # cross_attn multimodal layers, which attend over ALL text positions, pass False): the padded run may then at least skip what is
        # provably zero in its backward
        hs = self.encoder(e, attention_mask, masked_rows_dead=allow_unpad and cate_ids is None and not output_hidden_states)
        return BaseModelOutput(last_hidden_state=hs[-1], hidden_states=hs)

    def _forward_unpadded(self, input_ids, attention_mask, token_type_ids, position_ids):
        """IA_UNPAD=1: run the tower on the valid tokens only.  The reference pads every sequence to max length and computes on the
        padding too (data.py:558-559); nothing it returns depends on those rows (keys are masked, heads read valid positions), so
        the padded rows are dropped before the embedding kernel and every layer runs on sum(lengths) rows with the packed
        attention kernels.  Hidden states come back in the padded [B, L, H] shape (zeros at the padding) on demand."""
        emb, enc = self.embeddings, self.encoder
        B, L = input_ids.shape
        valid = (attention_mask != 0)
        lens = valid.sum(dim=1)
        T, Lmax = (int(v) for v in torch.stack((lens.sum(), lens.max())).tolist())       # one host sync per forward
        if T == B * L:
            return None                                                                # nothing to drop
        ids, tts, pids = emb._ids(input_ids, token_type_ids, position_ids)
        idx = valid.reshape(-1).nonzero(as_tuple=False).squeeze(1)                      # packed row -> padded row
        cu = torch.zeros(B + 1, device=ids.device, dtype=torch.int32)
        cu[1:] = torch.cumsum(lens, 0)
        training = self.training and torch.is_grad_enabled()
        p = emb.drop_p if training else 0.0
        take = lambda t: t.reshape(-1).index_select(0, idx).contiguous()
        pk = take(pids)
        order = torch.argsort(pk, stable=True).to(torch.int32)          # backward walks the rows position by position (embed.hip)
        e = Fn.EmbedLNFn.apply(emb.anchor, emb, take(ids), take(tts), pk, None, None, p, emb.stream_id, order)
        outs = Fn.EncoderStackFn.apply(e, enc.anchor, enc, None, B, Lmax, torch.is_grad_enabled(), cu)
        return _PaddedView((e,) + tuple(outs), idx, B, L)


def adopt(root, *children):
    """sub-models built on their own (CoCa passes ready-made encoders in) share the root's arena."""
    for c in children:
        for m in c.modules():
            m.__dict__["_root"] = root


class _PairTowerBase(HipModule, PretrainedMixin):
    def _finish(self, logits, probs, loss, src, tgt, labels, hidden_states=None):
        cfg = self.config
        if labels is not None and cfg.loss_type != "ce":
            loss = apply_loss(self.loss_fct, cfg, logits, labels, src, tgt)
        return SequenceClassifierOutput(loss=loss, logits=logits, probs=probs, src_embeds=src, tgt_embeds=tgt, hidden_states=hidden_states)


class AuxiliaryTaskPair(nn.Module):
    """reference text.py:66-102: for every (source attribute, target attribute) pair with the same key, the mean token vector of
    each `key:value` span -> dropout -> Linear(2H -> num_labels); labels say whether the values are equal.  `pair_indices` is
    the per-sample list the dataset builds (data.py:568-612): rows of (src start, src end, tgt start, tgt end, label).
    Returns the mean cross-entropy over all pairs of the batch (what `self.loss_fct(logits2, labels2)` adds at :1478-1480)."""

    def __init__(self, config):
        super().__init__()
        drop = getattr(config, "classifier_dropout", None)
        self.drop_p = float(drop if drop is not None else config.hidden_dropout_prob)
        self.out_proj = nn.Linear(config.hidden_size * 2, config.num_labels)

    def forward(self, taps, pair_indices, anchor, B, L, training):
        import numpy as np
        rows, ptr_, labels = [], [0], []
        for i, pi in enumerate(pair_indices):
            pi = pi.detach().cpu().numpy() if torch.is_tensor(pi) else np.asarray(pi)
            for j in range(pi.shape[0] if pi.ndim == 2 else 0):
                a0, a1, b0, b1, lab = (int(v) for v in pi[j])
                rows += [(i * L + a0, i * L + a1), (i * L + b0, i * L + b1)]
                labels.append(lab)
            ptr_.append(len(rows))
        ptr_ += [len(rows)] * (B + 1 - len(ptr_))
        if not labels:
            raise RuntimeError("auxiliary_task: no attribute pairs in this batch (the reference's torch.stack of an empty list fails likewise)")
        dev = taps[0].device
        spans = torch.tensor(rows, dtype=torch.int32, device=dev)
        span_ptr = torch.tensor(ptr_, dtype=torch.int32, device=dev)
        labels2 = torch.tensor(labels, dtype=torch.int64, device=dev)
        H = taps[0].shape[-1]
        means = [Fn.SpanMeanFn.apply(t.reshape(B * L, H), anchor, spans, span_ptr, B, L) for t in taps]
        m = means[0] if len(means) == 1 else torch.stack(means).mean(dim=0)
        x, y = m[0::2], m[1::2]
        if training and self.drop_p > 0:
            x, y = F.dropout(x, self.drop_p, True), F.dropout(y, self.drop_p, True)
        _logits, _probs, loss = Fn.PairHeadCEFn.apply(x, y, self.out_proj, labels2)
        return loss


class RobertaOneTower(_PairTowerBase):
    """reference text.py:1379-1492."""
    supports_auxiliary = True          # only this class builds AuxiliaryTaskPair in the reference (text.py:1412-1413)

    def __init__(self, config):
        super().__init__()
        self.num_labels = config.num_labels
        self.config = config
        if config.max_seq_len_pv is None:
            self.max_seq_len = config.max_seq_len
        elif config.max_seq_len is None:
            self.max_seq_len = config.max_seq_len_pv
        else:
            self.max_seq_len = config.max_seq_len + config.max_seq_len_pv
        self.cls_layers = [-int(i) for i in config.cls_layers.split(",")]
        self.roberta = self._make_backbone(config)
        if config.classification_method == "vec_sim":
            self.classifier = VecSimClassificationHead(config)
        else:
            self.classifier = RobertaClassificationHead(config)
        self.loss_fct = make_loss(config)
        if getattr(config, "auxiliary_task", False) and type(self).supports_auxiliary:
            self._build_auxiliary(config)
        init_bert_weights(self, getattr(config, "initializer_range", 0.02))
        adopt(self, self.roberta)

    def _build_auxiliary(self, config):
        if config.cls_pool != "avg" and len(self.cls_layers) > 1:
            raise ValueError("auxiliary_task needs a hidden_size-wide sequence output: one cls layer or cls_pool='avg' "
                             "(reference text.py:77: out_proj is Linear(2 * hidden_size, num_labels))")
        self.auxiliary_task = AuxiliaryTaskPair(config)

    def _make_backbone(self, config):
        return RobertaModel(config, add_pooling_layer=False)

    def _backbone(self, input_ids, attention_mask, token_type_ids, position_ids, cate_ids, inputs_embeds, image_indices):
        # the auxiliary attribute-pair task averages hidden states over caller-given spans (reference text.py:66-102), which may reach
        # into padded positions (the reference computes those rows like any other; the golden fixture's third sample does exactly that):
        # with it, the padded rows matter -- no unpadded run, no skipped query blocks in the attention backward
        # ... and so they do under vec_sim: the target embedding is read at the FIXED position max_seq_len (reference text.py:1463), which is a
        # padded position whenever a sequence is shorter than that (the eight-sample golden fixtures hold two such sequences: the row-wise
        # LayerNorm filter of round 6 caught it -- the 32-position attention blocks had not)
        rows_matter = hasattr(self, "auxiliary_task") or self.config.classification_method == "vec_sim"
        return self.roberta(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids,
                            cate_ids=cate_ids, allow_unpad=not rows_matter)

    def _tgt_index(self):
        return self.max_seq_len

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, cate_ids=None, head_mask=None,
                inputs_embeds=None, image_indices=None, labels=None, output_attentions=None, output_hidden_states=None,
                return_dict=None):
        self.ensure_arena()
        outputs = self._backbone(input_ids, attention_mask, token_type_ids, position_ids, cate_ids, inputs_embeds, image_indices)
        hs = outputs.hidden_states
        B, L, H = hs[-1].shape
        dev = hs[-1].device
        training = self.training and torch.is_grad_enabled()
        p = self.classifier.drop_p if training else 0.0
        taps = [hs[i] for i in self.cls_layers]

        def pick(index, sid):
            rows = cls_rows(B, L, index, dev)
            feats = [Fn.GatherRowsFn.apply(t.reshape(B * L, H), self.anchor, rows, 0.0, 0) for t in taps]
            f = torch.stack(feats).mean(dim=0) if self.config.cls_pool == "avg" else torch.cat(feats, dim=-1)
            if p > 0:
                f = F.dropout(f, p, True)
            return f

        if self.config.classification_method == "vec_sim":
            src, tgt, logits, probs = self.classifier(pick(0, 0), pick(self._tgt_index(), 1))
            loss = None
        else:
            ce = self.config.loss_type == "ce"
            logits, probs2, loss = self.classifier(pick(0, 0), labels if ce else None, inputs_embeds=inputs_embeds,
                                                   differentiable_logits=(labels is not None and not ce))
            src, tgt, probs = probs2[:, 0], probs2[:, 1], probs2[:, 1]
        out = self._finish(logits, probs, loss, src, tgt, labels, hidden_states=hs if output_hidden_states else None)
        if labels is not None and hasattr(self, "auxiliary_task"):
            if self.config.loss_type != "ce":
                raise ValueError("auxiliary_task adds self.loss_fct(logits2 [P, 2], labels2 [P]) (reference text.py:1478-1480): only "
                                 "--loss_type ce has that signature")
            out.loss = out.loss + self.auxiliary_task(taps, image_indices, self.anchor, B, L, training)
        return out


class RobertaTwoTower(_PairTowerBase):
    """reference text.py:1269-1376 (probs stay [B, 2], embeddings are the dropped-out CLS vectors: quirk A3)."""

    def __init__(self, config):
        super().__init__()
        self.num_labels = config.num_labels
        self.config = config
        self.roberta = self._make_backbone(config)
        self.classifier = TwoTowerClassificationHead(config.hidden_size, dropout=config.hidden_dropout_prob, num_labels=config.num_labels)
        self.loss_fct = make_loss(config)
        init_bert_weights(self, getattr(config, "initializer_range", 0.02))
        adopt(self, self.roberta)

    def _make_backbone(self, config):
        return RobertaModel(config, add_pooling_layer=False)

    def _both(self, a, b):
        if a is None or b is None:
            return None
        return torch.cat((a, b), dim=0)

    def _backbone(self, ids, mask, tts, pids, extra1=None, extra2=None):
        return self.roberta(ids, attention_mask=mask, token_type_ids=tts, position_ids=pids).last_hidden_state

    def forward(self, input_ids_1=None, attention_mask_1=None, token_type_ids_1=None, cate_ids_1=None, position_ids_1=None,
                input_ids_2=None, attention_mask_2=None, token_type_ids_2=None, cate_ids_2=None, position_ids_2=None,
                head_mask=None, inputs_embeds=None, labels=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                images_1=None, images_2=None):
        self.ensure_arena()
        if input_ids_1.shape != input_ids_2.shape:
            raise ValueError("both towers must be padded to the same length (the reference collate pads to max_length)")
        if attention_mask_1 is None:
            attention_mask_1 = torch.ones_like(input_ids_1)
        if attention_mask_2 is None:
            attention_mask_2 = torch.ones_like(input_ids_2)
        B, L = input_ids_1.shape
        seq = self._backbone(self._both(input_ids_1, input_ids_2), self._both(attention_mask_1, attention_mask_2),
                             self._both(token_type_ids_1, token_type_ids_2), self._both(position_ids_1, position_ids_2), images_1, images_2)
        L, H = seq.shape[1], seq.shape[-1]          # PKGM sequences grow after embedding (each relation -> 2 rows)
        dev = seq.device
        training = self.training and torch.is_grad_enabled()
        p = self.classifier.drop_p if training else 0.0
        flat = seq.reshape(2 * B * L, H)
        f1 = Fn.GatherRowsFn.apply(flat, self.anchor, cls_rows(B, L, 0, dev), p, 2001)
        f2 = Fn.GatherRowsFn.apply(flat, self.anchor, (cls_rows(B, L, 0, dev) + B * L).contiguous(), p, 2002)
        ce = self.config.loss_type == "ce"
        src, tgt, logits, probs, loss = self.classifier(f1, f2, labels if ce else None, differentiable_logits=(labels is not None and not ce))
        return self._finish(logits, probs, loss, src, tgt, labels)


# ------------------------------------------------------------------------------------------------ PKGM
class RobertaPKGMModel(HipModule, PretrainedMixin):
    """reference text.py:128-289."""
    weight_files = (ROBERTA_WEIGHTS_NAME, KG_WEIGHTS_NAME)

    def __init__(self, config, add_pooling_layer=False):
        super().__init__()
        self.config = config
        self.embeddings = RobertaPKGMEmbeddings(config)
        self.encoder = RobertaEncoder(config)
        self.pooler = None
        init_bert_weights(self, getattr(config, "initializer_range", 0.02))

    def forward(self, input_ids, attention_mask, token_type_ids, position_ids, inputs_embeds=None, head_mask=None, masked_rows_dead=False,
                **unused):
        """masked_rows_dead: the caller reads no hidden state of a masked position (RobertaEncoder.forward)"""
        (self._root if "_root" in self.__dict__ else self).ensure_arena()
        if input_ids is None or attention_mask is None or token_type_ids is None or position_ids is None:
            raise ValueError("You have to specify input_ids, attention_mask, token_type_ids and position_ids")
        e = self.embeddings(input_ids=input_ids, position_ids=position_ids, token_type_ids=token_type_ids)
        hs = self.encoder(e, attention_mask, masked_rows_dead=masked_rows_dead)
        return BaseModelOutput(last_hidden_state=hs[-1], hidden_states=hs)


class PKGMOneTower(RobertaOneTower):
    """reference text.py:691-783 (+ two-file from_pretrained :785-1080)."""
    supports_auxiliary = False
    weight_files = (ROBERTA_WEIGHTS_NAME, KG_WEIGHTS_NAME)

    def __init__(self, config):
        if not hasattr(config, "cls_layers"):
            config.cls_layers, config.cls_pool = "1", "cat"
        if getattr(config, "max_seq_len_pv", None) is None and not hasattr(config, "max_seq_len_pv"):
            config.max_seq_len_pv = None
        super().__init__(config)
        self.cls_layers = [-1]                      # PKGMOneTower always classifies on the last layer (text.py:753-760)

    def _make_backbone(self, config):
        return RobertaPKGMModel(config, add_pooling_layer=False)

    def _backbone(self, input_ids, attention_mask, token_type_ids, position_ids, cate_ids, inputs_embeds, image_indices):
        # the cls head reads position 0 of the last layer only; vec_sim reads a fixed second position (RobertaOneTower._backbone)
        return self.roberta(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids,
                            masked_rows_dead=self.config.classification_method != "vec_sim")

    def _tgt_index(self):
        return self.config.max_seq_len + 2 * self.config.max_pvs


class PKGMTwoTower(RobertaTwoTower):
    """reference text.py:292-391."""
    weight_files = (ROBERTA_WEIGHTS_NAME, KG_WEIGHTS_NAME)

    def _make_backbone(self, config):
        return RobertaPKGMModel(config, add_pooling_layer=False)


# --------------------------------------------------------------------------------------------- TextCNN
class _CpuEmbeddings(nn.Module):
    """RobertaEmbeddings for the TextCNN plumbing config (BASELINE.json configs[0], CPU via finetune_text.py):
    plain torch modules; this model family never touches the HIP engine (no kernel required)."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.padding_idx = config.pad_token_id
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size, padding_idx=self.padding_idx)

    def forward(self, input_ids):
        from .base import create_position_ids_from_input_ids
        pos = create_position_ids_from_input_ids(input_ids, self.padding_idx)
        e = self.word_embeddings(input_ids) + self.token_type_embeddings(torch.zeros_like(input_ids))
        e = e + self.position_embeddings(pos)
        return self.dropout(self.LayerNorm(e))


class TextCNN(nn.Module):
    """reference text.py:1496-1527."""

    def __init__(self, config, embedding_state_dict):
        super().__init__()
        filter_sizes = [int(i) for i in config.filter_sizes.split(",")]
        self.embedding1 = _CpuEmbeddings(config)
        self.embedding1.load_state_dict(embedding_state_dict, strict=False)
        self.embedding2 = _CpuEmbeddings(config)
        self.embedding2.load_state_dict(embedding_state_dict, strict=False)
        for value in self.embedding2.parameters():
            value.requires_grad = False
        self.convs1 = nn.ModuleList([nn.Conv2d(2, config.num_filters, (K, config.hidden_size)) for K in filter_sizes])
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, x):
        x = torch.stack((self.embedding1(x), self.embedding2(x)), dim=1)
        x = [F.relu(conv(x)).squeeze(3) for conv in self.convs1]
        x = [F.max_pool1d(i, i.size(2)).squeeze(2) for i in x]
        return self.dropout(torch.cat(x, 1))


class _TorchTwoTowerHead(nn.Module):
    def __init__(self, hidden_size, dropout=0.0, num_labels=2):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        self.out_proj = nn.Linear(hidden_size * 2, num_labels)

    def forward(self, a, b):
        x, y = self.dropout(a), self.dropout(b)
        logits = self.out_proj(torch.cat((x, y), dim=1))
        return x, y, logits, torch.softmax(logits, dim=1)


class TextCNNTwoTower(nn.Module):
    """reference text.py:1530-1609 — BASELINE.json configs[0], the reference's CPU-runnable plumbing case."""

    def __init__(self, config, embedding_state_dict):
        super().__init__()
        self.num_labels = config.num_labels
        self.config = config
        self.textcnn = TextCNN(config, embedding_state_dict)
        if config.classification_method == "vec_sim":
            raise NotImplementedError("TextCNN + vec_sim is not used by the reference scripts (finetune_text.py never sets similarity_measure, quirk A14)")
        hidden_size = len(config.filter_sizes.split(",")) * config.num_filters
        self.classifier = _TorchTwoTowerHead(hidden_size, config.hidden_dropout_prob)
        self.loss_fct = make_loss(config)

    def forward(self, input_ids_1=None, attention_mask_1=None, token_type_ids_1=None, position_ids_1=None, input_ids_2=None,
                attention_mask_2=None, token_type_ids_2=None, position_ids_2=None, head_mask=None, inputs_embeds=None, labels=None,
                output_attentions=None, output_hidden_states=None, return_dict=None):
        o1, o2 = self.textcnn(input_ids_1), self.textcnn(input_ids_2)
        src, tgt, logits, probs = self.classifier(o1, o2)
        src, tgt, probs = probs[:, 0], probs[:, 1], probs[:, 1]
        loss = apply_loss(self.loss_fct, self.config, logits, labels, src, tgt, ce_too=True) if labels is not None else None
        return SequenceClassifierOutput(loss=loss, probs=probs, logits=logits, src_embeds=src, tgt_embeds=tgt)
