"""Host-side mirror of the reference's src/models/multimodal.py: RobertaImage{Model,OneTower,TwoTower}
(RoBERTa + pre-extracted image embeddings) and CoCaForItemAlignment (RoBERTa text tower + ViT image tower).
"""
import torch
from torch import nn

from . import functional as Fn
from .base import (ACT_NONE, BaseModelOutput, HipModule, RobertaEmbeddings, RobertaEncoder, RobertaImageEmbeddings, RobertaPooler,
                   SequenceClassifierOutput, TwoTowerClassificationHead, VecSimClassificationHead, cls_rows, init_bert_weights)
from .loss import apply_loss, make_loss
from .text import PretrainedMixin, RobertaOneTower, RobertaTwoTower, adopt


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

    def embed_text(self, input_ids, attention_mask, token_type_ids, position_ids):
        out = self.text_encoder(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids)
        return out.last_hidden_state

    def embed_image(self, images):
        return self.img_encoder.forward_features(images)


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
        if config.ensemble == "cross_attn":
            raise NotImplementedError("--ensemble cross_attn (reference multimodal.py:529-706,1003-1013) is SURVEY §8(f) rank 3: "
                                      "not built this round (DESIGN.md, next)")
        img_dim = getattr(image_encoder, "num_features", config.hidden_size)
        if img_dim != config.hidden_size:
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
        cat = lambda a, b: None if a is None else torch.cat((a, b), dim=0)
        # both items of a pair go through each shared-weight encoder as one 2B batch
        img_tok = self.coca.embed_image(torch.cat((images_1, images_2), dim=0))                     # [2B, N, Hi] bf16
        txt = self.coca.embed_text(cat(input_ids_1, input_ids_2), cat(attention_mask_1, attention_mask_2),
                                   cat(token_type_ids_1, token_type_ids_2), cat(position_ids_1, position_ids_2))   # [2B, L, H]
        H = txt.shape[-1]
        dev = txt.device
        t_cls = Fn.GatherRowsFn.apply(txt.reshape(2 * B * L, H), self.anchor, cls_rows(2 * B, L, 0, dev), 0.0, 0)   # text_tokens[:, 0]
        i_cls = self.coca.img_encoder.forward_head(img_tok, pre_logits=True)                                       # [2B, Hi] fp32
        if self.img_proj is not None:
            i_cls = Fn.LinearSmallFn.apply(i_cls, self.img_proj.weight, self.img_proj, ACT_NONE)
        emb = t_cls + i_cls                                                                                          # multimodal.py:1015
        e1, e2 = emb[:B].contiguous(), emb[B:].contiguous()
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
