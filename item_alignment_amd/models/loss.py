"""Losses of the reference (src/models/loss.py) and the loss dispatch every tower shares
(text.py:1283-1292 construction, :1356-1364 use).  These act on [B]- or [B,2]-sized fp32 tensors after the
fused head; cross-entropy itself is fused into the head kernel (functional.PairHeadCEFn)."""
import torch
from torch import nn
from torch.nn.modules.loss import _Loss


class EuclideanDistanceLoss(_Loss):
    """reference loss.py:7-68 — note forward is pow(input, target) (quirk A5), kept as is."""

    def __init__(self, size_average=None, reduce=None, reduction="mean"):
        super().__init__(size_average, reduce, reduction)

    def forward(self, input, target):
        loss = torch.pow(input, target)
        if self.reduction == "sum":
            return loss.sum()
        if self.reduction == "mean":
            return loss.mean()
        return loss


class HingeLoss(_Loss):
    """reference loss.py:71-134: mean(max(0, margin - input * target))."""

    def __init__(self, margin=1.0, size_average=None, reduce=None, reduction="mean"):
        super().__init__(size_average, reduce, reduction)
        self.margin = margin

    def forward(self, input, target):
        loss = torch.clamp(self.margin - input * target, min=0)
        if self.reduction == "sum":
            return loss.sum()
        if self.reduction == "mean":
            return loss.mean()
        return loss


def make_loss(config):
    lt = config.loss_type
    if lt == "cosine":
        return nn.CosineEmbeddingLoss(margin=config.loss_margin)
    if lt == "bce":
        return nn.BCEWithLogitsLoss()
    if lt == "euclidean":
        return EuclideanDistanceLoss()
    if lt == "hinge":
        return HingeLoss(margin=config.loss_margin)
    return nn.CrossEntropyLoss()


def apply_loss(loss_fct, config, logits, labels, src_embeds, tgt_embeds, ce_too=False):
    lt = config.loss_type
    if lt == "cosine":
        return loss_fct(src_embeds, tgt_embeds, (labels * 2 - 1).view(-1))
    if lt == "ce":
        if not ce_too:
            raise RuntimeError("cross-entropy is computed by the fused head kernel")
        return loss_fct(logits.view(-1, config.num_labels), labels.view(-1))
    if lt in ("hinge", "euclidean"):
        return loss_fct(logits.view(-1), (labels * 2 - 1).view(-1))
    return loss_fct(logits.view(-1), labels.view(-1))
