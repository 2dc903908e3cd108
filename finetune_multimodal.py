#!/usr/bin/env python3
"""Multimodal pair matching: RoBERTa + pre-extracted image embeddings (roberta_image_*) and CoCa (RoBERTa + ViT).
CLI-compatible with the reference's finetune_multimodal.py (flags :33-93, dispatch :209-231, loop :371-468)."""
import argparse
import json
import os

import torch

from item_alignment_amd import train
from item_alignment_amd.cli_common import add_common_flags, freeze_and_resume, load_config, load_tokenizer, pick_device
from item_alignment_amd.data.datasets import (PairedMultimodalDataset, RobertaImageOneTowerDataset, RobertaImageTwoTowerDataset,
                                              collate_coca_pair, collate_multimodal, collate_multimodal_two_tower)
from item_alignment_amd.models.image import check_image_encoder_name, create_model
from item_alignment_amd.utils import VIT_WEIGHTS_NAME, logger
from src.models import CoCaForItemAlignment, RobertaImageOneTower, RobertaImageTwoTower, RobertaModel


def get_parser(extra_flags=None):
    p = argparse.ArgumentParser()
    add_common_flags(p)
    a = p.add_argument
    if extra_flags is not None:
        extra_flags(a)
    a("--interaction_type", required=True, type=str)
    a("--classification_method", required=True, type=str)
    a("--ensemble", required=True, type=str, help="begin | end (roberta_image) ; sum | cross_attn (coca)")
    a("--loss_type", required=True, type=str)
    a("--type_vocab_size", default=2, type=int)
    a("--similarity_measure", default="NA", type=str)
    a("--do_lower_case", default=True, type=bool)
    a("--max_seq_len", default=50, type=int)
    a("--max_seq_len_pv", default=305, type=int)
    a("--max_position_embeddings", default=512, type=int)
    a("--cls_layers", default="1", type=str)
    a("--cls_pool", default="cat", type=str)
    a("--image_size", default=384, type=int)
    a("--image_hidden_size", default=3072, type=int)
    a("--image_model_name", default="vit_base_patch16_384", type=str)
    a("--hflip", default=0.5, type=float)
    a("--color_jitter", default=None, type=float)
    args = p.parse_args()
    if "coca" in args.model_name:
        check_image_encoder_name(p, args.image_model_name)
    return args


def load_raw_data(args):
    """reference finetune_multimodal.py:96-165."""
    def rows(name):
        with open(os.path.join(args.data_dir, "processed", args.data_version, name), "r", encoding="utf-8") as r:
            return [line.strip("\n").split("\t") for line in r if line.strip("\n")]
    if "roberta_image" in args.model_name:
        return rows("finetune_train.tsv"), rows("finetune_test.tsv"), rows("finetune_test.tsv")
    id2image = {}
    with open(os.path.join(args.data_dir, "raw", "item_info.jsonl"), "r", encoding="utf-8") as r:
        for line in r:
            if line.strip():
                d = json.loads(line)
                id2image[d["item_id"]] = d["item_image_name"]

    def with_images(name):
        out = []
        for label, sid, st, sp, tid, tt, tp in rows(name):
            img = lambda i: os.path.join(args.data_dir, "raw", "item_images", id2image[i])
            out.append((label, sid, st, sp, img(sid), tid, tt, tp, img(tid)))
        return out
    return with_images("finetune_train.tsv"), with_images("finetune_test.tsv"), with_images("finetune_test.tsv")


def main(args=None, before_run=None):
    """before_run(args, model): hook between model construction and the train / eval / predict harness (model_soup_multimodal.py)"""
    args = args or get_parser()
    train.seed_everything(args.seed)
    tokenizer = load_tokenizer(args)
    config = load_config(args.config_file, interaction_type=args.interaction_type, type_vocab_size=args.type_vocab_size,
                         classification_method=args.classification_method, similarity_measure=args.similarity_measure,
                         loss_type=args.loss_type, max_seq_len=args.max_seq_len, max_seq_len_pv=args.max_seq_len_pv,
                         max_position_embeddings=args.max_position_embeddings, loss_margin=args.margin, cls_layers=args.cls_layers,
                         cls_pool=args.cls_pool, ensemble=args.ensemble, image_hidden_size=args.image_hidden_size, image_size=args.image_size)
    msl = args.max_seq_len if args.max_seq_len_pv is None else (args.max_seq_len_pv if args.max_seq_len is None else args.max_seq_len + args.max_seq_len_pv)
    one = args.interaction_type == "one_tower"
    if one:
        assert args.max_position_embeddings >= 2 * msl + 2
    if args.interaction_type not in ("one_tower", "two_tower"):
        raise ValueError("interaction type should be: one_tower or two_tower")
    coca = False
    if "roberta_image" in args.model_name:
        model = (RobertaImageOneTower if one else RobertaImageTwoTower).from_pretrained(args.pretrained_model_path, config=config,
                                                                                       ignore_mismatched_sizes=True)
    elif "coca" in args.model_name:
        coca = True
        text_encoder = RobertaModel.from_pretrained(args.pretrained_model_path, config=config)
        image_encoder = create_model(args.image_model_name, pretrained=True, img_size=args.image_size)
        f = os.path.join(args.pretrained_model_path or "", VIT_WEIGHTS_NAME)
        if os.path.exists(f):
            image_encoder.load_state_dict(torch.load(f, map_location="cpu"), strict=False)
        else:
            logger.warning(f"{f} not found: the image encoder keeps its random initialisation (no network for timm weights)")
        model = CoCaForItemAlignment(config, image_encoder, text_encoder)
    else:
        raise ValueError(f"Unsupported model name: {args.model_name}")
    freeze_and_resume(args, model)
    if before_run is not None:
        before_run(args, model)
    train_data, valid_data, test_data = load_raw_data(args)
    logger.info(f"# train samples: {len(train_data)}, # valid samples: {len(valid_data)}, # test samples: {len(test_data)}")

    def make(data, training):
        if coca:
            return PairedMultimodalDataset(data, ensemble=args.ensemble, image_size=args.image_size, is_training=training,
                                           text_tokenizer=tokenizer, max_seq_len=args.max_seq_len, max_seq_len_pv=args.max_seq_len_pv,
                                           hflip=args.hflip, color_jitter=args.color_jitter, raw=args.gpu_preproc)
        cls = RobertaImageOneTowerDataset if one else RobertaImageTwoTowerDataset
        return cls(data, tokenizer, max_seq_len=args.max_seq_len, max_seq_len_pv=args.max_seq_len_pv, ensemble=args.ensemble)

    collate = collate_coca_pair if coca else (collate_multimodal if one else collate_multimodal_two_tower)
    device = pick_device(model)
    model.to(device)

    def call(model, b):
        if not coca and one:
            image_indices, src_embs, tgt_embs, input_ids, segment_ids, input_mask, position_ids, labels = b
            return model(input_ids=input_ids, token_type_ids=segment_ids, attention_mask=input_mask, position_ids=position_ids, labels=labels,
                         output_hidden_states=True, inputs_embeds=[src_embs, tgt_embs], image_indices=image_indices)
        ids1, mask1, tt1, pos1, img1, ids2, mask2, tt2, pos2, img2, labels = b
        if coca:
            return model(ids1, mask1, tt1, pos1, img1, ids2, mask2, tt2, pos2, img2, labels=labels)
        return model(input_ids_1=ids1, token_type_ids_1=tt1, attention_mask_1=mask1, position_ids_1=pos1, images_1=img1, input_ids_2=ids2,
                     token_type_ids_2=tt2, attention_mask_2=mask2, position_ids_2=pos2, images_2=img2, labels=labels)

    # quirk A11 (reference :332): the CoCa validation set is built from train_data; kept
    valid_src = train_data if coca else valid_data
    train.run(args, model, dict(train=make(train_data, True) if args.do_train else None, valid=make(valid_src, False) if args.do_eval else None,
                                test=make(test_data, False) if args.do_pred else None),
              collate, call, "multimodal_finetune",
              ["model_name", "data_version", "interaction_type", "classification_method", "ensemble", "loss_type"], device)


if __name__ == "__main__":
    main()
