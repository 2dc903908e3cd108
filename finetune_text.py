#!/usr/bin/env python3
"""Text-only pair matching (RoBERTa / PKGM / TextCNN, one- or two-tower): CLI-compatible with the reference's
finetune_text.py (flags :33-88, model dispatch by substring of --model_name :218-241, loop :396-492), running on
the MI355X HIP engine (TextCNN = the reference's CPU-runnable plumbing config, plain torch)."""
import argparse
import json
import os

import torch

from item_alignment_amd import train
from item_alignment_amd.cli_common import add_common_flags, freeze_and_resume, load_config, load_tokenizer, pick_device
from item_alignment_amd.data.datasets import (PKGMOneTowerDataset, PKGMTwoTowerDataset, RobertaOneTowerDataset, RobertaTwoTowerDataset,
                                              collate_one_tower, collate_two_tower)
from item_alignment_amd.utils import ROBERTA_WEIGHTS_NAME, logger
from src.models import PKGMOneTower, PKGMTwoTower, RobertaOneTower, RobertaTwoTower, TextCNNTwoTower


def get_parser():
    p = argparse.ArgumentParser()
    add_common_flags(p)
    a = p.add_argument
    a("--interaction_type", required=True, type=str)
    a("--classification_method", required=True, type=str)
    a("--similarity_measure", required=True, type=str)
    a("--loss_type", required=True, type=str)
    a("--type_vocab_size", default=2, type=int)
    a("--do_lower_case", default=True, type=bool)
    a("--max_seq_len", default=None, type=int)
    a("--max_seq_len_pv", default=None, type=int)
    a("--max_position_embeddings", default=512, type=int)
    a("--max_pvs", default=30, type=int)
    a("--cls_layers", default="1", type=str)
    a("--cls_pool", default="cat", type=str)
    a("--auxiliary_task", action="store_true")
    a("--filter_sizes", default="1,2,3,5", type=str)
    a("--num_filters", default=36, type=int)
    return p.parse_args()


def load_raw_data(args):
    """reference finetune_text.py:91-150 (valid and test both read finetune_test.tsv, quirk A12)."""
    info = {}
    with open(os.path.join(args.data_dir, "raw", "item_info.jsonl"), "r", encoding="utf-8") as r:
        for line in r:
            if line.strip():
                d = json.loads(line)
                info[d["item_id"]] = d
    cate2id = json.load(open(os.path.join(args.data_dir, "processed", "cate2id.json"), "r", encoding="utf-8"))

    def read(name):
        rows = []
        with open(os.path.join(args.data_dir, "processed", args.data_version, name), "r", encoding="utf-8") as r:
            for line in r:
                if not line.strip("\n"):
                    continue
                label, sid, st, sp, tid, tt, tp = line.strip("\n").split("\t")
                rows.append((label, sid, cate2id[info[sid]["cate_name"]], st, sp, tid, cate2id[info[tid]["cate_name"]], tt, tp))
        return rows
    return read("finetune_train.tsv"), read("finetune_test.tsv"), read("finetune_test.tsv")


def load_kg_tokenizer(args):
    """reference finetune_text.py:153-172."""
    def read(name):
        out = {}
        with open(os.path.join(args.data_dir, "processed", name), "r", encoding="utf-8") as r:
            for line in r:
                if line.strip("\n"):
                    k, v = line.strip("\n").split("\t")
                    out[k] = int(v)
        return out
    return read("entity2id.txt"), read("relation2id.txt")


def main():
    args = get_parser()
    train.seed_everything(args.seed)
    tokenizer = load_tokenizer(args)
    config = load_config(args.config_file, interaction_type=args.interaction_type, type_vocab_size=args.type_vocab_size,
                         classification_method=args.classification_method, loss_type=args.loss_type, max_seq_len=args.max_seq_len,
                         max_seq_len_pv=args.max_seq_len_pv, max_pvs=args.max_pvs, max_position_embeddings=args.max_position_embeddings,
                         loss_margin=args.margin, cls_layers=args.cls_layers, cls_pool=args.cls_pool, filter_sizes=args.filter_sizes,
                         num_filters=args.num_filters, auxiliary_task=args.auxiliary_task, ensemble=None)
    # NB the reference never sets config.similarity_measure here (quirk A14); it is set so that vec_sim works at all
    config.similarity_measure = args.similarity_measure
    one = args.interaction_type == "one_tower"
    if args.interaction_type not in ("one_tower", "two_tower"):
        raise ValueError("interaction type should be: one_tower or two_tower")
    if "pkgm" in args.model_name:
        kg_ent, kg_rel = load_kg_tokenizer(args)
        logger.info(f"# kg entities: {len(kg_ent)}, # kg relations: {len(kg_rel)}")
        model = (PKGMOneTower if one else PKGMTwoTower).from_pretrained(args.pretrained_model_path, config=config, ignore_mismatched_sizes=True)
    elif "bert" in args.model_name:
        model = (RobertaOneTower if one else RobertaTwoTower).from_pretrained(args.pretrained_model_path, config=config,
                                                                             ignore_mismatched_sizes=True)
    elif "textcnn" in args.model_name:
        f = os.path.join(args.pretrained_model_path, ROBERTA_WEIGHTS_NAME)
        sd = torch.load(f, map_location="cpu") if os.path.exists(f) else {}
        model = TextCNNTwoTower(config=config, embedding_state_dict={k[11:]: v for k, v in sd.items() if "embedding" in k})
    else:
        raise ValueError("model name should be: roberta or pkgm")
    freeze_and_resume(args, model)
    train_data, valid_data, test_data = load_raw_data(args)
    logger.info(f"# train samples: {len(train_data)}, # valid samples: {len(valid_data)}, # test samples: {len(test_data)}")

    def make(data):
        if "pkgm" in args.model_name:
            cls = PKGMOneTowerDataset if one else PKGMTwoTowerDataset
            return cls(data, tokenizer, kg_ent, kg_rel, args.max_seq_len, args.max_pvs, args.classification_method)
        if one:
            return RobertaOneTowerDataset(data, tokenizer, args.max_seq_len, args.classification_method, args.max_seq_len_pv, args.auxiliary_task)
        return RobertaTwoTowerDataset(data, tokenizer, args.max_seq_len, args.max_seq_len_pv)

    device = pick_device(model)
    model.to(device)

    def call(model, b):
        if one:   # collate_one_tower[2:] = pair_indices, input_ids, token_type_ids, attention_mask, position_ids, labels
            pair_indices, input_ids, segment_ids, input_mask, position_ids, labels = b
            return model(input_ids=input_ids, token_type_ids=segment_ids, attention_mask=input_mask, position_ids=position_ids,
                         labels=labels, output_hidden_states=True, image_indices=pair_indices)
        ids1, mask1, tt1, ids2, mask2, tt2, position_ids, labels = b
        return model(input_ids_1=ids1, attention_mask_1=mask1, token_type_ids_1=tt1, position_ids_1=position_ids, input_ids_2=ids2,
                     attention_mask_2=mask2, token_type_ids_2=tt2, position_ids_2=position_ids, labels=labels)

    sim_or_ens = args.similarity_measure
    args.path_tail = sim_or_ens
    train.run(args, model, dict(train=make(train_data) if args.do_train else None, valid=make(valid_data) if args.do_eval else None,
                                test=make(test_data) if args.do_pred else None),
              collate_one_tower if one else collate_two_tower, call, "text_finetune",
              ["model_name", "data_version", "interaction_type", "classification_method", "path_tail", "loss_type"], device)


if __name__ == "__main__":
    main()
