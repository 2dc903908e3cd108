#!/usr/bin/env python3
"""Uniform model soup over the per-epoch checkpoints of finetune_multimodal.py, then prediction with the averaged weights:
CLI-compatible with the reference's model_soup_multimodal.py (flags :34-86 = the finetune flags + --epochs, with
--file_state_dict a pattern containing `{}` for the epoch; soup :223-240; output names :242-247,284).  Parameters are averaged
over the listed epochs, non-parameter entries (buffers) are taken from the last listed checkpoint, as the reference does.
The averaging is a host-side pass over the state_dicts; the prediction runs on the HIP engine like finetune_multimodal.py --do_pred."""
import os
from collections import OrderedDict

import torch

import finetune_multimodal as FT
from item_alignment_amd.utils import logger


def uniform_soup(model, pattern, epochs):
    """reference model_soup_multimodal.py:223-239"""
    names = set(dict(model.named_parameters()))
    st = OrderedDict()
    for epoch in epochs:
        sd = torch.load(pattern.format(epoch), map_location="cpu")
        for key, val in sd.items():
            if key not in names:
                st[key] = val                      # buffers: the last listed epoch wins
            elif key not in st:
                st[key] = val.clone()
            else:
                st[key] += val
    for key, val in st.items():
        if key in names:
            val /= len(epochs)
    return st


def main():
    args = FT.get_parser(lambda a: a("--epochs", required=True, type=str, help="epochs to be used for uniform soup, e.g. 0,1,2"))
    pattern, args.file_state_dict = args.file_state_dict, None           # the pattern is not a checkpoint to resume from
    if not pattern or "{}" not in pattern:
        raise ValueError("--file_state_dict must be a path pattern with {} in place of the epoch")
    args.do_train, args.do_eval, args.do_pred = False, False, True
    args.pred_tag = "uniform_soup_"

    def soup(args, model):
        epochs = args.epochs.split(",")
        st = uniform_soup(model, pattern, epochs)
        missing, unexpected = model.load_state_dict(st, strict=False)
        if unexpected:
            raise RuntimeError(f"unexpected keys in the checkpoints: {unexpected[:5]}")
        logger.info(f"Finished uniform soup on epochs: {epochs}")
        out_dir = os.path.join(args.output_dir, "-".join(str(getattr(args, f)) for f in ("model_name", "data_version", "interaction_type",
                                                                                          "classification_method", "ensemble", "loss_type")))
        os.makedirs(out_dir, exist_ok=True)
        torch.save(st, os.path.join(out_dir, f"multimodal_finetune-uniform_soup-epoch-{args.epochs}.bin"))
        logger.info("Finished saving uniform soup model")

    FT.main(args, before_run=soup)


if __name__ == "__main__":
    main()
