"""Pieces the three finetune_*.py scripts share: flags common to all of them, config /
tokenizer loading, the optional freeze / resume handling (reference finetune_multimodal.py:233-267)."""
import json
import os
from types import SimpleNamespace

import torch

from .utils import BOS_TOKEN, logger


def add_common_flags(parser, config_required=True):
    a = parser.add_argument
    a("--data_dir", required=True, type=str, help="模型训练数据地址")
    a("--output_dir", required=True, type=str, help="The output directory where the model checkpoints will be written.")
    a("--config_file", required=config_required, default=None, type=str, help="The config file which specified the model details.")
    a("--model_name", required=True, type=str, help="model saving name")
    a("--data_version", required=True, type=str, help="data version")
    a("--do_train", action="store_true")
    a("--do_eval", action="store_true")
    a("--do_pred", action="store_true")
    a("--seed", default=2345, type=int)
    a("--train_batch_size", default=64, type=int, help="Total batch size for training.")
    a("--eval_batch_size", default=64, type=int)
    a("--learning_rate", default=1e-3, type=float)
    a("--start_epoch", default=0, type=int)
    a("--num_train_epochs", default=10, type=int)
    a("--weight_decay", default=1e-5, type=float)
    a("--log_steps", default=10, type=int)
    a("--pretrained_model_path", default=None, type=str)
    a("--file_state_dict", default=None, type=str)
    a("--parameters_to_freeze", default=None, type=str)
    a("--threshold", default=0.5, type=float)
    a("--warmup_proportion", default=0.3, type=float)
    a("--gradient_accumulation_steps", default=1, type=int)
    a("--adam_epsilon", default=1e-8, type=float)
    a("--fp16", action="store_true", help="kept for CLI compatibility: the HIP engine always computes in bf16 with fp32 master weights")
    a("--margin", default=1.0, type=float)
    # not in the reference (its DataLoader has 0 workers and transforms run inside __getitem__): host-side decode workers and
    # the GPU resize / normalise kernels of SURVEY §8(f) rank 1.  Defaults keep the reference behaviour.
    a("--num_workers", default=0, type=int, help="DataLoader worker processes for decode / tokenisation")
    a("--gpu_preproc", action="store_true", help="decode images to uint8 on the host, the transform (resize / crop / flip / colour jitter / normalise) on the GPU, bit-identical to the Pillow path")
    a("--unpad", action="store_true", help="run the RoBERTa towers on the valid tokens only (no compute on the padding; same outputs and "
      "gradients, models/text.py RobertaModel._forward_unpadded).  Same as IA_UNPAD=1")


def load_config(path, **overrides):
    """BertConfig.from_json_file equivalent without the transformers dependency: every JSON key becomes an attribute,
    BertConfig defaults the reference relies on are filled in, then the script's flags are copied on top."""
    defaults = dict(hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12, pad_token_id=0,
                    type_vocab_size=2, initializer_range=0.02, num_labels=2, classifier_dropout=None, max_position_embeddings=512,
                    similarity_measure="NA", ensemble=None, auxiliary_task=False, cls_layers="1", cls_pool="cat", loss_margin=1.0,
                    max_seq_len=None, max_seq_len_pv=None, max_pvs=0)
    d = dict(defaults)
    if path:
        with open(path, "r", encoding="utf-8") as f:
            d.update(json.load(f))
    d.update(overrides)
    return SimpleNamespace(**d)


def load_tokenizer(args):
    """reference finetune_text.py:186-189."""
    from transformers import BertTokenizer
    tk = BertTokenizer.from_pretrained(args.pretrained_model_path, do_lower_case=args.do_lower_case)
    tk.do_basic_tokenize = False
    # transformers 4.20 (the reference's pin) with do_basic_tokenize=False keeps every whitespace-separated word whole, so the
    # image placeholder "[unused99]" (data.py:9-10, id 99) and the BOS marker reach the vocabulary lookup intact; the 5.x tokenizer
    # backend splits bracketed words at the punctuation unless they are registered as special tokens -- same ids either way
    # Only words the checkpoint's vocabulary already holds are registered: a word it lacks would be APPENDED with an id >= vocab_size,
    # beyond the embedding table the gather kernel indexes; the reference only sets tokenizer.bos_token (finetune_text.py:189), so such a
    # word resolves to [UNK] there -- and here.
    known = [w for w in ("[unused99]", BOS_TOKEN) if w in tk.vocab]
    if known:
        size = len(tk)
        try:
            tk.add_special_tokens({"additional_special_tokens": known})
        except TypeError as e:      # an older transformers without this keyword set: the whitespace path above already covers it
            logger.warning(f"could not register the image / BOS placeholders as special tokens: {e!r}")
        if len(tk) != size:
            raise RuntimeError(f"registering {known} grew the tokenizer from {size} to {len(tk)} entries")
    tk.bos_token = BOS_TOKEN
    logger.info(f"vocab size: {tk.vocab_size}")
    return tk


def freeze_and_resume(args, model):
    if args.parameters_to_freeze is not None:
        names = json.load(open(args.parameters_to_freeze, "r", encoding="utf-8"))
        frozen = []
        for key, value in dict(model.named_parameters()).items():
            if key.replace("roberta.", "") in names:
                frozen.append(key)
                value.requires_grad = False
        logger.info(f"Parameters freezed: {frozen}")
    if args.file_state_dict is not None:
        model.load_state_dict(torch.load(args.file_state_dict, map_location="cpu"))


def pick_device(model):
    """HIP models need the GPU.  TextCNN (BASELINE config C1, the reference's CPU plumbing case) is plain torch and stays on the host
    cores: there is no torch-eager GPU path in this package."""
    from .models.base import HipModule
    if isinstance(model, HipModule):
        if not torch.cuda.is_available():
            raise SystemExit("this model runs on the MI355X HIP engine only (no CPU path); no GPU is visible")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        return torch.device("cuda", local)
    return torch.device("cpu")
