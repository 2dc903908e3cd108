"""Train / eval / predict harness shared by finetune_{text,image,multimodal}.py.

It reproduces the reference's loop (finetune_multimodal.py:371-468 train, :470-563 eval, :565-569 checkpoint,
:661-775 predict; the same blocks in finetune_text.py:396-492 and finetune_image.py:310-348): zero_grad ->
forward -> loss (/ accumulation) -> backward -> AdamW(beta=(0.9,0.98)) + linear warm-up/decay schedule, loss
logged every log_steps, P/R/F1 swept over thresholds 0.1..0.9, state_dict saved per epoch under the reference's
file names.  What is new: the optimiser is one fused HIP launch over the parameter arena, `--fp16` selects the
engine's bf16 path (it is always on for the HIP models; there is no GradScaler), and the loop is data-parallel
when launched with torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK): the global batch (--train_batch_size
keeps its meaning "total batch size") is sharded across ranks and gradients are all-reduced over RCCL.
"""
import json
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

from . import dist as iadist
from .utils import logger


def seed_everything(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def linear_schedule_with_warmup(step, num_warmup_steps, num_training_steps):
    """transformers.get_linear_schedule_with_warmup lambda (reference finetune_multimodal.py:315)."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))


class TorchAdamW:
    """Reference optimiser construction (two parameter groups, finetune_multimodal.py:296-308) for models that do not
    live in a HIP parameter arena (the TextCNN CPU plumbing config)."""

    def __init__(self, model, lr, eps, weight_decay):
        no_decay = ["bias", "LayerNorm.weight"]
        groups = [{"params": [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
                  {"params": [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
        self.opt = torch.optim.AdamW(groups, lr=lr, eps=eps, betas=(0.9, 0.98))
        self.base_lr = lr

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self, lr_mult, grad_scale=1.0):
        for g in self.opt.param_groups:
            g["lr"] = self.base_lr * lr_mult
        self.opt.step()


class ArenaAdamW:
    """Fused AdamW over the flat parameter arena (one HIP launch; same update rule and parameter-group semantics)."""

    def __init__(self, model, lr, eps, weight_decay):
        self.arena = model.param_arena
        self.base_lr, self.eps, self.wd = lr, eps, weight_decay

    def zero_grad(self):
        self.arena.zero_grad()

    def step(self, lr_mult, grad_scale=1.0):
        self.arena.adamw_step(self.base_lr * lr_mult, betas=(0.9, 0.98), eps=self.eps, weight_decay=self.wd, grad_scale=grad_scale)


def model_dir(args, kind_fields):
    """reference finetune_multimodal.py:349 / finetune_text.py:373 / finetune_image.py:287 directory naming."""
    return os.path.join(args.output_dir, "-".join(str(getattr(args, f)) for f in kind_fields))


def run(args, model, datasets, collate_fn, call_model, checkpoint_name, path_fields, device):
    """datasets: dict(train=..., valid=..., test=...) (entries may be None); call_model(model, batch) -> output where
    batch = collate output from index 2 on, already on `device`."""
    from .models import functional as Fn
    if getattr(args, "unpad", False):
        from .models import text as _text
        _text.UNPAD = True
    rank, world, _ = iadist.init_from_env(device.type)
    is_hip = hasattr(model, "param_arena") and device.type == "cuda"
    out_dir = model_dir(args, path_fields)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)

    pipelines = {}

    def gpu_images(raw_batch):
        """RawImageBatch -> normalised [B, 3, S, S] on the device (data/gpu_preproc.py): the reference's transform, parameters drawn on
        the host, arithmetic on the GPU."""
        from .data.gpu_preproc import GpuImagePipeline
        size = int(getattr(args, "image_size", 0) or 0)
        pipe = pipelines.get(size)
        if pipe is None:
            pipe = pipelines[size] = GpuImagePipeline(size, device)
        return pipe.process(raw_batch.items)

    def to_dev(batch):
        from .data.datasets import RawImageBatch
        return tuple(gpu_images(t) if isinstance(t, RawImageBatch) else (t.to(device=device, non_blocking=True) if torch.is_tensor(t) else t)
                     for t in batch)

    nw = int(getattr(args, "num_workers", 0) or 0)
    dl_kw = dict(num_workers=nw, pin_memory=(device.type == "cuda"), persistent_workers=False)
    if nw > 0:
        # never fork() a process that owns HIP streams: workers come from a clean fork server and only see the pickled dataset
        dl_kw["multiprocessing_context"] = "forkserver"

    def evaluate(dataset, tag):
        """Scores the whole set.  Under data parallelism every rank scores positions rank::world and the (score, label) rows of
        all ranks are concatenated everywhere: no rank sits in a collective while another one evaluates alone.  The
        P/R/F1 sweep does not depend on the order, and a rank may come back with fewer rows than it was dealt (the collates drop
        samples whose image failed to load, data.py:44,84) or with none at all."""
        model.eval()
        n = len(dataset)
        mine = list(range(rank, n, world)) if world > 1 else None
        loader = DataLoader(dataset if mine is None else Subset(dataset, mine), batch_size=args.eval_batch_size, shuffle=False,
                            collate_fn=collate_fn, **dl_kw)
        probs_all, labels_all = None, None
        with torch.no_grad():
            for batch in loader:
                b = to_dev(batch[2:])
                out = call_model(model, b)
                probs, labels = out.probs.float().cpu().numpy(), b[-1].cpu().numpy()
                if probs.ndim == 2:
                    # RobertaTwoTower / PKGMTwoTower / RobertaImageTwoTower hand back softmax probs [B, 2] (quirk A3); the
                    # reference appends them flat and its sklearn call then fails on 2B vs B samples: score P(match) instead
                    probs = probs[:, 1]
                probs_all = probs if probs_all is None else np.append(probs_all, probs)
                labels_all = labels if labels_all is None else np.append(labels_all, labels)
        if world > 1:
            rows = np.empty((0, 2)) if probs_all is None else np.stack([np.asarray(probs_all, np.float64), np.asarray(labels_all, np.float64)], 1)
            rows = iadist.gather_rows(rows)
            probs_all, labels_all = rows[:, 0], rows[:, 1].astype(np.int64)
        if rank != 0 or probs_all is None:
            return
        from sklearn.metrics import f1_score, precision_score, recall_score
        for threshold in np.arange(0.1, 1.0, 0.1):
            pred = probs_all >= threshold
            p, r, f1 = precision_score(labels_all, pred, zero_division=0), recall_score(labels_all, pred, zero_division=0), f1_score(labels_all, pred, zero_division=0)
            logger.info(f"[{tag}] threshold={threshold}, precision={p}, recall={r}, f1={f1}")

    if args.do_train:
        train_ds = datasets["train"]
        per_rank = args.train_batch_size // world
        if per_rank * world != args.train_batch_size:
            raise ValueError(f"--train_batch_size {args.train_batch_size} must be divisible by the world size {world}")
        opt = (ArenaAdamW if is_hip else TorchAdamW)(model, args.learning_rate, args.adam_epsilon, args.weight_decay)
        steps_per_epoch = int(len(train_ds) / args.train_batch_size / args.gradient_accumulation_steps)
        total = steps_per_epoch * (args.num_train_epochs - args.start_epoch)
        warm = int(total * args.warmup_proportion)
        reducer = None
        if is_hip:
            iadist.broadcast_arena(model.param_arena)
            reducer = iadist.GradBucketReducer.for_arena(model.param_arena)
            Fn.clear_grad_ready_hooks()
            Fn.register_grad_ready_hook(reducer.grads_ready)
        if rank == 0:
            with open(os.path.join(out_dir, "hyperparamter.txt"), "w") as f:
                print(args, file=f)
                print("\n", file=f)
                print(getattr(model, "config", None), file=f)
        logger.info("***** Running training *****")
        logger.info("  Num examples = %d", len(train_ds))
        logger.info("  Batch size = %d (x %d ranks of %d)", args.train_batch_size, world, per_rank)
        logger.info("  Num steps = %d", total)
        global_step = 0
        for epoch in range(int(args.start_epoch), int(args.num_train_epochs)):
            model.train()
            idx = iadist.shard_indices(len(train_ds), rank, world, args.seed + epoch, shuffle=True)
            # augmentation draws (data/transforms.py: the global `random`): a different stream per epoch and per rank, in the main
            # process (num_workers 0) and, through the loader's base seed, in every worker (base_seed + worker_id)
            aug_seed = (args.seed * 1000003 + epoch * 1009 + rank * 7919) & 0x7FFFFFFF
            random.seed(aug_seed)
            loader = DataLoader(Subset(train_ds, idx.tolist()), batch_size=per_rank, shuffle=False, collate_fn=collate_fn,
                                generator=torch.Generator().manual_seed(aug_seed), **dl_kw)
            opt.zero_grad()
            for step, batch in enumerate(loader):
                b = to_dev(batch[2:])
                # dropout streams differ per rank (every rank holds other samples of the global batch: identical masks would correlate them)
                Fn.set_step_seed((args.seed * 1000003 + global_step * 131 + step + rank * 0x9E3779B1) & 0xFFFFFFFF)
                out = call_model(model, b)
                loss = out.loss
                if step % args.log_steps == 0:
                    logger.info(f"[Epoch-{epoch} Step-{step}] loss: {loss}")
                if args.gradient_accumulation_steps > 1:
                    loss = loss / args.gradient_accumulation_steps
                if reducer is not None:       # collectives only start in the backward that completes the accumulated gradient
                    reducer.armed = (step + 1) % args.gradient_accumulation_steps == 0
                loss.backward()
                if (step + 1) % args.gradient_accumulation_steps == 0:
                    if reducer is None and world > 1:     # plain torch models (TextCNN): average the gradients over the ranks
                        iadist.all_reduce_grads(model, world)
                    scale = reducer.finish() if reducer is not None else 1.0
                    opt.step(linear_schedule_with_warmup(global_step, warm, total), grad_scale=scale)
                    opt.zero_grad()
                    global_step += 1
            if args.do_eval and datasets.get("valid") is not None:
                logger.info(f"[Epoch-{epoch}] Starting evaluation ...")
                evaluate(datasets["valid"], f"Epoch-{epoch}")
            if rank == 0:
                logger.info(f"[Epoch-{epoch}] saving model")
                torch.save(model.state_dict(), os.path.join(out_dir, f"{checkpoint_name}_epoch-{epoch}.bin"))
    elif args.do_eval and datasets.get("valid") is not None:
        evaluate(datasets["valid"], "Eval")

    if args.do_pred and datasets.get("test") is not None and rank == 0:
        model.eval()
        head = model.classifier.out_proj
        json.dump({"w": head.weight.detach().cpu().numpy().tolist(), "b": head.bias.detach().cpu().numpy().tolist()},
                  open(os.path.join(out_dir, "weights.json"), "w", encoding="utf-8"), ensure_ascii=False)
        loader = DataLoader(datasets["test"], batch_size=args.eval_batch_size, shuffle=False, collate_fn=collate_fn, **dl_kw)
        with open(os.path.join(out_dir, f"deepAI_result_{getattr(args, 'pred_tag', '')}threshold={args.threshold}.jsonl"), "w", encoding="utf-8") as w, torch.no_grad():
            for step, batch in enumerate(loader):
                src_ids, tgt_ids = batch[:2]
                out = call_model(model, to_dev(batch[2:]))
                se, te = out.src_embeds.float().cpu().numpy(), out.tgt_embeds.float().cpu().numpy()
                for sid, tid, s, t in zip(src_ids, tgt_ids, se, te):
                    s = ",".join(str(x) for x in s) if isinstance(s, np.ndarray) else str(s)
                    t = ",".join(str(x) for x in t) if isinstance(t, np.ndarray) else str(t)
                    w.write(json.dumps({"src_item_id": sid, "src_item_emb": f"[{s}]", "tgt_item_id": tid, "tgt_item_emb": f"[{t}]",
                                        "threshold": args.threshold}) + "\n")
                if args.log_steps is not None and step % args.log_steps == 0:
                    logger.info(f"[Prediction] {step} samples processed")
        logger.info("[Prediction] Finished")
    return out_dir
