#!/usr/bin/env python3
"""Headline benchmark: item-pairs/sec of the train step (forward + loss + backward + gradient all-reduce +
fused AdamW) of the CoCa two-tower model roberta_large + ViT-B/16 @384, seq 50+205, bf16 activations /
fp32 master weights, on synthetic pairs (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One JSON line on rank 0.  `value` = pairs processed by ALL ranks / max-over-ranks wall time of the K timed
steps (inputs resident in HBM, weak scaling: per-GPU batch fixed).  The batches are drawn without replacement from the
50 000-pair synthetic set of SURVEY §8(d) (16 resident global batches; longer runs walk them again).  Extra objects:
  roofline     the dominant kernel (by summed time: the wgrad GEMM instantiation), timed per launch with HIP
               events on its launch stream: achieved = algorithmic FLOPs / time.  The headline region runs the image tower on a
               second HIP stream (round 5 default, +1.5-2 % pairs/s), where a kernel's event interval also contains whatever the
               other stream's kernels took from it -- so the event-timed leg is taken in a SEPARATE short single-stream pass of
               the same step right after the timed region (`roofline.measured_in` says so; `frac_in_timed_region` is the
               two-stream figure for comparison; --single-stream runs everything on one stream as in rounds 1-4).  `traffic` = HBM
               bytes per launch from two rocprofv3 --pmc child passes of this same command, single-stream (FETCH_SIZE x 2 on
               gfx950 + WRITE_SIZE, MI355X_MICROARCH.md HBM section), N = 1 only, null when rocprofv3 is unavailable.
  variants     the same step with every sequence at the full 255 tokens, and with the unpadded text towers (--unpad,
               DESIGN.md 4.8), a few steps each after the headline measurement (not part of `value`).
  cpu_baseline the CPU oracle (oracle/ref_models.py, a pure-torch port pinned to the reference by golden
               vectors) running the same train step on the host cores, bounded sample, rank 0 at N=1 only.
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_FLOPS_PER_PAIR = 1.628e12      # 3 x (2 x text(255) + 2 x ViT-B/16@384) forward FLOPs
WGRAD_VARIANT = 1101                 # gemm_kernel<A k-strided, B k-strided, EPI_NONE, fp32 out>  (dW = dY^T X)
WGRAD_KERNEL = "t256w::gemm_kernel<true, true, 0, true> (+ splitk_reduce_kernel)"
DATASET_PAIRS = 50_000               # size of the synthetic pair set the batches are drawn from
RESIDENT_BATCHES = 16                # global batches kept in HBM (4096 pairs at the default 256 per GPU; 0.9 GB each per rank)


def pmc_traffic(args):
    """HBM bytes per launch of the wgrad kernel, measured by running this same command (2 steps) under rocprofv3 --pmc, one pass
    per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).  FETCH_SIZE counts 64 B per 128 B request of a wide
    coalesced read on this part, so it is doubled (MI355X_MICROARCH.md, HBM).  Returns (bytes, note)."""
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None, "rocprofv3 not found"
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="ia_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        # the same workload as the parent (--unpad / --full-length are forwarded), 1 + 1 steps, two resident batches only
        cmd = [rocprof, "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
               "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--no-variants", "--pairs-per-gpu", str(args.pairs_per_gpu),
               "--image-model", args.image_model, "--seed", str(args.seed), "--resident-batches", "2", "--single-stream"]
        cmd += (["--unpad"] if args.unpad else []) + (["--full-length"] if args.full_length else [])
        try:
            # a plain single-process child even when this run was started by a launcher (no inherited rendezvous)
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                     "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
            # own session: on a timeout the whole group goes (rocprofv3 is a wrapper; killing only it would leave the python child
            # holding the GPU)
            errlog = open(os.path.join(out, "child.err"), "wb")
            child = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=errlog, cwd=out, env=env, start_new_session=True)
            import signal
            # the child lives in its own session: if THIS process is told to stop while a pass runs (driver timeout, Ctrl-C), take the
            # child's group down first -- it holds its own model in HBM
            def on_term(signum, frame, child=child):
                kill_group(child)
                raise SystemExit(128 + signum)
            old = {sg: signal.signal(sg, on_term) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
            try:
                rc = child.wait(timeout=args.pmc_timeout)
            except subprocess.TimeoutExpired:
                kill_group(child)
                return None, f"{counter} pass timed out after {args.pmc_timeout} s"
            except BaseException:
                kill_group(child)
                raise
            finally:
                for sg, h in old.items():
                    signal.signal(sg, h)
            errlog.close()
            # (rocprofv3 of ROCm 7.2 can return 1 after a complete run whose counter file is whole: the file decides, not the code)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            tot, n = 0.0, 0
            for r in (csv.DictReader(open(files[0])) if files else ()):
                if "gemm_kernel<true, true, 0, true>" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    tot += float(r["Counter_Value"]); n += 1
            if n == 0:
                # the child's own last words sit in front of the profiler's closing lines
                lines = [ln for ln in open(os.path.join(out, "child.err"), "rb").read().decode("utf-8", "replace").splitlines()
                         if "rocprofv3" not in ln and "rocprofiler" not in ln and ln.strip()]
                return None, f"{counter} pass exited with {rc}, kernel not in the trace: " + " | ".join(lines[-6:])[-900:]
            kb[counter] = tot / n
        except Exception as e:
            return None, f"{counter} pass failed: {e!r}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return int((2 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024), (f"rocprofv3 --pmc, mean per launch: FETCH_SIZE {kb['FETCH_SIZE']:.0f} KB x 2 "
                                                                    f"+ WRITE_SIZE {kb['WRITE_SIZE']:.0f} KB")


def roberta_large_config(**over):
    from types import SimpleNamespace
    cfg = SimpleNamespace(
        hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, vocab_size=21128,
        max_position_embeddings=512, type_vocab_size=2, pad_token_id=0, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
        layer_norm_eps=1e-12, hidden_act="gelu", initializer_range=0.02, num_labels=2,
        interaction_type="two_tower", classification_method="cls", similarity_measure="NA", loss_type="ce", max_seq_len=50,
        max_seq_len_pv=205, loss_margin=1.0, cls_layers="1", cls_pool="cat", ensemble="sum", image_hidden_size=3072, image_size=384,
        classifier_dropout=None, auxiliary_task=False)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def build_model(cfg, image_model="vit_base_patch16_384", seed=2345):
    import item_alignment_amd.models as M
    torch.manual_seed(seed)
    text = M.RobertaModel(cfg)
    vit = M.create_model(image_model)
    return M.CoCaForItemAlignment(cfg, vit, text)


def linear_schedule(step, total, warmup):
    """get_linear_schedule_with_warmup (reference finetune_multimodal.py:315)."""
    if step < warmup:
        return step / max(1, warmup)
    return max(0.0, (total - step) / max(1, total - warmup))


def cpu_baseline(cfg, image_model, pairs, steps, seed):
    """Same train step on the host cores with the CPU oracle: fp32, dropout on, AdamW."""
    from types import SimpleNamespace
    from oracle import ref_models as O
    from item_alignment_amd.data.synthetic import SyntheticCocaPairs
    from item_alignment_amd.models.image import VIT_CONFIGS
    # torch's intra-op pool stops scaling (and can thrash) far below the 256 hardware threads of the GPU box's host:
    # use at most 64 threads and report exactly that many as `cores`
    threads = min(os.cpu_count() or 1, int(os.environ.get("IA_CPU_BASELINE_THREADS", "64")))
    torch.set_num_threads(threads)
    model = build_model(cfg, image_model, seed)
    sd = {k: v.detach().clone().float().requires_grad_(v.requires_grad) for k, v in model.named_parameters()}
    names = [k for k, v in sd.items() if v.requires_grad]
    s, p, d, depth, h = VIT_CONFIGS[image_model]
    vcfg = SimpleNamespace(embed_dim=d, depth=depth, num_heads=h, patch_size=p, eps=1e-6)
    data = SyntheticCocaPairs(max(pairs * (steps + 1), pairs), image_size=cfg.image_size, seed=seed)
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    decay = [not any(nd in k for nd in ("bias", "LayerNorm.weight")) for k in names]
    times = []
    for it in range(steps + 1):
        b = data.batch(list(range(it * pairs, (it + 1) * pairs)), "cpu")
        t0 = time.perf_counter()
        out = O.coca_item_alignment(sd, cfg, vcfg, *b[:10], labels=b[10], training=True)
        grads = torch.autograd.grad(out.loss, [sd[k] for k in names], allow_unused=True)
        with torch.no_grad():
            O.adamw_step([sd[k] for k in names], [g if g is not None else torch.zeros_like(sd[k]) for g, k in zip(grads, names)],
                         m, v, it + 1, 1e-5, wd=1e-5, decay_mask=decay)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / max(1, len(times) - 1)
    return {"value": pairs / t, "unit": "item-pairs/sec", "cores": threads, "kind": "port",
            "sample": f"{steps} timed + 1 warm-up train steps of {pairs} pairs (same model/config, fp32, dropout on, oracle/ref_models.py on "
                      f"{torch.get_num_threads()} torch threads)"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher environment: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child process (never an exec: see the harness rules on processes that have initialised the GPU; this parent has not, but a
    child keeps that true by construction) on a free loopback port, and pass its stdout -- rank 0's one JSON line -- through."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    import signal
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    # the launcher and its N GPU ranks live in their own session (one killpg reaches all of them): if THIS process is told to stop
    # (driver timeout, Ctrl-C, hang-up) nothing else would -- take the group down first, then leave with the conventional code
    def on_term(signum, frame):
        kill_group(child)
        raise SystemExit(128 + signum)
    old = {sg: signal.signal(sg, on_term) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    try:
        return child.wait()
    except BaseException:
        kill_group(child)
        raise
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)


def kill_group(child):
    import signal
    try:
        os.killpg(child.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass
    try:
        child.wait(timeout=10)
    except Exception:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    # 256 pairs per GPU = 211 GiB of the 288 GB at the peak of a default run (activations of both towers for backward, 13 resident
    # batches, weights + AdamW state); per pair 2-3 % faster than 128 (the step's fixed costs -- optimizer, embedding backward, reduces --
    # and the GEMMs' tile-round tails amortise over twice the rows): 535.9 / 545.8 / 550.6 pairs/s at 128 / 192 / 256 on one box
    ap.add_argument("--pairs-per-gpu", type=int, default=256)
    ap.add_argument("--image-model", default="vit_base_patch16_384")
    ap.add_argument("--seed", type=int, default=2345)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-oracle steps (BASELINE.md 3: 3 timed + 1 warm-up)")
    ap.add_argument("--single-stream", action="store_true", help="both towers on one HIP stream (the round 1-4 headline); default: the "
                    "image tower on a second stream, the roofline leg in a separate single-stream pass")
    ap.add_argument("--roofline-steps", type=int, default=3, help="steps of the separate single-stream pass behind `roofline`")
    ap.add_argument("--full-length", action="store_true", help="all sequences at the full 255 tokens (mask all ones)")
    ap.add_argument("--unpad", action="store_true", help="text towers on the valid tokens only (IA_UNPAD=1, DESIGN.md 4.8); off by "
                    "default: the reference computes densely on the padding and the headline number keeps that workload")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes behind roofline.traffic")
    ap.add_argument("--no-variants", action="store_true", help="skip the full-length / unpadded side measurements")
    ap.add_argument("--variant-steps", type=int, default=4)
    ap.add_argument("--pmc-timeout", type=int, default=240, help="seconds per rocprofv3 --pmc child pass")
    ap.add_argument("--resident-batches", type=int, default=RESIDENT_BATCHES, help="global batches generated up front and kept in HBM")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N`): this process has not touched the GPU (no torch.cuda call so far) and never will --
        # it starts the N ranks as a CHILD launcher, forwards rank 0's single JSON line and returns the launcher's exit code
        raise SystemExit(self_launch(args.gpus))
    if args.unpad:
        os.environ["IA_UNPAD"] = "1"
    # stdout carries exactly ONE line, the result JSON of rank 0.  Native libraries write there too (RCCL prints a five-line version
    # banner when the pool exports NCCL_DEBUG=VERSION): everything written to fd 1 during the run goes to stderr, the result line to the
    # saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    from item_alignment_amd import _lib
    from item_alignment_amd import dist as iadist
    from item_alignment_amd.data.synthetic import SyntheticCocaPairs
    from item_alignment_amd.models import functional as Fn
    from item_alignment_amd.models import multimodal as _mm

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    lib = _lib.load()
    rank, world, local = iadist.init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # one process per GPU; IA_DP_BACKEND (tests: two gloo ranks sharing one GPU) folds surplus local ranks onto the visible devices,
    # exactly as dist.init_from_env does
    dev_index = local % max(1, torch.cuda.device_count()) if iadist.BACKEND else local
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    cfg = roberta_large_config()
    model = build_model(cfg, args.image_model, args.seed).to(dev).train()
    arena = model.param_arena
    iadist.broadcast_arena(arena)
    reducer = iadist.GradBucketReducer.for_arena(arena)
    Fn.clear_grad_ready_hooks()
    Fn.register_grad_ready_hook(reducer.grads_ready)

    B = args.pairs_per_gpu
    n_steps = args.warmup + args.steps
    # a 50 000-pair synthetic set; the run walks a seeded permutation of it in global batches of B * world pairs,
    # RESIDENT_BATCHES of them generated device-side and resident in HBM before timing starts (0.9 GB per 256-pair batch);
    # rank r owns pairs r::world of each global batch.  A run of more than RESIDENT_BATCHES steps (warm-up included) walks the resident
    # batches again from the start (13 steps by default: no repeat; the driver's 3 + 20: batches 0-6 are seen twice).
    data = SyntheticCocaPairs(DATASET_PAIRS, image_size=cfg.image_size, seed=args.seed, full_length=args.full_length)
    perm = torch.randperm(DATASET_PAIRS, generator=torch.Generator().manual_seed(args.seed)).tolist()
    n_batches = max(1, min(args.resident_batches, n_steps, DATASET_PAIRS // (B * world)))

    def make_batches(ds, count):
        return [ds.batch([perm[g * B * world + rank + world * i] for i in range(B)], dev, device_images=True) for g in range(count)]
    batches = make_batches(data, n_batches)
    total_opt_steps = max(n_steps, 100)
    warm = int(0.3 * total_opt_steps)
    base_lr = 1e-5

    def step(i, pool=None):
        pool = batches if pool is None else pool
        Fn.set_step_seed((args.seed * 1000003 + i + rank * 0x9E3779B1) & 0xFFFFFFFF)    # dropout streams differ per rank
        arena.zero_grad()
        b = pool[i % len(pool)]
        out = model(*b[:10], labels=b[10])
        out.loss.backward()
        scale = reducer.finish()
        arena.adamw_step(base_lr * linear_schedule(i, total_opt_steps, warm), betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-5,
                         grad_scale=scale)
        return out.loss

    def barrier():
        if world > 1 or iadist.FORCE:
            torch.distributed.barrier()

    two_streams = not args.single_stream
    _mm.TOWER_STREAMS = two_streams
    for i in range(args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    _lib.check(lib.ia_prof_begin(WGRAD_VARIANT, 400 * args.steps), "ia_prof_begin")
    t0 = time.perf_counter()
    for i in range(args.warmup, n_steps):
        loss = step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, fl, nl = C.c_double(), C.c_double(), C.c_int()
    _lib.check(lib.ia_prof_end(C.byref(ms), C.byref(fl), C.byref(nl)), "ia_prof_end")
    alg_bytes = lib.ia_prof_bytes()
    in_region = (ms.value, fl.value, nl.value)
    final_loss_region = float(loss.detach())
    if two_streams:
        # the roofline leg: the same step, both towers on ONE stream, so that an event interval around a launch holds that kernel alone
        _mm.TOWER_STREAMS = False
        try:
            step(n_steps)                                  # one untimed step: the allocator settles into the single-stream pattern
            torch.cuda.synchronize()
            _lib.check(lib.ia_prof_begin(WGRAD_VARIANT, 400 * args.roofline_steps), "ia_prof_begin")
            for i in range(args.roofline_steps):
                step(n_steps + 1 + i)
            torch.cuda.synchronize()
            _lib.check(lib.ia_prof_end(C.byref(ms), C.byref(fl), C.byref(nl)), "ia_prof_end")
            alg_bytes = lib.ia_prof_bytes()
        finally:
            _mm.TOWER_STREAMS = True
    n_steps_done = n_steps + (1 + args.roofline_steps if two_streams else 0)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1 or iadist.FORCE:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = tmax.item()
    final_loss = final_loss_region
    n_steps = n_steps_done                                 # the side measurements continue the step counter (seeds, schedule)

    def timed(pool, k, first, pairs_per_rank=None):
        """k more steps on `pool` (2 untimed first), max over ranks, like the headline region"""
        pairs_per_rank = B if pairs_per_rank is None else pairs_per_rank
        for i in range(2):
            step(first + i, pool)
        torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(k):
            step(first + 2 + i, pool)
        torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
        tm = torch.tensor([time.perf_counter() - t], device=dev, dtype=torch.float64)
        if world > 1 or iadist.FORCE:
            torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
        return {"value": pairs_per_rank * world * k / tm.item(), "unit": "item-pairs/sec", "ms_per_step": tm.item() / k * 1e3, "steps": k,
                "pairs_per_gpu": pairs_per_rank}

    variants = None
    if not args.no_variants and not args.unpad and not args.full_length:
        from item_alignment_amd.models import text as _text
        variants = {}
        # SURVEY 8(d)'s smallest batch: 16 pairs per GPU = a step of ~30 ms, where the 1.64 GB gradient all-reduce is a third of the
        # step instead of 2 % -- the point of the scaling curve that actually exercises the bucket overlap
        if B > 16:
            small = [data.batch([perm[g * 16 * world + rank + world * i] for i in range(16)], dev, device_images=True) for g in range(4)]
            variants["pairs_per_gpu_16"] = timed(small, 4 * args.variant_steps, n_steps, pairs_per_rank=16)
            del small
        full = SyntheticCocaPairs(B * world * 2, image_size=cfg.image_size, seed=args.seed + 1, full_length=True)
        pool = [full.batch([g * B * world + rank + world * i for i in range(B)], dev, device_images=True) for g in range(2)]
        variants["full_length_sequences"] = timed(pool, args.variant_steps, n_steps)
        del pool
        _text.UNPAD = True
        try:
            variants["unpadded_text_tower"] = timed(batches, args.variant_steps, n_steps + 2 + args.variant_steps)
        finally:
            _text.UNPAD = False
        # evaluation / prediction throughput (reference finetune_multimodal.py:470-563 runs the model under no_grad): forward-only
        # layers (ia_layer_fwd_infer: nothing written for a backward pass), same batch size
        def eval_pass(k):
            model.eval()
            try:
                with torch.no_grad():
                    for i in range(2):
                        model(*batches[i % len(batches)][:10], labels=batches[i % len(batches)][10])
                    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
                    t = time.perf_counter()
                    for i in range(k):
                        b = batches[i % len(batches)]
                        model(*b[:10], labels=b[10])
                    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
                    tm = torch.tensor([time.perf_counter() - t], device=dev, dtype=torch.float64)
            finally:
                model.train()
            if world > 1 or iadist.FORCE:
                torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
            return {"value": B * world * k / tm.item(), "unit": "item-pairs/sec", "ms_per_step": tm.item() / k * 1e3, "steps": k, "pairs_per_gpu": B,
                    "what": "forward + loss only (model.eval(), no_grad)"}
        variants["eval_forward_only"] = eval_pass(args.variant_steps)
        # the other stream arrangement than the headline's (one stream: the round 1-4 headline; two: this round's), same steps
        _mm.TOWER_STREAMS = not two_streams
        try:
            variants["single_stream" if two_streams else "image_tower_on_second_stream"] = timed(
                batches, args.variant_steps, n_steps + 4 + 2 * args.variant_steps)
        finally:
            _mm.TOWER_STREAMS = two_streams

    if rank == 0:
        pairs = B * world * args.steps
        achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # which roof bounds the kernel: its algorithmic intensity (FLOP per byte, each operand moved once) against the ridge of the
        # two peaks (2500 TFLOP/s dense bf16 MFMA, 8 TB/s HBM: 312.5 FLOP/B)
        intensity = fl.value / alg_bytes if alg_bytes > 0 else float("inf")
        bound = "mfma" if intensity >= 2500e12 / 8e12 else "hbm"
        res = {
            "metric": "item-pairs/sec (train step) RoBERTa-large+ViT-B two-tower", "value": pairs / dt, "unit": "item-pairs/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "coca(roberta_large + vit_base_patch16_384) two_tower cls/ce, ensemble=sum, seq 50+205, img 384, "
                                   "dropout 0.1, fused AdamW; train step = fwd+bwd+allreduce+optimizer",
                       "pairs_per_gpu": B, "global_batch": B * world, "seq_len": 255, "image_size": cfg.image_size,
                       "parallelism": f"dp{world}", "full_length_sequences": bool(args.full_length), "unpadded_text_tower": bool(args.unpad),
                       "deviation": "Linear(768->1024) on the image CLS (the named pairing is dimensionally inconsistent in the reference, SURVEY N1)"},
            # dense FLOP count of the padded workload; with --unpad the padded rows are not computed, so it does not apply
            "model_tflops_per_gpu": None if args.unpad else pairs / dt * TRAIN_FLOPS_PER_PAIR / 1e12 / world,
            "mfma_fraction_whole_step": None if args.unpad else pairs / dt * TRAIN_FLOPS_PER_PAIR / 1e12 / world / 2500.0,
            # the strictly dense figure: the same FLOP count over the run in which every sequence holds 255 tokens, so that no attention
            # kernel skips a padded key tile (the default batch's ragged lengths let them skip ~3 % of the attention work)
            "mfma_fraction_dense": (variants["full_length_sequences"]["value"] * TRAIN_FLOPS_PER_PAIR / 1e12 / world / 2500.0
                                    if variants and "full_length_sequences" in variants else
                                    (pairs / dt * TRAIN_FLOPS_PER_PAIR / 1e12 / world / 2500.0 if args.full_length else None)),
            "tower_streams": 2 if two_streams else 1,
            "final_loss": final_loss,
            "roofline": ({"bound": "mfma", "kernel": WGRAD_KERNEL, "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s",
                          "frac": achieved / 2500.0} if bound == "mfma" else
                         {"bound": "hbm", "kernel": WGRAD_KERNEL, "achieved": alg_bytes / (ms.value * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                          "frac": alg_bytes / (ms.value * 1e-3) / 1e9 / 8000.0}),
            "variants": variants,
        }
        res["roofline"].update({"traffic": None, "launches": nl.value, "avg_launch_us": ms.value * 1e3 / max(1, nl.value),
                                "flop_per_byte": intensity, "algorithmic_bytes_per_launch": alg_bytes / max(1, nl.value),
                                "measured_in": (f"separate single-stream pass of {args.roofline_steps} steps right after the timed region (HIP "
                                                "events on the launch stream; the timed region overlaps the towers on two streams)"
                                                if two_streams else "the timed region (HIP events on the launch stream)"),
                                "frac_in_timed_region": (in_region[1] / (in_region[0] * 1e-3) / 1e12 / 2500.0 if in_region[0] > 0 else None)})
        # Text attention: the kernels skip key tiles that hold no attendable key (right-padded batches), so a per-kernel MFMA fraction taken
        # from the DENSE 255 x 255 FLOP count flatters them.  From the attention masks of the timed batches: the share of
        # the dense key positions each kernel actually visits -- forward (attn_fwd3: 64-key tiles per sequence) and fused backward
        # (attn_bwd_fused: 32-key blocks per sequence); executed FLOPs per launch = dense FLOPs x this share.
        try:
            lens = torch.cat([torch.cat((b[1], b[6])).sum(1) for b in batches]).float()      # attention_mask_1 | attention_mask_2: tokens per sequence
            L = batches[0][1].shape[1]
            res["attention_text"] = {
                "mean_tokens": round(lens.mean().item(), 1), "padded_len": L,
                "executed_key_share_fwd": round((torch.ceil(lens / 64) * 64).clamp(max=L).mean().item() / L, 4),
                "executed_key_share_bwd": round((torch.ceil(lens / 32) * 32).clamp(max=L).mean().item() / L, 4),
                # the fused backward also leaves out 32-query blocks of padding (ia_layer_cfg::masked_rows_dead): live keys x live queries
                "executed_share_bwd_keys_x_queries": round((((torch.ceil(lens / 32) * 32).clamp(max=L) / L) ** 2).mean().item(), 4),
                "dense_flops_per_launch_fwd": 4.0 * 2 * B * 16 * L * L * 64, "dense_flops_per_launch_bwd": 10.0 * 2 * B * 16 * L * L * 64,
                "note": "per-kernel TFLOP/s on EXECUTED work = dense_flops_per_launch x executed_key_share / the kernel's launch time "
                        "(profiles/*_kernel_stats_summary.txt); the ViT kernels (no padding mask) execute 19 of their 20 32-key blocks"}
        except Exception as e:      # (a diagnostic: never fail the line over it)
            res["attention_text"] = {"error": repr(e)}
        res["config"]["dataset_pairs"] = DATASET_PAIRS
        res["config"]["resident_batches"] = n_batches
        res["peak_hbm_gib"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)      # this rank, incl. the resident batches
        if world == 1 and not args.no_pmc:
            # free this process's HBM first: the child runs build their own model and batches
            del batches
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            held = torch.cuda.memory_reserved() / 2 ** 30
            traffic, note = pmc_traffic(args)
            if traffic is None:
                note += f" (this process still held {held:.1f} GiB of HBM while the pass ran)"
            res["roofline"]["traffic"] = traffic
            res["roofline"]["traffic_source"] = note
            if traffic:
                res["roofline"]["traffic_GBps"] = traffic / (ms.value * 1e-3 / max(1, nl.value)) / 1e9
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(cfg, args.image_model, args.cpu_pairs, args.cpu_steps, args.seed)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "item-pairs/sec", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    os.close(result_fd)


if __name__ == "__main__":
    main()
