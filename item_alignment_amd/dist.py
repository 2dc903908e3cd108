"""Data parallelism for the pair batch: one process per GPU, gradients all-reduced with RCCL over xGMI
(torch.distributed backend "nccl" is RCCL on ROCm), overlapped with backward.

The reference is single-process / single-GPU (no torch.distributed anywhere), so this layer
has no reference implementation to match — only the single-GPU maths: the reference loss is a mean over
the batch (nn.CrossEntropyLoss default, text.py:1292), so equal shard sizes + averaged gradients reproduce
the global-batch gradient.

Design (MI355X-first): gradients already live in one flat fp32 arena (arena.py), so a bucket is a slice of
it — no packing copies.  Backward fills the arena from its end; as soon as every parameter overlapping a
bucket has been reported FINAL (models.functional.register_grad_ready_hook, called right after the layer's
kernels were enqueued; final = no other node of this backward adds to those gradients, which the encoder stacks
establish by counting their uses) the bucket's all-reduce is issued asynchronously: RCCL's stream waits for exactly
the kernels enqueued so far and the remaining backward keeps the compute stream busy.  The 1/world
averaging is folded into the optimiser's grad_scale, so no extra pass touches the gradients.
xGMI is point-to-point (7 links/GPU); large buckets (default 128 MiB) keep per-link rings bandwidth-bound.
"""
import datetime
import os

import numpy as np
import torch
import torch.distributed as dist

FORCE = os.environ.get("IA_DP_FORCE_COLLECTIVES") == "1"
# IA_DP_BF16=1: buckets travel as bf16 (half the xGMI bytes: 0.82 instead of 1.64 GB per step for the 411 M-parameter headline
# model); the sum is formed in bf16 by RCCL and written back into the fp32 gradient arena.  Off by default (fp32 = the exact sum).
BF16_BUCKETS = os.environ.get("IA_DP_BF16") == "1"
# IA_DP_BACKEND overrides the process-group backend (tests run two ranks on ONE GPU over gloo; RCCL refuses duplicate devices)
BACKEND = os.environ.get("IA_DP_BACKEND")


def init_from_env(device_type="cuda"):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the launcher (torch.distributed.run).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # IA_DP_FORCE_COLLECTIVES=1: initialise RCCL and issue every bucket all-reduce even with one rank (lets a 1-GPU box
    # exercise the exact multi-GPU code path: process group, stream hand-over, async bucket launches)
    if (world > 1 or FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = BACKEND or ("nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            ndev = max(1, torch.cuda.device_count())
            if local >= ndev and not BACKEND:
                # production launches: one process per GPU.  (Tests put two ranks on one GPU over gloo: IA_DP_BACKEND.)
                raise RuntimeError(f"LOCAL_RANK {local} but only {ndev} GPU(s) visible: launch one process per GPU")
            torch.cuda.set_device(local % ndev)
        # collective timeout: long enough for a rank that waits for the others' data loading or for rank 0's checkpoint write, short
        # enough that a dead rank surfaces; IA_DP_TIMEOUT_MIN overrides it
        minutes = float(os.environ.get("IA_DP_TIMEOUT_MIN", "30"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=minutes))
        if device_type == "cuda" and os.environ.get("IA_GEMM_DYNAMIC") is None:
            # a communication stream will hold CUs next to the backward GEMMs from now on: the persistent GEMM launches hand their tiles
            # out dynamically (a workgroup that cannot start keeps no share of the work hostage; profiles/r05_cu_contention.txt)
            from . import _lib
            _lib.load().ia_debug_gemm_dynamic(1)
    return rank, world, local


def shard_indices(n_items, rank, world, epoch_seed, shuffle=True):
    """Rank r takes positions r::world of the epoch permutation drawn from the shared seed: identical global
    order at any world size; shards are truncated to equal length (the loss is a batch mean)."""
    g = torch.Generator().manual_seed(int(epoch_seed))
    perm = torch.randperm(n_items, generator=g) if shuffle else torch.arange(n_items)
    per = n_items // world
    return perm[rank:per * world:world]


class GradBucketReducer:
    """Bucketed asynchronous all-reduce over a flat gradient buffer.

    flat_grad: 1-D tensor; params: list of (param, offset, numel) placed inside it (arena order).
    """

    def __init__(self, flat_grad, params, bucket_bytes=128 << 20, group=None, bf16=None):
        self.flat = flat_grad
        self.group = group
        self.bf16 = BF16_BUCKETS if bf16 is None else bool(bf16)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        esz = flat_grad.element_size()
        # bucket boundaries on multiples of ALIGN elements: the bf16 staging casts (ia_cast_f32_to_bf16 / ia_cast_bf16_to_f32) move 16-byte
        # vectors from flat[s:], so s * 4 and s * 2 bytes must both be 16-byte aligned (only the arena's end may be ragged)
        per = max(self.ALIGN, bucket_bytes // esz // self.ALIGN * self.ALIGN)
        n = flat_grad.numel()
        # buckets from the END of the arena (backward order) towards the start
        self.buckets, end = [], n
        while end > 0:
            start = max(0, (end - per) // self.ALIGN * self.ALIGN)
            self.buckets.append((start, end))
            end = start
        assert all(s % self.ALIGN == 0 for s, _ in self.buckets)
        self.stage = [None] * len(self.buckets)          # persistent bf16 staging slices (allocated on first use, bf16 buckets only)
        self.param_buckets = {}
        self.need = [0] * len(self.buckets)
        for p, off, numel in params:
            if not getattr(p, "requires_grad", True):
                continue
            hit = [i for i, (s, e) in enumerate(self.buckets) if off < e and off + numel > s]
            self.param_buckets[id(p)] = hit
            for i in hit:
                self.need[i] += 1
        self.reset()

    @classmethod
    def for_arena(cls, arena, **kw):
        r = cls(arena.grad, [(p, o, p.numel()) for p, o in zip(arena.params, arena.offsets)], **kw)
        r.arena = arena
        r.main_stream = torch.cuda.current_stream() if arena.grad.is_cuda else None
        return r

    def streams(self):
        """the stream the reducer was created on + the side streams the model registered in its arena"""
        a = getattr(self, "arena", None)
        main = getattr(self, "main_stream", None)
        return ([main] if main is not None else []) + (list(a.side_streams) if a is not None else [])

    # gradient accumulation: the arena sums micro-batches locally; only the LAST micro-step's backward may start collectives
    # (a bucket reduced early would be reduced with part of its sum and never again).  The train loop disarms the reducer for
    # the other micro-steps; finish() then reduces every bucket that was not launched.
    armed = True
    ALIGN = 8

    def reset(self):
        self.left = list(self.need)
        self.launched = [False] * len(self.buckets)
        self.seen = set()
        self.works = []

    def _launch(self, i):
        s, e = self.buckets[i]
        self.launched[i] = True
        if self.world > 1 or (FORCE and dist.is_initialized()):
            # Gradients of one bucket may have been produced on several HIP streams (models that run independent towers concurrently), so
            # the collective must be ordered behind all of them -- without making any of THEM wait for the others: the collective (and
            # its bf16 staging cast) is issued on a stream of its own that waits for the compute streams, and RCCL orders itself
            # behind that stream.  (Rounds 2-4 made the launching compute stream wait for the others: with the image tower on a second
            # stream the rest of that tower's backward then queued behind the whole text backward.)
            import contextlib
            ctx = contextlib.nullcontext()
            if self.flat.is_cuda:
                if getattr(self, "comm_stream", None) is None:
                    self.comm_stream = torch.cuda.Stream(device=self.flat.device)
                cur = torch.cuda.current_stream()
                self.comm_stream.wait_stream(cur)
                for st in self.streams():
                    if st != cur:
                        self.comm_stream.wait_stream(st)
                ctx = torch.cuda.stream(self.comm_stream)
            with ctx:
                if self.bf16:
                    # one pass fp32 -> bf16 into a staging slice (ia_cast_f32_to_bf16; torch's .to() only on CPU tensors, i.e. in the gloo tests)
                    if self.stage[i] is None:
                        self.stage[i] = torch.empty(e - s, dtype=torch.bfloat16, device=self.flat.device)
                    stage = self.stage[i]
                    if self.flat.is_cuda:
                        from . import ops
                        ops.cast_to_bf16(self.flat[s:e], stage)
                    else:
                        stage.copy_(self.flat[s:e])
                    self.works.append((dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group, async_op=True), stage, s, e))
                else:
                    self.works.append((dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, s, e))

    def grads_ready(self, params, final=False):
        """Hook target: these parameters' gradient kernels have been enqueued on the current stream.  Only a FINAL report
        (no other autograd node of this backward adds to the gradient: models.functional._notify) counts towards launching a
        bucket early; everything else is reduced by finish().  A module applied twice, the two sides of the KG embedding or a
        chunked conv tower report the same parameter from several nodes: treating the first report as final would all-reduce a
        partial sum and silently drop the rest."""
        if not self.armed or not final:
            return
        for p in params:
            k = id(p)
            if k in self.seen or k not in self.param_buckets:
                continue
            self.seen.add(k)
            for i in self.param_buckets[k]:
                self.left[i] -= 1
                if self.left[i] == 0 and not self.launched[i]:
                    self._launch(i)

    def finish(self):
        """Call after backward: reduce whatever is still pending (frozen / unreached parameters), then make the
        compute stream wait for every bucket.  Returns the grad_scale (1/world) for the optimiser."""
        for i in range(len(self.buckets)):
            if not self.launched[i]:
                self._launch(i)
        for w, stage, s, e in self.works:
            w.wait()                                     # the current (optimiser) stream waits for the collective
            if getattr(self, "comm_stream", None) is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)      # (gloo on device tensors completes on the issuing stream)
            if stage is not None:
                if self.flat.is_cuda:
                    from . import ops
                    ops.cast_to_f32(stage, self.flat[s:e])           # one pass back into the fp32 arena slice
                else:
                    self.flat[s:e].copy_(stage)
        self.reset()
        try:
            from .models import functional as Fn
            Fn.reset_use_counts()
        except Exception:        # the reducer is also used stand-alone (tests) without the model package
            pass
        return 1.0 / self.world


def broadcast_arena(arena, src=0):
    """Identical replicas: rank `src`'s master weights (and its bf16 shadow) everywhere."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or FORCE):
        dist.broadcast(arena.master, src=src)
        arena.refresh_shadow()


def all_reduce_scalar(x, op=None):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(x, op=op or dist.ReduceOp.SUM)
    return x


def gather_rows(local, device=None):
    """Every rank hands in a float array [n_r, k] (n_r may differ from rank to rank and may be 0: the collates drop samples whose
    image failed to load, data.py:44,84); every rank gets the concatenation over ranks (rank order) back.  Two tensor collectives
    (row counts, then the rows padded to the longest shard): no pickling, and no assumption about which positions a rank scored."""
    local = np.asarray(local if local is not None else [], dtype=np.float64)
    k = local.shape[1] if local.ndim == 2 else 1
    local = local.reshape(-1, k)
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    mine = torch.tensor([local.shape[0], k], dtype=torch.int64, device=device)
    shapes = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(shapes, mine)
    counts = [int(t[0]) for t in shapes]
    k = max(int(t[1]) for t in shapes)           # an empty shard does not know the row width
    if local.shape[0] == 0:
        local = local.reshape(0, k)
    cap = max(counts)
    if cap == 0:
        return np.empty((0, k), dtype=np.float64)
    buf = torch.zeros(cap, k, dtype=torch.float64, device=device)
    buf[:local.shape[0]] = torch.from_numpy(local).to(device)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)], axis=0)


def all_reduce_grads(model, world):
    """Gradient averaging for plain torch models without a parameter arena (TextCNN, config C1)."""
    for p in model.parameters():
        if p.grad is not None:
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
            p.grad.div_(world)
