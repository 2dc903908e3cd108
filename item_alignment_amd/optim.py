"""`torch.optim.Optimizer`-shaped front of the fused arena AdamW, so that the reference's own train loop keeps its lines:

    optimizer = AdamW(optimizer_grouped_parameters, lr=args.learning_rate, eps=args.adam_epsilon, betas=(0.9, 0.98))   # :296-308
    scheduler = get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_train_optimization_steps)            # :315
    ...
    optimizer.zero_grad(); loss.backward(); optimizer.step(); scheduler.step()                                        # :376, :458-468

(reference finetune_multimodal.py; the same lines in finetune_text.py / finetune_image.py).  Parameter groups are honoured per group --
`lr`, `betas`, `eps`, `weight_decay` are read from `param_groups[i]` at every step, so `LambdaLR` / `get_linear_schedule_with_warmup`
drive the learning rate exactly as they drive torch's -- and each group is ONE `ia_adamw_flat` launch over its slice list of the flat
arenas (fp32 master, gradient, both moments; the bf16 shadow the GEMMs read is rewritten by the same launch).  The arithmetic is
torch.optim.AdamW's (decoupled decay first, bias-corrected moments, eps added to sqrt(v) / sqrt(bc2)); `state_dict()` exposes the
moments as `exp_avg` / `exp_avg_sq` views per parameter plus a `step` tensor, i.e. torch's layout.

One documented difference: a group's launch updates EVERY parameter of the group, also one whose gradient torch would report as
`None` (a head the forward never used): its gradient slice of the arena is zero, so the moments decay and the decoupled weight decay is
applied where `torch.optim.AdamW` skips the parameter.  Every model of this package uses all of its parameters in every step; freeze an
unused head with `requires_grad_(False)` (such parameters are left out of the slice tables) to get torch's behaviour.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, stream_ptr
from .arena import CHUNK


def arena_of(p):
    arena = getattr(p, "_ia_arena", None)
    if arena is None:
        raise _lib.ItemAlignError("parameter is not in a GPU parameter arena: call model.cuda() and use the model once, or "
                                  "model.ensure_arena(), before building the optimizer")
    return arena


class AdamW(torch.optim.Optimizer):
    """Drop-in for `torch.optim.AdamW(params_or_groups, lr, betas, eps, weight_decay)` over a model that lives in a ParamArena."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.grad_scale = grad_scale            # the data-parallel 1 / world factor (dist.GradBucketReducer.finish()) when gradients are summed
        self.arena = None
        self._tables = None
        for group in self.param_groups:
            group.setdefault("step_count", 0)
        # The reference builds its optimizer BEFORE model.cuda() (finetune_multimodal.py:309 against :342), when no arena exists yet:
        # binding to the arena happens at the first step() / zero_grad() that finds one.
        if all(getattr(p, "_ia_arena", None) is not None for g in self.param_groups for p in g["params"]):
            self._bind()

    def _bind(self):
        if self._tables is not None:
            return True
        arena = None
        for group in self.param_groups:
            for p in group["params"]:
                a = arena_of(p)
                if arena is None:
                    arena = a
                elif a is not arena:
                    raise ValueError("all parameters of one optimizer must live in the same arena (one model)")
        if arena is None:
            raise ValueError("optimizer got an empty parameter list")
        self.arena = arena
        offset_of = {id(p): o for p, o in zip(arena.params, arena.offsets)}
        tables, step_tensors = [], []
        for group in self.param_groups:
            chunks = []
            # torch keeps a `step` tensor per parameter; all parameters of a group step together, so they share ONE tensor per group
            # that step() updates in place (no per-parameter host work in the loop)
            step_t = torch.full((), float(group.get("step_count", 0)), dtype=torch.float32)
            step_tensors.append(step_t)
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                o, n = offset_of[id(p)], p.numel()
                for c in range(0, n, CHUNK):
                    chunks.append(((o + c) & 0xFFFFFFFF, (o + c) >> 32, min(CHUNK, n - c), 1))
                # torch's per-parameter state, as views of the arenas (never re-allocated: the kernel writes them in place)
                self.state[p] = {"step": step_t,
                                 "exp_avg": arena.exp_avg[o:o + n].view(p.shape), "exp_avg_sq": arena.exp_avg_sq[o:o + n].view(p.shape)}
            table = torch.from_numpy(np.asarray(chunks, dtype=np.uint32).reshape(-1, 4)).to(arena.device) if chunks else None
            tables.append((table, len(chunks)))
        self._tables, self._step_tensors = tables, step_tensors
        arena.optimizer_bound = True
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._bind()
        arena, lib = self.arena, _lib.load()
        arena.join_side_streams()
        for group, (table, n), step_t in zip(self.param_groups, self._tables, self._step_tensors):
            if n == 0:
                continue
            group["step_count"] += 1
            b1, b2 = group["betas"]
            check(lib.ia_adamw_flat(arena.master.data_ptr(), arena.grad.data_ptr(), arena.exp_avg.data_ptr(), arena.exp_avg_sq.data_ptr(),
                                    arena.shadow.data_ptr(), table.data_ptr(), n, float(group["lr"]), b1, b2, group["eps"],
                                    group["weight_decay"], group["step_count"], float(self.grad_scale), stream_ptr()), "ia_adamw_flat")
            step_t.fill_(float(group["step_count"]))
        arena.step_count += 1
        arena.refresh_transposed()
        return loss

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self.param_groups[-1].setdefault("step_count", 0)
        if getattr(self, "_tables", None) is not None:          # bound already: rebuild the slice tables (the moments are arena views, nothing is lost)
            self._tables = None
            self._bind()

    def zero_grad(self, set_to_none=True):
        """One memset of the gradient arena; `p.grad` stays a view of it (the kernels write gradients there whatever `p.grad` says)."""
        if self._tables is None and any(getattr(p, "_ia_arena", None) is None for g in self.param_groups for p in g["params"]):
            return super().zero_grad(set_to_none=set_to_none)        # before the model's first use: nothing to clear but torch's own
        self._bind()
        self.arena.zero_grad()
        self.arena.reattach()

    def load_state_dict(self, state_dict):
        """torch's loader replaces the state tensors; the moments must stay views of the arenas, so copy the loaded values in."""
        self._bind()
        super().load_state_dict(state_dict)
        arena = self.arena
        offset_of = {id(p): o for p, o in zip(arena.params, arena.offsets)}
        for group, step_t in zip(self.param_groups, self._step_tensors):
            steps = []
            for p in group["params"]:
                st = self.state.get(p)
                if not st:
                    continue
                o, n = offset_of[id(p)], p.numel()
                for key, buf in (("exp_avg", arena.exp_avg), ("exp_avg_sq", arena.exp_avg_sq)):
                    view = buf[o:o + n].view(p.shape)
                    if st[key].data_ptr() != view.data_ptr():
                        view.copy_(st[key].to(view.device, torch.float32))
                        st[key] = view
                steps.append(int(float(st.get("step", 0))))
                st["step"] = step_t                       # back to the group's shared tensor (torch's loader made per-parameter copies)
            # a torch.optim.AdamW checkpoint (the reference's format) has no `step_count` in its groups, and one saved before the first
            # step has no per-parameter state either
            group.setdefault("step_count", 0)
            if steps:
                group["step_count"] = max(steps)
            step_t.fill_(float(group["step_count"]))
