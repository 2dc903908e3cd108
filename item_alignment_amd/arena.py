"""Flat parameter arena: every trainable fp32 master parameter of a model lives in ONE contiguous HBM
buffer, with four more buffers of identical element layout next to it (gradients, Adam exp_avg,
exp_avg_sq, and the bf16 shadow the MFMA GEMMs read).

Why (MI355X-first, 288 GB HBM): the optimiser is one HBM-bound launch over the arena instead of one per
tensor; gradient buckets for the RCCL all-reduce are plain slices of the gradient arena (no packing
copies); q/k/v projection weights are laid out back to back so the fused QKV GEMM reads one [3H, H]
operand although the state_dict keeps the reference's separate query/key/value keys.

The reference equivalent is torch.optim.AdamW over `model.named_parameters()` with two groups
(finetune_multimodal.py:296-308): names containing "bias" or "LayerNorm.weight" get weight_decay 0.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, stream_ptr

ALIGN = 64          # elements; 256 B for fp32, 128 B for the bf16 shadow
CHUNK = 4096        # elements handled by one workgroup of the AdamW kernel
NO_DECAY = ("bias", "LayerNorm.weight")


def _round_up(n, a):
    return (n + a - 1) // a * a


class ParamArena:
    def __init__(self, model, device, frozen_ok=True):
        self.side_streams = []
        self.device = torch.device(device)
        named = [(n, p) for n, p in model.named_parameters()]
        by_id = {id(p): n for n, p in named}
        # registration order, except that members of a contiguity group (fused QKV operands) are placed
        # together where the group's first member appears; layers therefore stay in order and the
        # backward pass completes the gradient arena from its end towards its start.
        group_of = {}
        for mod in model.modules():
            for grp in getattr(mod, "arena_groups", lambda: [])():
                for p in grp:
                    group_of[id(p)] = grp
        order, seen = [], set()
        for n, p in named:
            for q in group_of.get(id(p), (p,)):
                if id(q) not in seen:
                    order.append(q); seen.add(id(q))
        self.names = [by_id[id(p)] for p in order]
        self.params = order
        self.offsets, off = [], 0
        for p in order:
            self.offsets.append(off)
            off += _round_up(p.numel(), ALIGN)
        self.numel = off
        self.master = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.shadow = torch.zeros(off, device=self.device, dtype=torch.bfloat16)
        self._shadow_of = {}
        chunks = []
        for name, p, o in zip(self.names, order, self.offsets):
            n = p.numel()
            view = self.master[o:o + n].view(p.shape)
            view.copy_(p.data.to(self.device, torch.float32))
            p.data = view
            p.grad = self.grad[o:o + n].view(p.shape)
            p._ia_arena = self              # item_alignment_amd.optim.AdamW finds the arena from the parameters it is handed
            self._shadow_of[id(p)] = self.shadow[o:o + n].view(p.shape)
            if p.requires_grad:
                decay = 0 if any(nd in name for nd in NO_DECAY) else 1
                for c in range(0, n, CHUNK):
                    chunks.append(((o + c) & 0xFFFFFFFF, (o + c) >> 32, min(CHUNK, n - c), decay))
        self.n_chunks = len(chunks)
        self.chunk_table = torch.from_numpy(np.asarray(chunks, dtype=np.uint32).reshape(-1, 4)).to(self.device)
        # transposed bf16 copies of registered 2-D weights (register_transposed): the data-gradient GEMMs read W^T k-contiguously
        self.shadow_t = None
        self._transposed = {}              # element offset -> (rows, cols)
        self._transposed_table = None
        self.step_count = 0
        self.stale_refreshes = 0
        self.optimizer_bound = False
        self.refresh_shadow()

    # ------------------------------------------------------------------ views
    def shadow_of(self, p):
        """bf16 copy of parameter p (refreshed by every optimiser step)."""
        return self._shadow_of[id(p)]

    def fused_shadow(self, params):
        """bf16 view spanning several parameters that were laid out back to back (e.g. q|k|v)."""
        first = self._shadow_of[id(params[0])]
        total = sum(p.numel() for p in params)
        start = first.storage_offset()
        flat = self.shadow[start:start + total]
        exp = start
        for p in params:
            if self._shadow_of[id(p)].storage_offset() != exp:
                raise RuntimeError("parameters are not contiguous in the arena")
            exp += p.numel()
        return flat

    def fused_master(self, params, grad=False):
        buf = self.grad if grad else self.master
        start = params[0].data.storage_offset() if not grad else params[0].grad.storage_offset()
        total = sum(p.numel() for p in params)
        return buf[start:start + total]

    # ------------------------------------------------------------------ maintenance
    def refresh_shadow(self):
        lib = _lib.load()
        self.join_side_streams()
        check(lib.ia_cast_f32_to_bf16(self.master.data_ptr(), self.shadow.data_ptr(), self.numel, stream_ptr()), "ia_cast_f32_to_bf16")
        self._versions = self._param_versions()
        self.refresh_transposed()

    def register_transposed(self, params):
        """bf16 W^T [cols, rows] of a 2-D weight -- or of several stacked ones laid out back to back (q | k | v = one [3H, H] matrix) --
        kept in `shadow_t` at the weight's own element offset and rewritten after every optimiser step / shadow refresh (one batched
        launch, ~4 B per parameter).  Returns the bf16 view."""
        params = list(params) if isinstance(params, (tuple, list)) else [params]
        first = self._shadow_of[id(params[0])]
        off, cols = first.storage_offset(), params[0].shape[1]
        rows, exp = 0, off
        for p in params:
            if p.dim() != 2 or p.shape[1] != cols or self._shadow_of[id(p)].storage_offset() != exp:
                raise RuntimeError("register_transposed: 2-D weights of one width, contiguous in the arena")
            rows += p.shape[0]
            exp += p.numel()
        if self.shadow_t is None:
            self.shadow_t = torch.zeros_like(self.shadow)
        if self._transposed.get(off) != (rows, cols):
            self._transposed[off] = (rows, cols)
            self._transposed_table = None
            self._transposed_stale = True      # the new image is only written by the next refresh_transposed()
        return self.shadow_t[off:off + rows * cols].view(cols, rows)

    def transposed_ready(self):
        """bring the transposed images up to date if one was registered since the last refresh (a Linear that registers its weight in
        its first forward calls this in front of its first data-gradient GEMM)"""
        if getattr(self, "_transposed_stale", False):
            self.refresh_transposed()

    def refresh_transposed(self):
        if not self._transposed:
            return
        lib = _lib.load()
        if self._transposed_table is None:
            ent = [((o & 0xFFFFFFFF), o >> 32, r, c) for o, (r, c) in sorted(self._transposed.items())]
            self._transposed_table = torch.from_numpy(np.asarray(ent, dtype=np.uint32).reshape(-1, 4)).to(self.device)
            self._transposed_tiles = max(((r + 63) // 64) * ((c + 63) // 64) for _, _, r, c in ent)
        check(lib.ia_transpose_bf16_batched(self.shadow.data_ptr(), self.shadow_t.data_ptr(), self._transposed_table.data_ptr(),
                                            self._transposed_table.shape[0], self._transposed_tiles, stream_ptr()), "ia_transpose_bf16_batched")
        self._transposed_stale = False

    def _param_versions(self):
        return sum(p._version for p in self.params)

    def sync_shadow(self):
        """Staleness guard, called at the top of every model forward.  The GEMMs read the bf16 shadow, which the fused optimiser rewrites
        together with the fp32 masters; anything ELSE that writes a parameter through torch (a foreign `torch.optim.AdamW(model.parameters())`
        kept from the reference loop, `p.mul_()`, `copy_`, an initialiser) bumps that parameter's version counter -- the shadow is then
        re-cast from the masters (one 6 B / parameter pass) instead of silently serving the old weights.  Writes through `p.data` bypass
        the counter, as they bypass autograd's own checks: call refresh_shadow() after those."""
        dropped = [i for i, p in enumerate(self.params) if p.grad is None]
        if dropped:
            # `optimizer.zero_grad()` of a torch optimizer (set_to_none=True is its default) dropped the p.grad views: to torch that MEANS
            # "the gradients are zero".  The kernels accumulate into the gradient arena whatever p.grad says, so clear what was dropped
            # and re-point the views -- otherwise the old gradients would be added to, and torch's step() would skip every parameter.
            if len(dropped) == len(self.params):
                self.zero_grad()
            else:
                for i in dropped:
                    o = self.offsets[i]
                    self.grad[o:o + self.params[i].numel()].zero_()
            self.reattach()
        if self._param_versions() != self._versions:
            self.refresh_shadow()
            self.stale_refreshes += 1
            return True
        return False

    def join_side_streams(self):
        """Models that run independent towers on extra HIP streams register them in `side_streams`: everything that reads or
        rewrites the arenas on the current stream (optimiser step, zero_grad) first waits for those streams."""
        if self.side_streams:
            cur = torch.cuda.current_stream()
            for s in self.side_streams:
                cur.wait_stream(s)

    def zero_grad(self):
        self.join_side_streams()
        self.grad.zero_()

    def reattach(self):
        """p.grad can be dropped by user code (optimizer.zero_grad(set_to_none=True)); re-point it."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def adamw_step(self, lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-5, grad_scale=1.0):
        """One fused AdamW update over the whole arena + bf16 shadow refresh (single launch)."""
        lib = _lib.load()
        self.step_count += 1
        self.join_side_streams()
        check(lib.ia_adamw_flat(self.master.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                self.shadow.data_ptr(), self.chunk_table.data_ptr(), self.n_chunks, lr, betas[0], betas[1], eps,
                                weight_decay, self.step_count, grad_scale, stream_ptr()), "ia_adamw_flat")
        self.refresh_transposed()

    def grad_buckets(self, bucket_bytes=64 << 20):
        """Contiguous slices of the gradient arena, last-to-first (backward produces the last layers'
        gradients first), each about bucket_bytes: the units of the data-parallel all-reduce."""
        per = max(ALIGN, bucket_bytes // 4 // ALIGN * ALIGN)
        out, end = [], self.numel
        while end > 0:
            start = max(0, end - per)
            out.append(self.grad[start:end])
            end = start
        return out
