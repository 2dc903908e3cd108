"""torch.autograd glue between the model classes and the HIP engine.

Each Function's forward/backward is a handful of C-ABI calls (item_alignment_amd._lib); no torch math
runs on the hot path.  Parameter gradients are written straight into the fp32 gradient arena (the
parameters' `.grad` views, see arena.py) by the kernels, with accumulate (+=) semantics, so the Functions
return None for parameter inputs: autograd only carries activation gradients between Functions.
"""
import ctypes as C

import os

import torch

from .. import _lib
from .._lib import LayerCfg, LayerGrads, LayerWeights, check, ptr, stream_ptr

BF16, F32 = torch.bfloat16, torch.float32
_step_seed = [0x1234567]


def set_step_seed(seed):
    """Dropout streams are keyed by (step seed, layer, element); the train loop bumps this every step."""
    _step_seed[0] = int(seed) & 0xFFFFFFFF


def step_seed():
    return _step_seed[0]


def _need_gpu(t, what):
    if not t.is_cuda:
        raise _lib.ItemAlignError(f"{what} is on {t.device}: the MI355X engine has no CPU path (move the model and batch to cuda)")


_notify_hooks = []


def register_grad_ready_hook(fn):
    """fn(list_of_parameters) is called right after the kernels that finish those parameters' gradients
    have been enqueued on the current stream (used by the data-parallel bucket all-reduce)."""
    _notify_hooks.append(fn)
    return fn


def clear_grad_ready_hooks():
    del _notify_hooks[:]


def _notify(params, final=False):
    """final=True: no other autograd node of this backward will add to these gradients (only then may a data-parallel bucket
    be reduced before backward ends).  Most nodes cannot know that - a module applied twice, a two-sided embedding, a chunked
    tower all += into the same gradient from several nodes - so the default is False and such parameters are reduced after
    backward; the encoder stacks, which hold >90 % of the parameters, count their uses (below)."""
    for fn in _notify_hooks:
        fn(params, final)


# uses of an encoder stack whose backward has not run yet in this step (forward +1, backward -1): the layer gradients are
# final when the count returns to zero.  Cleared by the train loop / reducer every step (a forward whose output never
# reaches the loss would otherwise leave a count behind).
_pending_uses = {}


def reset_use_counts():
    _pending_uses.clear()


# ------------------------------------------------------------------------------------------ embeddings
class EmbedLNFn(torch.autograd.Function):
    """word (+ redirected extra rows) + token type + position -> LayerNorm -> dropout (reference base.py:238-279)."""

    @staticmethod
    def forward(ctx, anchor, emb, ids, tts, pids, extra_idx, extra, drop_p, stream_id, row_order=None):
        lib = _lib.load()
        _need_gpu(ids, "input_ids")
        M = ids.numel()
        H = emb.word_embeddings.weight.shape[1]
        dev = ids.device
        z = torch.empty((M, H), device=dev, dtype=BF16)
        y = torch.empty((M, H), device=dev, dtype=BF16)
        mean = torch.empty(M, device=dev, dtype=F32)
        rstd = torch.empty(M, device=dev, dtype=F32)
        seed = step_seed()
        check(lib.ia_embed_ln_fwd(ids.data_ptr(), tts.data_ptr(), pids.data_ptr(), ptr(extra_idx), emb.word_embeddings.weight.data_ptr(),
                                  emb.token_type_embeddings.weight.data_ptr(), emb.position_embeddings.weight.data_ptr(), ptr(extra),
                                  emb.LayerNorm.weight.data_ptr(), emb.LayerNorm.bias.data_ptr(), z.data_ptr(), y.data_ptr(),
                                  mean.data_ptr(), rstd.data_ptr(), M, H, emb.eps, drop_p, seed, stream_id, stream_ptr()), "ia_embed_ln_fwd")
        ctx.emb, ctx.saved = emb, (ids, tts, pids, extra_idx, z, mean, rstd)
        ctx.row_order = row_order
        ctx.drop, ctx.seed, ctx.stream_id = drop_p, seed, stream_id
        ctx.extra_shape = None if extra is None else extra.shape
        ctx.seq_len = ids.shape[-1] if ids.dim() == 2 else M
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        emb = ctx.emb
        ids, tts, pids, extra_idx, z, mean, rstd = ctx.saved
        M, H = z.shape
        dy = dy.contiguous()
        dextra = torch.zeros(ctx.extra_shape, device=dy.device, dtype=F32) if ctx.extra_shape is not None else None
        ws_bytes = lib.ia_embed_ln_bwd_workspace_bytes(M, H)
        ws = torch.empty(ws_bytes, device=dy.device, dtype=torch.uint8)

        def g(p):
            return p.grad.data_ptr() if p.requires_grad else None
        check(lib.ia_embed_ln_bwd(dy.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), emb.LayerNorm.weight.data_ptr(),
                                  ids.data_ptr(), tts.data_ptr(), pids.data_ptr(), ptr(extra_idx), ptr(ctx.row_order), g(emb.word_embeddings.weight),
                                  g(emb.token_type_embeddings.weight), g(emb.position_embeddings.weight), ptr(dextra),
                                  g(emb.LayerNorm.weight), g(emb.LayerNorm.bias), M, H, ctx.seq_len, emb.word_pad, emb.pos_pad, ctx.drop, ctx.seed,
                                  ctx.stream_id, ws.data_ptr(), ws_bytes, stream_ptr()), "ia_embed_ln_bwd")
        _notify([emb.word_embeddings.weight, emb.token_type_embeddings.weight, emb.position_embeddings.weight, emb.LayerNorm.weight,
                 emb.LayerNorm.bias])
        return None, None, None, None, None, None, dextra, None, None, None


# --------------------------------------------------------------------------------------- encoder stack
class EncoderStackFn(torch.autograd.Function):
    """N encoder layers (post-LN RoBERTa or pre-LN ViT) through ia_layer_fwd / ia_layer_bwd.
    Returns every layer's output (the reference's `hidden_states[1:]`, text.py:1452)."""

    @staticmethod
    def forward(ctx, x, anchor, stack, key_mask, B, L, keep, cu_seqlens=None):
        # `keep` = grad mode of the caller (inside Function.forward grad mode is always off): keep the
        # per-layer activation stash for backward and, in train mode, apply dropout.
        # cu_seqlens (int32 [B+1], device): x holds the unpadded token rows of the B sequences back to back, L = the longest one.
        lib = _lib.load()
        _need_gpu(x, "hidden states")
        ctx.set_materialize_grads(False)
        n = len(stack.layers)
        training = stack.training and keep
        seed = step_seed()
        cfgs = [stack.layer_cfg(i, B, L, training, seed) for i in range(n)]
        if cu_seqlens is not None:
            for c in cfgs:
                c.cu_seqlens, c.total_tokens = cu_seqlens.data_ptr(), x.shape[0]
        ctx.cu_seqlens = cu_seqlens
        outs, cur = [], x.contiguous()
        inputs = [cur]
        mp = ptr(key_mask)
        if keep:
            stash_bytes = lib.ia_layer_stash_bytes(C.byref(cfgs[0]))
            stash = torch.empty(n * stash_bytes, device=x.device, dtype=torch.uint8)
            for i in range(n):
                y = torch.empty_like(cur)
                check(lib.ia_layer_fwd(C.byref(cfgs[i]), C.byref(stack.weights(i)), cur.data_ptr(), mp, y.data_ptr(),
                                       stash.data_ptr() + i * stash_bytes, stream_ptr()), f"ia_layer_fwd[{i}]")
                outs.append(y)
                cur = y
                inputs.append(cur)
        else:
            # evaluation / prediction (no_grad: reference finetune_multimodal.py:470-563): forward-only layers -- nothing is written for
            # a backward pass (no gelu' stream, no pre-LayerNorm sums) and one transient scratch serves the whole stack
            stash_bytes, stash = 0, None
            sbytes = lib.ia_layer_infer_scratch_bytes(C.byref(cfgs[0]))
            scratch = torch.empty(sbytes, device=x.device, dtype=torch.uint8)
            for i in range(n):
                y = torch.empty_like(cur)
                check(lib.ia_layer_fwd_infer(C.byref(cfgs[i]), C.byref(stack.weights(i)), cur.data_ptr(), mp, y.data_ptr(), scratch.data_ptr(),
                                             sbytes, stream_ptr()), f"ia_layer_fwd_infer[{i}]")
                outs.append(y)
                cur = y
            inputs = None
        ctx.stack, ctx.cfgs, ctx.stash, ctx.stash_bytes = stack, cfgs, stash, stash_bytes
        ctx.inputs, ctx.key_mask = inputs, key_mask
        if keep:
            _pending_uses[id(stack)] = _pending_uses.get(id(stack), 0) + 1
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        stack, cfgs, n = ctx.stack, ctx.cfgs, len(ctx.cfgs)
        scratch_bytes = lib.ia_layer_bwd_scratch_bytes(C.byref(cfgs[0]))
        scratch = torch.empty(scratch_bytes, device=ctx.stash.device, dtype=torch.uint8)
        mp = ptr(ctx.key_mask)
        dy, dy2buf, have_dy2 = None, None, False
        colsum_done = False
        left = _pending_uses.get(id(stack), 1) - 1
        _pending_uses[id(stack)] = left
        for i in reversed(range(n)):
            g = grads[i]
            if g is not None:
                g = g.contiguous()
                dy = g.clone() if dy is None else dy.add_(g)     # several tapped layers (cls_layers "1,2,..")
            if dy is None:
                continue                                          # layers above the last tapped one get no gradient
            x = ctx.inputs[i]
            # post-LN stacks hand the input gradient down in two parts (ia_layer_bwd2): the data path (dy, in place) and the residual
            # path (dy2); the layer below sums them inside its first LayerNorm backward.  The bottom layer returns the sum.
            split = not cfgs[i].pre_ln and i > 0
            if split and dy2buf is None:
                dy2buf = torch.empty_like(dy)
            if cfgs[i].pre_ln:
                # the fc2 bias gradient of block i-1 = column sums of this block's dx: taken from this block's last LayerNorm backward
                # (ia_layer_cfg.dx_colsum_out) unless a tapped hidden state adds another gradient to that tensor in between
                slot = stack.grads(i - 1).b_fc2 if i > 0 else None
                below = i > 0 and grads[i - 1] is None and bool(slot)
                cfgs[i].dx_colsum_out = slot if below else None
                cfgs[i].dy_colsum_done = int(colsum_done)
                colsum_done = below
            check(lib.ia_layer_bwd2(C.byref(cfgs[i]), C.byref(stack.weights(i)), C.byref(stack.grads(i)), x.data_ptr(), mp,
                                    ctx.inputs[i + 1].data_ptr(), ctx.stash.data_ptr() + i * ctx.stash_bytes, dy.data_ptr(),
                                    dy2buf.data_ptr() if have_dy2 else None, dy.data_ptr(), dy2buf.data_ptr() if split else None,
                                    scratch.data_ptr(), scratch_bytes, stream_ptr()), f"ia_layer_bwd2[{i}]")
            have_dy2 = split
            _notify(stack.layer_params(i), final=left <= 0)
        ctx.stash = None
        ctx.inputs = None
        return dy, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------ small ops
class GatherRowsFn(torch.autograd.Function):
    """features[:, idx, :] -> dropout, as fp32 [B, H] (reference base.py:104 / :140-141)."""

    @staticmethod
    def forward(ctx, hidden, anchor, rows, drop_p, stream_id):
        lib = _lib.load()
        _need_gpu(hidden, "hidden states")
        B, H = rows.numel(), hidden.shape[-1]
        out = torch.empty((B, H), device=hidden.device, dtype=F32)
        seed = step_seed()
        check(lib.ia_gather_rows_fwd(hidden.data_ptr(), H, rows.data_ptr(), out.data_ptr(), B, H, drop_p, seed, stream_id, stream_ptr()),
              "ia_gather_rows_fwd")
        ctx.rows, ctx.shape, ctx.drop, ctx.seed, ctx.stream_id = rows, hidden.shape, drop_p, seed, stream_id
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        dsrc = torch.zeros(ctx.shape, device=dout.device, dtype=BF16)
        dout = dout.contiguous()
        B, H = dout.shape
        check(lib.ia_gather_rows_bwd(dout.data_ptr(), H, ctx.rows.data_ptr(), dsrc.data_ptr(), B, H, ctx.drop, ctx.seed, ctx.stream_id, 0,
                                     stream_ptr()), "ia_gather_rows_bwd")
        return dsrc, None, None, None, None


class SpanMeanFn(torch.autograd.Function):
    """Mean of the token rows of each span, fp32 [S, H] (reference text.py:82-83: sequence_output[i, a:b, :].mean(axis=0))."""

    @staticmethod
    def forward(ctx, hidden, anchor, spans, span_ptr, B, L):
        lib = _lib.load()
        _need_gpu(hidden, "hidden states")
        hidden = hidden.contiguous()
        S, H = spans.shape[0], hidden.shape[-1]
        out = torch.empty((S, H), device=hidden.device, dtype=F32)
        check(lib.ia_span_mean_fwd(hidden.data_ptr(), H, spans.data_ptr(), out.data_ptr(), S, H, stream_ptr()), "ia_span_mean_fwd")
        ctx.spans, ctx.span_ptr, ctx.dims, ctx.shape = spans, span_ptr, (B, L, H), hidden.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        B, L, H = ctx.dims
        dsrc = torch.empty(ctx.shape, device=dout.device, dtype=BF16)
        check(lib.ia_span_mean_bwd(dout.contiguous().to(F32).data_ptr(), ctx.spans.data_ptr(), ctx.span_ptr.data_ptr(), dsrc.data_ptr(), B, L, H,
                                   stream_ptr()), "ia_span_mean_bwd")
        return dsrc, None, None, None, None, None


class LinearSmallFn(torch.autograd.Function):
    """y = act(x W^T + b) on a few rows, fp32 (dense+tanh of the heads, img2txt; reference base.py:142-143,530)."""

    @staticmethod
    def forward(ctx, x, weight, lin, act):
        # `weight` (= lin.weight) is passed as a tensor input only so autograd records this node even when x
        # needs no gradient (image embeddings); its gradient is written into the arena, not returned.
        lib = _lib.load()
        _need_gpu(x, "features")
        x = x.contiguous()
        B, K = x.shape
        N = lin.weight.shape[0]
        y = torch.empty((B, N), device=x.device, dtype=F32)
        check(lib.ia_linear_small_fwd(x.data_ptr(), K, lin.weight.data_ptr(), ptr(lin.bias), y.data_ptr(), B, N, K, act, stream_ptr()),
              "ia_linear_small_fwd")
        ctx.lin, ctx.act, ctx.x, ctx.y = lin, act, x, y
        ctx.need_dx = ctx.needs_input_grad[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        lin, x, y = ctx.lin, ctx.x, ctx.y
        dy = dy.contiguous()
        B, K = x.shape
        N = lin.weight.shape[0]
        dx = torch.empty_like(x) if ctx.need_dx else None
        wg = lin.weight.requires_grad
        check(lib.ia_linear_small_bwd(dy.data_ptr(), y.data_ptr(), x.data_ptr(), K, lin.weight.data_ptr(), ptr(dx), K,
                                      lin.weight.grad.data_ptr() if wg else None,
                                      lin.bias.grad.data_ptr() if (wg and lin.bias is not None) else None, B, N, K, ctx.act, stream_ptr()),
              "ia_linear_small_bwd")
        _notify([p for p in (lin.weight, lin.bias) if p is not None])
        return dx, None, None, None


class PairHeadCEFn(torch.autograd.Function):
    """logits = [x | y] W^T + b ; probs = softmax ; loss = mean cross-entropy, fused (reference
    base.py:114-115 + text.py:1360).  y may be None (single-feature head, base.py:155)."""

    @staticmethod
    def forward(ctx, x, y, lin, labels):
        lib = _lib.load()
        _need_gpu(x, "features")
        x = x.contiguous()
        y = None if y is None else y.contiguous()
        B, D = x.shape
        Cn = lin.weight.shape[0]
        logits = torch.empty((B, Cn), device=x.device, dtype=F32)
        probs = torch.empty((B, Cn), device=x.device, dtype=F32)
        loss = torch.zeros((), device=x.device, dtype=F32)
        per = torch.empty(B, device=x.device, dtype=F32)
        check(lib.ia_pair_head_ce_fwd(x.data_ptr(), ptr(y), lin.weight.data_ptr(), ptr(lin.bias), ptr(labels), logits.data_ptr(),
                                      probs.data_ptr(), loss.data_ptr(), per.data_ptr(), B, D, Cn, stream_ptr()), "ia_pair_head_ce_fwd")
        ctx.lin, ctx.x, ctx.y, ctx.labels, ctx.probs = lin, x, y, labels, probs
        ctx.mark_non_differentiable(logits, probs)
        return logits, probs, loss

    @staticmethod
    def backward(ctx, _dlogits, _dprobs, dloss):
        lib = _lib.load()
        lin, x, y = ctx.lin, ctx.x, ctx.y
        if ctx.labels is None:
            raise RuntimeError("PairHeadCEFn.backward without labels")
        B, D = x.shape
        Cn = lin.weight.shape[0]
        dloss = dloss.contiguous().to(F32)
        dx = torch.empty_like(x)
        dyv = torch.empty_like(y) if y is not None else None
        wg = lin.weight.requires_grad
        check(lib.ia_pair_head_ce_bwd(ctx.probs.data_ptr(), ctx.labels.data_ptr(), dloss.data_ptr(), x.data_ptr(), ptr(y),
                                      lin.weight.data_ptr(), dx.data_ptr(), ptr(dyv), lin.weight.grad.data_ptr() if wg else None,
                                      lin.bias.grad.data_ptr() if (wg and lin.bias is not None) else None, B, D, Cn, stream_ptr()),
              "ia_pair_head_ce_bwd")
        _notify([p for p in (lin.weight, lin.bias) if p is not None])
        return dx, dyv, None, None


class LayerNormFn(torch.autograd.Function):
    """Plain LayerNorm over bf16 rows (final ViT norm; timm VisionTransformer.norm)."""

    @staticmethod
    def forward(ctx, x, anchor, ln, eps):
        lib = _lib.load()
        _need_gpu(x, "tokens")
        x = x.contiguous()
        M, H = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(M, device=x.device, dtype=F32)
        rstd = torch.empty(M, device=x.device, dtype=F32)
        check(lib.ia_ln_fwd(x.data_ptr(), None, None, None, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ln.weight.data_ptr(),
                            ln.bias.data_ptr(), M, H, eps, 0.0, 0, 0, stream_ptr()), "ia_ln_fwd")
        ctx.ln, ctx.saved = ln, (x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        ln = ctx.ln
        x, mean, rstd = ctx.saved
        M, H = x.shape
        dy = dy.contiguous()
        dz = torch.empty_like(x)
        ws_bytes = lib.ia_ln_bwd_workspace_bytes(M, H)
        ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
        wg = ln.weight.requires_grad
        check(lib.ia_ln_bwd(dy.data_ptr(), None, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ln.weight.data_ptr(), dz.data_ptr(), None,
                            ln.weight.grad.data_ptr() if wg else None, ln.bias.grad.data_ptr() if wg else None, None, M, H, 0.0, 0, 0,
                            ws.data_ptr(), ws_bytes, 1, stream_ptr()), "ia_ln_bwd")
        _notify([ln.weight, ln.bias])
        return dz, None, None, None


class PatchEmbedFn(torch.autograd.Function):
    """images [B,3,S,S] fp32 -> tokens [B*(np+1), H] bf16: im2col + MFMA GEMM (+bias) + cls/pos assembly
    (timm PatchEmbed conv16/16 + cls_token + pos_embed, reference multimodal.py:811)."""

    @staticmethod
    def forward(ctx, images, anchor, vit, *more):
        """`more`: further image batches of the same shape that continue the batch (a two-tower model's second item): their patches
        are gathered straight into the one patch matrix, no concatenated copy of the fp32 images (0.9 GB per 256 pairs at 384 x 384)."""
        lib = _lib.load()
        parts = []
        for im in (images,) + tuple(more):
            _need_gpu(im, "images")
            parts.append(im.contiguous().to(F32))
        _, Cc, S, _ = parts[0].shape
        B = sum(im.shape[0] for im in parts)
        P, H = vit.patch_size, vit.embed_dim
        NP = (S // P) ** 2
        K = Cc * P * P
        dev = parts[0].device
        patches = torch.empty((B * NP, K), device=dev, dtype=BF16)
        row = 0
        for im in parts:
            if tuple(im.shape[1:]) != (Cc, S, S):
                raise ValueError("PatchEmbedFn: all image batches must share one [C, S, S]")
            check(lib.ia_im2col_patch(im.data_ptr(), patches[row:].data_ptr(), im.shape[0], Cc, S, P, stream_ptr()), "ia_im2col_patch")
            row += im.shape[0] * NP
        w = vit.arena.shadow_of(vit.patch_embed.proj.weight)        # [H, C*P*P] bf16
        pe = torch.empty((B * NP, H), device=dev, dtype=BF16)
        check(lib.ia_gemm_bf16(patches.data_ptr(), 0, K, w.data_ptr(), 0, K, pe.data_ptr(), 0, H, B * NP, H, K, 1,
                               vit.patch_embed.proj.bias.data_ptr(), None, 0, None, 0, None, 0, stream_ptr()), "ia_gemm_bf16[patch]")
        tok = torch.empty((B * (NP + 1), H), device=dev, dtype=BF16)
        check(lib.ia_vit_tokens_fwd(pe.data_ptr(), vit.cls_token.data_ptr(), vit.pos_embed.data_ptr(), tok.data_ptr(), B, NP, H, stream_ptr()),
              "ia_vit_tokens_fwd")
        ctx.vit, ctx.patches, ctx.dims, ctx.n_more = vit, patches, (B, NP, H, K), len(more)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        lib = _lib.load()
        vit, patches = ctx.vit, ctx.patches
        B, NP, H, K = ctx.dims
        dtok = dtok.contiguous()
        dpe = torch.empty((B * NP, H), device=dtok.device, dtype=BF16)
        check(lib.ia_vit_tokens_bwd(dtok.data_ptr(), dpe.data_ptr(), vit.cls_token.grad.data_ptr(), vit.pos_embed.grad.data_ptr(), B, NP, H, 1,
                                    stream_ptr()), "ia_vit_tokens_bwd")
        ws_bytes = lib.ia_colsum_workspace_bytes(B * NP, H)
        ws = torch.empty(ws_bytes, device=dtok.device, dtype=torch.uint8)
        check(lib.ia_colsum(dpe.data_ptr(), H, B * NP, H, vit.patch_embed.proj.bias.grad.data_ptr(), 1, ws.data_ptr(), ws_bytes, stream_ptr()),
              "ia_colsum")
        # dW[H, K] += dpe^T patches
        gws_bytes = lib.ia_gemm_workspace_bytes(H, K, B * NP, 1)
        gws = torch.empty(gws_bytes, device=dtok.device, dtype=torch.uint8) if gws_bytes else None
        check(lib.ia_gemm_bf16(dpe.data_ptr(), 1, H, patches.data_ptr(), 1, K, vit.patch_embed.proj.weight.grad.data_ptr(), 1, K, H, K, B * NP, 0,
                               None, None, 0, None, 1, ptr(gws), gws_bytes, stream_ptr()), "ia_gemm_bf16[patch wgrad]")
        _notify([vit.cls_token, vit.pos_embed, vit.patch_embed.proj.weight, vit.patch_embed.proj.bias])
        return (None, None, None) + (None,) * ctx.n_more


# ------------------------------------------------------------------ CoCa multimodal layers (cross_attn ensemble)
_TRANSPOSED = os.environ.get("IA_TRANSPOSED_SHADOWS", "1") != "0"      # (models/base.py reads the same switch for the encoder layers)


def _gemm(lib, a, a_ks, lda, b, b_ks, ldb, c, c_f32, ldc, M, N, K, epi=0, aux=None, ldaux=0, accumulate=0, what="ia_gemm_bf16"):
    ws_bytes = lib.ia_gemm_workspace_bytes(M, N, K, int(c_f32))
    ws = torch.empty(ws_bytes, device=c.device, dtype=torch.uint8) if ws_bytes else None
    check(lib.ia_gemm_bf16(a.data_ptr(), a_ks, lda, b.data_ptr(), b_ks, ldb, c.data_ptr(), int(c_f32), ldc, M, N, K, epi, None, ptr(aux),
                           ldaux, None, accumulate, ptr(ws), ws_bytes, stream_ptr()), what)


class LinearBf16Fn(torch.autograd.Function):
    """y = x W^T (+ add) for the bias-free Linears of the CoCa blocks (reference multimodal.py:545-551,646-659):
    MFMA GEMM on the weight's bf16 shadow; dgrad k-strided; wgrad accumulated in fp32 straight into the arena.

    swiglu_src = (src, column offset, F): x is silu(gate) * x' of the columns [offset, offset + 2F) of `src` (SwiGLUFn / FusedSplitFn
    produced it from that tensor, which they keep for their own backward anyway).  x is then NOT kept for the weight gradient -- it is
    recomputed from `src` in backward (one ia_swiglu_fwd pass, ~1 % of the GEMM it feeds).  With ff_mult 12 (coca_large.json) the two
    SwiGLU outputs of a multimodal layer are 48 KiB per token, a third of the layer's stash."""

    @staticmethod
    def forward(ctx, x, add, weight, owner, swiglu_src=None):
        lib = _lib.load()
        _need_gpu(x, "tokens")
        x = x.contiguous()
        M, K = x.shape
        w = owner.arena.shadow_of(weight)
        N = w.shape[0]
        y = torch.empty((M, N), device=x.device, dtype=BF16)
        if add is not None:
            add = add.contiguous()
        _gemm(lib, x, 0, K, w, 0, K, y, False, N, M, N, K, epi=3 if add is not None else 0, aux=add, ldaux=N if add is not None else 0)
        ctx.weight, ctx.owner = weight, owner
        ctx.x, ctx.swiglu_src, ctx.x_shape = (None, swiglu_src, (M, K)) if swiglu_src is not None else (x, None, (M, K))
        # W^T image for the data gradient (k-contiguous NT form: x 1.02-1.16 per launch over the k-strided form, profiles/r05_nn_vs_nt.txt;
        # the encoder layers have had theirs since round 5).  No extra memory: shadow_t spans the whole arena once anybody registers.
        ctx.wt = owner.arena.register_transposed(weight) if (_TRANSPOSED and ctx.needs_input_grad[0] and M >= 1024) else None
        ctx.need_dx, ctx.has_add = ctx.needs_input_grad[0], add is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        weight, x = ctx.weight, ctx.x
        dy = dy.contiguous()
        M, K = ctx.x_shape
        if x is None:                                  # recompute the SwiGLU output this Linear consumed
            src, off, F_ = ctx.swiglu_src
            x = torch.empty((M, K), device=dy.device, dtype=BF16)
            check(lib.ia_swiglu_fwd(src.data_ptr() + 2 * off, src.shape[1], x.data_ptr(), M, F_, stream_ptr()), "ia_swiglu_fwd[recompute]")
        ctx.x = ctx.swiglu_src = None                  # (the node outlives backward while the caller holds the loss: drop the tensors now)
        N = weight.shape[0]
        dx = None
        if ctx.need_dx:
            dx = torch.empty((M, K), device=dy.device, dtype=BF16)
            if ctx.wt is not None:
                ctx.owner.arena.transposed_ready()
                _gemm(lib, dy, 0, N, ctx.wt, 0, N, dx, False, K, M, K, N, what="ia_gemm_bf16[dgrad, W^T]")
            else:
                w = ctx.owner.arena.shadow_of(weight)
                _gemm(lib, dy, 0, N, w, 1, K, dx, False, K, M, K, N, what="ia_gemm_bf16[dgrad]")
        if weight.requires_grad:
            _gemm(lib, dy, 1, N, x, 1, K, weight.grad, True, K, N, K, M, accumulate=1, what="ia_gemm_bf16[wgrad]")
            _notify([weight])
        return dx, (dy if ctx.has_add else None), None, None, None


class GammaLayerNormFn(torch.autograd.Function):
    """LayerNorm with a learned gamma and a constant zero beta (reference multimodal.py:475-482)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        lib = _lib.load()
        _need_gpu(x, "tokens")
        x = x.contiguous()
        M, H = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(M, device=x.device, dtype=F32)
        rstd = torch.empty(M, device=x.device, dtype=F32)
        check(lib.ia_ln_fwd(x.data_ptr(), None, None, None, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                            M, H, eps, 0.0, 0, 0, stream_ptr()), "ia_ln_fwd")
        ctx.gamma, ctx.saved = gamma, (x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        gamma = ctx.gamma
        x, mean, rstd = ctx.saved
        M, H = x.shape
        dy = dy.contiguous()
        dz = torch.empty_like(x)
        ws_bytes = lib.ia_ln_bwd_workspace_bytes(M, H)
        ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
        wg = gamma.requires_grad
        check(lib.ia_ln_bwd(dy.data_ptr(), None, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dz.data_ptr(), None,
                            gamma.grad.data_ptr() if wg else None, None, None, M, H, 0.0, 0, 0, ws.data_ptr(), ws_bytes, 1, stream_ptr()),
              "ia_ln_bwd")
        if wg:
            _notify([gamma])
        ctx.saved = None
        return dz, None, None, None


class FusedSplitFn(torch.autograd.Function):
    """Consumes the fused projection of a ParallelTransformerBlock, rows = [q (heads*64) | k (64) | v (64) | x (F) | gate (F)]
    (reference multimodal.py:586): returns rotary(q) [M, heads*64], rotary(k) | v [M, 128] and silu(gate) * x [M, F]
    (:596-603, :521-524).  Backward assembles the whole projection gradient in one buffer (no torch adds)."""

    @staticmethod
    def forward(ctx, fused, n, heads, F_):
        lib = _lib.load()
        M, ld = fused.shape
        if ld != heads * 64 + 128 + 2 * F_:
            raise ValueError("fused projection width does not match (heads*64, 64, 64, 2*ff_inner)")
        q = torch.empty((M, heads * 64), device=fused.device, dtype=BF16)
        kv = torch.empty((M, 128), device=fused.device, dtype=BF16)
        s = torch.empty((M, F_), device=fused.device, dtype=BF16)
        check(lib.ia_rotary_split_fwd(fused.data_ptr(), ld, q.data_ptr(), kv.data_ptr(), M, n, heads, stream_ptr()), "ia_rotary_split_fwd")
        check(lib.ia_swiglu_fwd(fused.data_ptr() + 2 * (heads * 64 + 128), ld, s.data_ptr(), M, F_, stream_ptr()), "ia_swiglu_fwd")
        ctx.fused, ctx.dims = fused, (M, ld, n, heads, F_)
        return q, kv, s

    @staticmethod
    def backward(ctx, dq, dkv, ds):
        lib = _lib.load()
        M, ld, n, heads, F_ = ctx.dims
        fused = ctx.fused
        dfused = torch.empty((M, ld), device=fused.device, dtype=BF16)
        off = 2 * (heads * 64 + 128)
        check(lib.ia_rotary_split_bwd(dq.contiguous().data_ptr(), dkv.contiguous().data_ptr(), dfused.data_ptr(), ld, M, n, heads,
                                      stream_ptr()), "ia_rotary_split_bwd")
        check(lib.ia_swiglu_bwd(ds.contiguous().data_ptr(), fused.data_ptr() + off, ld, dfused.data_ptr() + off, ld, M, F_, stream_ptr()),
              "ia_swiglu_bwd")
        ctx.fused = None
        return dfused, None, None, None


class SwiGLUFn(torch.autograd.Function):
    """silu(gate) * x for rows [x (F) | gate (F)] (reference multimodal.py:521-524, the CrossAttention feed-forward :655-659)."""

    @staticmethod
    def forward(ctx, src):
        lib = _lib.load()
        src = src.contiguous()
        M, ld = src.shape
        F_ = ld // 2
        out = torch.empty((M, F_), device=src.device, dtype=BF16)
        check(lib.ia_swiglu_fwd(src.data_ptr(), ld, out.data_ptr(), M, F_, stream_ptr()), "ia_swiglu_fwd")
        ctx.src = src
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        src = ctx.src
        M, ld = src.shape
        dsrc = torch.empty_like(src)
        check(lib.ia_swiglu_bwd(dout.contiguous().data_ptr(), src.data_ptr(), ld, dsrc.data_ptr(), ld, M, ld // 2, stream_ptr()), "ia_swiglu_bwd")
        ctx.src = None
        return dsrc


class AttentionXFn(torch.autograd.Function):
    """softmax(q k^T * scale) v with Lq queries on Lk keys per sequence (ia_attn_fwd_x / ia_attn_bwd_x).
    q: [B*Lq, nh*64]; kv: [B*Lk, 2*nh*64] = k | v.  Multi-query attention passes nh = 1 with the query heads folded
    into rows (q.view(B*n*heads, 64), Lq = n*heads)."""

    @staticmethod
    def forward(ctx, q, kv, B, nh, Lq, Lk, scale):
        lib = _lib.load()
        q, kv = q.contiguous(), kv.contiguous()
        H = nh * 64
        out = torch.empty((B * Lq, H), device=q.device, dtype=BF16)
        lse = torch.empty((B, nh, Lq), device=q.device, dtype=F32)
        check(lib.ia_attn_fwd_x(q.data_ptr(), H, kv.data_ptr(), kv.data_ptr() + 2 * H, 2 * H, None, out.data_ptr(), H, lse.data_ptr(), B, nh,
                                Lq, Lk, scale, 0.0, 0, stream_ptr()), "ia_attn_fwd_x")
        ctx.saved, ctx.dims = (q, kv, out, lse), (B, nh, Lq, Lk, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        q, kv, out, lse = ctx.saved
        B, nh, Lq, Lk, scale = ctx.dims
        H = nh * 64
        dout = dout.contiguous()
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        delta = torch.empty((B, nh, Lq), device=q.device, dtype=F32)
        check(lib.ia_attn_bwd_x(q.data_ptr(), H, kv.data_ptr(), kv.data_ptr() + 2 * H, 2 * H, None, out.data_ptr(), dout.data_ptr(), H,
                                lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), H, dkv.data_ptr(), dkv.data_ptr() + 2 * H, 2 * H, B, nh, Lq,
                                Lk, scale, 0.0, 0, stream_ptr()), "ia_attn_bwd_x")
        ctx.saved = None
        return dq, dkv, None, None, None, None, None


# -------------------------------------------------------------------------------- PKGM rows, similarity head
class KGGatherFn(torch.autograd.Function):
    """sign(ent_emb[e]) [B, Dk] and rel_emb[r_1..P] [B*P, Dk] for one item side (reference base.py:347-367)."""

    @staticmethod
    def forward(ctx, anchor, emb, input_ids, ent_col, rel_lo, P):
        lib = _lib.load()
        _need_gpu(input_ids, "input_ids")
        ids = input_ids.contiguous()
        B, ld = ids.shape
        Dk = emb.ent_emb.weight.shape[1]
        h = torch.empty((B, Dk), device=ids.device, dtype=F32)
        r = torch.empty((B * P, Dk), device=ids.device, dtype=F32)
        check(lib.ia_kg_gather_fwd(emb.ent_emb.weight.data_ptr(), emb.rel_emb.weight.data_ptr(), ids.data_ptr(), ld, ent_col, rel_lo,
                                   h.data_ptr(), r.data_ptr(), B, P, Dk, stream_ptr()), "ia_kg_gather_fwd")
        ctx.emb, ctx.ids, ctx.dims = emb, ids, (B, ld, rel_lo, P, Dk)
        ctx.mark_non_differentiable(h)        # d sign(x)/dx = 0: the entity table gets no gradient (quirk A1)
        return h, r

    @staticmethod
    def backward(ctx, _dh, dr):
        lib = _lib.load()
        B, ld, rel_lo, P, Dk = ctx.dims
        w = ctx.emb.rel_emb.weight
        if w.requires_grad and dr is not None:
            check(lib.ia_kg_gather_bwd(dr.contiguous().data_ptr(), ctx.ids.data_ptr(), ld, rel_lo, w.grad.data_ptr(), B, P, Dk, stream_ptr()),
                  "ia_kg_gather_bwd")
        _notify([ctx.emb.ent_emb.weight, w])
        return None, None, None, None, None, None


class KGRowsFn(torch.autograd.Function):
    """[h + r | M h - r] rows of every item side, written side by side into one [B, sides*2P, H] buffer
    (reference base.py:369-392).  Inputs: (h, r, hp) per side."""

    @staticmethod
    def forward(ctx, P, *sides):
        lib = _lib.load()
        n = len(sides) // 3
        B, H = sides[0].shape
        rows = torch.empty((B, n * 2 * P, H), device=sides[0].device, dtype=F32)
        for i in range(n):
            h, r, hp = (t.contiguous() for t in sides[3 * i:3 * i + 3])
            check(lib.ia_kg_rows_fwd(h.data_ptr(), r.data_ptr(), hp.data_ptr(), rows.data_ptr(), n * 2 * P, i * 2 * P, B, P, H, stream_ptr()),
                  "ia_kg_rows_fwd")
        ctx.dims = (n, B, P, H)
        return rows

    @staticmethod
    def backward(ctx, drows):
        lib = _lib.load()
        n, B, P, H = ctx.dims
        drows = drows.contiguous()
        out = []
        for i in range(n):
            dh = torch.empty((B, H), device=drows.device, dtype=F32)
            dr = torch.empty((B * P, H), device=drows.device, dtype=F32)
            dhp = torch.empty((B, H), device=drows.device, dtype=F32)
            check(lib.ia_kg_rows_bwd(drows.data_ptr(), n * 2 * P, i * 2 * P, dh.data_ptr(), dr.data_ptr(), dhp.data_ptr(), B, P, H, stream_ptr()),
                  "ia_kg_rows_bwd")
            out += [dh, dr, dhp]
        return (None, *out)


SIM_MEASURES = {"inner_product": 0, "cosine": 1, "l1": 2, "l2": 3}


class PairSimFn(torch.autograd.Function):
    """sim [B], probs [B] of two fp32 feature matrices (reference base.py:75-88)."""

    @staticmethod
    def forward(ctx, x, y, measure):
        lib = _lib.load()
        _need_gpu(x, "features")
        x, y = x.contiguous(), y.contiguous()
        B, D = x.shape
        sim = torch.empty(B, device=x.device, dtype=F32)
        probs = torch.empty(B, device=x.device, dtype=F32)
        check(lib.ia_pair_sim_fwd(x.data_ptr(), y.data_ptr(), sim.data_ptr(), probs.data_ptr(), B, D, measure, stream_ptr()), "ia_pair_sim_fwd")
        ctx.saved, ctx.measure = (x, y, sim, probs), measure
        ctx.set_materialize_grads(False)
        return sim, probs

    @staticmethod
    def backward(ctx, dsim, dprobs):
        lib = _lib.load()
        x, y, sim, probs = ctx.saved
        B, D = x.shape
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        dsim = None if dsim is None else dsim.contiguous().to(F32)
        dprobs = None if dprobs is None else dprobs.contiguous().to(F32)
        check(lib.ia_pair_sim_bwd(x.data_ptr(), y.data_ptr(), sim.data_ptr(), probs.data_ptr(), ptr(dsim), ptr(dprobs), dx.data_ptr(),
                                  dy.data_ptr(), B, D, ctx.measure, stream_ptr()), "ia_pair_sim_bwd")
        return dx, dy, None
