"""Raw (non-autograd) Python entry points over the C ABI: argument checking, output allocation and the
`data_ptr()` plumbing.  Every function launches HIP kernels from libitemalign_hip.so on torch's current
stream; tensors must be CUDA(HIP)-resident and contiguous.  No function here computes with torch ops.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_ADD, EPI_DGELU, EPI_BIAS_ADD, EPI_DGELU_COLSUM = 0, 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_TANH = 0, 1
BF16, F32 = torch.bfloat16, torch.float32


def _need(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.ItemAlignError(f"{name} must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def gemm(a, b, *, a_kstrided=False, b_kstrided=False, epilogue=EPI_NONE, bias=None, aux=None, out=None, out_f32=False,
         accumulate=False, pre_out=None, colsum_out=None):
    """C = A*B (+epilogue).  a: [M,K] (or [K,M] if a_kstrided); b: [N,K] (or [K,N] if b_kstrided).
    EPI_DGELU_COLSUM also adds the column sums of C into colsum_out (fp32 [N])."""
    lib = _lib.load()
    _need(a, BF16, "a"); _need(b, BF16, "b"); _need(bias, F32, "bias"); _need(aux, BF16, "aux")
    if a_kstrided:
        K, M = a.shape
    else:
        M, K = a.shape
    if b_kstrided:
        Kb, N = b.shape
    else:
        N, Kb = b.shape
    if K != Kb:
        raise ValueError(f"gemm: inner dims differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=F32 if out_f32 else BF16)
    _need(out, F32 if out_f32 else BF16, "out")
    if epilogue == EPI_BIAS_GELU and pre_out is None:
        pre_out = torch.empty((M, N), device=a.device, dtype=BF16)
    ws_bytes = lib.ia_gemm_workspace_bytes(M, N, K, int(out_f32))
    if epilogue == EPI_DGELU_COLSUM:
        _need(colsum_out, F32, "colsum_out")
        if colsum_out is None:
            raise ValueError("gemm: EPI_DGELU_COLSUM needs colsum_out")
        pre_out = colsum_out
        ws_bytes = max(ws_bytes, lib.ia_gemm_colsum_workspace_bytes(M, N))
    ws = torch.empty(ws_bytes, device=a.device, dtype=torch.uint8) if ws_bytes else None
    check(lib.ia_gemm_bf16(a.data_ptr(), int(a_kstrided), a.shape[1], b.data_ptr(), int(b_kstrided), b.shape[1], out.data_ptr(),
                           int(out_f32), N, M, N, K, epilogue, ptr(bias), ptr(aux), N if aux is not None else 0, ptr(pre_out),
                           int(accumulate), ptr(ws), ws_bytes, stream_ptr()), "ia_gemm_bf16")
    if epilogue == EPI_BIAS_GELU:
        return out, pre_out
    return out


def ln_fwd(x, gamma, beta, eps, *, bias=None, residual=None, write_z=True, drop_p=0.0, seed=0, stream_id=0):
    lib = _lib.load()
    _need(x, BF16, "x"); _need(gamma, F32, "gamma"); _need(beta, F32, "beta"); _need(bias, F32, "bias"); _need(residual, BF16, "residual")
    M, H = x.shape
    y = torch.empty_like(x)
    z = torch.empty_like(x) if write_z else None
    mean = torch.empty(M, device=x.device, dtype=F32)
    rstd = torch.empty(M, device=x.device, dtype=F32)
    check(lib.ia_ln_fwd(x.data_ptr(), ptr(bias), ptr(residual), ptr(z), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                        gamma.data_ptr(), ptr(beta), M, H, eps, drop_p, seed, stream_id, stream_ptr()), "ia_ln_fwd")
    return y, z, mean, rstd


def ln_bwd(dy, z, mean, rstd, gamma, *, dres=None, dgamma=None, dbeta=None, dbias=None, drop_p=0.0, seed=0, stream_id=0):
    lib = _lib.load()
    _need(dy, BF16, "dy"); _need(z, BF16, "z"); _need(dres, BF16, "dres")
    M, H = dy.shape
    dz = torch.empty_like(dy)
    dx = torch.empty_like(dy) if drop_p > 0 else None
    ws_bytes = lib.ia_ln_bwd_workspace_bytes(M, H)
    ws = torch.empty(ws_bytes, device=dy.device, dtype=torch.uint8)
    check(lib.ia_ln_bwd(dy.data_ptr(), ptr(dres), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dz.data_ptr(),
                        ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dbias), M, H, drop_p, seed, stream_id, ws.data_ptr(), ws_bytes, 1,
                        stream_ptr()), "ia_ln_bwd")
    return dz, dx


def colsum(x, out, accumulate=True):
    lib = _lib.load()
    _need(x, BF16, "x"); _need(out, F32, "out")
    M, N = x.shape
    ws_bytes = lib.ia_colsum_workspace_bytes(M, N)
    ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
    check(lib.ia_colsum(x.data_ptr(), N, M, N, out.data_ptr(), int(accumulate), ws.data_ptr(), ws_bytes, stream_ptr()), "ia_colsum")
    return out


def gemm_qscale(a, b, bias, scaled_cols, col_scale):
    """bf16 [M, N] = (a [M, K] @ b [N, K]^T + bias), columns < scaled_cols multiplied by col_scale before the bf16 rounding (the QKV
    projection that hands pre-scaled q to attn_fwd / attn_bwd with q_prescaled=True)"""
    lib = _lib.load()
    _need(a, BF16, "a"); _need(b, BF16, "b"); _need(bias, F32, "bias")
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), device=a.device, dtype=BF16)
    check(lib.ia_gemm_bf16_qscale(a.data_ptr(), K, b.data_ptr(), K, out.data_ptr(), N, M, N, K, bias.data_ptr(), scaled_cols, col_scale,
                                  stream_ptr()), "ia_gemm_bf16_qscale")
    return out


def attn_fwd(qkv, B, L, nh, *, key_mask=None, scale=0.125, drop_p=0.0, seed=0, q_prescaled=False):
    """qkv: packed [B*L, 3*nh*64] bf16.  Returns (ctx [B*L, nh*64], lse2 [B, nh, L]).
    q_prescaled: the q columns already hold q * scale * log2(e) (gemm_qscale)"""
    lib = _lib.load()
    _need(qkv, BF16, "qkv"); _need(key_mask, torch.uint8, "key_mask")
    H = nh * 64
    out = torch.empty((B * L, H), device=qkv.device, dtype=BF16)
    lse = torch.empty((B, nh, L), device=qkv.device, dtype=F32)
    base = qkv.data_ptr()
    fn = lib.ia_attn_fwd_ps if q_prescaled else lib.ia_attn_fwd
    check(fn(base, base + 2 * H, base + 4 * H, 3 * H, ptr(key_mask), out.data_ptr(), H, lse.data_ptr(), B, nh, L, scale, drop_p, seed,
             stream_ptr()), "ia_attn_fwd")
    return out, lse


def attn_bwd(qkv, ctx, d_ctx, lse, B, L, nh, *, key_mask=None, scale=0.125, drop_p=0.0, seed=0, dbias=None, q_prescaled=False, return_delta=False,
             masked_rows_dead=False):
    """dbias (fp32 [3H], optional): += column sums of dqkv, i.e. the bias gradient of the fused QKV projection, out of the same launches.
    return_delta: also hand back the [B, nh, L] scratch (the softmax-gradient delta where the dQ / dK,dV kernel pair ran: L > 256 or
    IA_ATTN_EXACT_DELTA=1; the fused backward keeps its key bits there)"""
    lib = _lib.load()
    _need(qkv, BF16, "qkv"); _need(ctx, BF16, "ctx"); _need(d_ctx, BF16, "d_ctx"); _need(lse, F32, "lse")
    H = nh * 64
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, nh, L), device=qkv.device, dtype=F32)
    base, dbase = qkv.data_ptr(), dqkv.data_ptr()
    if (q_prescaled or masked_rows_dead) and dbias is None:
        dbias = torch.zeros(3 * H, device=qkv.device, dtype=F32)
    if masked_rows_dead:          # (ia_attn_bwd_bias_ex: d_ctx must be zero at every masked position)
        ws = torch.empty(lib.ia_attn_bwd_bias_workspace_bytes(B, nh, L), device=qkv.device, dtype=torch.uint8)
        check(lib.ia_attn_bwd_bias_ex((1 if q_prescaled else 0) | 2, base, base + 2 * H, base + 4 * H, 3 * H, ptr(key_mask), ctx.data_ptr(),
                                      d_ctx.data_ptr(), H, lse.data_ptr(), delta.data_ptr(), dbase, dbase + 2 * H, dbase + 4 * H, 3 * H,
                                      dbias.data_ptr(), ws.data_ptr(), ws.numel(), B, nh, L, scale, drop_p, seed, stream_ptr()), "ia_attn_bwd_bias_ex")
        return (dqkv, delta) if return_delta else dqkv
    if dbias is not None:
        _need(dbias, F32, "dbias")
        ws = torch.empty(lib.ia_attn_bwd_bias_workspace_bytes(B, nh, L), device=qkv.device, dtype=torch.uint8)
        fn = lib.ia_attn_bwd_bias_ps if q_prescaled else lib.ia_attn_bwd_bias
        check(fn(base, base + 2 * H, base + 4 * H, 3 * H, ptr(key_mask), ctx.data_ptr(), d_ctx.data_ptr(), H, lse.data_ptr(),
                                   delta.data_ptr(), dbase, dbase + 2 * H, dbase + 4 * H, 3 * H, dbias.data_ptr(), ws.data_ptr(), ws.numel(),
                                   B, nh, L, scale, drop_p, seed, stream_ptr()), "ia_attn_bwd_bias")
        return (dqkv, delta) if return_delta else dqkv
    check(lib.ia_attn_bwd(base, base + 2 * H, base + 4 * H, 3 * H, ptr(key_mask), ctx.data_ptr(), d_ctx.data_ptr(), H, lse.data_ptr(),
                          delta.data_ptr(), dbase, dbase + 2 * H, dbase + 4 * H, 3 * H, B, nh, L, scale, drop_p, seed, stream_ptr()),
          "ia_attn_bwd")
    return (dqkv, delta) if return_delta else dqkv


def cast_to_bf16(src, dst=None):
    lib = _lib.load()
    _need(src, F32, "src")
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=BF16)
    check(lib.ia_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), stream_ptr()), "ia_cast_f32_to_bf16")
    return dst


def cast_to_f32(src, dst=None):
    lib = _lib.load()
    _need(src, BF16, "src")
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=F32)
    check(lib.ia_cast_bf16_to_f32(src.data_ptr(), dst.data_ptr(), src.numel(), stream_ptr()), "ia_cast_bf16_to_f32")
    return dst
