// Element-wise pieces of the CoCa multimodal layers (reference src/models/multimodal.py:495-626):
// rotary position embedding of the multi-query attention operands and the SwiGLU gate.  Both read their inputs in
// place out of the fused projection output ([tokens, heads*64 | 64 | 64 | 2*ff_inner], multimodal.py:586) so that
// projection is never split or copied by torch ops.  HBM-bound, 16-byte accesses.
#include "common.h"

namespace {

// RotaryEmbedding (multimodal.py:495-506): inv_freq[i] = 10000^(-2i/64), angle = position * inv_freq[i], i < 32;
// apply_rotary_pos_emb (:515-516) with rotate_half (:509-512): out[i] = t[i] cos - t[i+32] sin, out[i+32] = t[i+32] cos + t[i] sin.
// The backward pass is the transposed rotation (sign < 0).
IA_DEV void rotate8(const bf16x8& lo, const bf16x8& hi, int pos, int i0, float sign, bf16x8& olo, bf16x8& ohi) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float inv_freq = __builtin_exp2f(-(float)(2 * (i0 + j)) * (13.287712379549449f / 64.f));   // log2(10000) = 13.2877...
    float sn, cs;
    sincosf((float)pos * inv_freq, &sn, &cs);
    sn *= sign;
    const float a = bf2f(lo[j]), b = bf2f(hi[j]);
    olo[j] = f2bf(a * cs - b * sn);
    ohi[j] = f2bf(b * cs + a * sn);
  }
}

// one thread per (row, segment, j): segment < nh -> query head, == nh -> key, == nh + 1 -> value (copied)
__global__ __launch_bounds__(256) void rotary_split_kernel(const bf16* __restrict__ src, int ld_src, bf16* __restrict__ q_out,
                                                           bf16* __restrict__ kv_out, int M, int n, int nh) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = (nh + 2) * 4;
  if (idx >= (size_t)M * per_row) return;
  const int row = (int)(idx / per_row), rem = (int)(idx % per_row), seg = rem >> 2, j = rem & 3;
  const bf16* s = src + (size_t)row * ld_src + seg * 64 + j * 8;
  const bf16x8 lo = *reinterpret_cast<const bf16x8*>(s), hi = *reinterpret_cast<const bf16x8*>(s + 32);
  bf16* d = seg < nh ? q_out + (size_t)row * nh * 64 + seg * 64 + j * 8 : kv_out + (size_t)row * 128 + (seg - nh) * 64 + j * 8;
  bf16x8 olo = lo, ohi = hi;
  if (seg <= nh) rotate8(lo, hi, row % n, j * 8, 1.f, olo, ohi);
  *reinterpret_cast<bf16x8*>(d) = olo;
  *reinterpret_cast<bf16x8*>(d + 32) = ohi;
}

__global__ __launch_bounds__(256) void rotary_merge_bwd_kernel(const bf16* __restrict__ dq, const bf16* __restrict__ dkv,
                                                               bf16* __restrict__ dsrc, int ld_src, int M, int n, int nh) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = (nh + 2) * 4;
  if (idx >= (size_t)M * per_row) return;
  const int row = (int)(idx / per_row), rem = (int)(idx % per_row), seg = rem >> 2, j = rem & 3;
  const bf16* s = seg < nh ? dq + (size_t)row * nh * 64 + seg * 64 + j * 8 : dkv + (size_t)row * 128 + (seg - nh) * 64 + j * 8;
  const bf16x8 lo = *reinterpret_cast<const bf16x8*>(s), hi = *reinterpret_cast<const bf16x8*>(s + 32);
  bf16x8 olo = lo, ohi = hi;
  if (seg <= nh) rotate8(lo, hi, row % n, j * 8, -1.f, olo, ohi);
  bf16* d = dsrc + (size_t)row * ld_src + seg * 64 + j * 8;
  *reinterpret_cast<bf16x8*>(d) = olo;
  *reinterpret_cast<bf16x8*>(d + 32) = ohi;
}

// SwiGLU (multimodal.py:521-524): x, gate = chunk(2); out = silu(gate) * x
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16* __restrict__ src, int ld_src, bf16* __restrict__ out, int M, int F) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = F >> 3;
  if (idx >= (size_t)M * per_row) return;
  const int row = (int)(idx / per_row), c = (int)(idx % per_row) * 8;
  const bf16x8 x = *reinterpret_cast<const bf16x8*>(src + (size_t)row * ld_src + c);
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(src + (size_t)row * ld_src + F + c);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float gv = bf2f(g[j]);
    o[j] = f2bf(gv / (1.f + __expf(-gv)) * bf2f(x[j]));
  }
  *reinterpret_cast<bf16x8*>(out + (size_t)row * F + c) = o;
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ src, int ld_src,
                                                         bf16* __restrict__ dsrc, int ld_dsrc, int M, int F) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = F >> 3;
  if (idx >= (size_t)M * per_row) return;
  const int row = (int)(idx / per_row), c = (int)(idx % per_row) * 8;
  const bf16x8 x = *reinterpret_cast<const bf16x8*>(src + (size_t)row * ld_src + c);
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(src + (size_t)row * ld_src + F + c);
  const bf16x8 d = *reinterpret_cast<const bf16x8*>(dout + (size_t)row * F + c);
  bf16x8 dx, dg;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float gv = bf2f(g[j]), dv = bf2f(d[j]);
    const float sg = 1.f / (1.f + __expf(-gv));
    dx[j] = f2bf(dv * gv * sg);
    dg[j] = f2bf(dv * bf2f(x[j]) * sg * (1.f + gv * (1.f - sg)));
  }
  *reinterpret_cast<bf16x8*>(dsrc + (size_t)row * ld_dsrc + c) = dx;
  *reinterpret_cast<bf16x8*>(dsrc + (size_t)row * ld_dsrc + F + c) = dg;
}

}  // namespace

// src: [M, ld_src] bf16 rows holding q (nh heads x 64) | k (64) | v (64) from column 0; position = row % n.
// q_out: [M, nh*64] rotated queries; kv_out: [M, 128] = rotated k | v.
extern "C" int ia_rotary_split_fwd(const void* src, int ld_src, void* q_out, void* kv_out, int M, int n, int nh, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !q_out || !kv_out || M <= 0 || n <= 0 || nh <= 0 || (ld_src & 7) || ld_src < nh * 64 + 128) return IA_ERR_ARG;
  const size_t total = (size_t)M * (nh + 2) * 4;
  hipLaunchKernelGGL(rotary_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)src, ld_src,
                     (bf16*)q_out, (bf16*)kv_out, M, n, nh);
  return ia_check_launch();
}

// inverse of the above for gradients: dsrc[:, 0 : nh*64+128] = (R^T dq | R^T dk | dv)
extern "C" int ia_rotary_split_bwd(const void* dq, const void* dkv, void* dsrc, int ld_src, int M, int n, int nh, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dq || !dkv || !dsrc || M <= 0 || n <= 0 || nh <= 0 || (ld_src & 7) || ld_src < nh * 64 + 128) return IA_ERR_ARG;
  const size_t total = (size_t)M * (nh + 2) * 4;
  hipLaunchKernelGGL(rotary_merge_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)dq,
                     (const bf16*)dkv, (bf16*)dsrc, ld_src, M, n, nh);
  return ia_check_launch();
}

// src points at the first of 2F columns (x | gate) of rows with stride ld_src; out: [M, F]
extern "C" int ia_swiglu_fwd(const void* src, int ld_src, void* out, int M, int F, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !out || M <= 0 || F <= 0 || (F & 7) || (ld_src & 7) || ld_src < 2 * F) return IA_ERR_ARG;
  const size_t total = (size_t)M * (F >> 3);
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)src, ld_src, (bf16*)out, M, F);
  return ia_check_launch();
}

extern "C" int ia_swiglu_bwd(const void* dout, const void* src, int ld_src, void* dsrc, int ld_dsrc, int M, int F, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dout || !src || !dsrc || M <= 0 || F <= 0 || (F & 7) || (ld_src & 7) || (ld_dsrc & 7) || ld_src < 2 * F || ld_dsrc < 2 * F)
    return IA_ERR_ARG;
  const size_t total = (size_t)M * (F >> 3);
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)dout, (const bf16*)src,
                     ld_src, (bf16*)dsrc, ld_dsrc, M, F);
  return ia_check_launch();
}
