// Fused multi-head self-attention for the RoBERTa / ViT towers (head dim 64), gfx950.
//
// Forward:  O = softmax(Q K^T * scale + key_mask) V   per (sequence, head), flash style:
// K/V stream through LDS in 64-key tiles (buffer_load ... lds, double buffered), scores never
// leave registers. Reference arithmetic: transformers RobertaSelfAttention (eager path) called from
// src/models/text.py:1241, and timm Attention called from src/models/multimodal.py:811.
//
// Everything is computed transposed so that softmax statistics are lane-local:
//   S^T[key][q] = K Q^T   : MFMA 32x32x16, A = K fragment (LDS, ds_read_b128), B = Q fragment (registers)
//                            -> lane (q = lane&31) holds 32 keys of its own query, partner lane^32 the rest
//   O^T[d][q]  += V^T P^T : A = V^T fragment (ds_read_b64_tr_b16 from the row-major V tile), B = P^T taken
//                            straight from the S^T accumulators (bf16-packed), no cross-lane traffic.
// The k-slot <-> key permutation inside a 16-key block is the one the accumulator layout dictates
// (slot (half,j) <-> key (j&3) + 8*(j>>2) + 4*half) and the V^T transpose read follows it.
//
// Backward = two kernels, both recompute P from Q, K and the saved log-sum-exp:
//   attn_bwd_dq  (block owns 128 queries, S^T orientation):  dQ^T += K^T dS^T
//   attn_bwd_dkv (block owns 128 keys,   S orientation):     dV^T += dO^T P ; dK^T += Q^T dS
#include "common.h"

namespace {

constexpr uint32_t OOB = 0xFFFFFFF0u;
constexpr float LOG2E = 1.4426950408889634f;

struct AttnArgs {
  const bf16* q; const bf16* k; const bf16* v;   // row = token (b*L + l), head h at column h*64
  const bf16* o; const bf16* d_o;                // forward output / its gradient
  bf16* out;                                      // forward: O
  bf16* dq; bf16* dk; bf16* dv;
  const uint8_t* mask;                            // [B, L] 1 = attend, may be null
  float* lse2;                                    // [B, nh, L]  log2-domain log-sum-exp of scaled scores
  float* delta;                                   // [B, nh, L]  rowsum(dO * O)
  int B, nh, L;
  int ld_qkv, ld_o, ld_dqkv;                      // row strides in elements
  uint32_t qkv_bytes, o_bytes;
  float sc;                                       // softmax scale * log2(e)
  float scale;                                    // softmax scale
  uint32_t thr16; float inv_keep; uint32_t seed;
};

// K tile / Q tile read with ds_read_b128 (row = key or query, 128 B rows, 16 B chunk XOR row&7)
IA_DEV int swz_b128(int row) { return row & 7; }
// tile read with the transpose read (row-major [row][64 d]); 32 B slot XOR
IA_DEV int swz_tr(int row) { return ((row >> 1) & 1) << 2; }

template <bool TR>
IA_DEV void stage64(__amdgpu_buffer_rsrc_t rs, char* s, size_t row0, int nvalid, int ld, int col0, int tid, int wave) {
  // 64 rows x 64 columns (bf16) -> 8 KiB, two issues of 256 lanes x 16 B
#pragma unroll
  for (int issue = 0; issue < 2; ++issue) {
    const int row = issue * 32 + (tid >> 3);
    const int c = (tid & 7) ^ (TR ? swz_tr(row) : swz_b128(row));
    uint32_t off = (uint32_t)(((row0 + row) * ld + col0 + c * 8) * 2);
    if (row >= nvalid) off = OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, IA_LDS(s + issue * 4096 + wave * 1024), 16, off, 0, 0, 0);
  }
}

IA_DEV bf16x8 frag_b128(const char* s, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(s + row * 128 + ((chunk ^ swz_b128(row)) << 4));
}

// A^T fragment for MFMA 32x32x16 out of a row-major [row][64] tile: lane (i = lane&31 -> column,
// half = lane>>5) gets rows row0 + {0..3, 8..11} + 4*half of column col0 + i.
IA_DEV bf16x8 frag_tr(const char* s, int row0, int col0, int lane) {
  const int p = lane & 15, G = lane >> 4;
  const int row = row0 + 4 * (G >> 1) + (p >> 2);
  const int col = col0 + 16 * (G & 1) + (p & 3) * 4;
  const int addr = row * 128 + ((((col >> 3) ^ swz_tr(row))) << 4) + (col & 7) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + addr));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + addr + 8 * 128));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

IA_DEV f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// row index inside a 32-row accumulator block for register r of lane-half hh
#define ACC_ROW(r, hh) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hh))

// dropout keep decision for element (q, key) of stream (b, h)
IA_DEV bool drop_keep(uint32_t seed, uint32_t stream, int q, int key, uint32_t thr16) {
  const uint32_t r = ia_rng(seed, stream, (uint32_t)q * 1024u + ((uint32_t)key >> 1));
  const uint32_t u = (key & 1) ? (r >> 16) : (r & 0xFFFFu);
  return u >= thr16;
}

// ------------------------------------------------------------------------------------------ forward
template <bool DROPOUT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lq = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, h = blockIdx.y, L = p.L;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const size_t rowbase = (size_t)b * L;
  const bool active = q0 < L;
  const int q = q0 + lq;
  const int qc = q < L ? q : L - 1;

  bf16x8 qf[4];
  {
    const bf16* qp = p.q + (rowbase + qc) * p.ld_qkv + h * 64 + hh * 8;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) qf[kb] = *reinterpret_cast<const bf16x8*>(qp + kb * 16);
  }
  const __amdgpu_buffer_rsrc_t rsK = ia_rsrc(p.k, p.qkv_bytes);
  const __amdgpu_buffer_rsrc_t rsV = ia_rsrc(p.v, p.qkv_bytes);

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o0 = zero16(), o1 = zero16();
  const int nkt = (L + 63) >> 6;
  stage64<false>(rsK, smem, rowbase, L, p.ld_qkv, h * 64, tid, wave);
  stage64<true>(rsV, smem + 8192, rowbase, L, p.ld_qkv, h * 64, tid, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint32_t stream_id = (uint32_t)(b * p.nh + h);

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) {
      char* nb = smem + (buf ^ 1) * 16384;
      stage64<false>(rsK, nb, rowbase + (kt + 1) * 64, L - (kt + 1) * 64, p.ld_qkv, h * 64, tid, wave);
      stage64<true>(rsV, nb + 8192, rowbase + (kt + 1) * 64, L - (kt + 1) * 64, p.ld_qkv, h * 64, tid, wave);
    }
    if (active) {
      const char* sK = smem + buf * 16384;
      const char* sV = sK + 8192;
      const int key = kt * 64 + lane;
      const bool kv = key < L && (p.mask == nullptr || p.mask[rowbase + key] != 0);
      const uint64_t valid = __ballot(kv);
      uint32_t vlo = (uint32_t)valid, vhi = (uint32_t)(valid >> 32);
      if (hh) { vlo >>= 4; vhi >>= 4; }

      f32x16 s0 = zero16(), s1 = zero16();
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 k0 = frag_b128(sK, lq, kb * 2 + hh);
        const bf16x8 k1 = frag_b128(sK, 32 + lq, kb * 2 + hh);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[kb], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[kb], s1, 0, 0, 0);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int bit = (r & 3) + 8 * (r >> 2);
        s0[r] = ((vlo >> bit) & 1) ? s0[r] * p.sc : -INFINITY;
        s1[r] = ((vhi >> bit) & 1) ? s1[r] * p.sc : -INFINITY;
        mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_use);
        s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_use);
        rs += s0[r] + s1[r];
      }
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      if (DROPOUT) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int kl = ACC_ROW(r, hh);   // even key, r+1 is the odd neighbour
          const uint32_t ra = ia_rng(p.seed, stream_id, (uint32_t)q * 1024u + (uint32_t)((kt * 64 + kl) >> 1));
          const uint32_t rb = ia_rng(p.seed, stream_id, (uint32_t)q * 1024u + (uint32_t)((kt * 64 + 32 + kl) >> 1));
          if ((ra & 0xFFFFu) < p.thr16) s0[r] = 0.f;
          if ((ra >> 16) < p.thr16) s0[r + 1] = 0.f;
          if ((rb & 0xFFFFu) < p.thr16) s1[r] = 0.f;
          if ((rb >> 16) < p.thr16) s1[r + 1] = 0.f;
        }
      }
      bf16x8 pf[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pf[0][j] = f2bf(s0[j]); pf[1][j] = f2bf(s0[8 + j]);
        pf[2][j] = f2bf(s1[j]); pf[3][j] = f2bf(s1[8 + j]);
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 v0 = frag_tr(sV, kb * 16, 0, lane);
        const bf16x8 v1 = frag_tr(sV, kb * 16, 32, lane);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf[kb], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf[kb], o1, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (!active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? p.inv_keep / l_tot : 0.f;
  if (q < L) {
    if (hh == 0 && p.lse2) p.lse2[((size_t)b * p.nh + h) * L + q] = m_run + __builtin_amdgcn_logf(l_tot);
    bf16* op = p.out + (rowbase + q) * p.ld_o + h * 64;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = 8 * rg + 4 * hh;
      bf16x4 a = {f2bf(o0[rg * 4] * inv), f2bf(o0[rg * 4 + 1] * inv), f2bf(o0[rg * 4 + 2] * inv), f2bf(o0[rg * 4 + 3] * inv)};
      bf16x4 c = {f2bf(o1[rg * 4] * inv), f2bf(o1[rg * 4 + 1] * inv), f2bf(o1[rg * 4 + 2] * inv), f2bf(o1[rg * 4 + 3] * inv)};
      *reinterpret_cast<bf16x4*>(op + d) = a;
      *reinterpret_cast<bf16x4*>(op + 32 + d) = c;
    }
  }
}

// ------------------------------------------------------------------------------- delta = rowsum(dO*O)
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs p) {
  // one 8-lane group per (token, head): 64 columns = 8 lanes x 8 bf16
  const int gid = (blockIdx.x * 256 + threadIdx.x) >> 3, sub = threadIdx.x & 7;
  const int total = p.B * p.L * p.nh;
  if (gid >= total) return;
  const int tok = gid / p.nh, h = gid % p.nh;
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p.o + (size_t)tok * p.ld_o + h * 64 + sub * 8);
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(p.d_o + (size_t)tok * p.ld_o + h * 64 + sub * 8);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += bf2f(a[j]) * bf2f(g[j]);
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  if (sub == 0) {
    const int b = tok / p.L, l = tok % p.L;
    p.delta[((size_t)b * p.nh + h) * p.L + l] = s;
  }
}

// ------------------------------------------------------------------------------------- backward: dQ
template <bool DROPOUT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs p) {
  // per buffer: K (b128 layout) | K (transpose-read layout) | V (b128 layout) = 24 KiB
  __shared__ __attribute__((aligned(16))) char smem[2 * 24576];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lq = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, h = blockIdx.y, L = p.L;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const size_t rowbase = (size_t)b * L;
  const bool active = q0 < L;
  const int q = q0 + lq;
  const int qc = q < L ? q : L - 1;

  bf16x8 qf[4], gf[4];
  {
    const bf16* qp = p.q + (rowbase + qc) * p.ld_qkv + h * 64 + hh * 8;
    const bf16* gp = p.d_o + (rowbase + qc) * p.ld_o + h * 64 + hh * 8;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      qf[kb] = *reinterpret_cast<const bf16x8*>(qp + kb * 16);
      gf[kb] = *reinterpret_cast<const bf16x8*>(gp + kb * 16);
    }
  }
  const size_t sidx = ((size_t)b * p.nh + h) * L + qc;
  const float lse = p.lse2[sidx];
  const float dlt = p.delta[sidx];
  const __amdgpu_buffer_rsrc_t rsK = ia_rsrc(p.k, p.qkv_bytes);
  const __amdgpu_buffer_rsrc_t rsV = ia_rsrc(p.v, p.qkv_bytes);
  const uint32_t stream_id = (uint32_t)(b * p.nh + h);

  f32x16 dq0 = zero16(), dq1 = zero16();
  const int nkt = (L + 63) >> 6;
  stage64<false>(rsK, smem, rowbase, L, p.ld_qkv, h * 64, tid, wave);
  stage64<true>(rsK, smem + 8192, rowbase, L, p.ld_qkv, h * 64, tid, wave);
  stage64<false>(rsV, smem + 16384, rowbase, L, p.ld_qkv, h * 64, tid, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) {
      char* nb = smem + (buf ^ 1) * 24576;
      const size_t r0 = rowbase + (kt + 1) * 64; const int nv = L - (kt + 1) * 64;
      stage64<false>(rsK, nb, r0, nv, p.ld_qkv, h * 64, tid, wave);
      stage64<true>(rsK, nb + 8192, r0, nv, p.ld_qkv, h * 64, tid, wave);
      stage64<false>(rsV, nb + 16384, r0, nv, p.ld_qkv, h * 64, tid, wave);
    }
    if (active) {
      const char* sK = smem + buf * 24576;
      const char* sKt = sK + 8192;
      const char* sV = sK + 16384;
      const int key = kt * 64 + lane;
      const bool kv = key < L && (p.mask == nullptr || p.mask[rowbase + key] != 0);
      const uint64_t valid = __ballot(kv);
      uint32_t vlo = (uint32_t)valid, vhi = (uint32_t)(valid >> 32);
      if (hh) { vlo >>= 4; vhi >>= 4; }

      f32x16 s0 = zero16(), s1 = zero16(), dp0 = zero16(), dp1 = zero16();
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 k0 = frag_b128(sK, lq, kb * 2 + hh);
        const bf16x8 k1 = frag_b128(sK, 32 + lq, kb * 2 + hh);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[kb], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[kb], s1, 0, 0, 0);
        const bf16x8 v0 = frag_b128(sV, lq, kb * 2 + hh);
        const bf16x8 v1 = frag_b128(sV, 32 + lq, kb * 2 + hh);
        dp0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, gf[kb], dp0, 0, 0, 0);
        dp1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, gf[kb], dp1, 0, 0, 0);
      }
      // dS^T = P^T * (dP^T_eff - delta) * scale
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int bit = (r & 3) + 8 * (r >> 2);
        const float pa = ((vlo >> bit) & 1) ? __builtin_amdgcn_exp2f(s0[r] * p.sc - lse) : 0.f;
        const float pb = ((vhi >> bit) & 1) ? __builtin_amdgcn_exp2f(s1[r] * p.sc - lse) : 0.f;
        float da = dp0[r], db = dp1[r];
        if (DROPOUT) {
          const int kl = kt * 64 + ACC_ROW(r, hh);
          da = drop_keep(p.seed, stream_id, q, kl, p.thr16) ? da * p.inv_keep : 0.f;
          db = drop_keep(p.seed, stream_id, q, kl + 32, p.thr16) ? db * p.inv_keep : 0.f;
        }
        s0[r] = pa * (da - dlt) * p.scale;
        s1[r] = pb * (db - dlt) * p.scale;
      }
      bf16x8 sf[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sf[0][j] = f2bf(s0[j]); sf[1][j] = f2bf(s0[8 + j]);
        sf[2][j] = f2bf(s1[j]); sf[3][j] = f2bf(s1[8 + j]);
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 k0 = frag_tr(sKt, kb * 16, 0, lane);
        const bf16x8 k1 = frag_tr(sKt, kb * 16, 32, lane);
        dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, sf[kb], dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, sf[kb], dq1, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (!active || q >= L) return;
  bf16* op = p.dq + (rowbase + q) * p.ld_dqkv + h * 64;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int d = 8 * rg + 4 * hh;
    bf16x4 a = {f2bf(dq0[rg * 4]), f2bf(dq0[rg * 4 + 1]), f2bf(dq0[rg * 4 + 2]), f2bf(dq0[rg * 4 + 3])};
    bf16x4 c = {f2bf(dq1[rg * 4]), f2bf(dq1[rg * 4 + 1]), f2bf(dq1[rg * 4 + 2]), f2bf(dq1[rg * 4 + 3])};
    *reinterpret_cast<bf16x4*>(op + d) = a;
    *reinterpret_cast<bf16x4*>(op + 32 + d) = c;
  }
}

// ---------------------------------------------------------------------------------- backward: dK, dV
template <bool DROPOUT>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs p) {
  // per buffer: Q (b128) | Q (transpose-read) | dO (b128) | dO (transpose-read) = 32 KiB
  __shared__ __attribute__((aligned(16))) char smem[2 * 32768];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lk = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, h = blockIdx.y, L = p.L;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const size_t rowbase = (size_t)b * L;
  const bool active = k0 < L;
  const int key = k0 + lk;
  const int kc = key < L ? key : L - 1;
  const bool key_ok = key < L && (p.mask == nullptr || p.mask[rowbase + kc] != 0);

  bf16x8 kf[4], vf[4];
  {
    const bf16* kp = p.k + (rowbase + kc) * p.ld_qkv + h * 64 + hh * 8;
    const bf16* vp = p.v + (rowbase + kc) * p.ld_qkv + h * 64 + hh * 8;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      kf[kb] = *reinterpret_cast<const bf16x8*>(kp + kb * 16);
      vf[kb] = *reinterpret_cast<const bf16x8*>(vp + kb * 16);
    }
  }
  const __amdgpu_buffer_rsrc_t rsQ = ia_rsrc(p.q, p.qkv_bytes);
  const __amdgpu_buffer_rsrc_t rsG = ia_rsrc(p.d_o, p.o_bytes);
  const uint32_t stream_id = (uint32_t)(b * p.nh + h);
  const float* lse_base = p.lse2 + ((size_t)b * p.nh + h) * L;
  const float* dlt_base = p.delta + ((size_t)b * p.nh + h) * L;

  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  const int nqt = (L + 63) >> 6;
  auto stage_all = [&](char* s, int qt) {
    const size_t r0 = rowbase + (size_t)qt * 64; const int nv = L - qt * 64;
    stage64<false>(rsQ, s, r0, nv, p.ld_qkv, h * 64, tid, wave);
    stage64<true>(rsQ, s + 8192, r0, nv, p.ld_qkv, h * 64, tid, wave);
    stage64<false>(rsG, s + 16384, r0, nv, p.ld_o, h * 64, tid, wave);
    stage64<true>(rsG, s + 24576, r0, nv, p.ld_o, h * 64, tid, wave);
  };
  stage_all(smem, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    if (qt + 1 < nqt) stage_all(smem + (buf ^ 1) * 32768, qt + 1);
    if (active) {
      const char* sQ = smem + buf * 32768;
      const char* sQt = sQ + 8192;
      const char* sG = sQ + 16384;
      const char* sGt = sQ + 24576;
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        const int qb = qt * 64 + qs * 32;     // first query of this 32-row sub tile
        if (qb >= L) break;
        // S[q][key] = Q K^T ; dP[q][key] = dO V^T   (rows = q in registers, column = key = lane)
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const bf16x8 a = frag_b128(sQ, qs * 32 + lk, kb * 2 + hh);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[kb], s, 0, 0, 0);
          const bf16x8 g = frag_b128(sG, qs * 32 + lk, kb * 2 + hh);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g, vf[kb], dp, 0, 0, 0);
        }
        bf16x8 pf[2], sf[2];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int qrow = qb + 8 * rg + 4 * hh;                 // 4 consecutive queries
          f32x4 ls, dl;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int qi = qrow + j < L ? qrow + j : L - 1;
            ls[j] = lse_base[qi]; dl[j] = dlt_base[qi];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = rg * 4 + j;
            const bool ok = key_ok && (qrow + j < L);
            float pv = ok ? __builtin_amdgcn_exp2f(s[r] * p.sc - ls[j]) : 0.f;
            float d = dp[r];
            float pd = pv;
            if (DROPOUT) {
              const bool keep = drop_keep(p.seed, stream_id, qrow + j, key, p.thr16);
              d = keep ? d * p.inv_keep : 0.f;
              pd = keep ? pv * p.inv_keep : 0.f;
            }
            const float ds = pv * (d - dl[j]) * p.scale;
            pf[r >> 3][r & 7] = f2bf(pd);
            sf[r >> 3][r & 7] = f2bf(ds);
          }
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const bf16x8 g0 = frag_tr(sGt, qs * 32 + kb * 16, 0, lane);
          const bf16x8 g1 = frag_tr(sGt, qs * 32 + kb * 16, 32, lane);
          dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, pf[kb], dv0, 0, 0, 0);
          dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, pf[kb], dv1, 0, 0, 0);
          const bf16x8 a0 = frag_tr(sQt, qs * 32 + kb * 16, 0, lane);
          const bf16x8 a1 = frag_tr(sQt, qs * 32 + kb * 16, 32, lane);
          dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, sf[kb], dk0, 0, 0, 0);
          dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, sf[kb], dk1, 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (!active || key >= L) return;
  bf16* kp = p.dk + (rowbase + key) * p.ld_dqkv + h * 64;
  bf16* vp = p.dv + (rowbase + key) * p.ld_dqkv + h * 64;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int d = 8 * rg + 4 * hh;
    bf16x4 a = {f2bf(dk0[rg * 4]), f2bf(dk0[rg * 4 + 1]), f2bf(dk0[rg * 4 + 2]), f2bf(dk0[rg * 4 + 3])};
    bf16x4 c = {f2bf(dk1[rg * 4]), f2bf(dk1[rg * 4 + 1]), f2bf(dk1[rg * 4 + 2]), f2bf(dk1[rg * 4 + 3])};
    *reinterpret_cast<bf16x4*>(kp + d) = a;
    *reinterpret_cast<bf16x4*>(kp + 32 + d) = c;
    bf16x4 e = {f2bf(dv0[rg * 4]), f2bf(dv0[rg * 4 + 1]), f2bf(dv0[rg * 4 + 2]), f2bf(dv0[rg * 4 + 3])};
    bf16x4 f = {f2bf(dv1[rg * 4]), f2bf(dv1[rg * 4 + 1]), f2bf(dv1[rg * 4 + 2]), f2bf(dv1[rg * 4 + 3])};
    *reinterpret_cast<bf16x4*>(vp + d) = e;
    *reinterpret_cast<bf16x4*>(vp + 32 + d) = f;
  }
}

int fill_args(AttnArgs& a, int B, int nh, int L, int ld_qkv, int ld_o, float scale, float drop_p, uint32_t seed) {
  if (B <= 0 || nh <= 0 || L <= 0 || L > 2048 || (ld_qkv & 7) || (ld_o & 7)) return IA_ERR_ARG;
  const uint64_t qb = (uint64_t)B * L * ld_qkv * 2, ob = (uint64_t)B * L * ld_o * 2;
  if (qb >= 0x7FFFFFFFull || ob >= 0x7FFFFFFFull) return IA_ERR_ARG;
  a.B = B; a.nh = nh; a.L = L; a.ld_qkv = ld_qkv; a.ld_o = ld_o; a.ld_dqkv = ld_qkv;
  // the rsrc is based at the k / v / q pointer itself: it may start up to 3*H columns into the
  // packed row, so the window covers "to the end of the last row" from that pointer at most
  a.qkv_bytes = (uint32_t)(qb - (uint64_t)(ld_qkv - nh * 64) * 2);
  a.o_bytes = (uint32_t)ob;
  a.scale = scale; a.sc = scale * LOG2E;
  a.thr16 = drop_p > 0.f ? (uint32_t)(drop_p * 65536.f + 0.5f) : 0u;
  a.inv_keep = drop_p > 0.f ? 1.f / (1.f - (float)a.thr16 / 65536.f) : 1.f;
  a.seed = seed;
  return IA_OK;
}

}  // namespace

// q, k, v: pointers to the first column of head 0 of each operand; all three share row stride ld_qkv
// (packed [tokens, 3H] projection output, or three separate [tokens, H] tensors with ld_qkv = H).
extern "C" int ia_attn_fwd(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, void* out,
                           int ld_o, float* lse2, int B, int nh, int L, float scale, float drop_p, uint32_t seed,
                           hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!q || !k || !v || !out) return IA_ERR_ARG;
  AttnArgs a{};
  int rc = fill_args(a, B, nh, L, ld_qkv, ld_o, scale, drop_p, seed);
  if (rc) return rc;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out; a.mask = key_mask; a.lse2 = lse2;
  dim3 grid((L + 127) / 128, nh, B), blk(256);
  if (a.thr16) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, blk, 0, stream, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, blk, 0, stream, a);
  return ia_check_launch();
}

// delta: caller-provided scratch of B*nh*L floats.
extern "C" int ia_attn_bwd(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out,
                           const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv,
                           int ld_dqkv, int B, int nh, int L, float scale, float drop_p, uint32_t seed, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!q || !k || !v || !out || !d_out || !lse2 || !delta || !dq || !dk || !dv) return IA_ERR_ARG;
  AttnArgs a{};
  int rc = fill_args(a, B, nh, L, ld_qkv, ld_o, scale, drop_p, seed);
  if (rc) return rc;
  if (ld_dqkv & 3) return IA_ERR_ARG;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (const bf16*)out; a.d_o = (const bf16*)d_out;
  a.mask = key_mask; a.lse2 = const_cast<float*>(lse2); a.delta = delta;
  a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.ld_dqkv = ld_dqkv;
  const int total = B * L * nh;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((total * 8 + 255) / 256), dim3(256), 0, stream, a);
  dim3 grid((L + 127) / 128, nh, B), blk(256);
  if (a.thr16) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, blk, 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, blk, 0, stream, a);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, blk, 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, blk, 0, stream, a);
  }
  return ia_check_launch();
}
