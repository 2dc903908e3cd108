// LDS-resident fused self-attention for sequences whose K and V fit one CU's LDS (L <= 608 at head dim 64): the text
// towers (L = 255 / 510) and ViT-384 (577 tokens).  Reference arithmetic: transformers RobertaSelfAttention (eager path)
// called from src/models/text.py:1241, timm Attention called from src/models/multimodal.py:811.
//
// One workgroup owns one (sequence, head).  It pulls that head's whole K and V (forward, dQ) or Q and dO (dK/dV) into LDS
// once, tile by tile with buffer_load ... lds, and its waves then walk independent 32-row chunks against the resident
// operand with NO workgroup barrier in the loop (only the first pass waits, per 64-row tile, for the DMA to land).  The waves
// of a SIMD drift apart, so one wave's softmax VALU work runs under the other's MFMAs; inside a wave the scores of tile t+1 are
// issued before the softmax of tile t.
//
// Conventions that differ from attention.hip (the streaming kernels, which remain the general path):
//   * q is PRE-SCALED by softmax_scale * log2(e) (the QKV projection's epilogue does it, gemm.hip qs_cols), so q.k is
//     already the log2-domain logit;
//   * the running softmax reference enters through the MFMA's C operand: S = K Q^T + (-m) comes out of the matrix pipe
//     ready for exp2 (no per-score fma), and in the backward C = -lse (and C = -delta for dP) the same way;
//   * the saved statistic is nlse2 = -(m + log2 l), the negated log2-domain log-sum-exp;
//   * one XOR swizzle serves ds_read_b128 fragments and ds_read_b64_tr_b16 fragments of the same row-major [row][64] image
//     (fsw below), so an operand needed both ways is resident once.
#include "common.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr uint32_t OOB = 0xFFFFFFF0u;
constexpr int RES_MAX_KT = 10;
constexpr int RES_MAX_L = 608;            // two [L][64] bf16 images + 1.1 KiB of bookkeeping in 160 KiB of LDS
constexpr float RESCALE_THR = 8.f;        // log2 units: probabilities stay below 2^8 between re-basings
constexpr float LN2 = 0.6931471805599453f;

struct ResArgs {
  const bf16* q; const bf16* k; const bf16* v;   // row = token (b*L + l), head h at column h*64; q pre-scaled
  const bf16* o; const bf16* d_o;
  bf16* out; bf16* dq; bf16* dk; bf16* dv;
  const uint8_t* mask;                            // [B, L] 1 = attend, may be null
  float* nlse;                                    // [B, nh, L]
  float* delta;                                   // [B, nh, L]  rowsum(dO * O)
  int B, nh, L;
  int ld_q, ld_kv, ld_o, ld_dq, ld_dkv;
  uint32_t q_bytes, kv_bytes, o_bytes;
  float scale;                                    // softmax scale (applied to dQ), dK gets ln 2 (see header)
  uint32_t thr16; float inv_keep; uint32_t seed;
  int dbg;                                        // tuning switches (IA_ATTN_DBG), 0 in production
};

// 16-byte chunk XOR of row `row` of a [row][64 bf16] image.  ds_read_b128 (lane = row, fixed chunk) is served in 16-lane
// groups {0-3,12-15,20-27} ...: bits 2,3 and 1 of the row make every group hit 16 different 16-byte bank slots; the transpose
// read takes rows r..r+3 in one 32-lane group and needs rows r and r+2 in different 64-byte halves: bit 1 -> chunk bit 2.
IA_DEV int fsw(int row) { return ((row >> 2) & 3) | (((row >> 1) & 1) << 2); }

// stage rows [g*8, g*8+8) of one head's [L][64] operand: lane i -> row g*8 + i/8, LDS chunk position i%8
IA_DEV void stage8(__amdgpu_buffer_rsrc_t rs, char* s, size_t rowbase, int L, int ld, int col0, int g, int lane) {
  const int row = g * 8 + (lane >> 3);
  const int c = (lane & 7) ^ fsw(row);
  uint32_t off = (uint32_t)(((rowbase + row) * ld + col0 + c * 8) * 2);
  if (row >= L) off = OOB;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, IA_LDS(s + g * 1024), 16, off, 0, 0, 0);
}

// Geometry of a resident [L][64] image: full 64-row tiles, then a tail rounded up to 32 rows.
struct Geo {
  int L, nfull, tail, nkt, Lp, groups;
  IA_DEV explicit Geo(int L_) : L(L_) {
    nfull = L >> 6; tail = L & 63;
    nkt = nfull + (tail ? 1 : 0);
    Lp = nfull * 64 + ((tail + 31) & ~31);
    groups = Lp >> 3;
  }
  IA_DEV int tile_groups(int t) const { return t < nfull ? 8 : (Lp - nfull * 64) >> 3; }
};

IA_DEV uint32_t lds_addr(const void* p) { return ia_lds_addr(p); }

// byte offset of this lane's 8-byte piece for the transpose read of column block col0 (0 / 32), rows 4*half + (p>>2) + radd
IA_DEV uint32_t tr_off(int lane, int col0, int radd) {
  const int p = lane & 15, G = lane >> 4;
  const int row = 4 * (G >> 1) + (p >> 2) + radd;
  const int col = col0 + 16 * (G & 1) + (p & 3) * 4;
  return (uint32_t)(row * 128 + ((((col >> 3) ^ fsw(row))) << 4) + (col & 7) * 2);
}
struct TrBase { uint32_t lo0, hi0, lo1, hi1; };   // lane bases: columns 0..31 / 32..63, rows +0 / +8
IA_DEV TrBase tr_base(const void* s, int lane) {
  const uint32_t a = lds_addr(s);
  return {a + tr_off(lane, 0, 0), a + tr_off(lane, 0, 8), a + tr_off(lane, 32, 0), a + tr_off(lane, 32, 8)};
}
struct TrPair {   // the two A^T fragments (columns 0..31 and 32..63) of one 16-row step
  s16x4 lo0, hi0, lo1, hi1;
  IA_DEV bf16x8 a0() const { s16x8 r = {lo0[0], lo0[1], lo0[2], lo0[3], hi0[0], hi0[1], hi0[2], hi0[3]}; return __builtin_bit_cast(bf16x8, r); }
  IA_DEV bf16x8 a1() const { s16x8 r = {lo1[0], lo1[1], lo1[2], lo1[3], hi1[0], hi1[1], hi1[2], hi1[3]}; return __builtin_bit_cast(bf16x8, r); }
};
// rows ROW0 .. ROW0+15 (ROW0 a multiple of 16) below the lane bases
template <int ROW0>
IA_DEV void tr_issue(TrPair& f, const TrBase& b) {
  f.lo0 = ia_tr_read<ROW0 * 128>(b.lo0); f.hi0 = ia_tr_read<ROW0 * 128>(b.hi0);
  f.lo1 = ia_tr_read<ROW0 * 128>(b.lo1); f.hi1 = ia_tr_read<ROW0 * 128>(b.hi1);
}
template <int N>
IA_DEV void tr_wait(TrPair& f) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.lo0), "+v"(f.hi0), "+v"(f.lo1), "+v"(f.hi1) : "n"(N));
}
IA_DEV TrBase tr_shift(const TrBase& b, uint32_t bytes) { return {b.lo0 + bytes, b.hi0 + bytes, b.lo1 + bytes, b.hi1 + bytes}; }

// ds_read_b128 fragment of row `row`, chunk `chunk`
IA_DEV bf16x8 frag(const char* s, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(s + row * 128 + ((chunk ^ fsw(row)) << 4));
}

IA_DEV f32x16 splat16(float v) {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = v;
  return z;
}
#define ACC_ROW(r, hh) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hh))
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

IA_DEV float max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// per 64-key tile, the ballot of attendable keys (in range and not masked)
IA_DEV void build_valid_table(const ResArgs& p, uint32_t* s_valid, size_t rowbase, int L, int nkt, int lane, int wave, int nw) {
  for (int t = wave; t < nkt; t += nw) {
    const int key = t * 64 + lane;
    const bool kv = key < L && (p.mask == nullptr || p.mask[rowbase + key] != 0);
    const uint64_t vb = __ballot(kv);
    if (lane == 0) { s_valid[2 * t] = (uint32_t)vb; s_valid[2 * t + 1] = (uint32_t)(vb >> 32); }
  }
}

struct S2 { f32x16 a, b; };   // scores of one 64-key tile for the wave's 32 queries: keys 0..31 | 32..63

// ---- LDS fragment reads issued by hand.  hipcc places a ds_read right in front of its consumer (read, read, wait, MFMA, wait,
// MFMA ...), which exposes the LDS latency several times per tile; these go out as a block one phase early and are waited for
// with a counted lgkmcnt in front of the MFMAs that use them (LDS returns in order).
template <int OFF>
IA_DEV bf16x8 lds_read128(uint32_t addr) {
  bf16x8 d;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
  return d;
}
struct KFrags { bf16x8 a[4], b[4]; };            // A operands of S^T = K Q^T: rows 0..31 (a) and 32..63 (b) of a tile, 4 k-steps
struct KBase { uint32_t o[4]; };                 // lane byte offsets of the 4 k-step chunks inside a tile (row = lane & 31)
IA_DEV KBase k_base(const void* s, int lq, int hh) {
  KBase k;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) k.o[kb] = lds_addr(s) + (uint32_t)(lq * 128 + (((kb * 2 + hh) ^ fsw(lq)) << 4));
  return k;
}
template <bool FIRST, bool SECOND>
IA_DEV void k_issue(KFrags& f, const KBase& k, uint32_t tile_bytes) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (FIRST) f.a[kb] = lds_read128<0>(k.o[kb] + tile_bytes);
    if (SECOND) f.b[kb] = lds_read128<4096>(k.o[kb] + tile_bytes);
  }
}
// at most N LDS operations issued after these fragments are still outstanding
template <int N>
IA_DEV void k_wait(KFrags& f) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3])
               : "n"(N));
}
struct VFrags { TrPair s[4]; };                  // A operands of O^T += V^T P^T: 4 steps of 16 keys
template <int NKS>
IA_DEV void v_issue(VFrags& f, const TrBase& vt) {
  tr_issue<0>(f.s[0], vt);
  if (NKS > 1) tr_issue<16>(f.s[1], vt);
  if (NKS > 2) tr_issue<32>(f.s[2], vt);
  if (NKS > 3) tr_issue<48>(f.s[3], vt);
}
template <int N>
IA_DEV void v_wait(VFrags& f) {
  asm volatile("s_waitcnt lgkmcnt(%16)"
               : "+v"(f.s[0].lo0), "+v"(f.s[0].hi0), "+v"(f.s[0].lo1), "+v"(f.s[0].hi1), "+v"(f.s[1].lo0), "+v"(f.s[1].hi0), "+v"(f.s[1].lo1),
                 "+v"(f.s[1].hi1), "+v"(f.s[2].lo0), "+v"(f.s[2].hi0), "+v"(f.s[2].lo1), "+v"(f.s[2].hi1), "+v"(f.s[3].lo0), "+v"(f.s[3].hi0),
                 "+v"(f.s[3].lo1), "+v"(f.s[3].hi1)
               : "n"(N));
}

// S^T = K Q^T + C for one tile (C = -m broadcast per query: the lane's 16 rows all belong to its own query)
template <bool BOTH>
IA_DEV void scores(S2& s, const KFrags& f, const bf16x8 (&qf)[4], const f32x16& cinit) {
  s.a = MFMA(f.a[0], qf[0], cinit);
  if (BOTH) s.b = MFMA(f.b[0], qf[0], cinit);
#pragma unroll
  for (int kb = 1; kb < 4; ++kb) {
    s.a = MFMA(f.a[kb], qf[kb], s.a);
    if (BOTH) s.b = MFMA(f.b[kb], qf[kb], s.b);
  }
}

template <bool BOTH>
IA_DEV void mask_scores(S2& s, uint32_t valid_lo, uint32_t valid_hi, int hh) {
  const uint32_t vlo = hh ? valid_lo >> 4 : valid_lo, vhi = hh ? valid_hi >> 4 : valid_hi;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int bit = (r & 3) + 8 * (r >> 2);
    if (!((vlo >> bit) & 1)) s.a[r] = -INFINITY;
    if (BOTH && !((vhi >> bit) & 1)) s.b[r] = -INFINITY;
  }
}

// dropout keep decision for element (q, key) of stream (b, h): the same counter stream as attention.hip
IA_DEV bool drop_keep(uint32_t seed, uint32_t stream, int q, int key, uint32_t thr16) {
  const uint32_t r = ia_rng(seed, stream, (uint32_t)q * 1024u + ((uint32_t)key >> 1));
  const uint32_t u = (key & 1) ? (r >> 16) : (r & 0xFFFFu);
  return u >= thr16;
}

// ------------------------------------------------------------------------------------------ forward
// State of one 32-query chunk: the lazily updated softmax reference m (shared by lanes q and q^32), this lane's half of the
// row sum, and O^T accumulators.
struct FwdState { float m, l; f32x16 negm, o0, o1; };

// softmax of the (already reference-subtracted) scores of one tile -> P^T fragments
template <bool BOTH, bool DROPOUT>
IA_DEV void softmax_tile(const ResArgs& p, S2& s, FwdState& st, bf16x8 (&pf)[4], int hh, int q, int key0, uint32_t stream_id) {
  // four independent max chains (v_max3 through asm: the fmaxf builtin canonicalises every MFMA output first)
  float m0 = max3(s.a[0], s.a[1], s.a[2]), m1 = max3(s.a[3], s.a[4], s.a[5]), m2 = max3(s.a[6], s.a[7], s.a[8]),
        m3 = max3(s.a[9], s.a[10], s.a[11]);
  m0 = max3(m0, s.a[12], s.a[13]); m1 = max3(m1, s.a[14], s.a[15]);
  if (BOTH) {
    m2 = max3(m2, s.b[0], s.b[1]); m3 = max3(m3, s.b[2], s.b[3]); m0 = max3(m0, s.b[4], s.b[5]); m1 = max3(m1, s.b[6], s.b[7]);
    m2 = max3(m2, s.b[8], s.b[9]); m3 = max3(m3, s.b[10], s.b[11]); m0 = max3(m0, s.b[12], s.b[13]); m1 = max3(m1, s.b[14], s.b[15]);
  }
  const float tmax = max3(m0, m1, fmaxf(m2, m3));
  if (__ballot(tmax > RESCALE_THR) != 0ull) {   // wave-uniform, rare: move the reference up to the running maximum
    asm volatile("" ::: "memory");
    const float tm = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float delta = fmaxf(tm, 0.f);
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    st.m += delta;
    st.l *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s.a[r] -= delta; if (BOTH) s.b[r] -= delta;
      st.o0[r] *= alpha; st.o1[r] *= alpha; st.negm[r] -= delta;
    }
  }
  float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;   // four partial row sums: no 32-long dependent chain
#pragma unroll
  for (int r = 0; r < 16; r += 4) {
    s.a[r] = __builtin_amdgcn_exp2f(s.a[r]); s.a[r + 1] = __builtin_amdgcn_exp2f(s.a[r + 1]);
    s.a[r + 2] = __builtin_amdgcn_exp2f(s.a[r + 2]); s.a[r + 3] = __builtin_amdgcn_exp2f(s.a[r + 3]);
    r0 += s.a[r]; r1 += s.a[r + 1]; r2 += s.a[r + 2]; r3 += s.a[r + 3];
    if (BOTH) {
      s.b[r] = __builtin_amdgcn_exp2f(s.b[r]); s.b[r + 1] = __builtin_amdgcn_exp2f(s.b[r + 1]);
      s.b[r + 2] = __builtin_amdgcn_exp2f(s.b[r + 2]); s.b[r + 3] = __builtin_amdgcn_exp2f(s.b[r + 3]);
      r0 += s.b[r]; r1 += s.b[r + 1]; r2 += s.b[r + 2]; r3 += s.b[r + 3];
    }
  }
  st.l += (r0 + r1) + (r2 + r3);
  if (DROPOUT) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int kl = ACC_ROW(r, hh);   // even key, r+1 is the odd neighbour
      const uint32_t ra = ia_rng(p.seed, stream_id, (uint32_t)q * 1024u + (uint32_t)((key0 + kl) >> 1));
      if ((ra & 0xFFFFu) < p.thr16) s.a[r] = 0.f;
      if ((ra >> 16) < p.thr16) s.a[r + 1] = 0.f;
      if (BOTH) {
        const uint32_t rb = ia_rng(p.seed, stream_id, (uint32_t)q * 1024u + (uint32_t)((key0 + 32 + kl) >> 1));
        if ((rb & 0xFFFFu) < p.thr16) s.b[r] = 0.f;
        if ((rb >> 16) < p.thr16) s.b[r + 1] = 0.f;
      }
      if ((r & 3) == 2) __builtin_amdgcn_sched_barrier(0);   // a few hashes in flight at a time: their temporaries otherwise spill
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pf[0][j] = f2bf(s.a[j]); pf[1][j] = f2bf(s.a[8 + j]);
    if (BOTH) { pf[2][j] = f2bf(s.b[j]); pf[3][j] = f2bf(s.b[8 + j]); }
  }
}

// O^T += V^T P^T over the 16-key steps [KS0, NKS) of one tile
template <int NKS, int KS0 = 0>
IA_DEV void pv_tile(FwdState& st, const VFrags& f, const bf16x8 (&pf)[4]) {
#pragma unroll
  for (int ks = KS0; ks < NKS; ++ks) {
    st.o0 = MFMA(f.s[ks].a0(), pf[ks], st.o0);
    st.o1 = MFMA(f.s[ks].a1(), pf[ks], st.o1);
  }
}

// First-pass DMA pacing shared by the three kernels: every tile of the resident operands is requested up front (a CU needs far
// more than a few tiles in flight to stream 150 KB at HBM latency), every wave issues exactly CNT DMA instructions per tile
// (pieces a ragged tile does not have go to a 1 KiB dump area), so "tile t has landed" is s_waitcnt vmcnt(tiles after t * CNT)
// followed by one workgroup barrier.  The immediate comes from a switch: at most RES_MAX_KT - 1 tiles follow.
template <int CNT>
IA_DEV void wait_tiles_after(int after) {   // wave-uniform
#define IA_W(k) case k: asm volatile("s_waitcnt vmcnt(%0)" :: "n"((k) * CNT < 63 ? (k) * CNT : 63) : "memory"); break;
  switch (after) {
    IA_W(1) IA_W(2) IA_W(3) IA_W(4) IA_W(5) IA_W(6) IA_W(7) IA_W(8) IA_W(9) IA_W(10)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef IA_W
}

template <int NW, bool DROPOUT>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_res_kernel(ResArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lq = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x % p.nh, b = blockIdx.x / p.nh;
  const Geo g(p.L);
  const int L = p.L;
  char* sK = smem;
  char* sV = smem + g.Lp * 128;
  char* s_dump = smem + 2 * g.Lp * 128;
  uint32_t* s_valid = reinterpret_cast<uint32_t*>(s_dump + 1024);
  const size_t rowbase = (size_t)b * L;
  const uint32_t stream_id = (uint32_t)(b * p.nh + h);

  build_valid_table(p, s_valid, rowbase, L, g.nkt, lane, wave, NW);
  __syncthreads();
  // first and last tile with an attendable key (nothing outside is loaded or computed) and the tiles that need no selects
  int t_first = g.nkt, t_end = 0;
  uint32_t fullbits = 0;
  for (int t = 0; t < g.nkt; ++t) {
    const uint32_t lo = s_valid[2 * t], hi = s_valid[2 * t + 1];
    if (lo | hi) { if (t < t_first) t_first = t; t_end = t + 1; }
    if ((lo & hi) == 0xFFFFFFFFu) fullbits |= 1u << t;
  }
  t_first = __builtin_amdgcn_readfirstlane(t_first); t_end = __builtin_amdgcn_readfirstlane(t_end);
  fullbits = __builtin_amdgcn_readfirstlane(fullbits);

  const int nchunks = (L + 31) >> 5;
  int chunk = wave;                                // chunks wave, wave + NW, ...
  if (t_end == 0) {   // no attendable key at all (workgroup-uniform): the output is defined as zero
    for (; chunk < nchunks; chunk += NW) {
      const int q = chunk * 32 + lq;
      if (q >= L) continue;
      if (hh == 0 && p.nlse) p.nlse[((size_t)b * p.nh + h) * L + q] = 0.f;
      bf16* op = p.out + (rowbase + q) * p.ld_o + h * 64 + hh * 32;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(op + j * 8) = u32x4{0u, 0u, 0u, 0u};
    }
    return;
  }
  // Q fragments of the first chunk are requested (and waited for) before the DMA so that their wait does not cover it
  bf16x8 qf[4], qn[4];
  auto load_q = [&](bf16x8 (&dst)[4], int c) {
    const int qrow = c * 32 + lq;
    const int qc = qrow < L ? qrow : L - 1;
    const bf16* qp = p.q + (rowbase + qc) * p.ld_q + h * 64 + hh * 8;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) dst[kb] = *reinterpret_cast<const bf16x8*>(qp + kb * 16);
  };
  if (chunk < nchunks) {
    load_q(qf, chunk);
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
  }
  const __amdgpu_buffer_rsrc_t rsK = ia_rsrc(p.k, p.kv_bytes);
  const __amdgpu_buffer_rsrc_t rsV = ia_rsrc(p.v, p.kv_bytes);
  constexpr int CNT = 2 * (8 / NW);
  auto issue_tile = [&](int t) {
    const int tg = g.tile_groups(t);
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      const int j = wave + i * NW;
      if (j < tg) {
        stage8(rsK, sK, rowbase, L, p.ld_kv, h * 64, t * 8 + j, lane);
        stage8(rsV, sV, rowbase, L, p.ld_kv, h * 64, t * 8 + j, lane);
      } else {   // keeps the per-tile instruction count uniform: zeros into the dump area
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, IA_LDS(s_dump), 16, OOB, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, IA_LDS(s_dump), 16, OOB, 0, 0, 0);
      }
    }
  };
  if (!(p.dbg & 128)) for (int t = t_first; t < t_end; ++t) issue_tile(t);
  int landed = t_first;                            // tiles [t_first, landed) are known to be in LDS
  // every wave passes here exactly once per tile, in tile order
  auto need_tile = [&](int t) {
    while (landed <= t) {
      wait_tiles_after<CNT>(t_end - 1 - landed);
      __builtin_amdgcn_s_barrier();
      ++landed;
    }
  };

  const KBase kbase = k_base(sK, lq, hh);
  const TrBase vbase = tr_base(sV, lane);
  // a ragged last tile of at most 32 keys runs with half the score MFMAs and only the 16-key steps that hold keys
  const bool ragged = g.tail != 0 && t_end == g.nkt;
  const bool half_tail = ragged && g.tail <= 32;
  const int t_body_end = half_tail ? t_end - 1 : t_end;     // tiles [t_first, t_body_end) take the full-width path
  auto valid_lo = [&](int t) { return (uint32_t)__builtin_amdgcn_readfirstlane(s_valid[2 * t]); };
  auto valid_hi = [&](int t) { return (uint32_t)__builtin_amdgcn_readfirstlane(s_valid[2 * t + 1]); };

  if ((p.dbg & 2) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  while (chunk < nchunks) {
    const int q = chunk * 32 + lq;
    const int next_chunk = chunk + NW;
    if ((p.dbg & 1) && wave >= NW / 2 && landed >= t_end) __builtin_amdgcn_s_sleep(8);
    if ((p.dbg & 4) && wave >= NW / 2 && landed >= t_end) __builtin_amdgcn_s_sleep(4);
    FwdState st;
    st.l = 0.f; st.o0 = splat16(0.f); st.o1 = splat16(0.f);
    S2 s;
    KFrags kf;
    VFrags vf;
    bf16x8 pf[4];
    int t = t_first;
    need_tile(t);
    if (t < t_body_end) {
      // ---- first tile: establishes the reference
      k_issue<true, true>(kf, kbase, (uint32_t)t * 8192u);
      k_wait<0>(kf);
      scores<true>(s, kf, qf, splat16(0.f));
      v_issue<4>(vf, tr_shift(vbase, (uint32_t)t * 8192u));
      mask_scores<true>(s, valid_lo(t), valid_hi(t), hh);
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s.a[r], s.b[r]));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      st.m = tmax;
      st.negm = splat16(-tmax);
#pragma unroll
      for (int r = 0; r < 16; ++r) { s.a[r] -= tmax; s.b[r] -= tmax; }
      // ---- steady state: tile t's scores are in s and its V^T fragments in flight.  Nothing conditional touches the
      // accumulators inside the loop (a second definition of O or S under a branch costs a 32-register copy per tile).
      for (; t + 1 < t_body_end; ++t) {
        if (!(p.dbg & 8)) softmax_tile<true, DROPOUT>(p, s, st, pf, hh, q, t * 64, stream_id);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { pf[0][j] = f2bf(s.a[j]); pf[1][j] = f2bf(s.a[8 + j]); pf[2][j] = f2bf(s.b[j]); pf[3][j] = f2bf(s.b[8 + j]); }
        }
        need_tile(t + 1);
        if (!(p.dbg & 64)) v_wait<0>(vf);
        // the next tile's K fragments land under the P V MFMAs (second half into the registers the first two steps free)
        if (!(p.dbg & 64)) k_issue<true, false>(kf, kbase, (uint32_t)(t + 1) * 8192u);
        if (!(p.dbg & 16)) pv_tile<2>(st, vf, pf);
        if (!(p.dbg & 64)) k_issue<false, true>(kf, kbase, (uint32_t)(t + 1) * 8192u);
        if (!(p.dbg & 16)) pv_tile<4, 2>(st, vf, pf);
        else asm volatile("" :: "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]));
        if (!(p.dbg & 64)) k_wait<0>(kf);
        if (!(p.dbg & 32)) scores<true>(s, kf, qf, st.negm);
        if (!(p.dbg & 64)) v_issue<4>(vf, tr_shift(vbase, (uint32_t)(t + 1) * 8192u));      // lands under the softmax
        if (!((fullbits >> (t + 1)) & 1)) { asm volatile("" ::: "memory"); mask_scores<true>(s, valid_lo(t + 1), valid_hi(t + 1), hh); }
      }
      // ---- last full-width tile (possibly ragged with 33..63 keys); the next chunk's Q goes into registers the K fragments
      // no longer need
      if (next_chunk < nchunks) load_q(qn, next_chunk);
      softmax_tile<true, DROPOUT>(p, s, st, pf, hh, q, t * 64, stream_id);
      v_wait<0>(vf);
      if (t < g.nfull || g.tail > 48) pv_tile<4>(st, vf, pf);
      else pv_tile<3>(st, vf, pf);              // its last 16 rows hold no key
      ++t;
    }
    if (half_tail) {
      need_tile(t);
      k_issue<true, false>(kf, kbase, (uint32_t)t * 8192u);
      v_issue<2>(vf, tr_shift(vbase, (uint32_t)t * 8192u));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf.a[0]), "+v"(kf.a[1]), "+v"(kf.a[2]), "+v"(kf.a[3]));
      v_wait<0>(vf);
      if (t == t_first) {      // the ragged tile is the only one
        scores<false>(s, kf, qf, splat16(0.f));
        mask_scores<false>(s, valid_lo(t), 0u, hh);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s.a[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        st.m = tmax;
        st.negm = splat16(-tmax);
#pragma unroll
        for (int r = 0; r < 16; ++r) s.a[r] -= tmax;
      } else {
        scores<false>(s, kf, qf, st.negm);
        mask_scores<false>(s, valid_lo(t), 0u, hh);
      }
      softmax_tile<false, DROPOUT>(p, s, st, pf, hh, q, t * 64, stream_id);
      if (g.tail > 16) pv_tile<2>(st, vf, pf); else pv_tile<1>(st, vf, pf);
    }
    // ---- normalise and store
    const float l_tot = st.l + __shfl_xor(st.l, 32, 64);
    const float inv = l_tot > 0.f ? p.inv_keep / l_tot : 0.f;
    if (q < L && !(p.dbg & 512)) {
      if (hh == 0 && p.nlse) p.nlse[((size_t)b * p.nh + h) * L + q] = -(st.m + __builtin_amdgcn_logf(l_tot));
      bf16* op = p.out + (rowbase + q) * p.ld_o + h * 64;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 8 * rg + 4 * hh;
        bf16x4 a = {f2bf(st.o0[rg * 4] * inv), f2bf(st.o0[rg * 4 + 1] * inv), f2bf(st.o0[rg * 4 + 2] * inv), f2bf(st.o0[rg * 4 + 3] * inv)};
        bf16x4 c = {f2bf(st.o1[rg * 4] * inv), f2bf(st.o1[rg * 4 + 1] * inv), f2bf(st.o1[rg * 4 + 2] * inv), f2bf(st.o1[rg * 4 + 3] * inv)};
        *reinterpret_cast<bf16x4*>(op + d) = a;
        *reinterpret_cast<bf16x4*>(op + 32 + d) = c;
      }
    }
    chunk = next_chunk;
    if (chunk < nchunks) {
      if (t_first < t_body_end) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) qf[kb] = qn[kb];
      } else {
        load_q(qf, chunk);
      }
    }
  }
  // a wave that ran out of chunks before the first pass finished still owes the workgroup its barriers (and DMA pieces)
  need_tile(t_end - 1);
}

int fill_args(ResArgs& a, int B, int nh, int L, int ld_q, int ld_kv, int ld_o, float scale, float drop_p, uint32_t seed) {
  if (B <= 0 || nh <= 0 || L <= 0 || (ld_q & 7) || (ld_kv & 7) || (ld_o & 7)) return IA_ERR_ARG;
  if (L > RES_MAX_L) return IA_ERR_UNSUPPORTED;
  if (ld_q < nh * 64 || ld_kv < nh * 64 || ld_o < nh * 64) return IA_ERR_ARG;
  const uint64_t rows = (uint64_t)B * L;
  const uint64_t qb = rows * ld_q * 2, kb = rows * ld_kv * 2, ob = rows * ld_o * 2;
  if (qb >= 0x7FFFFFFFull || kb >= 0x7FFFFFFFull || ob >= 0x7FFFFFFFull) return IA_ERR_ARG;
  a.B = B; a.nh = nh; a.L = L; a.ld_q = ld_q; a.ld_kv = ld_kv; a.ld_o = ld_o; a.ld_dq = ld_q; a.ld_dkv = ld_kv;
  a.q_bytes = (uint32_t)(qb - (uint64_t)(ld_q - nh * 64) * 2);
  a.kv_bytes = (uint32_t)(kb - (uint64_t)(ld_kv - nh * 64) * 2);
  a.o_bytes = (uint32_t)ob;
  a.scale = scale;
  a.thr16 = drop_p > 0.f ? (uint32_t)(drop_p * 65536.f + 0.5f) : 0u;
  a.inv_keep = drop_p > 0.f ? 1.f / (1.f - (float)a.thr16 / 65536.f) : 1.f;
  a.seed = seed;
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("IA_ATTN_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
  return IA_OK;
}

size_t res_lds_bytes(int L) {
  const int nfull = L >> 6, tail = L & 63;
  const int Lp = nfull * 64 + ((tail + 31) & ~31);
  return (size_t)2 * Lp * 128 + 1024 + 2 * RES_MAX_KT * 4;
}

template <typename K>
int set_lds(K kern, size_t bytes) {
  return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? IA_OK : IA_ERR_LAUNCH;
}

}  // namespace

// Resident-K/V forward.  q must already be multiplied by softmax_scale * log2(e) (IA_ATTN_QSCALE(scale)); nlse2 receives the
// negated log2-domain log-sum-exp.  IA_ERR_UNSUPPORTED when L > 640: use ia_attn_fwd.
extern "C" int ia_attn_fwd_res(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, void* out, int ld_o,
                               float* nlse2, int B, int nh, int L, float drop_p, uint32_t seed, hipStream_t stream) {
  (void)hipGetLastError();
  if (!q || !k || !v || !out) return IA_ERR_ARG;
  ResArgs a{};
  int rc = fill_args(a, B, nh, L, ld_qkv, ld_qkv, ld_o, 0.f, drop_p, seed);
  if (rc) return rc;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out; a.mask = key_mask; a.nlse = nlse2;
  const size_t lds = res_lds_bytes(L);
  dim3 grid(B * nh);
  const bool small = lds <= 80 * 1024;       // two workgroups of four waves per CU instead of one of eight
#define IA_LAUNCH_FWD(NW, DROP)                                                                        \
  do {                                                                                                 \
    auto kern = attn_fwd_res_kernel<NW, DROP>;                                                         \
    if ((rc = set_lds(kern, lds))) return rc;                                                          \
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, a);                                     \
  } while (0)
  if (small) { if (a.thr16) IA_LAUNCH_FWD(4, true); else IA_LAUNCH_FWD(4, false); }
  else { if (a.thr16) IA_LAUNCH_FWD(8, true); else IA_LAUNCH_FWD(8, false); }
#undef IA_LAUNCH_FWD
  return ia_check_launch();
}
