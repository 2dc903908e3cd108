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
#include <type_traits>
#include <cstdlib>

// flag bits of ia_attn_bwd_bias_ex (include/itemalign.h; this file does not include the public header)
#define IA_ATTN_Q_PRESCALED 1
#define IA_ATTN_MASKED_ROWS_DEAD 2

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
  int B, nh, Lq, Lk;                              // queries / keys per sequence (self-attention: Lq == Lk)
  const int* cu;                                  // packed (unpadded) self-attention: sequence b owns rows cu[b] .. cu[b+1] and
                                                  // Lq == Lk is the longest sequence (grid size, stride of lse2 / delta); else null
  int ld_q, ld_kv, ld_o, ld_dq, ld_dkv;           // row strides in elements (q & o rows: b*Lq + i; k & v rows: b*Lk + j)
  uint32_t q_bytes, kv_bytes, o_bytes;
  float sc;                                       // softmax scale * log2(e)
  float scale;                                    // softmax scale
  int q_prescaled;                                // q rows already hold q * sc rounded to bf16 (ia_gemm_bf16_qscale): the round-3/4
                                                  // kernels skip their own pre-scaling; dq stays dL/dq of the UNSCALED q
  uint32_t thr16; float inv_keep; uint32_t seed;
  float* cs_part;                                 // backward, optional: [b*ntile + tile][3*nh*64] fp32 column sums of this workgroup's
                                                  // dq | dk | dv rows (the QKV bias gradient, summed over rows by ia_sum_rows_f32); null = off
  int exact_delta;                                // backward, IA_ATTN_EXACT_DELTA=1: `delta` already holds sum_k P dP in fp32 (attn_bwd3_delta_kernel)
  int dead_queries;                               // backward (fused kernel): the caller guarantees d_o == 0 at every masked position, so a 32-query
                                                  // block whose positions are all masked contributes nothing: it is skipped, its dq rows are zeros
};

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): renumber them so that consecutive
// work items -- the 128-row tiles of one (sequence, head), which all stream the same K/V -- share an XCD.
IA_DEV void attn_block_coords(const AttnArgs& p, int len, int& tile, int& h, int& b) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int per = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
#ifdef IA_NO_XCD
  int w = bid;
#else
  int w = (xcd < r ? xcd * (per + 1) : r * (per + 1) + (xcd - r) * per) + idx;
#endif
  const int nt = (len + 127) >> 7;
  tile = w % nt; w /= nt;
  h = w % p.nh; b = w / p.nh;
}

// the same with `rows` queries (keys) per workgroup instead of 128
IA_DEV void attn_block_coords_n(const AttnArgs& p, int len, int rows, int& tile, int& h, int& b) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int per = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  int w = (xcd < r ? xcd * (per + 1) : r * (per + 1) + (xcd - r) * per) + idx;
  const int nt = (len + rows - 1) / rows;
  tile = w % nt; w /= nt;
  h = w % p.nh; b = w / p.nh;
}

// Per 64-key tile, the ballot of attendable keys (in range and not masked), built once per workgroup: a mask byte
// fetched inside the tile loop would make the wave wait for the next tile's LDS-DMA as well (vmcnt is in-order).
constexpr int MAX_KT = 32;   // L <= 2048
IA_DEV void build_valid_table(const AttnArgs& p, uint32_t (*s_valid)[2], size_t rowbase, int L, int lane, int wave) {
  const int nkt = (L + 63) >> 6;
  for (int t = wave; t < nkt; t += 4) {
    const int key = t * 64 + lane;
    const bool kv = key < L && (p.mask == nullptr || p.mask[rowbase + key] != 0);
    const uint64_t vb = __ballot(kv);
    if (lane == 0) { s_valid[t][0] = (uint32_t)vb; s_valid[t][1] = (uint32_t)(vb >> 32); }
  }
}

// K tile / Q tile read with ds_read_b128 (row = key or query, 128 B rows, 16 B chunk XOR (row>>1)&7)
IA_DEV int swz_b128(int row) { return (row >> 1) & 7; }   // two 128-byte rows span the 64 banks: conflict-free for any 16 consecutive rows
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

// A wave's own 32 rows x 64 columns (this head's slice of q / k / v / o / dO rows row0 .. row0+31) into a wave-private 4 KiB LDS
// slot in the ds_read_b128 layout of frag_b128: four 16-byte LDS-DMA issues, each moving 8 whole 128-byte rows.  (Loaded straight into
// the fragment registers, every lane pair fetches 32 bytes of 32 different rows per instruction; that cost the forward kernel 15 %.)
// Rows >= nvalid arrive as zeros.  The reader waits vmcnt(0) itself; no workgroup barrier is involved.
IA_DEV void stage_rows32(__amdgpu_buffer_rsrc_t rs, char* slot, size_t row0, int nvalid, int ld, int col0, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_b128(row);
    uint32_t off = (uint32_t)(((row0 + row) * ld + col0 + c * 8) * 2);
    if (row >= nvalid) off = OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, IA_LDS(slot + i * 1024), 16, off, 0, 0, 0);
  }
}

IA_DEV bf16x8 frag_b128(const char* s, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(s + row * 128 + ((chunk ^ swz_b128(row)) << 4));
}

// A^T fragment for MFMA 32x32x16 out of a row-major [row][64] tile: lane (i = lane&31 -> column,
// half = lane>>5) gets rows row0 + {0..3, 8..11} + 4*half of column col0 + i  (ds_read_b64_tr_b16 x2).
// The reads go through ia_tr_read (inline asm, see common.h), so every use sits behind tr_wait<N>().
IA_DEV uint32_t lds_addr(const void* p) { return ia_lds_addr(p); }

// byte offset of this lane's element inside a tile for column block col0 (0 or 32), row block 0
IA_DEV uint32_t tr_lane_off(int lane, int col0) {
  const int p = lane & 15, G = lane >> 4;
  const int row = 4 * (G >> 1) + (p >> 2);
  const int col = col0 + 16 * (G & 1) + (p & 3) * 4;
  return (uint32_t)(row * 128 + ((((col >> 3) ^ swz_tr(row))) << 4) + (col & 7) * 2);
}

template <int OFF>
IA_DEV s16x4 tr_read(uint32_t base) { return ia_tr_read<OFF>(base); }

struct TrPair {   // the two A^T fragments (columns 0..31 and 32..63) of one 16-row step
  s16x4 lo0, hi0, lo1, hi1;
  IA_DEV bf16x8 a0() const { s16x8 r = {lo0[0], lo0[1], lo0[2], lo0[3], hi0[0], hi0[1], hi0[2], hi0[3]}; return __builtin_bit_cast(bf16x8, r); }
  IA_DEV bf16x8 a1() const { s16x8 r = {lo1[0], lo1[1], lo1[2], lo1[3], hi1[0], hi1[1], hi1[2], hi1[3]}; return __builtin_bit_cast(bf16x8, r); }
};
// rows ROW0 .. ROW0+15 (ROW0 a multiple of 16) of the tile whose lane bases are b0 / b1 (tr_lane_off for col0 = 0 / 32)
template <int ROW0>
IA_DEV void tr_issue(TrPair& f, uint32_t b0, uint32_t b1) {
  f.lo0 = tr_read<ROW0 * 128>(b0); f.hi0 = tr_read<ROW0 * 128 + 1024>(b0);
  f.lo1 = tr_read<ROW0 * 128>(b1); f.hi1 = tr_read<ROW0 * 128 + 1024>(b1);
}
// wait until at most N LDS operations issued after this pair are still outstanding (LDS returns in order)
template <int N>
IA_DEV void tr_wait(TrPair& f) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.lo0), "+v"(f.hi0), "+v"(f.lo1), "+v"(f.hi1) : "n"(N));
}

IA_DEV f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// row index inside a 32-row accumulator block for register r of lane-half hh
#define ACC_ROW(r, hh) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hh))

// dropout: element (q, key) of stream (b, h) is kept iff its 16-bit draw >= thr16; the draw is the low (even key) or high (odd key)
// half of ia_rng_pair(row key of q, (key >> 1) * IA_RNG_PAIR_C) (common.h).  The three kernels assemble the pair constant from parts
// they have for free (lane, tile, register index), so the loops carry no integer multiply beyond the one inside the mix.
IA_DEV uint32_t pair_c_of(int key) { return ((uint32_t)key >> 1) * IA_RNG_PAIR_C; }

// Epilogue store of a wave's 32 x 64 bf16 block that sits in two transposed accumulator blocks (a0: columns d = 0..31, a1: d = 32..63
// of the output row this lane's column lane&31 stands for).  Written straight from the accumulators a lane owns 8-byte pieces of 32
// different rows (a row stride apart): 16 partial-line requests per lane pair and store instruction, which made the store tail a
// third of the forward kernel at L = 255.  Staged through a wave-private LDS slot (32 rows x 144 B: the 16-byte pad keeps the
// 8-byte writes at two lanes per bank) the block leaves as whole 128-byte rows, 16 B per lane, 8 rows per instruction.
constexpr int EPI_ROW = 144, EPI_SLOT = 32 * EPI_ROW;
// cs_out != null (LDS): the 64 column sums of the block's stored rows (of the bf16 values as stored) go to cs_out[0..63] -- each lane adds
// up its 8 columns over its 4 rows, lanes with equal lane&7 are folded with row_ror / permlane swaps (the GEMM column-sum epilogue's
// pattern); the kernel adds the four waves' rows up behind a barrier and writes one row of the partial-sum matrix per workgroup.
IA_DEV void store_block_rows(char* slot, const f32x16& a0, const f32x16& a1, float mul, bool zero, bf16* out, size_t ld, int nrows, int lane,
                             float* cs_out = nullptr) {
  const int hh = lane >> 5, lq = lane & 31;
  char* w = slot + lq * EPI_ROW + hh * 8;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    bf16x4 a, c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // a masked column may hold inf / nan: select, do not multiply
      a[j] = f2bf(zero ? 0.f : a0[rg * 4 + j] * mul);
      c[j] = f2bf(zero ? 0.f : a1[rg * 4 + j] * mul);
    }
    *reinterpret_cast<bf16x4*>(w + rg * 16) = a;
    *reinterpret_cast<bf16x4*>(w + 64 + rg * 16) = c;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private slot: the wave's own LDS writes are in order, no barrier
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + (lane >> 3), c = lane & 7;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(slot + row * EPI_ROW + c * 16);
    if (row < nrows) {
      *reinterpret_cast<bf16x8*>(out + (size_t)row * ld + c * 8) = v;
      if (cs_out) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += bf2f(v[j]);
      }
    }
  }
  if (cs_out) {                                     // wave-uniform
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = cs[j];
      x += ia_dpp<0x128>(x);                        // row_ror 8: lane ^ 8
      cs[j] = ia_add_xor32(ia_add_xor16(x));
    }
    if (lane < 8) {
      *reinterpret_cast<f32x4*>(cs_out + lane * 8) = f32x4{cs[0], cs[1], cs[2], cs[3]};
      *reinterpret_cast<f32x4*>(cs_out + lane * 8 + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
    }
  }
}
// a wave without rows (past the end of the sequence) still contributes its (zero) row
IA_DEV void zero_cs_row(float* cs_out, int lane) {
  if (lane < 16) *reinterpret_cast<f32x4*>(cs_out + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// One 64-key tile of the forward pass for this wave's 32 queries (S^T orientation, see the header comment).
//
// Online softmax with a lazily updated reference: probabilities are formed as exp2(s*sc - m_ref) against the
// reference m_ref the lane already holds, and m_ref only moves (with the O / l rescale that implies) when some
// row of the wave would exceed 2^RESCALE_THR or has nothing accumulated yet. The result is the same softmax
// (any reference cancels in O / l); the common tile costs fma + max + exp2 per score and no cross-lane traffic.
// m_ref is shared by the two lanes of a query (lane, lane^32); l_run is this lane's half of the row sum.
constexpr float RESCALE_THR = 8.f;

template <int BUF, bool DROPOUT>
IA_DEV void fwd_tile(const AttnArgs& p, const char* smem, const bf16x8 (&qf)[4], uint32_t valid_lo, uint32_t valid_hi,
                     float& m_ref, float& l_run, f32x16& o0, f32x16& o1, int lane, int q, int kt, uint32_t rk) {
  const int hh = lane >> 5, lq = lane & 31;
  const char* sK = smem + BUF * 16384;
  f32x16 s0 = zero16(), s1 = zero16();
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const bf16x8 k0 = frag_b128(sK, lq, kb * 2 + hh);
    const bf16x8 k1 = frag_b128(sK, 32 + lq, kb * 2 + hh);
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[kb], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[kb], s1, 0, 0, 0);
  }
  // V^T fragments of the first two 16-key steps land while the softmax runs
  const uint32_t vb0 = lds_addr(smem) + tr_lane_off(lane, 0), vb1 = lds_addr(smem) + tr_lane_off(lane, 32);
  constexpr int VOFF = (BUF * 16384 + 8192) / 128;   // tr_issue takes its offset in 128-byte rows
  TrPair va, vb;
  tr_issue<VOFF>(va, vb0, vb1);
  tr_issue<VOFF + 16>(vb, vb0, vb1);
  if ((valid_lo & valid_hi) != 0xFFFFFFFFu) {   // wave-uniform: only a ragged / padded tile pays for the selects
    asm volatile("" ::: "memory");   // keeps hipcc from flattening this branch into 64 always-executed selects
    const uint32_t vlo = hh ? valid_lo >> 4 : valid_lo, vhi = hh ? valid_hi >> 4 : valid_hi;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bit = (r & 3) + 8 * (r >> 2);
      if (!((vlo >> bit) & 1)) s0[r] = -INFINITY;
      if (!((vhi >> bit) & 1)) s1[r] = -INFINITY;
    }
  }
  const float neg_m = -m_ref;
  float tmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s0[r] = __builtin_fmaf(s0[r], p.sc, neg_m);
    s1[r] = __builtin_fmaf(s1[r], p.sc, neg_m);
    tmax = fmaxf(tmax, fmaxf(s0[r], s1[r]));
  }
  if (__ballot(tmax > RESCALE_THR || l_run == 0.f) != 0ull) {   // wave-uniform, rare after the first tile
    const float tm = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const bool fresh = (l_run + __shfl_xor(l_run, 32, 64)) == 0.f;   // nothing accumulated: re-basing is free
    const float delta = (tm == -INFINITY) ? 0.f : (fresh ? tm : fmaxf(tm, 0.f));
    const float alpha = fresh ? 0.f : __builtin_amdgcn_exp2f(-delta);
    m_ref += delta;
    l_run *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] -= delta; s1[r] -= delta; o0[r] *= alpha; o1[r] *= alpha; }
  }
  float rs = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s0[r] = __builtin_amdgcn_exp2f(s0[r]);
    s1[r] = __builtin_amdgcn_exp2f(s1[r]);
    rs += s0[r] + s1[r];
  }
  l_run += rs;
  if (DROPOUT) {
    // key = kt*64 + ACC_ROW(r, hh) (+ 32 for the second block): pair constant = tile part (scalar) + lane part (hh) + immediate
    const uint32_t tile_c = (uint32_t)(kt * 32) * IA_RNG_PAIR_C + (uint32_t)(2 * hh) * IA_RNG_PAIR_C;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      constexpr uint32_t C = IA_RNG_PAIR_C;
      const uint32_t imm = (uint32_t)(((r & 3) >> 1) + 4 * (r >> 2)) * C;      // (ACC_ROW(r, 0) >> 1) * C, r even
      const uint32_t ra = ia_rng_pair(rk, tile_c + imm);
      const uint32_t rb = ia_rng_pair(rk, tile_c + imm + 16u * C);
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
  tr_wait<4>(va);
  o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va.a0(), pf[0], o0, 0, 0, 0);
  o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va.a1(), pf[0], o1, 0, 0, 0);
  TrPair vc, vd;
  tr_issue<VOFF + 32>(vc, vb0, vb1);
  tr_wait<4>(vb);
  o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb.a0(), pf[1], o0, 0, 0, 0);
  o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb.a1(), pf[1], o1, 0, 0, 0);
  tr_issue<VOFF + 48>(vd, vb0, vb1);
  tr_wait<4>(vc);
  o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc.a0(), pf[2], o0, 0, 0, 0);
  o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc.a1(), pf[2], o1, 0, 0, 0);
  tr_wait<0>(vd);
  o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vd.a0(), pf[3], o0, 0, 0, 0);
  o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vd.a1(), pf[3], o1, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------ forward
template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
  // K | V tile, double buffered, then the valid-key table. One __shared__ object only: with a second one hipcc
  // drains vmcnt(0) before every LDS read while a DMA is in flight.
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384 + MAX_KT * 8];
  uint32_t (*s_valid)[2] = reinterpret_cast<uint32_t (*)[2]>(smem + 2 * 16384);
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lq = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, h, b;
  attn_block_coords(p, p.Lq, tile, h, b);
  int Lq = p.Lq, L = p.Lk;                        // L: keys
  size_t qbase = (size_t)b * Lq, rowbase = (size_t)b * L;
  if (p.cu) {                                     // packed rows: this sequence's own length and first row
    const int s0 = p.cu[b];
    Lq = L = p.cu[b + 1] - s0;
    qbase = rowbase = (size_t)s0;
    if (tile * 128 >= Lq) return;                 // block-uniform, before any barrier
  }
  const int q0 = tile * 128 + wave * 32;
  const bool active = q0 < Lq;
  const int q = q0 + lq;

  const __amdgpu_buffer_rsrc_t rsK = ia_rsrc(p.k, p.kv_bytes);
  const __amdgpu_buffer_rsrc_t rsV = ia_rsrc(p.v, p.kv_bytes);

  bf16x8 qf[4];
  float m_ref = 0.f, l_run = 0.f;
  f32x16 o0 = zero16(), o1 = zero16();
  const int nkt = (L + 63) >> 6;
  const uint32_t rk = DROPOUT ? ia_rng_row(p.seed, (uint32_t)(b * p.nh + h), (uint32_t)q) : 0u;      // row key of this lane's query
  auto prefetch = [&](int buf, int kt) {
    if (kt < nkt) {
      char* nb = smem + buf * 16384;
      stage64<false>(rsK, nb, rowbase + kt * 64, L - kt * 64, p.ld_kv, h * 64, tid, wave);
      stage64<true>(rsV, nb + 8192, rowbase + kt * 64, L - kt * 64, p.ld_kv, h * 64, tid, wave);
    }
  };
  auto compute = [&](auto BUF, int kt) {
    if (active) {
      const uint32_t valid_lo = __builtin_amdgcn_readfirstlane(s_valid[kt][0]);
      const uint32_t valid_hi = __builtin_amdgcn_readfirstlane(s_valid[kt][1]);
      // a tile with no attendable key contributes nothing
      if ((valid_lo | valid_hi) != 0u)
        fwd_tile<decltype(BUF)::value, DROPOUT>(p, smem, qf, valid_lo, valid_hi, m_ref, l_run, o0, o1, lane, q, kt, rk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  // this wave's 32 query rows travel through its 4 KiB of the second ring slot (free until tile 1 is prefetched after the barrier)
  char* qslot = smem + 16384 + wave * 4096;
  stage_rows32(ia_rsrc(p.q, p.q_bytes), qslot, qbase + q0, Lq - q0, p.ld_q, h * 64, lane);
  prefetch(0, 0);
  build_valid_table(p, s_valid, rowbase, L, lane, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) qf[kb] = frag_b128(qslot, lq, kb * 2 + hh);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nkt; kt += 2) {
    prefetch(1, kt + 1);
    compute(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nkt) {
      prefetch(0, kt + 2);
      compute(std::integral_constant<int, 1>{}, kt + 1);
    }
  }
  if (!active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? p.inv_keep / l_tot : 0.f;
  if (q < Lq && hh == 0 && p.lse2) p.lse2[((size_t)b * p.nh + h) * p.Lq + q] = m_ref + __builtin_amdgcn_logf(l_tot);
  // the K/V ring is free after the loop's last barrier: wave-private staging slots at its start
  store_block_rows(smem + wave * EPI_SLOT, o0, o1, inv, false, p.out + (qbase + q0) * p.ld_o + h * 64, p.ld_o, Lq - q0, lane);
}

// ------------------------------------------------------------------------------------------ forward, round 3
// Same arithmetic and tile geometry (4 waves x 32 queries, 64-key tiles, S^T orientation) as attn_fwd_kernel above, rebuilt around
// what tools/abl/issue_model.hip and the PMC passes of round 3 measured on an MI355X (profiles/r03_issue_model.txt, r03_pmc_mfma.csv):
// at head dim 64 the kernel is bound by instruction ISSUE -- the waves of a SIMD spent their time issuing ~155 VALU + ~90 SALU
// instructions per 16 MFMAs -- so the loop is stripped to what the arithmetic needs:
//  * softmax = 32 exp2 + 16 packed adds + 16 packed converts per tile.  The query fragments are pre-scaled by scale*log2(e) once per
//    block, so an accumulator IS exp2's argument; no running maximum is taken: the reference m_ref starts at 0 and only moves when a
//    row sum leaves [2^-100, 2^60] (rebase(): the block's scores recomputed from the K fragments still in registers, true maximum
//    taken) -- any reference cancels in O / l, and exp2 arguments stay far inside the fp32 exponent range;
//  * masked / out-of-range keys and a moved reference enter through ONE extra MFMA per 32-key block (A = (penalty, 1) in k-slots 0 / 1
//    of the key's row, B = (1, -m_ref) in k-slots 0 / 1 of the query's column) instead of 64 compare + select pairs (v_cmp 8.6 cycles,
//    the v_cndmask behind it 4.8); tiles whose 64 keys are all attendable (a bit mask in an SGPR) start from the constant 0;
//  * no address arithmetic in the loop: the K / V windows are per-sequence buffer descriptors (rows past the sequence read as zero by
//    the bounds check), the tile is selected by the DMA's scalar offset, the ring slots are compile-time constants (loop unrolled by
//    the ring depth 3) and every LDS read uses a lane-constant base register + an immediate;
//  * all fragments are in registers before the MFMAs that use them (asm reads, counted lgkmcnt): the K fragments of tile t+1 and the
//    V^T fragments of tile t arrive under the PV MFMAs / the conversions of tile t;
//  * the DMA of tile t+2 is issued at the top of tile t and only K(t+1) is waited for there (counted vmcnt), one raw s_barrier per
//    tile; QB = 1 (32 queries per wave): 48 KiB of LDS and <= 168 VGPRs, three workgroups per CU; QB = 2 (64 queries per wave, half
//    the K / V traffic per MFMA): 64 KiB and <= 251 VGPRs, two per CU -- launch_fwd picks by how full the last 256-query block is.
#ifndef IA_F3_PRESCALE
#define IA_F3_PRESCALE 1     // 1: Q fragments carry scale * log2(e) (one more bf16 rounding of q, no multiply per score); 0: exp2(s * sc)
#endif
namespace fwd3 {
constexpr bool PRESCALE = IA_F3_PRESCALE != 0;
// LDS: K0 K1 | V0 V1 | K2 V2 | valid-key table.  Q is staged in K2 + V2 (first filled behind the first barrier), the epilogue rows in
// K0.. (behind a last barrier).
constexpr int k_slot(int s) { return s == 2 ? 32768 : s * 8192; }
constexpr int v_slot(int s) { return s == 2 ? 40960 : 16384 + s * 8192; }
constexpr int Q_OFF = 32768, EPI_OFF = 0, TAB_OFF = 49152, SMEM = TAB_OFF + MAX_KT * 8;
constexpr float L_HI = 1.1529215e18f;                                             // 2^60
constexpr uint32_t L_LO_BITS = 0x0D800000u, L_HI_BITS = 0x5D800000u;              // bits of 2^-100, 2^60
constexpr float NEG_BIG = -1e30f;
constexpr uint32_t NEG_BIG_BF16 = 0xF14Au;                                        // bf16(-1e30)

template <int OFF>
IA_DEV bf16x8 lds_read_b128(uint32_t addr) {
  bf16x8 d;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
  return d;
}
// wait until at most N LDS operations issued after these eight fragments are outstanding
template <int N>
IA_DEV void frag_wait(bf16x8 (&f)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(N));
}
IA_DEV float swap32(float x) {      // the partner lane's (lane ^ 32) value
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return (threadIdx.x & 32) ? a : b;
}
IA_DEV f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

struct Lane {          // lane constants (absolute LDS byte addresses of ring slot 0)
  uint32_t ka[4];      // this lane's K fragment (row lq, k-step kb); the second 32-key block is +4096
  uint32_t v0, v1;     // transpose-read bases of the V tile (d 0..31 / 32..63)
  uint32_t dk0, dk1, dv0, dv1;      // DMA: byte offset of this lane's 16 bytes inside a K / V tile of the sequence (issue 0 / 1)
};

// all eight K fragments of the tile in ring slot SLOT (kf[kb * 2 + blk])
template <int SLOT>
IA_DEV void read_k(bf16x8 (&kf)[8], const Lane& ln) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    kf[kb * 2] = lds_read_b128<k_slot(SLOT)>(ln.ka[kb]);
    kf[kb * 2 + 1] = lds_read_b128<k_slot(SLOT) + 4096>(ln.ka[kb]);
  }
}
// rows ROW0 .. ROW0+15 of the V tile in ring slot SLOT
template <int SLOT, int ROW0>
IA_DEV void read_v(TrPair& f, const Lane& ln) {
  constexpr int O = v_slot(SLOT) + ROW0 * 128;
  f.lo0 = tr_read<O>(ln.v0); f.hi0 = tr_read<O + 1024>(ln.v0);
  f.lo1 = tr_read<O>(ln.v1); f.hi1 = tr_read<O + 1024>(ln.v1);
}

// Scores of one 32-key block against this wave's 32 queries: s = K (Q * scale * log2 e)^T [- m_ref] [- 1e30 on keys that may not be
// attended].  The two optional terms come from one more MFMA (header comment); `plain` = neither is needed.  k0..k3: the block's K
// fragments of the four k-steps.  refw: this lane's B word {bf16 1.0, bf16 -m_ref} (lanes 32..63, which hold k-slots 8..15: 0).
IA_DEV void qk_block(f32x16& s, const bf16x8& k0, const bf16x8& k1, const bf16x8& k2, const bf16x8& k3, const bf16x8 (&qf)[4], bool plain,
                     uint32_t refw, uint32_t valid, int lane) {
  const f32x16 zero = zero16();
  if (plain) {                                                                      // wave-uniform
    s = mfma(k0, qf[0], zero);
  } else {
    const uint32_t low = (lane & 32) ? 0u : 1u;
    const uint32_t bad = ((~valid) >> (lane & 31)) & low;
    const u32x4 a = {bad * NEG_BIG_BF16 + low * 0x3F800000u, 0u, 0u, 0u};
    const u32x4 bw = {refw, 0u, 0u, 0u};
    const f32x16 c = mfma(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bw), zero);
    s = mfma(k0, qf[0], c);
  }
  s = mfma(k1, qf[1], s);
  s = mfma(k2, qf[2], s);
  s = mfma(k3, qf[3], s);
}

struct Row {           // per-lane softmax state of its query (the partner lane ^ 32 holds the other half of the keys)
  float m_ref, l_run;  // m_ref is a bf16-representable number (it travels through the matrix pipe as a bf16 operand)
  uint32_t refw;       // {bf16 1.0, bf16 -m_ref} in lanes 0..31, 0 in lanes 32..63
};

// Dropout on a register of two packed bf16 probabilities (keys 2i, 2i+1): h = their 32-bit draw (ia_rng_pair: low half = even key), kept
// iff the 16-bit draw >= thr16.  No compare / select (v_cmp 8.6 cycles + v_cndmask on one wave): saturating packed subtract of thr16-1
// (0 iff dropped), clamp to 0 / 1, integer multiply of the bf16 bit pattern.  thr1 = (thr16 - 1) in both halves.
IA_DEV uint32_t drop_pair(uint32_t w, uint32_t h, uint32_t thr1) {
  uint32_t d;
  asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]\n\tv_pk_mul_lo_u16 %0, %3, %0" : "=&v"(d) : "v"(h), "v"(thr1), "v"(w));
  return d;
}

// p = exp2(s [* sc]) in place
IA_DEV void exp_block(f32x16& s, float sc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(PRESCALE ? s[r] : s[r] * sc);
}
// The block's probabilities as the B fragments of its two PV k-steps (bf16), and this lane's part of the row sum taken from the
// ROUNDED values (v_dot2c_f32_bf16 against packed ones, exact in fp32): O = sum(p^ v) / sum(p^) is then a convex combination of the v
// rows -- a row with one attendable key returns that key's v exactly, and delta = rowsum(dO o) cancels against dP in the backward as
// it should.  (The builtin, not inline asm: the dot instruction's result needs wait states before an ordinary VALU read, which only
// the compiler's hazard recognizer inserts.)
IA_DEV float pack_sum(bf16x8& pa, bf16x8& pb, const f32x16& s) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { pa[j] = f2bf(s[j]); pb[j] = f2bf(s[8 + j]); }
  float ra = 0.f, rb = 0.f;
  const bf16x2 ones = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    ra = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{pa[i], pa[i + 1]}, ones, ra, false);
    rb = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{pb[i], pb[i + 1]}, ones, rb, false);
  }
  return ra + rb;
}

// Rare: a lane's row sum left [2^-100, 2^60] (or is not finite) at this block.  The block's scores are recomputed (its K fragments are
// still in their registers), the reference moves to the larger of the block's maximum and the running log-sum-exp, everything
// accumulated so far follows; `other`: scores of the tile's second block, already formed against the old reference (or null).
// sc: what exp2's argument is multiplied by (1 with pre-scaled queries); m_ref lives in the accumulators' units.
IA_DEV void rebase(f32x16& s, f32x16* other, f32x16& o0, f32x16& o1, Row& row, const bf16x8& k0, const bf16x8& k1, const bf16x8& k2,
                   const bf16x8& k3, const bf16x8 (&qf)[4], uint32_t valid, int lane, float sc) {
  if (PRESCALE) sc = 1.f;
  const float inv_sc = 1.f / sc;
  qk_block(s, k0, k1, k2, k3, qf, false, (lane & 32) ? 0u : 0x3F80u, valid, lane);      // reference 0
  float tm = NEG_BIG;
#pragma unroll
  for (int r = 0; r < 16; ++r) tm = fmaxf(tm, s[r]);
  tm = fmaxf(tm, swap32(tm));
  const float l_prev = row.l_run + swap32(row.l_run);
  const bool have_prev = l_prev > 0.f, have_blk = tm > 0.5f * NEG_BIG;
  float m_new = row.m_ref;
  if (have_prev) m_new = row.m_ref + __builtin_amdgcn_logf(l_prev) * inv_sc;      // v_log_f32 = log2
  if (have_blk) m_new = have_prev ? fmaxf(m_new, tm) : tm;
  m_new = bf2f(f2bf(m_new));
  const float shift = row.m_ref - m_new;
  const float alpha = have_prev ? __builtin_amdgcn_exp2f(shift * sc) : 0.f;
  row.m_ref = m_new;
  row.refw = (lane & 32) ? 0u : (0x3F80u | ((uint32_t)__builtin_bit_cast(uint16_t, f2bf(-m_new)) << 16));
  row.l_run *= alpha;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s[r] = __builtin_amdgcn_exp2f((s[r] - m_new) * sc);
    o0[r] *= alpha; o1[r] *= alpha;
  }
  if (other) {
#pragma unroll
    for (int r = 0; r < 16; ++r) (*other)[r] += shift;
  }
}
}  // namespace fwd3

// QB = 32-query blocks per wave: with 2, a workgroup owns 256 queries and every K / V^T fragment read from LDS feeds two MFMAs -- the
// per-tile costs that are neither MFMA nor softmax (LDS-DMA issue and latency, barrier, fragment reads) and the per-workgroup
// prologue / epilogue are spent once per 32 MFMAs instead of once per 16 (measured: profiles/r03_attention_variants.txt).
template <bool DROPOUT, int QB>
__global__ __launch_bounds__(256, QB == 1 ? 3 : 2) void attn_fwd3_kernel(AttnArgs p) {
  using namespace fwd3;
  constexpr int ROWS = 128 * QB;                          // queries per workgroup
  constexpr int QX_OFF = SMEM;                            // QB = 2: the second query block's staging rows (16 KiB) behind the table
  __shared__ __attribute__((aligned(16))) char smem[SMEM + (QB - 1) * 16384];
  uint32_t (*s_valid)[2] = reinterpret_cast<uint32_t (*)[2]>(smem + TAB_OFF);
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lq = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, h, b;
  attn_block_coords_n(p, p.Lq, ROWS, tile, h, b);
  int Lq = p.Lq, L = p.Lk;
  size_t qbase = (size_t)b * Lq, rowbase = (size_t)b * L;
  if (p.cu) {
    const int c0 = p.cu[b];
    Lq = L = p.cu[b + 1] - c0;
    qbase = rowbase = (size_t)c0;
    if (tile * ROWS >= Lq) return;
  }
  const int q0 = tile * ROWS + wave * 32 * QB;            // this wave's first query (block qb: + 32 qb)
  const bool active = q0 < Lq;
  // this sequence's K / V rows of head h as buffer windows: rows >= L are out of range and arrive as zeros
  const uint32_t win = (uint32_t)(((size_t)(L - 1) * p.ld_kv + 64) * 2);
  const __amdgpu_buffer_rsrc_t rsK = ia_rsrc(p.k + rowbase * p.ld_kv + h * 64, win);
  const __amdgpu_buffer_rsrc_t rsV = ia_rsrc(p.v + rowbase * p.ld_kv + h * 64, win);
  const uint32_t sbase = lds_addr(smem);
  const uint32_t tile_bytes = (uint32_t)p.ld_kv * 128u;                            // 64 rows
  const int nkt_all = (L + 63) >> 6;

  Lane ln;
  {
    const uint32_t a0 = (uint32_t)(lq * 128 + ((hh ^ swz_b128(lq)) << 4));
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) ln.ka[kb] = sbase + (a0 ^ (uint32_t)(kb << 5));
    ln.v0 = sbase + tr_lane_off(lane, 0); ln.v1 = sbase + tr_lane_off(lane, 32);
    const int r0 = tid >> 3, r1 = 32 + (tid >> 3), c = tid & 7;
    ln.dk0 = (uint32_t)((r0 * p.ld_kv + (c ^ swz_b128(r0)) * 8) * 2); ln.dk1 = (uint32_t)((r1 * p.ld_kv + (c ^ swz_b128(r1)) * 8) * 2);
    ln.dv0 = (uint32_t)((r0 * p.ld_kv + (c ^ swz_tr(r0)) * 8) * 2);   ln.dv1 = (uint32_t)((r1 * p.ld_kv + (c ^ swz_tr(r1)) * 8) * 2);
  }
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);        // one flat -> LDS cast; DMA destinations are LDS-pointer arithmetic from here
  // key tile kt into ring slot SLOT: two 16-byte-per-lane DMA issues per operand, tile selected by the scalar offset
  auto stage_k = [&](auto SLOT_T, int kt) {
    constexpr int S = decltype(SLOT_T)::value;
    const uint32_t so = (uint32_t)kt * tile_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lsm + k_slot(S) + wave * 1024), 16, ln.dk0, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lsm + k_slot(S) + 4096 + wave * 1024), 16, ln.dk1, so, 0, 0);
  };
  auto stage_v = [&](auto SLOT_T, int kt) {
    constexpr int S = decltype(SLOT_T)::value;
    const uint32_t so = (uint32_t)kt * tile_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lsm + v_slot(S) + wave * 1024), 16, ln.dv0, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lsm + v_slot(S) + 4096 + wave * 1024), 16, ln.dv1, so, 0, 0);
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;

  // ---- prologue: one round trip for Q, K0, V0, K1 and the mask bytes; V1 follows and stays in flight
  {
    const __amdgpu_buffer_rsrc_t rsQ = ia_rsrc(p.q, p.q_bytes);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
      stage_rows32(rsQ, smem + (qb ? QX_OFF : Q_OFF) + wave * 4096, qbase + q0 + 32 * qb, Lq - q0 - 32 * qb, p.ld_q, h * 64, lane);
  }
  stage_k(S0{}, 0); stage_v(S0{}, 0); stage_k(S1{}, 1);
  build_valid_table(p, s_valid, rowbase, L, lane, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stage_v(S1{}, 1);
  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const bf16x8 raw = frag_b128(smem + (qb ? QX_OFF : Q_OFF) + wave * 4096, lq, kb * 2 + hh);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[qb][kb][j] = (PRESCALE && !p.q_prescaled) ? f2bf(bf2f(raw[j]) * p.sc) : raw[j];
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // trailing tiles without any attendable key are never touched (right-padded batches); bit t of rag0 / rag1: the first / second
  // 32-key block of tile t has a key that may not be attended (such blocks take the penalty MFMA)
  int nkt = nkt_all;
  uint32_t rag0 = 0u, rag1 = 0u;
  for (int t = 0; t < nkt_all; ++t) {
    const uint32_t lo = s_valid[t][0], hi = s_valid[t][1];
    if (lo != 0xFFFFFFFFu) rag0 |= 1u << t;
    if (hi != 0xFFFFFFFFu) rag1 |= 1u << t;
    if ((lo | hi) != 0u) nkt = t + 1;
  }
  nkt = __builtin_amdgcn_readfirstlane(nkt);
  rag0 = __builtin_amdgcn_readfirstlane(rag0);
  rag1 = __builtin_amdgcn_readfirstlane(rag1);

  uint32_t rk[QB];                                        // dropout: row keys of this lane's queries
  Row row[QB];
  f32x16 o[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    rk[qb] = DROPOUT ? ia_rng_row(p.seed, (uint32_t)(b * p.nh + h), (uint32_t)(q0 + 32 * qb + lq)) : 0u;
    row[qb] = Row{0.f, 0.f, hh ? 0u : 0x3F80u};
    o[qb][0] = zero16(); o[qb][1] = zero16();
  }
  const uint32_t thr1 = DROPOUT ? (p.thr16 - 1u) * 0x10001u : 0u;
  bool has_ref = false;                                  // wave-uniform: some row of this wave has left the reference 0
  bf16x8 kf[8];
  if (active) read_k<0>(kf, ln);

  // one tile (key tile t sits in ring slot SLOT = t % 3)
  auto tile_step = [&](auto SLOT_T, int t) {
    constexpr int SLOT = decltype(SLOT_T)::value;
    using NEXT2 = std::integral_constant<int, (SLOT + 2) % 3>;
    const bool LAST = t + 1 >= nkt;                       // wave-uniform
    // K(t+1) has landed for this wave (V(t+1), issued behind it, may still be in flight), then for everybody
    if (LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nkt) { stage_k(NEXT2{}, t + 2); stage_v(NEXT2{}, t + 2); }
    if (!active) return;
    // The tile is worked in its two 32-key blocks: the softmax of one block (VALU) has the MFMAs of the other beside it.
    f32x16 s[QB][2];
    frag_wait<0>(kf);                                     // the K fragments of tile t
    const bool plain0 = !has_ref && !((rag0 >> t) & 1u), plain1 = !has_ref && !((rag1 >> t) & 1u);
    uint32_t cur_lo = 0xFFFFFFFFu, cur_hi = 0xFFFFFFFFu;
    if (!plain0 || !plain1) { cur_lo = __builtin_amdgcn_readfirstlane(s_valid[t][0]); cur_hi = __builtin_amdgcn_readfirstlane(s_valid[t][1]); }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) qk_block(s[qb][0], kf[0], kf[2], kf[4], kf[6], qf[qb], plain0, row[qb].refw, cur_lo, lane);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) qk_block(s[qb][1], kf[1], kf[3], kf[5], kf[7], qf[qb], plain1, row[qb].refw, cur_hi, lane);
    TrPair va, vb, vc, vd;
    read_v<SLOT, 0>(va, ln);
    read_v<SLOT, 16>(vb, ln);
    bf16x8 pf[QB][2];                                     // the current block's probabilities (two PV k-steps)
    const uint32_t tile_c = DROPOUT ? (uint32_t)(t * 32) * IA_RNG_PAIR_C + (uint32_t)(2 * hh) * IA_RNG_PAIR_C : 0u;
    // one block's probabilities: exp2, pack + row sum, range check (rare: rebase), dropout
    auto softmax_block = [&](auto BLK, auto QBI, bool plain_b, uint32_t valid_b) {
      constexpr int blk = decltype(BLK)::value, qb = decltype(QBI)::value;
      f32x16& sc = s[qb][blk];
      exp_block(sc, p.sc);
      float rs = pack_sum(pf[qb][0], pf[qb][1], sc);
      const float tot = row[qb].l_run + rs;
      if (__builtin_expect(__ballot(__builtin_bit_cast(uint32_t, tot) - L_LO_BITS > L_HI_BITS - L_LO_BITS) != 0ull, 0)) {
        if (plain_b) valid_b = __builtin_amdgcn_readfirstlane(s_valid[t][blk]);
        rebase(sc, blk == 0 ? &s[qb][1] : nullptr, o[qb][0], o[qb][1], row[qb], kf[blk], kf[2 + blk], kf[4 + blk], kf[6 + blk], qf[qb], valid_b,
               lane, p.sc);
        rs = pack_sum(pf[qb][0], pf[qb][1], sc);
        has_ref = true;
      }
      row[qb].l_run += rs;
      if (DROPOUT) {
        // word i of pf[qb][n] = keys (ACC_ROW(8 n + 2 i), +1) of the block: pair constant = tile part + lane part + immediate
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          u32x4 w = __builtin_bit_cast(u32x4, pf[qb][n]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            constexpr uint32_t C = IA_RNG_PAIR_C;
            const int r = 8 * n + 2 * i;
            const uint32_t imm = (uint32_t)(((r & 3) >> 1) + 4 * (r >> 2) + 16 * blk) * C;
            w[i] = drop_pair(w[i], ia_rng_pair(rk[qb], tile_c + imm), thr1);
          }
          pf[qb][n] = __builtin_bit_cast(bf16x8, w);
        }
      }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    softmax_block(I0{}, I0{}, plain0, cur_lo);
    if constexpr (QB == 2) softmax_block(I0{}, I1{}, plain0, cur_lo);
    tr_wait<4>(va);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { o[qb][0] = mfma(va.a0(), pf[qb][0], o[qb][0]); o[qb][1] = mfma(va.a1(), pf[qb][0], o[qb][1]); }
    read_v<SLOT, 32>(vc, ln);
    tr_wait<4>(vb);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { o[qb][0] = mfma(vb.a0(), pf[qb][1], o[qb][0]); o[qb][1] = mfma(vb.a1(), pf[qb][1], o[qb][1]); }
    read_v<SLOT, 48>(vd, ln);
    softmax_block(I1{}, I0{}, plain1, cur_hi);
    if constexpr (QB == 2) softmax_block(I1{}, I1{}, plain1, cur_hi);
    tr_wait<4>(vc);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { o[qb][0] = mfma(vc.a0(), pf[qb][0], o[qb][0]); o[qb][1] = mfma(vc.a1(), pf[qb][0], o[qb][1]); }
    tr_wait<0>(vd);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { o[qb][0] = mfma(vd.a0(), pf[qb][1], o[qb][0]); o[qb][1] = mfma(vd.a1(), pf[qb][1], o[qb][1]); }
    // the K fragments of tile t+1 land under the last MFMAs and the top of the next tile (no wait here depends on them)
    if (!LAST) read_k<(SLOT + 1) % 3>(kf, ln);
  };
  {
    int t = 0;
    for (;;) {
      tile_step(S0{}, t); if (++t >= nkt) break;
      tile_step(S1{}, t); if (++t >= nkt) break;
      tile_step(S2{}, t); if (++t >= nkt) break;
    }
  }
  __builtin_amdgcn_s_barrier();                           // the epilogue rows are staged in the ring
  if (!active) return;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qq0 = q0 + 32 * qb;
    if (qq0 >= Lq) break;                                 // wave-uniform
    const float l_tot = row[qb].l_run + swap32(row[qb].l_run);
    const float inv = l_tot > 0.f ? p.inv_keep / l_tot : 0.f;
    if (qq0 + lq < Lq && hh == 0 && p.lse2)
      p.lse2[((size_t)b * p.nh + h) * p.Lq + qq0 + lq] = row[qb].m_ref * (PRESCALE ? 1.f : p.sc) + __builtin_amdgcn_logf(l_tot);
    store_block_rows(smem + EPI_OFF + (wave * QB + qb) * EPI_SLOT, o[qb][0], o[qb][1], inv, false, p.out + (qbase + qq0) * p.ld_o + h * 64, p.ld_o,
                     Lq - qq0, lane);
  }
}

// ------------------------------------------------------------------------------------- backward: dQ
// One 64-key tile for this wave's 32 queries: dQ^T += K^T dS^T with dS^T = P^T (dP^T - delta) (the softmax scale is
// applied once, when dQ is stored).
template <bool DROPOUT>
IA_DEV void dq_tile(const AttnArgs& p, const char* sK, const bf16x8 (&qf)[4], const bf16x8 (&gf)[4], uint32_t valid_lo,
                    uint32_t valid_hi, float lse, float dlt, f32x16& dq0, f32x16& dq1, int lane, int q, int kt, uint32_t rk) {
  const int hh = lane >> 5, lq = lane & 31;
  const char* sKt = sK + 8192;
  const char* sV = sK + 16384;
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
  const uint32_t kb0 = lds_addr(sKt) + tr_lane_off(lane, 0), kb1 = lds_addr(sKt) + tr_lane_off(lane, 32);
  TrPair ka, kb_;
  tr_issue<0>(ka, kb0, kb1);
  tr_issue<16>(kb_, kb0, kb1);
  if ((valid_lo & valid_hi) != 0xFFFFFFFFu) {   // wave-uniform
    asm volatile("" ::: "memory");   // keeps hipcc from flattening this branch into 64 always-executed selects
    const uint32_t vlo = hh ? valid_lo >> 4 : valid_lo, vhi = hh ? valid_hi >> 4 : valid_hi;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bit = (r & 3) + 8 * (r >> 2);
      if (!((vlo >> bit) & 1)) s0[r] = -INFINITY;
      if (!((vhi >> bit) & 1)) s1[r] = -INFINITY;
    }
  }
  const float neg_lse = -lse;
  const uint32_t tile_c = (uint32_t)(kt * 32) * IA_RNG_PAIR_C + (uint32_t)(2 * hh) * IA_RNG_PAIR_C;      // as in fwd_tile
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    uint32_t ra = 0xFFFFFFFFu, rb = 0xFFFFFFFFu;       // one draw per pair of neighbouring keys (r, r+1)
    if (DROPOUT) {
      constexpr uint32_t C = IA_RNG_PAIR_C;
      const uint32_t imm = (uint32_t)(((r & 3) >> 1) + 4 * (r >> 2)) * C;
      ra = ia_rng_pair(rk, tile_c + imm);
      rb = ia_rng_pair(rk, tile_c + imm + 16u * C);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float pa = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r + e], p.sc, neg_lse));
      const float pb = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r + e], p.sc, neg_lse));
      float da = dp0[r + e], db = dp1[r + e];
      if (DROPOUT) {
        da = ((e ? ra >> 16 : ra & 0xFFFFu) >= p.thr16) ? da * p.inv_keep : 0.f;
        db = ((e ? rb >> 16 : rb & 0xFFFFu) >= p.thr16) ? db * p.inv_keep : 0.f;
      }
      s0[r + e] = pa * (da - dlt);
      s1[r + e] = pb * (db - dlt);
    }
  }
  bf16x8 sf[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sf[0][j] = f2bf(s0[j]); sf[1][j] = f2bf(s0[8 + j]);
    sf[2][j] = f2bf(s1[j]); sf[3][j] = f2bf(s1[8 + j]);
  }
  tr_wait<4>(ka);
  dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka.a0(), sf[0], dq0, 0, 0, 0);
  dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka.a1(), sf[0], dq1, 0, 0, 0);
  TrPair kc, kd;
  tr_issue<32>(kc, kb0, kb1);
  tr_wait<4>(kb_);
  dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_.a0(), sf[1], dq0, 0, 0, 0);
  dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_.a1(), sf[1], dq1, 0, 0, 0);
  tr_issue<48>(kd, kb0, kb1);
  tr_wait<4>(kc);
  dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc.a0(), sf[2], dq0, 0, 0, 0);
  dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc.a1(), sf[2], dq1, 0, 0, 0);
  tr_wait<0>(kd);
  dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kd.a0(), sf[3], dq0, 0, 0, 0);
  dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kd.a1(), sf[3], dq1, 0, 0, 0);
}

template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs p) {
  // per buffer: K (b128 layout) | K (transpose-read layout) | V (b128 layout) = 24 KiB; then the valid-key table
  // + 24 KiB: the prologue stages 12 KiB of q / dO / o rows per wave in the 48 KiB behind ring slot 0
  __shared__ __attribute__((aligned(16))) char smem[3 * 24576 + MAX_KT * 8];
  uint32_t (*s_valid)[2] = reinterpret_cast<uint32_t (*)[2]>(smem + 3 * 24576);
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lq = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, h, b;
  attn_block_coords(p, p.Lq, tile, h, b);
  int Lq = p.Lq, L = p.Lk;                        // L: keys
  size_t qbase = (size_t)b * Lq, rowbase = (size_t)b * L;
  if (p.cu) {                                     // packed rows: this sequence's own length and first row
    const int s0 = p.cu[b];
    Lq = L = p.cu[b + 1] - s0;
    qbase = rowbase = (size_t)s0;
    if (tile * 128 >= Lq) return;                 // block-uniform, before any barrier
  }
  const int q0 = tile * 128 + wave * 32;
  const bool active = q0 < Lq;
  const int q = q0 + lq;
  const int qc = q < Lq ? q : Lq - 1;
  const __amdgpu_buffer_rsrc_t rsK = ia_rsrc(p.k, p.kv_bytes);
  const __amdgpu_buffer_rsrc_t rsV = ia_rsrc(p.v, p.kv_bytes);
  const uint32_t rk = DROPOUT ? ia_rng_row(p.seed, (uint32_t)(b * p.nh + h), (uint32_t)q) : 0u;      // row key of this lane's query
  const int nkt = (L + 63) >> 6;

  // One memory round trip for the whole prologue: this wave's q / dO / o rows (wave-private 4 KiB slots behind ring slot 0), the
  // first key tile, the saved log-sum-exp and the mask bytes are all requested before the single wait.
  char* rslot = smem + 24576 + wave * 12288;
  stage_rows32(ia_rsrc(p.q, p.q_bytes), rslot, qbase + q0, Lq - q0, p.ld_q, h * 64, lane);
  stage_rows32(ia_rsrc(p.d_o, p.o_bytes), rslot + 4096, qbase + q0, Lq - q0, p.ld_o, h * 64, lane);
  stage_rows32(ia_rsrc(p.o, p.o_bytes), rslot + 8192, qbase + q0, Lq - q0, p.ld_o, h * 64, lane);
  stage64<false>(rsK, smem, rowbase, L, p.ld_kv, h * 64, tid, wave);
  stage64<true>(rsK, smem + 8192, rowbase, L, p.ld_kv, h * 64, tid, wave);
  stage64<false>(rsV, smem + 16384, rowbase, L, p.ld_kv, h * 64, tid, wave);
  const size_t sidx = ((size_t)b * p.nh + h) * p.Lq + qc;
  const float lse = p.lse2[sidx];
  build_valid_table(p, s_valid, rowbase, L, lane, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bf16x8 qf[4], gf[4];
  // delta = rowsum(dO * O) of this lane's query: each lane of the pair (lane, lane^32) holds half of the 64 columns.
  // Written out for the dK/dV kernel, which runs after this one on the same stream.
  float dlt = 0.f;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    qf[kb] = frag_b128(rslot, lq, kb * 2 + hh);
    gf[kb] = frag_b128(rslot + 4096, lq, kb * 2 + hh);
    const bf16x8 ov = frag_b128(rslot + 8192, lq, kb * 2 + hh);
#pragma unroll
    for (int j = 0; j < 8; ++j) dlt += bf2f(ov[j]) * bf2f(gf[kb][j]);
  }
  dlt += __shfl_xor(dlt, 32, 64);
  if (active && hh == 0 && q < Lq) p.delta[sidx] = dlt;
  f32x16 dq0 = zero16(), dq1 = zero16();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) {
      char* nb = smem + (buf ^ 1) * 24576;
      const size_t r0 = rowbase + (kt + 1) * 64; const int nv = L - (kt + 1) * 64;
      stage64<false>(rsK, nb, r0, nv, p.ld_kv, h * 64, tid, wave);
      stage64<true>(rsK, nb + 8192, r0, nv, p.ld_kv, h * 64, tid, wave);
      stage64<false>(rsV, nb + 16384, r0, nv, p.ld_kv, h * 64, tid, wave);
    }
    if (active) {
      const uint32_t valid_lo = __builtin_amdgcn_readfirstlane(s_valid[kt][0]);
      const uint32_t valid_hi = __builtin_amdgcn_readfirstlane(s_valid[kt][1]);
      if ((valid_lo | valid_hi) != 0u)
        dq_tile<DROPOUT>(p, smem + buf * 24576, qf, gf, valid_lo, valid_hi, lse, dlt, dq0, dq1, lane, q, kt, rk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float* cs_lds = p.cs_part ? reinterpret_cast<float*>(smem + 4 * EPI_SLOT) + wave * 64 : nullptr;      // behind the four store slots
  if (!active && !cs_lds) return;
  if (active) store_block_rows(smem + wave * EPI_SLOT, dq0, dq1, p.scale, false, p.dq + (qbase + q0) * p.ld_dq + h * 64, p.ld_dq, Lq - q0, lane, cs_lds);
  else zero_cs_row(cs_lds, lane);
  if (cs_lds) {           // workgroup-uniform: one row of the partial-sum matrix per workgroup
    __syncthreads();
    if (wave == 0) {
      const float* c = reinterpret_cast<const float*>(smem + 4 * EPI_SLOT);
      p.cs_part[(size_t)(b * ((p.Lq + 127) >> 7) + tile) * (3 * p.nh * 64) + h * 64 + lane] = (c[lane] + c[64 + lane]) + (c[128 + lane] + c[192 + lane]);
    }
  }
}

// ------------------------------------------------------------------------------------- backward, round 3: dQ
// The round-2 kernel above, rebuilt like the forward (issue-bound at head dim 64, so the loop carries only what the arithmetic needs):
//  * S^T = K (Q scale log2e)^T - lse and dP^T - delta come out of the matrix pipe: the first MFMA of each chain takes a 16-register
//    block holding -lse / -delta of the lane's query as its C operand (both are constants of the whole kernel), so P = exp2(acc) and
//    dS = P * acc': 32 exp2 + 32 multiplies + 16 packed converts per 24 MFMAs; the query fragments are pre-scaled exactly as in the
//    forward (the same bf16 rounding, so the recomputed P matches the forward's);
//  * masked keys take the penalty MFMA (P = exp2(-1e30) = 0), no compare / select pairs;
//  * per-sequence buffer windows + scalar DMA offsets + compile-time ring slots (3 stages of K | K for the transpose read | V), LDS
//    reads on lane-constant bases with immediates, fragments of k-step kb+1 requested before the MFMAs of kb (asm reads, counted waits).
// Dropout (text towers): dS = P (M / keep dP - delta); the draw of a key pair is turned into two 0 / 1 floats without compares
// (saturating packed subtract, clamp, v_cvt_f32_ubyte).
namespace bwd3 {
using namespace fwd3;
// One LDS image of a [64 rows][64 bf16] tile that BOTH read forms take without bank conflicts: the 16-byte chunk of row r is XORed
// with swz_u(r).  ds_read_b128 (a lane group = 16 rows of one column chunk) needs 8 different values over the even (and the odd)
// rows of its group; ds_read_b64_tr_b16 (a lane group = 4 rows x 32 bytes) needs rows r and r+2 to differ above bit 0.  With
// x = (r >> 1) & 7, swz_u = ((x & 1) << 2) | (x >> 1) does both.  The round-2 kernels kept two copies of K (dQ kernel) and of Q and dO
// (dK/dV kernel) in different layouts: a third / a half of their LDS-DMA traffic.
IA_DEV int swz_u(int row) { const int x = (row >> 1) & 7; return ((x & 1) << 2) | (x >> 1); }
// transpose-read lane offset inside such a tile (rows 0..7 of the 16-row step; rows 8..15 sit at (this ^ 32) + 1024)
IA_DEV uint32_t tr_lane_off_u(int lane, int col0) {
  const int p = lane & 15, G = lane >> 4;
  const int row = 4 * (G >> 1) + (p >> 2);
  const int col = col0 + 16 * (G & 1) + (p & 3) * 4;
  return (uint32_t)(row * 128 + ((((col >> 3) ^ swz_u(row))) << 4) + (col & 7) * 2);
}
constexpr int STAGE = 16384;                         // K | V, both in the unified layout
constexpr int DQ_ROWS_OFF = 3 * STAGE;               // 16 KiB: the prologue's q rows (dO / o rows go through stages 1 / 2)
constexpr int DQ_TAB_OFF = DQ_ROWS_OFF + 16384, DQ_SMEM = DQ_TAB_OFF + MAX_KT * 8;
template <int N>
IA_DEV void frag_wait4(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
// rows ROW0 .. ROW0+15 of the unified-layout tile at byte offset BASE, read transposed (b0 / b1: tr_lane_off_u for d 0..31 / 32..63,
// b0x / b1x = the same ^ 32: the swizzle of rows 8..15 differs in chunk bit 1)
template <int BASE, int ROW0>
IA_DEV void read_tr(TrPair& f, uint32_t b0, uint32_t b1, uint32_t b0x, uint32_t b1x) {
  constexpr int O = BASE + ROW0 * 128;
  f.lo0 = tr_read<O>(b0); f.hi0 = tr_read<O + 1024>(b0x);
  f.lo1 = tr_read<O>(b1); f.hi1 = tr_read<O + 1024>(b1x);
}
// two 0 / 1 floats (low / high 16-bit draw >= thr16) out of a 32-bit draw; thr1 = (thr16 - 1) in both halves
IA_DEV void keep_pair(uint32_t h, uint32_t thr1, float& k_lo, float& k_hi) {
  uint32_t d;
  asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=v"(d) : "v"(h), "v"(thr1));
  asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(k_lo) : "v"(d));
  asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(k_hi) : "v"(d));
}
}  // namespace bwd3

template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd3_dq_kernel(AttnArgs p) {
  using namespace bwd3;
  __shared__ __attribute__((aligned(16))) char smem[DQ_SMEM];
  uint32_t (*s_valid)[2] = reinterpret_cast<uint32_t (*)[2]>(smem + DQ_TAB_OFF);
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lq = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, h, b;
  attn_block_coords(p, p.Lq, tile, h, b);
  int Lq = p.Lq, L = p.Lk;
  size_t qbase = (size_t)b * Lq, rowbase = (size_t)b * L;
  if (p.cu) {
    const int c0 = p.cu[b];
    Lq = L = p.cu[b + 1] - c0;
    qbase = rowbase = (size_t)c0;
    if (tile * 128 >= Lq) return;
  }
  const int q0 = tile * 128 + wave * 32;
  const bool active = q0 < Lq;
  const int q = q0 + lq;
  const int qc = q < Lq ? q : Lq - 1;
  const uint32_t win = (uint32_t)(((size_t)(L - 1) * p.ld_kv + 64) * 2);
  const __amdgpu_buffer_rsrc_t rsK = ia_rsrc(p.k + rowbase * p.ld_kv + h * 64, win);
  const __amdgpu_buffer_rsrc_t rsV = ia_rsrc(p.v + rowbase * p.ld_kv + h * 64, win);
  const uint32_t sbase = lds_addr(smem);
  const uint32_t tile_bytes = (uint32_t)p.ld_kv * 128u, half_bytes = (uint32_t)p.ld_kv * 64u;
  const int nkt_all = (L + 63) >> 6;
  // lane constants: fragment read bases (stage 0) and the DMA offsets of this lane's 16 bytes inside a tile
  uint32_t ka[4], t0, t1, t0x, t1x, du;
  {
    const uint32_t a0 = (uint32_t)(lq * 128 + ((hh ^ swz_u(lq)) << 4));
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) ka[kb] = sbase + (a0 ^ (uint32_t)(kb << 5));
    t0 = sbase + tr_lane_off_u(lane, 0); t1 = sbase + tr_lane_off_u(lane, 32); t0x = t0 ^ 32u; t1x = t1 ^ 32u;
    const int r0 = tid >> 3, c = tid & 7;
    du = (uint32_t)((r0 * p.ld_kv + (c ^ swz_u(r0)) * 8) * 2);
  }
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);        // one flat -> LDS cast; DMA destinations are LDS-pointer arithmetic from here
  auto stage_tile = [&](auto SLOT_T, int kt) {
    constexpr int S = decltype(SLOT_T)::value * STAGE;
    const uint32_t so = (uint32_t)kt * tile_bytes, so2 = so + half_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lsm + S + wave * 1024), 16, du, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lsm + S + 4096 + wave * 1024), 16, du, so2, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lsm + S + 8192 + wave * 1024), 16, du, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lsm + S + 12288 + wave * 1024), 16, du, so2, 0, 0);
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;

  // ---- prologue: one round trip for this wave's q / dO / o rows (q behind the ring, dO / o in stages 1 / 2), key tile 0, lse and
  // the mask bytes
  char* const rq = smem + DQ_ROWS_OFF + wave * 4096;
  char* const rg = smem + STAGE + wave * 4096;
  char* const ro = smem + 2 * STAGE + wave * 4096;
  stage_rows32(ia_rsrc(p.q, p.q_bytes), rq, qbase + q0, Lq - q0, p.ld_q, h * 64, lane);
  stage_rows32(ia_rsrc(p.d_o, p.o_bytes), rg, qbase + q0, Lq - q0, p.ld_o, h * 64, lane);
  stage_rows32(ia_rsrc(p.o, p.o_bytes), ro, qbase + q0, Lq - q0, p.ld_o, h * 64, lane);
  stage_tile(S0{}, 0);
  const size_t sidx = ((size_t)b * p.nh + h) * p.Lq + qc;
  const float lse = p.lse2[sidx];
  build_valid_table(p, s_valid, rowbase, L, lane, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bf16x8 qf[4], gf[4];
  float dlt = 0.f;                                      // delta = rowsum(dO * O): each lane of the pair (lane, lane ^ 32) holds half the columns
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const bf16x8 raw = frag_b128(rq, lq, kb * 2 + hh);
    gf[kb] = frag_b128(rg, lq, kb * 2 + hh);
    const bf16x8 ov = frag_b128(ro, lq, kb * 2 + hh);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dlt += bf2f(ov[j]) * bf2f(gf[kb][j]);
      qf[kb][j] = (PRESCALE && !p.q_prescaled) ? f2bf(bf2f(raw[j]) * p.sc) : raw[j];
    }
  }
  dlt += swap32(dlt);
  // IA_ATTN_EXACT_DELTA=1: the pre-pass (attn_bwd3_delta_kernel) has left the exact fp32 sum_k P dP there -- rowsum(dO o O) above is the
  // flash-style form of the same number taken from the bf16-ROUNDED context, 2^-9 |dO . O| off
  if (p.exact_delta) dlt = p.delta[sidx];
  else if (active && hh == 0 && q < Lq) p.delta[sidx] = dlt;       // for the dK/dV kernel, which runs behind this one on the stream
  f32x16 nl, nd;                                              // C operands: -lse, -delta (dropout: the mask sits between dP and delta)
#pragma unroll
  for (int r = 0; r < 16; ++r) { nl[r] = PRESCALE ? -lse : -lse / p.sc; nd[r] = DROPOUT ? 0.f : -dlt; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int nkt = nkt_all;
  uint32_t ragged = 0u;
  for (int t = 0; t < nkt_all; ++t) {
    const uint32_t lo = s_valid[t][0], hi = s_valid[t][1];
    if ((lo & hi) != 0xFFFFFFFFu) ragged |= 1u << t;
    if ((lo | hi) != 0u) nkt = t + 1;
  }
  // AttnArgs::dead_queries (round 6): a wave whose 32 query positions are all masked has dO == 0 there (the caller's guarantee), hence
  // dS == 0 and dQ == 0: it keeps its share of the staging and skips the arithmetic; a workgroup without any live query walks one key
  // tile instead of all of them.  Self-attention (Lq == Lk): the validity words of the keys are those of the query positions.
  bool work = active;
  if (p.dead_queries && p.cu == nullptr && p.Lq == p.Lk) {
    const uint32_t mine = active ? s_valid[q0 >> 6][(q0 >> 5) & 1] : 0u;
    const int t128 = tile * 2;                                      // this workgroup's 128 positions = tiles t128, t128 + 1 of the table
    uint32_t any = s_valid[t128][0] | s_valid[t128][1];
    if (t128 + 1 < nkt_all) any |= s_valid[t128 + 1][0] | s_valid[t128 + 1][1];
    work = active && __builtin_amdgcn_readfirstlane(mine) != 0u;
    if (__builtin_amdgcn_readfirstlane(any) == 0u) nkt = 1;
  }
  nkt = __builtin_amdgcn_readfirstlane(nkt);
  ragged = __builtin_amdgcn_readfirstlane(ragged);
  if (nkt > 1) stage_tile(S1{}, 1);
  const uint32_t rk = DROPOUT ? ia_rng_row(p.seed, (uint32_t)(b * p.nh + h), (uint32_t)q) : 0u;
  const uint32_t thr1 = DROPOUT ? (p.thr16 - 1u) * 0x10001u : 0u;
  f32x16 dq0 = zero16(), dq1 = zero16();

  auto tile_step = [&](auto SLOT_T, int t) {
    constexpr int SB = decltype(SLOT_T)::value * STAGE;
    using NEXT2 = std::integral_constant<int, (decltype(SLOT_T)::value + 2) % 3>;
    const bool LAST = t + 1 >= nkt;
    // tile t has landed for this wave (tile t+1, issued behind it, may still be in flight), then for everybody
    if (LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nkt) stage_tile(NEXT2{}, t + 2);
    if (!work) return;
    bf16x8 xk0, xk1, xv0, xv1, yk0, yk1, yv0, yv1;        // K / V fragments of even / odd k-steps
    auto rd = [&](auto KB, bf16x8& k0, bf16x8& k1, bf16x8& v0, bf16x8& v1) {
      constexpr int kb = decltype(KB)::value;
      k0 = lds_read_b128<SB>(ka[kb]); k1 = lds_read_b128<SB + 4096>(ka[kb]);
      v0 = lds_read_b128<SB + 8192>(ka[kb]); v1 = lds_read_b128<SB + 12288>(ka[kb]);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    rd(I0{}, xk0, xk1, xv0, xv1);
    rd(I1{}, yk0, yk1, yv0, yv1);
    f32x16 s0, s1, dp0, dp1;
    const bool plain = !((ragged >> t) & 1u);
    frag_wait4<4>(xk0, xk1, xv0, xv1);
    s0 = mfma(xk0, qf[0], nl); s1 = mfma(xk1, qf[0], nl);
    dp0 = mfma(xv0, gf[0], nd); dp1 = mfma(xv1, gf[0], nd);
    if (!plain) {
      // masked keys: one more accumulate step adds -1e30 to their rows (A = the penalty in k-slot 0 of the key's row, B = 1.0 in
      // k-slot 0 of every query).  The fragment registers of the reads in flight are not touched in here: a wait that sat on the
      // far side of a branch from its read once had the register allocator copy a fragment before its data had arrived
      // (tools/lint_asm_waits.py checks the compiled kernels for exactly that).
      const uint32_t v_lo = __builtin_amdgcn_readfirstlane(s_valid[t][0]), v_hi = __builtin_amdgcn_readfirstlane(s_valid[t][1]);
      const uint32_t low = hh ? 0u : 1u;
      const uint32_t bad0 = ((~v_lo) >> lq) & low, bad1 = ((~v_hi) >> lq) & low;
      const u32x4 a0 = {bad0 * NEG_BIG_BF16, 0u, 0u, 0u}, a1 = {bad1 * NEG_BIG_BF16, 0u, 0u, 0u}, bw = {low * 0x3F80u, 0u, 0u, 0u};
      s0 = mfma(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bw), s0);
      s1 = mfma(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bw), s1);
    }
    rd(I2{}, xk0, xk1, xv0, xv1);
    frag_wait4<4>(yk0, yk1, yv0, yv1);
    s0 = mfma(yk0, qf[1], s0); s1 = mfma(yk1, qf[1], s1); dp0 = mfma(yv0, gf[1], dp0); dp1 = mfma(yv1, gf[1], dp1);
    rd(I3{}, yk0, yk1, yv0, yv1);
    frag_wait4<4>(xk0, xk1, xv0, xv1);
    s0 = mfma(xk0, qf[2], s0); s1 = mfma(xk1, qf[2], s1); dp0 = mfma(xv0, gf[2], dp0); dp1 = mfma(xv1, gf[2], dp1);
    TrPair ta, tb, tc, td;                                // K^T fragments of the dQ MFMAs
    read_tr<SB, 0>(ta, t0, t1, t0x, t1x);
    read_tr<SB, 16>(tb, t0, t1, t0x, t1x);
    frag_wait4<8>(yk0, yk1, yv0, yv1);
    s0 = mfma(yk0, qf[3], s0); s1 = mfma(yk1, qf[3], s1); dp0 = mfma(yv0, gf[3], dp0); dp1 = mfma(yv1, gf[3], dp1);
    // dS^T = P^T (dP^T - delta), P = exp2(s - lse)
    if (DROPOUT) {
      const uint32_t tile_c = (uint32_t)(t * 32) * IA_RNG_PAIR_C + (uint32_t)(2 * hh) * IA_RNG_PAIR_C;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        constexpr uint32_t C = IA_RNG_PAIR_C;
        const uint32_t imm = (uint32_t)(((r & 3) >> 1) + 4 * (r >> 2)) * C;
        float a_lo, a_hi, b_lo, b_hi;
        keep_pair(ia_rng_pair(rk, tile_c + imm), thr1, a_lo, a_hi);
        keep_pair(ia_rng_pair(rk, tile_c + imm + 16u * C), thr1, b_lo, b_hi);
        s0[r] = __builtin_amdgcn_exp2f(PRESCALE ? s0[r] : s0[r] * p.sc) * __builtin_fmaf(dp0[r] * a_lo, p.inv_keep, -dlt);
        s0[r + 1] = __builtin_amdgcn_exp2f(PRESCALE ? s0[r + 1] : s0[r + 1] * p.sc) * __builtin_fmaf(dp0[r + 1] * a_hi, p.inv_keep, -dlt);
        s1[r] = __builtin_amdgcn_exp2f(PRESCALE ? s1[r] : s1[r] * p.sc) * __builtin_fmaf(dp1[r] * b_lo, p.inv_keep, -dlt);
        s1[r + 1] = __builtin_amdgcn_exp2f(PRESCALE ? s1[r + 1] : s1[r + 1] * p.sc) * __builtin_fmaf(dp1[r + 1] * b_hi, p.inv_keep, -dlt);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(PRESCALE ? s0[r] : s0[r] * p.sc) * dp0[r];
        s1[r] = __builtin_amdgcn_exp2f(PRESCALE ? s1[r] : s1[r] * p.sc) * dp1[r];
      }
    }
    bf16x8 sf[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sf[0][j] = f2bf(s0[j]); sf[1][j] = f2bf(s0[8 + j]);
      sf[2][j] = f2bf(s1[j]); sf[3][j] = f2bf(s1[8 + j]);
    }
    tr_wait<4>(ta);
    dq0 = mfma(ta.a0(), sf[0], dq0); dq1 = mfma(ta.a1(), sf[0], dq1);
    read_tr<SB, 32>(tc, t0, t1, t0x, t1x);
    tr_wait<4>(tb);
    dq0 = mfma(tb.a0(), sf[1], dq0); dq1 = mfma(tb.a1(), sf[1], dq1);
    read_tr<SB, 48>(td, t0, t1, t0x, t1x);
    tr_wait<4>(tc);
    dq0 = mfma(tc.a0(), sf[2], dq0); dq1 = mfma(tc.a1(), sf[2], dq1);
    tr_wait<0>(td);
    dq0 = mfma(td.a0(), sf[3], dq0); dq1 = mfma(td.a1(), sf[3], dq1);
  };
  {
    int t = 0;
    for (;;) {
      tile_step(S0{}, t); if (++t >= nkt) break;
      tile_step(S1{}, t); if (++t >= nkt) break;
      tile_step(S2{}, t); if (++t >= nkt) break;
    }
  }
  __builtin_amdgcn_s_barrier();                           // the epilogue rows are staged in the ring
  float* cs_lds = p.cs_part ? reinterpret_cast<float*>(smem + 4 * EPI_SLOT) + wave * 64 : nullptr;      // behind the four store slots
  if (!active && !cs_lds) return;
  if (active) store_block_rows(smem + wave * EPI_SLOT, dq0, dq1, p.scale, !work, p.dq + (qbase + q0) * p.ld_dq + h * 64, p.ld_dq, Lq - q0, lane, cs_lds);
  else zero_cs_row(cs_lds, lane);
  if (cs_lds) {           // workgroup-uniform: one row of the partial-sum matrix per workgroup
    __syncthreads();
    if (wave == 0) {
      const float* c = reinterpret_cast<const float*>(smem + 4 * EPI_SLOT);
      p.cs_part[(size_t)(b * ((p.Lq + 127) >> 7) + tile) * (3 * p.nh * 64) + h * 64 + lane] = (c[lane] + c[64 + lane]) + (c[128 + lane] + c[192 + lane]);
    }
  }
}

// ---------------------------------------------------------------------- backward, round 6: exact softmax-gradient delta (opt-in)
// IA_ATTN_EXACT_DELTA=1.  A flash-style backward takes delta_q = rowsum(dO o O) from the context the forward STORED, i.e. rounded to
// bf16: delta is off by ~2^-9 |dO . O| and enters dS = P (dP - delta) as eps * P -- invisible where a layer's 255 query rows average it
// out, but the one place it shows is a top layer whose gradient arrives through a single row (CoCa ensemble = sum: only the CLS row
// of the last text layer carries gradient; layer-23 query / key weight gradients 0.27 / 0.25 off the fp32 reference,
// profiles/r05_c5_delta_probe.txt).  This pre-pass forms the same number the exact way, sum_k P_qk dP_qk in fp32 (the dQ kernel's S
// and dP chains without the dQ half: 16 of its 24 MFMAs per tile), into `delta`; the dQ and dK/dV kernels then read it instead of
// forming rowsum(dO o O).  Costs ~60 % of the dQ kernel per launch; the fused one-kernel backward (L <= 256) is bypassed while it is on.
template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd3_delta_kernel(AttnArgs p) {
  using namespace bwd3;
  __shared__ __attribute__((aligned(16))) char smem[DQ_SMEM];
  uint32_t (*s_valid)[2] = reinterpret_cast<uint32_t (*)[2]>(smem + DQ_TAB_OFF);
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lq = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, h, b;
  attn_block_coords(p, p.Lq, tile, h, b);
  int Lq = p.Lq, L = p.Lk;
  size_t qbase = (size_t)b * Lq, rowbase = (size_t)b * L;
  if (p.cu) {
    const int c0 = p.cu[b];
    Lq = L = p.cu[b + 1] - c0;
    qbase = rowbase = (size_t)c0;
    if (tile * 128 >= Lq) return;
  }
  const int q0 = tile * 128 + wave * 32;
  const bool active = q0 < Lq;
  const int q = q0 + lq;
  const int qc = q < Lq ? q : Lq - 1;
  const uint32_t win = (uint32_t)(((size_t)(L - 1) * p.ld_kv + 64) * 2);
  const __amdgpu_buffer_rsrc_t rsK = ia_rsrc(p.k + rowbase * p.ld_kv + h * 64, win);
  const __amdgpu_buffer_rsrc_t rsV = ia_rsrc(p.v + rowbase * p.ld_kv + h * 64, win);
  const uint32_t sbase = lds_addr(smem);
  const uint32_t tile_bytes = (uint32_t)p.ld_kv * 128u, half_bytes = (uint32_t)p.ld_kv * 64u;
  const int nkt_all = (L + 63) >> 6;
  // lane constants: fragment read bases (stage 0) and the DMA offsets of this lane's 16 bytes inside a tile
  uint32_t ka[4], du;
  {
    const uint32_t a0 = (uint32_t)(lq * 128 + ((hh ^ swz_u(lq)) << 4));
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) ka[kb] = sbase + (a0 ^ (uint32_t)(kb << 5));
    const int r0 = tid >> 3, c = tid & 7;
    du = (uint32_t)((r0 * p.ld_kv + (c ^ swz_u(r0)) * 8) * 2);
  }
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);        // one flat -> LDS cast; DMA destinations are LDS-pointer arithmetic from here
  auto stage_tile = [&](auto SLOT_T, int kt) {
    constexpr int S = decltype(SLOT_T)::value * STAGE;
    const uint32_t so = (uint32_t)kt * tile_bytes, so2 = so + half_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lsm + S + wave * 1024), 16, du, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lsm + S + 4096 + wave * 1024), 16, du, so2, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lsm + S + 8192 + wave * 1024), 16, du, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lsm + S + 12288 + wave * 1024), 16, du, so2, 0, 0);
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;

  // ---- prologue: one round trip for this wave's q / dO / o rows (q behind the ring, dO / o in stages 1 / 2), key tile 0, lse and
  // the mask bytes
  char* const rq = smem + DQ_ROWS_OFF + wave * 4096;
  char* const rg = smem + STAGE + wave * 4096;
  stage_rows32(ia_rsrc(p.q, p.q_bytes), rq, qbase + q0, Lq - q0, p.ld_q, h * 64, lane);
  stage_rows32(ia_rsrc(p.d_o, p.o_bytes), rg, qbase + q0, Lq - q0, p.ld_o, h * 64, lane);
  stage_tile(S0{}, 0);
  const size_t sidx = ((size_t)b * p.nh + h) * p.Lq + qc;
  const float lse = p.lse2[sidx];
  build_valid_table(p, s_valid, rowbase, L, lane, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bf16x8 qf[4], gf[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const bf16x8 raw = frag_b128(rq, lq, kb * 2 + hh);
    gf[kb] = frag_b128(rg, lq, kb * 2 + hh);
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[kb][j] = (PRESCALE && !p.q_prescaled) ? f2bf(bf2f(raw[j]) * p.sc) : raw[j];
  }
  f32x16 nl, nd;                                              // C operands: -lse, 0
#pragma unroll
  for (int r = 0; r < 16; ++r) { nl[r] = PRESCALE ? -lse : -lse / p.sc; nd[r] = 0.f; }
  float dsum = 0.f;                                           // this lane's part of sum_k P dP of its query (its 16 key rows of every block)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int nkt = nkt_all;
  uint32_t ragged = 0u;
  for (int t = 0; t < nkt_all; ++t) {
    const uint32_t lo = s_valid[t][0], hi = s_valid[t][1];
    if ((lo & hi) != 0xFFFFFFFFu) ragged |= 1u << t;
    if ((lo | hi) != 0u) nkt = t + 1;
  }
  nkt = __builtin_amdgcn_readfirstlane(nkt);
  ragged = __builtin_amdgcn_readfirstlane(ragged);
  if (nkt > 1) stage_tile(S1{}, 1);
  const uint32_t rk = DROPOUT ? ia_rng_row(p.seed, (uint32_t)(b * p.nh + h), (uint32_t)q) : 0u;
  const uint32_t thr1 = DROPOUT ? (p.thr16 - 1u) * 0x10001u : 0u;

  auto tile_step = [&](auto SLOT_T, int t) {
    constexpr int SB = decltype(SLOT_T)::value * STAGE;
    using NEXT2 = std::integral_constant<int, (decltype(SLOT_T)::value + 2) % 3>;
    const bool LAST = t + 1 >= nkt;
    // tile t has landed for this wave (tile t+1, issued behind it, may still be in flight), then for everybody
    if (LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nkt) stage_tile(NEXT2{}, t + 2);
    if (!active) return;
    bf16x8 xk0, xk1, xv0, xv1, yk0, yk1, yv0, yv1;        // K / V fragments of even / odd k-steps
    auto rd = [&](auto KB, bf16x8& k0, bf16x8& k1, bf16x8& v0, bf16x8& v1) {
      constexpr int kb = decltype(KB)::value;
      k0 = lds_read_b128<SB>(ka[kb]); k1 = lds_read_b128<SB + 4096>(ka[kb]);
      v0 = lds_read_b128<SB + 8192>(ka[kb]); v1 = lds_read_b128<SB + 12288>(ka[kb]);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    rd(I0{}, xk0, xk1, xv0, xv1);
    rd(I1{}, yk0, yk1, yv0, yv1);
    f32x16 s0, s1, dp0, dp1;
    const bool plain = !((ragged >> t) & 1u);
    frag_wait4<4>(xk0, xk1, xv0, xv1);
    s0 = mfma(xk0, qf[0], nl); s1 = mfma(xk1, qf[0], nl);
    dp0 = mfma(xv0, gf[0], nd); dp1 = mfma(xv1, gf[0], nd);
    if (!plain) {
      // masked keys: one more accumulate step adds -1e30 to their rows (A = the penalty in k-slot 0 of the key's row, B = 1.0 in
      // k-slot 0 of every query).  The fragment registers of the reads in flight are not touched in here: a wait that sat on the
      // far side of a branch from its read once had the register allocator copy a fragment before its data had arrived
      // (tools/lint_asm_waits.py checks the compiled kernels for exactly that).
      const uint32_t v_lo = __builtin_amdgcn_readfirstlane(s_valid[t][0]), v_hi = __builtin_amdgcn_readfirstlane(s_valid[t][1]);
      const uint32_t low = hh ? 0u : 1u;
      const uint32_t bad0 = ((~v_lo) >> lq) & low, bad1 = ((~v_hi) >> lq) & low;
      const u32x4 a0 = {bad0 * NEG_BIG_BF16, 0u, 0u, 0u}, a1 = {bad1 * NEG_BIG_BF16, 0u, 0u, 0u}, bw = {low * 0x3F80u, 0u, 0u, 0u};
      s0 = mfma(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bw), s0);
      s1 = mfma(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bw), s1);
    }
    rd(I2{}, xk0, xk1, xv0, xv1);
    frag_wait4<4>(yk0, yk1, yv0, yv1);
    s0 = mfma(yk0, qf[1], s0); s1 = mfma(yk1, qf[1], s1); dp0 = mfma(yv0, gf[1], dp0); dp1 = mfma(yv1, gf[1], dp1);
    rd(I3{}, yk0, yk1, yv0, yv1);
    frag_wait4<4>(xk0, xk1, xv0, xv1);
    s0 = mfma(xk0, qf[2], s0); s1 = mfma(xk1, qf[2], s1); dp0 = mfma(xv0, gf[2], dp0); dp1 = mfma(xv1, gf[2], dp1);
    frag_wait4<0>(yk0, yk1, yv0, yv1);
    s0 = mfma(yk0, qf[3], s0); s1 = mfma(yk1, qf[3], s1); dp0 = mfma(yv0, gf[3], dp0); dp1 = mfma(yv1, gf[3], dp1);
    // sum_k P dP (dropout: the kept entries of dP, scaled by 1 / keep -- what the forward's O = sum P M / keep V contracts with dO)
    if (DROPOUT) {
      const uint32_t tile_c = (uint32_t)(t * 32) * IA_RNG_PAIR_C + (uint32_t)(2 * hh) * IA_RNG_PAIR_C;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        constexpr uint32_t C = IA_RNG_PAIR_C;
        const uint32_t imm = (uint32_t)(((r & 3) >> 1) + 4 * (r >> 2)) * C;
        float a_lo, a_hi, b_lo, b_hi;
        keep_pair(ia_rng_pair(rk, tile_c + imm), thr1, a_lo, a_hi);
        keep_pair(ia_rng_pair(rk, tile_c + imm + 16u * C), thr1, b_lo, b_hi);
        dsum += __builtin_amdgcn_exp2f(PRESCALE ? s0[r] : s0[r] * p.sc) * (dp0[r] * a_lo * p.inv_keep);
        dsum += __builtin_amdgcn_exp2f(PRESCALE ? s0[r + 1] : s0[r + 1] * p.sc) * (dp0[r + 1] * a_hi * p.inv_keep);
        dsum += __builtin_amdgcn_exp2f(PRESCALE ? s1[r] : s1[r] * p.sc) * (dp1[r] * b_lo * p.inv_keep);
        dsum += __builtin_amdgcn_exp2f(PRESCALE ? s1[r + 1] : s1[r + 1] * p.sc) * (dp1[r + 1] * b_hi * p.inv_keep);
      }
    } else {
      float u0 = 0.f, u1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        u0 += __builtin_amdgcn_exp2f(PRESCALE ? s0[r] : s0[r] * p.sc) * dp0[r];
        u1 += __builtin_amdgcn_exp2f(PRESCALE ? s1[r] : s1[r] * p.sc) * dp1[r];
      }
      dsum += u0 + u1;
    }
  };
  {
    int t = 0;
    for (;;) {
      tile_step(S0{}, t); if (++t >= nkt) break;
      tile_step(S1{}, t); if (++t >= nkt) break;
      tile_step(S2{}, t); if (++t >= nkt) break;
    }
  }
  dsum += swap32(dsum);                                   // the partner lane holds the other half of every block's key rows
  if (active && hh == 0 && q < Lq) p.delta[sidx] = dsum;
}

// ---------------------------------------------------------------------------------- backward: dK, dV
// S orientation: rows = queries (accumulator registers), column = key = lane. A column of P / dS only ever reaches
// that key's dK / dV, so the key mask needs no per-element work: a masked key's outputs are simply stored as zero.
// Rows past the end of the sequence cost nothing either: their Q, dO, lse and delta arrive zero-filled from the DMA
// (P = 1, dP = 0, dS = 0).
template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnArgs p) {
  // per buffer: Q (b128) | Q (transpose-read) | dO (b128) | dO (transpose-read) | lse[64] | delta[64]
  constexpr int BUF = 32768 + 512;
  // + 1 KiB: under dropout every wave keeps the 64 row keys (ia_rng_row) of the current query tile in a private 256-byte slot
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF + 1024];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lk = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, h, b;
  attn_block_coords(p, p.Lk, tile, h, b);
  int Lq = p.Lq, L = p.Lk;                        // L: keys
  size_t qbase = (size_t)b * Lq, rowbase = (size_t)b * L;
  if (p.cu) {
    const int s0 = p.cu[b];
    Lq = L = p.cu[b + 1] - s0;
    qbase = rowbase = (size_t)s0;
    if (tile * 128 >= L) return;
  }
  const int k0 = tile * 128 + wave * 32;
  const bool active = k0 < L;
  const int key = k0 + lk;
  const int kc = key < L ? key : L - 1;
  const bool key_ok = key < L && (p.mask == nullptr || p.mask[rowbase + kc] != 0);

  const __amdgpu_buffer_rsrc_t rsQ = ia_rsrc(p.q, p.q_bytes);
  const __amdgpu_buffer_rsrc_t rsG = ia_rsrc(p.d_o, p.o_bytes);
  const __amdgpu_buffer_rsrc_t rsL = ia_rsrc(p.lse2 + ((size_t)b * p.nh + h) * p.Lq, (uint32_t)Lq * 4u);
  const __amdgpu_buffer_rsrc_t rsD = ia_rsrc(p.delta + ((size_t)b * p.nh + h) * p.Lq, (uint32_t)Lq * 4u);
  const uint32_t stream_id = (uint32_t)(b * p.nh + h);
  // dropout draw of (q, key): key is this lane -> its pair constant and the half of the draw it reads are lane constants
  const uint32_t pc = pair_c_of(key), ush = (uint32_t)(key & 1) * 16u;
  uint32_t* const s_rk = reinterpret_cast<uint32_t*>(smem + 2 * BUF) + wave * 64;

  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  const int nqt = (Lq + 63) >> 6;
  auto stage_all = [&](char* s, int qt) {
    const size_t r0 = qbase + (size_t)qt * 64; const int nv = Lq - qt * 64;
    stage64<false>(rsQ, s, r0, nv, p.ld_q, h * 64, tid, wave);
    stage64<true>(rsQ, s + 8192, r0, nv, p.ld_q, h * 64, tid, wave);
    stage64<false>(rsG, s + 16384, r0, nv, p.ld_o, h * 64, tid, wave);
    stage64<true>(rsG, s + 24576, r0, nv, p.ld_o, h * 64, tid, wave);
    // 64 x fp32 each, one 4-byte-per-lane DMA; out-of-range rows read as zero
    if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, IA_LDS(s + 32768), 4, (uint32_t)(qt * 64 + lane) * 4u, 0, 0, 0);
    if (wave == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, IA_LDS(s + 32768 + 256), 4, (uint32_t)(qt * 64 + lane) * 4u, 0, 0, 0);
  };
  // this wave's 32 key rows of K and V travel through wave-private 4 KiB slots of the second ring slot (free until the loop
  // prefetches query tile 1 behind the barrier), requested together with query tile 0
  char* kslot = smem + BUF + wave * 8192;
  stage_rows32(ia_rsrc(p.k, p.kv_bytes), kslot, rowbase + k0, L - k0, p.ld_kv, h * 64, lane);
  stage_rows32(ia_rsrc(p.v, p.kv_bytes), kslot + 4096, rowbase + k0, L - k0, p.ld_kv, h * 64, lane);
  stage_all(smem, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    kf[kb] = frag_b128(kslot, lk, kb * 2 + hh);
    vf[kb] = frag_b128(kslot + 4096, lk, kb * 2 + hh);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    if (qt + 1 < nqt) stage_all(smem + (buf ^ 1) * BUF, qt + 1);
    if (active) {
      if (DROPOUT) {      // row keys of this tile's 64 queries, one per lane (the wave's LDS operations complete in order: no barrier)
        s_rk[lane] = ia_rng_row(p.seed, stream_id, (uint32_t)(qt * 64 + lane));
        __builtin_amdgcn_wave_barrier();
      }
      const char* sQ = smem + buf * BUF;
      const char* sG = sQ + 16384;
      const uint32_t qt0 = lds_addr(sQ + 8192) + tr_lane_off(lane, 0), qt1 = lds_addr(sQ + 8192) + tr_lane_off(lane, 32);
      const uint32_t gt0 = lds_addr(sQ + 24576) + tr_lane_off(lane, 0), gt1 = lds_addr(sQ + 24576) + tr_lane_off(lane, 32);
      const float* sL = reinterpret_cast<const float*>(sQ + 32768);
      const float* sD = sL + 64;
      auto sub_tile = [&](auto QS) {
        constexpr int qs = decltype(QS)::value;
        const int qb = qt * 64 + qs * 32;     // first query of this 32-row sub tile
        // S[q][key] = Q K^T ; dP[q][key] = dO V^T   (rows = q in registers, column = key = lane)
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const bf16x8 a = frag_b128(sQ, qs * 32 + lk, kb * 2 + hh);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[kb], s, 0, 0, 0);
          const bf16x8 g = frag_b128(sG, qs * 32 + lk, kb * 2 + hh);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g, vf[kb], dp, 0, 0, 0);
        }
        TrPair g0, a0, g1, a1;
        tr_issue<qs * 32>(g0, gt0, gt1);
        tr_issue<qs * 32>(a0, qt0, qt1);
        tr_issue<qs * 32 + 16>(g1, gt0, gt1);
        tr_issue<qs * 32 + 16>(a1, qt0, qt1);
        bf16x8 pf[2], sf[2];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int qoff = qs * 32 + 8 * rg + 4 * hh;              // 4 consecutive queries
          const f32x4 ls = *reinterpret_cast<const f32x4*>(sL + qoff);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(sD + qoff);
          u32x4 rkq = {0u, 0u, 0u, 0u};
          if (DROPOUT) rkq = *reinterpret_cast<const u32x4*>(s_rk + qoff);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = rg * 4 + j;
            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.sc, -ls[j]));
            float d = dp[r];
            float pd = pv;
            if (DROPOUT) {
              const bool keep = ((ia_rng_pair(rkq[j], pc) >> ush) & 0xFFFFu) >= p.thr16;
              d = keep ? d * p.inv_keep : 0.f;
              pd = keep ? pv * p.inv_keep : 0.f;
            }
            const float ds = pv * (d - dl[j]);
            pf[r >> 3][r & 7] = f2bf(pd);
            sf[r >> 3][r & 7] = f2bf(ds);
          }
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
        tr_wait<12>(g0);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0.a0(), pf[0], dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0.a1(), pf[0], dv1, 0, 0, 0);
        tr_wait<8>(a0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.a0(), sf[0], dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.a1(), sf[0], dk1, 0, 0, 0);
        tr_wait<4>(g1);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1.a0(), pf[1], dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1.a1(), pf[1], dv1, 0, 0, 0);
        tr_wait<0>(a1);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.a0(), sf[1], dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.a1(), sf[1], dk1, 0, 0, 0);
      };
      sub_tile(std::integral_constant<int, 0>{});
      if (qt * 64 + 32 < Lq) sub_tile(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float* cs_lds = p.cs_part ? reinterpret_cast<float*>(smem + 8 * EPI_SLOT) + wave * 128 : nullptr;     // behind the eight store slots: dk | dv sums
  if (!active && !cs_lds) return;
  if (active) {
    // a masked key's outputs are zero (its P is not bounded by the saved log-sum-exp, so the accumulators may hold inf / nan)
    store_block_rows(smem + wave * 2 * EPI_SLOT, dk0, dk1, p.scale, !key_ok, p.dk + (rowbase + k0) * p.ld_dkv + h * 64, p.ld_dkv, L - k0, lane, cs_lds);
    store_block_rows(smem + (wave * 2 + 1) * EPI_SLOT, dv0, dv1, 1.f, !key_ok, p.dv + (rowbase + k0) * p.ld_dkv + h * 64, p.ld_dkv, L - k0, lane,
                     cs_lds ? cs_lds + 64 : nullptr);
  } else { zero_cs_row(cs_lds, lane); zero_cs_row(cs_lds + 64, lane); }
  if (cs_lds) {
    __syncthreads();
    if (wave < 2) {         // wave 0: the dk columns, wave 1: the dv columns
      const float* c = reinterpret_cast<const float*>(smem + 8 * EPI_SLOT) + wave * 64;
      p.cs_part[(size_t)(b * ((p.Lk + 127) >> 7) + tile) * (3 * p.nh * 64) + (1 + wave) * p.nh * 64 + h * 64 + lane] =
          (c[lane] + c[128 + lane]) + (c[256 + lane] + c[384 + lane]);
    }
  }
}

// ------------------------------------------------------------------------------ backward, round 3: dK, dV
// attn_bwd_dkv_kernel rebuilt on the same lines as attn_bwd3_dq_kernel.  S orientation (rows = queries in the accumulator registers,
// column = key = lane), one 32-query sub tile at a time:
//  * the wave's K / V fragments are negated (round 4: the scale * log2 e sits on the Q tile, rounded in LDS exactly as the forward
//    rounds its pre-scaled query fragments) and the chains start from the +lse / +delta values of the
//    sub tile's queries, read from LDS straight into the C operands: acc = lse - s, acc' = delta - dP, so P = exp2(-acc) (a source
//    modifier) and -dS = P * acc': 16 exp2 + 16 multiplies + 16 packed converts per 16 MFMAs; the sign goes into the final scale of dK;
//  * per-sequence buffer windows for the q / dO rows (rows past the sequence read as zeros: P = 1, dP = 0, dS = 0), scalar DMA
//    offsets, compile-time ring slots, lane-constant LDS bases + immediates, fragments requested a k-step ahead.
// A masked key's outputs are stored as zero (its P is not bounded by the saved log-sum-exp; the garbage stays in its own column).
namespace bwd3 {
constexpr int KV_STAGE = 16384 + 512;                // Q | dO (unified layout: b128 and transpose reads) | lse[64] | delta[64]
constexpr int KV_SMEM = 3 * KV_STAGE + 1024 + 512 + 16;   // + under dropout: 64 row keys (ia_rng_row) of the current query tile per wave; + 2 spare DMA slots; + 4 words (round 6: last live query per wave)
}

template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd3_dkv_kernel(AttnArgs p) {
  using namespace bwd3;
  __shared__ __attribute__((aligned(16))) char smem[KV_SMEM];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lk = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, h, b;
  attn_block_coords(p, p.Lk, tile, h, b);
  int Lq = p.Lq, L = p.Lk;
  size_t qbase = (size_t)b * Lq, rowbase = (size_t)b * L;
  if (p.cu) {
    const int c0 = p.cu[b];
    Lq = L = p.cu[b + 1] - c0;
    qbase = rowbase = (size_t)c0;
    if (tile * 128 >= L) return;
  }
  const int k0 = tile * 128 + wave * 32;
  const bool active = k0 < L;
  const int key = k0 + lk;
  const int kc = key < L ? key : L - 1;
  const bool key_ok = key < L && (p.mask == nullptr || p.mask[rowbase + kc] != 0);
  const __amdgpu_buffer_rsrc_t rsQ = ia_rsrc(p.q + qbase * p.ld_q + h * 64, (uint32_t)(((size_t)(Lq - 1) * p.ld_q + 64) * 2));
  const __amdgpu_buffer_rsrc_t rsG = ia_rsrc(p.d_o + qbase * p.ld_o + h * 64, (uint32_t)(((size_t)(Lq - 1) * p.ld_o + 64) * 2));
  const __amdgpu_buffer_rsrc_t rsL = ia_rsrc(p.lse2 + ((size_t)b * p.nh + h) * p.Lq, (uint32_t)Lq * 4u);
  const __amdgpu_buffer_rsrc_t rsD = ia_rsrc(p.delta + ((size_t)b * p.nh + h) * p.Lq, (uint32_t)Lq * 4u);
  const uint32_t sbase = lds_addr(smem);
  const uint32_t q_tile = (uint32_t)p.ld_q * 128u, q_half = (uint32_t)p.ld_q * 64u, g_tile = (uint32_t)p.ld_o * 128u, g_half = (uint32_t)p.ld_o * 64u;
  int nqt = (Lq + 63) >> 6;
  // lane constants
  uint32_t ka[4], t0, t1, t0x, t1x, dqu, dgu;
  {
    const uint32_t a0 = (uint32_t)(lk * 128 + ((hh ^ swz_u(lk)) << 4));
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) ka[kb] = sbase + (a0 ^ (uint32_t)(kb << 5));
    t0 = sbase + tr_lane_off_u(lane, 0); t1 = sbase + tr_lane_off_u(lane, 32); t0x = t0 ^ 32u; t1x = t1 ^ 32u;
    const int r0 = tid >> 3, c = tid & 7;
    dqu = (uint32_t)((r0 * p.ld_q + (c ^ swz_u(r0)) * 8) * 2);
    dgu = (uint32_t)((r0 * p.ld_o + (c ^ swz_u(r0)) * 8) * 2);
  }
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);        // one flat -> LDS cast; DMA destinations are LDS-pointer arithmetic from here
  auto stage_tile = [&](auto SLOT_T, int qt) {
    constexpr int S = decltype(SLOT_T)::value * KV_STAGE;
    const uint32_t sq = (uint32_t)qt * q_tile, sg = (uint32_t)qt * g_tile;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lsm + S + wave * 1024), 16, dqu, sq, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lsm + S + 4096 + wave * 1024), 16, dqu, sq + q_half, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lsm + S + 8192 + wave * 1024), 16, dgu, sg, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lsm + S + 12288 + wave * 1024), 16, dgu, sg + g_half, 0, 0);
    // 64 x fp32 each, one 4-byte-per-lane DMA piece: wave 0 the lse values, wave 1 delta; waves 2 / 3 issue a piece of zeros into a
    // spare slot, so that every wave has the same number of pieces in flight (the counted vmcnt waits assume it); out-of-range rows
    // read as zero
    {
      const uint32_t o4 = wave < 2 ? (uint32_t)(qt * 64 + lane) * 4u : OOB;
      lds_char* const dst = wave < 2 ? lsm + S + 16384 + wave * 256 : lsm + 3 * KV_STAGE + 1024 + (wave - 2) * 256;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 1 ? rsD : rsL, dst, 4, o4, 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
  const uint32_t stream_id = (uint32_t)(b * p.nh + h);
  const uint32_t pc = pair_c_of(key), ush = (uint32_t)(key & 1) * 16u;
  const uint32_t thr1 = DROPOUT ? p.thr16 - 1u : 0u, seed = p.seed;
  const float sc = p.sc, inv_keep = p.inv_keep;
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
  lds_u32* const s_rk = (lds_u32*)IA_LDS(smem + 3 * KV_STAGE) + wave * 64;

  // ---- prologue: this wave's 32 key rows of K and V through wave-private slots of stages 1-2, query tile 0 into stage 0
  char* kslot = smem + KV_STAGE + wave * 8192;
  stage_rows32(ia_rsrc(p.k, p.kv_bytes), kslot, rowbase + k0, L - k0, p.ld_kv, h * 64, lane);
  stage_rows32(ia_rsrc(p.v, p.kv_bytes), kslot + 4096, rowbase + k0, L - k0, p.ld_kv, h * 64, lane);
  stage_tile(S0{}, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const bf16x8 kr = frag_b128(kslot, lk, kb * 2 + hh), vr = frag_b128(kslot + 4096, lk, kb * 2 + hh);
#pragma unroll
    for (int j = 0; j < 8; ++j) { kf[kb][j] = f2bf(-bf2f(kr[j])); vf[kb][j] = f2bf(-bf2f(vr[j])); }
  }
  const bool scale_q = PRESCALE && !p.q_prescaled;        // q arrives pre-scaled from the QKV projection: nothing to do per tile
  if (scale_q) {                                          // query tile 0 (stage 0) is pre-scaled here, see prescale() below
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bf16x8* const qp = reinterpret_cast<bf16x8*>(smem + i * 4096 + wave * 1024 + lane * 16);
      bf16x8 v = *qp;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * p.sc);
      *qp = v;
    }
  }
  // AttnArgs::dead_queries (round 6): the query tiles behind the last unmasked position bring dO == 0 (the caller's guarantee) -- nothing
  // for dK / dV -- so the query loop ends with that position's tile (right-padded batches: ~45 % of the tiles at the bench's lengths).
  // Each wave scans a quarter of the positions with ballots; the four maxima meet in LDS across the prologue's barrier.
  int* const s_last = reinterpret_cast<int*>(smem + 3 * KV_STAGE + 1024 + 512);
  const bool trim = p.dead_queries && p.mask != nullptr && p.cu == nullptr && p.Lq == p.Lk;
  if (trim) {
    int last = -1;
    for (int base = wave * 64; base < Lq; base += 256) {
      const int pos = base + lane;
      const bool ok = pos < Lq && p.mask[rowbase + pos] != 0;
      const uint64_t bal = __ballot(ok);
      if (bal) last = base + 63 - __builtin_clzll(bal);
    }
    if (lane == 0) s_last[wave] = last;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                           // the K / V row slots (stages 1-2) are free for query tiles
  if (trim) {
    const int last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    const int live = __builtin_amdgcn_readfirstlane(last < 0 ? 1 : (last >> 6) + 1);
    if (live < nqt) nqt = live;
  }
  if (nqt > 1) stage_tile(S1{}, 1);
  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();

  // Q <- bf16(Q * scale * log2 e) in place, each wave on the two 1-KiB pieces of a query tile it fetched itself: exactly the operand
  // the forward and the dQ kernel multiply with K, so the P recomputed here is the P the saved log-sum-exp and delta belong to
  // (round 3 scaled the K fragments instead: a different rounding of every score, P rows that no longer sum to one).  Tile 0 in the
  // prologue; tile qt+1 between the two sub tiles of tile qt, where the matrix pipe is busy (between the wait and the barrier at the
  // top of a tile the same work cost the ViT launch 130 us).
  auto prescale = [&](int stage_base) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bf16x8* const qp = reinterpret_cast<bf16x8*>(smem + stage_base + i * 4096 + wave * 1024 + lane * 16);
      bf16x8 v = *qp;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * sc);
      *qp = v;
    }
  };
  auto tile_step = [&](auto SLOT_T, int qt) {
    constexpr int SB = decltype(SLOT_T)::value * KV_STAGE;
    using NEXT2 = std::integral_constant<int, (decltype(SLOT_T)::value + 2) % 3>;
    // query tile qt has landed for this wave (tile qt+1, issued behind it: 5 pieces, may still be in flight), then for everybody
    if (qt + 1 >= nqt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's pre-scaling writes into tile qt (below) are done
    __builtin_amdgcn_s_barrier();
    if (qt + 2 < nqt) stage_tile(NEXT2{}, qt + 2);
    if (!active) {                                        // a wave without keys still owns DMA pieces of every query tile: pre-scale them
      if (scale_q && qt + 1 < nqt) {
        if (qt + 2 < nqt) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        prescale(((decltype(SLOT_T)::value + 1) % 3) * KV_STAGE);
      }
      return;
    }
    if (DROPOUT) {      // row keys of this tile's 64 queries, one per lane (the wave's LDS operations complete in order: no barrier)
      s_rk[lane] = ia_rng_row(seed, stream_id, (uint32_t)(qt * 64 + lane));
      __builtin_amdgcn_wave_barrier();
    }
    auto sub_tile = [&](auto QS) {
      constexpr int qs = decltype(QS)::value;
      constexpr int QB = SB + qs * 4096, GB = SB + 8192 + qs * 4096;        // the sub tile's 32 rows of the Q / dO tiles
      // C operands: +lse / +delta of the rows this lane's accumulator registers stand for (q = 8 rg + 4 hh + j)
      f32x16 s, dp, dlr;
      {
        const float* sL = reinterpret_cast<const float*>(smem + SB + 16384) + qs * 32 + 4 * hh;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const f32x4 ls = *reinterpret_cast<const f32x4*>(sL + 8 * rg);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(sL + 64 + 8 * rg);
#pragma unroll
          for (int j = 0; j < 4; ++j) { s[rg * 4 + j] = PRESCALE ? ls[j] : ls[j] / sc; dlr[rg * 4 + j] = dl[j]; dp[rg * 4 + j] = DROPOUT ? 0.f : dl[j]; }
        }
      }
      bf16x8 xq, xg, yq, yg;                              // Q / dO fragments of even / odd k-steps
      xq = lds_read_b128<QB>(ka[0]); xg = lds_read_b128<GB>(ka[0]);
      yq = lds_read_b128<QB>(ka[1]); yg = lds_read_b128<GB>(ka[1]);
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xq), "+v"(xg));
      s = mfma(xq, kf[0], s); dp = mfma(xg, vf[0], dp);
      xq = lds_read_b128<QB>(ka[2]); xg = lds_read_b128<GB>(ka[2]);
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(yq), "+v"(yg));
      s = mfma(yq, kf[1], s); dp = mfma(yg, vf[1], dp);
      yq = lds_read_b128<QB>(ka[3]); yg = lds_read_b128<GB>(ka[3]);
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xq), "+v"(xg));
      s = mfma(xq, kf[2], s); dp = mfma(xg, vf[2], dp);
      TrPair g0, a0, g1, a1;                              // dO^T and Q^T fragments of the sub tile's two 16-query steps
      read_tr<SB + 8192, qs * 32>(g0, t0, t1, t0x, t1x);
      read_tr<SB, qs * 32>(a0, t0, t1, t0x, t1x);
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(yq), "+v"(yg));
      s = mfma(yq, kf[3], s); dp = mfma(yg, vf[3], dp);
      read_tr<SB + 8192, qs * 32 + 16>(g1, t0, t1, t0x, t1x);
      read_tr<SB, qs * 32 + 16>(a1, t0, t1, t0x, t1x);
      bf16x8 pf[2], sf[2];
      u32x4 rkq[4];
      if (DROPOUT) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) rkq[rg] = *(const lds_u32x4*)(s_rk + qs * 32 + 8 * rg + 4 * hh);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(PRESCALE ? -s[r] : -s[r] * sc);      // exp2(s' - lse)
        float pd = pv, nds;
        if (DROPOUT) {
          const uint32_t hsh = ia_rng_pair(rkq[r >> 2][r & 3], pc);
          const uint32_t dr = (hsh >> ush) & 0xFFFFu;                                    // this key's half of the pair's draw
          const float mk = __builtin_fminf(__builtin_fmaxf((float)((int)dr - (int)thr1), 0.f), 1.f);      // 0 iff dropped (draw < thr16), else 1
          pd = pv * mk;
          // dp = -dP here; -(M dP / keep - delta) = dp mk / keep + delta
          nds = pv * __builtin_fmaf(dp[r] * mk, inv_keep, dlr[r]);
        } else {
          nds = pv * dp[r];
        }
        pf[r >> 3][r & 7] = f2bf(pd);
        sf[r >> 3][r & 7] = f2bf(nds);
      }
      // dV^T[d][key] += dO^T[d][q] P[q][key] ; -dK^T[d][key] += Q^T[d][q] (-dS)[q][key]
      tr_wait<12>(g0);
      dv0 = mfma(g0.a0(), pf[0], dv0); dv1 = mfma(g0.a1(), pf[0], dv1);
      tr_wait<8>(a0);
      dk0 = mfma(a0.a0(), sf[0], dk0); dk1 = mfma(a0.a1(), sf[0], dk1);
      tr_wait<4>(g1);
      dv0 = mfma(g1.a0(), pf[1], dv0); dv1 = mfma(g1.a1(), pf[1], dv1);
      tr_wait<0>(a1);
      dk0 = mfma(a1.a0(), sf[1], dk0); dk1 = mfma(a1.a1(), sf[1], dk1);
    };
    sub_tile(std::integral_constant<int, 0>{});
    if (scale_q && qt + 1 < nqt) {
      // tile qt+1 (issued a whole step ago) has landed for this wave: only the 5 pieces of tile qt+2, just issued, may be in flight
      if (qt + 2 < nqt) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      prescale(((decltype(SLOT_T)::value + 1) % 3) * KV_STAGE);
    }
    if (qt * 64 + 32 < Lq) sub_tile(std::integral_constant<int, 1>{});
  };
  {
    int qt = 0;
    for (;;) {
      tile_step(S0{}, qt); if (++qt >= nqt) break;
      tile_step(S1{}, qt); if (++qt >= nqt) break;
      tile_step(S2{}, qt); if (++qt >= nqt) break;
    }
  }
  __builtin_amdgcn_s_barrier();                           // the epilogue rows are staged in the ring
  float* cs_lds = p.cs_part ? reinterpret_cast<float*>(smem + 8 * EPI_SLOT) + wave * 128 : nullptr;     // behind the eight store slots: dk | dv sums
  if (!active && !cs_lds) return;
  if (active) {
    // dK = scale sum dS q; the accumulators hold sum (-dS) q' with q' = q scale log2 e
    store_block_rows(smem + wave * 2 * EPI_SLOT, dk0, dk1, PRESCALE ? -1.f / LOG2E : -p.scale, !key_ok, p.dk + (rowbase + k0) * p.ld_dkv + h * 64, p.ld_dkv,
                     L - k0, lane, cs_lds);
    store_block_rows(smem + (wave * 2 + 1) * EPI_SLOT, dv0, dv1, DROPOUT ? p.inv_keep : 1.f, !key_ok, p.dv + (rowbase + k0) * p.ld_dkv + h * 64,
                     p.ld_dkv, L - k0, lane, cs_lds ? cs_lds + 64 : nullptr);
  } else { zero_cs_row(cs_lds, lane); zero_cs_row(cs_lds + 64, lane); }
  if (cs_lds) {
    __syncthreads();
    if (wave < 2) {         // wave 0: the dk columns, wave 1: the dv columns
      const float* c = reinterpret_cast<const float*>(smem + 8 * EPI_SLOT) + wave * 64;
      p.cs_part[(size_t)(b * ((p.Lk + 127) >> 7) + tile) * (3 * p.nh * 64) + (1 + wave) * p.nh * 64 + h * 64 + lane] =
          (c[lane] + c[128 + lane]) + (c[256 + lane] + c[384 + lane]);
    }
  }
}

// ------------------------------------------------------------------------- backward, round 4: ONE kernel for L <= 256
// Self-attention backward of a whole (sequence, head) item by one 512-thread workgroup (8 waves, two per SIMD), persistent over
// items.  dQ, dK and dV come out of ONE evaluation of S, P, dP and dS per 32 x 32 block: 20 MFMA-equivalents per block instead of the
// 12 + 16 of the dQ / dK,dV kernel pair, one exp2 per score instead of two, and q, k, v, dO, o are read from HBM once (the pair reads
// q, k, v, dO twice).  Layout of the work:
//  * wave w owns key block w (32 keys; L <= 256 -> at most 8 blocks): its K / V fragments sit in registers as the B operands of
//    S = Q' K^T and dP = dO V^T (S orientation of the dK/dV kernels: rows = queries in the accumulator registers, column = key =
//    lane), dK and dV accumulate in registers over the item's query blocks;
//  * the query blocks (Q | dO | O rows, lse) stream through a 4-deep LDS ring, one block per step, fetched 4 steps ahead by LDS-DMA
//    (continuously across items); before a block is published, the waves that fetched its pieces prepare it: Q <- bf16(Q * scale *
//    log2 e) in place (exactly the forward's pre-scaled operand, so the recomputed P is the forward's P), delta = rowsum(dO o O);
//  * dQ needs the contraction over the keys, i.e. over lanes AND waves: every wave writes its -dS block (bf16, [key][query]) into
//    an exchange slot; behind the step's barrier wave (dq, qh) forms the 16 x 16 tile dQ^T[16 dq .. +16][16 qh .. +16] of the step's
//    query block with MFMA 16x16x32 over all key blocks (A = K^T fragments held in registers for the whole item, B = transpose
//    reads of the exchange slots) and stores it; one workgroup barrier per step;
//  * the next item's K / V / mask bytes are fetched into a stage during the current item, so nothing but the very first fill is
//    exposed.  No global load is issued inside the loop except through LDS-DMA (a register load would drain the in-order VMEM
//    queue), and every wait on the DMA is a counted vmcnt that is exact or stricter than needed whether or not stores are counted.
// Deterministic (no atomics).  A masked / out-of-range key gets +1e30 on its column's reference (P = 0 exactly).
#ifndef IA_FUSED_SWAP
#define IA_FUSED_SWAP 1      // 1: the second wave of every SIMD runs its part A one step ahead inside an item (below); 0: all waves in step
#endif
#ifndef IA_FUSED_ABL
#define IA_FUSED_ABL 0       // development: timing ablations of the fused backward (tools/abl/run_fused_abl.sh); results are wrong when != 0
#endif
namespace bwdf {
using namespace bwd3;
constexpr int ABL = IA_FUSED_ABL;
constexpr int NW = 8, RING = 4;
// ring slot: Q | dO | O rows of one 32-query block (4 KiB each, unified layout) | lse[32] | 128 B the lse piece's out-of-range lanes
// zero | delta[32] | dropout row keys[32].  The lse piece is a 64-lane DMA of wave 0 whose lanes 32 .. 63 write zeros behind the 32
// words: nothing another wave writes may live there -- delta did until round 4's last day, and a late lse piece (first item of a
// workgroup, cold lse lines) zeroed the delta waves 4 .. 7 had just written: dQ / dK of that block off, dV fine, one launch in ten.
constexpr int SL_Q = 0, SL_G = 4096, SL_O = 8192, SL_LSE = 12288, SL_DLT = SL_LSE + 256, SL_RK = SL_LSE + 384, SLOT = SL_LSE + 512;
constexpr int DLT_F = (SL_DLT - SL_LSE) / 4;          // delta's offset from lse in floats
constexpr int KST = 0, VST = 32768, RING_OFF = 65536, X_OFF = RING_OFF + RING * SLOT;
// Exchange tile [32 keys][32 queries] bf16.  Round 6 layout: DENSE 64-byte rows, the 8-byte chunk c of row r stored at chunk position
// c ^ xkey(r), xkey(r) = ((r >> 2) & 1) << 2 | (r >> 3) & 3 -- conflict-free for BOTH access forms: a 32-lane pass of the part-A b64 writes
// is 32 rows x one chunk (the eight rows that share r mod 4, i.e. a 16-bank group, differ in (r >> 2): xkey is a bijection of those
// three bits), a 32-lane pass of the part-C transpose reads is 8 rows x 32 bytes (rows r and r + 4 share a bank group and take
// different 32-byte halves: xkey's bit 2 = bit 2 of r).  Rounds 4-5 used 72-byte rows: conflict-free writes, but rows r and r + 7 of a
// read pass overlapped in six banks -- every read a two-way conflict (26 % of the kernel's LDS-active cycles in SQ_LDS_BANK_CONFLICT).
// The tile keeps its 2304-byte slot (2 x XT = the epilogue staging slot of store_block_rows).
constexpr int XP = 64, XT = 2304;
IA_DEV int xkey(int r) { return (((r >> 2) & 1) << 2) | ((r >> 3) & 3); }
static_assert(2 * XT == EPI_SLOT && 32 * XP <= XT && XT % 64 == 0, "a wave's two exchange tiles double as its epilogue staging slot");
constexpr int CS_OFF = X_OFF + NW * 2 * XT;          // column sums: 8 x (dk[64] | dv[64]) + 8 x dq tile[16] floats
constexpr int MSK_OFF = CS_OFF + NW * 128 * 4 + NW * 16 * 4;      // the next item's attendable-key bits: 8 words (one per key block), 256 B reserved
constexpr int DUMMY_OFF = MSK_OFF + 256;             // 8 x 256 B: targets of the count-equalising out-of-range DMA pieces
constexpr int F_SMEM = DUMMY_OFF + NW * 256;
static_assert(F_SMEM <= 160 * 1024, "LDS budget");
constexpr float PEN = 1e30f;

IA_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
IA_DEV bf16x8 join(s16x4 lo, s16x4 hi) { s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; return __builtin_bit_cast(bf16x8, r); }
template <int N> IA_DEV void wait2(s16x4& a, s16x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
IA_DEV bf16x8 frag_u(const char* s, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(s + row * 128 + ((chunk ^ swz_u(row)) << 4));
}
// wait until at most n VMEM operations of this wave are outstanding (n is wave-uniform, one of the values the schedule uses)
IA_DEV void wait_vm(int n) {
  if (n >= 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
}  // namespace bwdf

template <bool DROPOUT>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(AttnArgs p) {
  using namespace bwdf;
  __shared__ __attribute__((aligned(128))) char smem[F_SMEM];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, lk = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.Lq, nb = (L + 31) >> 5, nh = p.nh;
  // items (sequence, head), head fastest; every XCD takes a contiguous run of items (the heads of a sequence share its token rows)
  const int nitems = p.B * nh;
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int per_x = (nitems + 7) >> 3;
  const int x_lo = xcd * per_x, x_hi = (x_lo + per_x < nitems) ? x_lo + per_x : nitems;
  const int first_item = x_lo + wslot;
  if (first_item >= x_hi) return;
  const int cnt = (x_hi - first_item + wpx - 1) / wpx;
  // item cursors carry (item, sequence, head) and advance by the workgroup stride without divisions
  struct Cur { int it, b, h; };
  const int wpx_b = wpx / nh, wpx_h = wpx - wpx_b * nh;
  auto adv = [&](Cur& c) { c.it += wpx; c.b += wpx_b; c.h += wpx_h; if (c.h >= nh) { c.h -= nh; ++c.b; } };

  const uint32_t sbase = lds_addr(smem);
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);

  // ---- lane constants
  // fragment read bases inside a ring slot (slot offset added per step; RING_OFF and SLOT are multiples of 128, so the k-step /
  // row-half XORs of 32 / 64 / 96 bytes can be applied to the absolute address): b128 reads of row lk, k-step 0; transpose reads
  // of columns 0 .. 31 (t0) and 32 .. 63 (t0 + t1d)
  const uint32_t ka0 = sbase + RING_OFF + (uint32_t)(lk * 128 + ((hh ^ swz_u(lk)) << 4));
  const uint32_t t0 = sbase + RING_OFF + tr_lane_off_u(lane, 0), t1d = tr_lane_off_u(lane, 32) - tr_lane_off_u(lane, 0);
  const int grp = wave >> 2;                              // the two waves of a SIMD (w, w + 4) are in different groups
  const int dqt = wave & 3, qh = wave >> 2;               // this wave's dQ^T tile: d 16 dqt .. +16, queries 16 qh .. +16 of the block
  const int g4 = lane >> 4, p16 = lane & 15;
  // transpose-read bases for MFMA 16x16x32 operands: 16-lane group g4 reads rows 4 g4 .. +3 (second read: + 16 rows) of 16 columns
  const uint32_t ktb = sbase + KST + (uint32_t)((4 * g4 + (p16 >> 2)) * 128 + ((((16 * dqt + 4 * (p16 & 3)) >> 3) ^ swz_u(4 * g4 + (p16 >> 2))) << 4) +
                                                ((4 * (p16 & 3)) & 7) * 2);
  // part C: rows 4 g4 + (p16 >> 2) (second read: + 16 rows, whose key differs in bit 1: address ^ 16, + 16 rows), chunk 4 qh + (p16 & 3)
  const uint32_t xrb = sbase + X_OFF + (uint32_t)((4 * g4 + (p16 >> 2)) * XP + (((4 * qh + (p16 & 3)) ^ xkey(4 * g4 + (p16 >> 2))) << 3));
  // part A: this lane's row of the wave's exchange tile, at its chunk hh; its chunks hh + 2 j (j = 0 .. 3) sit at this address ^ (j << 4)
  // (hh + 2 j = hh ^ 2 j, the row base is a multiple of 64 and the XOR stays inside the row)
  const int xwo = X_OFF + wave * 2 * XT + lk * XP + ((hh ^ xkey(lk)) << 3);
  const int r8 = lane >> 3, c8 = lane & 7;                // DMA piece: 8 rows x 8 chunks of 16 bytes
  // ---- DMA issue.  Every tensor keeps ONE descriptor base; per item only its size field moves: "up to the end of the item's last
  // row of this head", so rows >= L and the pieces of items past the end read as zeros without per-lane selects, the item / block /
  // piece position is the instruction's scalar offset, and the only lane-dependent operand is the piece-local offset (row r8 of the
  // piece, swizzled chunk).  Piece pc of a 32-row block covers rows 8 pc .. 8 pc + 7: swz_u of odd pieces = swz_u of even pieces | 2,
  // i.e. the lane offset XOR 32 bytes.  Every wave issues the same number of pieces at every point of the schedule (out-of-range
  // dummies fill up), so one set of counted waits serves all waves.
  const uint32_t lo_q = (uint32_t)((r8 * p.ld_q + ((c8 ^ swz_u(r8)) << 3)) * 2);
  const uint32_t lo_kv = (uint32_t)((r8 * p.ld_kv + ((c8 ^ swz_u(r8)) << 3)) * 2);
  const uint32_t lo_o = (uint32_t)((r8 * p.ld_o + ((c8 ^ swz_u(r8)) << 3)) * 2);
  const uint32_t lo_w = lane < 32 ? (uint32_t)lane * 4u : OOB;      // 32 words
  const uint32_t lo_m = lane < 8 ? (uint32_t)lane * 4u : OOB;       // 8 words
  // bytes from the tensor's base to the end of row L-1 of the item's head column block; 0 for an item past the end
  auto lim = [&](const Cur& c, int ld) { return c.it < x_hi ? (uint32_t)((((uint32_t)c.b * L + L - 1) * ld + c.h * 64 + 64) * 2) : 0u; };
  auto org = [&](const Cur& c, int ld, int row) { return (uint32_t)((((uint32_t)c.b * L + row) * ld + c.h * 64) * 2); };
  // ring pieces of query block j of item c into ring slot `slot`: 2 pieces per wave
  auto issue_ring = [&](const Cur& c, int j, int slot) {
    const int so = RING_OFF + slot * SLOT;
    if (wave < 4) {
      const uint32_t vo = lo_q ^ (uint32_t)((wave & 1) << 5);
      const __amdgpu_buffer_rsrc_t wq = ia_rsrc(p.q, lim(c, p.ld_q));
      const uint32_t sof = org(c, p.ld_q, j * 32 + wave * 8);      // (a lambda call written directly as a builtin argument makes hipcc
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wq, lsm + so + SL_Q + wave * 1024, 16, vo, sof, 0, 0);      // drop the kernel's host stub)
      const bool real = wave == 0 && c.it < x_hi;
      const __amdgpu_buffer_rsrc_t wl = ia_rsrc(p.lse2, real ? (uint32_t)(c.it * L + L) * 4u : 0u);
      const int dst = wave == 0 ? so + SL_LSE : DUMMY_OFF + wave * 256;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wl, lsm + dst, 4, lo_w, (uint32_t)(c.it * L + j * 32) * 4u, 0, 0);
    } else {
      const int pc = wave - 4;
      const uint32_t vo = lo_o ^ (uint32_t)((pc & 1) << 5), sof = org(c, p.ld_o, j * 32 + pc * 8);
      const __amdgpu_buffer_rsrc_t wg = ia_rsrc(p.d_o, lim(c, p.ld_o)), wo = ia_rsrc(p.o, lim(c, p.ld_o));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wg, lsm + so + SL_G + pc * 1024, 16, vo, sof, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wo, lsm + so + SL_O + pc * 1024, 16, vo, sof, 0, 0);
    }
  };
  // K / V rows and the attendable-key words of item c into the stage: 9 pieces per wave (its own key block + a word piece)
  auto issue_kv = [&](const Cur& c) {
    const __amdgpu_buffer_rsrc_t wk = ia_rsrc(p.k, lim(c, p.ld_kv)), wv = ia_rsrc(p.v, lim(c, p.ld_kv));
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      const uint32_t vo = lo_kv ^ (uint32_t)((pc & 1) << 5), sof = org(c, p.ld_kv, wave * 32 + pc * 8);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wk, lsm + KST + wave * 4096 + pc * 1024, 16, vo, sof, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wv, lsm + VST + wave * 4096 + pc * 1024, 16, vo, sof, 0, 0);
    }
    // attendable-key bits per sequence: 8 words (bit i of word k: key 32 k + i in range and not masked), built by
    // attn_key_bits_kernel into the `delta` scratch the pair kernels use for their hand-over (this kernel needs none)
    const bool real = wave == 0 && c.it < x_hi;
    const __amdgpu_buffer_rsrc_t wm = ia_rsrc(p.delta, real ? (uint32_t)(c.b * 8 + 8) * 4u : 0u);
    const int dst = wave == 0 ? MSK_OFF : DUMMY_OFF + wave * 256;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wm, lsm + dst, 4, lo_m, (uint32_t)c.b * 32u, 0, 0);
  };
  // preparation of a landed block by the waves that fetched its pieces (before the barrier that publishes it)
  auto prep = [&](int it, int j, int slot) {
    char* const sl = smem + RING_OFF + slot * SLOT;
    if (wave < 4) {
      if (!p.q_prescaled) {
        bf16x8* const qp = reinterpret_cast<bf16x8*>(sl + SL_Q + wave * 1024 + lane * 16);
        bf16x8 v = *qp;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f2bf(bf2f(v[i]) * p.sc);
        *qp = v;
      }
      if (DROPOUT && wave == 1 && lane < 32)
        *reinterpret_cast<uint32_t*>(sl + SL_RK + lane * 4) = ia_rng_row(p.seed, (uint32_t)it, (uint32_t)(j * 32 + lane));
    } else {
      const bf16x8 gv = *reinterpret_cast<const bf16x8*>(sl + SL_G + (wave - 4) * 1024 + lane * 16);
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(sl + SL_O + (wave - 4) * 1024 + lane * 16);
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) d += bf2f(gv[i]) * bf2f(ov[i]);
      d += ia_dpp<0xB1>(d); d += ia_dpp<0x4E>(d); d += ia_dpp<0x141>(d);      // the 8 lanes of a row
      if (c8 == 0) *reinterpret_cast<float*>(sl + SL_DLT + ((wave - 4) * 8 + r8) * 4) = d;
    }
  };

  const uint32_t thr1 = DROPOUT ? p.thr16 - 1u : 0u;
  const float inv_keep = p.inv_keep;
  const int ntile = (L + 127) >> 7;

  // ---- prologue: stage of item 0, ring blocks 0 .. RING-1, preparation of block 0
  Cur prod{first_item, first_item / nh, first_item % nh}; // producer cursor of the ring (item, block pj)
  Cur cons = prod, kvn = prod;                            // consumer cursor; the item whose K / V is staged next
  int pj = 0;
  issue_kv(kvn); adv(kvn);
#pragma unroll 1
  for (int s = 0; s < RING; ++s) {
    issue_ring(prod, pj, s);
    if (++pj == nb) { pj = 0; adv(prod); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  prep(cons.it, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 kf[4], vf[4], KT[8];
  f32x16 dk0, dk1, dv0, dv1;
  f32x4 csq;
  bool key_ok = false, mine = false;
  uint32_t vbits = 0u;                                    // bit k: key block k has an attendable key
  float pen = 0.f;
  bool any_bad = false;
  uint32_t pc = 0u, ush = 0u;
  // One trip per step g, closed by the step's barrier: part C of step g-1 (this wave's dQ tile from all exchange slots, which the
  // previous barrier published), then part A of step g (S, dP, P, dS of this wave's key block, dV, dK, -dS to the exchange slot).
  // Run like that by all eight waves, the two waves of a SIMD hit the matrix pipe together and leave it idle together (measured:
  // MFMA busy 19 %, every part's time adding up).  So the second wave of every SIMD (waves 4 .. 7) runs its part A one step AHEAD:
  // behind the barrier of trip g it already does part A of step g+1 -- whose block that barrier has just published -- and in trip
  // g+1 only part C.  One wave's exp2 / multiply / convert stretch then has the other's MFMA stretch beside it (626 against 667 us
  // at 512 x 255 x 16).  Every wave still arrives at every barrier once; the dependencies are the same ones (part C of step g needs
  // every wave's part A of step g in front of barrier g: group 1 finished it a trip earlier; an exchange tile is rewritten two steps
  // later, behind the barrier every reader arrives at after reading it).  At an item boundary the run-ahead pauses: the epilogue
  // needs dK / dV and part C the K^T fragments of the old item before the item start overwrites them.
  const int G = cnt * nb;
  int cj = 0;                                             // block of step g inside its item
  Cur prev = cons;                                        // item of step g-1
  int pcj = 0;                                            // its block
#pragma unroll 1
  for (int g = 0;; ++g) {
    const bool haveA = g < G;
    const bool first = cj == 0, last = cj == nb - 1;
    // blk: the query block of step ga inside its item.  Round 6: with AttnArgs::dead_queries a block whose 32 positions are all masked
    // (right padding: 115 of 255 positions on the bench's ragged batches) is skipped -- its dO rows are exactly zero (the caller's
    // guarantee: no output of a masked position reaches the loss, the position is masked as a key in every layer), hence dS = 0: nothing
    // for dK / dV, and part C stores zeros for its dQ rows.  The step keeps its barrier, its DMA and its place in the ring.
    auto part_a = [&](int ga, bool first_a, int blk) {
      {
        const int slot = ga & (RING - 1), par = ga & 1;
        if (first_a) {
          // ---- item start: this wave's K / V fragments (B operands), the K^T fragments of its dQ tile, the mask
          const char* kb_ = smem + KST + wave * 4096;
          const char* vb_ = smem + VST + wave * 4096;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const bf16x8 kr = frag_u(kb_, lk, kb * 2 + hh), vr = frag_u(vb_, lk, kb * 2 + hh);
#pragma unroll
            for (int i = 0; i < 8; ++i) { kf[kb][i] = f2bf(-bf2f(kr[i])); vf[kb][i] = f2bf(-bf2f(vr[i])); }
          }
          {
            s16x4 lo[8], hi[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { lo[k] = tr_read<0>(ktb + k * 4096); hi[k] = tr_read<2048>(ktb + k * 4096); }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]));
            asm volatile("" : "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]));
#pragma unroll
            for (int k = 0; k < 8; ++k) KT[k] = join(lo[k], hi[k]);
          }
          const int key = wave * 32 + lk;
          // attendable keys of the item: one word per key block (from the stage)
          {
            const uint32_t* const mw = reinterpret_cast<const uint32_t*>(smem + MSK_OFF);
            uint32_t vb = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) vb |= (mw[k] != 0u ? 1u : 0u) << k;
            vbits = __builtin_amdgcn_readfirstlane(vb);
            key_ok = ((mw[wave] >> lk) & 1u) != 0u;
          }
          mine = ((vbits >> wave) & 1u) != 0u;            // this wave's key block takes part
          pen = key_ok ? 0.f : PEN;
          any_bad = __ballot(!key_ok) != 0ull;
          pc = pair_c_of(key); ush = (uint32_t)(key & 1) * 16u;
          dk0 = zero16(); dk1 = zero16(); dv0 = zero16(); dv1 = zero16();
          csq = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- part A of step g
        if (mine && (!p.dead_queries || ((vbits >> blk) & 1u))) {
          const uint32_t so = (uint32_t)(slot * SLOT);
          f32x16 s, dp;
          const float* const sL = reinterpret_cast<const float*>(smem + RING_OFF + slot * SLOT + SL_LSE) + 4 * hh;
          {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const f32x4 ls = *reinterpret_cast<const f32x4*>(sL + 8 * rg);
              const f32x4 dl = *reinterpret_cast<const f32x4*>(sL + DLT_F + 8 * rg);
#pragma unroll
              for (int j = 0; j < 4; ++j) { s[rg * 4 + j] = ls[j]; dp[rg * 4 + j] = DROPOUT ? 0.f : dl[j]; }
            }
            if (any_bad) {                                // wave-uniform: masked / out-of-range keys end up with P = exp2(-1e30) = 0
#pragma unroll
              for (int r = 0; r < 16; ++r) s[r] += pen;
            }
          }
          const uint32_t k0_ = ka0 + so;
          bf16x8 xq, xg, yq, yg;
          if (!(ABL & 128)) {
          xq = lds_read_b128<SL_Q>(k0_); xg = lds_read_b128<SL_G>(k0_);
          yq = lds_read_b128<SL_Q>(k0_ ^ 32u); yg = lds_read_b128<SL_G>(k0_ ^ 32u);
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xq), "+v"(xg));
          s = mfma(xq, kf[0], s); dp = mfma(xg, vf[0], dp);
          xq = lds_read_b128<SL_Q>(k0_ ^ 64u); xg = lds_read_b128<SL_G>(k0_ ^ 64u);
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(yq), "+v"(yg));
          s = mfma(yq, kf[1], s); dp = mfma(yg, vf[1], dp);
          yq = lds_read_b128<SL_Q>(k0_ ^ 96u); yg = lds_read_b128<SL_G>(k0_ ^ 96u);
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xq), "+v"(xg));
          s = mfma(xq, kf[2], s); dp = mfma(xg, vf[2], dp);
          }
          const uint32_t a0_ = t0 + so, a1_ = a0_ + t1d, a0x = a0_ ^ 32u, a1x = a1_ ^ 32u;
          TrPair g0, q0, g1, q1;                          // dO^T and Q'^T fragments of the block's two 16-query steps
          if (!(ABL & 64)) {
          read_tr<SL_G, 0>(g0, a0_, a1_, a0x, a1x);
          read_tr<SL_Q, 0>(q0, a0_, a1_, a0x, a1x);
          }
          if (!(ABL & 128)) {
          asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(yq), "+v"(yg));
          s = mfma(yq, kf[3], s); dp = mfma(yg, vf[3], dp);
          }
          bf16x8 pf[2], sf[2];
          const uint32_t* const rkp = reinterpret_cast<const uint32_t*>(smem + RING_OFF + slot * SLOT + SL_RK) + 4 * hh;
          // probabilities and -dS of 8 accumulator rows (one 16-query MFMA step): HALF = 0 / 1
          auto half = [&](auto HALF) {
            constexpr int hf = decltype(HALF)::value;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const int r = hf * 8 + rr;
              const float pv = (ABL & 2) ? s[r] : __builtin_amdgcn_exp2f(-s[r]);      // exp2(q' . k - lse)
              float pd = pv, nds;
              if (ABL & 2) nds = dp[r];
              else if (DROPOUT) {
                const int qo = 8 * (r >> 2) + (r & 3);    // this register's query inside the block, minus 4 hh
                const uint32_t dr = (ia_rng_pair(rkp[qo], pc) >> ush) & 0xFFFFu;
                const float mk = __builtin_fminf(__builtin_fmaxf((float)((int)dr - (int)thr1), 0.f), 1.f);      // 0 iff dropped
                pd = pv * mk;
                nds = pv * __builtin_fmaf(dp[r] * mk, inv_keep, sL[DLT_F + qo]);      // dp = -dP: -(M dP / keep - delta)
              } else {
                nds = pv * dp[r];                         // dp = delta - dP
              }
              pf[hf][rr] = f2bf(pd);
              sf[hf][rr] = f2bf(nds);
            }
            // -dS^T to the exchange slot: row = key (this lane), two groups of 4 consecutive queries
            const int xa = xwo + par * XT;
            if (!(ABL & 8)) {
            *reinterpret_cast<bf16x4*>(smem + (xa ^ (hf * 32))) = bf16x4{sf[hf][0], sf[hf][1], sf[hf][2], sf[hf][3]};
            *reinterpret_cast<bf16x4*>(smem + (xa ^ (hf * 32 + 16))) = bf16x4{sf[hf][4], sf[hf][5], sf[hf][6], sf[hf][7]};
            }
          };
          half(std::integral_constant<int, 0>{});
          // the second step's fragments are requested only now: the first step's MFMAs run under the second half's exp2 / multiplies
          if (!(ABL & 64)) {
          read_tr<SL_G, 16>(g1, a0_, a1_, a0x, a1x);
          read_tr<SL_Q, 16>(q1, a0_, a1_, a0x, a1x);
          tr_wait<12>(g0);
          dv0 = mfma(g0.a0(), pf[0], dv0); dv1 = mfma(g0.a1(), pf[0], dv1);
          tr_wait<8>(q0);
          dk0 = mfma(q0.a0(), sf[0], dk0); dk1 = mfma(q0.a1(), sf[0], dk1);
          }
          half(std::integral_constant<int, 1>{});
          if (!(ABL & 64)) {
          tr_wait<4>(g1);
          dv0 = mfma(g1.a0(), pf[1], dv0); dv1 = mfma(g1.a1(), pf[1], dv1);
          tr_wait<0>(q1);
          dk0 = mfma(q1.a0(), sf[1], dk0); dk1 = mfma(q1.a1(), sf[1], dk1);
          } else { dv0[0] += bf2f(pf[0][0]) + bf2f(pf[1][0]); dk0[0] += bf2f(sf[0][0]) + bf2f(sf[1][0]); }
        }
      }
    };
    // ---- the other half of the interval: look-ahead DMA, part C of step g-1, and at an item boundary the epilogue
    auto part_c = [&]() {
      if (haveA && g >= 1 && !(ABL & 16)) {
        // the slot of step g-1 is free (every wave passed the barrier behind its part A): block g+3 goes there; the stage is free once
        // every wave has read it, i.e. behind the barrier of the item's first step
        if (cj == 1) { issue_kv(kvn); adv(kvn); }
        issue_ring(prod, pj, (g - 1) & (RING - 1));
        if (++pj == nb) { pj = 0; adv(prod); }
      }
      if (g == 0) return;
      {
        const uint32_t xr = xrb + ((g - 1) & 1) * XT, xr2 = xr ^ 16u;      // rows + 16: the swizzle key differs in bit 1
        s16x4 lo[8], hi[8];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // (vbits still belongs to the item of step g-1 here: a new item's part A runs behind this part C)
        if (!(ABL & 1) && (!p.dead_queries || ((vbits >> pcj) & 1u))) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { lo[k] = tr_read<0>(xr + k * 2 * XT); hi[k] = tr_read<16 * XP>(xr2 + k * 2 * XT); }
#define IA_DQ_STEP(k, n)                                                                                   \
        wait2<n>(lo[k], hi[k]);                                                                            \
        if ((vbits >> k) & 1u) acc = mfma16(KT[k], join(lo[k], hi[k]), acc);
        IA_DQ_STEP(0, 14) IA_DQ_STEP(1, 12) IA_DQ_STEP(2, 10) IA_DQ_STEP(3, 8) IA_DQ_STEP(4, 6) IA_DQ_STEP(5, 4) IA_DQ_STEP(6, 2) IA_DQ_STEP(7, 0)
#undef IA_DQ_STEP
        }
        // lane (query n = p16 of the half, rows d = 16 dqt + 4 g4 .. +3): 8 bytes of row q
        const int q = pcj * 32 + 16 * qh + p16;
        bf16x4 o4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o4[i] = f2bf(-p.scale * acc[i]); }
        if (q < L) {
          *reinterpret_cast<bf16x4*>(p.dq + ((size_t)prev.b * L + q) * p.ld_dq + prev.h * 64 + 16 * dqt + 4 * g4) = o4;
#pragma unroll
          for (int i = 0; i < 4; ++i) csq[i] += bf2f(o4[i]);
        }
      }
      if (!first) return;
      // ---- item end (of the item of step g-1): dK, dV through the wave's exchange tiles (free once every wave is past part C),
      // column sums
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const size_t row0 = (size_t)prev.b * L;
      float* const cs_w = reinterpret_cast<float*>(smem + CS_OFF) + wave * 128;
      if (wave < nb) {
        // dK = scale sum dS q = -(1 / log2 e) sum (-dS) q'   (q' = q scale log2 e)
        store_block_rows(smem + X_OFF + wave * EPI_SLOT, dk0, dk1, -1.f / LOG2E, false, p.dk + (row0 + wave * 32) * p.ld_dkv + prev.h * 64, p.ld_dkv,
                         L - wave * 32, lane, p.cs_part ? cs_w : nullptr);
        store_block_rows(smem + X_OFF + wave * EPI_SLOT, dv0, dv1, DROPOUT ? inv_keep : 1.f, false, p.dv + (row0 + wave * 32) * p.ld_dkv + prev.h * 64,
                         p.ld_dkv, L - wave * 32, lane, p.cs_part ? cs_w + 64 : nullptr);
      } else if (p.cs_part) { zero_cs_row(cs_w, lane); zero_cs_row(cs_w + 64, lane); }
      if (p.cs_part) {                                    // workgroup-uniform
        // dq: the tile's column sums over its 16 queries (lanes with equal g4), then the two query halves (waves dqt, dqt + 4)
        // (lane-derived addresses of this once-per-item block are rebuilt from an opaque copy of the lane id: hoisted out of the item
        // loop they were spilled, and a scratch reload drains the whole in-order VMEM queue -- the look-ahead DMA -- with vmcnt(0))
        int ln = lane;
        asm volatile("" : "+v"(ln));
        float* const cq = reinterpret_cast<float*>(smem + CS_OFF + NW * 128 * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) csq[i] = row16_sum(csq[i]);
        if ((ln & 15) == 0) *reinterpret_cast<f32x4*>(cq + wave * 16 + 4 * (ln >> 4)) = csq;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* const dst = p.cs_part + (size_t)(prev.b * ntile) * (3 * nh * 64) + prev.h * 64 + ln;
        if (wave == 0) {
          dst[0] = cq[(ln >> 4) * 16 + (ln & 15)] + cq[((ln >> 4) + 4) * 16 + (ln & 15)];
        } else if (wave == 1 || wave == 2) {
          const float* c = reinterpret_cast<const float*>(smem + CS_OFF) + (wave - 1) * 64 + ln;
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) t += c[w * 128];
          dst[wave * nh * 64] = t;
        } else if (wave == 3 && ntile == 2) {             // the pair kernels write one row per 128-row tile: keep the workspace shape
          float* const z = p.cs_part + (size_t)(prev.b * ntile + 1) * (3 * nh * 64) + prev.h * 64 + ln;
          z[0] = 0.f; z[nh * 64] = 0.f; z[2 * nh * 64] = 0.f;
        }
      }
    };
    part_c();
    if (!haveA) break;
    if (!IA_FUSED_SWAP || grp == 0 || first) part_a(g, first, cj);
    // ---- the next block: wait for this wave's pieces of it, prepare it, publish everything with the step's barrier
    {
      // pieces issued after the ones waited for: the ring pieces of this and the previous interval (4), plus the 9 stage pieces when
      // they went out in one of them (block 1 / 2 of the item); at an item's last step the stage pieces must have landed as well
      int n = 4;
      if (last) n = nb == 2 ? 2 : 4;
      else if (cj == 1 || cj == 2) n = 13;
      wait_vm(n);
      int nj = cj + 1;
      Cur nc = cons;
      if (nj == nb) { nj = 0; adv(nc); }
      if (!(ABL & 4)) prep(nc.it, nj, (g + 1) & (RING - 1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(ABL & 32)) __builtin_amdgcn_s_barrier();
    if (IA_FUSED_SWAP && grp != 0 && !last) part_a(g + 1, false, cj + 1);      // group 1 runs one part A ahead inside an item
    prev = cons; pcj = cj;
    if (++cj == nb) { cj = 0; adv(cons); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the look-ahead pieces still in flight target this workgroup's LDS
}

// bit i of word [b][k] = key 32 k + i of sequence b may be attended (in range, mask byte != 0); 8 words per sequence (L <= 256)
__global__ void attn_key_bits_kernel(const uint8_t* mask, uint32_t* bits, int L) {
  const int b = blockIdx.x, lane = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int key = k * 64 + lane;
    const bool ok = key < L && (mask == nullptr || mask[(size_t)b * L + key] != 0);
    const uint64_t bal = __ballot(ok);
    if (lane == 0) { bits[b * 8 + 2 * k] = (uint32_t)bal; bits[b * 8 + 2 * k + 1] = (uint32_t)(bal >> 32); }
  }
}

// packed_rows > 0: packed self-attention over that many token rows in total (AttnArgs::cu), Lq == Lk == longest sequence
int fill_args(AttnArgs& a, int B, int nh, int Lq, int Lk, int ld_q, int ld_kv, int ld_o, float scale, float drop_p, uint32_t seed,
              long packed_rows = 0) {
  if (B <= 0 || nh <= 0 || Lq <= 0 || Lk <= 0 || Lk > 64 * MAX_KT || Lq > (1 << 20) || (ld_q & 7) || (ld_kv & 7) || (ld_o & 7))
    return IA_ERR_ARG;
  if (ld_q < nh * 64 || ld_kv < nh * 64 || ld_o < nh * 64) return IA_ERR_ARG;
  const uint64_t rq = packed_rows > 0 ? (uint64_t)packed_rows : (uint64_t)B * Lq, rk = packed_rows > 0 ? (uint64_t)packed_rows : (uint64_t)B * Lk;
  const uint64_t qb = rq * ld_q * 2, kb = rk * ld_kv * 2, ob = rq * ld_o * 2;
  if (qb >= 0x7FFFFFFFull || kb >= 0x7FFFFFFFull || ob >= 0x7FFFFFFFull) return IA_ERR_ARG;
  a.B = B; a.nh = nh; a.Lq = Lq; a.Lk = Lk; a.ld_q = ld_q; a.ld_kv = ld_kv; a.ld_o = ld_o; a.ld_dq = ld_q; a.ld_dkv = ld_kv;
  // each rsrc is based at the operand pointer itself, which may start some columns into a packed row, so the
  // window covers "to the end of the last row" from that pointer at most
  a.q_bytes = (uint32_t)(qb - (uint64_t)(ld_q - nh * 64) * 2);
  a.kv_bytes = (uint32_t)(kb - (uint64_t)(ld_kv - nh * 64) * 2);
  a.o_bytes = (uint32_t)ob;
  a.scale = scale; a.sc = scale * LOG2E;
  a.thr16 = drop_p > 0.f ? (uint32_t)(drop_p * 65536.f + 0.5f) : 0u;
  a.inv_keep = drop_p > 0.f ? 1.f / (1.f - (float)a.thr16 / 65536.f) : 1.f;
  a.seed = seed;
  return IA_OK;
}

// development switches (round 3): IA_ATTN_FWD=2 / IA_ATTN_BWD=0 run the round-2 kernels for A/B measurements on one box
int bwd_version() {
  // bit 0: round-3 dQ kernel, bit 1: round-3 dK/dV kernel, bit 2: the fused one-kernel backward where it applies (L <= 256)
  static const int v = [] { const char* e = getenv("IA_ATTN_BWD"); return e ? atoi(e) : 7; }();
  return v;
}
template <bool D> void launch_dkv(const AttnArgs& a, dim3 grid, hipStream_t st) {
  if ((bwd_version() & 2) || a.q_prescaled) hipLaunchKernelGGL(attn_bwd3_dkv_kernel<D>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(attn_bwd_dkv_kernel<D>, grid, dim3(256), 0, st, a);
}
template <bool D> void launch_dq(const AttnArgs& a, dim3 grid, hipStream_t st) {
  if ((bwd_version() & 1) || a.q_prescaled) hipLaunchKernelGGL(attn_bwd3_dq_kernel<D>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(attn_bwd_dq_kernel<D>, grid, dim3(256), 0, st, a);
}
// read on every call (a test / a fine-tuning script may switch it inside one process)
bool exact_delta_on() { const char* e = getenv("IA_ATTN_EXACT_DELTA"); return e && atoi(e) != 0; }
// The fused backward serves plain self-attention (one length for queries and keys, padded rows) of 33 .. 256 tokens
bool fused_applies(const AttnArgs& a) {
  return (bwd_version() & 4) && !a.exact_delta && a.cu == nullptr && a.Lq == a.Lk && a.Lq > 32 && a.Lq <= 256 && a.delta != nullptr;
}
// dQ then dK/dV (the pair of kernels behind every backward entry point that is not served by the fused kernel)
void launch_pair(AttnArgs& a, dim3 gq, dim3 gk, hipStream_t st) {
  if (a.exact_delta) {
    if (a.thr16) hipLaunchKernelGGL(attn_bwd3_delta_kernel<true>, gq, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_bwd3_delta_kernel<false>, gq, dim3(256), 0, st, a);
    // (the round-3 kernels only: the round-2 dQ kernel knows nothing of the flag)
    if (a.thr16) { hipLaunchKernelGGL(attn_bwd3_dq_kernel<true>, gq, dim3(256), 0, st, a); hipLaunchKernelGGL(attn_bwd3_dkv_kernel<true>, gk, dim3(256), 0, st, a); }
    else { hipLaunchKernelGGL(attn_bwd3_dq_kernel<false>, gq, dim3(256), 0, st, a); hipLaunchKernelGGL(attn_bwd3_dkv_kernel<false>, gk, dim3(256), 0, st, a); }
    return;
  }
  if (a.thr16) { launch_dq<true>(a, gq, st); launch_dkv<true>(a, gk, st); }
  else { launch_dq<false>(a, gq, st); launch_dkv<false>(a, gk, st); }
}
void launch_fused(const AttnArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(attn_key_bits_kernel, dim3(a.B), dim3(64), 0, st, a.mask, reinterpret_cast<uint32_t*>(a.delta), a.Lk);
  const int per_x = (a.B * a.nh + 7) / 8;
  const dim3 grid(8 * (per_x < 32 ? per_x : 32));        // one 512-thread workgroup per CU, a multiple of the 8 XCDs
  if (a.thr16) hipLaunchKernelGGL(attn_bwd_fused_kernel<true>, grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL(attn_bwd_fused_kernel<false>, grid, dim3(512), 0, st, a);
}
int fwd_version() {
  // 2: round-2 kernel, 3: 128 queries per workgroup, 4: 256, default 0: by shape
  static const int v = [] { const char* e = getenv("IA_ATTN_FWD"); return e ? atoi(e) : 0; }();
  return v;
}
// grid: the workgroup count for 128 queries per workgroup (the round-2 geometry); the 256-query kernel derives its own
void launch_fwd(const AttnArgs& a, dim3 grid, hipStream_t stream) {
  const dim3 blk(256);
  const int v = fwd_version();
  if (v == 2 && !a.q_prescaled) {                        // (the round-2 kernels multiply every score by sc themselves)
    if (a.thr16) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, blk, 0, stream, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, blk, 0, stream, a);
  } else if (v == 3 || (v != 4 && ((a.Lq - 1) & 255) < 128)) {
    // 128 queries per workgroup when the last 256-query block would be less than half full (ViT: 577 = 2 x 256 + 65); measured
    // (profiles/r03_attention_variants.txt): 256-query workgroups are 3-6 % faster at L = 220 / 255, equal at 510, 3 % slower at 577
    if (a.thr16) hipLaunchKernelGGL((attn_fwd3_kernel<true, 1>), grid, blk, 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd3_kernel<false, 1>), grid, blk, 0, stream, a);
  } else {
    const dim3 g2(((a.Lq + 255) / 256) * a.nh * a.B);
    if (a.thr16) hipLaunchKernelGGL((attn_fwd3_kernel<true, 2>), g2, blk, 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd3_kernel<false, 2>), g2, blk, 0, stream, a);
  }
}

}  // namespace

// General form: Lq queries attend to Lk keys per (sequence, head).  q / out / d_out rows are b*Lq + i (strides ld_q,
// ld_o), k / v rows are b*Lk + j (stride ld_kv); head h sits at column h*64 of each.  Multi-query attention (one K/V
// head shared by all query heads, reference multimodal.py:590-616) is the nh = 1 case with the query heads folded
// into rows: q viewed as [B, n*heads, 64] (ld_q = 64), Lq = n*heads.  key_mask is [B, Lk].
static int attn_fwd_impl(int q_prescaled, const void* q, int ld_q, const void* k, const void* v, int ld_kv, const uint8_t* key_mask, void* out,
                             int ld_o, float* lse2, int B, int nh, int Lq, int Lk, float scale, float drop_p, uint32_t seed,
                             hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!q || !k || !v || !out) return IA_ERR_ARG;
  AttnArgs a{};
  int rc = fill_args(a, B, nh, Lq, Lk, ld_q, ld_kv, ld_o, scale, drop_p, seed);
  if (rc) return rc;
  if (q_prescaled && !fwd3::PRESCALE) return IA_ERR_ARG;
  a.q_prescaled = q_prescaled;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out; a.mask = key_mask; a.lse2 = lse2;
  dim3 grid(((Lq + 127) / 128) * nh * B), blk(256);
  launch_fwd(a, grid, stream);
  return ia_check_launch();
}

extern "C" int ia_attn_fwd_x(const void* q, int ld_q, const void* k, const void* v, int ld_kv, const uint8_t* key_mask, void* out,
                             int ld_o, float* lse2, int B, int nh, int Lq, int Lk, float scale, float drop_p, uint32_t seed,
                             hipStream_t stream) {
  return attn_fwd_impl(0, q, ld_q, k, v, ld_kv, key_mask, out, ld_o, lse2, B, nh, Lq, Lk, scale, drop_p, seed, stream);
}

// delta: caller-provided scratch of B*nh*Lq floats (filled by the dQ kernel, read by the dK/dV kernel).
extern "C" int ia_attn_bwd_x(const void* q, int ld_q, const void* k, const void* v, int ld_kv, const uint8_t* key_mask,
                             const void* out, const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, int ld_dq,
                             void* dk, void* dv, int ld_dkv, int B, int nh, int Lq, int Lk, float scale, float drop_p, uint32_t seed,
                             hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!q || !k || !v || !out || !d_out || !lse2 || !delta || !dq || !dk || !dv) return IA_ERR_ARG;
  AttnArgs a{};
  int rc = fill_args(a, B, nh, Lq, Lk, ld_q, ld_kv, ld_o, scale, drop_p, seed);
  if (rc) return rc;
  if ((ld_dq & 7) || (ld_dkv & 7) || ld_dq < nh * 64 || ld_dkv < nh * 64) return IA_ERR_ARG;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (const bf16*)out; a.d_o = (const bf16*)d_out;
  a.mask = key_mask; a.lse2 = const_cast<float*>(lse2); a.delta = delta;
  a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.ld_dq = ld_dq; a.ld_dkv = ld_dkv;
  dim3 gq(((Lq + 127) / 128) * nh * B), gk(((Lk + 127) / 128) * nh * B), blk(256);
  a.exact_delta = exact_delta_on() ? 1 : 0;
  if (fused_applies(a)) launch_fused(a, stream);
  else launch_pair(a, gq, gk, stream);
  return ia_check_launch();
}

// Self-attention over a packed projection: q, k, v point at the first column of head 0 of each operand and share
// row stride ld_qkv (packed [tokens, 3H] projection output, or three separate [tokens, H] tensors with ld_qkv = H).
extern "C" int ia_attn_fwd(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, void* out,
                           int ld_o, float* lse2, int B, int nh, int L, float scale, float drop_p, uint32_t seed,
                           hipStream_t stream) {
  return ia_attn_fwd_x(q, ld_qkv, k, v, ld_qkv, key_mask, out, ld_o, lse2, B, nh, L, L, scale, drop_p, seed, stream);
}

// ia_attn_fwd on a projection whose q columns already hold q * scale * log2(e) rounded to bf16 (ia_gemm_bf16_qscale): no kernel of the
// forward / backward pair scales q again, so all of them exponentiate bit-identical products
extern "C" int ia_attn_fwd_ps(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, void* out, int ld_o,
                              float* lse2, int B, int nh, int L, float scale, float drop_p, uint32_t seed, hipStream_t stream) {
  return attn_fwd_impl(1, q, ld_qkv, k, v, ld_qkv, key_mask, out, ld_o, lse2, B, nh, L, L, scale, drop_p, seed, stream);
}

extern "C" int ia_attn_bwd(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out,
                           const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv,
                           int ld_dqkv, int B, int nh, int L, float scale, float drop_p, uint32_t seed, hipStream_t stream) {
  return ia_attn_bwd_x(q, ld_qkv, k, v, ld_qkv, key_mask, out, d_out, ld_o, lse2, delta, dq, ld_dqkv, dk, dv, ld_dqkv, B, nh, L, L, scale,
                       drop_p, seed, stream);
}

// ia_attn_bwd that also returns the bias gradient of the fused QKV projection: dbias[3*nh*64] (q | k | v order) += column sums of
// dq, dk, dv over all tokens, taken from the rows as they are stored (each workgroup's sums go to a row of the workspace matrix, one
// fixed-order fold afterwards: deterministic) -- the separate column-sum pass over [tokens, 3H] disappears from the layer backward.
extern "C" size_t ia_attn_bwd_bias_workspace_bytes(int B, int nh, int L) {
  if (B <= 0 || nh <= 0 || L <= 0) return 0;
  return (size_t)B * ((L + 127) / 128) * 3 * nh * 64 * sizeof(float);
}

static int attn_bwd_bias_impl(int flags, const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out,
                              const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv,
                              float* dbias, void* workspace, size_t workspace_bytes, int B, int nh, int L, float scale, float drop_p,
                              uint32_t seed, hipStream_t stream) {
  (void)hipGetLastError();
  if (!q || !k || !v || !out || !d_out || !lse2 || !delta || !dq || !dk || !dv || !dbias) return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_attn_bwd_bias_workspace_bytes(B, nh, L)) return IA_ERR_WORKSPACE;
  AttnArgs a{};
  int rc = fill_args(a, B, nh, L, L, ld_qkv, ld_qkv, ld_o, scale, drop_p, seed);
  if (rc) return rc;
  if ((ld_dqkv & 7) || ld_dqkv < nh * 64) return IA_ERR_ARG;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (const bf16*)out; a.d_o = (const bf16*)d_out;
  a.mask = key_mask; a.lse2 = const_cast<float*>(lse2); a.delta = delta;
  a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.ld_dq = ld_dqkv; a.ld_dkv = ld_dqkv;
  a.cs_part = (float*)workspace;
  const int q_prescaled = flags & IA_ATTN_Q_PRESCALED;
  if (q_prescaled && !fwd3::PRESCALE) return IA_ERR_ARG;
  a.q_prescaled = q_prescaled ? 1 : 0;
  a.dead_queries = ((flags & IA_ATTN_MASKED_ROWS_DEAD) && key_mask) ? 1 : 0;
  dim3 grid(((L + 127) / 128) * nh * B), blk(256);
  a.exact_delta = exact_delta_on() ? 1 : 0;
  if (fused_applies(a)) launch_fused(a, stream);
  else launch_pair(a, grid, grid, stream);
  rc = ia_check_launch();
  if (rc) return rc;
  return ia_sum_rows_f32((const float*)workspace, B * ((L + 127) / 128), 3 * nh * 64, dbias, 1, stream);
}

extern "C" int ia_attn_bwd_bias(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out,
                                const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv,
                                float* dbias, void* workspace, size_t workspace_bytes, int B, int nh, int L, float scale, float drop_p,
                                uint32_t seed, hipStream_t stream) {
  return attn_bwd_bias_impl(0, q, k, v, ld_qkv, key_mask, out, d_out, ld_o, lse2, delta, dq, dk, dv, ld_dqkv, dbias, workspace, workspace_bytes,
                            B, nh, L, scale, drop_p, seed, stream);
}
// backward of ia_attn_fwd_ps: q as the forward saw it (pre-scaled); dq is still dL/d(q before the scale), i.e. what the QKV projection's
// weight / input gradients consume unchanged
extern "C" int ia_attn_bwd_bias_ps(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out,
                                   const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv,
                                   float* dbias, void* workspace, size_t workspace_bytes, int B, int nh, int L, float scale, float drop_p,
                                   uint32_t seed, hipStream_t stream) {
  return attn_bwd_bias_impl(IA_ATTN_Q_PRESCALED, q, k, v, ld_qkv, key_mask, out, d_out, ld_o, lse2, delta, dq, dk, dv, ld_dqkv, dbias, workspace,
                            workspace_bytes, B, nh, L, scale, drop_p, seed, stream);
}
// ia_attn_bwd_bias with flags (IA_ATTN_*): IA_ATTN_Q_PRESCALED = the _ps form; IA_ATTN_MASKED_ROWS_DEAD = the caller guarantees that d_out is
// zero at every masked position (an encoder whose masked positions never reach the loss: they are masked as keys in every layer and no
// head reads them) -- the one-kernel backward then skips the 32-query blocks that hold only masked positions; results are identical.
extern "C" int ia_attn_bwd_bias_ex(int flags, const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out,
                                   const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv,
                                   float* dbias, void* workspace, size_t workspace_bytes, int B, int nh, int L, float scale, float drop_p,
                                   uint32_t seed, hipStream_t stream) {
  if (flags & ~(IA_ATTN_Q_PRESCALED | IA_ATTN_MASKED_ROWS_DEAD)) return IA_ERR_ARG;
  return attn_bwd_bias_impl(flags, q, k, v, ld_qkv, key_mask, out, d_out, ld_o, lse2, delta, dq, dk, dv, ld_dqkv, dbias, workspace, workspace_bytes,
                            B, nh, L, scale, drop_p, seed, stream);
}

// Packed ("unpadded") self-attention: the token rows of all sequences lie back to back, sequence b owning rows
// cu_seqlens[b] .. cu_seqlens[b+1] (int32 [B+1], device; total_tokens = cu_seqlens[B]); no key mask — every key of a sequence is
// attendable.  Lmax = longest sequence (sets the grid and the row stride of lse2 / delta, which stay [B, nh, Lmax]).  Same
// arithmetic as ia_attn_fwd / ia_attn_bwd on the valid tokens of a right-padded batch; padded positions are simply absent.
static int attn_fwd_varlen_impl(int q_prescaled, const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens,
                                void* out, int ld_o, float* lse2, int B, int nh, int Lmax, float scale, float drop_p, uint32_t seed,
                                hipStream_t stream) {
  (void)hipGetLastError();
  if (!q || !k || !v || !out || !cu_seqlens || total_tokens <= 0) return IA_ERR_ARG;
  AttnArgs a{};
  int rc = fill_args(a, B, nh, Lmax, Lmax, ld_qkv, ld_qkv, ld_o, scale, drop_p, seed, total_tokens);
  if (rc) return rc;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out; a.mask = nullptr; a.lse2 = lse2; a.cu = cu_seqlens;
  if (q_prescaled && !fwd3::PRESCALE) return IA_ERR_ARG;
  a.q_prescaled = q_prescaled;
  dim3 grid(((Lmax + 127) / 128) * nh * B);
  launch_fwd(a, grid, stream);
  return ia_check_launch();
}
extern "C" int ia_attn_fwd_varlen(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens, void* out,
                                  int ld_o, float* lse2, int B, int nh, int Lmax, float scale, float drop_p, uint32_t seed, hipStream_t stream) {
  return attn_fwd_varlen_impl(0, q, k, v, ld_qkv, cu_seqlens, total_tokens, out, ld_o, lse2, B, nh, Lmax, scale, drop_p, seed, stream);
}
extern "C" int ia_attn_fwd_varlen_ps(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens, void* out,
                                     int ld_o, float* lse2, int B, int nh, int Lmax, float scale, float drop_p, uint32_t seed,
                                     hipStream_t stream) {
  return attn_fwd_varlen_impl(1, q, k, v, ld_qkv, cu_seqlens, total_tokens, out, ld_o, lse2, B, nh, Lmax, scale, drop_p, seed, stream);
}

static int attn_bwd_varlen_impl(int q_prescaled, const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens,
                                const void* out, const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv,
                                int ld_dqkv, int B, int nh, int Lmax, float scale, float drop_p, uint32_t seed, hipStream_t stream) {
  (void)hipGetLastError();
  if (!q || !k || !v || !out || !d_out || !lse2 || !delta || !dq || !dk || !dv || !cu_seqlens || total_tokens <= 0) return IA_ERR_ARG;
  AttnArgs a{};
  int rc = fill_args(a, B, nh, Lmax, Lmax, ld_qkv, ld_qkv, ld_o, scale, drop_p, seed, total_tokens);
  if (rc) return rc;
  if ((ld_dqkv & 7) || ld_dqkv < nh * 64) return IA_ERR_ARG;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (const bf16*)out; a.d_o = (const bf16*)d_out;
  a.mask = nullptr; a.lse2 = const_cast<float*>(lse2); a.delta = delta; a.cu = cu_seqlens;
  a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.ld_dq = ld_dqkv; a.ld_dkv = ld_dqkv;
  if (q_prescaled && !fwd3::PRESCALE) return IA_ERR_ARG;
  a.q_prescaled = q_prescaled;
  dim3 grid(((Lmax + 127) / 128) * nh * B), blk(256);
  a.exact_delta = exact_delta_on() ? 1 : 0;
  launch_pair(a, grid, grid, stream);
  return ia_check_launch();
}
extern "C" int ia_attn_bwd_varlen(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens,
                                  const void* out, const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv,
                                  int ld_dqkv, int B, int nh, int Lmax, float scale, float drop_p, uint32_t seed, hipStream_t stream) {
  return attn_bwd_varlen_impl(0, q, k, v, ld_qkv, cu_seqlens, total_tokens, out, d_out, ld_o, lse2, delta, dq, dk, dv, ld_dqkv, B, nh, Lmax, scale,
                              drop_p, seed, stream);
}
extern "C" int ia_attn_bwd_varlen_ps(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens,
                                     const void* out, const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv,
                                     int ld_dqkv, int B, int nh, int Lmax, float scale, float drop_p, uint32_t seed, hipStream_t stream) {
  return attn_bwd_varlen_impl(1, q, k, v, ld_qkv, cu_seqlens, total_tokens, out, d_out, ld_o, lse2, delta, dq, dk, dv, ld_dqkv, B, nh, Lmax, scale,
                              drop_p, seed, stream);
}
