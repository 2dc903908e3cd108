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
  uint32_t thr16; float inv_keep; uint32_t seed;
  float* cs_part;                                 // backward, optional: [b*ntile + tile][3*nh*64] fp32 column sums of this workgroup's
                                                  // dq | dk | dv rows (the QKV bias gradient, summed over rows by ia_sum_rows_f32); null = off
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
// Same arithmetic and tile geometry (4 waves x 32 queries, 64-key tiles, S^T orientation) as attn_fwd_kernel above, restructured after
// tools/abl/issue_model.hip had priced the instruction streams on an MI355X (profiles/r03_issue_model.txt):
//  * the softmax is down to its 80 unavoidable VALU instructions per tile (32 exp2, 32 adds, 16 packed converts): the query fragments
//    are pre-scaled by scale*log2(e) once per block, and -m_ref comes out of the matrix pipe as the C operand of the first QK^T MFMA
//    (a 16-register block holding -m_ref), so an accumulator IS exp2's argument; no running maximum is taken at all -- m_ref only
//    moves when a row's sum leaves [2^-60, 2^60] (rebase(): scores recomputed from the K tile still in the ring, true maximum taken);
//    any reference cancels in O / l, and exp2 arguments stay far inside the fp32 exponent range;
//  * masked / out-of-range keys get -1e30 from one extra MFMA per 32-key block (A = the penalty in k-slot 0 of the key's row, B = ones
//    in k-slot 0) instead of 64 compare+select pairs (v_cmp 8.6 cycles, v_cndmask behind it 4.8: 13 cycles per select on one wave);
//  * software pipelined inside the wave: QK^T of tile t+1 is issued under the softmax of tile t (two score blocks live), the V^T
//    fragments of tile t and the K fragments of tile t+2 are in registers before the MFMAs that use them (asm reads, counted waits);
//  * K ring of 4, V ring of 3 LDS slots; the DMA of K(t+3) / V(t+2) is issued at the top of iteration t and only K(t+2) is waited for
//    there (counted vmcnt), one raw s_barrier per tile.
#ifndef IA_F3_ABL
#define IA_F3_ABL 0          // ablation bits for tools/abl/attn_dev.hip builds: 1 no K/V DMA in the loop, 2 no barrier, 4 no compute, 8 no exp, 16 no LDS reads
#endif
#ifndef IA_F3_PRESCALE
#define IA_F3_PRESCALE 1     // 1: Q fragments carry scale * log2(e) (one more bf16 rounding of q, no multiply per score); 0: exp2(s * sc)
#endif
namespace fwd3 {
constexpr bool PRESCALE = IA_F3_PRESCALE != 0;
// LDS: K ring | V ring | [PIPE: Q staging, later the epilogue's row staging] | valid-key table.  Without PIPE the K ring is 3 deep as
// well (the fragments of tile t+1 are read during tile t), Q is staged in the last V slot (first filled behind the first barrier) and
// the epilogue rows in the K ring (behind a last barrier): 48 KiB, three workgroups per CU.
template <bool PIPE>
struct Cfg {
  static constexpr int KRING = PIPE ? 4 : 3, VRING = 3;
  // PIPE: K0..K3 | V0..V2 | Q / epilogue rows | table.   otherwise: K0 K1 | V0 V1 | K2 V2 (= Q staging, 16 KiB) | table
  static constexpr int Q_OFF = PIPE ? (KRING + VRING) * 8192 : 32768;
  static constexpr int EPI_OFF = PIPE ? Q_OFF : 0;
  static constexpr int TAB_OFF = PIPE ? Q_OFF + 4 * EPI_SLOT : 49152;
  static constexpr int SMEM = TAB_OFF + MAX_KT * 8;
  static IA_DEV int k_slot(int kt) { const int s = kt % KRING; return PIPE ? s * 8192 : (s == 2 ? 32768 : s * 8192); }
  static IA_DEV int v_slot(int kt) { const int s = kt % VRING; return PIPE ? (KRING + s) * 8192 : (s == 2 ? 40960 : 16384 + s * 8192); }
};
constexpr float L_LO = 8.6736174e-19f, L_HI = 1.1529215e18f;                      // 2^-60, 2^60
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

struct Lane {          // lane constants of the fragment reads
  uint32_t ka[4];      // byte offset of this lane's K fragment (row lq, k-step kb) inside a K tile; the second 32-key block is +4096
  uint32_t v0, v1;     // transpose-read offsets inside a V tile (d 0..31 / 32..63)
};

// all eight K fragments of a tile (kf[kb * 2 + blk])
IA_DEV void read_k(bf16x8 (&kf)[8], const Lane& ln, uint32_t tile_addr) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const uint32_t a = tile_addr + ln.ka[kb];
    kf[kb * 2] = lds_read_b128<0>(a);
    kf[kb * 2 + 1] = lds_read_b128<4096>(a);
  }
}

// Scores of one 64-key tile against this wave's 32 queries: s = K (Q * scale * log2 e)^T [- m_ref] [- 1e30 on keys that may not be
// attended].  The two optional terms come from one more MFMA per 32-key block whose A operand carries (penalty, 1) in k-slots 0 / 1 of
// the key's row and whose B operand carries (1, -m_ref) in k-slots 0 / 1 of the query's column; a tile whose 64 keys are all
// attendable in a wave that never left the reference 0 (the common case) starts from the inline constant 0 instead.
// refw: this lane's B word {bf16 1.0, bf16 -m_ref} (lanes 32..63, which hold k-slots 8..15: 0).
IA_DEV void qk_tile(f32x16& s0, f32x16& s1, const bf16x8 (&kf)[8], const bf16x8 (&qf)[4], bool plain, uint32_t refw, uint32_t valid_lo,
                    uint32_t valid_hi, int lane) {
  const f32x16 zero = zero16();
  if (plain) {                                                                      // wave-uniform
    s0 = mfma(kf[0], qf[0], zero);
    s1 = mfma(kf[1], qf[0], zero);
  } else {
    const uint32_t low = (lane & 32) ? 0u : 1u;
    const uint32_t bad0 = ((~valid_lo) >> (lane & 31)) & low, bad1 = ((~valid_hi) >> (lane & 31)) & low;
    const u32x4 a0 = {bad0 * NEG_BIG_BF16 + low * 0x3F800000u, 0u, 0u, 0u}, a1 = {bad1 * NEG_BIG_BF16 + low * 0x3F800000u, 0u, 0u, 0u};
    const u32x4 bw = {refw, 0u, 0u, 0u};
    const f32x16 c0 = mfma(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bw), zero);
    const f32x16 c1 = mfma(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bw), zero);
    s0 = mfma(kf[0], qf[0], c0);
    s1 = mfma(kf[1], qf[0], c1);
  }
#pragma unroll
  for (int kb = 1; kb < 4; ++kb) {
    s0 = mfma(kf[kb * 2], qf[kb], s0);
    s1 = mfma(kf[kb * 2 + 1], qf[kb], s1);
  }
}

struct Row {           // per-lane softmax state of its query (the partner lane ^ 32 holds the other half of the keys)
  float m_ref, l_run;  // m_ref is a bf16-representable number (it travels through the matrix pipe as a bf16 operand)
  uint32_t refw;       // {bf16 1.0, bf16 -m_ref} in lanes 0..31, 0 in lanes 32..63
};

// p = exp2(s [* sc]) in place; returns this lane's half of the tile's row sum
IA_DEV float exp_sum(f32x16& s0, f32x16& s1, float sc) {
  float ra = 0.f, rb = 0.f;
  if (IA_F3_ABL & 8) return 1.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s0[r] = __builtin_amdgcn_exp2f(PRESCALE ? s0[r] : s0[r] * sc);
    s1[r] = __builtin_amdgcn_exp2f(PRESCALE ? s1[r] : s1[r] * sc);
    ra += s0[r];
    rb += s1[r];
  }
  return ra + rb;
}

// Rare: a row sum left [L_LO, L_HI] (or is not finite).  The tile's scores are recomputed from the K tile that is still in its ring
// slot, the reference moves to the larger of the tile's maximum and the running log-sum-exp, everything accumulated so far follows.
// snx0 / snx1: the next tile's scores, already formed against the old reference (null for the last tile).
// sc: what exp2's argument is multiplied by (1 with pre-scaled queries); m_ref lives in the accumulators' units.
IA_DEV float rebase(f32x16& s0, f32x16& s1, f32x16* snx0, f32x16* snx1, f32x16& o0, f32x16& o1, Row& row, const Lane& ln,
                    uint32_t ktile_addr, const bf16x8 (&qf)[4], uint32_t valid_lo, uint32_t valid_hi, int lane, float sc) {
  if (PRESCALE) sc = 1.f;
  const float inv_sc = 1.f / sc;
  bf16x8 kf[8];
  read_k(kf, ln, ktile_addr);
  frag_wait<0>(kf);
  qk_tile(s0, s1, kf, qf, false, (lane & 32) ? 0u : 0x3F80u, valid_lo, valid_hi, lane);      // reference 0
  float tm = NEG_BIG;
#pragma unroll
  for (int r = 0; r < 16; ++r) tm = fmaxf(tm, fmaxf(s0[r], s1[r]));
  tm = fmaxf(tm, swap32(tm));
  const float l_prev = row.l_run + swap32(row.l_run);
  const bool have_prev = l_prev > 0.f, have_tile = tm > 0.5f * NEG_BIG;
  float m_new = row.m_ref;
  if (have_prev) m_new = row.m_ref + __builtin_amdgcn_logf(l_prev) * inv_sc;      // v_log_f32 = log2
  if (have_tile) m_new = have_prev ? fmaxf(m_new, tm) : tm;
  m_new = bf2f(f2bf(m_new));
  const float shift = row.m_ref - m_new;
  const float alpha = have_prev ? __builtin_amdgcn_exp2f(shift * sc) : 0.f;
  row.m_ref = m_new;
  row.refw = (lane & 32) ? 0u : (0x3F80u | ((uint32_t)__builtin_bit_cast(uint16_t, f2bf(-m_new)) << 16));
  row.l_run *= alpha;
  float rs = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s0[r] = __builtin_amdgcn_exp2f((s0[r] - m_new) * sc);
    s1[r] = __builtin_amdgcn_exp2f((s1[r] - m_new) * sc);
    rs += s0[r] + s1[r];
    o0[r] *= alpha; o1[r] *= alpha;
  }
  if (snx0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { (*snx0)[r] += shift; (*snx1)[r] += shift; }
  }
  return rs;
}
}  // namespace fwd3

// PIPE: the QK^T of tile t+1 runs under the softmax of tile t inside the wave (two score blocks live, 2 waves per SIMD); without it a wave
// alternates an MFMA batch (PV of tile t, QK^T of tile t+1) with the softmax, and the waves sharing its SIMD fill the other pipe.
template <bool DROPOUT, bool PIPE, int WPS>
__global__ __launch_bounds__(256, WPS) void attn_fwd3_kernel(AttnArgs p) {
  using namespace fwd3;
  using C = Cfg<PIPE>;
  constexpr int Q_OFF = C::Q_OFF, TAB_OFF = C::TAB_OFF;
  constexpr int AHEAD = PIPE ? 2 : 1;                     // the K fragments requested during tile t are those of tile t + AHEAD
  __shared__ __attribute__((aligned(16))) char smem[C::SMEM];
  uint32_t (*s_valid)[2] = reinterpret_cast<uint32_t (*)[2]>(smem + TAB_OFF);
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
  const __amdgpu_buffer_rsrc_t rsK = ia_rsrc(p.k, p.kv_bytes);
  const __amdgpu_buffer_rsrc_t rsV = ia_rsrc(p.v, p.kv_bytes);
  const uint32_t sbase = lds_addr(smem);
  const int nkt_all = (L + 63) >> 6;

  auto stage_k = [&](int kt) { stage64<false>(rsK, smem + C::k_slot(kt), rowbase + kt * 64, L - kt * 64, p.ld_kv, h * 64, tid, wave); };
  auto stage_v = [&](int kt) { stage64<true>(rsV, smem + C::v_slot(kt), rowbase + kt * 64, L - kt * 64, p.ld_kv, h * 64, tid, wave); };

  // ---- prologue: one round trip for Q, K0, V0, K1 and the mask bytes; K2 / V1 follow and stay in flight
  char* qslot = smem + Q_OFF + wave * 4096;
  stage_rows32(ia_rsrc(p.q, p.q_bytes), qslot, qbase + q0, Lq - q0, p.ld_q, h * 64, lane);
  stage_k(0); stage_v(0); stage_k(1);
  build_valid_table(p, s_valid, rowbase, L, lane, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (PIPE) stage_k(2);
  stage_v(1);
  bf16x8 qf[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const bf16x8 raw = frag_b128(qslot, lq, kb * 2 + hh);
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[kb][j] = PRESCALE ? f2bf(bf2f(raw[j]) * p.sc) : raw[j];
  }
  Lane ln;
  {
    const uint32_t a0 = (uint32_t)(lq * 128 + ((hh ^ swz_b128(lq)) << 4));
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) ln.ka[kb] = a0 ^ (uint32_t)(kb << 5);
    ln.v0 = tr_lane_off(lane, 0); ln.v1 = tr_lane_off(lane, 32);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // trailing tiles without any attendable key are never touched (right-padded batches); holes inside take the penalty path
  int nkt = nkt_all;
  while (nkt > 1 && (s_valid[nkt - 1][0] | s_valid[nkt - 1][1]) == 0u) --nkt;
  nkt = __builtin_amdgcn_readfirstlane(nkt);

  Row row{0.f, 0.f, hh ? 0u : 0x3F80u};
  bool has_ref = false;                                  // wave-uniform: some row of this wave has left the reference 0
  f32x16 o0 = zero16(), o1 = zero16();
  f32x16 sa0, sa1, sb0, sb1;
  bf16x8 kf[8];
  if (active) {
    read_k(kf, ln, sbase + C::k_slot(0));
    if (PIPE) {
      frag_wait<0>(kf);
      const uint32_t v_lo = __builtin_amdgcn_readfirstlane(s_valid[0][0]), v_hi = __builtin_amdgcn_readfirstlane(s_valid[0][1]);
      qk_tile(sa0, sa1, kf, qf, (v_lo & v_hi) == 0xFFFFFFFFu, row.refw, v_lo, v_hi, lane);
      if (nkt > 1) read_k(kf, ln, sbase + C::k_slot(1));
    }
  }

  // one tile.  PIPE: sc = scores of tile t (in), sn = scores of tile t+1 (out, unless LAST); otherwise sc is scratch, sn unused
  auto tile_step = [&](auto LAST_T, f32x16& sc0, f32x16& sc1, f32x16& sn0, f32x16& sn1, int t) {
    constexpr bool LAST = decltype(LAST_T)::value;
    // K(t+2) has landed for this wave (V(t+1), issued behind it, may still be in flight), then for everybody
    if (LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (!(IA_F3_ABL & 2)) __builtin_amdgcn_s_barrier();
    if (!(IA_F3_ABL & 1)) {
      if (t + AHEAD + 1 < nkt) stage_k(t + AHEAD + 1);
      if (t + 2 < nkt) stage_v(t + 2);
    }
    if (!active || (IA_F3_ABL & 4)) return;
    const uint32_t vt = sbase + (uint32_t)C::v_slot(t);
    const uint32_t vb0 = vt + ln.v0, vb1 = vt + ln.v1;
    const uint32_t cur_lo = __builtin_amdgcn_readfirstlane(s_valid[t][0]), cur_hi = __builtin_amdgcn_readfirstlane(s_valid[t][1]);
    TrPair va, vb, vc, vd;
    if (PIPE) {
      if (!LAST) {
        frag_wait<0>(kf);                                 // the K fragments of tile t+1, requested during the previous tile's PV
        const uint32_t n_lo = __builtin_amdgcn_readfirstlane(s_valid[t + 1][0]), n_hi = __builtin_amdgcn_readfirstlane(s_valid[t + 1][1]);
        qk_tile(sn0, sn1, kf, qf, !has_ref && (n_lo & n_hi) == 0xFFFFFFFFu, row.refw, n_lo, n_hi, lane);
      }
    } else {
      frag_wait<0>(kf);                                   // the K fragments of tile t
      qk_tile(sc0, sc1, kf, qf, !has_ref && (cur_lo & cur_hi) == 0xFFFFFFFFu, row.refw, cur_lo, cur_hi, lane);
    }
    float rs = exp_sum(sc0, sc1, p.sc);
    {
      float tot = row.l_run + rs;
      tot += swap32(tot);
      if (__builtin_expect(__ballot(!(tot >= L_LO && tot <= L_HI)) != 0ull, 0)) {
        rs = rebase(sc0, sc1, (PIPE && !LAST) ? &sn0 : nullptr, (PIPE && !LAST) ? &sn1 : nullptr, o0, o1, row, ln,
                    sbase + (uint32_t)C::k_slot(t), qf, cur_lo, cur_hi, lane, p.sc);
        has_ref = true;
      }
    }
    row.l_run += rs;
    if (IA_F3_ABL & 16) {                                 // no LDS reads: whatever is in the K fragment registers stands in for V^T
      va.lo0 = va.lo1 = vb.lo0 = vb.lo1 = vc.lo0 = vc.lo1 = vd.lo0 = vd.lo1 = __builtin_bit_cast(s16x8, kf[0]).lo;
      va.hi0 = va.hi1 = vb.hi0 = vb.hi1 = vc.hi0 = vc.hi1 = vd.hi0 = vd.hi1 = __builtin_bit_cast(s16x8, kf[1]).hi;
    } else {
    tr_issue<0>(va, vb0, vb1);                            // the V^T fragments land under the conversions
    tr_issue<16>(vb, vb0, vb1);
    }
    bf16x8 pf[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pf[0][j] = f2bf(sc0[j]); pf[1][j] = f2bf(sc0[8 + j]);
      pf[2][j] = f2bf(sc1[j]); pf[3][j] = f2bf(sc1[8 + j]);
    }
    tr_wait<4>(va);
    o0 = mfma(va.a0(), pf[0], o0); o1 = mfma(va.a1(), pf[0], o1);
    if (!(IA_F3_ABL & 16)) tr_issue<32>(vc, vb0, vb1);
    tr_wait<4>(vb);
    o0 = mfma(vb.a0(), pf[1], o0); o1 = mfma(vb.a1(), pf[1], o1);
    if (!(IA_F3_ABL & 16)) tr_issue<48>(vd, vb0, vb1);
    const bool more = !LAST && t + AHEAD < nkt && !(IA_F3_ABL & 16);
    if (more) read_k(kf, ln, sbase + (uint32_t)C::k_slot(t + AHEAD));
    if (more) { tr_wait<12>(vc); } else { tr_wait<4>(vc); }
    o0 = mfma(vc.a0(), pf[2], o0); o1 = mfma(vc.a1(), pf[2], o1);
    if (more) { tr_wait<8>(vd); } else { tr_wait<0>(vd); }
    o0 = mfma(vd.a0(), pf[3], o0); o1 = mfma(vd.a1(), pf[3], o1);
  };
  if (PIPE) {
    int t = 0;
    for (;;) {
      if (t + 1 >= nkt) { tile_step(std::true_type{}, sa0, sa1, sb0, sb1, t); break; }
      tile_step(std::false_type{}, sa0, sa1, sb0, sb1, t); ++t;
      if (t + 1 >= nkt) { tile_step(std::true_type{}, sb0, sb1, sa0, sa1, t); break; }
      tile_step(std::false_type{}, sb0, sb1, sa0, sa1, t); ++t;
    }
  } else {
    for (int t = 0; t + 1 < nkt; ++t) tile_step(std::false_type{}, sa0, sa1, sb0, sb1, t);
    tile_step(std::true_type{}, sa0, sa1, sb0, sb1, nkt - 1);
    __builtin_amdgcn_s_barrier();                         // the epilogue rows are staged in the K ring
  }
  if (!active) return;
  const float l_tot = row.l_run + swap32(row.l_run);
  const float inv = l_tot > 0.f ? p.inv_keep / l_tot : 0.f;
  if (q < Lq && hh == 0 && p.lse2) p.lse2[((size_t)b * p.nh + h) * p.Lq + q] = row.m_ref * (PRESCALE ? 1.f : p.sc) + __builtin_amdgcn_logf(l_tot);
  store_block_rows(smem + C::EPI_OFF + wave * EPI_SLOT, o0, o1, inv, false, p.out + (qbase + q0) * p.ld_o + h * 64, p.ld_o, Lq - q0, lane);
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

// development switch (round 3): IA_ATTN_FWD=2 runs the round-2 forward kernel for A/B measurements on one box
int fwd_version() {
  static const int v = [] { const char* e = getenv("IA_ATTN_FWD"); return e ? atoi(e) : 2; }();
  return v;
}

}  // namespace

// General form: Lq queries attend to Lk keys per (sequence, head).  q / out / d_out rows are b*Lq + i (strides ld_q,
// ld_o), k / v rows are b*Lk + j (stride ld_kv); head h sits at column h*64 of each.  Multi-query attention (one K/V
// head shared by all query heads, reference multimodal.py:590-616) is the nh = 1 case with the query heads folded
// into rows: q viewed as [B, n*heads, 64] (ld_q = 64), Lq = n*heads.  key_mask is [B, Lk].
extern "C" int ia_attn_fwd_x(const void* q, int ld_q, const void* k, const void* v, int ld_kv, const uint8_t* key_mask, void* out,
                             int ld_o, float* lse2, int B, int nh, int Lq, int Lk, float scale, float drop_p, uint32_t seed,
                             hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!q || !k || !v || !out) return IA_ERR_ARG;
  AttnArgs a{};
  int rc = fill_args(a, B, nh, Lq, Lk, ld_q, ld_kv, ld_o, scale, drop_p, seed);
  if (rc) return rc;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out; a.mask = key_mask; a.lse2 = lse2;
  dim3 grid(((Lq + 127) / 128) * nh * B), blk(256);
  if (a.thr16) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, blk, 0, stream, a);
  else if (fwd_version() == 3) hipLaunchKernelGGL((attn_fwd3_kernel<false, false, 3>), grid, blk, 0, stream, a);
  else if (fwd_version() == 4) hipLaunchKernelGGL((attn_fwd3_kernel<false, false, 2>), grid, blk, 0, stream, a);
  else if (fwd_version() == 5) hipLaunchKernelGGL((attn_fwd3_kernel<false, true, 2>), grid, blk, 0, stream, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, blk, 0, stream, a);
  return ia_check_launch();
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
  if (a.thr16) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, gq, blk, 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, gk, blk, 0, stream, a);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, gq, blk, 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, gk, blk, 0, stream, a);
  }
  return ia_check_launch();
}

// Self-attention over a packed projection: q, k, v point at the first column of head 0 of each operand and share
// row stride ld_qkv (packed [tokens, 3H] projection output, or three separate [tokens, H] tensors with ld_qkv = H).
extern "C" int ia_attn_fwd(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, void* out,
                           int ld_o, float* lse2, int B, int nh, int L, float scale, float drop_p, uint32_t seed,
                           hipStream_t stream) {
  return ia_attn_fwd_x(q, ld_qkv, k, v, ld_qkv, key_mask, out, ld_o, lse2, B, nh, L, L, scale, drop_p, seed, stream);
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

extern "C" int ia_attn_bwd_bias(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out,
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
  dim3 grid(((L + 127) / 128) * nh * B), blk(256);
  if (a.thr16) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, blk, 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, blk, 0, stream, a);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, blk, 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, blk, 0, stream, a);
  }
  rc = ia_check_launch();
  if (rc) return rc;
  return ia_sum_rows_f32((const float*)workspace, B * ((L + 127) / 128), 3 * nh * 64, dbias, 1, stream);
}

// Packed ("unpadded") self-attention: the token rows of all sequences lie back to back, sequence b owning rows
// cu_seqlens[b] .. cu_seqlens[b+1] (int32 [B+1], device; total_tokens = cu_seqlens[B]); no key mask — every key of a sequence is
// attendable.  Lmax = longest sequence (sets the grid and the row stride of lse2 / delta, which stay [B, nh, Lmax]).  Same
// arithmetic as ia_attn_fwd / ia_attn_bwd on the valid tokens of a right-padded batch; padded positions are simply absent.
extern "C" int ia_attn_fwd_varlen(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens, void* out,
                                  int ld_o, float* lse2, int B, int nh, int Lmax, float scale, float drop_p, uint32_t seed, hipStream_t stream) {
  (void)hipGetLastError();
  if (!q || !k || !v || !out || !cu_seqlens || total_tokens <= 0) return IA_ERR_ARG;
  AttnArgs a{};
  int rc = fill_args(a, B, nh, Lmax, Lmax, ld_qkv, ld_qkv, ld_o, scale, drop_p, seed, total_tokens);
  if (rc) return rc;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out; a.mask = nullptr; a.lse2 = lse2; a.cu = cu_seqlens;
  dim3 grid(((Lmax + 127) / 128) * nh * B), blk(256);
  if (a.thr16) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, blk, 0, stream, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, blk, 0, stream, a);
  return ia_check_launch();
}

extern "C" int ia_attn_bwd_varlen(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens,
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
  dim3 grid(((Lmax + 127) / 128) * nh * B), blk(256);
  if (a.thr16) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, blk, 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, blk, 0, stream, a);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, blk, 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, blk, 0, stream, a);
  }
  return ia_check_launch();
}
