// bf16 MFMA GEMM for the encoder hot path (gfx950).
//
//   C[M,N] = A[M,K] * B[K,N]   (+ fused epilogue), fp32 accumulate in MFMA accumulators.
//
// Operand memory layouts (no explicit transposes anywhere on the path):
//   A k-contiguous : A[m][k]  (activations / output-gradients, row = token)
//   A k-strided    : At[k][m] (dW = dY^T X : the "A" of that product is dY stored [token][n])
//   B k-contiguous : Bt[n][k] (a torch Linear weight W[N,K] used in the forward)
//   B k-strided    : B[k][n]  (the same W used for dX = dY W, and X in dW = dY^T X)
// k-strided operands are staged row-major into LDS exactly as they lie in HBM and are turned into
// MFMA fragments with ds_read_b64_tr_b16 (the LDS transpose read), so the reference's three GEMM
// flavours (y = xW^T, dx = dy W, dW = dy^T x; torch.nn.Linear under autograd, reference
// src/models/text.py:1241 -> transformers RobertaLayer) all stream each operand from HBM once.
//
// Two tile configurations share staging / epilogue code:
//   T256: 256x256x64 per 512-thread workgroup (8 waves as 2x4, 128x64 per wave, 4x2 MFMA 32x32x16
//         fragments, 128 KiB LDS double buffer) - the large-M encoder GEMMs.  The 128x64 wave tile halves
//         the LDS bytes read per MFMA relative to T128 (whose 64x64 wave tile saturates the LDS port).
//   T128: 128x128x64 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave, 4x4 MFMA 16x16x32) -
//         small outputs (heads, tiny test shapes).
// LDS is filled by buffer_load ... lds (16 B/lane; out-of-range rows arrive as zeros, so ragged M / K
// tails need no extra code) with the XOR swizzle applied on the source side.  The MFMA is issued with
// swapped operands (C^T fragments) so each lane ends up with runs of consecutive output columns.
#include "common.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

// The IA_GEMM_DBG timing ablations (switching the DMA, the barriers, the epilogue, its stores or its math off) exist in tools builds
// only (-DIA_GEMM_DBG_HOOKS=1): left in as run-time tests they put a branch around every GELU pair of the FFN1 epilogue (334 branches
// and 1882 s_nops in that kernel: each pair in its own basic block, the four independent chains of a call never interleaved).
#ifndef IA_GEMM_DBG_HOOKS
#define IA_GEMM_DBG_HOOKS 0
#endif
#define IA_DBG(p) (IA_GEMM_DBG_HOOKS ? (p).dbg : 0)

namespace {

constexpr int BK = 64;
constexpr uint32_t OOB = 0xFFFFFFF0u;

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_ADD = 3, EPI_DGELU = 4, EPI_BIAS_ADD = 5, EPI_DGELU_CS = 6, EPI_BIAS_GELU_ACT = 7 };
// EPI_BIAS_GELU_ACT: the forward-only form of EPI_BIAS_GELU (activation only: no derivative is evaluated or stored)

struct GemmArgs {
  const bf16* A; const bf16* B; void* C; bf16* C2; const float* bias; const bf16* aux;
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  uint64_t a_bytes, b_bytes;  // bytes addressable from A / B (to the end of the tensor); each workgroup re-bases a < 2 GiB buffer window inside them
  int accumulate;
  // IA_EPI_BIAS only: columns n < qcols leave as (acc + bias) * qscale -- the fused QKV projection hands the attention kernels
  // q * (softmax scale * log2 e), rounded to bf16 ONCE, so that forward, dQ, dK/dV and the fused backward exponentiate identical
  // products without any of them re-scaling q (ia_gemm_bf16_qscale); 0 = off
  int qcols; float qscale;
  int tiles_m, tiles_n;
  int splits, nk_per_split;   // split-K (fp32 output only): split s owns k-tiles [s*nk_per_split, ...)
  float* csum_part;           // EPI_DGELU_CS: [ceil(M/128)][N] fp32 column sums of each 128-row block of the output (T256 only)
  int split_id;               // set inside the kernels (derived from the XCD-aware work order)
  float* ws;                  // [splits][M][N] fp32 partials when splits > 1
  // weight-gradient form (A, B k-strided, fp32 C), T128: rsum_out[group * M + m] += sum_k A[k][m] -- the bias gradient of the layer,
  // taken from the A tiles already in LDS by one more MFMA against a fragment of ones (no second pass over dy).
  float* rsum_out;            // NULL = off
  float* rsum_ws;             // [groups][splits][M] partials when splits > 1 (folded by the split-K reduce kernels)
  int dbg;                    // ablation switches for tuning runs (IA_GEMM_DBG env): results are WRONG when non-zero
  // Shifted operand views (T128 only) that turn the GEMM into a 3x3 convolution over a zero-bordered NHWC tensor
  // [B, H+2, W+2, C] without a patch matrix: the k axis is (tap, channel) with a power-of-two channel count per tap, and tap t
  // reads the same rows shifted by tap_delta(t) = (t/3 - 1) * pw + (t%3 - 1).
  //   a_view +1 / -1 (k-contiguous A): k = (tap << lca) + ch reads row + / - tap_delta(tap), column ch        (forward / data grad)
  //   b_view 1 (k-strided B = W[o][(tap << lcbn) + c]): k = (tap << lcbk) + o, column n reads W[o][(tap << lcbn) + n]   (data grad)
  //   b_view 2 (k-strided B = the activations): column n = (tap << lcbn) + ch reads row k + tap_delta(tap), column ch  (weight grad)
  int a_view, b_view, pw, lca, lcbk, lcbn;
  // channel groups: operands advance by ga / gb / gc / gbias elements per group (all groups in one launch)
  int groups;
  long ga, gb, gc, gbias;
  // Dynamic tile claim (t256w, persistent launches): 8 per-XCD claim counters + 1 exit counter of this launch's slot (zero on entry,
  // zeroed again by the last workgroup to leave); NULL = the static order (tile + gridDim.x).
  uint32_t* tile_ctr;
};

// One slot per launch in flight (launches of different streams may overlap; a slot comes round again after CTR_SLOTS launches).
constexpr int CTR_SLOTS = 1024, CTR_WORDS = 16;
__device__ uint32_t g_tile_ctr[CTR_SLOTS * CTR_WORDS];

IA_DEV int tap_delta(int t, int pw) { return (t / 3 - 1) * pw + (t % 3 - 1); }

// Buffer window that starts `origin` elements into an operand of `total_bytes` bytes.  Buffer offsets are 32-bit, operands (an
// 800x800 feature map of 64 images, a 9x patch matrix) are not: every workgroup addresses its tile / k-slab relative to its own
// origin, and the range check still ends at the end of the tensor (lanes past it read zeros).
IA_DEV __amdgpu_buffer_rsrc_t rsrc_at(const bf16* base, uint64_t total_bytes, uint64_t origin) {
  const uint64_t ob = origin * 2, rem = total_bytes > ob ? total_bytes - ob : 0;
  return ia_rsrc(base + origin, (uint32_t)(rem < 0x7FFFFFF0ull ? rem : 0x7FFFFFF0ull));
}

// ---------------------------------------------------------------------------------- shared epilogue
// v = 4 consecutive output columns n..n+3 of row m
template <int EPI, bool OUTF32>
IA_DEV void epi_store4(const GemmArgs& p, int m, int n, f32x4 v) {
  if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_ACT || EPI == EPI_BIAS_ADD) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
    v += b;
    if (EPI == EPI_BIAS && n < p.qcols) v *= p.qscale;
  }
  if (EPI == EPI_BIAS_GELU) {
    // C = gelu(pre), C2 = gelu'(pre): the derivative shares the exp / rcp of the activation (gelu_pair, common.h) and turns the
    // data-gradient epilogue (EPI_DGELU) into one multiply per element
    bf16x4 der;
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      f32x2_t a, d;
      gelu_pair(f32x2_t{v[r], v[r + 1]}, a, d);
      v[r] = a[0]; v[r + 1] = a[1];
      der[r] = f2bf(d[0]); der[r + 1] = f2bf(d[1]);
    }
    *reinterpret_cast<bf16x4*>(p.C2 + (size_t)m * p.ldc + n) = der;
  }
  if (EPI == EPI_BIAS_GELU_ACT) {
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const f32x2_t a = gelu_act_pair(f32x2_t{v[r], v[r + 1]});
      v[r] = a[0]; v[r + 1] = a[1];
    }
  }
  if (EPI == EPI_ADD || EPI == EPI_BIAS_ADD) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bf2f(a[r]);
  }
  if (EPI == EPI_DGELU || EPI == EPI_DGELU_CS) {   // aux = the saved gelu'(pre-activation); T128 leaves the column sums to ia_colsum
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= bf2f(a[r]);
  }
  if (OUTF32) {
    if (p.splits > 1) {   // partial sums; the second-stage kernel adds them into C in a fixed order
      *reinterpret_cast<f32x4*>(p.ws + ((size_t)p.split_id * p.M + m) * p.N + n) = v;
      return;
    }
    float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
    if (p.accumulate) { const f32x4 o = *reinterpret_cast<const f32x4*>(c); v += o; }
    *reinterpret_cast<f32x4*>(c) = v;
  } else {
    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n) = o;
  }
}

// One 16-byte global store = one VMEM instruction (hipcc emits a single global_store_dwordx4 for an aligned 16-byte vector store;
// tools/check_gemm_isa.sh counts them).  The persistent T256 kernel leaves a tile's output stores in flight while the next tile's
// main loop starts and waits with a COUNTED s_waitcnt vmcnt(N) for the k-tile DMA issued before them (vmcnt retires in order), so
// the number of store instructions per epilogue must be known exactly.  The store is a plain (compiler-visible) one on purpose: the
// epilogue's aux / bias loads are software-pipelined two slices ahead of the stores, and only when hipcc counts the stores too does
// it wait for such a load with vmcnt(k > 0) and leave the younger stores in flight (an asm store is invisible to its counters: every
// load wait became vmcnt(0), i.e. a full store + load round trip per 8-row slice, 16 times per tile).
template <typename V>
IA_DEV void gstore16(void* ptr, V v) {
  static_assert(sizeof(V) == 16, "16-byte store");
  *reinterpret_cast<V*>(ptr) = v;      // (a non-temporal store measured the same: tools/README.md, GEMM ablations)
}
template <int EPI, bool OUTF32>
constexpr int epi_stores_per_call() { return OUTF32 ? 2 : (EPI == EPI_BIAS_GELU ? 2 : 1); }
template <int EPI>
constexpr int epi_extra_stores() { return EPI == EPI_DGELU_CS ? 2 : 0; }     // the wave's column-sum partial: two 16-byte stores

// v = 8 consecutive output columns n..n+7 of row m (row-coalesced epilogue of the T256 kernel).  PRE: bias (pb0 | pb1) and aux (ax)
// were fetched ahead by the caller; else they are loaded here.
// BUF (full tiles of the one-wave-per-SIMD kernel, bf16 outputs): the output streams are addressed through buffer windows over the
// wave's rows with one 32-bit lane offset (io.off) instead of 64-bit pointer arithmetic per access (3 VALU instructions each)
struct BufIO { __amdgpu_buffer_rsrc_t rsC, rsC2; uint32_t off; };
template <int EPI, bool OUTF32, bool PRE, bool BUF = false>
IA_DEV void epi_store8(const GemmArgs& p, int m, int n, f32x4 lo, f32x4 hi, f32x4 pb0, f32x4 pb1, bf16x8 ax, float (&cs)[8], const BufIO* io = nullptr) {
  float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_ACT || EPI == EPI_BIAS_ADD) {
    const f32x4 b0 = PRE ? pb0 : *reinterpret_cast<const f32x4*>(p.bias + n), b1 = PRE ? pb1 : *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] += b0[r]; v[4 + r] += b1[r]; }
    if (EPI == EPI_BIAS && n < p.qcols) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] *= p.qscale;
    }
  }
  if (EPI == EPI_BIAS_GELU) {   // see epi_store4
    bf16x8 der;
    {
      // the four pairs of the call stage by stage (gelu_pair's arithmetic, common.h): four independent chains side by side, so the
      // transcendental / packed-op latencies of one hide behind the others (one pair at a time left ~1.8 s_nops per exp2 / rcp)
      constexpr float L2E = 1.4426950408889634f;
      constexpr float C0 = 1.5949398799788077f, C1 = 0.07403000634661838f, C2 = -0.0007007124749191571f;
      constexpr float U_MAX = C1 / (2.f * 0.0007007124749191571f);
      f32x2_t x[4], u[4], t[4], e[4], sg[4], act[4], q[4], d[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[i] = f32x2_t{v[2 * i], v[2 * i + 1]}; u[i] = x[i] * x[i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { u[i][0] = __builtin_fminf(u[i][0], U_MAX); u[i][1] = __builtin_fminf(u[i][1], U_MAX); }
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = x[i] * ((u[i] * (-C2 * L2E) + (-C1 * L2E)) * u[i] + (-C0 * L2E));
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = f32x2_t{__builtin_amdgcn_exp2f(t[i][0]), __builtin_amdgcn_exp2f(t[i][1])};
#pragma unroll
      for (int i = 0; i < 4; ++i) { e[i] = e[i] + 1.0f; q[i] = (u[i] * (5.f * C2) + (3.f * C1)) * u[i] + C0; }
#pragma unroll
      for (int i = 0; i < 4; ++i) sg[i] = f32x2_t{__builtin_amdgcn_rcpf(e[i][0]), __builtin_amdgcn_rcpf(e[i][1])};
#pragma unroll
      for (int i = 0; i < 4; ++i) { act[i] = x[i] * sg[i]; d[i] = (act[i] * (1.0f - sg[i])) * q[i] + sg[i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[2 * i] = act[i][0]; v[2 * i + 1] = act[i][1];
        der[2 * i] = f2bf(d[i][0]); der[2 * i + 1] = f2bf(d[i][1]);
      }
    }
    if (BUF) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, der), io->rsC2, (int)io->off, 0, 0);
    else gstore16(((IA_DBG(p) & 1024) ? reinterpret_cast<bf16*>(p.C) : p.C2) + (size_t)m * p.ldc + n, der);
  }
  if (EPI == EPI_BIAS_GELU_ACT) {      // gelu_act_pair's arithmetic, the four pairs stage by stage (see above)
    constexpr float L2E = 1.4426950408889634f;
    constexpr float C0 = 1.5949398799788077f, C1 = 0.07403000634661838f, C2 = -0.0007007124749191571f;
    constexpr float U_MAX = C1 / (2.f * 0.0007007124749191571f);
    f32x2_t x[4], u[4], t[4], e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = f32x2_t{v[2 * i], v[2 * i + 1]}; u[i] = x[i] * x[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[i][0] = __builtin_fminf(u[i][0], U_MAX); u[i][1] = __builtin_fminf(u[i][1], U_MAX); }
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = x[i] * ((u[i] * (-C2 * L2E) + (-C1 * L2E)) * u[i] + (-C0 * L2E));
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = f32x2_t{__builtin_amdgcn_exp2f(t[i][0]), __builtin_amdgcn_exp2f(t[i][1])} + 1.0f;
    f32x2_t sg[4], act[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sg[i] = f32x2_t{__builtin_amdgcn_rcpf(e[i][0]), __builtin_amdgcn_rcpf(e[i][1])};
#pragma unroll
    for (int i = 0; i < 4; ++i) act[i] = x[i] * sg[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = act[i][0]; v[2 * i + 1] = act[i][1]; }
  }
  if (EPI == EPI_ADD || EPI == EPI_BIAS_ADD || EPI == EPI_DGELU || EPI == EPI_DGELU_CS) {
    const bf16x8 a = PRE ? ax : *reinterpret_cast<const bf16x8*>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (EPI == EPI_DGELU || EPI == EPI_DGELU_CS) v[r] *= bf2f(a[r]);
      else v[r] += bf2f(a[r]);
      if (EPI == EPI_DGELU_CS) cs[r] += v[r];       // this lane's 8 columns, summed over the rows it stores
    }
  }
  if (OUTF32) {
    float* c = p.splits > 1 ? p.ws + ((size_t)p.split_id * p.M + m) * p.N + n : reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
    if (p.splits <= 1 && p.accumulate) { o0 += *reinterpret_cast<const f32x4*>(c); o1 += *reinterpret_cast<const f32x4*>(c + 4); }
    gstore16(c, o0);
    gstore16(c + 4, o1);
  } else {
    bf16x8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = f2bf(v[r]);
    if (BUF) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), io->rsC, (int)io->off, 0, 0);
    else gstore16(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n, o);
  }
}

// XCD-aware work order.  Workgroups are dealt round-robin to the 8 XCDs (each with a private L2): xcd_chunk renumbers them so that
// every XCD takes a contiguous run of the work list (bijective for any grid size) ...
IA_DEV int xcd_chunk(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// ... and the work list sweeps 8 m-tiles per n, so a run of 32 tiles is an 8 x 4 block: 8 A panels and 4 B panels feed 32 tiles
// out of one XCD's L2.
IA_DEV void tile_of_order(const GemmArgs& p, int t, int& bm, int& bn) {
  const int GM = 8;
  const int group = t / (GM * p.tiles_n);
  const int first_m = group * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  bm = first_m + (t % (GM * p.tiles_n)) % gsz;
  bn = (t % (GM * p.tiles_n)) / gsz;
}
IA_DEV void tile_of_index(const GemmArgs& p, int bid, int nwg, int& bm, int& bn) { tile_of_order(p, xcd_chunk(bid, nwg), bm, bn); }

// ============================================================================== T128 (4 waves, 16x16x32)
namespace t128 {
constexpr int BM = 128, BN = 128, TILE_BYTES = 16384;
// "row3" form of a shifted-view convolution GEMM (below): two 136-row A images + two B tiles; else two (A, B) k-tile pairs
constexpr int A3_ROWS = 136, A3_BYTES = A3_ROWS * 128, SMEM_BYTES = 2 * A3_BYTES + 2 * TILE_BYTES;
static_assert(SMEM_BYTES >= 4 * TILE_BYTES && 2 * SMEM_BYTES <= 160 * 1024, "two workgroups per CU");

// 32-byte-slot swizzle of a k-strided tile row (row = k index within the 64-row tile, 256 B rows)
IA_DEV int ks_swz(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

// x0 (k-contiguous) / korg (k-strided, views 0 and 2) are relative to the origin row of the buffer window `rs`
template <bool KS>
IA_DEV void stage_tile(__amdgpu_buffer_rsrc_t rs, char* s, int kt, int x0, int ld, int K, int tid, int wave, int korg, int view = 0, int pw = 0,
                       int lck = 6, int lcn = 6) {
#pragma unroll
  for (int issue = 0; issue < 4; ++issue) {
    uint32_t off;
    if (!KS) {
      const int row = issue * 32 + (tid >> 3);
      const int c = (tid & 7) ^ (row & 7);
      const int k = kt * BK + c * 8;
      if (view == 0) off = (uint32_t)(((x0 + row) * ld + k) * 2);
      else off = (uint32_t)(((x0 + row + view * tap_delta(k >> lck, pw)) * ld + (k & ((1 << lck) - 1))) * 2);   // rows before the tensor wrap to out-of-range
      if (k >= K) off = OOB;
    } else {
      const int row = issue * 16 + (tid >> 4);
      const int c = (tid & 15) ^ ks_swz(row);
      const int k = kt * BK + row;
      if (view == 0) off = (uint32_t)(((k - korg) * ld + x0 + c * 8) * 2);
      else if (view == 1) off = (uint32_t)(((k & ((1 << lck) - 1)) * ld + ((k >> lck) << lcn) + x0 + c * 8) * 2);
      else { const int col = x0 + c * 8; off = (uint32_t)(((k - korg + tap_delta(col >> lcn, pw)) * ld + (col & ((1 << lcn) - 1))) * 2); }
      if (k >= K) off = OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, IA_LDS(s + issue * 4096 + wave * 1024), 16, off, 0, 0, 0);
  }
}

IA_DEV bf16x8 frag_kc(const char* s, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(s + row * 128 + ((chunk ^ (row & 7)) << 4));
}

IA_DEV bf16x8 frag_ks(const char* s, int k, int col) {
  // 16-lane group reads a [4 k][16 col] block twice (k, k+4); lane p supplies row k+(p>>2), 4 cols.
  const int addr = k * 256 + ((((col >> 3) ^ ks_swz(k))) << 4) + (col & 7) * 2;
  const uint32_t a = ia_lds_addr(s) + (uint32_t)addr;   // asm reads: the caller waits lgkmcnt(0) before the MFMAs
  s16x4 lo = ia_tr_read<0>(a);
  s16x4 hi = ia_tr_read<4 * 256>(a);
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <bool AKS, bool BKS, int EPI, bool OUTF32>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // 1-D grid over (k-slab, m-tile, channel group, n-tile), n-tile fastest, renumbered so that every XCD takes a contiguous run of it:
  // the work items that read the same rows of A -- the n-tiles of one slab (for a 3x3 weight gradient: its nine taps, i.e. nine
  // shifted views of the same rows) and the channel groups (128-byte column slices of the same rows) -- then run side by side on ONE
  // XCD and share its L2.  (As a 3-D grid they were dealt round-robin over the 8 XCDs and each XCD fetched the slab for itself.)
  int bm, bn, grp = 0;
  p.split_id = 0;
  if (p.groups > 1 || p.splits > 1) {
    int w = xcd_chunk(blockIdx.x, gridDim.x);
    bn = w % p.tiles_n; w /= p.tiles_n;
    grp = w % p.groups; w /= p.groups;
    bm = w % p.tiles_m;
    p.split_id = w / p.tiles_m;
  } else {
    tile_of_index(p, blockIdx.x, gridDim.x, bm, bn);
  }
  const int m0 = bm * BM, n0 = bn * BN;
  if (p.groups > 1) {                  // uniform: the kernel argument copy is ours to edit
    const long z = grp;
    p.A += z * p.ga; p.B += z * p.gb;
    p.a_bytes -= (uint64_t)(z * p.ga * 2); p.b_bytes -= (uint64_t)(z * p.gb * 2);     // the windows end where the tensors end
    p.C = reinterpret_cast<char*>(p.C) + z * p.gc * (OUTF32 ? 4 : 2);
    if (p.bias) p.bias += z * p.gbias;
    if (p.ws) p.ws += z * (long)p.splits * p.M * p.N;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_all = (p.K + BK - 1) / BK;
  const int kt0 = p.split_id * p.nk_per_split;
  const int nk = min(nk_all, kt0 + p.nk_per_split);
  // buffer windows of this workgroup (rsrc_at): a k-contiguous operand starts at its tile's first row, a k-strided one at its
  // k-slab's first row, both moved back by the largest tap shift (pw + 1 rows) when the operand is read through shifted views
  const int a_org = !AKS ? max(0, m0 - (p.a_view ? p.pw + 1 : 0)) : kt0 * BK;
  const int b_org = !BKS ? n0 : (p.b_view == 1 ? 0 : max(0, kt0 * BK - (p.b_view == 2 ? p.pw + 1 : 0)));
  const __amdgpu_buffer_rsrc_t rsA = rsrc_at(p.A, p.a_bytes, (uint64_t)a_org * p.lda);
  const __amdgpu_buffer_rsrc_t rsB = rsrc_at(p.B, p.b_bytes, (uint64_t)b_org * p.ldb);
  const int xa = AKS ? m0 : m0 - a_org, xb = BKS ? n0 : 0;
  // ---- "row3": a 3x3 shifted-view convolution whose taps are whole k-tiles (64 channels per group: k-tile t = tap t) reads the SAME
  // 128 rows nine times, shifted by (dy, dx) rows.  The L2 -> LDS path is what bounds this kernel on those shapes (9 x the activation
  // bytes at the ~6.4 TB/s the chip's LDS-DMA engines sustain: 0.19 ms for 32 images x 200 x 200 x 64, measured), so the three taps of
  // one dy share ONE image of rows m0-1 .. m0+128 (136 rows fetched) and differ only in the LDS row their fragments start at: 3.2 x
  // the bytes instead of 9 x.  Forward and data gradient (a_view = +-1, k-contiguous A, N <= 64); the weight tiles stay per tap.
  const bool row3 = !AKS && p.a_view != 0 && p.lca == 6 && p.K == 9 * BK && p.splits == 1 && p.N <= 64;
  if (!row3) {
    stage_tile<AKS>(rsA, smem, kt0, xa, p.lda, p.K, tid, wave, a_org, p.a_view, p.pw, p.lca, 0);
    stage_tile<BKS>(rsB, smem + TILE_BYTES, kt0, xb, p.ldb, p.K, tid, wave, b_org, p.b_view, p.pw, p.lcbk, p.lcbn);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const int g = lane >> 4, li = lane & 15;
  // A wave whose 64 x 64 quadrant lies entirely outside C (M or N <= 64: the 64-channel groups of the convolutions -- forward and data
  // gradient have N = 64, the weight gradient M = 64 --, attention heads) would only help with the staging.  Round 5: such a wave
  // JOINS the live quadrant beside it and takes the second k-step of every k-tile (the quadrant's owner keeps the first); the pair adds
  // its accumulators up through LDS behind the loop.  The 64-wide shapes are MFMA-bound in this kernel (a 3x3 convolution over 64
  // channels has 288 FLOP per HBM byte), so halving each live wave's MFMA chain is worth up to 2 x there.
  const bool m_narrow = p.M - m0 <= 64, n_narrow = p.N - n0 <= 64;           // uniform per workgroup
  const bool ksplit = m_narrow || n_narrow;
  int qm = wm, qn = wn, my_ks = 0;
  bool helper_idle = false;
  if (n_narrow && !m_narrow) { qn = 0; my_ks = wn; }
  else if (m_narrow && !n_narrow) { qm = 0; my_ks = wm; }
  else if (m_narrow && n_narrow) { qm = 0; qn = 0; my_ks = wave & 1; helper_idle = wave >= 2; }
  const bool quadrant_live = !helper_idle && m0 + qm * 64 < p.M && n0 + qn * 64 < p.N;
  constexpr bool WGRAD = AKS && BKS && OUTF32 && EPI == EPI_NONE;
  const bool row_sums = WGRAD && p.rsum_out != nullptr && bn == 0 && qn == 0;      // uniform per wave
  f32x4 racc[4];
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 4; ++i) racc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = f2bf(1.0f);
  if (row3) {
    char* const A3 = smem;
    char* const B2 = smem + 2 * A3_BYTES;
    // LDS row j of image gy <- tensor row m0 + j - 1 + a_view (gy - 1) pw (rows before the tensor wrap out of range: zeros), chunk
    // position XOR (j & 7) as frag_kc reads it; 4 issues of 32 rows by all waves + 8 rows by wave 0
    auto stage_a3 = [&](int gy, int b3) {
      char* const s3 = A3 + b3 * A3_BYTES;
      const int shift = p.a_view * (gy - 1) * p.pw - 1;
#pragma unroll
      for (int issue = 0; issue < 4; ++issue) {
        const int j = issue * 32 + (tid >> 3);
        const uint32_t off = (uint32_t)(((xa + j + shift) * p.lda + (((tid & 7) ^ (j & 7)) * 8)) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, IA_LDS(s3 + issue * 4096 + wave * 1024), 16, off, 0, 0, 0);
      }
      if (wave == 0) {
        const int j = 128 + (lane >> 3);
        const uint32_t off = (uint32_t)(((xa + j + shift) * p.lda + (((lane & 7) ^ (j & 7)) * 8)) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, IA_LDS(s3 + 16384), 16, off, 0, 0, 0);
      }
    };
    stage_a3(0, 0);
    stage_tile<BKS>(rsB, B2, 0, xb, p.ldb, p.K, tid, wave, b_org, p.b_view, p.pw, p.lcbk, p.lcbn);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < 9; ++kt) {
      const int gy = kt / 3, dxi = kt - 3 * gy, bb = kt & 1;
      if (kt + 1 < 9) stage_tile<BKS>(rsB, B2 + (bb ^ 1) * TILE_BYTES, kt + 1, xb, p.ldb, p.K, tid, wave, b_org, p.b_view, p.pw, p.lcbk, p.lcbn);
      if (dxi == 0 && gy < 2) stage_a3(gy + 1, (gy + 1) & 1);
      const char* sA = A3 + (gy & 1) * A3_BYTES;
      const char* sB = B2 + bb * TILE_BYTES;
      const int rowoff = 1 + p.a_view * (dxi - 1);
      if (quadrant_live)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ksplit && ks != my_ks) continue;
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = frag_kc(sA, qm * 64 + mi * 16 + li + rowoff, ks * 4 + g);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          if (!BKS) bfr[ni] = frag_kc(sB, qn * 64 + (li >> 2) * 16 + ni * 4 + (li & 3), ks * 4 + g);
          else      bfr[ni] = frag_ks(sB, ks * 32 + g * 8 + (li >> 2), qn * 64 + ni * 16 + (li & 3) * 4);
        }
        if (BKS) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else
  for (int kt = kt0; kt < nk; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < nk) {
      char* nb = smem + (buf ^ 1) * 2 * TILE_BYTES;
      stage_tile<AKS>(rsA, nb, kt + 1, xa, p.lda, p.K, tid, wave, a_org, p.a_view, p.pw, p.lca, 0);
      stage_tile<BKS>(rsB, nb + TILE_BYTES, kt + 1, xb, p.ldb, p.K, tid, wave, b_org, p.b_view, p.pw, p.lcbk, p.lcbn);
    }
    const char* sA = smem + buf * 2 * TILE_BYTES;
    const char* sB = sA + TILE_BYTES;
    if (quadrant_live)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ksplit && ks != my_ks) continue;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if (!AKS) af[mi] = frag_kc(sA, qm * 64 + mi * 16 + li, ks * 4 + g);
        else      af[mi] = frag_ks(sA, ks * 32 + g * 8 + (li >> 2), qm * 64 + mi * 16 + (li & 3) * 4);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        // k-contiguous B: fragment row i <-> n = (i>>2)*16 + ni*4 + (i&3), so lane group g ends up
        // holding 16 consecutive n; k-strided B: plain n = ni*16 + i (conflict-free transpose read).
        if (!BKS) bfr[ni] = frag_kc(sB, qn * 64 + (li >> 2) * 16 + ni * 4 + (li & 3), ks * 4 + g);
        else      bfr[ni] = frag_ks(sB, ks * 32 + g * 8 + (li >> 2), qn * 64 + ni * 16 + (li & 3) * 4);
      }
      if (AKS || BKS) {   // the transpose reads are asm (common.h): order the MFMAs behind their data
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
      if (WGRAD && row_sums)          // D[n][m] = sum_k 1 * A[m][k] for every n: each lane's column li holds the sum of its m
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) racc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[mi], racc[mi], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (ksplit) {
    // the helper (k-step 1) hands its sums to the quadrant's owner: 20 x 16-byte columns per lane, lane-major inside a column so
    // that both sides move whole 1-KiB lines; the k-tile buffers are free behind the loop's last barrier
    float* const xch = reinterpret_cast<float*>(smem) + (size_t)(qm * 2 + qn) * (20 * 64 * 4);
    if (quadrant_live && my_ks == 1) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(xch + ((mi * 4 + ni) * 64 + lane) * 4) = acc[mi][ni];
        *reinterpret_cast<f32x4*>(xch + ((16 + mi) * 64 + lane) * 4) = racc[mi];
      }
    }
    __syncthreads();
    if (!quadrant_live || my_ks == 1) return;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] += *reinterpret_cast<const f32x4*>(xch + ((mi * 4 + ni) * 64 + lane) * 4);
      racc[mi] += *reinterpret_cast<const f32x4*>(xch + ((16 + mi) * 64 + lane) * 4);
    }
  }

#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + qm * 64 + mi * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + qn * 64 + (BKS ? ni * 16 + g * 4 : g * 16 + ni * 4);
      if (n >= p.N) continue;
      epi_store4<EPI, OUTF32>(p, m, n, acc[mi][ni]);
    }
  }
  if (WGRAD && row_sums && g == 0) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = m0 + qm * 64 + mi * 16 + li;
      if (m >= p.M) continue;
      if (p.splits > 1) p.rsum_ws[((size_t)grp * p.splits + p.split_id) * p.M + m] = racc[mi][0];
      else p.rsum_out[(size_t)grp * p.M + m] += racc[mi][0];
    }
  }
}
}  // namespace t128

// ============================================================================== T256 (8 waves, 32x32x16)
namespace t256 {
constexpr int BM = 256, BN = 256, TILE_BYTES = 32768;
constexpr int STAGE_BYTES = 16 * 64 * 4;                   // per-wave epilogue slot: 16 rows x 64 columns fp32
constexpr int LDS_BYTES = 2 * 2 * TILE_BYTES + 8 * STAGE_BYTES;   // 128 KiB k-tile double buffer + 32 KiB = all 160 KiB of the CU

template <bool KS>
IA_DEV void stage_tile(__amdgpu_buffer_rsrc_t rs, char* s, int kt, int x0, int ld, int K, int tid, int wave) {
  // k-contiguous: [256 rows(x)][64 k], 128 B rows, chunk XOR (row>>1)&7 (conflict-free ds_read_b128 for any
  // 16 rows distinct mod 16); k-strided: [64 rows(k)][256 x], 512 B rows, 32 B slot XOR (k&3)
#pragma unroll
  for (int issue = 0; issue < 4; ++issue) {
    uint32_t off;
    if (!KS) {
      const int row = issue * 64 + (tid >> 3);
      const int c = (tid & 7) ^ ((row >> 1) & 7);
      const int k = kt * BK + c * 8;
      off = (uint32_t)(((x0 + row) * ld + k) * 2);
      if (k >= K) off = OOB;
    } else {
      const int row = issue * 16 + (tid >> 5);
      const int c = (tid & 31) ^ ((row & 3) << 2);
      const int k = kt * BK + row;
      off = (uint32_t)((k * ld + x0 + c * 8) * 2);
      if (k >= K) off = OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, IA_LDS(s + issue * 8192 + wave * 1024), 16, off, 0, 0, 0);
  }
}

// fragment of MFMA 32x32x16: lane (i = lane&31, half = lane>>5) holds operand row i, k = half*8 .. +7
IA_DEV bf16x8 frag_kc(const char* s, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(s + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// same fragment out of a k-strided tile: each 16-lane group transposes a [4 k][16 col] block; lane (i -> column
// col0 + i, half) gets k rows k0 + 8*half + {0..3} from the first read and + {4..7} from the second, i.e. the
// standard k order of the MFMA operand.
IA_DEV bf16x8 frag_ks(const char* s, int k0, int col0, int lane) {
  const int p = lane & 15, G = lane >> 4;
  const int row = k0 + 8 * (G >> 1) + (p >> 2);
  const int col = col0 + 16 * (G & 1) + (p & 3) * 4;
  const int addr = row * 512 + ((((col >> 3) ^ ((row & 3) << 2))) << 4) + (col & 7) * 2;
  const uint32_t a = ia_lds_addr(s) + (uint32_t)addr;   // asm reads: main_loop waits lgkmcnt(0) before the MFMAs
  s16x4 lo = ia_tr_read<0>(a);
  s16x4 hi = ia_tr_read<4 * 512>(a);
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// Ping-pong main loop of one wave group (GRP 0: rows 0..127 of the block tile and the A-operand DMA;
// GRP 1: rows 128..255 and the B-operand DMA).  See the schedule comment in gemm_kernel.
template <int GRP, bool AKS, bool BKS, int PEND>
IA_DEV void main_loop(const GemmArgs& p, char* smem, f32x16 (&acc)[4][2], __amdgpu_buffer_rsrc_t rs, int x0, int ld, int kt0, int kta0,
                      int n_tiles, int nk_all, int wn, int lane, bool prologue_only, bool stores_in_flight) {
  // rs is this workgroup's buffer window (rsrc_at): x0 and the k-tile index used for ADDRESSES (kta0 + u) are relative to its
  // origin -- tile row 0 of a k-contiguous operand (kta0 = kt0: the k offset is a column), slab row 0 of a k-strided one (kta0 = 0)
  constexpr bool MYKS = GRP ? BKS : AKS;            // layout of the operand this group streams
  const int hh = lane >> 5, li = lane & 31;
  const int gt = wn * 64 + lane;                    // thread index inside the group (0..255)
  // One per-lane byte offset serves all 8 DMA pieces of a half tile: the swizzled chunk a lane fetches does not
  // depend on the piece (the row advance per piece is a multiple of the swizzle period), so piece and k-tile
  // advances are both scalars folded into the instruction's soffset.
  uint32_t voff0, piece_step;
  if (!MYKS) {
    const int row = gt >> 3;
    voff0 = (uint32_t)(((x0 + row) * ld + (((gt & 7) ^ ((row >> 1) & 7)) * 8)) * 2);
    piece_step = (uint32_t)(32 * ld * 2);
  } else {
    const int row = gt >> 5;
    voff0 = (uint32_t)((row * ld + x0 + (((gt & 31) ^ ((row & 3) << 2)) * 8)) * 2);
    piece_step = (uint32_t)(8 * ld * 2);
  }
  const uint32_t kstep = MYKS ? (uint32_t)(BK * ld * 2) : (uint32_t)(BK * 2);
  const bool ragged_k = (p.K & (BK - 1)) != 0;
  char* const my_half = smem + (GRP ? TILE_BYTES : 0) + wn * 1024;

  auto dma_piece = [&](int u, int i) {             // piece i of this group's half of k-tile u -> buffer u&1
    char* dst = my_half + (u & 1) * 2 * TILE_BYTES + i * 4096;
    const int kt = kt0 + u, kta = kta0 + u;
    if (!ragged_k || kt != nk_all - 1) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, IA_LDS(dst), 16, voff0, (int)(kta * kstep + i * piece_step), 0, 0);
    } else {                                        // last, partial k-tile: lanes past K fetch zeros
      const int k = MYKS ? kt * BK + i * 8 + (gt >> 5) : kt * BK + ((gt & 7) ^ (((gt >> 3) >> 1) & 7)) * 8;
      const uint32_t off = k < p.K ? voff0 + kta * kstep + i * piece_step : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, IA_LDS(dst), 16, off, 0, 0, 0);
    }
  };

  if (prologue_only) {       // called ahead of time (before the previous tile's epilogue): just start the first two k-tiles
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(0, i);
    if (n_tiles > 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dma_piece(1, i);
    }
    return;
  }
  // the prologue DMA of this tile.  After a full-tile epilogue exactly PEND store instructions were issued behind it and
  // may stay in flight (vmcnt retires in order: at most PEND outstanding <=> every older DMA piece has landed).
  if (stores_in_flight) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PEND) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (GRP == 1) __builtin_amdgcn_s_barrier();      // G1 idles through phase 0

  // B fragment row i <-> n so that a lane ends up with 16 consecutive output columns (k-contiguous B only)
  const int nperm = ((li >> 2) & 1) * 16 + (li >> 3) * 4 + (li & 3);
  const bool dma_on = !(IA_DBG(p) & 2);

  for (int u = 0; u < n_tiles; ++u) {
    const char* sA = smem + (u & 1) * 2 * TILE_BYTES;
    const char* sB = sA + TILE_BYTES;
    // ------------------------------------------------------------------ LOAD phase
    if (GRP == 0 && u >= 1 && u + 1 < n_tiles && dma_on) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dma_piece(u + 1, i);
    }
    bf16x8 af[4][4], bfr[4][2];
    if (!(IA_DBG(p) & 8) || u == 0)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if (AKS) af[ks][mi] = frag_ks(sA, ks * 16, GRP * 128 + mi * 32, lane);
        else af[ks][mi] = frag_kc(sA, GRP * 128 + mi * 32 + li, ks * 2 + hh);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (BKS) bfr[ks][ni] = frag_ks(sB, ks * 16, wn * 64 + ni * 32, lane);
        else bfr[ks][ni] = frag_kc(sB, wn * 64 + ni * 32 + nperm, ks * 2 + hh);
      }
    }
    // G1's DMA issued in its previous COMPUTE (none yet at u == 0: do not drain the previous tile's stores there)
    if (GRP == 1 && u >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(IA_DBG(p) & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ------------------------------------------------------------------ COMPUTE phase
    // G1 streams its 8 DMA pieces of k-tile u+2 in the shadow of its own MFMAs: one piece per 4 MFMAs
    const bool stage_now = GRP == 1 && u + 2 < n_tiles && dma_on;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if (GRP == 1 && (mi & 1) == 0 && stage_now) dma_piece(u + 2, ks * 2 + (mi >> 1));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
    if (GRP == 0 && u >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // G0's DMA issued in this iteration's LOAD (u >= 1)
    __builtin_amdgcn_sched_barrier(0);
    if (!(IA_DBG(p) & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (GRP == 0) __builtin_amdgcn_s_barrier();      // matches G1's last phase
}

template <bool AKS, bool BKS, int EPI, bool OUTF32>
__global__ __launch_bounds__(512) void gemm_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves, 128 x 64 each; waves w and w+4 share a SIMD
  const int nk_all = (p.K + BK - 1) / BK;
  const int total_tiles = p.tiles_m * p.tiles_n;
  // Split-K launches are a 1-D grid of tiles x splits workgroups whose XCD-aware order has the split OUTERMOST: one XCD's run of
  // 32 work items is then an 8 x 4 block of tiles of ONE k-slab, i.e. 12 operand panel streams per XCD instead of 36 when the
  // splits of a tile shared an XCD (HBM bytes of the fc1 weight gradient: 2.2 GB -> 0.9 GB per launch, algorithmic 0.67 GB).
  int first_tile = blockIdx.x;          // index into the tile work order
  bool ordered = false;                 // first_tile already is a position of that order (no XCD renumbering inside run())
  p.split_id = 0;
  if (p.splits > 1) {
    const int w = xcd_chunk(blockIdx.x, gridDim.x);
    p.split_id = w / total_tiles;
    first_tile = w % total_tiles;
    ordered = true;
  }
  const int kt0 = p.split_id * p.nk_per_split;
  const int n_tiles = min(nk_all, kt0 + p.nk_per_split) - kt0;

  // Ping-pong schedule.  The 8 waves form two groups (grp = wm: rows 0..127 / 128..255 of the block tile); every
  // SIMD hosts one wave of each group.  A wave alternates a LOAD phase (all 24 fragment reads of one k-tile into
  // registers) and a COMPUTE phase (its 32 MFMAs from registers); group 1 runs one phase behind group 0, so on
  // each SIMD one wave feeds the matrix pipe while the other one reads LDS.  One s_barrier per phase boundary.
  //   phase 2u   : G0 LOAD(u)     | G1 COMPUTE(u-1)
  //   phase 2u+1 : G0 COMPUTE(u)  | G1 LOAD(u)
  // LDS holds two k-tiles; tile u+2 goes into the buffer of tile u once both groups have read it (after phase
  // 2u+1): G0 streams the A half during its LOAD(u+1), G1 the B half between the MFMAs of its COMPUTE(u), and
  // each waits for its own DMA one phase later, i.e. before the barrier that ends phase 2u+3.
  //
  // The workgroup is persistent over output tiles (one workgroup per CU): the DMA of the NEXT tile's first two
  // k-tiles is issued before the current tile's epilogue (which stages through LDS outside the k-tile buffers), so the
  // ~2 us HBM round trip of a tile prologue hides behind the epilogue instead of idling the CU.
  constexpr int PEND = 16 * epi_stores_per_call<EPI, OUTF32>() + epi_extra_stores<EPI>();   // store instructions of one full-tile epilogue, per wave
  static_assert(PEND < 60, "vmcnt is a 6-bit counter");
  auto coords = [&](int tile, int& bm, int& bn) {
    if (ordered) tile_of_order(p, tile, bm, bn); else tile_of_index(p, tile, total_tiles, bm, bn);
  };
  auto run = [&](int tile, bool prologue_only, f32x16 (&acc)[4][2], bool stores_in_flight) {
    int bm, bn;
    coords(tile, bm, bn);
    // keep per-lane address arithmetic from being hoisted out of the tile loop (it would stay live across the
    // epilogue and push the 256-register kernel into scratch): every call derives it afresh from an opaque lane id
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    if (wm == 0) main_loop<0, AKS, BKS, PEND>(p, smem, acc, rsrc_at(p.A, p.a_bytes, (uint64_t)(AKS ? kt0 * BK : bm * BM) * p.lda), AKS ? bm * BM : 0, p.lda,
                                               kt0, AKS ? 0 : kt0, n_tiles, nk_all, wn, lane, prologue_only, stores_in_flight);
    else         main_loop<1, AKS, BKS, PEND>(p, smem, acc, rsrc_at(p.B, p.b_bytes, (uint64_t)(BKS ? kt0 * BK : bn * BN) * p.ldb), BKS ? bn * BN : 0, p.ldb,
                                               kt0, BKS ? 0 : kt0, n_tiles, nk_all, wn, lane, prologue_only, stores_in_flight);
  };

  if ((IA_DBG(p) & 2048) && ((blockIdx.x >> 3) & 1)) {     // ablation: every other CU of an XCD starts (dbg >> 16) us late
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)(IA_DBG(p) >> 16) * 100ull) __builtin_amdgcn_s_sleep(8);
  }
  f32x16 acc[4][2];
  int tile = first_tile;
  run(tile, true, acc, false);
  bool stores_in_flight = false;
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    run(tile, false, acc, stores_in_flight);
    const int next = ordered ? total_tiles : tile + gridDim.x;      // a split-K workgroup owns exactly one (tile, k-slab)

    // The k-tile buffers are free once the main loop's last barrier has passed: start the NEXT tile's first two k-tiles
    // now, so their HBM round trip (~2 us) runs under this tile's epilogue.
    if (next < total_tiles) run(next, true, acc, false);

    // Epilogue through a wave-private LDS slot outside the k-tile buffers (no workgroup barrier).  The MFMA C^T fragments
    // give each lane 4-element runs scattered over 32 rows, which as direct global stores cost ~11 us per tile (64 separate
    // segments per store instruction).  Instead a wave stages 16 rows x 64 columns of fp32 (4 KiB, 16-byte chunks XOR-swizzled
    // by row: conflict-free both ways), then streams them out: bias / residual / GELU inputs are read and all outputs written
    // as full 128-byte row segments, 8 rows per instruction.  C^T fragment: lane (m = li, half hh) register r <-> fragment
    // row i = (r&3) + 8*(r>>2) + 4*hh; k-contiguous B: n = hh*16 + r; k-strided B: n = i.
    int bm, bn;
    coords(tile, bm, bn);
    const int m0 = bm * BM + wm * 128, n0 = bn * BN + wn * 64;
    int lane_e = lane0;
    asm volatile("" : "+v"(lane_e));
    const int hh = lane_e >> 5, li = lane_e & 31;
    char* stg = smem + 2 * 2 * TILE_BYTES + wave * STAGE_BYTES;
    const int wrow = li & 15, rrow = lane_e >> 3, c8 = lane_e & 7;
    // this wave's 128 x 64 part lies entirely inside C: no row / column guards, the store count of the tile is exact
    const bool full = m0 + 128 <= p.M && n0 + 64 <= p.N;
    constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_ACT || EPI == EPI_BIAS_ADD;
    constexpr bool HAS_AUX = EPI == EPI_ADD || EPI == EPI_BIAS_ADD || EPI == EPI_DGELU || EPI == EPI_DGELU_CS;
    // Slice c (c = 0..15) = rows (c>>2)*32 + ((c>>1)&1)*16 + (c&1)*8 + rrow of this wave's part, the lane's 8 columns n0 + c8*8.
    // PRE (full parts): the bias - the same 8 columns for all slices - is read once per tile, and the aux operand runs AHEAD slices
    // ahead of the stores: VMEM retires in order, so a load issued behind a store can only be waited for by draining that store;
    // issued ahead, hipcc's own counters leave the younger stores in flight (vmcnt(2 * AHEAD) in the steady state).
    constexpr int AHEAD = 4;
    auto drain_tile = [&](auto PREFETCHED) {
      constexpr bool PRE = decltype(PREFETCHED)::value;
      f32x4 pb0, pb1;          // only read when the epilogue has a bias (left undefined otherwise: no registers, no spill)
      float cs[8];
      if (EPI == EPI_DGELU_CS) {      // zeroed here, opaquely: a hoisted zero vector would live (and spill) across the whole main loop
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_mov_b32 %0, 0" : "=v"(cs[j]));
      }
      bf16x8 ax[AHEAD + 1];
      auto aux_of = [&](int c) {
        const int row = m0 + (c >> 2) * 32 + ((c >> 1) & 1) * 16 + (c & 1) * 8 + rrow;
        return *reinterpret_cast<const bf16x8*>(p.aux + (size_t)row * p.ldaux + n0 + c8 * 8);
      };
      if (PRE && HAS_BIAS) { pb0 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c8 * 8); pb1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c8 * 8 + 4); }
      if (PRE && HAS_AUX) {
#pragma unroll
        for (int c = 0; c < AHEAD; ++c) ax[c] = aux_of(c);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int h16 = 0; h16 < 2; ++h16) {
          if ((li >> 4) == h16) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) {
                const int chunk = (ni * 32 + (BKS ? rg * 8 + hh * 4 : hh * 16 + rg * 4)) >> 2;
                const f32x4 v = {acc[mi][ni][rg * 4], acc[mi][ni][rg * 4 + 1], acc[mi][ni][rg * 4 + 2], acc[mi][ni][rg * 4 + 3]};
                *reinterpret_cast<f32x4*>(stg + wrow * 256 + ((chunk ^ wrow) << 4)) = v;
              }
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int row = it * 8 + rrow;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * c8) ^ row) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * c8 + 1) ^ row) << 4));
            const int m = m0 + mi * 32 + h16 * 16 + row, n = n0 + c8 * 8;
            const int c = mi * 4 + h16 * 2 + it;
            if (PRE) {
              if (HAS_AUX && c + AHEAD < 16) {
                ax[(c + AHEAD) % (AHEAD + 1)] = aux_of(c + AHEAD);
                asm volatile("" ::: "memory");      // the load stays in front of this slice's store (hipcc would sink it behind)
              }
              if (!(IA_DBG(p) & 64)) epi_store8<EPI, OUTF32, true>(p, m, n, lo, hi, pb0, pb1, ax[c % (AHEAD + 1)], cs);
            } else {
              if (m < p.M && n < p.N && !(IA_DBG(p) & 64)) epi_store8<EPI, OUTF32, false>(p, m, n, lo, hi, pb0, pb1, ax[0], cs);
            }
            if (IA_DBG(p) & 64) asm volatile("" : : "v"(lo), "v"(hi));
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      if (EPI == EPI_DGELU_CS) {
        // column sums of this wave's 128 rows: the 8 lanes that share a column group (lane & 7) differ in lane bits 3..5
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float v = cs[r];
          v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x128, 0xF, 0xF,
                                                                     false));                    // row_ror:8   -> lane ^ 8
          v = ia_add_xor32(ia_add_xor16(v));                                                      // lane ^ 16, lane ^ 32 (common.h)
          cs[r] = v;
        }
        if (rrow == 0 && n0 + c8 * 8 < p.N) {      // lanes 0..7: 8 consecutive columns each (two 16-byte stores, counted in PEND)
          float* dst = p.csum_part + (size_t)(m0 >> 7) * p.N + n0 + c8 * 8;
          gstore16(dst, f32x4{cs[0], cs[1], cs[2], cs[3]});
          gstore16(dst + 4, f32x4{cs[4], cs[5], cs[6], cs[7]});
        }
      }
    };
    if (!(IA_DBG(p) & 32)) {
      if (full && (HAS_BIAS || HAS_AUX) && !(IA_DBG(p) & 256)) drain_tile(std::true_type{}); else drain_tile(std::false_type{});
    }
    if (next >= total_tiles) break;
    // a wave whose 128 x 64 part of the tile was clipped by M or N issued fewer stores than PEND: drain instead of counting
    stores_in_flight = !(IA_DBG(p) & 96) && full;
    if (!stores_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile = next;
  }
}
}  // namespace t256

// ============================================================================== T256W (4 waves, 128 x 128 per wave, 32x32x16)
// Same 256 x 256 x 64 block tile, LDS image, DMA pieces, tile order and epilogue as T256, but ONE wave per SIMD owning a 128 x 128
// part (256 accumulators in AGPRs): 8 fragment reads per 16 MFMAs instead of 12, i.e. 128 + 64 KiB of LDS traffic per k-tile where
// T256 moves 192 + 64 KiB.  With one wave per SIMD nothing else hides latency, so the wave pipelines itself, on the schedule of
// the library's hand-written 256x256x64 kernels (read off their disassembly): the WHOLE k-tile of fragments lives in registers
// (four sets of 8 fragments), so the LDS buffer of k-tile u is free after the first HALF of iteration u; k-tile u+2 is then
// requested piece by piece between the MFMAs of the second half (one 1-KiB LDS-DMA per two MFMAs, never a burst), stays in flight
// for a whole iteration and is only waited for with a counted vmcnt(8) in front of the last k-step of iteration u+1, where the first
// fragments of k-tile u+1 are read.  Two workgroup barriers per k-tile, each behind issued MFMAs.
//   k-step 0: MFMAs on set 0 | fragment reads of sets 1 and 2 (one per MFMA)
//   k-step 1: MFMAs on set 1 | reads of set 3;  lgkmcnt(0), barrier B1: every wave holds all of k-tile u -> its buffer is free
//   k-step 2: MFMAs on set 2 | DMA pieces 0..7 of k-tile u+2
//   k-step 3: vmcnt(8), barrier B2: k-tile u+1 has landed;  MFMAs on set 3 | reads of set 0 of k-tile u+1, DMA pieces 8..15
#ifndef IA_T256W_ROUND
#define IA_T256W_ROUND 1       // which GEMM forms run the ROUND k-loop schedule: 0 none, 1 k-contiguous A and B (forward GEMMs), 2 all but the weight-gradient form
#endif
#ifndef IA_T256W_MFMA16
#define IA_T256W_MFMA16 0
#endif
#ifndef IA_T256W_NOREADS
#define IA_T256W_NOREADS 0
#endif
#ifndef IA_T256W_SPREAD3
#define IA_T256W_SPREAD3 0     // second k-loop schedule (a k-strided operand): DMA pieces one per three MFMAs over k-steps 2, 3 and the next k-step 0
                               // -- measured: data gradient +-0, weight gradient -0.9 % (what helps the forward forms does not help these)
#endif
#ifndef IA_T256W_KS_LANEOFF
#define IA_T256W_KS_LANEOFF 1  // k-strided operands: whole DMA address in the lane offset (hardware range check) instead of compare + select per piece
#endif

namespace t256w {
using t256::BM;
using t256::BN;
using t256::TILE_BYTES;
using t256::STAGE_BYTES;
using t256::LDS_BYTES;

// One operand's four fragments of a k-step.  A k-strided operand's fragment arrives as two transpose reads: the halves are kept
// apart until the wait (the wait asm ties the RAW read destinations; assembling the 128-bit value earlier could be scheduled as
// register copies in front of the wait).
template <bool KS> struct Op;
template <> struct Op<false> { bf16x8 v[4]; };
template <> struct Op<true> { s16x4 lo[4], hi[4]; };
IA_DEV bf16x8 frag_of(const Op<false>& o, int j) { return o.v[j]; }
IA_DEV bf16x8 frag_of(const Op<true>& o, int j) {
  s16x8 r = {o.lo[j][0], o.lo[j][1], o.lo[j][2], o.lo[j][3], o.hi[j][0], o.hi[j][1], o.hi[j][2], o.hi[j][3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <int IMM>
IA_DEV bf16x8 rd128(uint32_t addr) {
  bf16x8 d;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
  return d;
}
template <int IMM>
IA_DEV s16x4 rd_tr(uint32_t addr) {
  s16x4 d;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
  return d;
}

// Per-lane LDS byte offsets of one operand's fragments inside a k-tile buffer.  k-contiguous tile ([256 rows][64 k], chunk XOR
// (row>>1)&7): fragment (j, ks) of rows x0 + j*32 + li sits at base[ks] + j*4096 -- the XOR only depends on the lane and ks.
// k-strided tile ([64 k][256 x], 32-byte slot XOR (k&3)<<2): fragment (j, ks) of columns x0 + j*32.. sits at base[j] + ks*8192 --
// the XOR lands on the bits j occupies, so it is folded per j.  (x0 = 0 or 128: the wave's half of the tile.)
template <bool KS>
IA_DEV void frag_bases(uint32_t (&base)[4], uint32_t tile_addr, int x0, int lane, int nperm_row) {
  if (!KS) {
    const int hh = lane >> 5;
    const int row = x0 + nperm_row;                       // li, or the permuted row of a k-contiguous B operand
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) base[ks] = tile_addr + row * 128 + ((((ks * 2 + hh) ^ ((row >> 1) & 7))) << 4);
  } else {
    const int p = lane & 15, G = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 8 * (G >> 1) + (p >> 2);            // + ks*16 through the immediate
      const int col = x0 + j * 32 + 16 * (G & 1) + (p & 3) * 4;
      base[j] = tile_addr + row * 512 + ((((col >> 3) ^ ((row & 3) << 2))) << 4) + (col & 7) * 2;
    }
  }
}

// wait until at most N LDS operations are outstanding and tie the read destinations to the wait (the MFMAs that consume them
// cannot be scheduled above it)
template <int N> IA_DEV void tie(Op<false>& o) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]) : "n"(N));
}
// several fragment sets behind ONE wait (k-contiguous sets: 4 registers each; six sets are 24 of the 30 operands an asm may have)
template <int N> IA_DEV void tie2(Op<false>& a, Op<false>& b) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]) : "n"(N));
}
template <int N> IA_DEV void tie6(Op<false>& a, Op<false>& b, Op<false>& c, Op<false>& d, Op<false>& e, Op<false>& f) {
  asm volatile("s_waitcnt lgkmcnt(%24)"
               : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]),
                 "+v"(c.v[0]), "+v"(c.v[1]), "+v"(c.v[2]), "+v"(c.v[3]), "+v"(d.v[0]), "+v"(d.v[1]), "+v"(d.v[2]), "+v"(d.v[3]),
                 "+v"(e.v[0]), "+v"(e.v[1]), "+v"(e.v[2]), "+v"(e.v[3]), "+v"(f.v[0]), "+v"(f.v[1]), "+v"(f.v[2]), "+v"(f.v[3])
               : "n"(N));
}
template <int N> IA_DEV void tie(Op<true>& o) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(o.lo[0]), "+v"(o.lo[1]), "+v"(o.lo[2]), "+v"(o.lo[3]), "+v"(o.hi[0]), "+v"(o.hi[1]), "+v"(o.hi[2]), "+v"(o.hi[3]) : "n"(N));
}
template <int N> IA_DEV void tie2(Op<true>& a, Op<true>& b) { tie<N>(a); tie<N>(b); }
template <int N> IA_DEV void tie6(Op<true>& a, Op<true>& b, Op<true>& c, Op<true>& d, Op<true>& e, Op<true>& f) {
  tie<N>(a); tie<N>(b); tie<N>(c); tie<N>(d); tie<N>(e); tie<N>(f);
}


// fragment j (0..3: A, 4..7: B) of k-step S of the k-tile at bufoff
template <int S, bool KS>
IA_DEV void read_frag(Op<KS>& f, int j, const uint32_t (&base)[4], uint32_t bufoff) {
#if IA_T256W_NOREADS      // timing experiment only: the MFMAs run on whatever the registers hold
  return;
#endif
  if constexpr (!KS) {
    const uint32_t a = base[S] + bufoff;
    if (j == 0) f.v[0] = rd128<0>(a);
    if (j == 1) f.v[1] = rd128<4096>(a);
    if (j == 2) f.v[2] = rd128<8192>(a);
    if (j == 3) f.v[3] = rd128<12288>(a);
  } else {
    f.lo[j] = rd_tr<S * 8192>(base[j] + bufoff);
    f.hi[j] = rd_tr<S * 8192 + 4 * 512>(base[j] + bufoff);
  }
}

#ifndef IA_T256W_PEEL
#define IA_T256W_PEEL 1
#endif
template <bool AKS, bool BKS, int PEND, bool PEEL_OK = true>
IA_DEV void main_loop(const GemmArgs& p, char* smem, f32x16 (&acc)[4][4], __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, int xa, int xb,
                      int kt0, int ktaA0, int ktaB0, int n_tiles, int nk_all, int wm, int wn, int wave, int lane, bool prologue_only,
                      bool stores_in_flight) {
  constexpr bool ROUND = IA_T256W_ROUND == 2 ? !(AKS && BKS) : IA_T256W_ROUND == 1 ? !AKS && !BKS : false;      // the k loop's schedule (below)
  const int li = lane & 31;
  const int gt = wave * 64 + lane;                  // thread index inside the workgroup (0..255)
  // ---- DMA: 8 pieces per operand and k-tile, one lane offset per operand (see t256::main_loop)
  uint32_t voffA, stepA, voffB, stepB;
  if (!AKS) { const int row = gt >> 3; voffA = (uint32_t)(((xa + row) * p.lda + (((gt & 7) ^ ((row >> 1) & 7)) * 8)) * 2); stepA = (uint32_t)(32 * p.lda * 2); }
  else { const int row = gt >> 5; voffA = (uint32_t)((row * p.lda + xa + (((gt & 31) ^ ((row & 3) << 2)) * 8)) * 2); stepA = (uint32_t)(8 * p.lda * 2); }
  if (!BKS) { const int row = gt >> 3; voffB = (uint32_t)(((xb + row) * p.ldb + (((gt & 7) ^ ((row >> 1) & 7)) * 8)) * 2); stepB = (uint32_t)(32 * p.ldb * 2); }
  else { const int row = gt >> 5; voffB = (uint32_t)((row * p.ldb + xb + (((gt & 31) ^ ((row & 3) << 2)) * 8)) * 2); stepB = (uint32_t)(8 * p.ldb * 2); }
  const uint32_t kstepA = AKS ? (uint32_t)(BK * p.lda * 2) : (uint32_t)(BK * 2), kstepB = BKS ? (uint32_t)(BK * p.ldb * 2) : (uint32_t)(BK * 2);
  // the k index of this lane's 16 bytes inside a k-tile: k-strided piece j holds k rows j*8 + (gt>>5), a k-contiguous piece the chunk
  const int klA = AKS ? (gt >> 5) : ((gt & 7) ^ (((gt >> 3) >> 1) & 7)) * 8;
  const int klB = BKS ? (gt >> 5) : ((gt & 7) ^ (((gt >> 3) >> 1) & 7)) * 8;
  char* const my_part = smem + wave * 1024;
  const int dbg = IA_GEMM_DBG_HOOKS ? p.dbg : 0;       // the timing ablations (IA_GEMM_DBG bits 2 / 4 / 16) exist in tools builds only
  const bool dma_on = !(dbg & 2);

  // piece i (0..7: A, 8..15: B) of k-tile u -> buffer u & 1.  Branch-free (a branch next to the accumulator updates makes hipcc copy
  // all 256 of them): lim = number of valid k in this k-tile (0 for a k-tile past the end: the whole piece goes out of range and
  // writes zeros into a buffer nobody reads any more); the address advance per piece and k-tile is a scalar (soffset).
  auto dma_piece = [&](int u, int i) {
    const bool isB = i >= 8;
    const int j = i & 7;
    char* dst = my_part + (isB ? TILE_BYTES : 0) + (u & 1) * 2 * TILE_BYTES + j * 4096;
    const int kt = kt0 + u, kta = (dbg & 4) ? 0 : (isB ? ktaB0 : ktaA0) + u;      // dbg 4: every k-tile re-fetches k-tile 0 (cache-resident)
    const bool ks = isB ? BKS : AKS;
    const uint32_t soff = (uint32_t)kta * (isB ? kstepB : kstepA) + (uint32_t)j * (isB ? stepB : stepA);
    if (ks && IA_T256W_KS_LANEOFF) {
      // k-strided operand: k is the ROW of the tensor, so a piece past K (the tail of the last k-tile, the look-ahead k-tiles behind it)
      // lies behind the end of the buffer window -- provided its whole address sits in the LANE offset (the hardware's range check does
      // not see the scalar offset): one v_add per piece instead of add + compare + select.  (A split-K slab's look-ahead reads the next
      // slab's first rows instead of zeros: nobody consumes that buffer.)  Data gradient +4 %, weight gradient +0.7 %; the memory-side
      // fetches of both went UP 4-8 % with it (FETCH_SIZE 0.996 -> 1.076 GB per weight-gradient launch, same box, IA_T256W_KS_LANEOFF=0
      // against 1): far more than the two look-ahead k-tiles -- the workgroups that share a panel out of one L2 drift further apart.
      const uint32_t off = dma_on ? (isB ? voffB : voffA) + soff : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rsB : rsA, IA_LDS(dst), 16, off, 0, 0, 0);
    } else {
      const int lim = (u < n_tiles && dma_on) ? p.K - kt * BK : 0;
      const uint32_t off = (isB ? klB : klA) + (ks ? j * 8 : 0) < lim ? (isB ? voffB : voffA) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rsB : rsA, IA_LDS(dst), 16, off, (int)soff, 0, 0);
    }
  };

  // ROUND schedule (k-contiguous operands): the lane offset of a k-tile's pieces is the same for all eight pieces of an operand -- whole
  // k-tile, K tail, or out of range past the last k-tile -- and is formed where the schedule has room, not at the first piece
  auto off_of = [&](int u, bool isB) -> uint32_t {
    const int lim = (u < n_tiles && dma_on) ? p.K - (kt0 + u) * BK : 0;
    return (isB ? klB : klA) < lim ? (isB ? voffB : voffA) : OOB;
  };
  auto dma_c = [&](int u, int i, uint32_t off) {
    const bool isB = i >= 8;
    const int j = i & 7;
    char* dst = my_part + (isB ? TILE_BYTES : 0) + (u & 1) * 2 * TILE_BYTES + j * 4096;
    const int kta = (dbg & 4) ? 0 : (isB ? ktaB0 : ktaA0) + u;
    const uint32_t soff = (uint32_t)kta * (isB ? kstepB : kstepA) + (uint32_t)j * (isB ? stepB : stepA);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rsB : rsA, IA_LDS(dst), 16, off, (int)soff, 0, 0);
  };

  // second schedule (a k-strided operand): in-loop pieces of k-tile u+2; the k-strided operand's k-tile advance lives in a running lane offset
  uint32_t runA = voffA + (uint32_t)((dbg & 4) ? 0 : ktaA0 + 2) * kstepA, runB = voffB + (uint32_t)((dbg & 4) ? 0 : ktaB0 + 2) * kstepB;
  auto dma_run = [&](int u, int i) {
    const bool isB = i >= 8;
    if (!(isB ? BKS : AKS) || !IA_T256W_KS_LANEOFF || (dbg & 4)) { dma_piece(u, i); return; }
    const int j = i & 7;
    char* dst = my_part + (isB ? TILE_BYTES : 0) + (u & 1) * 2 * TILE_BYTES + j * 4096;
    const uint32_t off = (isB ? runB : runA) + (uint32_t)j * (isB ? stepB : stepA);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rsB : rsA, IA_LDS(dst), 16, dma_on ? off : OOB, 0, 0, 0);
  };

  // pieces 10 .. 15 (B pieces 2 .. 7) of the k-tile BEFORE the running one (the spread form of the second schedule issues them a trip late)
  auto dma_prev = [&](int u, int i) {
    if (!BKS || !IA_T256W_KS_LANEOFF || (dbg & 4)) { dma_piece(u, i); return; }
    const int j = i & 7;
    char* dst = my_part + TILE_BYTES + (u & 1) * 2 * TILE_BYTES + j * 4096;
    const uint32_t off = runB - kstepB + (uint32_t)j * stepB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, IA_LDS(dst), 16, dma_on ? off : OOB, 0, 0, 0);
  };

  if (prologue_only) {       // called ahead of time (before the previous tile's epilogue): just start the first two k-tiles
#pragma unroll
    for (int i = 0; i < 16; ++i) dma_piece(0, i);
#pragma unroll
    for (int i = 0; i < (ROUND ? 12 : IA_T256W_SPREAD3 ? 10 : 16); ++i) dma_piece(1, i);      // (zero-filled when it does not exist; the rest: see the loops)
    return;
  }
  // the prologue DMA of this tile.  After a full-tile epilogue exactly PEND store instructions were issued behind it and
  // may stay in flight (vmcnt retires in order: at most PEND outstanding <=> every older DMA piece has landed).
  if (stores_in_flight) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PEND < 60 ? PEND : 60) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- fragment addresses
  const int nperm = ((li >> 2) & 1) * 16 + (li >> 3) * 4 + (li & 3);      // B row <-> n so that a lane ends up with 16 consecutive columns
  uint32_t baseA[4], baseB[4];
  const uint32_t smem_addr = ia_lds_addr(smem);
  frag_bases<AKS>(baseA, smem_addr, wm * 128, lane, li);
  frag_bases<BKS>(baseB, smem_addr + TILE_BYTES, wn * 128, lane, nperm);
  constexpr int NRA = AKS ? 8 : 4, NRB = BKS ? 8 : 4, NR = NRA + NRB;      // LDS operations of one fragment set
  constexpr int NR15 = NR > 15 ? 15 : NR;

  Op<AKS> a0, a1, a2, a3;
  Op<BKS> b0, b1, b2, b3;
#pragma unroll
  for (int j = 0; j < 4; ++j) read_frag<0>(a0, j, baseA, 0u);
#pragma unroll
  for (int j = 0; j < 4; ++j) read_frag<0>(b0, j, baseB, 0u);

  // the 16 MFMAs of one k-step; filler(i) is issued right behind MFMA i and pinned there
  // FRESH (the first k-step of a tile under IA_T256W_PEEL): the MFMAs take a zero C operand, so nothing has to clear the accumulators
  auto step = [&](const auto& fa, const auto& fb, auto&& filler, auto FRESH_T) {
    constexpr bool FRESH = decltype(FRESH_T)::value;
    bf16x8 va[4], vb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { va[j] = frag_of(fa, j); vb[j] = frag_of(fb, j); }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
#if IA_T256W_MFMA16      // timing experiment only (results are garbage): two 16x16x32 MFMAs in place of one 32x32x16, same operand registers
        {
          f32x16& c = acc[mi][ni];
          f32x4 c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[ni], va[mi], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[ni], va[mi], c1, 0, 0, 0);
          c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3]; c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
        }
#else
        if constexpr (FRESH) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[ni], va[mi], z, 0, 0, 0);
        } else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[ni], va[mi], acc[mi][ni], 0, 0, 0);
#endif
        filler(mi * 4 + ni);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  using Acc = std::false_type;

  uint32_t bo = 0;
  int u = 0;
  // Two schedules.  ROUND (both operands k-contiguous: the forward GEMMs): every fragment of k-tile u is requested under k-step 0 (24 / 36
  // LDS reads behind 12 MFMAs), the buffer is handed back after a QUARTER of the k-tile, and the 16 DMA pieces of a k-tile go out one
  // per FOUR MFMAs all the way round: k-tile u+2's pieces 0 .. 11 under k-steps 1 .. 3, 12 .. 15 under k-step 0 of the next trip (the
  // tile prologue issues 16 + 12 pieces to match).  The VMEM port takes 60-180 cycles per piece and two MFMAs are 64: one piece per two
  // MFMAs queued up behind it (K = 4096 forward shapes +4.7 %, profiles/r04_gemm_loop_schedules.txt).  The forms with a k-strided operand
  // (transpose reads: 36 / 48 per k-tile) lose 3 % under it (data gradient 1148 -> 1184, weight gradient frac 0.522 -> 0.537 back
  // on the round-2 order below).
  if constexpr (ROUND) {
  uint32_t offA = OOB, offB = off_of(1, true);
  // one trip = one k-tile.  IA_T256W_PEEL: the first trip of a tile is its own copy of the body whose k-step 0 writes the accumulators
  // with a zero C operand -- the 256 v_accvgpr_write per tile and wave that cleared them (in the epilogue: an issue-bound stretch) are
  // gone; the loop behind it runs at least once more (a k-tile past the end reads zeros: K <= 64 pays one empty trip).
  auto trip = [&](auto FIRST_T) {
    const uint32_t bn = bo ^ (uint32_t)(2 * TILE_BYTES);
    tie2<0>(a0, b0);
    step(a0, b0, [&](int i) {      // (no reads behind the last four MFMAs: they cover the latency of the last reads)
      if (i < 4) { read_frag<1>(a1, i, baseA, bo); read_frag<1>(b1, i, baseB, bo); }
      else if (i < 8) { read_frag<2>(a2, i - 4, baseA, bo); read_frag<2>(b2, i - 4, baseB, bo); }
      else if (i < 12) { read_frag<3>(a3, i - 8, baseA, bo); read_frag<3>(b3, i - 8, baseB, bo); }
      if (i % 4 == 3) dma_c(u + 1, 12 + i / 4, offB);      // the last four pieces of k-tile u+1 (into the other buffer)
      if (i == 12) offA = off_of(u + 2, false);
    }, FIRST_T);
    tie6<0>(a1, b1, a2, b2, a3, b3);
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    step(a1, b1, [&](int i) {
      if (i % 4 == 1) dma_c(u + 2, i / 4, offA);
    }, Acc{});
    step(a2, b2, [&](int i) {
      if (i % 4 == 1) dma_c(u + 2, 4 + i / 4, offA);
      if (i == 14) offB = off_of(u + 2, true);
    }, Acc{});
    // k-tile u+1 has landed (its last four pieces went out under k-step 0): only the 8 pieces just issued may be outstanding
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    step(a3, b3, [&](int i) {      // set 0 of k-tile u+1 in the first half: the second half covers the reads' latency
      if (i < 4) read_frag<0>(a0, i, baseA, bn);
      else if (i < 8) read_frag<0>(b0, i - 4, baseB, bn);
      if (i % 4 == 3) dma_c(u + 2, 8 + i / 4, offB);
    }, Acc{});
    bo = bn;
    ++u;
    };
  if constexpr (IA_T256W_PEEL && PEEL_OK) {
    trip(std::true_type{});
    do { trip(Acc{}); } while (u < n_tiles);
  } else {
    do { trip(Acc{}); } while (u < n_tiles);
  }
  // the reads of "set 0 of the k-tile after the last" are dead, but in flight: their destinations must not be handed out before they land
  tie2<0>(a0, b0);
  return;
  }
  do {
    const uint32_t bn = bo ^ (uint32_t)(2 * TILE_BYTES);
    // k-step 0: sets 1 and 2 requested, one fragment behind each MFMA
    tie<0>(a0); tie<0>(b0);
    step(a0, b0, [&](int i) {
      if (i < 4) read_frag<1>(a1, i, baseA, bo);
      else if (i < 8) read_frag<1>(b1, i - 4, baseB, bo);
      else if (i < 12) read_frag<2>(a2, i - 8, baseA, bo);
      else read_frag<2>(b2, i - 12, baseB, bo);
      // spread form: the DMA pieces of a k-tile one per THREE MFMAs over k-steps 2, 3 and the next trip's k-step 0 (5 + 5 + 6; the
      // tile prologue issues 16 + 10): at one per two the VMEM port queues up (see the ROUND schedule)
      if (IA_T256W_SPREAD3 && i % 3 == 0) dma_prev(u + 1, 10 + i / 3);
    }, Acc{});
    // k-step 1: set 3 requested in the first half
    tie<NR15>(a1); tie<NR15>(b1);
    step(a1, b1, [&](int i) {
      if (i < 4) read_frag<3>(a3, i, baseA, bo);
      else if (i < 8) read_frag<3>(b3, i - 4, baseB, bo);
    }, Acc{});
    // every fragment of k-tile u is in registers: once all waves are here its buffer is free for k-tile u+2
    tie<0>(a2); tie<0>(b2); tie<0>(a3); tie<0>(b3);
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // k-step 2: the A half of k-tile u+2, one piece per two MFMAs
    step(a2, b2, [&](int i) {
      if (IA_T256W_SPREAD3) { if (i % 3 == 1) dma_run(u + 2, i / 3); }
      else if (i & 1) dma_run(u + 2, i >> 1);
    }, Acc{});
    // k-step 3: k-tile u+1 has landed (this wave's share: all but the 8 / 5 pieces just issued; everybody's: the barrier)
    if (IA_T256W_SPREAD3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    step(a3, b3, [&](int i) {
      if (IA_T256W_SPREAD3) { if (i % 3 == 2) dma_run(u + 2, 5 + i / 3); }
      else if (i & 1) dma_run(u + 2, 8 + (i >> 1));
      if (!(i & 1)) {
        if (i < 8) read_frag<0>(a0, i >> 1, baseA, bn);
        else read_frag<0>(b0, (i - 8) >> 1, baseB, bn);
      }
    }, Acc{});
    bo = bn;
    ++u;
    // one running lane offset per k-strided operand, opaque to the loop optimiser: left alone it keeps SIXTEEN induction variables
    // (one per piece) and bumps them all in the last MFMA gap of the trip
    if (AKS) { runA += kstepA; asm volatile("" : "+v"(runA)); }
    if (BKS) { runB += kstepB; asm volatile("" : "+v"(runB)); }
  } while (u < n_tiles);
  tie<0>(a0); tie<0>(b0);      // dead, but in flight (see the ROUND loop's exit)
}

// Plain bf16 output (no bias / aux operand): the accumulators are rounded to bf16 BEFORE the trip through LDS, so one pass stages
// 32 rows x 64 columns in 4 KiB (LDS writes run at 64-85 B/clk: they, not the reads or the global stores, are the price of the
// row-major staging) and a lane reads back exactly the 16 bytes it stores.  Same pipelining and accumulator clearing as drain_half.
// WITH_BIAS (k-contiguous B only): bct[ni][q] = the bias of the lane's columns ni*32 + hh*16 + 4q .. +3 (accumulator layout, fetched
// before the main loop), added in fp32 before the rounding.
template <bool BKS, int NH, bool WITH_BIAS>
IA_DEV void drain_half_plain(const GemmArgs& p, f32x16 (&acc)[4][4], int m0, int n0, char* stg, int lane_e, const f32x4 (&bct)[4][4], float ts = 1.f) {
  static_assert(!(WITH_BIAS && BKS), "bias rows are laid out for the k-contiguous B fragment permutation");
  const int hh = lane_e >> 5, li = lane_e & 31, rrow = lane_e >> 3, c8 = lane_e & 7;
  auto stage = [&](int mi) {
    char* const sl = stg + (mi & 1) * 4096;      // two 4-KiB slots per wave: block mi+1 is written while block mi is being read back
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      f32x16& a = acc[mi][NH * 2 + ni];
      if (!BKS) {          // 16 consecutive columns per lane: two 16-byte chunks
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          // accumulators live in AGPRs, the conversions are VALU work: copy eight at a time into VGPRs HERE (opaque to the register
          // allocator, which otherwise moves the whole accumulator array into arch VGPRs for the epilogue and spills the main loop): explicit
          // v_accvgpr_read with the source constrained to an AGPR
          f32x4 t0, t1;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t0[j]) : "a"(a[h8 * 8 + j]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t1[j]) : "a"(a[h8 * 8 + 4 + j]));
          }
          // bct arrives multiplied by ts already (ts = the tile's column scale, 1 outside the q columns: a * 1 + b rounds like a + b)
          if (WITH_BIAS) { t0 = t0 * ts + bct[NH * 2 + ni][h8 * 2]; t1 = t1 * ts + bct[NH * 2 + ni][h8 * 2 + 1]; }
          const bf16x8 v = {f2bf(t0[0]), f2bf(t0[1]), f2bf(t0[2]), f2bf(t0[3]), f2bf(t1[0]), f2bf(t1[1]), f2bf(t1[2]), f2bf(t1[3])};
          const int chunk = (ni * 32 + hh * 16 + h8 * 8) >> 3;
          *reinterpret_cast<bf16x8*>(sl + li * 128 + ((chunk ^ (li & 7)) << 4)) = v;
        }
      } else {             // four runs of 4 columns, 8 apart: 8-byte pieces
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          f32x4 t0;
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t0[j]) : "a"(a[rg * 4 + j]));
          const bf16x4 v = {f2bf(t0[0]), f2bf(t0[1]), f2bf(t0[2]), f2bf(t0[3])};
          const int col = ni * 32 + rg * 8 + hh * 4;
          *reinterpret_cast<bf16x4*>(sl + li * 128 + (((col >> 3) ^ (li & 7)) << 4) + (col & 7) * 2) = v;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;      // dead now: cleared for the next tile in the shadow of the LDS round trip
      __builtin_amdgcn_sched_barrier(0);             // one block at a time: hoisting every accumulator read ran the kernel into scratch
    }
  };
  // Stores through a buffer window over rows m0 .. M-1 of C: a row past M lies behind the window's end and a lane past N starts 2 GiB
  // out, so the hardware drops both -- no execution-mask branches, one 32-bit add per store instead of a 64-bit multiply-add chain
  // (the global_store form of rounds 1-3 spent ~400 scalar / branch / address instructions per wave and tile on clipping that a full
  // tile never needs).  The row advance sits in the LANE offset: the range check does not see a scalar offset.
  const int rows_left = p.M - m0;
  const uint64_t wbytes = rows_left > 0 ? (uint64_t)rows_left * p.ldc * 2 : 0;
  const __amdgpu_buffer_rsrc_t rsC = ia_rsrc(reinterpret_cast<bf16*>(p.C) + (size_t)m0 * p.ldc, (uint32_t)(wbytes < 0x7FFFFFF0ull ? wbytes : 0x7FFFFFF0ull));
  const int ncol = n0 + c8 * 8;
  const uint32_t voffC = ncol < p.N ? (uint32_t)((rrow * p.ldc + ncol) * 2) : 0x80000000u;
  stage(0);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    __builtin_amdgcn_wave_barrier();
    const char* const sl = stg + (mi & 1) * 4096;
    u32x4 v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + rrow;
      v[it] = *reinterpret_cast<const u32x4*>(sl + row * 128 + ((c8 ^ (row & 7)) << 4));
    }
    __builtin_amdgcn_wave_barrier();
    if (mi < 3) stage(mi + 1);        // the other slot: no wait for the reads above (LDS operations of a wave complete in order)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint32_t off = voffC + (uint32_t)((mi * 32 + it * 8) * p.ldc * 2);
      if (!IA_GEMM_DBG_HOOKS || !(p.dbg & 64)) __builtin_amdgcn_raw_buffer_store_b128(v[it], rsC, (int)off, 0, 0);
      else asm volatile("" : : "v"(v[it]));
    }
  }
}

// the T256 epilogue arithmetic (see t256::gemm_kernel) for one 128 x 64 half (NH = 0 / 1) of the wave's 128 x 128 part
template <int EPI, bool OUTF32, bool BKS, int NH>
IA_DEV void drain_half(const GemmArgs& p, f32x16 (&acc)[4][4], int m0, int n0, char* stg, int lane_e, bool full, bool bias_ready, f32x4 bias_lo,
                       f32x4 bias_hi) {
  const int hh = lane_e >> 5, li = lane_e & 31;
  const int rrow = lane_e >> 3, c8 = lane_e & 7;
  constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_ACT || EPI == EPI_BIAS_ADD;
  constexpr bool HAS_AUX = EPI == EPI_ADD || EPI == EPI_BIAS_ADD || EPI == EPI_DGELU || EPI == EPI_DGELU_CS;
  constexpr int AHEAD = 4;
  auto drain = [&](auto PREFETCHED) {
    constexpr bool PRE = decltype(PREFETCHED)::value;
    f32x4 pb0, pb1;
    float cs[8];
    if (EPI == EPI_DGELU_CS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mov_b32 %0, 0" : "=v"(cs[j]));
    }
    bf16x8 ax[AHEAD + 1];
    // full tiles with bf16 outputs: buffer windows over the wave's rows (from row m0 to the end of the tensor), one lane offset each
    constexpr bool BUF = PRE && !OUTF32;
    const auto window = [&](const void* base, int ld) {
      const uint64_t bytes = (uint64_t)(p.M - m0) * ld * 2;
      return ia_rsrc(reinterpret_cast<const bf16*>(base) + (size_t)m0 * ld, (uint32_t)(bytes < 0x7FFFFFF0ull ? bytes : 0x7FFFFFF0ull));
    };
    BufIO io{ia_rsrc(nullptr, 0), ia_rsrc(nullptr, 0), 0u};
    __amdgpu_buffer_rsrc_t rsAux = ia_rsrc(nullptr, 0);
    uint32_t voffC = 0, voffAux = 0;
    if (BUF) {
      io.rsC = window(p.C, p.ldc);
      if (EPI == EPI_BIAS_GELU) io.rsC2 = window(p.C2, p.ldc);
      voffC = (uint32_t)((rrow * p.ldc + n0 + c8 * 8) * 2);
      if (HAS_AUX) { rsAux = window(p.aux, p.ldaux); voffAux = (uint32_t)((rrow * p.ldaux + n0 + c8 * 8) * 2); }
    }
    auto aux_of = [&](int c) {
      const int row_in = (c >> 2) * 32 + ((c >> 1) & 1) * 16 + (c & 1) * 8;
      if (BUF) return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsAux, (int)(voffAux + (uint32_t)(row_in * p.ldaux * 2)), 0, 0));
      return *reinterpret_cast<const bf16x8*>(p.aux + (size_t)(m0 + row_in + rrow) * p.ldaux + n0 + c8 * 8);
    };
    if (PRE && HAS_BIAS) {      // normally fetched before the main loop (a load issued here queues behind the previous stores)
      if (bias_ready) { pb0 = bias_lo; pb1 = bias_hi; }
      else { pb0 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c8 * 8); pb1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c8 * 8 + 4); }
    }
    if (PRE && HAS_AUX) {
#pragma unroll
      for (int c = 0; c < AHEAD; ++c) ax[c] = aux_of(c);
    }
    // One pass = one 32-row block (mi): all 64 lanes stage their 8 quads (32 rows x 64 columns fp32 = 8 KiB, the wave's two 4-KiB
    // slots: with four waves the eight slots of the block are two per wave), then the wave reads them back as rows.  The wave is
    // alone on its SIMD, so the passes are pipelined by hand: the quads of block mi+1 are written as soon as the rows of block mi
    // have been read, and the math / stores of block mi run while that write is in flight.
    auto stage = [&](int mi) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int chunk = (ni * 32 + (BKS ? rg * 8 + hh * 4 : hh * 16 + rg * 4)) >> 2;
          f32x16& a = acc[mi][NH * 2 + ni];
          const f32x4 v = {a[rg * 4], a[rg * 4 + 1], a[rg * 4 + 2], a[rg * 4 + 3]};
          *reinterpret_cast<f32x4*>(stg + li * 256 + ((chunk ^ (li & 15)) << 4)) = v;
          // the block is dead now: clear it for the next tile here, in the shadow of the LDS round trip (256 accumulator writes in
          // front of the next main loop were ~0.5 us per tile)
          a[rg * 4] = 0.f; a[rg * 4 + 1] = 0.f; a[rg * 4 + 2] = 0.f; a[rg * 4 + 3] = 0.f;
        }
    };
    stage(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      __builtin_amdgcn_wave_barrier();
      f32x4 lo[4], hi[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rrow;
        lo[it] = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
        hi[it] = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
      __builtin_amdgcn_wave_barrier();
      if (mi < 3) stage(mi + 1);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rrow;
        const int m = m0 + mi * 32 + row, n = n0 + c8 * 8;
        const int c = mi * 4 + it;
        if (PRE) {
          if (HAS_AUX && c + AHEAD < 16) {
            ax[(c + AHEAD) % (AHEAD + 1)] = aux_of(c + AHEAD);
            asm volatile("" ::: "memory");
          }
          io.off = voffC + (uint32_t)((mi * 32 + it * 8) * p.ldc * 2);
          if (!(IA_DBG(p) & 64)) epi_store8<EPI, OUTF32, true, BUF>(p, m, n, lo[it], hi[it], pb0, pb1, ax[c % (AHEAD + 1)], cs, &io);
        } else {
          if (m < p.M && n < p.N && !(IA_DBG(p) & 64)) epi_store8<EPI, OUTF32, false>(p, m, n, lo[it], hi[it], pb0, pb1, ax[0], cs);
        }
        if (IA_DBG(p) & 64) asm volatile("" : : "v"(lo[it]), "v"(hi[it]));
      }
    }
    if (EPI == EPI_DGELU_CS) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float v = cs[r];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
        v = ia_add_xor32(ia_add_xor16(v));
        cs[r] = v;
      }
      if (rrow == 0 && n0 + c8 * 8 < p.N) {
        float* dst = p.csum_part + (size_t)(m0 >> 7) * p.N + n0 + c8 * 8;
        gstore16(dst, f32x4{cs[0], cs[1], cs[2], cs[3]});
        gstore16(dst + 4, f32x4{cs[4], cs[5], cs[6], cs[7]});
      }
    }
  };
  if (full && (HAS_BIAS || HAS_AUX) && !(IA_DBG(p) & 256)) drain(std::true_type{}); else drain(std::false_type{});
}

template <bool AKS, bool BKS, int EPI, bool OUTF32>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves, 128 x 128 each, one per SIMD
  const int nk_all = (p.K + BK - 1) / BK;
  const int total_tiles = p.tiles_m * p.tiles_n;
  int first_tile = blockIdx.x;
  bool ordered = false;
  p.split_id = 0;
  if (p.splits > 1) {                  // 1-D grid of tiles x splits, XCD-aware with the split outermost (see t256::gemm_kernel)
    const int w = xcd_chunk(blockIdx.x, gridDim.x);
    p.split_id = w / total_tiles;
    first_tile = w % total_tiles;
    ordered = true;
  }
  const int kt0 = p.split_id * p.nk_per_split;
  const int n_tiles = min(nk_all, kt0 + p.nk_per_split) - kt0;
  constexpr int PEND = 2 * (16 * epi_stores_per_call<EPI, OUTF32>() + epi_extra_stores<EPI>());   // store instructions of one full-tile epilogue, per wave
  auto coords = [&](int tile, int& bm, int& bn) {
    if (ordered) tile_of_order(p, tile, bm, bn); else tile_of_index(p, tile, total_tiles, bm, bn);
  };
  auto run = [&](int tile, bool prologue_only, f32x16 (&acc)[4][4], bool stores_in_flight) {
    int bm, bn;
    coords(tile, bm, bn);
    int lane = lane0;
    asm volatile("" : "+v"(lane));      // keep per-lane address arithmetic from being hoisted across the tile loop
    const uint64_t oa = (uint64_t)(AKS ? kt0 * BK : bm * BM) * p.lda, ob = (uint64_t)(BKS ? kt0 * BK : bn * BN) * p.ldb;
    // (the bf16 bias form keeps 64 bias values in registers across the main loop -- 468 of 512: the peeled first trip spilled there)
    constexpr bool PEEL_OK = !(EPI == EPI_BIAS && !OUTF32 && !BKS);
    main_loop<AKS, BKS, PEND, PEEL_OK>(p, smem, acc, rsrc_at(p.A, p.a_bytes, oa), rsrc_at(p.B, p.b_bytes, ob), AKS ? bm * BM : 0, BKS ? bn * BN : 0, kt0,
                              AKS ? 0 : kt0, BKS ? 0 : kt0, n_tiles, nk_all, wm, wn, wave, lane, prologue_only, stores_in_flight);
  };

  // ---- Dynamic tile claim (persistent launches).  With the static order every workgroup owns the tiles bid, bid + gridDim.x, ...: a
  // workgroup that cannot start -- its CU is held by another stream's kernel (an RCCL all-reduce next to the backward GEMMs: this
  // kernel takes all 160 KiB of LDS, nothing co-resides) -- keeps its 1/256 share hostage until a sibling has finished ALL of its
  // own tiles, i.e. the launch takes two rounds.  Claimed tiles instead: every XCD's contiguous run of the work order (the same runs
  // as before, so the panels still share that XCD's L2) is handed out position by position through one counter per XCD; a workgroup
  // whose own run is exhausted takes from the other XCDs' runs; one that starts late finds nothing left and exits.  Claims run two
  // tiles ahead: issued (wave 0, one lane) in front of a tile's epilogue, read behind the NEXT tile's main loop; an outstanding claim
  // is older than every DMA piece the loops wait for, so their counted vmcnt waits only get stricter by it.
  const bool dyn = p.tile_ctr != nullptr && !ordered;
  const int xcd = blockIdx.x & 7, q8 = total_tiles >> 3, r8 = total_tiles & 7;
  auto run_start = [&](int x) { return x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8; };
  auto run_len = [&](int x) { return q8 + (x < r8 ? 1 : 0); };
  int* const mailbox = reinterpret_cast<int*>(smem + 2 * 2 * TILE_BYTES);     // wave 0's staging slot, idle outside the epilogue
  uint32_t claimed = 0;
  // (inline asm: handed the builtin, LLVM's atomic optimizer folds the active lanes into one add and consumes the result -- with an
  // s_waitcnt vmcnt(0) -- on the spot, i.e. in front of the main loop, where the previous tile's stores are still draining.  The
  // returned value stays in `claimed` across the main loop; tools/lint_asm_waits.py checks the ISA for any use of that register
  // ahead of the vmcnt(0) in claim_resolve.)
  auto claim_issue = [&]() {
    if (wave == 0 && lane0 == 0)
      // s_nop 4: the counter address may just have been rebuilt by VALU instructions (v_readlane of a spilled SGPR pair) -- a VALU
      // write of an SGPR needs 5 wait states before a VMEM instruction reads it, and the compiler's hazard recogniser does not look
      // inside inline asm (found the hard way: without it the first build's atomic went to a wild address)
      // (the address through readfirstlane: where the compiler keeps the uniform pointer in a VGPR it cannot hand it to an "s" operand)
      {
        const uint64_t a64 = (uint64_t)(p.tile_ctr + xcd);
        const uint32_t alo = __builtin_amdgcn_readfirstlane((uint32_t)a64), ahi = __builtin_amdgcn_readfirstlane((uint32_t)(a64 >> 32));
        const uint64_t addr = ((uint64_t)ahi << 32) | alo;
        asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(claimed) : "v"(0u), "v"(1u), "s"(addr) : "memory");
      }
  };
  // -> position in the work order (tile_of_order), or -1: nothing left.  Workgroup-uniform (LDS mailbox between two barriers).
  auto claim_resolve = [&]() -> int {
    // (every wave waits, not only the claimer: behind the main loop nothing but its past-the-end dummy pieces is in flight, and the
    // ISA lint can then see the wait on every path from the atomic)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(claimed));
    if (wave == 0) {
      int pos = (int)__builtin_amdgcn_readfirstlane(claimed), w = -1;
      if (pos < run_len(xcd)) w = run_start(xcd) + pos;
      else {
        // Own run exhausted (the launch's tail): take from the other XCDs' runs.  One snapshot of the eight counters (lanes 0-7, one
        // load each) tells which runs are used up, so the tail -- where all runs end within a tile time of each other -- costs one load
        // round trip instead of seven returning atomics in a row in front of the last tile's epilogue (+19 % on a 113-us launch).
        uint32_t seen = 0xFFFFFFFFu;
        if (lane0 < 8) seen = __hip_atomic_load(p.tile_ctr + lane0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t mask = (uint32_t)__ballot((int)seen >= q8 + (lane0 < r8 ? 1 : 0)) | (1u << xcd);
        for (int d = 1; d < 8 && w < 0; ++d) {
          const int x = (xcd + d) & 7;
          if ((mask >> x) & 1u) continue;
          uint32_t got = 0;
          if (lane0 == 0) got = __hip_atomic_fetch_add(p.tile_ctr + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pos = (int)__builtin_amdgcn_readfirstlane(got);
          if (pos < run_len(x)) w = run_start(x) + pos;
        }
      }
      if (lane0 == 0) *reinterpret_cast<volatile int*>(mailbox) = w;
    }
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(mailbox));
    __syncthreads();                                                  // wave 0's epilogue writes its staging slot next
    return w;
  };
  auto leave = [&]() {                                                // the last workgroup out re-arms the slot for its next launch
    // every claim of this workgroup has been PERFORMED before it counts itself out (the last one's value may be unused)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(claimed) : : "memory");
    if (dyn && wave == 0 && lane0 == 0) {
      const uint32_t d = __hip_atomic_fetch_add(p.tile_ctr + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == gridDim.x - 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) __hip_atomic_store(p.tile_ctr + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if (dyn) {
    ordered = true;                                                   // `tile` below is a position of the work order
    claim_issue();
    first_tile = claim_resolve();
    if (first_tile < 0) { leave(); return; }
    claim_issue();                                                    // the second tile's claim travels under the first main loop
  }

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;      // later tiles: the epilogue clears every block it has staged
  int tile = first_tile;
  run(tile, true, acc, false);
  bool stores_in_flight = false;
  constexpr bool HAS_BIAS_K = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_ACT || EPI == EPI_BIAS_ADD;
  while (true) {
    int bm, bn;
    coords(tile, bm, bn);
    const int m0_pre = bm * BM + wm * 128, n0_pre = bn * BN + wn * 128;
    const bool full = m0_pre + 128 <= p.M && n0_pre + 128 <= p.N;      // both halves inside C: the store count of the tile is exact
    // the bias of this tile's columns is fetched BEFORE the main loop: VMEM retires in order, so a load issued in the epilogue
    // could only be awaited by draining the stores in front of it (the bias epilogue cost 27 us more than the plain one at
    // 32640 x 4096 x 1024).  Waited for right behind the main loop, where nothing slow is in flight yet.
    // Plain bias + bf16 output: in ACCUMULATOR layout (16 consecutive columns per lane and 32-column block: 64 floats, the wave has
    // 140 spare VGPRs), so the sum is rounded before the LDS trip (drain_half_plain); the other bias epilogues: row layout.
    constexpr bool BIAS_CT = EPI == EPI_BIAS && !OUTF32 && !BKS;
    f32x4 pbv[4] = {};
    f32x4 bct[4][4] = {};
    int lane_b = lane0;
    asm volatile("" : "+v"(lane_b));
    const bool bias_pre = HAS_BIAS_K && full;
    if (bias_pre && BIAS_CT) {
      const float* bp = p.bias + n0_pre + (lane_b >> 5) * 16;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) bct[ni][q] = *reinterpret_cast<const f32x4*>(bp + ni * 32 + q * 4);
    } else if (bias_pre) {
      const float* bp = p.bias + n0_pre + (lane_b & 7) * 8;
      pbv[0] = *reinterpret_cast<const f32x4*>(bp); pbv[1] = *reinterpret_cast<const f32x4*>(bp + 4);
      pbv[2] = *reinterpret_cast<const f32x4*>(bp + 64); pbv[3] = *reinterpret_cast<const f32x4*>(bp + 68);
    }
    run(tile, false, acc, stores_in_flight);
    float ts = 1.f;
    if (bias_pre && BIAS_CT) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bct[ni][0]), "+v"(bct[ni][1]), "+v"(bct[ni][2]), "+v"(bct[ni][3]));
      if (n0_pre < p.qcols) {                  // wave-uniform: the 128 columns of this wave lie inside the scaled (q) columns
        ts = p.qscale;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q) bct[ni][q] *= ts;
      }
    } else if (bias_pre) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pbv[0]), "+v"(pbv[1]), "+v"(pbv[2]), "+v"(pbv[3]));
    int next = ordered ? total_tiles : tile + gridDim.x;
    if (dyn) {
      next = claim_resolve();
      // the claim for the tile AFTER next goes out here, in front of next's prologue DMA and this tile's epilogue: VMEM retires in
      // order, so a claim issued right in front of a main loop holds that loop's first counted wait up for the atomic's round trip
      // (1-3 us under load: +14 % on 25-us tiles at K = 1024, N = 1024); here it has the whole epilogue to come back
      // (issued also when nothing is left -- written as `else claim_issue()` the backend fails with "illegal VGPR to SGPR copy";
      // the stray claim only overshoots a counter, and leave() waits for it before it reports this workgroup done)
      claim_issue();
      if (next < 0) next = total_tiles;
    }
    if (next < total_tiles) run(next, true, acc, false);      // the next tile's first two k-tiles travel under this epilogue

    int lane_e = lane0;
    asm volatile("" : "+v"(lane_e));
    // (the tile origin is known before the main loop now: keep the epilogue's address arithmetic from being scheduled in front of
    // the loop, where it would stay live across it and spill)
    int m0 = m0_pre, n0 = n0_pre;
    asm volatile("" : "+s"(m0), "+s"(n0));
    char* stg = smem + 2 * 2 * TILE_BYTES + wave * 2 * STAGE_BYTES;      // 8 KiB per wave (t256's eight 4-KiB slots, two per wave)
    if (!(IA_DBG(p) & 32)) {
      if constexpr (EPI == EPI_NONE && !OUTF32) {
        drain_half_plain<BKS, 0, false>(p, acc, m0, n0, stg, lane_e, bct);
        drain_half_plain<BKS, 1, false>(p, acc, m0, n0 + 64, stg, lane_e, bct);
      } else if (BIAS_CT && bias_pre) {
        drain_half_plain<false, 0, BIAS_CT>(p, acc, m0, n0, stg, lane_e, bct, ts);
        drain_half_plain<false, 1, BIAS_CT>(p, acc, m0, n0 + 64, stg, lane_e, bct, ts);
      } else {
        drain_half<EPI, OUTF32, BKS, 0>(p, acc, m0, n0, stg, lane_e, full || (m0 + 128 <= p.M && n0 + 64 <= p.N), bias_pre, pbv[0], pbv[1]);
        drain_half<EPI, OUTF32, BKS, 1>(p, acc, m0, n0 + 64, stg, lane_e, full, bias_pre, pbv[2], pbv[3]);
      }
    }
    if (next >= total_tiles) break;
    stores_in_flight = !(IA_DBG(p) & 96) && full;
    if (!stores_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile = next;
  }
  leave();
}
}  // namespace t256w



// ============================================================================== T256LA (round 6): T256W with a k pipeline that never drains
// Plain NT form only (k-contiguous A and B, bf16 output, no bias / aux operand), static tile order, one workgroup per CU.  T256W ends
// every tile with an empty pipeline: the last two trips of its k loop fetch past the end, the next tile starts with a burst of 28 DMA
// pieces (~41 cycles of blocked issue each), a wait, a workgroup barrier and the exposed latency of its first fragment reads.  Here the
// last two trips fetch the NEXT tile's k-tiles 0 and 1 (a second descriptor pair, chosen by scalar selects), so when a tile's last trip
// ends the next tile's k-tile 0 sits in LDS, its first fragment set is already requested (what T256W reads as "dead" look-ahead), and
// k-tile 1 is in flight: the drain runs, and the next tile's first trip starts where a steady-state trip would -- no prologue burst, no
// prologue wait, no barrier, no exposed read.  The loop-carried pipeline state is the fragment set a0 / b0 (32 VGPRs, live across the
// drain: the plain kernel has ~100 to spare there) and the LDS buffer parity.  One tile boundary = drain only.
#ifndef IA_T256LA
#define IA_T256LA 1
#endif
namespace t256la {
using namespace t256w;

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nk_all = (p.K + BK - 1) / BK;             // >= 2 (the host sends shorter K to T256W)
  const int total_tiles = p.tiles_m * p.tiles_n;
  const int n_tiles = nk_all;
  auto coords = [&](int tile, int& bm, int& bn) { tile_of_index(p, tile, total_tiles, bm, bn); };
  // a tile's operand windows; a tile index past the end gives empty windows: every piece of it reads zeros
  auto window_a = [&](int tile) {
    int bm = 0, bn = 0;
    if (tile < total_tiles) coords(tile, bm, bn);
    return tile < total_tiles ? rsrc_at(p.A, p.a_bytes, (uint64_t)bm * BM * p.lda) : ia_rsrc(p.A, 0u);
  };
  auto window_b = [&](int tile) {
    int bm = 0, bn = 0;
    if (tile < total_tiles) coords(tile, bm, bn);
    return tile < total_tiles ? rsrc_at(p.B, p.b_bytes, (uint64_t)bn * BN * p.ldb) : ia_rsrc(p.B, 0u);
  };

  int lane = lane0;
  asm volatile("" : "+v"(lane));
  const int li = lane & 31;
  const int gt = wave * 64 + lane;
  // ---- DMA lane constants (t256w::main_loop): 8 pieces per operand and k-tile, one lane offset per operand
  const int rowd = gt >> 3;
  const uint32_t voffA = (uint32_t)((rowd * p.lda + (((gt & 7) ^ ((rowd >> 1) & 7)) * 8)) * 2), stepA = (uint32_t)(32 * p.lda * 2);
  const uint32_t voffB = (uint32_t)((rowd * p.ldb + (((gt & 7) ^ ((rowd >> 1) & 7)) * 8)) * 2), stepB = (uint32_t)(32 * p.ldb * 2);
  const int kl = ((gt & 7) ^ (((gt >> 3) >> 1) & 7)) * 8;          // the k index of this lane's 16 bytes inside a k-tile
  char* const my_part = smem + wave * 1024;
  // ---- fragment addresses
  const int nperm = ((li >> 2) & 1) * 16 + (li >> 3) * 4 + (li & 3);
  uint32_t baseA[4], baseB[4];
  const uint32_t smem_addr = ia_lds_addr(smem);
  frag_bases<false>(baseA, smem_addr, wm * 128, lane, li);
  frag_bases<false>(baseB, smem_addr + TILE_BYTES, wn * 128, lane, nperm);

  int tile = blockIdx.x;
  __amdgpu_buffer_rsrc_t rsA = window_a(tile), rsB = window_b(tile);
  __amdgpu_buffer_rsrc_t rsAn = window_a(tile + (int)gridDim.x), rsBn = window_b(tile + (int)gridDim.x);

  // k-tile v of the CURRENT tile (v >= n_tiles: k-tile v - n_tiles of the NEXT tile) -> lane offset of its pieces, or out of range
  auto off_of = [&](int v) -> uint32_t {
    const int kt = v >= n_tiles ? v - n_tiles : v;
    return kl < p.K - kt * BK ? 1u : 0u;                 // (1 = in range; the operand's own lane offset is selected by the caller)
  };
  // piece i (0..7: A, 8..15: B) of k-tile v into the buffer at byte offset `buf`
  auto dma = [&](int v, int i, uint32_t ok, uint32_t buf) {
    const bool isB = i >= 8, nx = v >= n_tiles;
    const int j = i & 7, kt = nx ? v - n_tiles : v;
    char* dst = my_part + (isB ? TILE_BYTES : 0) + buf + j * 4096;
    const uint32_t soff = (uint32_t)kt * (uint32_t)(BK * 2) + (uint32_t)j * (isB ? stepB : stepA);
    const uint32_t off = ok ? (isB ? voffB : voffA) : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? (nx ? rsBn : rsB) : (nx ? rsAn : rsA), IA_LDS(dst), 16, off, (int)soff, 0, 0);
  };

  // ---- the first tile's prologue: k-tiles 0 and 1 whole, landed (once per launch), first set.  Every tile's first trip then finds the
  // same state: k-tile 1 requested in full BEFORE anything else that is in flight (below).
#pragma unroll
  for (int i = 0; i < 16; ++i) dma(0, i, off_of(0), 0u);
#pragma unroll
  for (int i = 0; i < 16; ++i) dma(1, i, off_of(1), (uint32_t)(2 * TILE_BYTES));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  Op<false> a0, a1, a2, a3, b0, b1, b2, b3;
#pragma unroll
  for (int j = 0; j < 4; ++j) read_frag<0>(a0, j, baseA, 0u);
#pragma unroll
  for (int j = 0; j < 4; ++j) read_frag<0>(b0, j, baseB, 0u);
  uint32_t bo = 0;                                       // LDS byte offset of the buffer that holds the coming trip's k-tile

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto step = [&](const Op<false>& fa, const Op<false>& fb, auto&& filler, auto FRESH_T) {
    constexpr bool FRESH = decltype(FRESH_T)::value;
    bf16x8 va[4], vb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { va[j] = frag_of(fa, j); vb[j] = frag_of(fb, j); }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        if constexpr (FRESH) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[ni], va[mi], z, 0, 0, 0);
        } else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[ni], va[mi], acc[mi][ni], 0, 0, 0);
        filler(mi * 4 + ni);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  using Acc = std::false_type;

  const f32x4 no_bias[4][4] = {};
  while (true) {
    int bm, bn;
    coords(tile, bm, bn);
    const int m0_pre = bm * BM + wm * 128, n0_pre = bn * BN + wn * 128;
    int u = 0;
    uint32_t okA = 0u, okB = off_of(1);
    // one trip = one k-tile (t256w::main_loop, ROUND schedule, with the look-ahead reaching into the next tile)
    // A tile's FIRST trip: the last four pieces of its k-tile 1 went out in front of the previous tile's drain (or in the launch
    // prologue), so the counted wait of this trip -- "k-tile 1 has landed" -- can leave the drain's 32 output stores in flight with the 8
    // pieces it has just issued: vmcnt(40).  VMEM retires in order: with those four pieces issued BEHIND the stores (T256W's order) the
    // wait was for the stores' completion, i.e. for the whole drain of 128 KiB at the ~18 B per clock a CU's stores sustain.
    auto trip = [&](auto FIRST_T) {
      constexpr bool FIRST = decltype(FIRST_T)::value;
      const uint32_t bn_ = bo ^ (uint32_t)(2 * TILE_BYTES);
      tie2<0>(a0, b0);
      step(a0, b0, [&](int i) {
        if (i < 4) { read_frag<1>(a1, i, baseA, bo); read_frag<1>(b1, i, baseB, bo); }
        else if (i < 8) { read_frag<2>(a2, i - 4, baseA, bo); read_frag<2>(b2, i - 4, baseB, bo); }
        else if (i < 12) { read_frag<3>(a3, i - 8, baseA, bo); read_frag<3>(b3, i - 8, baseB, bo); }
        if (!FIRST && i % 4 == 3) dma(u + 1, 12 + i / 4, okB, bn_);       // the last four pieces of k-tile u+1 (into the other buffer)
        if (i == 12) okA = off_of(u + 2);
      }, FIRST_T);
      tie6<0>(a1, b1, a2, b2, a3, b3);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      step(a1, b1, [&](int i) {
        if (i % 4 == 1) dma(u + 2, i / 4, okA, bo);
      }, Acc{});
      step(a2, b2, [&](int i) {
        if (i % 4 == 1) dma(u + 2, 4 + i / 4, okA, bo);
        if (i == 14) okB = off_of(u + 2);
      }, Acc{});
      if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");      // 8 pieces + the previous tile's 32 stores
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      step(a3, b3, [&](int i) {
        if (i < 4) read_frag<0>(a0, i, baseA, bn_);
        else if (i < 8) read_frag<0>(b0, i - 4, baseB, bn_);
        if (i % 4 == 3) dma(u + 2, 8 + i / 4, okB, bo);
      }, Acc{});
      bo = bn_;
      ++u;
    };
    trip(std::true_type{});
    do { trip(Acc{}); } while (u < n_tiles);
    // the next tile's k-tile 1: its last four pieces in front of the drain's stores (see trip)
    {
      const uint32_t ok1 = off_of(n_tiles + 1), b1_ = bo ^ (uint32_t)(2 * TILE_BYTES);
#pragma unroll
      for (int i = 12; i < 16; ++i) dma(n_tiles + 1, i, ok1, b1_);
    }

    // ---- drain (t256w's plain epilogue); the pipeline stays loaded: a0 / b0 requested, k-tile 1 of the next tile in flight
    int lane_e = lane0;
    asm volatile("" : "+v"(lane_e));
    int m0 = m0_pre, n0 = n0_pre;
    asm volatile("" : "+s"(m0), "+s"(n0));
    char* stg = smem + 2 * 2 * TILE_BYTES + wave * 2 * STAGE_BYTES;
    drain_half_plain<false, 0, false>(p, acc, m0, n0, stg, lane_e, no_bias);
    drain_half_plain<false, 1, false>(p, acc, m0, n0 + 64, stg, lane_e, no_bias);
    tile += (int)gridDim.x;
    if (tile >= total_tiles) break;
    rsA = rsAn; rsB = rsBn;
    rsAn = window_a(tile + (int)gridDim.x); rsBn = window_b(tile + (int)gridDim.x);
  }
  tie2<0>(a0, b0);                                        // dead, but in flight
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the look-ahead pieces of the tile that does not exist target this workgroup's LDS
}
}  // namespace t256la

// Optional per-launch timing of ONE kernel instantiation with HIP events on the launch stream
// (bench.py's roofline leg): variant id = AKS*1000 + BKS*100 + EPI*10 + OUTF32.
struct GemmProf {
  bool on = false;
  int variant = -1, n = 0, cap = 0;
  double flops = 0.0, bytes = 0.0;      // algorithmic: 2 M N K, and A + B read once + C written once
  hipEvent_t* ev = nullptr;
};
GemmProf g_prof;

// C[m][n] (+)= sum_s ws[s][m][n]
// out[group * M + m] += sum_s rsum_ws[group][s][m]: workgroup x of a group folds m = 32x .. 32x+31 (workgroups beyond M/32 skip it);
// 8 lanes per m, lane l adds partials l, l+8, ... in order and lane 0 folds the 8 lane sums in order (fixed order -> deterministic)
IA_DEV void fold_row_sums(const float* rsum_ws, float* rsum_out, int M, int splits) {
  __shared__ float rs[256];
  if (!rsum_ws || (int)blockIdx.x * 32 >= M) return;          // uniform per workgroup
  const int lane = threadIdx.x & 7, m = blockIdx.x * 32 + (threadIdx.x >> 3);
  float a = 0.f;
  if (m < M)
    for (int sp = lane; sp < splits; sp += 8) a += rsum_ws[((size_t)blockIdx.y * splits + sp) * M + m];
  rs[threadIdx.x] = a;
  __syncthreads();
  if (lane == 0 && m < M) {
    float sum = rs[threadIdx.x];
#pragma unroll
    for (int l = 1; l < 8; ++l) sum += rs[threadIdx.x + l];
    rsum_out[(size_t)blockIdx.y * M + m] += sum;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N, int ldc,
                                                            int splits, int accumulate, long gc, const float* rsum_ws, float* rsum_out) {
  fold_row_sums(rsum_ws, rsum_out, M, splits);
  ws += (size_t)blockIdx.y * splits * M * N;       // blockIdx.y = channel group of a batched convolution GEMM
  C += (size_t)blockIdx.y * gc;
  const size_t total4 = (size_t)M * N / 4;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (size_t)gridDim.x * 256) {
    const size_t e = t * 4;
    const int m = (int)(e / N), n = (int)(e % N);
    f32x4 acc = *reinterpret_cast<const f32x4*>(ws + e);
    for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(ws + (size_t)s * M * N + e);
    float* c = C + (size_t)m * ldc + n;
    if (accumulate) acc += *reinterpret_cast<const f32x4*>(c);
    *reinterpret_cast<f32x4*>(c) = acc;
  }
}

// the same sum for deep cuts (conv weight gradients: hundreds of partials of a few KB each): 8 lanes share one output quad,
// lane l adds partials l, l+8, ... in order and lane 0 folds the 8 lane sums in order (fixed order -> deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_deep_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N, int ldc,
                                                                 int splits, int accumulate, long gc, const float* rsum_ws, float* rsum_out) {
  __shared__ f32x4 red[256];
  fold_row_sums(rsum_ws, rsum_out, M, splits);
  ws += (size_t)blockIdx.y * splits * M * N;
  C += (size_t)blockIdx.y * gc;
  const size_t total4 = (size_t)M * N / 4;
  const int lane = threadIdx.x & 7;
  const size_t t = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (t < total4)
    for (int sp = lane; sp < splits; sp += 8) acc += *reinterpret_cast<const f32x4*>(ws + (size_t)sp * M * N + t * 4);
  red[threadIdx.x] = acc;
  __syncthreads();
  if (lane == 0 && t < total4) {
    f32x4 sum = red[threadIdx.x];
#pragma unroll
    for (int l = 1; l < 8; ++l) sum += red[threadIdx.x + l];
    const size_t e = t * 4;
    float* c = C + (size_t)(e / N) * ldc + (e % N);
    if (accumulate) sum += *reinterpret_cast<const f32x4*>(c);
    *reinterpret_cast<f32x4*>(c) = sum;
  }
}

struct Plan { bool big; int splits; };

// Tile choice + split-K factor.  Weight-gradient GEMMs have outputs of only a few dozen tiles (1024x1024 ->
// 16 tiles of 256x256) while K = #tokens is tens of thousands, so K is cut until every CU has a workgroup.
Plan make_plan(int M, int N, int K, bool f32_out, int groups = 1) {
  Plan pl;
  const int nk = (K + BK - 1) / BK;
  const long t256n = (long)((M + 255) / 256) * ((N + 255) / 256);
  const int smax = f32_out ? (nk / 8 > 32 ? 32 : (nk / 8 < 1 ? 1 : nk / 8)) : 1;
  pl.big = M >= 256 && N >= 256 && (N & 7) == 0 && t256n * smax >= 160;
  if (groups > 1) pl.big = false;
  const long tiles = pl.big ? t256n : (long)((M + 127) / 128) * ((N + 127) / 128) * groups;
  // Pick the cut that minimises the makespan: ceil(workgroups / resident slots) rounds of (k-tiles per workgroup + a fixed
  // prologue/epilogue cost), plus the fixed-order reduction that reads one fp32 copy of C per split.  Overshooting the
  // slot count by a few workgroups (36 tiles x 8 = 288 on 256 CUs) would cost a whole extra round.
  const long slots = pl.big ? 256 : 512;
  // outputs of one or two tiles under a huge K (convolution weight gradients: 64 x 576 over millions of pixels) are pure
  // streaming reads: cut K until the resident slots are full, the partials are a few KB each
  int smax2 = smax;
  if (f32_out && slots / tiles > smax2) { smax2 = (int)(slots / tiles); if (smax2 > nk / 8) smax2 = nk / 8 < 1 ? 1 : nk / 8; if (smax2 < smax) smax2 = smax; }
  const double reduce_per_split = (double)tiles * (pl.big ? 0.0244 : 0.0061);   // in k-tile times of one workgroup
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= smax2; ++s) {
    const long rounds = (tiles * s + slots - 1) / slots;
    const double cost = (double)rounds * ((nk + s - 1) / s + 6) + (s > 1 ? s * reduce_per_split : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
  }
  pl.splits = best;
  return pl;
}

// Counter slot of the next dynamic-claim launch (nullptr: static order).  File-scope state: `launch` is a template, its function-local
// statics would exist once per instantiation -- and two instantiations running on two streams would share slot 0.
// Host threads may launch concurrently (ctypes releases the GIL): the sequence is atomic, so two launches never share a slot, and the
// counter array's address is kept per device (a __device__ symbol has one instance per device).  A slot comes round again after
// CTR_SLOTS launches of this process; the kernel's last workgroup re-arms it, so a reuse is only unsafe while >= CTR_SLOTS persistent
// GEMM launches are in flight at once -- the step has ~400 of them in total, each queue holds far fewer.
int g_dynamic = -1;
constexpr int MAX_DEVICES = 16;
uint32_t* g_ctr_base[MAX_DEVICES] = {};
std::atomic<unsigned> g_launch_seq{0};
uint32_t* next_ctr_slot() {
  // Default: the static order.  The claims cost the 256-wide GEMMs 0-1.5 % (two workgroup barriers per tile; 14 % on a 113-us launch of
  // four 25-us tiles per workgroup) and buy nothing while no other stream holds CUs; the data-parallel host switches them on when it
  // creates a communicator (dist.init_from_env at world size > 1 -> ia_debug_gemm_dynamic(1)), IA_GEMM_DYNAMIC=0/1 forces either.
  if (g_dynamic < 0) {
    const char* e = getenv("IA_GEMM_DYNAMIC");
    g_dynamic = e ? atoi(e) : 0;
  }
  if (!g_dynamic) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
  uint32_t* base = __atomic_load_n(&g_ctr_base[dev], __ATOMIC_ACQUIRE);
  if (!base) {
    if (hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_tile_ctr)) != hipSuccess || !base) return nullptr;
    __atomic_store_n(&g_ctr_base[dev], base, __ATOMIC_RELEASE);
  }
  return base + (size_t)(g_launch_seq.fetch_add(1u, std::memory_order_relaxed) % CTR_SLOTS) * CTR_WORDS;
}

template <bool AKS, bool BKS, int EPI, bool OUTF32>
int launch(GemmArgs a, bool big, hipStream_t st) {
  constexpr int vid = AKS * 1000 + BKS * 100 + EPI * 10 + (OUTF32 ? 1 : 0);
  const bool rec = g_prof.on && g_prof.variant == vid && g_prof.n < g_prof.cap;
  if (rec) (void)hipEventRecord(g_prof.ev[2 * g_prof.n], st);
  // Two 256 x 256 kernels: t256w (one wave per SIMD, 128 x 128 wave tiles) has the faster k loop, t256 (two waves per SIMD) had the
  // faster epilogue when that is VALU-heavy (GELU forward, x GELU' data gradient) -- until round 4's k loop: since then t256w wins
  // those too (bias + GELU 412 -> 405 us, x GELU' + column sums 317 -> 313 us at 32640 x 4096 x 1024, step +0.65 % same box).
  // IA_GEMM_WIDE=0 runs everything on t256, 3 the round-3 split (heavy epilogues on t256) for A/B runs.
  static int wide = -1;
  if (wide < 0) { const char* e = getenv("IA_GEMM_WIDE"); wide = e ? atoi(e) : 1; }
  constexpr bool heavy_epi = EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_ACT || EPI == EPI_DGELU || EPI == EPI_DGELU_CS;
  if (big && (wide == 1 || wide == 2 || (wide == 3 && !heavy_epi))) {
    a.tiles_m = (a.M + t256::BM - 1) / t256::BM; a.tiles_n = (a.N + t256::BN - 1) / t256::BN;
    static bool attr_set_w = false;
    auto kern = t256w::gemm_kernel<AKS, BKS, EPI, OUTF32>;
    if (!attr_set_w) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, t256::LDS_BYTES) != hipSuccess) return IA_ERR_LAUNCH;
      attr_set_w = true;
    }
    const int ntile = a.tiles_m * a.tiles_n;
    const int gx = a.splits > 1 ? ntile * a.splits : (ntile < 256 ? ntile : 256);
    // persistent launches with more than one tile per workgroup claim their tiles dynamically (IA_GEMM_DYNAMIC=0: the static order)
    a.tile_ctr = (a.splits == 1 && ntile > gx) ? next_ctr_slot() : nullptr;
    // plain NT form, several tiles per workgroup, static order: the look-ahead kernel (t256la; IA_GEMM_LA=0 keeps t256w)
    int la = IA_T256LA;
    { const char* e = getenv("IA_GEMM_LA"); if (e) la = atoi(e); }       // (read per launch: tests and A/B runs switch it in one process)
    if constexpr (!AKS && !BKS && EPI == EPI_NONE && !OUTF32) {
      if (la && a.splits == 1 && ntile > gx && !a.tile_ctr && a.K >= 2 * BK && !(IA_DBG(a))) {
        static bool attr_set_la = false;
        if (!attr_set_la) {
          if (hipFuncSetAttribute((const void*)t256la::gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, t256::LDS_BYTES) != hipSuccess)
            return IA_ERR_LAUNCH;
          attr_set_la = true;
        }
        hipLaunchKernelGGL(t256la::gemm_kernel, dim3(gx), dim3(256), t256::LDS_BYTES, st, a);
        if (rec) {
          (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st);
          g_prof.flops += 2.0 * a.M * a.N * a.K;
          g_prof.bytes += 2.0 * ((double)a.M * a.K + (double)a.N * a.K) + 2.0 * a.M * a.N;
          ++g_prof.n;
        }
        return ia_check_launch();
      }
    }
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256), t256::LDS_BYTES, st, a);
  } else if (big) {
    a.tiles_m = (a.M + t256::BM - 1) / t256::BM; a.tiles_n = (a.N + t256::BN - 1) / t256::BN;
    static bool attr_set = false;
    auto kern = t256::gemm_kernel<AKS, BKS, EPI, OUTF32>;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, t256::LDS_BYTES) != hipSuccess) return IA_ERR_LAUNCH;
      attr_set = true;
    }
    const int ntile = a.tiles_m * a.tiles_n;
    // split-K: one workgroup per (tile, k-slab), 1-D so the kernel can order them XCD-aware with the slab outermost;
    // otherwise persistent over tiles, one workgroup per CU
    const int gx = a.splits > 1 ? ntile * a.splits : (ntile < 256 ? ntile : 256);
    hipLaunchKernelGGL(kern, dim3(gx), dim3(512), t256::LDS_BYTES, st, a);
  } else {
    a.tiles_m = (a.M + t128::BM - 1) / t128::BM; a.tiles_n = (a.N + t128::BN - 1) / t128::BN;
    hipLaunchKernelGGL((t128::gemm_kernel<AKS, BKS, EPI, OUTF32>), dim3(a.tiles_m * a.tiles_n * a.splits * a.groups), dim3(256), 0, st, a);
  }
  if (OUTF32 && a.splits > 16) {
    size_t g = ((size_t)a.M * a.N / 4 + 31) / 32;
    if (a.rsum_out && g < (size_t)(a.M + 31) / 32) g = (size_t)(a.M + 31) / 32;
    hipLaunchKernelGGL(splitk_reduce_deep_kernel, dim3((unsigned)g, a.groups), dim3(256), 0, st, a.ws, (float*)a.C, a.M, a.N, a.ldc, a.splits,
                       a.accumulate, a.gc, a.rsum_out ? a.rsum_ws : nullptr, a.rsum_out);
  } else if (OUTF32 && a.splits > 1) {
    size_t g = ((size_t)a.M * a.N / 4 + 255) / 256; if (g > 4096) g = 4096;
    if (a.rsum_out && g < (size_t)(a.M + 31) / 32) g = (size_t)(a.M + 31) / 32;      // fold_row_sums: one workgroup per 32 rows
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)g, a.groups), dim3(256), 0, st, a.ws, (float*)a.C, a.M, a.N, a.ldc, a.splits, a.accumulate,
                       a.gc, a.rsum_out ? a.rsum_ws : nullptr, a.rsum_out);
  }
  if (rec) {
    (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st);
    g_prof.flops += 2.0 * a.M * a.N * a.K;
    g_prof.bytes += 2.0 * ((double)a.M * a.K + (double)a.N * a.K) + (OUTF32 ? 4.0 : 2.0) * a.M * a.N;
    ++g_prof.n;
  }
  return ia_check_launch();
}

}  // namespace

// workspace an fp32-output GEMM can use for split-K partial sums (0 = none needed)
extern "C" size_t ia_gemm_workspace_bytes(int M, int N, int K, int c_is_f32) {
  if (!c_is_f32 || M <= 0 || N <= 0 || K <= 0) return 0;
  const Plan pl = make_plan(M, N, K, true);
  return pl.splits > 1 ? (size_t)pl.splits * M * (N + 1) * sizeof(float) : 0;      // + the row-sum partials of the weight-gradient form
}

static int gemm_core(const void* A, int a_kstrided, int lda, const void* B, int b_kstrided, int ldb, void* C, int c_is_f32, int ldc, int M,
                     int N, int K, int epilogue, const float* bias, const void* aux, int ldaux, void* C2, int accumulate, void* workspace,
                     size_t workspace_bytes, const IaViewGemm* view, hipStream_t stream, int qcols = 0, float qscale = 1.f);

// workspace of an IA_EPI_DGELU_COLSUM GEMM: one fp32 row of N partial sums per 128-row block of the output (and never less than
// the stand-alone column-sum kernel needs, which small shapes fall back to)
extern "C" size_t ia_gemm_colsum_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  const size_t fused = (size_t)((M + 255) / 256) * 2 * N * sizeof(float), plain = ia_colsum_workspace_bytes(M, N);
  return fused > plain ? fused : plain;
}

extern "C" int ia_gemm_bf16(const void* A, int a_kstrided, int lda, const void* B, int b_kstrided, int ldb,
                            void* C, int c_is_f32, int ldc, int M, int N, int K, int epilogue,
                            const float* bias, const void* aux, int ldaux, void* C2, int accumulate, void* workspace,
                            size_t workspace_bytes, hipStream_t stream) {
  return gemm_core(A, a_kstrided, lda, B, b_kstrided, ldb, C, c_is_f32, ldc, M, N, K, epilogue, bias, aux, ldaux, C2, accumulate, workspace,
                   workspace_bytes, nullptr, stream);
}

// IA_EPI_BIAS GEMM (k-contiguous operands, bf16 output) whose first scaled_cols columns (a multiple of 128) are multiplied by col_scale
// after the bias: the fused QKV projection of the encoder layers, q leaving as q * softmax scale * log2 e (GemmArgs::qcols)
extern "C" int ia_gemm_bf16_qscale(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                                   int scaled_cols, float col_scale, hipStream_t stream) {
  return gemm_core(A, 0, lda, B, 0, ldb, C, 0, ldc, M, N, K, EPI_BIAS, bias, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, stream, scaled_cols,
                   col_scale);
}

// GEMM with shifted operand views and channel groups batched into one launch (see GemmArgs): the building block of the
// patch-matrix-free 3x3 convolution in conv.hip (library-internal, declared in common.h).
int ia_gemm_view(const IaViewGemm& v, hipStream_t stream) {
  if ((v.a_view && (v.a_kstrided || (v.a_view != 1 && v.a_view != -1))) || (v.b_view && (!v.b_kstrided || v.b_view < 1 || v.b_view > 2)) ||
      v.pw <= 2 || (!v.a_view && !v.b_view) || v.groups < 1 || v.lca < 3 || v.lcbk < 3 || v.lcbn < 3)
    return IA_ERR_ARG;
  return gemm_core(v.A, v.a_kstrided, v.lda, v.B, v.b_kstrided, v.ldb, v.C, v.c_is_f32, v.ldc, v.M, v.N, v.K, v.bias ? EPI_BIAS : EPI_NONE,
                   v.bias, nullptr, 0, v.rsum_out, 0, v.workspace, v.workspace_bytes, &v, stream);
}

size_t ia_gemm_view_workspace_bytes(int M, int N, int K, int groups) {
  if (M <= 0 || N <= 0 || K <= 0 || groups <= 0) return 0;
  const Plan pl = make_plan(M, N, K, true, groups);
  return pl.splits > 1 ? (size_t)groups * pl.splits * M * (N + 1) * sizeof(float) : 0;      // + the row-sum partials (bias gradient)
}

static int gemm_core(const void* A, int a_kstrided, int lda, const void* B, int b_kstrided, int ldb, void* C, int c_is_f32, int ldc, int M,
                     int N, int K, int epilogue, const float* bias, const void* aux, int ldaux, void* C2, int accumulate, void* workspace,
                     size_t workspace_bytes, const IaViewGemm* view, hipStream_t stream, int qcols, float qscale) {
  const uint64_t a_window = view ? view->a_window : 0, b_window = view ? view->b_window : 0;
  const int groups = view ? view->groups : 1;
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return IA_ERR_ARG;
  if ((lda & 7) || (ldb & 7) || (ldc & 3) || (N & 3)) return IA_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) return IA_ERR_ARG;
  GemmArgs g;
  g.A = (const bf16*)A; g.B = (const bf16*)B; g.C = C; g.C2 = (bf16*)C2; g.bias = bias; g.aux = (const bf16*)aux;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux; g.accumulate = accumulate;
  if (qcols < 0 || qcols > N || (qcols & 127) || (qcols && (epilogue != EPI_BIAS || c_is_f32 || a_kstrided || b_kstrided))) return IA_ERR_ARG;
  g.qcols = qcols; g.qscale = qscale;
  const uint64_t ab = a_window ? a_window : (a_kstrided ? ((uint64_t)(K - 1) * lda + M) * 2 : ((uint64_t)(M - 1) * lda + K) * 2);
  const uint64_t bb = b_window ? b_window : (b_kstrided ? ((uint64_t)(K - 1) * ldb + N) * 2 : ((uint64_t)(N - 1) * ldb + K) * 2);
  g.csum_part = nullptr; g.split_id = 0; g.tile_ctr = nullptr;
  g.a_view = g.b_view = g.pw = 0; g.lca = g.lcbk = g.lcbn = 6;
  g.groups = 1; g.ga = g.gb = g.gc = g.gbias = 0;
  if (view) {
    g.a_view = view->a_view; g.b_view = view->b_view; g.pw = view->pw; g.lca = view->lca; g.lcbk = view->lcbk; g.lcbn = view->lcbn;
    g.groups = groups; g.ga = view->ga; g.gb = view->gb; g.gc = view->gc; g.gbias = view->gbias;
  }
  if (!a_kstrided && (K & 7)) return IA_ERR_ARG;
  g.a_bytes = ab; g.b_bytes = bb;
  g.tiles_m = g.tiles_n = 0;
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("IA_GEMM_DBG"); dbg = e ? atoi(e) : 0; } g.dbg = dbg; }
  const int nk = (K + BK - 1) / BK;
  Plan pl = make_plan(M, N, K, c_is_f32 != 0, groups);
  g.splits = 1; g.nk_per_split = nk; g.ws = nullptr;
  const bool big = pl.big && !view && !(g.dbg & 128);     // the shifted views and the group batching live in the T128 kernel only (dbg 128: force T128)
  // weight-gradient form with C2: the fp32 row sums of A^T (bias gradient) come out of the same launch (T128 kernel)
  const bool wgrad_form = a_kstrided && b_kstrided && c_is_f32 && epilogue == EPI_NONE;
  g.rsum_out = (wgrad_form && !big) ? (float*)C2 : nullptr;
  g.rsum_ws = nullptr;
  // split-K sums partial products: only the plain epilogue may be cut (a bias would be added once per k-slab -- the fp32 + bias form
  // came out with 2-8 x the bias until round 2's edge-shape test)
  if (c_is_f32 && epilogue == EPI_NONE && pl.splits > 1 && workspace &&
      workspace_bytes >= (size_t)groups * pl.splits * M * (N + (g.rsum_out ? 1 : 0)) * sizeof(float)) {
    g.nk_per_split = (nk + pl.splits - 1) / pl.splits;
    g.splits = (nk + g.nk_per_split - 1) / g.nk_per_split;
    g.ws = (float*)workspace;
    g.rsum_ws = g.ws + (size_t)groups * g.splits * M * N;
  }
  {  // every workgroup addresses its operands through a 32-bit buffer window that starts at its tile / k-slab (rsrc_at): that
     // window -- not the tensor -- has to stay below 2 GiB
    const uint64_t margin = view ? 2 * (uint64_t)(view->pw + 1) : 0, slab = (uint64_t)g.nk_per_split * BK + margin, tile = 256 + margin;
    const uint64_t wa = (a_kstrided ? slab : tile) * (uint64_t)lda * 2 + (a_kstrided ? 0 : (uint64_t)K * 2);
    const uint64_t wb = (b_kstrided ? slab : tile) * (uint64_t)ldb * 2 + (b_kstrided ? 0 : (uint64_t)K * 2);
    if (wa >= 0x7FFFFFF0ull || wb >= 0x7FFFFFF0ull) return IA_ERR_ARG;
  }
  const bool needs_bias = epilogue == EPI_BIAS || epilogue == EPI_BIAS_GELU || epilogue == EPI_BIAS_GELU_ACT || epilogue == EPI_BIAS_ADD;
  const bool needs_aux = epilogue == EPI_ADD || epilogue == EPI_DGELU || epilogue == EPI_BIAS_ADD || epilogue == EPI_DGELU_CS;
  if (needs_bias && !bias) return IA_ERR_ARG;
  if (needs_aux && (!aux || (ldaux & 3))) return IA_ERR_ARG;
  if (epilogue == EPI_BIAS_GELU && !C2) return IA_ERR_ARG;

  if (!a_kstrided && !b_kstrided && !c_is_f32) {
    switch (epilogue) {
      case EPI_NONE: return launch<false, false, EPI_NONE, false>(g, big, stream);
      case EPI_BIAS: return launch<false, false, EPI_BIAS, false>(g, big, stream);
      case EPI_BIAS_GELU: return launch<false, false, EPI_BIAS_GELU, false>(g, big, stream);
      case EPI_BIAS_GELU_ACT: return launch<false, false, EPI_BIAS_GELU_ACT, false>(g, big, stream);
      case EPI_BIAS_ADD: return launch<false, false, EPI_BIAS_ADD, false>(g, big, stream);
      case EPI_ADD: return launch<false, false, EPI_ADD, false>(g, big, stream);
      // (the data-gradient epilogues on a k-contiguous B: the transposed weight shadows of ia_layer_weights::wt_*)
      case EPI_DGELU: return launch<false, false, EPI_DGELU, false>(g, big, stream);
      case EPI_DGELU_CS: {
        if (!C2 || !workspace || workspace_bytes < ia_gemm_colsum_workspace_bytes(M, N) || ldc != N) return IA_ERR_WORKSPACE;
        if (!big) {
          int rc = launch<false, false, EPI_DGELU, false>(g, false, stream);
          return rc ? rc : ia_colsum(C, ldc, M, N, (float*)C2, 1, workspace, workspace_bytes, stream);
        }
        g.csum_part = (float*)workspace;
        int rc = launch<false, false, EPI_DGELU_CS, false>(g, true, stream);
        return rc ? rc : ia_sum_rows_f32((const float*)workspace, ((M + 255) / 256) * 2, N, (float*)C2, 1, stream);
      }
    }
  } else if (!a_kstrided && b_kstrided && !c_is_f32) {
    switch (epilogue) {
      case EPI_NONE: return launch<false, true, EPI_NONE, false>(g, big, stream);
      case EPI_ADD: return launch<false, true, EPI_ADD, false>(g, big, stream);
      case EPI_DGELU: return launch<false, true, EPI_DGELU, false>(g, big, stream);
      case EPI_DGELU_CS: {
        // C2 (fp32 [N]) += column sums of the output: the bias gradient of the Linear in front of the GELU
        if (!C2 || !workspace || workspace_bytes < ia_gemm_colsum_workspace_bytes(M, N) || ldc != N) return IA_ERR_WORKSPACE;
        if (!big) {                                  // small shapes: plain epilogue, then the stand-alone column-sum kernel
          int rc = launch<false, true, EPI_DGELU, false>(g, false, stream);
          return rc ? rc : ia_colsum(C, ldc, M, N, (float*)C2, 1, workspace, workspace_bytes, stream);
        }
        g.csum_part = (float*)workspace;
        int rc = launch<false, true, EPI_DGELU_CS, false>(g, true, stream);
        return rc ? rc : ia_sum_rows_f32((const float*)workspace, ((M + 255) / 256) * 2, N, (float*)C2, 1, stream);
      }
    }
  } else if (a_kstrided && b_kstrided && c_is_f32) {
    if (epilogue == EPI_NONE) {
      int rc = launch<true, true, EPI_NONE, true>(g, big, stream);
      // 256x256 kernel: no spare accumulators for the row sums -> the stand-alone column-sum pass over A (stream-ordered after the
      // split-K reduce, so it may reuse the workspace)
      if (!rc && C2 && big) rc = ia_colsum(A, lda, K, M, (float*)C2, 1, workspace, workspace_bytes, stream);
      return rc;
    }
  } else if (!a_kstrided && !b_kstrided && c_is_f32) {
    if (epilogue == EPI_NONE) return launch<false, false, EPI_NONE, true>(g, big, stream);
    if (epilogue == EPI_BIAS) return launch<false, false, EPI_BIAS, true>(g, big, stream);
  }
  return IA_ERR_UNSUPPORTED;
}

// Start recording HIP events around every launch of GEMM instantiation `variant` (at most max_launches).
extern "C" int ia_prof_begin(int variant, int max_launches) {
  (void)hipGetLastError();
  if (max_launches <= 0) return IA_ERR_ARG;
  if (g_prof.cap < max_launches) {
    for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
    g_prof.ev = new hipEvent_t[2 * max_launches];
    for (int i = 0; i < 2 * max_launches; ++i)
      if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return IA_ERR_LAUNCH;
    g_prof.cap = max_launches;
  }
  g_prof.variant = variant; g_prof.n = 0; g_prof.flops = 0.0; g_prof.bytes = 0.0; g_prof.on = true;
  return IA_OK;
}

// Stop recording; synchronises on the recorded events and returns summed kernel time, FLOPs and launch count.
extern "C" int ia_prof_end(double* total_ms, double* total_flops, int* launches) {
  (void)hipGetLastError();
  g_prof.on = false;
  double ms = 0.0;
  for (int i = 0; i < g_prof.n; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return IA_ERR_LAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return IA_ERR_LAUNCH;
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = g_prof.flops;
  if (launches) *launches = g_prof.n;
  return IA_OK;
}

// Diagnostics: `workgroups` workgroups that each take a whole CU's LDS and spin for `milliseconds` (100 MHz wall clock)
namespace {
__global__ __launch_bounds__(256) void cu_hog_kernel(unsigned long long ticks) {
  extern __shared__ __attribute__((aligned(16))) char hog_smem[];
  if (threadIdx.x == 0) *reinterpret_cast<volatile int*>(hog_smem) = 1;      // the allocation is real
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
// Diagnostics: switch the dynamic tile claim of the persistent GEMM launches on / off at run time (the IA_GEMM_DYNAMIC default otherwise);
// returns the previous setting.
extern "C" int ia_debug_gemm_dynamic(int on) {
  if (g_dynamic < 0) (void)next_ctr_slot();
  const int prev = g_dynamic;
  if (on >= 0) g_dynamic = on ? 1 : 0;        // (negative: query only)
  return prev;
}
extern "C" int ia_debug_cu_hog(int workgroups, float milliseconds, hipStream_t stream) {
  (void)hipGetLastError();
  if (workgroups <= 0 || workgroups > 256 || !(milliseconds > 0.f) || milliseconds > 1000.f) return IA_ERR_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, t256::LDS_BYTES) != hipSuccess) return IA_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(cu_hog_kernel, dim3(workgroups), dim3(256), t256::LDS_BYTES, stream, (unsigned long long)(milliseconds * 1e5f));
  return ia_check_launch();
}

// algorithmic bytes (each operand read once, the output written once) of the launches recorded between ia_prof_begin / ia_prof_end
extern "C" double ia_prof_bytes(void) { return g_prof.bytes; }
