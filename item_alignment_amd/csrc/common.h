// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the item-pair matching engine.
// Hardware facts used here were verified on an MI355X with tools/probe_layouts.hip
// (MFMA fragment maps, ds_read_tr16_b64 gather, buffer->LDS out-of-range zero fill).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define IA_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define IA_DEV __device__ __forceinline__

// ---- LDS transpose read through inline asm.  hipcc (ROCm 7.2) treats __builtin_amdgcn_ds_read_tr16_b64 as possibly
// aliasing a pending LDS-DMA (buffer_load ... lds) and drains vmcnt(0) in front of it, which serialises the next
// tile's fetch with this tile's math.  An asm read is invisible to the compiler's counters: every use must sit behind
// an explicit s_waitcnt lgkmcnt (plus a sched_barrier or a "+v" dependency so the consumer cannot be hoisted above it).
IA_DEV uint32_t ia_lds_addr(const void* p) { return (uint32_t)(uintptr_t)IA_LDS(p); }
template <int OFF>
IA_DEV s16x4 ia_tr_read(uint32_t lds_byte_addr) {
  s16x4 d;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(lds_byte_addr), "n"(OFF));
  return d;
}

// error codes of the C-ABI (ia_strerror in capi.hip)
#define IA_OK 0
#define IA_ERR_ARG (-1)      // bad shape / alignment / null pointer
#define IA_ERR_LAUNCH (-2)   // hipGetLastError after a launch
#define IA_ERR_WORKSPACE (-3)// workspace too small
#define IA_ERR_UNSUPPORTED (-4)

IA_DEV float bf2f(bf16 v) { return (float)v; }
IA_DEV bf16 f2bf(float v) { return (bf16)v; }

// buffer descriptor over [ptr, ptr+bytes): out-of-range lanes read 0 (registers and LDS-DMA alike).
IA_DEV __amdgpu_buffer_rsrc_t ia_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// x[lane] + x[lane ^ 16] and x[lane] + x[lane ^ 32] through v_permlane16_swap / v_permlane32_swap (the instruction exchanges the odd
// rows / upper half of its first register with the even rows / lower half of the second).  Inline asm on purpose: handed the same
// value twice, __builtin_amdgcn_permlane16_swap came back from hipcc (ROCm 7.2) as `v_permlane16_swap v0, v8 ; v_add v0, v0, v0` --
// the second result dropped, i.e. 2 * x[lane ^ 16] instead of the sum (seen in the attention column sums).  s_nop 1 = the two wait
// states between a VALU write of an operand and the swap (guide T21).
IA_DEV float ia_add_xor16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
IA_DEV float ia_add_xor32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// wave-wide reductions (64 lanes) on DPP (no LDS round trips: __shfl_xor lowers to ds_bpermute_b32, ~100 cycles a step):
// four in-row butterfly steps leave every lane with the total of its 16-lane row, the four row totals are then combined
// through scalar registers.  The result is wave-uniform.
template <int CTRL>
IA_DEV float ia_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
IA_DEV float ia_lane(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
IA_DEV float row16_sum(float v) {
  v += ia_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += ia_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += ia_dpp<0x141>(v);   // row_half_mirror
  v += ia_dpp<0x140>(v);   // row_mirror
  return v;
}
IA_DEV float wave_sum(float v) {
  v = row16_sum(v);
  return (ia_lane(v, 0) + ia_lane(v, 16)) + (ia_lane(v, 32) + ia_lane(v, 48));
}
// sums over lanes 0..31 and 32..63 separately
IA_DEV void half_wave_sums(float v, float& lo, float& hi) {
  v = row16_sum(v);
  lo = ia_lane(v, 0) + ia_lane(v, 16);
  hi = ia_lane(v, 32) + ia_lane(v, 48);
}
IA_DEV float wave_max(float v) {
  v = fmaxf(v, ia_dpp<0xB1>(v));
  v = fmaxf(v, ia_dpp<0x4E>(v));
  v = fmaxf(v, ia_dpp<0x141>(v));
  v = fmaxf(v, ia_dpp<0x140>(v));
  return fmaxf(fmaxf(ia_lane(v, 0), ia_lane(v, 16)), fmaxf(ia_lane(v, 32), ia_lane(v, 48)));
}

// Counter-based dropout RNG: one 32-bit mix per (seed, stream, index) -> two 16-bit uniforms.
// keep element iff u16 >= thr16 where thr16 = round(p * 65536).
IA_DEV uint32_t ia_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
IA_DEV uint32_t ia_rng(uint32_t seed, uint32_t stream, uint32_t idx) {
  return ia_mix32(idx ^ ia_mix32(stream ^ (seed * 0x9E3779B9U)));
}

// Row-keyed variant for the attention dropout (element = (row q, column key) of stream (b, h)): one full mix per (stream, row) --
// ia_rng_row, computed once per row -- and per PAIR of columns one xor + multiply + xorshift of the row key against the column
// pair's constant (pair * IA_RNG_PAIR_C, assembled from lane / tile / immediate parts by the kernels so that no integer multiply
// sits in the inner loops): 7 VALU issue slots per 32-bit draw instead of 15.  Statistics checked on 40 x 256 x 256 masks against
// the two-round hash (keep rate, adjacent-key / adjacent-row / in-pair / cross-stream correlations, binomial row sums).
constexpr uint32_t IA_RNG_PAIR_C = 0x9E3779B1U;
IA_DEV uint32_t ia_rng_row(uint32_t seed, uint32_t stream, uint32_t row) { return ia_mix32(row ^ ia_mix32(stream ^ (seed * 0x9E3779B9U))); }
IA_DEV uint32_t ia_rng_pair(uint32_t rowkey, uint32_t pair_c) {      // pair_c = (column >> 1) * IA_RNG_PAIR_C
  const uint32_t h = (rowkey ^ pair_c) * 0x846ca68bU;
  return h ^ (h >> 16);                                               // column & 1 ? high : low 16 bits
}

// erf-GELU (the reference's hidden_act = "gelu": x * Phi(x)).  erf comes from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// far below bf16 resolution) on one v_rcp + one v_exp, because the GELU epilogues run on the VALU in the shadow of no
// MFMA work (libm erff costs ~2x as many instructions); exp(-x^2/2) is shared by Phi and the density phi.
IA_DEV void gelu_parts(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));      // exp(-x^2/2) = exp(-z^2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.3989422804014327f * e;
}
// The same pair for the GEMM epilogues (FFN1 forward: 128 outputs per thread behind every 256 x 256 tile, no MFMA work to hide
// behind), two outputs at a time on the packed fp32 pipe:  Phi(x) ~ sigmoid(x (c0 + c1 u + c2 u^2)), u = min(x^2, U_MAX), the odd
// polynomial fitted to the erf form on [-8, 8] (tools/fit_gelu.py: |gelu - x Phi(x)| <= 3.0e-5, |gelu' error| <= 1.1e-4, a tenth of
// a bf16 ulp at the magnitudes where it matters).  U_MAX = c1 / (2 |c2|) is where the fitted polynomial peaks (|x| = 7.27, the
// sigmoid is 1 - 6e-12 there); beyond it the argument keeps growing linearly instead of following the polynomial back down.
// 12 issue slots per output (6 packed, 1 min, exp2 + rcp, 2 converts) against 21 for the A&S form above.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
IA_DEV void gelu_pair(f32x2_t x, f32x2_t& act, f32x2_t& der) {
  constexpr float L2E = 1.4426950408889634f;
  constexpr float C0 = 1.5949398799788077f, C1 = 0.07403000634661838f, C2 = -0.0007007124749191571f;
  constexpr float U_MAX = C1 / (2.f * 0.0007007124749191571f);
  f32x2_t u = x * x;
  u[0] = __builtin_fminf(u[0], U_MAX); u[1] = __builtin_fminf(u[1], U_MAX);
  const f32x2_t np = (u * (-C2 * L2E) + (-C1 * L2E)) * u + (-C0 * L2E);      // -log2(e) p(u)
  const f32x2_t t = x * np;
  const f32x2_t e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};    // exp(-x p): inf for very negative x -> s = 0
  const f32x2_t den = e + 1.0f;
  const f32x2_t s = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  act = x * s;
  const f32x2_t q = (u * (5.f * C2) + (3.f * C1)) * u + C0;                  // d/dx [x p(x^2)]
  const f32x2_t r = 1.0f - s;
  der = (act * r) * q + s;
}
// the activation alone (forward-only layers): 8 issue slots per output
IA_DEV f32x2_t gelu_act_pair(f32x2_t x) {
  constexpr float L2E = 1.4426950408889634f;
  constexpr float C0 = 1.5949398799788077f, C1 = 0.07403000634661838f, C2 = -0.0007007124749191571f;
  constexpr float U_MAX = C1 / (2.f * 0.0007007124749191571f);
  f32x2_t u = x * x;
  u[0] = __builtin_fminf(u[0], U_MAX); u[1] = __builtin_fminf(u[1], U_MAX);
  const f32x2_t t = x * ((u * (-C2 * L2E) + (-C1 * L2E)) * u + (-C0 * L2E));
  const f32x2_t e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  const f32x2_t den = e + 1.0f;
  return x * f32x2_t{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
}
IA_DEV float gelu_erf(float x) { float c, d; gelu_parts(x, c, d); return x * c; }
IA_DEV float gelu_erf_grad(float x) { float c, d; gelu_parts(x, c, d); return c + x * d; }

// last HIP status seen by a failed launch check (reported by ia_strerror(IA_ERR_LAUNCH))
inline hipError_t g_ia_last_hip_error = hipSuccess;
static inline int ia_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) g_ia_last_hip_error = e;
  return e == hipSuccess ? IA_OK : IA_ERR_LAUNCH;
}

// Library-internal GEMM with shifted operand views + channel groups batched in one launch (gemm.hip; used by the
// patch-matrix-free 3x3 convolution in conv.hip).  Fields as in ia_gemm_bf16; the view fields are described at GemmArgs.
struct IaViewGemm {
  const void* A; int a_kstrided, lda;
  const void* B; int b_kstrided, ldb;
  void* C; int c_is_f32, ldc;
  int M, N, K;
  const float* bias;                 // NULL = no bias
  float* rsum_out;                   // weight-gradient form only: [groups * M] fp32 += sum_k A[k][m] (the bias gradient); NULL = off
  void* workspace; size_t workspace_bytes;
  int a_view, b_view, pw, lca, lcbk, lcbn;
  size_t a_window, b_window;         // bytes addressable from A / B of group 0 (to the end of the tensor)
  int groups; long ga, gb, gc, gbias;   // per-group element strides of A, B, C, bias
};
__attribute__((visibility("hidden"))) int ia_gemm_view(const IaViewGemm& v, hipStream_t stream);
// out[c] (+)= sum_b part[b][c], b < nblk, fixed order (layernorm.hip); the second stage of fused column sums
__attribute__((visibility("hidden"))) int ia_sum_rows_f32(const float* part, int nblk, int N, float* out, int accumulate, hipStream_t stream);
extern "C" size_t ia_colsum_workspace_bytes(int M, int N);
extern "C" int ia_colsum(const void* x, int ld, int M, int N, float* out, int accumulate, void* workspace, size_t workspace_bytes,
                         hipStream_t stream);
__attribute__((visibility("hidden"))) size_t ia_gemm_view_workspace_bytes(int M, int N, int K, int groups);

