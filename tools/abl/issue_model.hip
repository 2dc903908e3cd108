// Issue-model microbenchmark for the attention restructure (round 3): how many VALU instructions of which kind fit
// beside a v_mfma_f32_32x32x16_bf16 stream at 1 / 2 waves per SIMD, and what plain VALU streams cost.  No memory traffic at all.
// One workgroup per CU; WAVES = waves per SIMD (block = 256 * WAVES threads, waves w and w+4 share a SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o issue_model.bin issue_model.hip ; run: ./issue_model.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x, c, d) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define ADD(x, c) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c))
#define MAX3(x, c, d) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
#define CVT(u, c, d) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(c), "v"(d))
#define MULLO(u, c) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(c))
#define XOR(u, c) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u) : "v"(c))
#define PKADD(x2, c2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x2) : "v"(c2))
#define CNDMASK(x, c) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c))
#define CNDMASK_S(x, c, m) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "s"(m))
#define MED3(x, c, d) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
#define BFI(u, c, d) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(u) : "v"(c), "v"(d))
#define AND(u, c) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(c))
#define PERM(u, c, d) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u) : "v"(c), "v"(d))
#define CMPGE(c, d) asm volatile("v_cmp_ge_u32 vcc, %0, %1" : : "v"(c), "v"(d) : "vcc")
#define CMPGE16(c, d) asm volatile("v_cmp_ge_u16 vcc, %0, %1" : : "v"(c), "v"(d) : "vcc")
#define BFE(u, c) asm volatile("v_bfe_u32 %0, %0, %1, 16" : "+v"(u) : "v"(c))
#define ASHR(u) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(u))
#define PKSUBC(u, c) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(u) : "v"(c))
#define PKASHR(u) asm volatile("v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "+v"(u))
#define LSHRXOR(u, c) asm volatile("v_lshrrev_b32 %1, 16, %0\n\tv_xor_b32 %0, %0, %1" : "+v"(u), "=&v"(c))
#define ACCRD(x, a) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a))
#define ACCWR(a, x) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(x))

// FILL: 0 fma, 1 exp, 2 add, 3 cvt_pk, 4 softmax mix (2 exp, 2 add, 1 cvt per 5), 5 mul_lo_u32, 6 max3, 7 xor, 8 pk_add, 9 cndmask
template <int FILL>
__device__ __forceinline__ void filler(float (&x)[16], uint32_t (&u)[8], float c, float d, int i) {
  const int r = i & 15;
  if (FILL == 0) FMA(x[r], c, d);
  if (FILL == 1) EXP(x[r]);
  if (FILL == 2) ADD(x[r], c);
  if (FILL == 3) CVT(u[i & 7], x[r], x[(r + 1) & 15]);
  if (FILL == 4) {
    const int m = i % 5;
    if (m == 0 || m == 2) EXP(x[r]);
    else if (m == 1 || m == 3) ADD(x[r], c);
    else CVT(u[i & 7], x[r], x[(r + 1) & 15]);
  }
  if (FILL == 5) MULLO(u[i & 7], u[(i + 1) & 7]);
  if (FILL == 6) MAX3(x[r], c, d);
  if (FILL == 7) XOR(u[i & 7], u[(i + 1) & 7]);
  if (FILL == 8) { typedef float f2 __attribute__((ext_vector_type(2))); f2 t = {x[r & 14], x[(r & 14) + 1]}; f2 cc = {c, d}; PKADD(t, cc); x[r & 14] = t[0]; x[(r & 14) + 1] = t[1]; }
  if (FILL == 9) CNDMASK(x[r], c);
  if (FILL == 10) { const uint64_t m = 0x00000000FFFFFFFFull; CNDMASK_S(x[r], c, m); }
  if (FILL == 11) MED3(x[r], c, d);
  if (FILL == 12) BFI(u[i & 7], u[(i + 1) & 7], u[(i + 2) & 7]);
  if (FILL == 13) AND(u[i & 7], u[(i + 1) & 7]);
  if (FILL == 14) PERM(u[i & 7], u[(i + 1) & 7], u[(i + 2) & 7]);
  if (FILL == 15) { if (i & 1) CNDMASK(x[r], c); else CMPGE(u[i & 7], u[(i + 1) & 7]); }     // cmp -> vcc -> cndmask pairs
  if (FILL == 16) BFE(u[i & 7], u[(i + 1) & 7]);
  if (FILL == 17) ASHR(u[i & 7]);
  if (FILL == 18) PKSUBC(u[i & 7], u[(i + 1) & 7]);
  if (FILL == 19) PKASHR(u[i & 7]);
  if (FILL == 20) CMPGE(u[i & 7], u[(i + 1) & 7]);
  if (FILL == 21) CMPGE16(u[i & 7], u[(i + 1) & 7]);
  if (FILL == 22) { if (i & 1) CNDMASK(x[r], c); else FMA(x[r], c, d); }     // cndmask alternating with fma
}

// NM MFMAs per iteration (0 = VALU only: then K fillers x 16 per iteration), K fillers behind each MFMA; NACC independent accumulators
template <int NM, int K, int FILL, int NACC, int WAVES, int ROLES>
__global__ __launch_bounds__(256 * WAVES) void kern(const float* in, int iters, long long* cyc, float* out) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float x[16]; uint32_t u[8];
  for (int i = 0; i < 16; ++i) x[i] = in[tid + i * 64] * 1e-3f;
  for (int i = 0; i < 8; ++i) u[i] = (uint32_t)tid * 2654435761u + i;
  const float c = in[tid & 63] * 1e-6f + 1.0f, d = in[(tid + 1) & 63] * 1e-6f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[(tid + i) & 255]; b[i] = (__bf16)in[(tid * 3 + i) & 255]; }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // ROLES: waves 0-3 run the MFMAs only, waves 4-7 the fillers only (two waves per SIMD, different pipes)
  const bool do_m = ROLES == 1 ? wave < 4 : true, do_v = ROLES == 1 ? wave >= 4 : true;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (NM == 0) {
#pragma unroll
      for (int i = 0; i < 16 * K; ++i) filler<FILL>(x, u, c, d, i);
    } else if (!ROLES) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        MFMA(acc[m % NACC], a, b);
#pragma unroll
        for (int i = 0; i < K; ++i) filler<FILL>(x, u, c, d, m * K + i);
      }
    } else if (ROLES == 3 || ROLES == 4) {
      // the attention loop's own shape, free running: every wave alternates a batch of NM MFMAs with NM * K fillers whose inputs are
      // the accumulators (so the batch must have drained); ROLES 4: waves 4-7 start with the filler batch (anti-phase)
      if (ROLES == 4 && wave >= 4 && it == 0) {
#pragma unroll
        for (int i = 0; i < NM * K; ++i) filler<FILL>(x, u, c, d, i);
      }
#pragma unroll
      for (int m = 0; m < NM; ++m) MFMA(acc[m % NACC], a, b);
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[r]) : "v"(acc[r & (NACC - 1)][r]));      // wait for the MFMAs
#pragma unroll
      for (int i = 0; i < NM * K; ++i) filler<FILL>(x, u, c, d, i);
    } else if (ROLES == 2) {
      // ping-pong: waves 0-3 {MFMA phase, barrier, filler phase, barrier}, waves 4-7 the opposite order
      if (wave < 4) {
#pragma unroll
        for (int m = 0; m < NM; ++m) MFMA(acc[m % NACC], a, b);
      } else {
#pragma unroll
        for (int i = 0; i < NM * K; ++i) filler<FILL>(x, u, c, d, i);
      }
      __builtin_amdgcn_s_barrier();
      if (wave >= 4) {
#pragma unroll
        for (int m = 0; m < NM; ++m) MFMA(acc[m % NACC], a, b);
      } else {
#pragma unroll
        for (int i = 0; i < NM * K; ++i) filler<FILL>(x, u, c, d, i);
      }
      __builtin_amdgcn_s_barrier();
    } else {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < NM; ++m) MFMA(acc[m % NACC], a, b);
      }
      if (do_v) {
#pragma unroll
        for (int i = 0; i < NM * K; ++i) filler<FILL>(x, u, c, d, i);
      }
    }
  }
  const long long t1 = clock64();
  if ((tid & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int i = 0; i < 8; ++i) s += (float)u[i];
  for (int j = 0; j < 4; ++j) s += acc[j][0];
  if (s == 12345.678f) out[tid] = s;
}

static float* g_in; static long long* g_cyc; static float* g_out;

template <int NM, int K, int FILL, int NACC, int WAVES, int ROLES>
void run(const char* what) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int pass = 0; pass < 2; ++pass) {
    hipMemset(g_cyc, 0, 256 * 16 * 8);
    hipEventRecord(e0);
    kern<NM, K, FILL, NACC, WAVES, ROLES><<<256, 256 * WAVES>>>(g_in, iters, g_cyc, g_out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> h(256 * 16);
  hipMemcpy(h.data(), g_cyc, 256 * 16 * 8, hipMemcpyDeviceToHost);
  double lo = 0, hi = 0; int nlo = 0, nhi = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 4 * WAVES; ++w) { if (w < 4) { lo += h[b * 16 + w]; ++nlo; } else { hi += h[b * 16 + w]; ++nhi; } }
  lo /= nlo * (double)iters; if (nhi) hi /= nhi * (double)iters;
  const int nmf = NM ? NM : 0, nfl = NM ? NM * K : 16 * K;
  printf("%-44s waves/SIMD %d  %s  cyc/iter w0-3 %8.1f", what, WAVES, ROLES == 1 ? "roles" : ROLES == 2 ? "pingp" : ROLES == 3 ? "batch" : ROLES == 4 ? "batcA" : "     ", lo);
  if (nhi) printf("  w4-7 %8.1f", hi); else printf("               ");
  if (nmf) printf("  | per MFMA %6.1f", lo / nmf);
  if (nfl) printf("  | per filler %6.2f", (nhi && ROLES == 1 ? hi : lo) / nfl);
  printf("  | wall %.3f ms -> %.2f GHz\n", ms, lo * iters / (ms * 1e-3) / 1e9);
}

int main() {
  std::vector<float> h(4096);
  uint32_t s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; }
  hipMalloc(&g_in, 4096 * 4); hipMemcpy(g_in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipMalloc(&g_cyc, 256 * 16 * 8); hipMalloc(&g_out, 4096 * 4);
  printf("== plain VALU streams (16*K instructions per iteration) ==\n");
  run<0, 4, 0, 1, 1, 0>("fma x64"); run<0, 4, 0, 1, 2, 0>("fma x64"); run<0, 4, 0, 1, 4, 0>("fma x64");
  run<0, 4, 1, 1, 1, 0>("exp x64"); run<0, 4, 1, 1, 2, 0>("exp x64"); run<0, 4, 1, 1, 4, 0>("exp x64");
  run<0, 4, 2, 1, 1, 0>("add x64"); run<0, 4, 2, 1, 2, 0>("add x64");
  run<0, 4, 3, 1, 1, 0>("cvt_pk x64"); run<0, 4, 3, 1, 2, 0>("cvt_pk x64");
  run<0, 4, 5, 1, 1, 0>("mul_lo_u32 x64"); run<0, 4, 5, 1, 2, 0>("mul_lo_u32 x64");
  run<0, 4, 6, 1, 1, 0>("max3 x64"); run<0, 4, 6, 1, 2, 0>("max3 x64");
  run<0, 4, 7, 1, 1, 0>("xor x64"); run<0, 4, 7, 1, 2, 0>("xor x64");
  run<0, 4, 8, 1, 1, 0>("pk_add_f32 x64"); run<0, 4, 8, 1, 2, 0>("pk_add_f32 x64");
  run<0, 4, 9, 1, 1, 0>("cndmask x64"); run<0, 4, 9, 1, 2, 0>("cndmask x64");
  run<0, 4, 10, 1, 1, 0>("cndmask sgpr-pair x64"); run<0, 4, 10, 1, 2, 0>("cndmask sgpr-pair x64");
  run<0, 4, 11, 1, 1, 0>("med3 x64"); run<0, 4, 11, 1, 2, 0>("med3 x64");
  run<0, 4, 12, 1, 1, 0>("bfi x64"); run<0, 4, 12, 1, 2, 0>("bfi x64");
  run<0, 4, 13, 1, 1, 0>("and x64"); run<0, 4, 13, 1, 2, 0>("and x64");
  run<0, 4, 14, 1, 1, 0>("perm x64"); run<0, 4, 14, 1, 2, 0>("perm x64");
  run<0, 4, 15, 1, 1, 0>("cmp+cndmask pairs x32"); run<0, 4, 15, 1, 2, 0>("cmp+cndmask pairs x32");
  run<0, 4, 16, 1, 1, 0>("bfe x64"); run<0, 4, 16, 1, 2, 0>("bfe x64");
  run<0, 4, 17, 1, 1, 0>("ashr x64"); run<0, 4, 17, 1, 2, 0>("ashr x64");
  run<0, 4, 18, 1, 1, 0>("pk_sub_i16 clamp x64"); run<0, 4, 18, 1, 2, 0>("pk_sub_i16 clamp x64");
  run<0, 4, 19, 1, 1, 0>("pk_ashrrev_i16 x64"); run<0, 4, 19, 1, 2, 0>("pk_ashrrev_i16 x64");
  run<0, 4, 20, 1, 1, 0>("cmp_ge_u32 x64"); run<0, 4, 20, 1, 2, 0>("cmp_ge_u32 x64");
  run<0, 4, 21, 1, 1, 0>("cmp_ge_u16 x64"); run<0, 4, 21, 1, 2, 0>("cmp_ge_u16 x64");
  run<0, 4, 22, 1, 1, 0>("cndmask/fma alternating x64"); run<0, 4, 22, 1, 2, 0>("cndmask/fma alternating x64");
  run<0, 5, 4, 1, 1, 0>("softmax mix x80"); run<0, 5, 4, 1, 2, 0>("softmax mix x80");
  printf("== MFMA 32x32x16 alone (16 per iteration) ==\n");
  run<16, 0, 0, 1, 1, 0>("mfma, 1 accumulator"); run<16, 0, 0, 2, 1, 0>("mfma, 2 accumulators"); run<16, 0, 0, 4, 1, 0>("mfma, 4 accumulators");
  run<16, 0, 0, 2, 2, 0>("mfma, 2 accumulators"); run<16, 0, 0, 4, 2, 0>("mfma, 4 accumulators");
  printf("== MFMA + K fma fillers per MFMA, 2 accumulators ==\n");
  run<16, 2, 0, 2, 1, 0>("K=2 fma"); run<16, 4, 0, 2, 1, 0>("K=4 fma"); run<16, 5, 0, 2, 1, 0>("K=5 fma"); run<16, 6, 0, 2, 1, 0>("K=6 fma");
  run<16, 7, 0, 2, 1, 0>("K=7 fma"); run<16, 8, 0, 2, 1, 0>("K=8 fma"); run<16, 10, 0, 2, 1, 0>("K=10 fma"); run<16, 12, 0, 2, 1, 0>("K=12 fma");
  run<16, 4, 0, 2, 2, 0>("K=4 fma"); run<16, 5, 0, 2, 2, 0>("K=5 fma"); run<16, 6, 0, 2, 2, 0>("K=6 fma"); run<16, 7, 0, 2, 2, 0>("K=7 fma");
  run<16, 8, 0, 2, 2, 0>("K=8 fma"); run<16, 10, 0, 2, 2, 0>("K=10 fma"); run<16, 12, 0, 2, 2, 0>("K=12 fma");
  printf("== MFMA + K softmax-mix fillers per MFMA (2 exp, 2 add, 1 cvt per 5), 2 accumulators ==\n");
  run<16, 5, 4, 2, 1, 0>("K=5 mix"); run<16, 6, 4, 2, 1, 0>("K=6 mix"); run<16, 7, 4, 2, 1, 0>("K=7 mix"); run<16, 8, 4, 2, 1, 0>("K=8 mix"); run<16, 10, 4, 2, 1, 0>("K=10 mix");
  run<16, 5, 4, 2, 2, 0>("K=5 mix"); run<16, 6, 4, 2, 2, 0>("K=6 mix"); run<16, 7, 4, 2, 2, 0>("K=7 mix"); run<16, 8, 4, 2, 2, 0>("K=8 mix"); run<16, 10, 4, 2, 2, 0>("K=10 mix");
  printf("== roles: waves 0-3 MFMA only, waves 4-7 fillers only (same SIMDs) ==\n");
  run<16, 5, 0, 2, 2, 1>("16 mfma | 80 fma"); run<16, 8, 0, 2, 2, 1>("16 mfma | 128 fma"); run<16, 10, 4, 2, 2, 1>("16 mfma | 160 mix"); run<16, 16, 0, 2, 2, 1>("16 mfma | 256 fma");
  printf("== phase-alternating: 8 MFMAs back to back, then 8*K fillers (what a non-interleaved loop does) ==\n");
  printf("== batches: every wave {16 MFMA, wait, 16*K fillers}, free running (batch) / waves 4-7 starting half a period late (batcA) ==\n");
  run<16, 5, 4, 2, 1, 3>("1 wave: 16 mfma then 80 mix"); run<16, 5, 4, 2, 2, 3>("2 waves: 16 mfma then 80 mix"); run<16, 5, 4, 2, 2, 4>("2 waves anti-phase start");
  run<16, 5, 4, 2, 3, 3>("3 waves: 16 mfma then 80 mix");
  run<16, 8, 4, 2, 2, 3>("2 waves: 16 mfma then 128 mix"); run<16, 8, 4, 2, 2, 4>("2 waves anti-phase: 128 mix");
  run<8, 5, 4, 2, 2, 3>("2 waves: 8 mfma then 40 mix"); run<8, 5, 4, 2, 3, 3>("3 waves: 8 mfma then 40 mix");
  run<16, 5, 0, 2, 2, 2>("16 mfma / 80 fma per wave"); run<16, 8, 0, 2, 2, 2>("16 mfma / 128 fma per wave"); run<16, 5, 4, 2, 2, 2>("16 mfma / 80 mix per wave");
  run<16, 8, 4, 2, 2, 2>("16 mfma / 128 mix per wave"); run<16, 10, 4, 2, 2, 2>("16 mfma / 160 mix per wave");
  return 0;
}
