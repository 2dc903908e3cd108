// Sustained MFMA rate under the power cap as a function of the operand DATA and the MFMA shape (round 2).
// One 256-thread workgroup per CU (one wave per SIMD, like the t256w GEMM), 128 x 128 wave tile held in registers: per "k-step" every
// accumulator block gets one MFMA from 4 (32x32x16) or 8 (16x16x32) A fragments x as many B fragments, exactly the GEMM's register
// traffic, no memory at all.  DATA 0: all operands zero, 1: small integers, 2: normal(0,1) bf16 (what torch.randn feeds the kernels).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_power.bin mfma_power.hip ; run: ./mfma_power.bin [ms per measurement]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const bf16x8* data, int iters, float* out) {
  const int tid = threadIdx.x;
  constexpr int NF = SHAPE == 32 ? 4 : 8;
  bf16x8 a[2][NF], b[2][NF];                               // two k-steps' worth of fragments, alternated
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < NF; ++i) { a[s][i] = data[((s * NF + i) * 2 + 0) * 256 + tid]; b[s][i] = data[((s * NF + i) * 2 + 1) * 256 + tid]; }
  f32x16 acc32[4][4];
  f32x4 acc16[8][8];
  if (SHAPE == 32) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f; }
  else { for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (SHAPE == 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[s][j], a[s][i], acc32[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[s][j], a[s][i], acc16[i][j], 0, 0, 0);
      }
    }
  }
  float sum = 0.f;
  if (SHAPE == 32) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc32[i][j][0]; }
  else { for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) sum += acc16[i][j][0]; }
  if (sum == 12345.678f) out[tid] = sum;
}

template <int SHAPE>
void run(const bf16x8* data, const char* what, double target_ms, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 2000;
  for (int pass = 0; pass < 3; ++pass) {          // calibrate, then two long measurements (the second one is the sustained rate)
    hipEventRecord(e0);
    k<SHAPE><<<256, 256>>>(data, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration: 2 k-steps x 16 (32x32x16) or 64 (16x16x32) MFMAs = 2 x 128 x 128 x 16|32 x 2 flops per wave
    const double flops = 256.0 * 4 * iters * 2.0 * 128 * 128 * (SHAPE == 32 ? 16 : 32) * 2;
    if (pass) printf("MFMA %dx%d  %-18s %8.2f ms  %7.1f TFLOP/s\n", SHAPE, SHAPE, what, ms, flops / (ms * 1e-3) / 1e12);
    iters = (int)(iters * target_ms / ms) + 1;
  }
}

int main(int argc, char** argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 200.0;
  const size_t n = 64 * 256;       // bf16x8 per data set
  std::vector<uint16_t> h(n * 8);
  bf16x8* d[3]; float* out; hipMalloc(&out, 4096);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0f; };
  for (int kind = 0; kind < 3; ++kind) {
    for (auto& x : h) {
      float f = 0.f;
      if (kind == 1) f = (float)((int)(rnd() * 4));
      if (kind == 2) { const float u1 = rnd() + 1e-7f, u2 = rnd(); f = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
      uint32_t u; std::memcpy(&u, &f, 4); x = (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
    hipMalloc(&d[kind], n * 16); hipMemcpy(d[kind], h.data(), n * 16, hipMemcpyHostToDevice);
  }
  static const char* names[] = {"zeros", "small integers", "normal(0,1)"};
  for (int kind = 0; kind < 3; ++kind) { run<32>(d[kind], names[kind], target_ms, out); run<16>(d[kind], names[kind], target_ms, out); }
  return 0;
}
