// Does fetching real data slow the MFMA stream, and does it depend on the MFMA shape?  (round 2)
// One 512-thread workgroup per CU (2 waves per SIMD, like the T256 GEMM).  Every wave issues back-to-back independent MFMAs on register
// operands; MODE adds, per 4 (32x32x16) or 8 (16x16x32) MFMAs, one 16-byte-per-lane load: 0 none, 1 buffer_load ... lds (LDS-DMA)
// of real data, 2 the same sent out of range (zero fill, no fetch), 3 global load into VGPRs.  The loads walk a per-CU window of
// a buffer (window = argv[2] KiB: small = L2 / L1 resident, large = streams through the caches).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_vs_fetch mfma_vs_fetch.hip ; run: ./mfma_vs_fetch [iters] [window KiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define LDS(p) ((__attribute__((address_space(3))) void*)(p))

template <int SHAPE, int MODE>
__global__ __launch_bounds__(512) void k(const char* buf, size_t window, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const char* base = buf + (size_t)blockIdx.x * window;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (uint32_t)window, 0x00020000);
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(tid & 3); b[i] = (__bf16)1.0f; }
  f32x16 acc32[8];
  f32x4 acc16[32];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 4; ++j) acc16[i][j] = 0.f;
  u32x4 sink = {0u, 0u, 0u, 0u};
  uint32_t off = (uint32_t)(tid * 16);
  const uint32_t step = 512 * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {                       // 8 groups: 4 x 32x32x16 or 8 x 16x16x32 each (= 131072 flops per lane-group)
      if (MODE == 1 || MODE == 2) {
        const uint32_t o = MODE == 2 ? 0xFFFFFFF0u : off;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS(smem + (g & 3) * 8192 + wave * 1024), 16, o, 0, 0, 0);
      } else if (MODE == 3) {
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        sink ^= v;
      }
      off += step; if (off + step > window) off = (uint32_t)(tid * 16);
      if (SHAPE == 32) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc32[(g & 1) * 4 + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[(g & 1) * 4 + m], 0, 0, 0);
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc16[(g & 3) * 8 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[(g & 3) * 8 + m], 0, 0, 0);
      }
    }
    if (MODE == 1 || MODE == 2) { if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc32[i][0];
  for (int i = 0; i < 32; ++i) s += acc16[i][0];
  if (s == 12345.678f || sink[0] == 0xdeadbeef) out[tid] = s + smem[tid];
}

template <int SHAPE, int MODE>
void run(const char* buf, size_t window, int iters, float* out) {
  hipFuncSetAttribute((const void*)k<SHAPE, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<SHAPE, MODE><<<256, 512, 32768>>>(buf, window, iters / 4, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<SHAPE, MODE><<<256, 512, 32768>>>(buf, window, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 /*waves*/ * iters * 8 * (SHAPE == 32 ? 4 * 32768.0 : 8 * 16384.0);
  const double bytes = MODE ? 256.0 * 512 * 16 * 8.0 * iters : 0.0;
  static const char* names[] = {"no loads", "LDS-DMA, real data", "LDS-DMA, out of range", "global load to VGPR"};
  printf("MFMA %dx%d  %-24s %8.1f us  %7.1f TFLOP/s  %6.2f TB/s fetched\n", SHAPE, SHAPE, names[MODE], ms * 1e3, flops / (ms * 1e-3) / 1e12,
         bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  const size_t window = (size_t)(argc > 2 ? atoi(argv[2]) : 4096) * 1024;
  char* buf; float* out;
  hipMalloc(&buf, window * 256); hipMemset(buf, 1, window * 256); hipMalloc(&out, 4096);
  printf("window per CU %zu KiB, %d iterations, 1 x 16 B/lane load per 131072-flop MFMA group (the GEMM's ratio is 1 per 131072)\n", window / 1024, iters);
  run<32, 0>(buf, window, iters, out); run<32, 1>(buf, window, iters, out); run<32, 2>(buf, window, iters, out); run<32, 3>(buf, window, iters, out);
  run<16, 0>(buf, window, iters, out); run<16, 1>(buf, window, iters, out); run<16, 2>(buf, window, iters, out); run<16, 3>(buf, window, iters, out);
  return 0;
}
