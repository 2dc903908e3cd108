// What a 64-MFMA k-tile body costs at one wave per SIMD, piece by piece (round 4, GEMM main loop): bare v_mfma_f32_32x32x16_bf16 stream over 16
// accumulators, + barriers, + ds_read_b128 fragment reads, + LDS-DMA pieces (L2-resident source).  Cycles per MFMA from s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_stream.bin mfma_stream.hip ; run: ./mfma_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) char lds_char;

template <int NBAR, int NREAD, int NDMA, int ORDER>
__global__ __launch_bounds__(256) void k(const char* src, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[4], b[4], ra[8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(float)(lane + i + j); b[i][j] = (__bf16)(float)(lane - i - j); }
  // fragment read of the production kernel: row = lane & 31, 16-byte chunk (lane >> 5) XOR (row >> 1) & 7 of a 128-byte row (conflict-free)
  const uint32_t laddr = (uint32_t)(uintptr_t)(lds_char*)smem + (uint32_t)((wave & 1) * 16384 + (lane & 31) * 128 + (((lane >> 5) ^ (((lane & 31) >> 1) & 7)) << 4));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 1u << 22, 0x00020000);
  lds_char* const ldst = (lds_char*)smem + 65536 + wave * 16384;
  const uint32_t voff = (uint32_t)(blockIdx.x & 63) * 65536u + lane * 16;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      const int ai = ORDER ? (m >> 2) & 3 : m & 3, bi = ORDER ? m & 3 : (m >> 2) & 3;
      acc[m & 15] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ai], b[bi], acc[m & 15], 0, 0, 0);
      if (NREAD && (m * NREAD) / 64 != ((m + 1) * NREAD) / 64) {
        const int r = (m * NREAD) / 64;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ra[r & 7]) : "v"(laddr ^ (uint32_t)((r & 3) << 5)), "n"((r & 12) * 1024));
      }
      if (NDMA && (m * NDMA) / 64 != ((m + 1) * NDMA) / 64) {
        const int r = (m * NDMA) / 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ldst + (r & 15) * 1024, 16, voff, (r & 15) * 4096 + (it & 7) * 128, 0, 0);
      }
      if (NBAR && (m * NBAR) / 64 != ((m + 1) * NBAR) / 64) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NREAD) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]));
    if (NDMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][lane & 15];
  if (NREAD) s += (float)ra[0][0] + (float)ra[7][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NBAR, int NREAD, int NDMA, int ORDER>
void run(const char* name, const char* src, float* out, unsigned long long* cyc, int nwg) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<NBAR, NREAD, NDMA, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NBAR, NREAD, NDMA, ORDER>), dim3(nwg), dim3(256), 128 * 1024, 0, src, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= nwg;
  const double per = avg / (iters * 64.0);
  printf("%-44s %6.2f cycles/MFMA  %8.1f us  %.3f GHz  %7.1f TF/s\n", name, per, ms * 1e3, avg / (ms * 1e-3) * 1e-9,
         (double)nwg * 4 * iters * 64 * 32768.0 / (ms * 1e-3) * 1e-12);
}


// VALU stream (an epilogue's arithmetic: 512 independent v_fma_f32 per trip) with 16 LDS-DMA pieces either in one burst in front of it or
// one per 32 fmas: does spreading a tile prologue's pieces over the epilogue hide their issue cost?
template <int MODE>
__global__ __launch_bounds__(256) void kv(const char* src, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (float)(lane + i);
  const float c = 1.0001f, d = 0.5f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 1u << 22, 0x00020000);
  lds_char* const ldst = (lds_char*)smem + wave * 16384;
  const uint32_t voff = (uint32_t)(blockIdx.x & 63) * 65536u + lane * 16;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ldst + r * 1024, 16, voff, r * 4096 + (it & 7) * 128, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 512; ++m) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[m & 15]) : "v"(c), "v"(d));
      if (MODE == 2 && (m & 31) == 31) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ldst + (m >> 5) * 1024, 16, voff, (m >> 5) * 4096 + (it & 7) * 128, 0, 0);
    }
    if (MODE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void runv(const char* name, const char* src, float* out, unsigned long long* cyc) {
  const int iters = 2000, nwg = 256;
  hipFuncSetAttribute((const void*)kv<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((kv<MODE>), dim3(nwg), dim3(256), 64 * 1024, 0, src, out, cyc, iters); hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= nwg;
  printf("%-44s %8.1f cycles per trip (512 fma%s)\n", name, avg / iters, MODE ? " + 16 pieces" : "");
}

int main() {
  char* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 1 << 23); hipMemset(src, 1, 1 << 23);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<0, 0, 0, 0>("bare MFMA stream, b fixed for 4", src, out, cyc, 256);
  run<0, 0, 0, 1>("bare MFMA stream, a fixed for 4", src, out, cyc, 256);
  run<2, 0, 0, 0>("+ 2 barriers / 64", src, out, cyc, 256);
  run<0, 32, 0, 0>("+ 32 ds_read_b128 / 64", src, out, cyc, 256);
  run<0, 0, 16, 0>("+ 16 LDS-DMA pieces / 64", src, out, cyc, 256);
  run<2, 32, 0, 0>("+ barriers + reads", src, out, cyc, 256);
  run<2, 32, 16, 0>("+ barriers + reads + DMA", src, out, cyc, 256);
  run<0, 16, 0, 0>("+ 16 ds_read_b128 / 64", src, out, cyc, 256);
  run<0, 32, 16, 0>("+ reads + DMA", src, out, cyc, 256);
  run<0, 0, 8, 0>("+ 8 LDS-DMA pieces / 64", src, out, cyc, 256);
  run<0, 0, 0, 0>("bare, one workgroup only", src, out, cyc, 1);
  runv<0>("VALU stream alone", src, out, cyc);
  runv<1>("VALU stream, 16 pieces in a burst in front", src, out, cyc);
  runv<2>("VALU stream, one piece per 32 fmas", src, out, cyc);
  return 0;
}
