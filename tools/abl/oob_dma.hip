// What does an LDS-DMA (buffer_load ... lds) do for out-of-range lanes on gfx950?  Each workgroup pre-fills a piece of LDS with a
// pattern, issues 16-byte-per-lane DMAs whose lanes are (mode 0) out of range through a huge voffset, (mode 1) out of range through
// voffset + soffset >= num_records, (mode 2) half in range, waits vmcnt(0) and classifies every dword it finds:
// zero / still the pattern (not written) / the buffer's data / anything else.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((address_space(3))) char lds_char;
__device__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }
constexpr uint32_t PAT = 0x3F803F80u, DATA = 0x40004000u;
__global__ __launch_bounds__(256) void k(const uint32_t* buf, uint32_t window, int mode, int rounds, unsigned long long* cnt, const uint32_t* big, uint32_t big_bytes) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t* s32 = reinterpret_cast<uint32_t*>(smem);
  lds_char* lsm = (lds_char*)((__attribute__((address_space(3))) void*)smem);
  const __amdgpu_buffer_rsrc_t rs = rsrc(buf + (size_t)(blockIdx.x & 1023) * (window / 4), window);
  const __amdgpu_buffer_rsrc_t rb = rsrc(big, big_bytes);
  unsigned long long c_zero = 0, c_pat = 0, c_data = 0, c_other = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int i = tid; i < 4096; i += 256) s32[i] = PAT;                 // 16 KiB target region
    __syncthreads();
    // background traffic into the other 48 KiB (valid loads)
    for (int j = 0; j < 12; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lsm + 16384 + j * 4096 + wave * 1024, 16, (uint32_t)lane * 16u,
                                               (uint32_t)(((blockIdx.x * 131u + r * 17u + j * 7u + wave) * 4096u) % (big_bytes - 4096u)) & ~15u, 0, 0);
    // the DMAs under test: 4 per wave, 1 KiB each -> 16 KiB
    for (int j = 0; j < 4; ++j) {
      uint32_t vo = (uint32_t)lane * 16u, so = (uint32_t)(j * 4 + wave) * 1024u;
      if (mode == 0) { vo = 0xFFFFFFF0u; }
      else if (mode == 1) { so += window; }
      else if (mode == 2) { so = window - 512u; }                      // lanes 0..31 in range, 32..63 out
      else if (mode == 3) { so = so % (window - 1024u); }              // all in range
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lsm + (j * 4 + wave) * 1024, 16, vo, so, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4096; i += 256) {
      const uint32_t v = s32[i];
      if (v == 0u) ++c_zero; else if (v == PAT) ++c_pat; else if (v == DATA) ++c_data; else ++c_other;
    }
    __syncthreads();
  }
  atomicAdd(cnt + 0, c_zero); atomicAdd(cnt + 1, c_pat); atomicAdd(cnt + 2, c_data); atomicAdd(cnt + 3, c_other);
}
int main() {
  const uint32_t window = 64 * 1024, big_bytes = 256u << 20;
  uint32_t *buf, *big; unsigned long long* cnt;
  hipMalloc(&buf, (size_t)1024 * window + 4096); hipMalloc(&big, big_bytes); hipMalloc(&cnt, 32);
  hipMemsetD32(buf, DATA, (size_t)1024 * window / 4 + 1024); hipMemsetD32(big, 0x7F7F7F7Fu, big_bytes / 4);
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(cnt, 0, 32);
    hipLaunchKernelGGL(k, dim3(8192), dim3(256), 0, 0, buf, window, mode, 8, cnt, big, big_bytes);
    hipDeviceSynchronize();
    unsigned long long h[4]; hipMemcpy(h, cnt, 32, hipMemcpyDeviceToHost);
    printf("mode %d: zero %llu  not-written %llu  data %llu  other %llu\n", mode, h[0], h[1], h[2], h[3]);
  }
  return 0;
}
