// Do out-of-range LDS-DMA instructions retire in order with the real ones in front of them?  The attention and GEMM kernels wait with
// COUNTED s_waitcnt vmcnt(N) and rely on "at most N outstanding <=> every older DMA has landed" also when the youngest N are
// (partly) out of range -- the last key tile of a sequence, a k-tile past the end of K.  Each wave: four real DMAs from a cold 1 GiB
// region into its own LDS slice (pre-filled with a pattern), then four DMAs whose lanes are all out of range, then vmcnt(4) and an
// immediate read of the slice.  A pattern word still there = the real DMA had not landed when the counter said so.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) char lds_char;
__device__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }
constexpr uint32_t PAT = 0x3F803F80u;
__global__ __launch_bounds__(256) void k(const uint32_t* big, uint32_t big_bytes, int rounds, int mode, unsigned long long* cnt) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t* mine = reinterpret_cast<uint32_t*>(smem + wave * 8192);              // 4 KiB real target + 4 KiB out-of-range target
  lds_char* lsm = (lds_char*)((__attribute__((address_space(3))) void*)smem) + wave * 8192;
  const __amdgpu_buffer_rsrc_t rb = rsrc(big, big_bytes);
  unsigned long long late = 0, total = 0, nonzero_oob = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int i = lane; i < 2048; i += 64) mine[i] = PAT;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t base = (uint32_t)((((blockIdx.x * 977u + r * 131u + wave * 17u) * 65536u) % (big_bytes - 65536u)) & ~15u);
    for (int j = 0; j < 4; ++j)                                                  // real: 1 KiB each, far apart (cold lines)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lsm + j * 1024, 16, (uint32_t)lane * 16u, (int)(base + j * 16384u), 0, 0);
    for (int j = 0; j < 4; ++j) {                                                // out of range: huge voffset (mode 0) or soffset past the end (mode 1)
      if (mode == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lsm + 4096 + j * 1024, 16, 0xFFFFFFF0u, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lsm + 4096 + j * 1024, 16, (uint32_t)lane * 16u, (int)big_bytes, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    for (int i = lane; i < 1024; i += 64) { late += mine[i] == PAT; ++total; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = lane; i < 1024; i += 64) nonzero_oob += mine[1024 + i] != 0u;
  }
  atomicAdd(cnt + 0, late); atomicAdd(cnt + 1, total); atomicAdd(cnt + 2, nonzero_oob);
}
int main() {
  const uint32_t big_bytes = 1u << 30;
  uint32_t* big; unsigned long long* cnt;
  hipMalloc(&big, big_bytes); hipMalloc(&cnt, 32);
  hipMemsetD32(big, 0x40004000u, big_bytes / 4);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(cnt, 0, 32);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, big, big_bytes, 16, mode, cnt);
    hipDeviceSynchronize();
    unsigned long long h[4]; hipMemcpy(h, cnt, 32, hipMemcpyDeviceToHost);
    printf("mode %d: words of the real DMAs not landed at vmcnt(4): %llu of %llu; non-zero words in the out-of-range targets: %llu\n", mode, h[0], h[1], h[2]);
  }
  return 0;
}
