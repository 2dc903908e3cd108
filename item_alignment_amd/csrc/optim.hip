// Optimiser step over the flat parameter arena (HBM-bound: 28 B/param read+write + 2 B shadow), gfx950.
//
// The reference builds torch.optim.AdamW with two parameter groups (weight decay everywhere except
// names containing "bias" / "LayerNorm.weight", finetune_multimodal.py:296-308) and steps it once per
// optimiser step (:460-468).  Here all fp32 master parameters, their gradients and both Adam moments
// live in four flat arenas with identical element offsets; one launch walks a static chunk table
// {offset, count, weight-decay flag} and also refreshes the bf16 shadow copy the GEMMs read.
#include "common.h"
#include "../../include/itemalign.h"
#include <cstdio>

namespace {

struct Chunk { uint32_t offset_lo; uint32_t offset_hi; uint32_t count; uint32_t decay; };

// PyTorch AdamW update order: p *= 1 - lr*wd ; m,v update ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ shadow, const Chunk* __restrict__ table,
                                                    float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    float grad_scale) {
  const Chunk c = table[blockIdx.x];
  const size_t base = ((size_t)c.offset_hi << 32) | c.offset_lo;
  const float decay = c.decay ? 1.f - lr * wd : 1.f;
  const float step = lr / bc1;
  for (uint32_t i = threadIdx.x * 4; i < c.count; i += 256 * 4) {
    const size_t e = base + i;
    if (i + 4 <= c.count) {
      f32x4 pp = *reinterpret_cast<f32x4*>(p + e);
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + e);
      f32x4 mm = *reinterpret_cast<f32x4*>(m + e), vv = *reinterpret_cast<f32x4*>(v + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gr = gg[j] * grad_scale;
        pp[j] *= decay;
        mm[j] = beta1 * mm[j] + (1.f - beta1) * gr;
        vv[j] = beta2 * vv[j] + (1.f - beta2) * gr * gr;
        pp[j] -= step * mm[j] / (sqrtf(vv[j]) / bc2_sqrt + eps);
      }
      *reinterpret_cast<f32x4*>(p + e) = pp;
      *reinterpret_cast<f32x4*>(m + e) = mm;
      *reinterpret_cast<f32x4*>(v + e) = vv;
      if (shadow) {
        bf16x4 s = {f2bf(pp[0]), f2bf(pp[1]), f2bf(pp[2]), f2bf(pp[3])};
        *reinterpret_cast<bf16x4*>(shadow + e) = s;
      }
    } else {
      for (uint32_t j = i; j < c.count; ++j) {
        const size_t ee = base + j;
        const float gr = g[ee] * grad_scale;
        float pp = p[ee] * decay;
        const float mm = beta1 * m[ee] + (1.f - beta1) * gr;
        const float vv = beta2 * v[ee] + (1.f - beta2) * gr * gr;
        pp -= step * mm / (sqrtf(vv) / bc2_sqrt + eps);
        p[ee] = pp; m[ee] = mm; v[ee] = vv;
        if (shadow) shadow[ee] = f2bf(pp);
      }
    }
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
    if (i + 4 <= n) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + i);
      bf16x4 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
      *reinterpret_cast<bf16x4*>(dst + i) = o;
    } else {
      for (size_t j = i; j < n; ++j) dst[j] = f2bf(src[j]);
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = bf2f(src[i]);
}

// dst[c][r] = src[r][c] for every matrix of the table: 64 x 64 tiles through LDS (both sides move 128-byte rows), one workgroup per tile
struct TransposeEntry { uint32_t offset_lo; uint32_t offset_hi; uint32_t rows; uint32_t cols; };
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                                const TransposeEntry* __restrict__ table) {
  __shared__ bf16 tile[64][66];
  const TransposeEntry e = table[blockIdx.y];
  const uint32_t tiles_c = (e.cols + 63) / 64, tiles_r = (e.rows + 63) / 64;
  if (blockIdx.x >= tiles_c * tiles_r) return;
  const size_t base = ((size_t)e.offset_hi << 32) | e.offset_lo;
  const uint32_t r0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 64;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 lanes x 2 columns, 8 row groups
  for (int i = ty; i < 64; i += 8) {
    const uint32_t r = r0 + i, c = c0 + 2 * tx;
    bf16 a = f2bf(0.f), b = f2bf(0.f);
    if (r < e.rows && c < e.cols) a = src[base + (size_t)r * e.cols + c];
    if (r < e.rows && c + 1 < e.cols) b = src[base + (size_t)r * e.cols + c + 1];
    tile[i][2 * tx] = a; tile[i][2 * tx + 1] = b;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 8) {
    const uint32_t c = c0 + i, r = r0 + 2 * tx;                          // output row = source column
    if (c < e.cols && r < e.rows) dst[base + (size_t)c * e.rows + r] = tile[2 * tx][i];
    if (c < e.cols && r + 1 < e.rows) dst[base + (size_t)c * e.rows + r + 1] = tile[2 * tx + 1][i];
  }
}

}  // namespace

// Transposed bf16 copies of 2-D weights (the data-gradient GEMMs dx = dy W read W^T k-contiguously: the faster operand form).
// table: device array of n {offset_lo, offset_hi, rows, cols} (element offsets into src / dst, the same on both sides);
// dst[offset .. offset + rows*cols) receives the [cols, rows] transpose of src's [rows, cols] matrix.  max_tiles = the largest
// ceil(rows/64) * ceil(cols/64) of the table.
extern "C" int ia_transpose_bf16_batched(const void* src, void* dst, const void* table, int n, int max_tiles, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !dst || !table || n <= 0 || max_tiles <= 0 || n > 65535) return IA_ERR_ARG;
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(max_tiles, n), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst, (const TransposeEntry*)table);
  return ia_check_launch();
}

// table: device array of n_chunks {offset_lo, offset_hi, count (<= 4096, offsets 4-element aligned), decay}.
extern "C" int ia_adamw_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                             const void* chunk_table, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int step, float grad_scale, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!params || !grads || !exp_avg || !exp_avg_sq || !chunk_table || n_chunks <= 0 || step <= 0) return IA_ERR_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, (bf16*)shadow_bf16,
                     (const Chunk*)chunk_table, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
  return ia_check_launch();
}

extern "C" int ia_cast_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!src || !dst || n == 0) return IA_ERR_ARG;
  size_t g = (n / 4 + 255) / 256; if (g > 8192) g = 8192; if (g == 0) g = 1;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((int)g), dim3(256), 0, stream, src, (bf16*)dst, n);
  return ia_check_launch();
}

extern "C" int ia_cast_bf16_to_f32(const void* src, float* dst, size_t n, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!src || !dst || n == 0) return IA_ERR_ARG;
  size_t g = (n + 255) / 256; if (g > 8192) g = 8192;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((int)g), dim3(256), 0, stream, (const bf16*)src, dst, n);
  return ia_check_launch();
}

extern "C" const char* ia_strerror(int code) {
  switch (code) {
    case IA_OK: return "ok";
    case IA_ERR_ARG: return "invalid argument (null pointer, shape or alignment)";
    case IA_ERR_LAUNCH: {
      static char buf[256];
      snprintf(buf, sizeof buf, "HIP kernel launch failed: %s", hipGetErrorString(g_ia_last_hip_error));
      return buf;
    }
    case IA_ERR_WORKSPACE: return "workspace missing or too small";
    case IA_ERR_UNSUPPORTED: return "unsupported operand layout / epilogue combination";
  }
  return "unknown error";
}

extern "C" int ia_abi_version(void) { return IA_ABI_VERSION; }
