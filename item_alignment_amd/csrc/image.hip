// GPU side of the image input pipeline (reference data.py:838-866: timm create_transform(is_training=False)
// = PIL bicubic resize -> ToTensor -> Normalize, which the reference runs per sample on one CPU thread inside __getitem__).
// Decoded uint8 RGB frames go through Pillow's own resampling arithmetic — two separable passes over 8-bit data with 8.22
// fixed-point coefficients (Pillow src/libImaging/Resample.c: precompute_coeffs / normalize_coeffs_8bpc /
// ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc; the coefficient tables are computed on the host exactly as
// Pillow does and passed in) — so the result is bit-identical to Image.resize(..., BICUBIC); then /255, mean/std and the
// optional horizontal flip in fp32 with IEEE division, identical to ToTensor + Normalize.  HBM-bound byte kernels.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;     // Pillow Resample.c

IA_DEV uint8_t clip8(int v) {                   // clip8_lookups[v >> PRECISION_BITS]
  v >>= PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: dst[b][y][xx][c] = clip8(half + sum_x src[b][y][xmin + x][c] * k[xx][x]),  src [B,H,Win,3] -> dst [B,H,Wout,3]
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                                                       const int* __restrict__ kk, int ksize, int H, int Win, int Wout, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int xx = (int)(idx % Wout);
  const size_t row = idx / Wout;                                   // b*H + y
  const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
  const int* k = kk + (size_t)xx * ksize;
  const uint8_t* s = src + (row * Win + xmin) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < xmax; ++x) {
    const int w = k[x];
    s0 += s[3 * x] * w; s1 += s[3 * x + 1] * w; s2 += s[3 * x + 2] * w;
  }
  uint8_t* d = dst + idx * 3;
  d[0] = clip8(s0); d[1] = clip8(s1); d[2] = clip8(s2);
}

// vertical pass: dst[b][yy][x][c] = clip8(half + sum_y src[b][ymin + y][x][c] * k[yy][y]),  src [B,Hin,W,3] -> dst [B,Hout,W,3]
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                                                       const int* __restrict__ kk, int ksize, int Hin, int Hout, int W, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // over B*Hout*W*3 bytes
  if (idx >= total) return;
  const int xc = (int)(idx % ((size_t)W * 3));
  const size_t r = idx / ((size_t)W * 3);
  const int yy = (int)(r % Hout);
  const size_t b = r / Hout;
  const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
  const int* k = kk + (size_t)yy * ksize;
  const uint8_t* s = src + ((b * Hin + ymin) * W) * 3 + xc;
  int acc = 1 << (PRECISION_BITS - 1);
  for (int y = 0; y < ymax; ++y) acc += s[(size_t)y * W * 3] * k[y];
  dst[idx] = clip8(acc);
}

// ToTensor + Normalize (+ RandomHorizontalFlip decided by the caller): out[b][c][y][x] = (u8[b][y][x'][c] / 255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void u8_to_nchw_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ flip, float* __restrict__ out,
                                                         int S0, int S1, float m0, float m1, float m2, float d0, float d1, float d2, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // over B*S0*S1 pixels
  if (idx >= total) return;
  const int x = (int)(idx % S1);
  const size_t r = idx / S1;
  const int y = (int)(r % S0);
  const size_t b = r / S0;
  const int xs = (flip && flip[b]) ? S1 - 1 - x : x;
  const uint8_t* s = src + ((b * S0 + y) * S1 + xs) * 3;
  const size_t plane = (size_t)S0 * S1;
  float* o = out + b * 3 * plane + (size_t)y * S1 + x;
  o[0] = (__fdiv_rn((float)s[0], 255.f) - m0) / d0;
  o[plane] = (__fdiv_rn((float)s[1], 255.f) - m1) / d1;
  o[2 * plane] = (__fdiv_rn((float)s[2], 255.f) - m2) / d2;
}

}  // namespace

// One separable resampling pass over a batch of equally sized uint8 RGB frames.  horizontal != 0: src [B,H,Win,3] -> dst
// [B,H,Wout,3]; else src [B,Hin,W,3] -> dst [B,Hout,W,3].  bounds [n_out][2] = (first input index, tap count), coeffs
// [n_out][ksize] int32 in 8.22 fixed point: Pillow's precompute_coeffs + normalize_coeffs_8bpc tables for this size pair.
extern "C" int ia_resize_pass_u8(const uint8_t* src, uint8_t* dst, const int* bounds, const int* coeffs, int ksize, int B, int in_len,
                                 int out_len, int other_len, int horizontal, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !dst || !bounds || !coeffs || ksize <= 0 || B <= 0 || in_len <= 0 || out_len <= 0 || other_len <= 0) return IA_ERR_ARG;
  if (horizontal) {
    const size_t total = (size_t)B * other_len * out_len;        // other_len = H
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, dst, bounds, coeffs, ksize, other_len,
                       in_len, out_len, total);
  } else {
    const size_t total = (size_t)B * out_len * other_len * 3;    // other_len = W
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, dst, bounds, coeffs, ksize, in_len,
                       out_len, other_len, total);
  }
  return ia_check_launch();
}

// uint8 [B,S0,S1,3] -> fp32 [B,3,S0,S1], (x/255 - mean)/std, optional per-image horizontal flip (flip: [B] device bytes or
// NULL); mean3 / std3 are HOST arrays of three floats
extern "C" int ia_u8_to_nchw_normalized(const uint8_t* src, const uint8_t* flip, float* out, int B, int S0, int S1, const float* mean3,
                                        const float* std3, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !out || !mean3 || !std3 || B <= 0 || S0 <= 0 || S1 <= 0) return IA_ERR_ARG;
  const size_t total = (size_t)B * S0 * S1;
  hipLaunchKernelGGL(u8_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, flip, out, S0, S1, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2], total);
  return ia_check_launch();
}
