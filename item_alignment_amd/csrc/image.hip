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

// horizontal pass: dst[b][y][xx][c] = clip8(half + sum_x src[b][y][xmin + x][c] * k[xx][x]),  src rows `pitch` bytes apart, frames
// `frame` bytes apart (a crop window of a larger frame is just a shifted base pointer with the frame's pitch) -> dst [B,H,Wout,3]
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                                                       const int* __restrict__ kk, int ksize, int H, size_t pitch, size_t frame, int Wout,
                                                       size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int xx = (int)(idx % Wout);
  const size_t row = idx / Wout;                                   // b*H + y
  const size_t b = row / H, y = row % H;
  const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
  const int* k = kk + (size_t)xx * ksize;
  const uint8_t* s = src + b * frame + y * pitch + (size_t)xmin * 3;
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

// ---- torchvision ColorJitter on uint8 frames = Pillow ImageEnhance: out = blend(degenerate, image, factor) with
// (Pillow src/libImaging/Blend.c, float arithmetic, truncating store; factors outside [0,1] clip)
//   brightness  degenerate = 0
//   contrast    degenerate = int(mean(L) + 0.5) in every channel, L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16 (Convert.c rgb2l)
//   saturation  degenerate = L of the pixel in every channel
IA_DEV int luma601(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

IA_DEV uint8_t blend8(int deg, int v, float alpha) {
  const float t = __fadd_rn((float)deg, __fmul_rn(alpha, (float)(v - deg)));     // no fma: Pillow rounds the product first
  if (alpha >= 0.f && alpha <= 1.f) return (uint8_t)(int)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (uint8_t)(int)t);
}

// sums[b] += sum of L over this block's pixels of frame b (frames [B, npix, 3]); grid (blocks per frame, B)
__global__ __launch_bounds__(256) void luma_sum_kernel(const uint8_t* __restrict__ frames, unsigned long long* __restrict__ sums, int npix) {
  const size_t b = blockIdx.y;
  unsigned int acc = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const uint8_t* p = frames + (b * npix + i) * 3;
    acc += (unsigned)luma601(p[0], p[1], p[2]);
  }
  __shared__ unsigned int red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&sums[b], (unsigned long long)red[0]);
}

// one ColorJitter step on every frame, in place: op[b] 0 none / 1 brightness / 2 contrast / 3 saturation, factor[b]
__global__ __launch_bounds__(256) void jitter_step_kernel(uint8_t* __restrict__ frames, const int* __restrict__ op, const float* __restrict__ factor,
                                                          const unsigned long long* __restrict__ sums, int npix) {
  const size_t b = blockIdx.y;
  const int o = op[b];
  if (o == 0) return;
  const float f = factor[b];
  int mean = 0;
  if (o == 2) mean = (int)((double)sums[b] / (double)npix + 0.5);     // int(ImageStat.Stat(L).mean[0] + 0.5), double like Python
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    uint8_t* p = frames + (b * npix + i) * 3;
    const int r = p[0], g = p[1], bl = p[2];
    const int deg = o == 1 ? 0 : (o == 2 ? mean : luma601(r, g, bl));
    p[0] = blend8(deg, r, f); p[1] = blend8(deg, g, f); p[2] = blend8(deg, bl, f);
  }
}

}  // namespace

// One separable resampling pass over a batch of equally sized uint8 RGB frames.  horizontal != 0: src [B,H,Win,3] -> dst
// [B,H,Wout,3]; else src [B,Hin,W,3] -> dst [B,Hout,W,3].  bounds [n_out][2] = (first input index, tap count), coeffs
// [n_out][ksize] int32 in 8.22 fixed point: Pillow's precompute_coeffs + normalize_coeffs_8bpc tables for this size pair.
extern "C" int ia_resize_pass_u8_ex(const uint8_t* src, size_t src_pitch_bytes, size_t src_frame_bytes, uint8_t* dst, const int* bounds,
                                    const int* coeffs, int ksize, int B, int in_len, int out_len, int other_len, int horizontal,
                                    hipStream_t stream);

extern "C" int ia_resize_pass_u8(const uint8_t* src, uint8_t* dst, const int* bounds, const int* coeffs, int ksize, int B, int in_len,
                                 int out_len, int other_len, int horizontal, hipStream_t stream) {
  // packed frames: rows in_len * 3 bytes apart (horizontal pass), frames other_len rows apart
  return ia_resize_pass_u8_ex(src, (size_t)in_len * 3, (size_t)other_len * in_len * 3, dst, bounds, coeffs, ksize, B, in_len, out_len, other_len,
                              horizontal, stream);
}

// The same pass with an explicit source layout for the HORIZONTAL pass: rows src_pitch_bytes apart, frames src_frame_bytes apart.
// A crop window (RandomResizedCrop) of a decoded frame is the frame's pitch with src pointing at the window's first pixel and
// in_len = the window width; out_len may also be a sub-range of a resize (CenterCrop) when the tables are the matching slice.
// The vertical pass reads packed [B, in_len, other_len, 3] input (the horizontal pass's output); the two layout arguments are ignored.
extern "C" int ia_resize_pass_u8_ex(const uint8_t* src, size_t src_pitch_bytes, size_t src_frame_bytes, uint8_t* dst, const int* bounds,
                                    const int* coeffs, int ksize, int B, int in_len, int out_len, int other_len, int horizontal,
                                    hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !dst || !bounds || !coeffs || ksize <= 0 || B <= 0 || in_len <= 0 || out_len <= 0 || other_len <= 0) return IA_ERR_ARG;
  if (horizontal) {
    if (src_pitch_bytes < (size_t)in_len * 3) return IA_ERR_ARG;
    const size_t total = (size_t)B * other_len * out_len;        // other_len = H
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, dst, bounds, coeffs, ksize, other_len,
                       src_pitch_bytes, src_frame_bytes, out_len, total);
  } else {
    const size_t total = (size_t)B * out_len * other_len * 3;    // other_len = W
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, dst, bounds, coeffs, ksize, in_len,
                       out_len, other_len, total);
  }
  return ia_check_launch();
}

// One step of torchvision's ColorJitter (= Pillow ImageEnhance blends) on uint8 frames [B, H, W, 3] in place: op [B] int32 (0 none,
// 1 brightness, 2 contrast, 3 saturation), factor [B] fp32, scratch [B] uint64 (device, used for the contrast means).  The caller
// issues one call per position of each image's random op order.
extern "C" int ia_color_jitter_step_u8(uint8_t* frames, const int* op, const float* factor, void* scratch, int B, int H, int W,
                                       int any_contrast, hipStream_t stream) {
  (void)hipGetLastError();
  if (!frames || !op || !factor || !scratch || B <= 0 || H <= 0 || W <= 0) return IA_ERR_ARG;
  const int npix = H * W;
  int blocks = (npix + 255) / 256; if (blocks > 64) blocks = 64;
  if (any_contrast) {
    if (hipMemsetAsync(scratch, 0, (size_t)B * 8, stream) != hipSuccess) return IA_ERR_LAUNCH;
    hipLaunchKernelGGL(luma_sum_kernel, dim3(blocks, B), dim3(256), 0, stream, frames, (unsigned long long*)scratch, npix);
  }
  hipLaunchKernelGGL(jitter_step_kernel, dim3(blocks, B), dim3(256), 0, stream, frames, op, factor, (const unsigned long long*)scratch, npix);
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
