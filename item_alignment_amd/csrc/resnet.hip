// Pieces of the pre-activation ResNetV2 image tower (reference src/models/image.py:298-378 ResNetTwoTower -> timm 0.6.5
// resnetv2.py ResNetV2 / PreActBottleneck with norm_layer=BatchNormAct2d, conv_layer=create_conv2d, stem_type ''): activations are
// NHWC rows [B*H*W, C] bf16 as in conv.hip, so the 1x1 / 3x3 convolutions are the GEMMs of gemm.hip / conv.hip.  What is new
// here: training-mode BatchNorm + ReLU with per-call batch statistics (the reference runs the two towers as two separate
// forward calls, so a batch of 2B images is normalised in two segments of B), the 7x7 stem patch gather straight from the
// NCHW fp32 images, MaxPool 3x3/2, the row subsampling of a strided 1x1 convolution and the plain (unstandardised) weight
// re-layout.  All HBM-bound kernels with 16-byte accesses; reductions have a fixed order (deterministic).
#include "common.h"
#include "../../include/itemalign.h"

namespace {

inline unsigned blocks_of(size_t total) { return (unsigned)((total + 255) / 256); }

// --------------------------------------------------------------------------------------- BatchNorm statistics
// part[(seg * nsplit + s) * C + c] (two planes: plane 0 = sum a, plane 1 = sum b) over rows of slice s of segment seg:
//   MODE 0 (forward):  a = x,           b = x*x
//   MODE 1 (backward): a = g,           b = g * xhat      with g = dy * [act > 0] (relu) and xhat = (x - mean) * rstd
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ part, int rows_per_seg, int C,
                                                         int nsplit, int relu, size_t plane) {
  __shared__ float red[256 * 16];
  const int seg = blockIdx.y, s = blockIdx.x;
  const int per = (rows_per_seg + nsplit - 1) / nsplit, r0 = s * per, r1 = min(rows_per_seg, r0 + per);
  const int c8n = C >> 3;
  const int ncol = c8n < 256 ? c8n : 256, nlane = 256 / ncol;
  const int col = threadIdx.x % ncol, lane = threadIdx.x / ncol;
  const size_t base = (size_t)seg * rows_per_seg;
  for (int c0 = 0; c0 < c8n; c0 += ncol) {
    const int c = (c0 + col) * 8;
    const bool live = lane < nlane && c0 + col < c8n;
    float a[8], b[8], mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = b[j] = 0.f; mu[j] = rs[j] = ga[j] = be[j] = 0.f; }
    if (MODE == 1 && live) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { mu[j] = mean[seg * C + c + j]; rs[j] = rstd[seg * C + c + j]; ga[j] = gamma[c + j]; be[j] = beta[c + j]; }
    }
    if (live)
      for (int r = r0 + lane; r < r1; r += nlane) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (base + r) * C + c);
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]); a[j] += f; b[j] += f * f; }
        } else {
          const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(dy + (base + r) * C + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xh = (bf2f(v[j]) - mu[j]) * rs[j];
            float g = bf2f(g8[j]);
            if (relu && xh * ga[j] + be[j] <= 0.f) g = 0.f;
            a[j] += g; b[j] += g * xh;
          }
        }
      }
    if (nlane > 1) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = a[j]; red[threadIdx.x * 16 + 8 + j] = b[j]; }
      __syncthreads();
      if (lane == 0)
        for (int l = 1; l < nlane; ++l)
#pragma unroll
          for (int j = 0; j < 8; ++j) { a[j] += red[(l * ncol + col) * 16 + j]; b[j] += red[(l * ncol + col) * 16 + 8 + j]; }
    }
    if (lane == 0 && c0 + col < c8n)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        part[((size_t)seg * nsplit + s) * C + c + j] = a[j];
        part[plane + ((size_t)seg * nsplit + s) * C + c + j] = b[j];
      }
  }
}

// sum over the nsplit partials of (segment, channel): 16 lanes take every 16th partial, then lane 0 folds the 16 lane sums in
// order (fixed order -> deterministic).  Block = 16 channels x 16 lanes.
IA_DEV void fold_partials(const float* __restrict__ part, size_t plane, int seg, int nsplit, int C, int c, int lane, float* red, float& s, float& q) {
  s = 0.f; q = 0.f;
  if (c < C)
    for (int i = lane; i < nsplit; i += 16) { s += part[((size_t)seg * nsplit + i) * C + c]; q += part[plane + ((size_t)seg * nsplit + i) * C + c]; }
  __syncthreads();
  red[threadIdx.x * 2] = s; red[threadIdx.x * 2 + 1] = q;
  __syncthreads();
  if (lane == 0) {
    const int ch = threadIdx.x >> 4;
    s = 0.f; q = 0.f;
    for (int l = 0; l < 16; ++l) { s += red[(ch * 16 + l) * 2]; q += red[(ch * 16 + l) * 2 + 1]; }
  }
}

// forward finish: mean / rstd per (segment, channel) from the partial sums; running statistics follow nn.BatchNorm2d (momentum
// update with the unbiased variance), one segment after the other exactly as two consecutive forward calls would do
__global__ __launch_bounds__(256) void bn_finish_fwd_kernel(const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var, int C, int nsplit,
                                                            int segments, int rows_per_seg, float eps, float momentum, size_t plane) {
  __shared__ float red[512];
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4), lane = threadIdx.x & 15;
  const bool owner = lane == 0 && c < C;
  float rm = (owner && running_mean) ? running_mean[c] : 0.f, rv = (owner && running_var) ? running_var[c] : 0.f;
  for (int seg = 0; seg < segments; ++seg) {
    float s, q;
    fold_partials(part, plane, seg, nsplit, C, c, lane, red, s, q);
    if (!owner) continue;
    const float n = (float)rows_per_seg, mu = s / n;
    float var = q / n - mu * mu;
    var = var < 0.f ? 0.f : var;
    mean[seg * C + c] = mu;
    rstd[seg * C + c] = rsqrtf(var + eps);
    rm = (1.f - momentum) * rm + momentum * mu;
    rv = (1.f - momentum) * rv + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
  }
  if (owner && running_mean) running_mean[c] = rm;
  if (owner && running_var) running_var[c] = rv;
}

// eval mode: mean / rstd from the running statistics (every segment alike)
__global__ __launch_bounds__(256) void bn_running_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int C, int segments, float eps) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  for (int seg = 0; seg < segments; ++seg) { mean[seg * C + c] = running_mean[c]; rstd[seg * C + c] = rsqrtf(running_var[c] + eps); }
}

// Row-slab layout shared by the element-wise passes: blockIdx.y = segment, blockIdx.x = slab of `per` rows; the block is
// (C/8 column threads) x (row lanes), so a thread keeps its 8 channels for the whole slab and loads their statistics once.
struct Slab { int r0, r1, ncol, nlane, col, lane, c8n; size_t base; };
IA_DEV Slab slab_of(int rows_per_seg, int C, int per) {
  Slab t;
  t.r0 = blockIdx.x * per; t.r1 = min(rows_per_seg, t.r0 + per);
  t.c8n = C >> 3;
  t.ncol = t.c8n < 256 ? t.c8n : 256; t.nlane = 256 / t.ncol;
  t.col = threadIdx.x % t.ncol; t.lane = threadIdx.x / t.ncol;
  t.base = (size_t)blockIdx.y * rows_per_seg;
  return t;
}
IA_DEV void load8(const float* __restrict__ p, float* v) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// y = act((x - mean) * rstd * gamma + beta)
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ y,
                                                       int rows_per_seg, int C, int relu, int per) {
  const Slab t = slab_of(rows_per_seg, C, per);
  const int seg = blockIdx.y;
  for (int c0 = 0; c0 < t.c8n; c0 += t.ncol) {
    if (t.lane >= t.nlane || c0 + t.col >= t.c8n) continue;
    const int c = (c0 + t.col) * 8;
    float mu[8], sc[8], sh[8], ga[8];
    load8(mean + seg * C + c, mu); load8(rstd + seg * C + c, sc); load8(gamma + c, ga); load8(beta + c, sh);
    for (int r = t.r0 + t.lane; r < t.r1; r += t.nlane) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (t.base + r) * C + c);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = (bf2f(v[j]) - mu[j]) * sc[j] * ga[j] + sh[j];
        if (relu && f < 0.f) f = 0.f;
        o[j] = f2bf(f);
      }
      *reinterpret_cast<bf16x8*>(y + (t.base + r) * C + c) = o;
    }
  }
}

// backward finish: sums[seg][c] = (sum g, sum g*xhat); dgamma += sum over segments of sum g*xhat, dbeta += sum g
__global__ __launch_bounds__(256) void bn_finish_bwd_kernel(const float* __restrict__ part, float* __restrict__ sums, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int C, int nsplit, int segments, size_t plane) {
  __shared__ float red[512];
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4), lane = threadIdx.x & 15;
  const bool owner = lane == 0 && c < C;
  float dg = 0.f, db = 0.f;
  for (int seg = 0; seg < segments; ++seg) {
    float s, q;
    fold_partials(part, plane, seg, nsplit, C, c, lane, red, s, q);
    if (!owner) continue;
    sums[(seg * 2) * C + c] = s;
    sums[(seg * 2 + 1) * C + c] = q;
    db += s; dg += q;
  }
  if (owner && dgamma) dgamma[c] += dg;
  if (owner && dbeta) dbeta[c] += db;
}

// dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat))  [training]   or   gamma * rstd * g  [eval: statistics are constants]
// (+ extra: a second gradient reaching x, e.g. the identity shortcut of a pre-activation block)
// GN (GroupNorm): mean / rstd are per (image, channel) copies of the group statistics and sums holds the GROUP means
// M1 = mean over the group of gamma * g, M2 = mean of gamma * g * xhat:  dx = rstd * (gamma * g - M1 - xhat * M2)
template <bool GN>
__global__ __launch_bounds__(256) void bn_dx_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ sums, const bf16* __restrict__ extra, bf16* __restrict__ dx,
                                                    int rows_per_seg, int C, int relu, int training, int per) {
  const Slab t = slab_of(rows_per_seg, C, per);
  const int seg = blockIdx.y;
  const float inv_n = 1.f / (float)rows_per_seg;
  for (int c0 = 0; c0 < t.c8n; c0 += t.ncol) {
    if (t.lane >= t.nlane || c0 + t.col >= t.c8n) continue;
    const int c = (c0 + t.col) * 8;
    float mu[8], rs[8], ga[8], be[8], k1[8], k2[8], k3[8];
    load8(mean + seg * C + c, mu); load8(rstd + seg * C + c, rs); load8(gamma + c, ga); load8(beta + c, be);
    load8(sums + (seg * 2) * C + c, k2); load8(sums + (seg * 2 + 1) * C + c, k3);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      k1[j] = ga[j] * rs[j];
      if (!GN) {
        k2[j] = training ? k2[j] * inv_n : 0.f;          // mean(g)
        k3[j] = training ? k3[j] * inv_n : 0.f;          // mean(g * xhat)
      }
    }
    for (int r = t.r0 + t.lane; r < t.r1; r += t.nlane) {
      const size_t off = (t.base + r) * C + c;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + off);
      const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(dy + off);
      bf16x8 e8;
      if (extra) e8 = *reinterpret_cast<const bf16x8*>(extra + off);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (bf2f(v[j]) - mu[j]) * rs[j];
        float g = bf2f(g8[j]);
        if (relu && xh * ga[j] + be[j] <= 0.f) g = 0.f;
        float d = GN ? k1[j] * g - rs[j] * (k2[j] + xh * k3[j]) : k1[j] * (g - k2[j] - xh * k3[j]);
        if (extra) d += bf2f(e8[j]);
        o[j] = f2bf(d);
      }
      *reinterpret_cast<bf16x8*>(dx + off) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------- GroupNorm (BiT towers)
// GroupNormAct (timm layers/norm_act.py: nn.GroupNorm(32, C) + ReLU) on NHWC rows: every image is one "segment" of the BatchNorm
// passes above, so bn_partial_kernel delivers per (image, slice, channel) sums; the kernels below fold them over the slices and then
// over the channels of a group.  Everything here touches [images][C] floats: a few microseconds; fixed orders (deterministic).
__global__ __launch_bounds__(256) void gn_fold_kernel(const float* __restrict__ part, float* __restrict__ sums, int C, int nsplit, size_t plane,
                                                      int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int seg = idx / C, c = idx - seg * C;
  float s = 0.f, q = 0.f;
  for (int i = 0; i < nsplit; ++i) { s += part[((size_t)seg * nsplit + i) * C + c]; q += part[plane + ((size_t)seg * nsplit + i) * C + c]; }
  sums[((size_t)seg * 2) * C + c] = s;
  sums[((size_t)seg * 2 + 1) * C + c] = q;
}

// forward: mean / rstd of (image, group) from sums = (sum x, sum x^2) per (image, channel), written once per channel of the group
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ sums, float* __restrict__ mean, float* __restrict__ rstd, int C,
                                                       int groups, int rows_per_seg, float eps, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int seg = idx / groups, g = idx - seg * groups, cg = C / groups, c0 = g * cg;
  float s = 0.f, q = 0.f;
  for (int c = c0; c < c0 + cg; ++c) { s += sums[((size_t)seg * 2) * C + c]; q += sums[((size_t)seg * 2 + 1) * C + c]; }
  const float n = (float)cg * (float)rows_per_seg, mu = s / n;
  float var = q / n - mu * mu;
  var = var < 0.f ? 0.f : var;
  const float rs = rsqrtf(var + eps);
  for (int c = c0; c < c0 + cg; ++c) { mean[(size_t)seg * C + c] = mu; rstd[(size_t)seg * C + c] = rs; }
}

// backward: gmean[image][0][c] = M1, [1][c] = M2 of the channel's group from sums = (sum g, sum g * xhat) per (image, channel)
__global__ __launch_bounds__(256) void gn_bwd_group_kernel(const float* __restrict__ sums, const float* __restrict__ gamma, float* __restrict__ gmean,
                                                           int C, int groups, int rows_per_seg, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int seg = idx / groups, g = idx - seg * groups, cg = C / groups, c0 = g * cg;
  float m1 = 0.f, m2 = 0.f;
  for (int c = c0; c < c0 + cg; ++c) { m1 += gamma[c] * sums[((size_t)seg * 2) * C + c]; m2 += gamma[c] * sums[((size_t)seg * 2 + 1) * C + c]; }
  const float inv_n = 1.f / ((float)cg * (float)rows_per_seg);
  m1 *= inv_n; m2 *= inv_n;
  for (int c = c0; c < c0 + cg; ++c) { gmean[((size_t)seg * 2) * C + c] = m1; gmean[((size_t)seg * 2 + 1) * C + c] = m2; }
}

// dgamma[c] += sum over the images of sum g * xhat, dbeta[c] += sum g
__global__ __launch_bounds__(256) void gn_dparam_kernel(const float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                                        int segments) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float db = 0.f, dg = 0.f;
  for (int seg = 0; seg < segments; ++seg) { db += sums[((size_t)seg * 2) * C + c]; dg += sums[((size_t)seg * 2 + 1) * C + c]; }
  if (dgamma) dgamma[c] += dg;
  if (dbeta) dbeta[c] += db;
}

// ------------------------------------------------------------------------------------------ stem patch gather
// cols[m][(ky*k + kx)*C + c] = images[b, c, oy*stride + ky - pad, ox*stride + kx - pad] (fp32 NCHW -> bf16), zero outside the
// image and in the padding columns [k*k*C, Kp)
// (KC, CC > 0: the kernel size and channel count as compile-time constants -- the timm stem is 7 x 7 over 3 channels: the four
// divisions per gathered element become multiplications; the run-time form spent its time in them, 3.1 ms for 64 images of 800 x 800)
template <int KC, int CC>
__global__ __launch_bounds__(256) void patches_nchw_kernel(const float* __restrict__ img, bf16* __restrict__ cols, int C_, int H, int W, int Ho,
                                                           int Wo, int k_, int stride, int pad, int Kp, size_t total) {
  const int C = CC > 0 ? CC : C_, k = KC > 0 ? KC : k_;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // over m * Kp/8: one 16-byte store of 8 columns per thread
  if (idx >= total) return;
  const int k8n = Kp >> 3, kk0 = (int)(idx % k8n) * 8;
  const size_t m = idx / k8n;
  const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho);
  const size_t b = m / ((size_t)Wo * Ho);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kk = kk0 + j;
    float v = 0.f;
    if (kk < k * k * C) {
      const int c = kk % C, t = kk / C, ky = t / k, kx = t % k;
      const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = img[((b * C + c) * H + iy) * (size_t)W + ix];
    }
    o[j] = f2bf(v);
  }
  *reinterpret_cast<bf16x8*>(cols + m * Kp + kk0) = o;
}

// The timm stem (7 x 7, stride 2, pad 3, 3 channels, Kp = 152) through LDS: a workgroup owns 64 consecutive output pixels of one output
// row.  Their 7 input rows x 133 columns x 3 planes are read ONCE, contiguously along x (the gather form above reads every input value
// ~12 times in 4-byte pieces scattered over 21 row segments per pixel), and the 64 x 152 patch rows leave as one contiguous 19 KB run
// of 16-byte stores.
__global__ __launch_bounds__(256) void patches_stem7_kernel(const float* __restrict__ img, bf16* __restrict__ cols, int H, int W, int Ho, int Wo) {
  constexpr int K = 7, C = 3, PX = 64, IW = 2 * (PX - 1) + K, Kp = 152, NCH = Kp / 8;       // IW = 133 input columns per segment
  __shared__ bf16 tile[C][K][IW + 3];
  const int seg = blockIdx.x, oy = blockIdx.y, b = blockIdx.z;
  const int ox0 = seg * PX, ix0 = 2 * ox0 - 3, iy0 = 2 * oy - 3;
  for (int i = threadIdx.x; i < C * K * IW; i += 256) {
    const int x = i % IW, r = i / IW, ky = r % K, c = r / K;
    const int iy = iy0 + ky, ix = ix0 + x;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = img[(((size_t)b * C + c) * H + iy) * W + ix];
    tile[c][ky][x] = f2bf(v);
  }
  __syncthreads();
  const int npx = min(PX, Wo - ox0);
  bf16* const out = cols + (((size_t)b * Ho + oy) * Wo + ox0) * Kp;
  for (int i = threadIdx.x; i < npx * NCH; i += 256) {
    const int px = i / NCH, j = i - px * NCH;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = j * 8 + e;                     // (ky * 7 + kx) * 3 + c; columns 147 .. 151 are padding
      const int t = kk / C, c = kk - t * C, ky = t / K, kx = t - ky * K;
      o[e] = kk < K * K * C ? tile[c][ky][2 * px + kx] : f2bf(0.f);
    }
    *reinterpret_cast<bf16x8*>(out + (size_t)i * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------- MaxPool 3x3 / 2 / pad 1
// y[b, oy, ox, c] = max over the 3x3 window at (2oy-1, 2ox-1); arg = window position (ky*3+kx) of the FIRST maximum in scan
// order (the element PyTorch's max_pool2d backward routes the gradient to)
// pad_zero: the window positions outside the image count as ZEROS (timm's 'fixed' stem of the BiT towers: ConstantPad2d(1, 0.) in
// front of MaxPool2d(3, 2, padding 0)); a recorded maximum that is such a zero has no pixel, so its gradient goes nowhere
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, uint8_t* __restrict__ arg, int H, int W,
                                                          int C, int Ho, int Wo, int pad_zero, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t m = idx / c8n;
  const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho);
  const size_t b = m / ((size_t)Wo * Ho);
  float best[8];
  int at[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; at[j] = 0; }
  bool first = true;
  for (int t = 0; t < 9; ++t) {
    const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
    const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
    if (!inside && !pad_zero) continue;
    bf16x8 v;
    if (inside) v = *reinterpret_cast<const bf16x8*>(x + ((b * H + iy) * W + ix) * C + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = inside ? bf2f(v[j]) : 0.f;
      if (first || f > best[j] || f != f) { best[j] = f; at[j] = t; }
    }
    first = false;
  }
  bf16x8 o;
  uint8_t a8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { o[j] = f2bf(best[j]); a8[j] = (uint8_t)at[j]; }
  *reinterpret_cast<bf16x8*>(y + m * C + c) = o;
  *reinterpret_cast<uint2*>(arg + m * C + c) = *reinterpret_cast<const uint2*>(a8);
}

// dx[b, iy, ix, c] = sum of dy over the (at most 4) windows whose recorded maximum is this pixel (gather form)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16* __restrict__ dy, const uint8_t* __restrict__ arg, bf16* __restrict__ dx, int H,
                                                          int W, int C, int Ho, int Wo, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t pix = idx / c8n;
  const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
  const size_t b = pix / ((size_t)W * H);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int t = 0; t < 9; ++t) {
    const int ny = iy + 1 - t / 3, nx = ix + 1 - t % 3;          // 2*oy = ny, 2*ox = nx
    if (ny < 0 || nx < 0 || (ny & 1) || (nx & 1)) continue;
    const int oy = ny >> 1, ox = nx >> 1;
    if (oy >= Ho || ox >= Wo) continue;
    const size_t m = (b * Ho + oy) * Wo + ox;
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + m * C + c);
    const uint2 a2 = *reinterpret_cast<const uint2*>(arg + m * C + c);
    const uint8_t* a8 = reinterpret_cast<const uint8_t*>(&a2);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (a8[j] == t) acc[j] += bf2f(g[j]);
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j]);
  *reinterpret_cast<bf16x8*>(dx + pix * C + c) = o;
}

// ---------------------------------------------------------------------------------- strided 1x1 convolution rows
// y[b, oy, ox, :] = x[b, oy*s, ox*s, :]
__global__ __launch_bounds__(256) void subsample_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int H, int W, int C, int Ho, int Wo,
                                                            int stride, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t m = idx / c8n;
  const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho);
  const size_t b = m / ((size_t)Wo * Ho);
  *reinterpret_cast<bf16x8*>(y + m * C + c) = *reinterpret_cast<const bf16x8*>(x + ((b * H + oy * stride) * W + ox * stride) * C + c);
}

// dx[b, iy, ix, :] = (base ? base[b, iy, ix, :] : 0) + (iy, ix on the stride grid ? dy[b, iy/s, ix/s, :] : 0)
__global__ __launch_bounds__(256) void subsample_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ base, bf16* __restrict__ dx, int H,
                                                            int W, int C, int Ho, int Wo, int stride, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t pix = idx / c8n;
  const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
  const size_t b = pix / ((size_t)W * H);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (base) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + pix * C + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bf2f(v[j]);
  }
  if (iy % stride == 0 && ix % stride == 0 && iy / stride < Ho && ix / stride < Wo) {
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + ((b * Ho + iy / stride) * Wo + ix / stride) * C + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += bf2f(g[j]);
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j]);
  *reinterpret_cast<bf16x8*>(dx + pix * C + c) = o;
}

// ------------------------------------------------------------------------------------- plain weight re-layout
// what[o][t*Cgp + c] = w[o][c][t] (c < Cg) else 0, columns [kk*Cgp, ldw) zero      (PyTorch conv weight -> tap-major bf16)
__global__ __launch_bounds__(256) void weight_pack_kernel(const float* __restrict__ w, bf16* __restrict__ what, int Cg, int kk, int Cgp, int ldw,
                                                          size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int i = (int)(idx % ldw);
  const size_t o = idx / ldw;
  float v = 0.f;
  if (i < kk * Cgp) {
    const int t = i / Cgp, c = i % Cgp;
    if (c < Cg) v = w[(o * Cg + c) * kk + t];
  }
  what[idx] = f2bf(v);
}

// dw[o][c][t] += dwhat[o][t*Cgp + c]
__global__ __launch_bounds__(256) void weight_unpack_grad_kernel(const float* __restrict__ dwhat, float* __restrict__ dw, int Cg, int kk, int Cgp,
                                                                 int ldw, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // over Cout * Cg * kk
  if (idx >= total) return;
  const int t = (int)(idx % kk), c = (int)((idx / kk) % Cg);
  const size_t o = idx / ((size_t)kk * Cg);
  dw[idx] += dwhat[o * ldw + t * Cgp + c];
}

// rows per block of the element-wise passes: 16 rows per row lane
inline int slab_rows(int C) { const int c8n = C >> 3, ncol = c8n < 256 ? c8n : 256; return (256 / ncol) * 16; }
inline int bn_nsplit(int rows_per_seg) { int n = rows_per_seg / 64; return n < 1 ? 1 : (n > 512 ? 512 : n); }   // >= 1024 blocks on big maps
// GroupNorm: one segment per image, so the images already spread the work -- ~4096 blocks in all instead of 512 per image (64 images of
// 200 x 200: 625 rows per block instead of 78, 8 MB of partial sums instead of 67)
inline int gn_nsplit(int rows_per_image, int images) {
  int cap = 4096 / images;
  cap = cap < 8 ? 8 : cap;
  int n = rows_per_image / 64;
  return n < 1 ? 1 : (n > cap ? cap : n);
}

}  // namespace

// workspace of ia_bn_act_fwd / ia_bn_act_bwd: two planes of [segments][nsplit][C] partial sums + [segments][2][C] finished sums
extern "C" size_t ia_bn_act_workspace_bytes(int rows, int C, int segments) {
  if (rows <= 0 || C <= 0 || segments <= 0 || rows % segments) return 0;
  const int ns = bn_nsplit(rows / segments);
  return ((size_t)2 * segments * ns * C + (size_t)2 * segments * C) * sizeof(float);
}

// BatchNormAct2d (timm layers/norm_act.py: nn.BatchNorm2d followed by ReLU) over NHWC rows x [rows, C] bf16.  The rows are
// `segments` equal consecutive runs, each normalised with its own batch statistics (= that many separate forward calls).
// training != 0: batch statistics; running_mean / running_var [C] (may be NULL) are updated in place with `momentum`, segment
// by segment.  training == 0: the running statistics are used.  mean / rstd [segments][C] fp32 are written for the backward.
extern "C" int ia_bn_act_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y, float* mean,
                             float* rstd, int rows, int C, int segments, float eps, float momentum, int training, int relu, void* workspace,
                             size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || C <= 0 || (C & 7) || segments <= 0 || rows % segments) return IA_ERR_ARG;
  const int rps = rows / segments;
  if (training) {
    if (!workspace || workspace_bytes < ia_bn_act_workspace_bytes(rows, C, segments)) return IA_ERR_WORKSPACE;
    const int ns = bn_nsplit(rps);
    const size_t plane = (size_t)segments * ns * C;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(ns, segments), dim3(256), 0, stream, (const bf16*)x, (const bf16*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)workspace, rps, C, ns, 0, plane);
    hipLaunchKernelGGL(bn_finish_fwd_kernel, dim3((C + 15) / 16), dim3(256), 0, stream, (const float*)workspace, mean, rstd, running_mean,
                       running_var, C, ns, segments, rps, eps, momentum, plane);
  } else {
    if (!running_mean || !running_var) return IA_ERR_ARG;
    hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, (const float*)running_mean, (const float*)running_var, mean,
                       rstd, C, segments, eps);
  }
  const int per = slab_rows(C);
  hipLaunchKernelGGL(bn_apply_kernel, dim3((rps + per - 1) / per, segments), dim3(256), 0, stream, (const bf16*)x, (const float*)mean,
                     (const float*)rstd, gamma, beta, (bf16*)y, rps, C, relu, per);
  return ia_check_launch();
}

// dx [rows, C] bf16 (+ extra [rows, C] bf16 if not NULL), dgamma / dbeta [C] fp32 accumulated (either may be NULL)
extern "C" int ia_bn_act_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                             const void* extra, void* dx, float* dgamma, float* dbeta, int rows, int C, int segments, int training, int relu,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !x || !gamma || !beta || !mean || !rstd || !dx || rows <= 0 || C <= 0 || (C & 7) || segments <= 0 || rows % segments) return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_bn_act_workspace_bytes(rows, C, segments)) return IA_ERR_WORKSPACE;
  const int rps = rows / segments, ns = bn_nsplit(rps);
  const size_t plane = (size_t)segments * ns * C;
  float* sums = (float*)workspace + 2 * plane;
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(ns, segments), dim3(256), 0, stream, (const bf16*)x, (const bf16*)dy, mean, rstd, gamma, beta,
                     (float*)workspace, rps, C, ns, relu, plane);
  hipLaunchKernelGGL(bn_finish_bwd_kernel, dim3((C + 15) / 16), dim3(256), 0, stream, (const float*)workspace, sums, dgamma, dbeta, C, ns, segments,
                     plane);
  const int per = slab_rows(C);
  hipLaunchKernelGGL(bn_dx_kernel<false>, dim3((rps + per - 1) / per, segments), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, mean, rstd, gamma,
                     beta, (const float*)sums, (const bf16*)extra, (bf16*)dx, rps, C, relu, training, per);
  return ia_check_launch();
}

// workspace of ia_gn_act_fwd / ia_gn_act_bwd: two planes of [images][nsplit][C] partial sums + [images][2][C] folded sums + [images][2][C]
// group means
extern "C" size_t ia_gn_act_workspace_bytes(int rows, int C, int images) {
  if (rows <= 0 || C <= 0 || images <= 0 || rows % images) return 0;
  const int ns = gn_nsplit(rows / images, images);
  return ((size_t)2 * images * ns * C + (size_t)4 * images * C) * sizeof(float);
}

// GroupNormAct (timm layers/norm_act.py GroupNormAct: nn.GroupNorm(groups, C, eps) followed by ReLU -- the norm layer of the BiT
// `resnetv2_*_bit*` towers, timm resnetv2.py) over NHWC rows x [rows, C] bf16 of `images` images (rows / images pixels each): biased
// statistics per (image, group of C / groups channels); the same arithmetic in training and eval mode.  mean / rstd [images][C] fp32
// (the group's value repeated for each of its channels) are written for the backward.  C % 8 == 0, C % groups == 0.
extern "C" int ia_gn_act_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int C, int images,
                             int groups, float eps, int relu, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || C <= 0 || (C & 7) || images <= 0 || images > 65535 || rows % images ||
      groups <= 0 || C % groups)
    return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_gn_act_workspace_bytes(rows, C, images)) return IA_ERR_WORKSPACE;
  const int rps = rows / images, ns = gn_nsplit(rps, images);
  const size_t plane = (size_t)images * ns * C;
  float* sums = (float*)workspace + 2 * plane;
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(ns, images), dim3(256), 0, stream, (const bf16*)x, (const bf16*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)workspace, rps, C, ns, 0, plane);
  hipLaunchKernelGGL(gn_fold_kernel, dim3(blocks_of((size_t)images * C)), dim3(256), 0, stream, (const float*)workspace, sums, C, ns, plane, images * C);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(blocks_of((size_t)images * groups)), dim3(256), 0, stream, (const float*)sums, mean, rstd, C, groups, rps, eps,
                     images * groups);
  const int per = slab_rows(C);
  hipLaunchKernelGGL(bn_apply_kernel, dim3((rps + per - 1) / per, images), dim3(256), 0, stream, (const bf16*)x, (const float*)mean,
                     (const float*)rstd, gamma, beta, (bf16*)y, rps, C, relu, per);
  return ia_check_launch();
}

// dx [rows, C] bf16 (+ extra [rows, C] bf16 if not NULL), dgamma / dbeta [C] fp32 accumulated (either may be NULL)
extern "C" int ia_gn_act_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                             const void* extra, void* dx, float* dgamma, float* dbeta, int rows, int C, int images, int groups, int relu,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !x || !gamma || !beta || !mean || !rstd || !dx || rows <= 0 || C <= 0 || (C & 7) || images <= 0 || images > 65535 || rows % images ||
      groups <= 0 || C % groups)
    return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_gn_act_workspace_bytes(rows, C, images)) return IA_ERR_WORKSPACE;
  const int rps = rows / images, ns = gn_nsplit(rps, images);
  const size_t plane = (size_t)images * ns * C;
  float* sums = (float*)workspace + 2 * plane;
  float* gmean = sums + (size_t)2 * images * C;
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(ns, images), dim3(256), 0, stream, (const bf16*)x, (const bf16*)dy, mean, rstd, gamma, beta,
                     (float*)workspace, rps, C, ns, relu, plane);
  hipLaunchKernelGGL(gn_fold_kernel, dim3(blocks_of((size_t)images * C)), dim3(256), 0, stream, (const float*)workspace, sums, C, ns, plane, images * C);
  hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(blocks_of((size_t)images * groups)), dim3(256), 0, stream, (const float*)sums, gamma, gmean, C, groups, rps,
                     images * groups);
  if (dgamma || dbeta)
    hipLaunchKernelGGL(gn_dparam_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, (const float*)sums, dgamma, dbeta, C, images);
  const int per = slab_rows(C);
  hipLaunchKernelGGL(bn_dx_kernel<true>, dim3((rps + per - 1) / per, images), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, mean, rstd, gamma,
                     beta, (const float*)gmean, (const bf16*)extra, (bf16*)dx, rps, C, relu, 1, per);
  return ia_check_launch();
}

// cols [B*Ho*Wo, Kp] bf16 patch matrix of a k x k / stride / pad convolution read straight from NCHW fp32 images
// (Ho = (H + 2 pad - k) / stride + 1); column (ky*k + kx)*C + c, columns >= k*k*C zero.  Kp % 8 == 0.
extern "C" int ia_patches_nchw(const float* images, void* cols, int B, int C, int H, int W, int k, int stride, int pad, int Kp, hipStream_t stream) {
  (void)hipGetLastError();
  if (!images || !cols || B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0 || Kp < k * k * C || (Kp & 7)) return IA_ERR_ARG;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return IA_ERR_ARG;
  const size_t total = (size_t)B * Ho * Wo * (Kp >> 3);
  if (k == 7 && C == 3 && stride == 2 && pad == 3 && Kp == 152 && Ho <= 65535 && B <= 65535)
    hipLaunchKernelGGL(patches_stem7_kernel, dim3((Wo + 63) / 64, Ho, B), dim3(256), 0, stream, images, (bf16*)cols, H, W, Ho, Wo);
  else if (k == 7 && C == 3)
    hipLaunchKernelGGL((patches_nchw_kernel<7, 3>), dim3(blocks_of(total)), dim3(256), 0, stream, images, (bf16*)cols, C, H, W, Ho, Wo, k, stride, pad, Kp, total);
  else
    hipLaunchKernelGGL((patches_nchw_kernel<0, 0>), dim3(blocks_of(total)), dim3(256), 0, stream, images, (bf16*)cols, C, H, W, Ho, Wo, k, stride, pad, Kp, total);
  return ia_check_launch();
}

// MaxPool2d(3, stride 2, padding 1) on NHWC rows: y [B*Ho*Wo, C] bf16, arg [B*Ho*Wo, C] u8 (window position of the maximum)
extern "C" int ia_maxpool3s2_fwd_ex(const void* x, void* y, uint8_t* arg, int B, int H, int W, int C, int pad_zero, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !y || !arg || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return IA_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t total = (size_t)B * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, arg, H, W, C, Ho, Wo, pad_zero, total);
  return ia_check_launch();
}
extern "C" int ia_maxpool3s2_fwd(const void* x, void* y, uint8_t* arg, int B, int H, int W, int C, hipStream_t stream) {
  return ia_maxpool3s2_fwd_ex(x, y, arg, B, H, W, C, 0, stream);
}

extern "C" int ia_maxpool3s2_bwd(const void* dy, const uint8_t* arg, void* dx, int B, int H, int W, int C, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !arg || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return IA_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t total = (size_t)B * H * W * (C >> 3);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)dy, arg, (bf16*)dx, H, W, C, Ho, Wo, total);
  return ia_check_launch();
}

// rows of a strided 1x1 convolution: y [B*Ho*Wo, C] = x[b, oy*stride, ox*stride, :], Ho = (H-1)/stride + 1
extern "C" int ia_rows_subsample_fwd(const void* x, void* y, int B, int H, int W, int C, int stride, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || stride < 1) return IA_ERR_ARG;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const size_t total = (size_t)B * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(subsample_fwd_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, H, W, C, Ho, Wo, stride, total);
  return ia_check_launch();
}

// dx [B*H*W, C] = base (may be NULL = zeros; may alias dx) + dy scattered onto the stride grid
extern "C" int ia_rows_subsample_bwd(const void* dy, const void* base, void* dx, int B, int H, int W, int C, int stride, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || stride < 1) return IA_ERR_ARG;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const size_t total = (size_t)B * H * W * (C >> 3);
  hipLaunchKernelGGL(subsample_bwd_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)base, (bf16*)dx, H, W, C, Ho, Wo,
                     stride, total);
  return ia_check_launch();
}

// what [Cout][ldw] bf16 (tap-major, channels padded to Cgp, row padded to ldw) from the PyTorch weight w [Cout][Cg][kk] fp32
extern "C" int ia_conv_weight_pack(const float* w, void* what, int Cout, int Cg, int kk, int Cgp, int ldw, hipStream_t stream) {
  (void)hipGetLastError();
  if (!w || !what || Cout <= 0 || Cg <= 0 || kk <= 0 || Cgp < Cg || ldw < kk * Cgp) return IA_ERR_ARG;
  const size_t total = (size_t)Cout * ldw;
  hipLaunchKernelGGL(weight_pack_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, w, (bf16*)what, Cg, kk, Cgp, ldw, total);
  return ia_check_launch();
}

// dw [Cout][Cg][kk] fp32 += dwhat [Cout][ldw] fp32 (same layout as ia_conv_weight_pack writes)
extern "C" int ia_conv_weight_unpack_grad(const float* dwhat, float* dw, int Cout, int Cg, int kk, int Cgp, int ldw, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dwhat || !dw || Cout <= 0 || Cg <= 0 || kk <= 0 || Cgp < Cg || ldw < kk * Cgp) return IA_ERR_ARG;
  const size_t total = (size_t)Cout * Cg * kk;
  hipLaunchKernelGGL(weight_unpack_grad_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, dwhat, dw, Cg, kk, Cgp, ldw, total);
  return ia_check_launch();
}
