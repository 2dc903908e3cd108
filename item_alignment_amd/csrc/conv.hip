// NHWC convolution tower pieces for the norm-free ResNets (ECA-NFNet; reference src/models/image.py:191-199 -> timm
// NormFreeNet / NormFreeBlock / ScaledStdConv2d / EcaModule): activations are [B, H, W, C] bf16 (= [B*H*W, C] rows),
// so a 1x1 convolution IS the bf16 MFMA GEMM of gemm.hip and a (grouped) 3x3 convolution is a patch gather
// (im2col, HBM-bound) followed by one GEMM per channel group; weights arrive standardised and laid out [Cout][tap][Cin/g].
// Everything else here is an HBM-bound element-wise / reduction kernel with 16-byte accesses.
#include "common.h"
#include "../../include/itemalign.h"

namespace {

// ----------------------------------------------------------------------------------- layout conversion
// images [B, C, H, W] fp32 (what the collate functions produce, reference data.py:92) -> [B, H, W, Cp] bf16, channels >= C zero
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, bf16* __restrict__ out, int C, int HW, int Cp,
                                                           size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % Cp);
  const size_t pix = idx / Cp, b = pix / HW, hw = pix % HW;
  out[idx] = f2bf(c < C ? in[(b * C + c) * HW + hw] : 0.f);
}

// Cp = 8 (the stem's 3 -> 8 padded channels): one thread per pixel -- C plane reads that are contiguous across the wave (256 B per
// plane) and one 16-byte store, instead of one 2-byte store per thread and plane-strided 4-byte reads (0.84 -> ~0.2 ms for 64 images
// of 800 x 800)
__global__ __launch_bounds__(256) void nchw_to_nhwc8_kernel(const float* __restrict__ in, bf16* __restrict__ out, int C, int HW, size_t npix) {
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  const size_t b = pix / HW, hw = pix % HW;
  bf16x8 v;
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = f2bf(c < C ? in[(b * C + c) * HW + hw] : 0.f);
  *reinterpret_cast<bf16x8*>(out + pix * 8) = v;
}

// ------------------------------------------------------------------------------------------- im2col
// cols[m][g*9*Cg + t*Cg + c] = x[b, oy*s + ky - 1, ox*s + kx - 1, g*Cg + c]   (t = ky*3 + kx, zero outside the image)
__global__ __launch_bounds__(256) void im2col3_kernel(const bf16* __restrict__ x, bf16* __restrict__ cols, int H, int W, int C, int Cg,
                                                      int Ho, int Wo, int stride, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3;
  const int c = (int)(idx % c8n) * 8;
  const int t = (int)((idx / c8n) % 9);
  const size_t m = idx / ((size_t)c8n * 9);
  const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho);
  const size_t b = m / ((size_t)Wo * Ho);
  const int iy = oy * stride + t / 3 - 1, ix = ox * stride + t % 3 - 1;
  bf16x8 v;
  if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const bf16x8*>(x + ((b * H + iy) * W + ix) * C + c);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f2bf(0.f);
  }
  const int g = c / Cg, cg = c % Cg;
  *reinterpret_cast<bf16x8*>(cols + m * (size_t)(9 * C) + (size_t)g * 9 * Cg + t * Cg + cg) = v;
}

// dx[b, y, x, c] = sum over taps of dcols at the output positions that read this pixel (gather form: deterministic)
__global__ __launch_bounds__(256) void col2im3_kernel(const bf16* __restrict__ dcols, bf16* __restrict__ dx, int H, int W, int C, int Cg,
                                                      int Ho, int Wo, int stride, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3;
  const int c = (int)(idx % c8n) * 8;
  const size_t pix = idx / c8n;
  const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
  const size_t b = pix / ((size_t)W * H);
  const int g = c / Cg, cg = c % Cg;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ny = iy + 1 - t / 3, nx = ix + 1 - t % 3;
    if (ny < 0 || nx < 0 || (ny % stride) || (nx % stride)) continue;
    const int oy = ny / stride, ox = nx / stride;
    if (oy >= Ho || ox >= Wo) continue;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(dcols + ((b * Ho + oy) * Wo + ox) * (size_t)(9 * C) + (size_t)g * 9 * Cg + t * Cg + cg);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j]);
  *reinterpret_cast<bf16x8*>(dx + pix * C + c) = o;
}

// the same gather between BORDERED layouts (stride 2): dcols rows are the pixels of [B, Ho + 2, Wo + 2], dx is [B, H + 2, W + 2, C]; only the
// interior of dx is written (its reader, the SiLU backward in front, looks at nothing else)
__global__ __launch_bounds__(256) void col2im3_s2_padded_kernel(const bf16* __restrict__ dcols, bf16* __restrict__ dx, int H, int W, int C, int Cg,
                                                                int Ho, int Wo, int y_compact, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3;
  const int c = (int)(idx % c8n) * 8;
  const size_t pix = idx / c8n;
  const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
  const size_t b = pix / ((size_t)W * H);
  const int g = c / Cg, cg = c % Cg;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ny = iy + 1 - t / 3, nx = ix + 1 - t % 3;
    if (ny < 0 || nx < 0 || (ny & 1) || (nx & 1)) continue;
    const int oy = ny >> 1, ox = nx >> 1;
    if (oy >= Ho || ox >= Wo) continue;
    const size_t orow = y_compact ? (b * Ho + oy) * Wo + ox : (b * (Ho + 2) + oy + 1) * (Wo + 2) + ox + 1;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(dcols + orow * (size_t)(9 * C) + (size_t)g * 9 * Cg + t * Cg + cg);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j]);
  *reinterpret_cast<bf16x8*>(dx + ((b * (H + 2) + iy + 1) * (size_t)(W + 2) + ix + 1) * C + c) = o;
}

// ------------------------------------------------------------------------------ weight standardisation
// ScaledStdConv2d: what[o][t*Cgp + c] = (w[o][c][t] - mean_o) * rstd_o * gain[o] * scale, statistics over the real fan-in
// (Cg*kk, biased variance, eps inside the sqrt); channels c >= Cg (padding of the 3-channel stem) are zero.  One wave per o.
__global__ __launch_bounds__(256) void ws_weight_fwd_kernel(const float* __restrict__ w, const float* __restrict__ gain,
                                                            bf16* __restrict__ what, float* __restrict__ mean, float* __restrict__ rstd,
                                                            int Cout, int Cg, int kk, int Cgp, float scale, float eps) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= Cout) return;
  const int fan = Cg * kk;
  const float* wo = w + (size_t)o * fan;
  float s = 0.f, ss = 0.f;
  for (int i = lane; i < fan; i += 64) { const float v = wo[i]; s += v; ss += v * v; }
  s = wave_sum(s); ss = wave_sum(ss);
  const float mu = s / fan;
  const float var = fmaxf(ss / fan - mu * mu, 0.f);
  const float rs = rsqrtf(var + eps);
  if (lane == 0) { mean[o] = mu; rstd[o] = rs; }
  const float a = rs * gain[o] * scale;
  for (int i = lane; i < kk * Cgp; i += 64) {
    const int t = i / Cgp, c = i % Cgp;
    what[(size_t)o * kk * Cgp + i] = f2bf(c < Cg ? (wo[c * kk + t] - mu) * a : 0.f);
  }
}

// dgain[o] (+)= scale * sum_i g_i xhat_i ; dw_i (+)= gain*scale*rstd * (g_i - mean(g) - xhat_i * mean(g * xhat))
__global__ __launch_bounds__(256) void ws_weight_bwd_kernel(const float* __restrict__ dwhat, const float* __restrict__ w,
                                                            const float* __restrict__ gain, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dw, float* __restrict__ dgain,
                                                            int Cout, int Cg, int kk, int Cgp, float scale) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= Cout) return;
  const int fan = Cg * kk;
  const float* wo = w + (size_t)o * fan;
  const float* go = dwhat + (size_t)o * kk * Cgp;
  const float mu = mean[o], rs = rstd[o];
  float sg = 0.f, sgx = 0.f;
  for (int i = lane; i < fan; i += 64) {
    const int c = i / kk, t = i % kk;
    const float g = go[t * Cgp + c], xh = (wo[i] - mu) * rs;
    sg += g; sgx += g * xh;
  }
  sg = wave_sum(sg); sgx = wave_sum(sgx);
  if (lane == 0 && dgain) dgain[o] += scale * sgx;
  if (!dw) return;
  const float a = gain[o] * scale * rs, mg = sg / fan, mgx = sgx / fan;
  for (int i = lane; i < fan; i += 64) {
    const int c = i / kk, t = i % kk;
    const float g = go[t * Cgp + c], xh = (wo[i] - mu) * rs;
    dw[(size_t)o * fan + i] += a * (g - mg - xh * mgx);
  }
}

// ------------------------------------------------------------------------------------------- SiLU
__global__ __launch_bounds__(256) void silu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t n8, float scale) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float a = bf2f(v[j]); o[j] = f2bf(a / (1.f + __expf(-a)) * scale); }
  *reinterpret_cast<bf16x8*>(y + i * 8) = o;
}

// dx = (dy [+ dy2]) * scale * silu'(x) [+ dadd]: dy2 = the gradient of a second consumer of y (the projected shortcut of a
// downsampling NormFreeBlock reads the same activation as conv1), dadd = the gradient of an identity shortcut that bypassed the SiLU
__global__ __launch_bounds__(256) void silu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ dy2, const bf16* __restrict__ x,
                                                       const bf16* __restrict__ dadd, bf16* __restrict__ dx, size_t n8, float scale) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + i * 8), v = *reinterpret_cast<const bf16x8*>(x + i * 8);
  bf16x8 e, g2;
  if (dadd) e = *reinterpret_cast<const bf16x8*>(dadd + i * 8);
  if (dy2) g2 = *reinterpret_cast<const bf16x8*>(dy2 + i * 8);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = bf2f(v[j]), sg = 1.f / (1.f + __expf(-a));
    const float gy = bf2f(g[j]) + (dy2 ? bf2f(g2[j]) : 0.f);
    o[j] = f2bf(gy * scale * sg * (1.f + a * (1.f - sg)) + (dadd ? bf2f(e[j]) : 0.f));
  }
  *reinterpret_cast<bf16x8*>(dx + i * 8) = o;
}

// ----------------------------------------------------------------------------------------- avg-pool
// AvgPool2d(2, stride 2, ceil_mode=True, count_include_pad=False) on NHWC (timm DownsampleAvg)
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int H, int W, int C, int Ho,
                                                           int Wo, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t m = idx / c8n;
  const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho);
  const size_t b = m / ((size_t)Wo * Ho);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  int cnt = 0;
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      const int iy = oy * 2 + dy, ix = ox * 2 + dx;
      if (iy >= H || ix >= W) continue;
      ++cnt;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((b * H + iy) * W + ix) * C + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
    }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j] / cnt);
  *reinterpret_cast<bf16x8*>(y + m * C + c) = o;
}

__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ dx, int H, int W, int C, int Ho,
                                                           int Wo, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t pix = idx / c8n;
  const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
  const size_t b = pix / ((size_t)W * H);
  const int oy = iy >> 1, ox = ix >> 1;
  const int cnt = ((oy * 2 + 1 < H) ? 2 : 1) * ((ox * 2 + 1 < W) ? 2 : 1);
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + ((b * Ho + oy) * Wo + ox) * C + c);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(g[j]) / cnt);
  *reinterpret_cast<bf16x8*>(dx + pix * C + c) = o;
}

// ------------------------------------------------------------------------------- spatial reductions
// part[b][s][c] = sum over the s-th slice of HW of x[b, hw, c] (* y[b, hw, c] if y); fixed slice order -> deterministic
__global__ __launch_bounds__(256) void spatial_sum_kernel(const bf16* __restrict__ x, const bf16* __restrict__ y, float* __restrict__ part,
                                                          int HW, int C, int nsplit) {
  // C/8 column threads x as many row lanes as fit in the block; row lanes are folded through LDS in a fixed order
  __shared__ float red[256 * 8];
  const int b = blockIdx.y, s = blockIdx.x;
  const int per = (HW + nsplit - 1) / nsplit, h0 = s * per, h1 = min(HW, h0 + per);
  const int c8n = C >> 3;
  const int ncol = c8n < 256 ? c8n : 256, nlane = 256 / ncol;
  const int col = threadIdx.x % ncol, lane = threadIdx.x / ncol;
  for (int c0 = 0; c0 < c8n; c0 += ncol) {
    const int c = (c0 + col) * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (lane < nlane && c0 + col < c8n)
      for (int hw = h0 + lane; hw < h1; hw += nlane) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((size_t)b * HW + hw) * C + c);
        if (y) {
          const bf16x8 u = *reinterpret_cast<const bf16x8*>(y + ((size_t)b * HW + hw) * C + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]) * bf2f(u[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
        }
      }
    if (nlane > 1) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
      __syncthreads();
      if (lane == 0)
        for (int l = 1; l < nlane; ++l)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += red[(l * ncol + col) * 8 + j];
    }
    if (lane == 0 && c0 + col < c8n)
#pragma unroll
      for (int j = 0; j < 8; ++j) part[((size_t)b * nsplit + s) * C + c + j] = acc[j];
  }
}

// The NormFreeBlock tail's backward meets the next block's opening activation (round 6): dtot = (dy [+ dy2]) * scale * silu'(o) [+ dadd]
// -- silu_bwd_kernel's arithmetic: the whole gradient of the block output o -- written out AND, in the same pass, part[b][s][c] = sum over
// the s-th slice of HW of dtot * x (x = the conv3 output the ECA gate scaled: the gate's gradient), which spatial_sum_kernel otherwise
// takes by reading dtot and x again.  The products use the ROUNDED dtot, i.e. exactly what the two-kernel form multiplies.  Same
// thread layout and fixed fold order as spatial_sum_kernel.
__global__ __launch_bounds__(256) void silu_bwd_dot_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ dy2, const bf16* __restrict__ o,
                                                           const bf16* __restrict__ dadd, const bf16* __restrict__ x, bf16* __restrict__ dtot,
                                                           float* __restrict__ part, int HW, int C, int nsplit, float scale) {
  __shared__ float red[256 * 8];
  const int b = blockIdx.y, s = blockIdx.x;
  const int per = (HW + nsplit - 1) / nsplit, h0 = s * per, h1 = min(HW, h0 + per);
  const int c8n = C >> 3;
  const int ncol = c8n < 256 ? c8n : 256, nlane = 256 / ncol;
  const int col = threadIdx.x % ncol, lane = threadIdx.x / ncol;
  for (int c0 = 0; c0 < c8n; c0 += ncol) {
    const int c = (c0 + col) * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (lane < nlane && c0 + col < c8n)
      for (int hw = h0 + lane; hw < h1; hw += nlane) {
        const size_t at = ((size_t)b * HW + hw) * C + c;
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + at), v = *reinterpret_cast<const bf16x8*>(o + at);
        const bf16x8 u = *reinterpret_cast<const bf16x8*>(x + at);
        bf16x8 e, g2;
        if (dadd) e = *reinterpret_cast<const bf16x8*>(dadd + at);
        if (dy2) g2 = *reinterpret_cast<const bf16x8*>(dy2 + at);
        bf16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = bf2f(v[j]), sg = 1.f / (1.f + __expf(-a));
          const float gy = bf2f(g[j]) + (dy2 ? bf2f(g2[j]) : 0.f);
          t[j] = f2bf(gy * scale * sg * (1.f + a * (1.f - sg)) + (dadd ? bf2f(e[j]) : 0.f));
          acc[j] += bf2f(t[j]) * bf2f(u[j]);
        }
        *reinterpret_cast<bf16x8*>(dtot + at) = t;
      }
    if (nlane > 1) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
      __syncthreads();
      if (lane == 0)
        for (int l = 1; l < nlane; ++l)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += red[(l * ncol + col) * 8 + j];
    }
    if (lane == 0 && c0 + col < c8n)
#pragma unroll
      for (int j = 0; j < 8; ++j) part[((size_t)b * nsplit + s) * C + c + j] = acc[j];
  }
}

__global__ __launch_bounds__(256) void spatial_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int nsplit,
                                                             float scale, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = idx / C, c = idx % C;
  float s = 0.f;
  for (int i = 0; i < nsplit; ++i) s += part[((size_t)b * nsplit + i) * C + c];
  out[idx] = s * scale;
}

// dx[b, hw, c] = g[b, c] * scale  (backward of the global average pool)
__global__ __launch_bounds__(256) void spatial_bcast_kernel(const float* __restrict__ g, bf16* __restrict__ dx, int HW, int C, float scale,
                                                            size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t b = idx / ((size_t)c8n * HW);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(g[b * C + c + j] * scale);
  *reinterpret_cast<bf16x8*>(dx + idx * 8) = o;
}

// ---------------------------------------------------------------------------------------------- ECA
// gate[b, c] = sigmoid(sum_j w[j] * pooled[b, c + j - pad])   (timm EcaModule: conv1d over the channel axis, zero padded)
__global__ __launch_bounds__(256) void eca_gate_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w,
                                                           float* __restrict__ gate, int C, int k, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = idx / C, c = idx % C, pad = (k - 1) / 2;
  float s = 0.f;
  for (int j = 0; j < k; ++j) {
    const int cc = c + j - pad;
    if (cc >= 0 && cc < C) s += w[j] * pooled[b * C + cc];
  }
  gate[idx] = 1.f / (1.f + __expf(-s));
}

// ds = dgate * gate (1 - gate); dpooled[b, c] = sum_j w[j] ds[b, c - j + pad]
__global__ __launch_bounds__(256) void eca_gate_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                           const float* __restrict__ w, float* __restrict__ dpooled, int C, int k, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = idx / C, c = idx % C, pad = (k - 1) / 2;
  float s = 0.f;
  for (int j = 0; j < k; ++j) {
    const int cc = c - j + pad;
    if (cc >= 0 && cc < C) { const float gt = gate[b * C + cc]; s += w[j] * dgate[b * C + cc] * gt * (1.f - gt); }
  }
  dpooled[idx] = s;
}

// dw[j] += sum_{b, c} ds[b, c] * pooled[b, c + j - pad]     (one 1024-thread block per tap j)
__global__ __launch_bounds__(1024) void eca_gate_wgrad_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                              const float* __restrict__ pooled, float* __restrict__ dw, int B, int C, int k) {
  __shared__ float red[16];
  const int j = blockIdx.x, pad = (k - 1) / 2;
  float s = 0.f;
  for (int idx = threadIdx.x; idx < B * C; idx += 1024) {
    const int b = idx / C, c = idx % C, cc = c + j - pad;
    if (cc >= 0 && cc < C) { const float gt = gate[idx]; s += dgate[idx] * gt * (1.f - gt) * pooled[b * C + cc]; }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i];
    dw[j] += tot;
  }
}

// pooled[b, c] = sum_k what[c][k] * mean_a[b][k] + bias[c]: the spatial mean of a 1x1 convolution's OUTPUT taken from the spatial mean of
// its INPUT (the mean over pixels commutes with a per-pixel linear map), one thread per (b, c)
__global__ __launch_bounds__(256) void eca_pool_linear_kernel(const float* __restrict__ mean_a, const bf16* __restrict__ what,
                                                              const float* __restrict__ bias, float* __restrict__ pooled, int C, int K, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = idx / C, c = idx % C;
  const float* m = mean_a + (size_t)b * K;
  const bf16* w = what + (size_t)c * K;
  float s0 = 0.f, s1 = 0.f;
  for (int k = 0; k < K; k += 8) {
    const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + k);
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(m + k), m1 = *reinterpret_cast<const f32x4*>(m + k + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s0 += bf2f(wv[j]) * m0[j]; s1 += bf2f(wv[4 + j]) * m1[j]; }
  }
  pooled[idx] = s0 + s1 + (bias ? bias[c] : 0.f);
}

// out = x * gate[b, c] * coef + shortcut      (attn_gain * ECA(x) * alpha + shortcut, timm NormFreeBlock.forward tail)
// ACT: also act = silu(out) * act_scale, the activation the NEXT block opens with (timm NormFreeBlock.forward: act1(x) * beta), taken from
// the ROUNDED out -- bit for bit what silu_fwd_kernel makes of the stored tensor -- so that tensor is not read back by a pass of its own
template <bool ACT>
__global__ __launch_bounds__(256) void scale_residual_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gate,
                                                                 const bf16* __restrict__ shortcut, bf16* __restrict__ out, bf16* __restrict__ act,
                                                                 int HW, int C, float coef, float act_scale, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t b = idx / ((size_t)c8n * HW);
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + idx * 8), sc = *reinterpret_cast<const bf16x8*>(shortcut + idx * 8);
  bf16x8 o, a;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o[j] = f2bf(bf2f(v[j]) * gate[b * C + c + j] * coef + bf2f(sc[j]));
    if (ACT) { const float r = bf2f(o[j]); a[j] = f2bf(r / (1.f + __expf(-r)) * act_scale); }
  }
  *reinterpret_cast<bf16x8*>(out + idx * 8) = o;
  if (ACT) *reinterpret_cast<bf16x8*>(act + idx * 8) = a;
}

// dx = dout * gate * coef + dpooled[b, c] / HW
__global__ __launch_bounds__(256) void scale_residual_bwd_kernel(const bf16* __restrict__ dout, const float* __restrict__ gate,
                                                                 const float* __restrict__ dpooled, bf16* __restrict__ dx, int HW, int C,
                                                                 float coef, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t b = idx / ((size_t)c8n * HW);
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(dout + idx * 8);
  const float inv = 1.f / HW;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(g[j]) * gate[b * C + c + j] * coef + dpooled[b * C + c + j] * inv);
  *reinterpret_cast<bf16x8*>(dx + idx * 8) = o;
}

// --------------------------------------------------------------------------- SiLU between padded / compact layouts
// The patch-matrix-free 3x3 convolution (ia_conv3x3_padded_*) reads [B, H+2, W+2, C] with a zero border and produces its
// output on the same padded domain (border rows hold garbage).  These SiLU kernels move between that domain and the
// compact [B, H, W, C] one: the border of a padded output is written as zero, the border of a padded input is ignored.
IA_DEV size_t pad_row(size_t b, int y, int x, int H, int W) { return (b * (H + 2) + y + 1) * (size_t)(W + 2) + x + 1; }

template <bool ACT>
__global__ __launch_bounds__(256) void silu_pad_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int H, int W, int C, float scale,
                                                           int in_padded, int out_padded, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // over output rows x C/8
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t row = idx / c8n;
  const int OW = out_padded ? W + 2 : W, OH = out_padded ? H + 2 : H;
  const int ox = (int)(row % OW), oy = (int)((row / OW) % OH);
  const size_t b = row / ((size_t)OW * OH);
  const int yy = out_padded ? oy - 1 : oy, xx = out_padded ? ox - 1 : ox;
  bf16x8 o;
  if (yy < 0 || yy >= H || xx < 0 || xx >= W) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(0.f);
  } else {
    const size_t irow = in_padded ? pad_row(b, yy, xx, H, W) : (b * H + yy) * (size_t)W + xx;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + irow * C + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float a = bf2f(v[j]); o[j] = ACT ? f2bf(a / (1.f + __expf(-a)) * scale) : v[j]; }
  }
  *reinterpret_cast<bf16x8*>(y + row * C + c) = o;
}

// dx (layout of x) = dy (layout of y) * scale * silu'(x); the border of a padded dx is zero
__global__ __launch_bounds__(256) void silu_pad_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, bf16* __restrict__ dx, int H,
                                                           int W, int C, float scale, int in_padded, int out_padded, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // over rows of x (its own layout) x C/8
  if (idx >= total) return;
  const int c8n = C >> 3, c = (int)(idx % c8n) * 8;
  const size_t row = idx / c8n;
  const int IW = in_padded ? W + 2 : W, IH = in_padded ? H + 2 : H;
  const int ix = (int)(row % IW), iy = (int)((row / IW) % IH);
  const size_t b = row / ((size_t)IW * IH);
  const int yy = in_padded ? iy - 1 : iy, xx = in_padded ? ix - 1 : ix;
  bf16x8 o;
  if (yy < 0 || yy >= H || xx < 0 || xx >= W) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(0.f);
  } else {
    const size_t orow = out_padded ? pad_row(b, yy, xx, H, W) : (b * H + yy) * (size_t)W + xx;
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + orow * C + c), v = *reinterpret_cast<const bf16x8*>(x + row * C + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = bf2f(v[j]), sg = 1.f / (1.f + __expf(-a));
      o[j] = f2bf(bf2f(g[j]) * scale * sg * (1.f + a * (1.f - sg)));
    }
  }
  *reinterpret_cast<bf16x8*>(dx + row * C + c) = o;
}

inline unsigned blocks_of(size_t total) { return (unsigned)((total + 255) / 256); }
inline int nsplit_of(int HW) { int n = HW / 64; return n < 1 ? 1 : (n > 64 ? 64 : n); }

struct ConvGeom { int B, H, W, C, Cout, k, stride, groups, Ho, Wo, Cg, Ng, K; size_t M; };
int geom(ConvGeom& g, int B, int H, int W, int C, int Cout, int k, int stride, int groups) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0 || groups <= 0 || (k != 1 && k != 3) || stride < 1 || stride > 2) return IA_ERR_ARG;
  if (k == 1 && (stride != 1 || groups != 1)) return IA_ERR_UNSUPPORTED;
  if ((C % groups) || (Cout % groups)) return IA_ERR_ARG;
  g = ConvGeom{B, H, W, C, Cout, k, stride, groups, 0, 0, C / groups, Cout / groups, 0, 0};
  if ((g.Cg & 7) || (g.Ng & 7)) return IA_ERR_ARG;
  g.Ho = k == 1 ? H : (H + 2 - 3) / stride + 1;
  g.Wo = k == 1 ? W : (W + 2 - 3) / stride + 1;
  g.K = k * k * g.Cg;
  g.M = (size_t)B * g.Ho * g.Wo;
  // operands may exceed 2 GiB (the GEMM re-bases a 32-bit buffer window per workgroup); row counts are ints
  if (g.M >= 0x7FFFFFFFull || (size_t)B * H * W >= 0x7FFFFFFFull) return IA_ERR_ARG;
  return IA_OK;
}

}  // namespace

extern "C" int ia_nchw_to_nhwc_bf16(const float* in, void* out, int B, int C, int H, int W, int Cp, hipStream_t stream) {
  (void)hipGetLastError();
  if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C) return IA_ERR_ARG;
  const size_t total = (size_t)B * H * W * Cp;
  if (Cp == 8 && C <= 8) {
    const size_t npix = (size_t)B * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc8_kernel, dim3(blocks_of(npix)), dim3(256), 0, stream, in, (bf16*)out, C, H * W, npix);
  } else {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, in, (bf16*)out, C, H * W, Cp, total);
  }
  return ia_check_launch();
}

// workspace: the im2col buffer (3x3 only) + the split-K scratch of the weight-gradient GEMMs
extern "C" size_t ia_conv_nhwc_workspace_bytes(int B, int H, int W, int C, int Cout, int k, int stride, int groups) {
  ConvGeom g;
  if (geom(g, B, H, W, C, Cout, k, stride, groups)) return 0;
  size_t cols = k == 3 ? g.M * (size_t)(9 * C) * 2 : 0;
  cols = (cols + 255) & ~(size_t)255;
  const size_t gw = ia_gemm_workspace_bytes(g.Ng, g.K, (int)g.M, 1), cs = ia_colsum_workspace_bytes((int)g.M, Cout);
  return cols + (gw > cs ? gw : cs);
}

// y [B*Ho*Wo, Cout] = conv(x [B*H*W, C], what [Cout][k*k*Cg]) + bias
extern "C" int ia_conv_nhwc_fwd(const void* x, const void* what, const float* bias, void* y, int B, int H, int W, int C, int Cout, int k,
                                int stride, int groups, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  ConvGeom g;
  int rc = geom(g, B, H, W, C, Cout, k, stride, groups);
  if (rc) return rc;
  if (!x || !what || !y) return IA_ERR_ARG;
  const int epi = bias ? IA_EPI_BIAS : IA_EPI_NONE;
  if (k == 1) return ia_gemm_bf16(x, 0, C, what, 0, C, y, 0, Cout, (int)g.M, Cout, C, epi, bias, nullptr, 0, nullptr, 0, nullptr, 0, stream);
  if (!workspace || workspace_bytes < ia_conv_nhwc_workspace_bytes(B, H, W, C, Cout, k, stride, groups)) return IA_ERR_WORKSPACE;
  bf16* cols = (bf16*)workspace;
  const size_t total = g.M * 9 * (size_t)(C >> 3);
  hipLaunchKernelGGL(im2col3_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, cols, H, W, C, g.Cg, g.Ho, g.Wo, stride, total);
  rc = ia_check_launch();
  for (int gi = 0; gi < groups && !rc; ++gi)
    rc = ia_gemm_bf16(cols + (size_t)gi * g.K, 0, 9 * C, (const bf16*)what + (size_t)gi * g.Ng * g.K, 0, g.K, (bf16*)y + gi * g.Ng, 0, Cout,
                      (int)g.M, g.Ng, g.K, epi, bias ? bias + gi * g.Ng : nullptr, nullptr, 0, nullptr, 0, nullptr, 0, stream);
  return rc;
}

// dx [B*H*W, C] from dy [B*Ho*Wo, Cout]
extern "C" int ia_conv_nhwc_bwd_data(const void* dy, const void* what, void* dx, int B, int H, int W, int C, int Cout, int k, int stride,
                                     int groups, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  ConvGeom g;
  int rc = geom(g, B, H, W, C, Cout, k, stride, groups);
  if (rc) return rc;
  if (!dy || !what || !dx) return IA_ERR_ARG;
  if (k == 1) return ia_gemm_bf16(dy, 0, Cout, what, 1, C, dx, 0, C, (int)g.M, C, Cout, IA_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, stream);
  if (!workspace || workspace_bytes < ia_conv_nhwc_workspace_bytes(B, H, W, C, Cout, k, stride, groups)) return IA_ERR_WORKSPACE;
  bf16* dcols = (bf16*)workspace;
  for (int gi = 0; gi < groups && !rc; ++gi)
    rc = ia_gemm_bf16((const bf16*)dy + gi * g.Ng, 0, Cout, (const bf16*)what + (size_t)gi * g.Ng * g.K, 1, g.K, dcols + (size_t)gi * g.K, 0,
                      9 * C, (int)g.M, g.K, g.Ng, IA_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, stream);
  if (rc) return rc;
  const size_t total = (size_t)B * H * W * (C >> 3);
  hipLaunchKernelGGL(col2im3_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, dcols, (bf16*)dx, H, W, C, g.Cg, g.Ho, g.Wo, stride, total);
  return ia_check_launch();
}

// dwhat [Cout][k*k*Cg] fp32 (overwritten) and dbias [Cout] (+=, may be NULL) from x and dy
// cols_valid != 0: the workspace still holds the patch matrix ia_conv_nhwc_fwd left there for the same x and geometry
// (the caller kept that buffer), so the gather is not repeated
extern "C" int ia_conv_nhwc_bwd_weight(const void* x, const void* dy, float* dwhat, float* dbias, int B, int H, int W, int C, int Cout, int k,
                                       int stride, int groups, int cols_valid, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  ConvGeom g;
  int rc = geom(g, B, H, W, C, Cout, k, stride, groups);
  if (rc) return rc;
  if (!x || !dy || !dwhat) return IA_ERR_ARG;
  const size_t need = ia_conv_nhwc_workspace_bytes(B, H, W, C, Cout, k, stride, groups);
  if (need && (!workspace || workspace_bytes < need)) return IA_ERR_WORKSPACE;
  size_t cols_bytes = k == 3 ? g.M * (size_t)(9 * C) * 2 : 0;
  cols_bytes = (cols_bytes + 255) & ~(size_t)255;
  const bf16* a = (const bf16*)x;
  int lda = C;
  if (k == 3) {
    bf16* cols = (bf16*)workspace;
    if (!cols_valid) {
      const size_t total = g.M * 9 * (size_t)(C >> 3);
      hipLaunchKernelGGL(im2col3_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, cols, H, W, C, g.Cg, g.Ho, g.Wo, stride, total);
      rc = ia_check_launch();
    }
    a = cols; lda = 9 * C;
  }
  void* gws = need > cols_bytes ? (char*)workspace + cols_bytes : nullptr;
  // C2 = the bias gradient of the group's output channels: row sums of dy^T out of the same GEMM (a separate column-sum pass
  // only where the 256x256 kernel serves the shape)
  for (int gi = 0; gi < groups && !rc; ++gi)
    rc = ia_gemm_bf16((const bf16*)dy + gi * g.Ng, 1, Cout, a + (size_t)gi * g.K, 1, lda, dwhat + (size_t)gi * g.Ng * g.K, 1, g.K, g.Ng, g.K,
                      (int)g.M, IA_EPI_NONE, nullptr, nullptr, 0, dbias ? dbias + gi * g.Ng : nullptr, 0, gws, need - cols_bytes, stream);
  return rc;
}

// what [Cout][kk*Cgp] bf16 from w [Cout][Cg][kk] fp32 (PyTorch conv weight), gain [Cout]; saves mean / rstd [Cout]
extern "C" int ia_ws_conv_weight_fwd(const float* w, const float* gain, void* what, float* mean, float* rstd, int Cout, int Cg, int kk,
                                     int Cgp, float scale, float eps, hipStream_t stream) {
  (void)hipGetLastError();
  if (!w || !gain || !what || !mean || !rstd || Cout <= 0 || Cg <= 0 || kk <= 0 || Cgp < Cg) return IA_ERR_ARG;
  hipLaunchKernelGGL(ws_weight_fwd_kernel, dim3((Cout + 3) / 4), dim3(256), 0, stream, w, gain, (bf16*)what, mean, rstd, Cout, Cg, kk, Cgp, scale, eps);
  return ia_check_launch();
}

// dw [Cout][Cg][kk] and dgain [Cout] (both +=, either may be NULL) from dwhat [Cout][kk*Cgp] fp32
extern "C" int ia_ws_conv_weight_bwd(const float* dwhat, const float* w, const float* gain, const float* mean, const float* rstd, float* dw,
                                     float* dgain, int Cout, int Cg, int kk, int Cgp, float scale, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dwhat || !w || !gain || !mean || !rstd || Cout <= 0 || Cg <= 0 || kk <= 0 || Cgp < Cg) return IA_ERR_ARG;
  hipLaunchKernelGGL(ws_weight_bwd_kernel, dim3((Cout + 3) / 4), dim3(256), 0, stream, dwhat, w, gain, mean, rstd, dw, dgain, Cout, Cg, kk, Cgp, scale);
  return ia_check_launch();
}

// y = silu(x) * scale over n bf16 elements (n % 8 == 0)
extern "C" int ia_silu_fwd(const void* x, void* y, size_t n, float scale, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !y || !n || (n & 7)) return IA_ERR_ARG;
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(blocks_of(n >> 3)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, n >> 3, scale);
  return ia_check_launch();
}

// dx = dy * scale * silu'(x) (+ dadd if not NULL: gradient arriving on a second use of x, e.g. the identity shortcut)
extern "C" int ia_silu_bwd(const void* dy, const void* x, const void* dadd, void* dx, size_t n, float scale, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !x || !dx || !n || (n & 7)) return IA_ERR_ARG;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(blocks_of(n >> 3)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)nullptr, (const bf16*)x, (const bf16*)dadd,
                     (bf16*)dx, n >> 3, scale);
  return ia_check_launch();
}

extern "C" int ia_silu_bwd_sum(const void* dy, const void* dy2, const void* x, const void* dadd, void* dx, size_t n, float scale, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !dy2 || !x || !dx || !n || (n & 7)) return IA_ERR_ARG;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(blocks_of(n >> 3)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)dy2, (const bf16*)x, (const bf16*)dadd,
                     (bf16*)dx, n >> 3, scale);
  return ia_check_launch();
}

extern "C" int ia_avgpool2_fwd(const void* x, void* y, int B, int H, int W, int C, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return IA_ERR_ARG;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const size_t total = (size_t)B * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, H, W, C, Ho, Wo, total);
  return ia_check_launch();
}

extern "C" int ia_avgpool2_bwd(const void* dy, void* dx, int B, int H, int W, int C, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return IA_ERR_ARG;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const size_t total = (size_t)B * H * W * (C >> 3);
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)dy, (bf16*)dx, H, W, C, Ho, Wo, total);
  return ia_check_launch();
}

extern "C" size_t ia_gap_workspace_bytes(int B, int HW, int C) { return (size_t)B * nsplit_of(HW) * C * sizeof(float); }

// pooled [B, C] fp32 = mean over HW of x [B, HW, C] bf16   (head.global_pool, reference image.py:255)
extern "C" int ia_gap_fwd(const void* x, float* pooled, int B, int HW, int C, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !pooled || B <= 0 || HW <= 0 || C <= 0 || (C & 7)) return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_gap_workspace_bytes(B, HW, C)) return IA_ERR_WORKSPACE;
  const int ns = nsplit_of(HW);
  hipLaunchKernelGGL(spatial_sum_kernel, dim3(ns, B), dim3(256), 0, stream, (const bf16*)x, (const bf16*)nullptr, (float*)workspace, HW, C, ns);
  hipLaunchKernelGGL(spatial_finish_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, (const float*)workspace, pooled, C, ns, 1.f / HW, B * C);
  return ia_check_launch();
}

extern "C" int ia_gap_bwd(const float* dpooled, void* dx, int B, int HW, int C, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dpooled || !dx || B <= 0 || HW <= 0 || C <= 0 || (C & 7)) return IA_ERR_ARG;
  const size_t total = (size_t)B * HW * (C >> 3);
  hipLaunchKernelGGL(spatial_bcast_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, dpooled, (bf16*)dx, HW, C, 1.f / HW, total);
  return ia_check_launch();
}

// out = x * sigmoid(conv1d_k(mean_HW(x)))[b, c] * coef + shortcut ; saves pooled, gate [B, C] fp32 for the backward pass
extern "C" int ia_eca_fwd(const void* x, const float* conv_w, int k, const void* shortcut, void* out, float* pooled, float* gate, int B,
                          int HW, int C, float coef, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !conv_w || !shortcut || !out || !pooled || !gate || k <= 0 || k > 16 || !(k & 1)) return IA_ERR_ARG;
  int rc = ia_gap_fwd(x, pooled, B, HW, C, workspace, workspace_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(eca_gate_fwd_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, (const float*)pooled, conv_w, gate, C, k, B * C);
  const size_t total = (size_t)B * HW * (C >> 3);
  hipLaunchKernelGGL(scale_residual_fwd_kernel<false>, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, (const float*)gate,
                     (const bf16*)shortcut, (bf16*)out, (bf16*)nullptr, HW, C, coef, 1.f, total);
  return ia_check_launch();
}

// The same block tail with `pooled` taken from the INPUT of the 1x1 convolution that produced x (x = a what^T + bias per pixel, so
// mean_HW x = (mean_HW a) what^T + bias exactly): the spatial reduction reads a [B, HW, Cmid] -- a quarter of x's bytes in a
// NormFreeBlock (conv3: mid -> 4 mid channels) -- and the fp32 mean no longer carries x's per-element bf16 rounding.  what
// [C][Cmid] bf16 (ia_ws_conv_weight_fwd's output), bias [C] or NULL.  Backward is ia_eca_bwd unchanged (the function is the same).
extern "C" size_t ia_eca_fwd_linear_workspace_bytes(int B, int HW, int Cmid) {
  return ia_gap_workspace_bytes(B, HW, Cmid) + (size_t)B * Cmid * sizeof(float);
}
// act_out (may be NULL): silu(out) * act_scale, the next block's opening activation, written by the same pass that writes out.
extern "C" int ia_eca_fwd_linear(const void* x, const void* a, const void* what, const float* bias, int Cmid, const float* conv_w, int k,
                                 const void* shortcut, void* out, void* act_out, float act_scale, float* pooled, float* gate, int B, int HW, int C,
                                 float coef, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !a || !what || !conv_w || !shortcut || !out || !pooled || !gate || k <= 0 || k > 16 || !(k & 1) || Cmid <= 0 || (Cmid & 7) || (C & 7))
    return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_eca_fwd_linear_workspace_bytes(B, HW, Cmid)) return IA_ERR_WORKSPACE;
  float* mean_a = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ia_gap_workspace_bytes(B, HW, Cmid));
  int rc = ia_gap_fwd(a, mean_a, B, HW, Cmid, workspace, workspace_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(eca_pool_linear_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, (const float*)mean_a, (const bf16*)what, bias, pooled,
                     C, Cmid, B * C);
  hipLaunchKernelGGL(eca_gate_fwd_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, (const float*)pooled, conv_w, gate, C, k, B * C);
  const size_t total = (size_t)B * HW * (C >> 3);
  if (act_out)
    hipLaunchKernelGGL(scale_residual_fwd_kernel<true>, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, (const float*)gate,
                       (const bf16*)shortcut, (bf16*)out, (bf16*)act_out, HW, C, coef, act_scale, total);
  else
    hipLaunchKernelGGL(scale_residual_fwd_kernel<false>, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, (const float*)gate,
                       (const bf16*)shortcut, (bf16*)out, (bf16*)nullptr, HW, C, coef, 1.f, total);
  return ia_check_launch();
}

// dx (gradient of x); the shortcut's gradient is dout itself.  dconv_w [k] +=.  scratch: 2*B*C floats after the gap workspace.
extern "C" size_t ia_eca_bwd_workspace_bytes(int B, int HW, int C) { return ia_gap_workspace_bytes(B, HW, C) + (size_t)2 * B * C * sizeof(float); }
// everything behind the spatial partial sums of dout * x (part, in the workspace): gate gradient, conv1d weight gradient, dx
static int eca_bwd_tail(const void* dout, const float* conv_w, int k, const float* pooled, const float* gate, void* dx, float* dconv_w, int B,
                        int HW, int C, float coef, void* workspace, hipStream_t stream);

extern "C" int ia_eca_bwd(const void* dout, const void* x, const float* conv_w, int k, const float* pooled, const float* gate, void* dx,
                          float* dconv_w, int B, int HW, int C, float coef, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dout || !x || !conv_w || !pooled || !gate || !dx || k <= 0 || k > 16 || !(k & 1) || (C & 7)) return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_eca_bwd_workspace_bytes(B, HW, C)) return IA_ERR_WORKSPACE;
  const int ns = nsplit_of(HW);
  hipLaunchKernelGGL(spatial_sum_kernel, dim3(ns, B), dim3(256), 0, stream, (const bf16*)dout, (const bf16*)x, (float*)workspace, HW, C, ns);
  return eca_bwd_tail(dout, conv_w, k, pooled, gate, dx, dconv_w, B, HW, C, coef, workspace, stream);
}

// ia_eca_bwd for a block tail that also wrote the next block's opening activation act = silu(out) * act_scale (ia_eca_fwd_linear's act_out):
// dtot = (dact [+ dact2]) * act_scale * silu'(out) [+ dout_direct] is out's whole gradient (ia_silu_bwd / ia_silu_bwd_sum's arithmetic,
// bit for bit), written to dtot -- the shortcut's gradient -- and the gate gradient's spatial sums come out of the same pass; dx, dconv_w
// as ia_eca_bwd.  dact2 / dout_direct may be NULL.  Workspace: ia_eca_bwd_workspace_bytes.
extern "C" int ia_eca_silu_bwd(const void* dact, const void* dact2, const void* out, const void* dout_direct, float act_scale, const void* x,
                               const float* conv_w, int k, const float* pooled, const float* gate, void* dtot, void* dx, float* dconv_w, int B,
                               int HW, int C, float coef, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dact || !out || !x || !conv_w || !pooled || !gate || !dtot || !dx || k <= 0 || k > 16 || !(k & 1) || (C & 7)) return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_eca_bwd_workspace_bytes(B, HW, C)) return IA_ERR_WORKSPACE;
  const int ns = nsplit_of(HW);
  hipLaunchKernelGGL(silu_bwd_dot_kernel, dim3(ns, B), dim3(256), 0, stream, (const bf16*)dact, (const bf16*)dact2, (const bf16*)out,
                     (const bf16*)dout_direct, (const bf16*)x, (bf16*)dtot, (float*)workspace, HW, C, ns, act_scale);
  return eca_bwd_tail(dtot, conv_w, k, pooled, gate, dx, dconv_w, B, HW, C, coef, workspace, stream);
}

static int eca_bwd_tail(const void* dout, const float* conv_w, int k, const float* pooled, const float* gate, void* dx, float* dconv_w, int B,
                        int HW, int C, float coef, void* workspace, hipStream_t stream) {
  const int ns = nsplit_of(HW);
  float* part = (float*)workspace;
  float* dgate = part + (size_t)B * ns * C;
  float* dpooled = dgate + (size_t)B * C;
  hipLaunchKernelGGL(spatial_finish_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, (const float*)part, dgate, C, ns, coef, B * C);
  hipLaunchKernelGGL(eca_gate_bwd_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, (const float*)dgate, gate, conv_w, dpooled, C, k, B * C);
  if (dconv_w) hipLaunchKernelGGL(eca_gate_wgrad_kernel, dim3(k), dim3(1024), 0, stream, (const float*)dgate, gate, pooled, dconv_w, B, C, k);
  const size_t total = (size_t)B * HW * (C >> 3);
  hipLaunchKernelGGL(scale_residual_bwd_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)dout, gate, (const float*)dpooled,
                     (bf16*)dx, HW, C, coef, total);
  return ia_check_launch();
}

// ------------------------------------------------------------------------ 3x3 stride-1 convolution without a patch matrix
// xp / dxp: [B, H+2, W+2, Cin], yp / dyp: [B, H+2, W+2, Cout] bf16 ("padded domain"); inputs (xp, dyp) must have a ZERO border,
// outputs carry garbage in their border rows.  Channels per group (Cin/groups, Cout/groups) must be powers of two >= 8.  The
// k axis of the GEMMs is (tap, channel): tap t reads the tensor itself shifted by (t/3-1)(W+2) + (t%3-1) rows (ia_gemm_view), so
// the activations are read in place (9 shifted reads served by L2) instead of through a 9x larger gathered matrix, and all
// channel groups run in one launch.

// ====================================================================================== direct 3x3 convolution (few channels per group)
// Round 5.  The shifted-view GEMM above fetches every activation row nine times through the L2 -> LDS path and synchronises once per
// tap; on the 64-channel groups of the NF-Net stages (288 FLOP per HBM byte: not an HBM-bound shape) it ran latency-bound at one DMA
// round trip per tap and workgroup (0.19 ms for 32 x 200 x 200 x 64 -> 64, 500 TFLOP/s), and on the stem's 16 -> 32 / 32 -> 64
// convolutions (HBM-bound shapes) at 1.1-1.7 TB/s.  This kernel is the layout-native form for CI -> CO channels per group, CI and CO
// in {16, 32, 64}:
//   * persistent workgroups, each bound to ONE channel group: the group's whole filter bank (<= 72 KiB) is fetched into LDS once;
//   * the work list is (image, 8 x 30 output tile); a tile's input -- 10 x 32 pixels, halo included -- arrives by LDS-DMA as one-KiB
//     pieces (16-byte chunks XOR-swizzled so that the 16 pixels of an MFMA fragment read conflict-free) into one of two buffers, the
//     NEXT tile's while the current one is computed: 1.33 x the activation bytes through the DMA path instead of 9 x, one wait + one
//     barrier per tile instead of nine;
//   * the nine taps are nine LDS row offsets of the same image: wave w owns output pixels 64 w .. 64 w + 63 of the tile (4 blocks of
//     16 pixels x CO output channels, MFMA 16x16x32; one k-step = 32 input channels of a tap -- two taps at CI = 16), fragments read
//     through inline asm one k-step ahead (a plain LDS load behind an LDS-DMA makes hipcc drain vmcnt(0): the DMA would be
//     serialised with the arithmetic -- measured, the parts added up);
//   * outputs leave straight from the accumulators: a lane holds CO / 4 consecutive channels of one pixel, the four lanes of a pixel
//     its whole row; the stores of tile t drain under tile t + 1.
// The data gradient is the same kernel (<CO, CI>) run on dy with the tap-flipped, transposed filter bank (ia_conv3x3_flip_weights).
namespace dconv {
constexpr int TH = 8, TW = 30, TWP = 32, IN_PX = (TH + 2) * TWP, TILE_PX = TH * TW;      // 240 output pixels per tile

struct Args {
  const bf16* xp; const bf16* w; const float* bias; bf16* yp;
  int B, H, W, Cin, Cout, groups;
  int tiles_x, tiles_y;
  int dbg;                     // IA_CONV_DBG (timing ablations, results wrong): 1 = no stores, 4 = no input DMA after the first tile
};

template <int CI> IA_DEV int akey(int P) { return CI == 64 ? (P & 7) : CI == 32 ? ((P >> 1) & 3) : 0; }     // swizzle key of input pixel P
template <int CO> IA_DEV int bkey(int row) { return (row >> (CO == 64 ? 3 : CO == 32 ? 2 : 1)) & 3; }      // of filter row n (64-byte rows)
// Fragment reads through inline asm, destinations tied to counted lgkmcnt waits (tools/lint_asm_waits.py checks the ISA for uses ahead of them)
IA_DEV void lds_read128(bf16x8& dst, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); }
template <int N, int NI>
IA_DEV void wait_frags(bf16x8 (&a)[4], bf16x8 (&b)[NI]) {
  if constexpr (NI == 4) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
  else if constexpr (NI == 2) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
  else asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]) : "n"(N));
}

template <int CI, int CO>
struct Geo {
  static constexpr int PB = CI * 2, NC = CI / 8, PPP = 64 / NC;            // bytes / 16-byte chunks per pixel, pixels per DMA piece
  static constexpr int NS = CI == 64 ? 18 : CI == 32 ? 9 : 5, NI = CO / 16;  // k-steps of 32, output-channel blocks of 16
  static constexpr int IN_BYTES = IN_PX * PB, W_BYTES = NS * CO * 64, LDS_BYTES = W_BYTES + 2 * IN_BYTES;
  static constexpr int IN_PIECES = IN_BYTES / 1024, W_PIECES = W_BYTES / 1024;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int CI, int CO>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(Args p) {
  using G = Geo<CI, CO>;
  constexpr int PB = G::PB, NC = G::NC, PPP = G::PPP, NS = G::NS, NI = G::NI, IN_BYTES = G::IN_BYTES, W_BYTES = G::W_BYTES;
  constexpr int IN_PIECES = G::IN_PIECES, W_PIECES = G::W_PIECES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int grp = blockIdx.x % p.groups, slot = blockIdx.x / p.groups, nslots = gridDim.x / p.groups;
  const int ntiles = p.B * p.tiles_y * p.tiles_x;
  if (slot >= ntiles) return;
  const int PW = p.W + 2, PH = p.H + 2;
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);
  const uint32_t sbase = ia_lds_addr(smem);

  // ---- the filter bank of this group: NS k-step tiles [CO n][32 k] with 64-byte rows, chunk position XOR bkey(n).  k-step s holds
  // tap s / 2, channels 32 (s & 1) .. (CI = 64); tap s (CI = 32); taps 2 s and 2 s + 1 (CI = 16; the tenth tap does not exist: zeros)
  {
    const bf16* wg = p.w + (size_t)grp * CO * 9 * CI;
    const __amdgpu_buffer_rsrc_t rsW = ia_rsrc(wg, (uint32_t)(CO * 9 * CI * 2));
#pragma unroll 1
    for (int pc = wave; pc < W_PIECES; pc += 4) {          // piece = 16 rows of one k-step tile
      const int s = pc / (CO / 16), r16 = pc - s * (CO / 16);
      const int row = r16 * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ bkey<CO>(row);
      int k;
      bool ok = true;
      if (CI == 64) k = (s >> 1) * 64 + (s & 1) * 32 + chunk * 8;
      else if (CI == 32) k = s * 32 + chunk * 8;
      else { const int tap = 2 * s + (chunk >> 1); k = tap * 16 + (chunk & 1) * 8; ok = tap < 9; }
      const uint32_t off = ok ? (uint32_t)((row * 9 * CI + k) * 2) : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lsm + pc * 1024, 16, off, 0, 0, 0);
    }
  }
  // ---- lane constants
  // A fragments: pixel q = 64 wave + 16 mi + li of the tile (clamped: the last 16 of the 256 fragment rows have no pixel) at
  // (y, x) = (q / 30, q % 30); LDS pixel of tap (dy, dx): P = (y + dy) * 32 + x + dx, byte P * PB + ((chunk ^ akey(P)) << 4)
  int a_p0[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    int q = wave * 64 + mi * 16 + li;
    q = q < TILE_PX ? q : TILE_PX - 1;
    const int y = q / TW, x = q - y * TW;
    a_p0[mi] = y * TWP + x;
  }
  // B fragments: row n = (li >> 2) * (CO / 4) + ni * 4 + (li & 3) (lane group g then holds CO / 4 consecutive n over its ni), chunk g
  int b_off[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = (li >> 2) * (CO / 4) + ni * 4 + (li & 3);
    b_off[ni] = row * 64 + ((g ^ bkey<CO>(row)) << 4);
  }
  f32x4 bv[NI];                                            // this lane's bias values, fetched once (a load in the epilogue would drain the stores)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
    bv[ni] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + grp * CO + g * (CO / 4) + ni * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  // DMA piece of the input: pixel lane / NC of the piece, chunk position lane % NC holds chunk (lane % NC) ^ akey(pixel)
  const uint32_t lane_in = (uint32_t)(((lane / NC) * p.Cin + (((lane % NC) ^ akey<CI>(lane / NC)) * 8)) * 2);
  const size_t total_in = (size_t)p.B * PH * PW * p.Cin;       // elements

  auto tile_of = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % p.tiles_x, r = t / p.tiles_x;
    const int ty = r % p.tiles_y;
    b = r / p.tiles_y; y0 = 1 + ty * TH; x0 = 1 + tx * TW;      // first OUTPUT pixel of the tile, bordered coordinates
  };
  // input of tile t -> buffer: rows y0 - 1 .. y0 + 8, pixels x0 - 1 .. x0 + 30 of image b, channels of this group.  The buffer window
  // starts at the tile's first pixel and ends with the tensor: pieces past the last image read zeros; pixels past a row's end are the
  // next row's first ones (they only feed output pixels that are never stored).
  auto stage = [&](int t, int buf) {
    int b, y0, x0;
    tile_of(t, b, y0, x0);
    const size_t org = (((size_t)b * PH + (y0 - 1)) * PW + (x0 - 1)) * p.Cin + (size_t)grp * CI;
    const size_t rem = (total_in - org) * 2;
    const __amdgpu_buffer_rsrc_t rs = ia_rsrc(p.xp + org, (uint32_t)(rem < 0x7FFFFFF0ull ? rem : 0x7FFFFFF0ull));
#pragma unroll
    for (int i = 0; i < (IN_PIECES + 3) / 4; ++i) {
      const int pc = wave + 4 * i;
      if (pc < IN_PIECES) {
        const int r = pc / (TWP / PPP), c = pc - r * (TWP / PPP);      // piece c of tile row r
        // (the row / piece advance sits in the LANE offset: the hardware's range check does not see a scalar offset, and the last
        // image's bottom / right tiles reach past the end of the tensor whenever H % 8 or W % 30 is not zero -- those lanes must read zeros)
        const uint32_t adv = (uint32_t)((r * PW + c * PPP) * p.Cin * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lsm + W_BYTES + buf * IN_BYTES + pc * 1024, 16, lane_in + adv, 0, 0, 0);
      }
    }
  };

  stage(slot, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
#pragma unroll 1
  for (int t = slot; t < ntiles; t += nslots, buf ^= 1) {
    if (t + nslots < ntiles && !(p.dbg & 4)) stage(t + nslots, buf ^ 1);
    const uint32_t in = sbase + W_BYTES + buf * IN_BYTES;
    f32x4 acc[4][NI];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // NS k-steps, the fragments of step s + 1 requested before the MFMAs of step s (nothing else covers the LDS latency); LDS returns
    // in order, so "at most 4 + NI reads outstanding" = step s has arrived
    bf16x8 af[2][4], bfr[2][NI];
    auto load = [&](int s, int slot2) {
      int off, chunk;
      if (CI == 64) { const int tap = s >> 1; off = (tap / 3) * TWP + tap % 3; chunk = (s & 1) * 4 + g; }
      else if (CI == 32) { off = (s / 3) * TWP + s % 3; chunk = g; }
      else {                                                   // two taps per k-step: lane groups 0, 1 the first, 2, 3 the second
        const int t0 = 2 * s, t1 = 2 * s + 1 < 9 ? 2 * s + 1 : 8;      // (the tenth tap's weights are zeros: any finite data will do)
        const int o0 = (t0 / 3) * TWP + t0 % 3, o1 = (t1 / 3) * TWP + t1 % 3;
        off = o0 + (g >> 1) * (o1 - o0); chunk = g & 1;
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int P = a_p0[mi] + off;
        lds_read128(af[slot2][mi], in + (uint32_t)(P * PB + ((chunk ^ akey<CI>(P)) << 4)));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) lds_read128(bfr[slot2][ni], sbase + (uint32_t)(s * (CO * 64) + b_off[ni]));
    };
    load(0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      // (no branch may sit between a read and its wait: a merge point makes hipcc copy the fragment registers -- before the data is there)
      if (s + 1 < NS) { load(s + 1, (s + 1) & 1); wait_frags<4 + NI, NI>(af[s & 1], bfr[s & 1]); }
      else wait_frags<0, NI>(af[s & 1], bfr[s & 1]);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s & 1][ni], af[s & 1][mi], acc[mi][ni], 0, 0, 0);
    }
    // the next tile's input has landed (and this workgroup's previous stores, issued before it, are done); everybody is through with `in`
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- stores: pixel q -> bordered (y0 + y, x0 + x); a lane's CO / 4 channels g * CO / 4 .. = acc[mi][0 .. NI - 1]
    int b, y0, x0;
    tile_of(t, b, y0, x0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int q = wave * 64 + mi * 16 + li;
      const int y = q / TW, x = q - y * TW;
      if (q >= TILE_PX || y0 + y > p.H || x0 + x > p.W || (p.dbg & 1)) continue;
      bf16* const dst = p.yp + (((size_t)b * PH + (y0 + y)) * PW + (x0 + x)) * p.Cout + (size_t)grp * CO + g * (CO / 4);
      if constexpr (NI == 1) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(acc[mi][0][j] + bv[0][j]);
        *reinterpret_cast<bf16x4*>(dst) = o;
      } else {
#pragma unroll
        for (int h = 0; h < NI / 2; ++h) {
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) { o[j] = f2bf(acc[mi][2 * h][j] + bv[2 * h][j]); o[4 + j] = f2bf(acc[mi][2 * h + 1][j] + bv[2 * h + 1][j]); }
          *reinterpret_cast<bf16x8*>(dst + 8 * h) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- stride-2 forward (round 6)
// y[oy][ox] = sum over taps of w[dy][dx] . x[2 oy + dy][2 ox + dx] (bordered input coordinates).  Rows and columns of the input split
// by parity: the EVEN bordered rows 2 i serve the taps dy = 0 (i = oy) and dy = 2 (i = oy + 1), the ODD rows 2 i + 1 serve dy = 1 (i = oy)
// -- likewise the columns.  So the convolution is the sum of four STRIDE-1 pieces over four views of x, each a dense (8 + 1) x (30 + 1)
// pixel tile once it sits in LDS: view (even, even) with the four corner taps at shifts {0, 1}^2, (even, odd) / (odd, even) with two
// taps each, (odd, odd) with the centre tap.  The kernel is the stride-1 one with the work list (tile, view): the LDS-DMA gathers a
// view's tile with a pixel stride of two (16-byte lanes, 128 contiguous bytes per pixel and group: full sectors), the k loop runs that
// view's taps out of the resident filter bank, the accumulators stay over the four views and the tile is stored after the last.  Every
// input byte is fetched once (+ halo) and no patch matrix exists: 1.6 units of traffic instead of 5.75.  64 channels per
// group in, CO out; `in_gstride` = 0 lets several output groups share one input slice (the stem's 64 -> 128: two output halves).
struct S2Args {
  const bf16* xp; const bf16* w; const float* bias; bf16* yp;
  int B, H, W;                 // OUTPUT height / width
  int XH, XW;                  // input height / width (H = (XH - 1) / 2 + 1)
  int Cin, Cout, groups, in_gstride;
  int tiles_x, tiles_y;
  int out_compact;             // yp is [B, H, W, Cout] without a border (the stem's last convolution feeds compact consumers)
};

template <int CO>
__global__ __launch_bounds__(256) void conv3x3_s2_kernel(S2Args p) {
  constexpr int CI = 64;
  using G = Geo<CI, CO>;
  constexpr int PB = G::PB, NC = G::NC, PPP = G::PPP, NI = G::NI, IN_BYTES = G::IN_BYTES, W_BYTES = G::W_BYTES;
  constexpr int IN_PIECES = G::IN_PIECES, W_PIECES = G::W_PIECES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int grp = blockIdx.x % p.groups, slot = blockIdx.x / p.groups, nslots = gridDim.x / p.groups;
  const int ntiles = p.B * p.tiles_y * p.tiles_x;
  if (slot >= ntiles) return;
  const int PW = p.W + 2, PH = p.H + 2, XPW = p.XW + 2, XPH = p.XH + 2;
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);
  const uint32_t sbase = ia_lds_addr(smem);
  {   // the filter bank, as in the stride-1 kernel: k-step s = tap s / 2, channels 32 (s & 1) ..
    const bf16* wg = p.w + (size_t)grp * CO * 9 * CI;
    const __amdgpu_buffer_rsrc_t rsW = ia_rsrc(wg, (uint32_t)(CO * 9 * CI * 2));
#pragma unroll 1
    for (int pc = wave; pc < W_PIECES; pc += 4) {
      const int s = pc / (CO / 16), r16 = pc - s * (CO / 16);
      const int row = r16 * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ bkey<CO>(row);
      const int k = (s >> 1) * 64 + (s & 1) * 32 + chunk * 8;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lsm + pc * 1024, 16, (uint32_t)((row * 9 * CI + k) * 2), 0, 0, 0);
    }
  }
  int a_p0[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    int q = wave * 64 + mi * 16 + li;
    q = q < TILE_PX ? q : TILE_PX - 1;
    const int y = q / TW, x = q - y * TW;
    a_p0[mi] = y * TWP + x;
  }
  int b_off[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = (li >> 2) * (CO / 4) + ni * 4 + (li & 3);
    b_off[ni] = row * 64 + ((g ^ bkey<CO>(row)) << 4);
  }
  f32x4 bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
    bv[ni] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + grp * CO + g * (CO / 4) + ni * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  // DMA piece of a view: pixel lane / NC of the piece lies 2 (lane / NC) input pixels further
  const uint32_t lane_in = (uint32_t)(((lane / NC) * 2 * p.Cin + (((lane % NC) ^ akey<CI>(lane / NC)) * 8)) * 2);
  const size_t total_in = (size_t)p.B * XPH * XPW * p.Cin;

  auto tile_of = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % p.tiles_x, r = t / p.tiles_x;
    const int ty = r % p.tiles_y;
    b = r / p.tiles_y; y0 = 1 + ty * TH; x0 = 1 + tx * TW;      // first OUTPUT pixel of the tile, bordered output coordinates
  };
  // view v = 2 (rows odd) + (columns odd) of tile t -> buffer: view pixel (r, c) = bordered input pixel (2 (y0 - 1 + r) + rows odd,
  // 2 (x0 - 1 + c) + columns odd); rows 0 .. 8 (the tenth row of the stride-1 tile has no reader)
  auto stage = [&](int t, int v, int buf) {
    int b, y0, x0;
    tile_of(t, b, y0, x0);
    const size_t org = (((size_t)b * XPH + 2 * (y0 - 1) + (v >> 1)) * XPW + 2 * (x0 - 1) + (v & 1)) * p.Cin + (size_t)grp * p.in_gstride;
    const size_t rem = org < total_in ? (total_in - org) * 2 : 0;
    const __amdgpu_buffer_rsrc_t rs = ia_rsrc(p.xp + (rem ? org : 0), (uint32_t)(rem < 0x7FFFFFF0ull ? rem : 0x7FFFFFF0ull));
#pragma unroll
    for (int i = 0; i < (IN_PIECES + 3) / 4; ++i) {
      const int pc = wave + 4 * i;
      const int r = pc / (TWP / PPP), c = pc - r * (TWP / PPP);
      if (pc < IN_PIECES && r <= TH) {
        const uint32_t adv = (uint32_t)((r * 2 * XPW + c * PPP * 2) * p.Cin * 2);      // (in the lane offset: range-checked, see the stride-1 kernel)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lsm + W_BYTES + buf * IN_BYTES + pc * 1024, 16, lane_in + adv, 0, 0, 0);
      }
    }
  };

  stage(slot, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
#pragma unroll 1
  for (int t = slot; t < ntiles; t += nslots) {
    f32x4 acc[4][NI];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (v < 3) stage(t, v + 1, buf ^ 1);
      else if (t + nslots < ntiles) stage(t + nslots, 0, buf ^ 1);
      const uint32_t in = sbase + W_BYTES + buf * IN_BYTES;
      // the view's taps: rows even -> dy in {0, 2} at row shifts {0, 1}; rows odd -> dy = 1 at shift 0; the same for the columns
      const int nry = (v >> 1) ? 1 : 2, nrx = (v & 1) ? 1 : 2, NK = nry * nrx * 2;
      bf16x8 af[2][4], bfr[2][NI];
      auto load = [&](int i, int slot2) {                       // i-th k-step of the view: (row tap, column tap, channel half)
        const int half = i & 1, ix = (i >> 1) % nrx, iy = (i >> 1) / nrx;
        const int dy = (v >> 1) ? 1 : 2 * iy, dx = (v & 1) ? 1 : 2 * ix;
        const int sy = (v >> 1) ? 0 : iy, sx = (v & 1) ? 0 : ix;
        const int s = (dy * 3 + dx) * 2 + half, off = sy * TWP + sx, chunk = half * 4 + g;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int P = a_p0[mi] + off;
          lds_read128(af[slot2][mi], in + (uint32_t)(P * PB + ((chunk ^ akey<CI>(P)) << 4)));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) lds_read128(bfr[slot2][ni], sbase + (uint32_t)(s * (CO * 64) + b_off[ni]));
      };
      load(0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < NK) {
          if (i + 1 < NK) { load(i + 1, (i + 1) & 1); wait_frags<4 + NI, NI>(af[i & 1], bfr[i & 1]); }
          else wait_frags<0, NI>(af[i & 1], bfr[i & 1]);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[i & 1][ni], af[i & 1][mi], acc[mi][ni], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next view has landed; everybody is through with `in`
      __syncthreads();
      buf ^= 1;
    }
    int b, y0, x0;
    tile_of(t, b, y0, x0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int q = wave * 64 + mi * 16 + li;
      const int y = q / TW, x = q - y * TW;
      if (q >= TILE_PX || y0 + y > p.H || x0 + x > p.W) continue;
      const size_t orow = p.out_compact ? ((size_t)b * p.H + (y0 + y - 1)) * p.W + (x0 + x - 1) : ((size_t)b * PH + (y0 + y)) * PW + (x0 + x);
      bf16* const dst = p.yp + orow * p.Cout + (size_t)grp * CO + g * (CO / 4);
#pragma unroll
      for (int h = 0; h < NI / 2; ++h) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = f2bf(acc[mi][2 * h][j] + bv[2 * h][j]); o[4 + j] = f2bf(acc[mi][2 * h + 1][j] + bv[2 * h + 1][j]); }
        *reinterpret_cast<bf16x8*>(dst + 8 * h) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- stride-2 data gradient (round 6)
// dx[2 j + py][2 i + px] (interior coordinates) = sum over the taps (dy, dx) with dy = py + 1 (mod 2), dx = px + 1 (mod 2) of
// w[dy][dx]^T . g[(2 j + py + 1 - dy) / 2][(2 i + px + 1 - dx) / 2]: the four parity classes of dx are four stride-1 pieces over the SAME
// (8 + 1) x (30 + 1) tile of the incoming gradient g -- class (0, 0) the centre tap at g[j][i], (0, 1) / (1, 0) two taps, (1, 1) the
// four corner taps at g[j + a][i + b], a = (dy == 0), b = (dx == 0).  One LDS tile of g per 8 x 30 cells serves all four classes; every
// class has its own accumulators and leaves as soon as it is done (its pixels lie two apart: 128-byte segments, the other classes fill
// the gaps).  The filter bank is the tap-flipped, transposed one of the stride-1 data gradient (flip_weights_kernel).  g needs a ZERO
// border (row Ho / column Wo are read for odd maps' last cells); 64 -> 64 channels per group.
struct S2DArgs {
  const bf16* gp; const bf16* wt; bf16* dxp;
  int B, H, W;                 // height / width of dx (the convolution's input)
  int Ho, Wo;                  // of g (its output)
  int Cin, Cout, groups;       // channels of dx / of g
  int tiles_x, tiles_y;        // over the cells (j, i): ceil(H / 2) x ceil(W / 2)
  // a contraction wider than 64 channels (the stem's 64 -> 128) runs as one launch per 64-channel slice of g: slice `g_choff` with its own
  // bank, every launch after the first ADDS to dx (a bf16 read-modify-write: one rounding more per slice, as the patch-matrix path's
  // per-tap bf16 partial sums had)
  int g_choff, accumulate;
  int g_compact;               // g is [B, Ho, Wo, Cout] without a border: the lanes that would read past a row / past the image fetch zeros
};

__global__ __launch_bounds__(256) void conv3x3_s2_dgrad_kernel(S2DArgs p) {
  constexpr int CI = 64, CO = 64;                              // CI: contraction (channels of g), CO: channels of dx
  using G = Geo<CI, CO>;
  constexpr int PB = G::PB, NC = G::NC, PPP = G::PPP, NI = G::NI, IN_BYTES = G::IN_BYTES, W_BYTES = G::W_BYTES;
  constexpr int IN_PIECES = G::IN_PIECES, W_PIECES = G::W_PIECES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int grp = blockIdx.x % p.groups, slot = blockIdx.x / p.groups, nslots = gridDim.x / p.groups;
  const int ntiles = p.B * p.tiles_y * p.tiles_x;
  if (slot >= ntiles) return;
  const int GPW = p.g_compact ? p.Wo : p.Wo + 2, GPH = p.g_compact ? p.Ho : p.Ho + 2, gb = p.g_compact ? 0 : 1, XPW = p.W + 2, XPH = p.H + 2;
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);
  const uint32_t sbase = ia_lds_addr(smem);
  {
    const bf16* wg = p.wt + (size_t)grp * CO * 9 * CI;
    const __amdgpu_buffer_rsrc_t rsW = ia_rsrc(wg, (uint32_t)(CO * 9 * CI * 2));
#pragma unroll 1
    for (int pc = wave; pc < W_PIECES; pc += 4) {
      const int s = pc / (CO / 16), r16 = pc - s * (CO / 16);
      const int row = r16 * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ bkey<CO>(row);
      const int k = (s >> 1) * 64 + (s & 1) * 32 + chunk * 8;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lsm + pc * 1024, 16, (uint32_t)((row * 9 * CI + k) * 2), 0, 0, 0);
    }
  }
  int a_p0[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    int q = wave * 64 + mi * 16 + li;
    q = q < TILE_PX ? q : TILE_PX - 1;
    const int y = q / TW, x = q - y * TW;
    a_p0[mi] = y * TWP + x;
  }
  int b_off[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = (li >> 2) * (CO / 4) + ni * 4 + (li & 3);
    b_off[ni] = row * 64 + ((g ^ bkey<CO>(row)) << 4);
  }
  const uint32_t lane_in = (uint32_t)(((lane / NC) * p.Cout + (((lane % NC) ^ akey<CI>(lane / NC)) * 8)) * 2);
  const size_t total_in = (size_t)p.B * GPH * GPW * p.Cout;

  auto tile_of = [&](int t, int& b, int& j0, int& i0) {
    const int tx = t % p.tiles_x, r = t / p.tiles_x;
    const int ty = r % p.tiles_y;
    b = r / p.tiles_y; j0 = ty * TH; i0 = tx * TW;               // first cell of the tile
  };
  // g tile: LDS pixel (r, c) = g[j0 + r][i0 + c] (interior coordinates; bordered: + 1), rows 0 .. 8
  auto stage = [&](int t, int buf) {
    int b, j0, i0;
    tile_of(t, b, j0, i0);
    const size_t org = (((size_t)b * GPH + (j0 + gb)) * GPW + (i0 + gb)) * p.Cout + (size_t)grp * CI + p.g_choff;
    const size_t rem = org < total_in ? (total_in - org) * 2 : 0;
    const __amdgpu_buffer_rsrc_t rs = ia_rsrc(p.gp + (rem ? org : 0), (uint32_t)(rem < 0x7FFFFFF0ull ? rem : 0x7FFFFFF0ull));
#pragma unroll
    for (int i = 0; i < (IN_PIECES + 3) / 4; ++i) {
      const int pc = wave + 4 * i;
      const int r = pc / (TWP / PPP), c = pc - r * (TWP / PPP);
      if (pc < IN_PIECES && r <= TH) {
        const uint32_t adv = (uint32_t)((r * GPW + c * PPP) * p.Cout * 2);
        const bool ok = !p.g_compact || (j0 + r < p.Ho && i0 + c * PPP + lane / NC < p.Wo);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lsm + W_BYTES + buf * IN_BYTES + pc * 1024, 16, ok ? lane_in + adv : 0xFFFFFFF0u, 0, 0, 0);
      }
    }
  };

  stage(slot, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
#pragma unroll 1
  for (int t = slot; t < ntiles; t += nslots, buf ^= 1) {
    if (t + nslots < ntiles) stage(t + nslots, buf ^ 1);
    const uint32_t in = sbase + W_BYTES + buf * IN_BYTES;
    int b, j0, i0;
    tile_of(t, b, j0, i0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {                                // parity class (py, px) = (v >> 1, v & 1) of dx
      f32x4 acc[4][NI];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nry = (v >> 1) ? 2 : 1, nrx = (v & 1) ? 2 : 1, NK = nry * nrx * 2;
      bf16x8 af[2][4], bfr[2][NI];
      auto load = [&](int i, int slot2) {
        const int half = i & 1, ix = (i >> 1) % nrx, iy = (i >> 1) / nrx;
        const int dy = (v >> 1) ? 2 * iy : 1, dx = (v & 1) ? 2 * ix : 1;      // class 1: taps 0 and 2; class 0: the middle tap
        const int sy = dy == 0 ? 1 : 0, sx = dx == 0 ? 1 : 0;
        const int s = (8 - (dy * 3 + dx)) * 2 + half, off = sy * TWP + sx, chunk = half * 4 + g;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int P = a_p0[mi] + off;
          lds_read128(af[slot2][mi], in + (uint32_t)(P * PB + ((chunk ^ akey<CI>(P)) << 4)));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) lds_read128(bfr[slot2][ni], sbase + (uint32_t)(s * (CO * 64) + b_off[ni]));
      };
      load(0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < NK) {
          if (i + 1 < NK) { load(i + 1, (i + 1) & 1); wait_frags<4 + NI, NI>(af[i & 1], bfr[i & 1]); }
          else wait_frags<0, NI>(af[i & 1], bfr[i & 1]);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[i & 1][ni], af[i & 1][mi], acc[mi][ni], 0, 0, 0);
        }
      }
      // this class's pixels: cell q -> dx (2 (j0 + y) + py, 2 (i0 + x) + px), bordered + 1
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int q = wave * 64 + mi * 16 + li;
        const int y = q / TW, x = q - y * TW;
        const int iy = 2 * (j0 + y) + (v >> 1), ix = 2 * (i0 + x) + (v & 1);
        if (q >= TILE_PX || iy >= p.H || ix >= p.W) continue;
        bf16* const dst = p.dxp + (((size_t)b * XPH + (iy + 1)) * XPW + (ix + 1)) * p.Cin + (size_t)grp * CO + g * (CO / 4);
#pragma unroll
        for (int h = 0; h < NI / 2; ++h) {
          bf16x8 o;
          if (p.accumulate) {
            const bf16x8 old = *reinterpret_cast<const bf16x8*>(dst + 8 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = f2bf(acc[mi][2 * h][j] + bf2f(old[j])); o[4 + j] = f2bf(acc[mi][2 * h + 1][j] + bf2f(old[4 + j])); }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = f2bf(acc[mi][2 * h][j]); o[4 + j] = f2bf(acc[mi][2 * h + 1][j]); }
          }
          *reinterpret_cast<bf16x8*>(dst + 8 * h) = o;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the next tile of g has landed; everybody is through with `in`
    __syncthreads();
  }
}

// wt[g][ci][(8 - tap) * CO + co] = w[g][co][tap * CI + ci]: the filter bank of the data gradient (a correlation with the flipped taps)
__global__ __launch_bounds__(256) void flip_weights_kernel(const bf16* __restrict__ w, bf16* __restrict__ wt, int CI, int CO, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ci = idx % CI, tap = (idx / CI) % 9, co = (idx / (9 * CI)) % CO, grp = idx / (9 * CI * CO);
  wt[((size_t)grp * CI + ci) * (9 * CO) + (8 - tap) * CO + co] = w[idx];
}

static bool enabled() {
  static const bool on = [] { const char* e = getenv("IA_CONV_DIRECT"); return !e || atoi(e) != 0; }();
  return on;
}
static bool wgrad_enabled() {
  static const bool on = [] { const char* e = getenv("IA_CONV_DIRECT_WGRAD"); return !e || atoi(e) != 0; }();
  return on;
}
// smallest B * H * W the direct weight gradient takes (below it the split-K GEMM is faster); IA_CONV_DIRECT_WGRAD_MIN overrides it --
// read on every call, so that a test can route small ragged maps through the direct kernel
static size_t wgrad_min_pixels() {
  const char* e = getenv("IA_CONV_DIRECT_WGRAD_MIN");
  return e ? (size_t)atol(e) : 100000;
}
static bool pair_ok(int ci, int co) {
  return (ci == 64 && co == 64) || (ci == 16 && co == 32) || (ci == 32 && co == 64) || (ci == 64 && co == 32) || (ci == 32 && co == 16);
}
template <int CI, int CO>
static int launch_t(Args a, hipStream_t stream) {
  using G = Geo<CI, CO>;
  auto kern = conv3x3_direct_kernel<CI, CO>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) return IA_ERR_LAUNCH;
    attr_set = true;
  }
  const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
  const int per_cu = (160 * 1024) / G::LDS_BYTES >= 2 ? 2 : 1;      // (the kernel's ~220 registers allow two workgroups per CU)
  long per_group = (256L * per_cu) / a.groups;
  if (per_group > ntiles) per_group = ntiles;
  if (per_group < 1) per_group = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(per_group * a.groups)), dim3(256), G::LDS_BYTES, stream, a);
  return ia_check_launch();
}
// y (bordered) = conv3x3(x bordered) + bias for `groups` groups of ci -> co channels
static int launch(const void* xp, const void* w, const float* bias, void* yp, int B, int H, int W, int ci, int co, int groups, hipStream_t stream) {
  Args a;
  a.xp = (const bf16*)xp; a.w = (const bf16*)w; a.bias = bias; a.yp = (bf16*)yp;
  a.B = B; a.H = H; a.W = W; a.Cin = groups * ci; a.Cout = groups * co; a.groups = groups;
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("IA_CONV_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
  if (ci == 64 && co == 64) return launch_t<64, 64>(a, stream);
  if (ci == 16 && co == 32) return launch_t<16, 32>(a, stream);
  if (ci == 32 && co == 64) return launch_t<32, 64>(a, stream);
  if (ci == 64 && co == 32) return launch_t<64, 32>(a, stream);
  if (ci == 32 && co == 16) return launch_t<32, 16>(a, stream);
  return IA_ERR_UNSUPPORTED;
}
// ---------------------------------------------------------------------------------------------- direct weight gradient
// dW[co][tap * CI + ci] = sum over the image interiors of dy[px][co] x[px + tap][ci] (+ dbias[co] = sum dy[px][co]).  The same tiles
// as the forward kernel: per 8 x 30 output tile the dy rows (8 x 32 pixels; the two surplus pixels of a row and everything outside
// the image arrive as ZEROS -- out-of-range lanes of the DMA piece) and the x tile with its halo sit in LDS, one k-step = one tile
// row of 32 pixels.  Both operands have the contraction index (the pixel) as their LDS row, so the fragments are transpose reads
// (ds_read_b64_tr_b16, two per fragment).  D = mfma(x fragment, dy fragment): lane li <-> co, four registers <-> four consecutive ci.
// The 9 CI / 16 column blocks (tap, 16 input channels) are dealt to the four waves; every wave keeps its blocks' sums for all CO in
// registers over ALL tiles of the workgroup (persistent, one group per workgroup) and writes one fp32 partial bank at the end; a
// fixed-order fold adds the workgroups' banks (deterministic).
template <int CI, int CO>
struct WGeo {
  static constexpr int XB = CI * 2, YB = CO * 2;                       // bytes per pixel of the x / dy tiles
  static constexpr int NB = 9 * CI / 16, NBW = (NB + 3) / 4, MB = CO / 16;
  static constexpr int X_BYTES = IN_PX * XB + 4 * XB, Y_BYTES = TH * TWP * YB;      // (+ the two pixels a shifted read of the last row overhangs)
  static constexpr int STAGE = ((X_BYTES + Y_BYTES + 1023) / 1024) * 1024, LDS_BYTES = 2 * STAGE;
  static constexpr int X_PIECES = IN_PX * XB / 1024, Y_PIECES = Y_BYTES / 1024;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};
// swizzle of a tile read ONLY by transpose reads: 128-byte rows as the attention kernels' V tiles, narrower rows unswizzled
template <int C> IA_DEV int tkey(int P) { return C == 64 ? (((P >> 1) & 1) << 2) : 0; }

// transposed fragment (8 consecutive pixels P .. P + 3, P + 4 .. P + 7 of channel column ch .. ) out of a [pixel][C channels] tile
template <int C>
IA_DEV bf16x8 tr_pair(uint32_t base, int P, int ch) {
  const uint32_t a0 = base + (uint32_t)(P * (C * 2) + (((ch >> 3) ^ tkey<C>(P)) << 4) + (ch & 7) * 2);
  const uint32_t a1 = base + (uint32_t)((P + 4) * (C * 2) + (((ch >> 3) ^ tkey<C>(P + 4)) << 4) + (ch & 7) * 2);
  const s16x4 lo = ia_tr_read<0>(a0), hi = ia_tr_read<0>(a1);
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

struct WArgs {
  const bf16* xp; const bf16* dyp; float* part;      // part: [workgroups][CO * 9 * CI + CO] fp32 partial banks (+ bias sums)
  int B, H, W, Cin, Cout, groups;                    // H, W: height / width of dy
  int tiles_x, tiles_y;
  // x as a strided view (the stride-2 convolution's weight gradient, below): bordered x pixel of view pixel (r, c) = (xs r + offy, xs c + offx)
  // in a tensor of XH x XW interior pixels; group g reads channels g * in_gstride ..  Stride 1: xs = 1, off = 0, XH = H, XW = W, in_gstride = CI.
  int XH, XW, xs, offy, offx, in_gstride;
  int dy_compact;              // dyp is [B, H, W, Cout] without a border
};

// TAPS: bit t = kernel tap t is wanted (a parity view of the stride-2 form needs 1, 2 or 4 of the nine).  A compile-time mask: a run-time
// branch around the transpose reads is a merge point at which hipcc copies the fragment registers before the data has arrived
// (tools/lint_asm_waits.py caught exactly that); with CI = 64 a wave's j-th column block IS tap j, so the mask folds away.
template <int CI, int CO, int TAPS = 0x1FF>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WArgs p) {
  static_assert(TAPS == 0x1FF || CI == 64, "tap masks need the block <-> tap identity of CI = 64");
  using G = WGeo<CI, CO>;
  // (every Geo constant used inside the lambdas is copied to a local first: `X_BYTES` written inside an argument of the LDS-DMA
  // builtin made hipcc drop this kernel's host launch stub -- undefined symbol at load time, tests/test_cabi_symbols.py)
  constexpr int XB = G::XB, YB = G::YB, NB = G::NB, NBW = G::NBW, MB = G::MB, STAGE = G::STAGE, XNC = CI / 8, YNC = CO / 8;
  constexpr int X_BYTES = G::X_BYTES, X_PIECES = G::X_PIECES, Y_PIECES = G::Y_PIECES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int grp = blockIdx.x % p.groups, slot = blockIdx.x / p.groups, nslots = gridDim.x / p.groups;
  const int ntiles = p.B * p.tiles_y * p.tiles_x;
  const int PW = p.W + 2, PH = p.H + 2;
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);
  const uint32_t sbase = ia_lds_addr(smem);
  const int XPW = p.XW + 2, XPH = p.XH + 2;
  const int YW = p.dy_compact ? p.W : PW, YH = p.dy_compact ? p.H : PH, yo = p.dy_compact ? 1 : 0;      // dy's own row pitch / origin shift
  const size_t total_x = (size_t)p.B * XPH * XPW * p.Cin, total_y = (size_t)p.B * YH * YW * p.Cout;

  auto tile_of = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % p.tiles_x, r = t / p.tiles_x;
    const int ty = r % p.tiles_y;
    b = r / p.tiles_y; y0 = 1 + ty * TH; x0 = 1 + tx * TW;
  };
  const uint32_t lane_x = (uint32_t)(((lane / XNC) * p.xs * p.Cin + (((lane % XNC) ^ tkey<CI>(lane / XNC)) * 8)) * 2);
  const int ypix = lane / YNC;                               // pixel of this lane inside a dy piece
  const uint32_t lane_y = (uint32_t)((ypix * p.Cout + (((lane % YNC) ^ tkey<CO>(ypix)) * 8)) * 2);
  auto stage = [&](int t, int buf) {
    int b, y0, x0;
    tile_of(t, b, y0, x0);
    {   // x tile: rows y0 - 1 .. y0 + 8, pixels x0 - 1 .. x0 + 30 (as the forward kernel)
      const size_t org = (((size_t)b * XPH + p.xs * (y0 - 1) + p.offy) * XPW + p.xs * (x0 - 1) + p.offx) * p.Cin + (size_t)grp * p.in_gstride;
      const size_t rem = org < total_x ? (total_x - org) * 2 : 0;
      const __amdgpu_buffer_rsrc_t rs = ia_rsrc(p.xp + (rem ? org : 0), (uint32_t)(rem < 0x7FFFFFF0ull ? rem : 0x7FFFFFF0ull));
      constexpr int PPP = 64 / XNC;
#pragma unroll
      for (int i = 0; i < (X_PIECES + 3) / 4; ++i) {
        const int pc = wave + 4 * i;
        if (pc < X_PIECES) {
          const int r = pc / (TWP / PPP), c = pc - r * (TWP / PPP);
          // whole address in the lane offset (range-checked; see the forward kernel): x past the end of the tensor arrives as zeros --
          // read through the scalar offset it was whatever lies behind the allocation, and 0 (masked dy) x NaN = NaN in dW
          const uint32_t adv = (uint32_t)((r * XPW + c * PPP) * p.xs * p.Cin * 2);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lsm + buf * STAGE + pc * 1024, 16, lane_x + adv, 0, 0, 0);
        }
      }
    }
    {   // dy tile: rows y0 .. y0 + 7, pixels x0 .. x0 + 31; pixels 30, 31 of a row, pixels right of the image and rows below it -> zeros
      const size_t org = (((size_t)b * YH + (y0 - yo)) * YW + (x0 - yo)) * p.Cout + (size_t)grp * CO;
      const size_t rem = (total_y - org) * 2;
      const __amdgpu_buffer_rsrc_t rs = ia_rsrc(p.dyp + org, (uint32_t)(rem < 0x7FFFFFF0ull ? rem : 0x7FFFFFF0ull));
      constexpr int PPP = 64 / YNC;
#pragma unroll
      for (int i = 0; i < (Y_PIECES + 3) / 4; ++i) {
        const int pc = wave + 4 * i;
        if (pc < Y_PIECES) {
          const int r = pc / (TWP / PPP), c = pc - r * (TWP / PPP);
          const int col = c * PPP + ypix;
          const bool ok = col < TW && x0 + col <= p.W && y0 + r <= p.H;
          const uint32_t voff = ok ? lane_y + (uint32_t)((r * YW + c * PPP) * p.Cout * 2) : 0xFFFFFFF0u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lsm + buf * STAGE + X_BYTES + pc * 1024, 16, voff, 0, 0, 0);
        }
      }
    }
  };

  f32x4 acc[MB][NBW], racc[MB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) {
    racc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NBW; ++j) acc[mi][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = f2bf(1.0f);
  // transpose-read lane geometry: a 16-lane group addresses a [4 pixels][16 channels] block -- lane li: pixel (li >> 2), channels
  // 4 (li & 3) .. + 3 -- and receives column li; lane group g takes pixels 8 g .. 8 g + 7 of the k-step (two reads: + 0, + 4)
  const int tpx = g * 8 + (li >> 2), tch = (li & 3) * 4;
  // (a free function template, not a generic lambda: with a lambda handed to a lambda hipcc dropped this kernel's host launch stub --
  // the library then fails to load with an undefined symbol, tests/test_cabi_symbols.py)

  // the shifted reads of the x tile's last row overhang it by up to two pixels (they only ever meet dy = 0): keep finite bytes there
  if (tid < (4 * XB) / 4) {
    *reinterpret_cast<uint32_t*>(smem + IN_PX * XB + tid * 4) = 0u;
    *reinterpret_cast<uint32_t*>(smem + STAGE + IN_PX * XB + tid * 4) = 0u;
  }
  if (slot < ntiles) {
    stage(slot, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  int buf = 0;
#pragma unroll 1
  for (int t = slot; t < ntiles; t += nslots, buf ^= 1) {
    if (t + nslots < ntiles) stage(t + nslots, buf ^ 1);
    const uint32_t xs = sbase + buf * STAGE, ys = xs + X_BYTES;
    // k-step = tile row r (32 pixels).  Two register sets: the fragments of row r + 1 are requested right after row r's have arrived and
    // stay in flight under row r's MFMAs.
    bf16x8 af[2][MB], bf[2][NBW];
    auto load = [&](int r, int set) {
#pragma unroll
      for (int mi = 0; mi < MB; ++mi) af[set][mi] = tr_pair<CO>(ys, r * TWP + tpx, mi * 16 + tch);
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int nb = wave + 4 * j < NB ? wave + 4 * j : NB - 1;      // column block (tap, 16 input channels), wave-uniform (clamped: unused)
        const int tap = nb / (CI / 16), cb = nb - tap * (CI / 16);
        if (TAPS == 0x1FF || ((TAPS >> j) & 1)) bf[set][j] = tr_pair<CI>(xs, r * TWP + tpx + (tap / 3) * TWP + tap % 3, cb * 16 + tch);
      }
    };
    load(0, 0);
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // row r has arrived (lgkmcnt is 4 bits: no counted form for 26 reads)
      __builtin_amdgcn_sched_barrier(0);
      if (r + 1 < TH) load(r + 1, (r + 1) & 1);                        // in flight under this row's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NBW; ++j)
        if (wave + 4 * j < NB && (TAPS == 0x1FF || ((TAPS >> j) & 1))) {
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[r & 1][j], af[r & 1][mi], acc[mi][j], 0, 0, 0);
        }
      if (wave == 0) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) racc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[r & 1][mi], racc[mi], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // ---- this workgroup's partial bank: part[wg][co][9 CI] then [CO] bias sums.  Lane li <-> co = 16 mi + li, registers <-> ci 4 g .. + 3
  float* const bank = p.part + (size_t)blockIdx.x * (CO * 9 * CI + CO);
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) {
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int nb = wave + 4 * j;
      if (nb < NB) *reinterpret_cast<f32x4*>(bank + (size_t)(mi * 16 + li) * (9 * CI) + nb * 16 + g * 4) = acc[mi][j];
    }
    if (wave == 0 && g == 0) bank[CO * 9 * CI + mi * 16 + li] = racc[mi][0];
  }
}

// The stride-2 form in ONE launch (64 -> 64 channels per group): work list (tile, parity view of x) as in the forward kernel -- the dy tile is
// fetched once per tile and serves the four views, the x buffers alternate per view, and every view's taps accumulate straight into the
// column blocks of the REAL taps (rows even: kernel tap ty -> dy = 2 ty; rows odd: ty = 0 -> dy = 1; the same for the columns), so the bank has
// the stride-1 layout and the stride-1 fold applies.  Against four masked launches: dy read once instead of four times, one launch.
__global__ __launch_bounds__(256) void conv3x3_wgrad_s2_kernel(WArgs p) {
  constexpr int CI = 64, CO = 64;
  using G = WGeo<CI, CO>;
  constexpr int XB = G::XB, MB = G::MB, STAGE = G::STAGE, XNC = CI / 8, YNC = CO / 8;
  constexpr int X_BYTES = G::X_BYTES, X_PIECES = G::X_PIECES, Y_PIECES = G::Y_PIECES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int grp = blockIdx.x % p.groups, slot = blockIdx.x / p.groups, nslots = gridDim.x / p.groups;
  const int ntiles = p.B * p.tiles_y * p.tiles_x;
  const int PW = p.W + 2, PH = p.H + 2, XPW = p.XW + 2, XPH = p.XH + 2;
  const int YW = p.dy_compact ? p.W : PW, YH = p.dy_compact ? p.H : PH, yo = p.dy_compact ? 1 : 0;
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lsm = (lds_char*)IA_LDS(smem);
  const uint32_t sbase = ia_lds_addr(smem);
  const size_t total_x = (size_t)p.B * XPH * XPW * p.Cin, total_y = (size_t)p.B * YH * YW * p.Cout;

  auto tile_of = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % p.tiles_x, r = t / p.tiles_x;
    const int ty = r % p.tiles_y;
    b = r / p.tiles_y; y0 = 1 + ty * TH; x0 = 1 + tx * TW;
  };
  const uint32_t lane_x = (uint32_t)(((lane / XNC) * 2 * p.Cin + (((lane % XNC) ^ tkey<CI>(lane / XNC)) * 8)) * 2);
  const int ypix = lane / YNC;
  const uint32_t lane_y = (uint32_t)((ypix * p.Cout + (((lane % YNC) ^ tkey<CO>(ypix)) * 8)) * 2);
  // x view v = 2 (rows odd) + (columns odd) of tile t -> x buffer xb: view pixel (r, c) = bordered x pixel (2 (y0 - 1 + r) + rows odd, ...)
  auto stage_x = [&](int t, int v, int xb) {
    int b, y0, x0;
    tile_of(t, b, y0, x0);
    const size_t org = (((size_t)b * XPH + 2 * (y0 - 1) + (v >> 1)) * XPW + 2 * (x0 - 1) + (v & 1)) * p.Cin + (size_t)grp * p.in_gstride;
    const size_t rem = org < total_x ? (total_x - org) * 2 : 0;
    const __amdgpu_buffer_rsrc_t rs = ia_rsrc(p.xp + (rem ? org : 0), (uint32_t)(rem < 0x7FFFFFF0ull ? rem : 0x7FFFFFF0ull));
    constexpr int PPP = 64 / XNC;
#pragma unroll
    for (int i = 0; i < (X_PIECES + 3) / 4; ++i) {
      const int pc = wave + 4 * i;
      const int r = pc / (TWP / PPP), c = pc - r * (TWP / PPP);
      if (pc < X_PIECES && r <= TH) {                           // rows 0 .. 8: the views' taps reach one row down, not two
        const uint32_t adv = (uint32_t)((r * XPW + c * PPP) * 2 * p.Cin * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lsm + xb * STAGE + pc * 1024, 16, lane_x + adv, 0, 0, 0);
      }
    }
  };
  auto stage_y = [&](int t, int yb) {
    int b, y0, x0;
    tile_of(t, b, y0, x0);
    const size_t org = (((size_t)b * YH + (y0 - yo)) * YW + (x0 - yo)) * p.Cout + (size_t)grp * CO;
    const size_t rem = (total_y - org) * 2;
    const __amdgpu_buffer_rsrc_t rs = ia_rsrc(p.dyp + org, (uint32_t)(rem < 0x7FFFFFF0ull ? rem : 0x7FFFFFF0ull));
    constexpr int PPP = 64 / YNC;
#pragma unroll
    for (int i = 0; i < (Y_PIECES + 3) / 4; ++i) {
      const int pc = wave + 4 * i;
      if (pc < Y_PIECES) {
        const int r = pc / (TWP / PPP), c = pc - r * (TWP / PPP);
        const int col = c * PPP + ypix;
        const bool ok = col < TW && x0 + col <= p.W && y0 + r <= p.H;
        const uint32_t voff = ok ? lane_y + (uint32_t)((r * YW + c * PPP) * p.Cout * 2) : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lsm + yb * STAGE + X_BYTES + pc * 1024, 16, voff, 0, 0, 0);
      }
    }
  };

  f32x4 acc[MB][9], racc[MB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) {
    racc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[mi][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = f2bf(1.0f);
  const int tpx = g * 8 + (li >> 2), tch = (li & 3) * 4;
  // the tail of both x buffers (row 9 is not fetched, shifted reads of row 8 overhang by one pixel): finite bytes, they only meet dy = 0
  for (int i = tid; i < (X_BYTES - (TH + 1) * TWP * XB) / 4; i += 256) {
    *reinterpret_cast<uint32_t*>(smem + (TH + 1) * TWP * XB + i * 4) = 0u;
    *reinterpret_cast<uint32_t*>(smem + STAGE + (TH + 1) * TWP * XB + i * 4) = 0u;
  }
  if (slot < ntiles) {
    stage_y(slot, 0);
    stage_x(slot, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  int yb = 0;
#pragma unroll 1
  for (int t = slot; t < ntiles; t += nslots, yb ^= 1) {
    const uint32_t ys = sbase + yb * STAGE + X_BYTES;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (v < 3) stage_x(t, v + 1, (v + 1) & 1);
      else if (t + nslots < ntiles) { stage_x(t + nslots, 0, 0); stage_y(t + nslots, yb ^ 1); }
      const uint32_t xs = sbase + (v & 1) * STAGE;
      const int nty = (v >> 1) ? 1 : 2, ntx = (v & 1) ? 1 : 2, NT = nty * ntx;      // the view's kernel taps (ty, tx), ty < nty, tx < ntx
      bf16x8 af[2][MB], bf[2][4];
      auto load = [&](int r, int set) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) af[set][mi] = tr_pair<CO>(ys, r * TWP + tpx, mi * 16 + tch);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < NT) bf[set][k] = tr_pair<CI>(xs, r * TWP + tpx + (k / ntx) * TWP + k % ntx, wave * 16 + tch);
      };
      load(0, 0);
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (r + 1 < TH) load(r + 1, (r + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < NT) {
            const int ty = k / ntx, tx = k % ntx;
            const int T = ((v >> 1) ? 1 : 2 * ty) * 3 + ((v & 1) ? 1 : 2 * tx);       // the real tap
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) acc[mi][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[r & 1][k], af[r & 1][mi], acc[mi][T], 0, 0, 0);
          }
        if (v == 0 && wave == 0) {
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) racc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[r & 1][mi], racc[mi], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  float* const bank = p.part + (size_t)blockIdx.x * (CO * 9 * CI + CO);
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) {
#pragma unroll
    for (int j = 0; j < 9; ++j) *reinterpret_cast<f32x4*>(bank + (size_t)(mi * 16 + li) * (9 * CI) + (wave + 4 * j) * 16 + g * 4) = acc[mi][j];
    if (wave == 0 && g == 0) bank[CO * 9 * CI + mi * 16 + li] = racc[mi][0];
  }
}

// dwhat[g][co][9 CI] = sum over the group's workgroups of their banks (fixed order), dbias[g * CO + co] += the bias sums
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ dwhat, float* __restrict__ dbias, int groups,
                                                         int per_group, int bank, int wsize) {
  const int idx = blockIdx.x * 256 + threadIdx.x;      // element of [groups][bank]
  if (idx >= groups * bank) return;
  const int grp = idx / bank, e = idx - grp * bank;
  float sum = 0.f;
  for (int s = 0; s < per_group; ++s) sum += part[(size_t)(s * groups + grp) * bank + e];      // workgroup blockIdx = slot * groups + grp
  if (e < wsize) dwhat[(size_t)grp * wsize + e] = sum;
  else if (dbias) dbias[grp * (bank - wsize) + (e - wsize)] += sum;
}

// the stride-2 form: four launches of the weight-gradient kernel, one per parity view of x (view v = 2 (rows odd) + (columns odd)), each with
// its own region of `part`; tap (dy, dx) of the convolution is kernel tap (ty, tx) of one view -- dy = 0 -> (rows even, ty = 0), dy = 1 ->
// (rows odd, ty = 0), dy = 2 -> (rows even, ty = 1), the same for the columns -- the other 27 tap sums the four launches produce are dropped
__global__ __launch_bounds__(256) void wgrad_fold_s2_kernel(const float* __restrict__ part, float* __restrict__ dwhat, float* __restrict__ dbias, int groups,
                                                            int per_group, int bank, int wsize, int CI, size_t region) {
  const int idx = blockIdx.x * 256 + threadIdx.x;      // element of [groups][bank]
  if (idx >= groups * bank) return;
  const int grp = idx / bank, e = idx - grp * bank;
  int v = 0, src = e;
  if (e < wsize) {
    const int co = e / (9 * CI), rest = e - co * (9 * CI), tap = rest / CI, ci = rest - tap * CI;
    const int dy = tap / 3, dx = tap - dy * 3;
    v = ((dy == 1) << 1) | (dx == 1);
    const int tk = (dy == 2 ? 3 : 0) + (dx == 2 ? 1 : 0);
    src = co * (9 * CI) + tk * CI + ci;
  }
  const float* pv = part + (size_t)v * region;
  float sum = 0.f;
  for (int s = 0; s < per_group; ++s) sum += pv[(size_t)(s * groups + grp) * bank + src];
  if (e < wsize) dwhat[(size_t)grp * wsize + e] = sum;
  else if (dbias) dbias[grp * (bank - wsize) + (e - wsize)] += sum;
}

// IA_CONV_S2_WGRAD_MERGED=0: the four masked view launches instead of the one-launch kernel (read per call: A/B in one process)
static bool wgrad_s2_merged() {
  const char* e = getenv("IA_CONV_S2_WGRAD_MERGED");
  return !e || atoi(e) != 0;
}
template <int TAPS>
static int wgrad_view_launch(const WArgs& a, unsigned grid, hipStream_t stream) {
  using G = WGeo<64, 64>;
  auto kern = conv3x3_wgrad_kernel<64, 64, TAPS>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) return IA_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, stream, a);
  return IA_OK;
}

template <int CI, int CO>
static int wgrad_t(WArgs a, float* dwhat, float* dbias, hipStream_t stream) {
  using G = WGeo<CI, CO>;
  auto kern = conv3x3_wgrad_kernel<CI, CO>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) return IA_ERR_LAUNCH;
    attr_set = true;
  }
  const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
  const int per_cu = (160 * 1024) / G::LDS_BYTES >= 2 ? 2 : 1;
  long per_group = (256L * per_cu) / a.groups;
  if (per_group > ntiles) per_group = ntiles;
  if (per_group < 1) per_group = 1;
  const int wsize = CO * 9 * CI, bank = wsize + CO;
  if (a.xs == 2 && wgrad_s2_merged()) {
    if constexpr (CI == 64 && CO == 64) {
      auto k2 = conv3x3_wgrad_s2_kernel;
      static bool attr2 = false;
      if (!attr2) {
        if (hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) return IA_ERR_LAUNCH;
        attr2 = true;
      }
      hipLaunchKernelGGL(k2, dim3((unsigned)(per_group * a.groups)), dim3(256), G::LDS_BYTES, stream, a);
      hipLaunchKernelGGL(wgrad_fold_kernel, dim3((a.groups * bank + 255) / 256), dim3(256), 0, stream, a.part, dwhat, dbias, a.groups, (int)per_group, bank, wsize);
      return ia_check_launch();
    } else {
      return IA_ERR_UNSUPPORTED;
    }
  }
  if (a.xs == 2) {
    const size_t region = (size_t)per_group * a.groups * bank;
    float* const part = a.part;
    if constexpr (CI == 64 && CO == 64) {
      for (int v = 0; v < 4; ++v) {      // taps wanted: rows even -> ty = 0, 1, rows odd -> ty = 0; the same for the columns (tap = 3 ty + tx)
        a.offy = v >> 1; a.offx = v & 1; a.part = part + (size_t)v * region;
        const unsigned grid = (unsigned)(per_group * a.groups);
        const int rc = v == 0 ? wgrad_view_launch<0x1B>(a, grid, stream) : v == 1 ? wgrad_view_launch<0x09>(a, grid, stream)
                     : v == 2 ? wgrad_view_launch<0x03>(a, grid, stream) : wgrad_view_launch<0x01>(a, grid, stream);
        if (rc) return rc;
      }
    } else {
      return IA_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(wgrad_fold_s2_kernel, dim3((a.groups * bank + 255) / 256), dim3(256), 0, stream, part, dwhat, dbias, a.groups, (int)per_group, bank,
                       wsize, CI, region);
    return ia_check_launch();
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(per_group * a.groups)), dim3(256), G::LDS_BYTES, stream, a);
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3((a.groups * bank + 255) / 256), dim3(256), 0, stream, a.part, dwhat, dbias, a.groups, (int)per_group, bank, wsize);
  return ia_check_launch();
}
static size_t wgrad_workspace(int ci, int co, int groups) {
  const int per_cu = 2;                                        // upper bound of wgrad_t's choice
  return (size_t)((256L * per_cu) / groups) * groups * ((size_t)co * 9 * ci + co) * sizeof(float);
}
static int wgrad(const void* xp, const void* dyp, float* dwhat, float* dbias, int B, int H, int W, int ci, int co, int groups, void* ws, hipStream_t stream) {
  WArgs a;
  a.xp = (const bf16*)xp; a.dyp = (const bf16*)dyp; a.part = (float*)ws;
  a.B = B; a.H = H; a.W = W; a.Cin = groups * ci; a.Cout = groups * co; a.groups = groups;
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  a.XH = H; a.XW = W; a.xs = 1; a.offy = a.offx = 0; a.in_gstride = ci; a.dy_compact = 0;
  if (ci == 64 && co == 64) return wgrad_t<64, 64>(a, dwhat, dbias, stream);
  if (ci == 16 && co == 32) return wgrad_t<16, 32>(a, dwhat, dbias, stream);
  if (ci == 32 && co == 64) return wgrad_t<32, 64>(a, dwhat, dbias, stream);
  return IA_ERR_UNSUPPORTED;
}
// stride 2, 64 -> 64 channels per group: x [B, XH + 2, XW + 2, Cin] bordered, dy [B, H + 2, W + 2, Cout] bordered (H = (XH - 1) / 2 + 1);
// shared_input: every output group reads input channels 0 .. 63 (Cin = 64)
static int wgrad_s2(const void* xp, const void* dyp, float* dwhat, float* dbias, int B, int XH, int XW, int Cin, int Cout, int groups, int shared_input,
                    int dy_compact, void* ws, hipStream_t stream) {
  WArgs a;
  a.xp = (const bf16*)xp; a.dyp = (const bf16*)dyp; a.part = (float*)ws;
  a.H = (XH - 1) / 2 + 1; a.W = (XW - 1) / 2 + 1;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.groups = groups;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = (a.H + TH - 1) / TH;
  a.XH = XH; a.XW = XW; a.xs = 2; a.offy = a.offx = 0; a.in_gstride = shared_input ? 0 : 64; a.dy_compact = dy_compact;
  return wgrad_t<64, 64>(a, dwhat, dbias, stream);
}
static size_t wgrad_s2_workspace(int groups) { return 4 * wgrad_workspace(64, 64, groups); }

// slices > 1: groups = 1, Cin = 64, Cout = 64 * slices; wt = `slices` banks back to back (ia_conv3x3_flip_weights with groups = slices)
static int launch_s2_dgrad(const void* gp, const void* wt, void* dxp, int B, int H, int W, int Cin, int Cout, int groups, int slices, int g_compact,
                           hipStream_t stream) {
  S2DArgs a;
  a.gp = (const bf16*)gp; a.wt = (const bf16*)wt; a.dxp = (bf16*)dxp;
  a.B = B; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.Cin = Cin; a.Cout = Cout; a.groups = groups; a.g_choff = 0; a.accumulate = 0; a.g_compact = g_compact;
  a.tiles_x = (a.Wo + TW - 1) / TW; a.tiles_y = (a.Ho + TH - 1) / TH;      // cells: ceil(H / 2) x ceil(W / 2) = Ho x Wo
  using G = Geo<64, 64>;
  auto kern = conv3x3_s2_dgrad_kernel;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) return IA_ERR_LAUNCH;
    attr_set = true;
  }
  const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
  long per_group = 256L / groups;
  if (per_group > ntiles) per_group = ntiles;
  if (per_group < 1) per_group = 1;
  for (int sl = 0; sl < slices; ++sl) {
    a.g_choff = sl * 64; a.accumulate = sl > 0; a.wt = (const bf16*)wt + (size_t)sl * 64 * 9 * 64;
    hipLaunchKernelGGL(kern, dim3((unsigned)(per_group * groups)), dim3(256), G::LDS_BYTES, stream, a);
  }
  return ia_check_launch();
}

// launcher of the stride-2 forward kernel
static int launch_s2(const void* xp, const void* w, const float* bias, void* yp, int B, int XH, int XW, int Cin, int Cout, int groups, int shared_input,
                     int out_compact, hipStream_t stream) {
  S2Args a;
  a.xp = (const bf16*)xp; a.w = (const bf16*)w; a.bias = bias; a.yp = (bf16*)yp;
  a.B = B; a.XH = XH; a.XW = XW; a.H = (XH - 1) / 2 + 1; a.W = (XW - 1) / 2 + 1;
  a.Cin = Cin; a.Cout = Cout; a.groups = groups; a.in_gstride = shared_input ? 0 : 64; a.out_compact = out_compact;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = (a.H + TH - 1) / TH;
  using G = Geo<64, 64>;
  auto kern = conv3x3_s2_kernel<64>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) return IA_ERR_LAUNCH;
    attr_set = true;
  }
  const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
  long per_group = 256L / groups;
  if (per_group > ntiles) per_group = ntiles;
  if (per_group < 1) per_group = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(per_group * groups)), dim3(256), G::LDS_BYTES, stream, a);
  return ia_check_launch();
}
static bool wgrad_ok(int ci, int co) { return (ci == 64 && co == 64) || (ci == 16 && co == 32) || (ci == 32 && co == 64); }
}  // namespace dconv

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static int padded_ok(int B, int H, int W, int Cin, int Cout, int groups) {
  if (B <= 0 || H <= 0 || W <= 0 || groups <= 0 || Cin <= 0 || Cout <= 0 || Cin % groups || Cout % groups) return IA_ERR_ARG;
  const int ci = Cin / groups, co = Cout / groups;
  if (ci < 8 || co < 8 || (ci & (ci - 1)) || (co & (co - 1))) return IA_ERR_UNSUPPORTED;
  if ((size_t)B * (H + 2) * (W + 2) >= 0x7FFFFFFFull) return IA_ERR_ARG;       // rows are ints; the byte size is not limited
  return IA_OK;
}

extern "C" int ia_conv3x3_padded_fwd(const void* xp, const void* what, const float* bias, void* yp, int B, int H, int W, int Cin, int Cout,
                                     int groups, hipStream_t stream) {
  (void)hipGetLastError();
  const int rc = padded_ok(B, H, W, Cin, Cout, groups);
  if (rc) return rc;
  if (!xp || !what || !yp) return IA_ERR_ARG;
  const size_t Mp = (size_t)B * (H + 2) * (W + 2);
  const int ci = Cin / groups, co = Cout / groups;
  if (dconv::pair_ok(ci, co) && groups <= 64 && dconv::enabled()) return dconv::launch(xp, what, bias, yp, B, H, W, ci, co, groups, stream);
  IaViewGemm v{};
  v.A = xp; v.lda = Cin; v.B = what; v.ldb = 9 * ci; v.C = yp; v.ldc = Cout;
  v.M = (int)Mp; v.N = co; v.K = 9 * ci; v.bias = bias;
  v.a_view = 1; v.pw = W + 2; v.lca = ilog2(ci); v.lcbk = v.lcbn = 6;
  v.a_window = Mp * Cin * 2; v.b_window = (size_t)Cout * 9 * ci * 2;
  v.groups = groups; v.ga = ci; v.gb = (long)co * 9 * ci; v.gc = co; v.gbias = co;
  return ia_gemm_view(v, stream);
}

extern "C" int ia_conv3x3_padded_bwd_data(const void* dyp, const void* what, void* dxp, int B, int H, int W, int Cin, int Cout, int groups,
                                          hipStream_t stream) {
  (void)hipGetLastError();
  const int rc = padded_ok(B, H, W, Cin, Cout, groups);
  if (rc) return rc;
  if (!dyp || !what || !dxp) return IA_ERR_ARG;
  const size_t Mp = (size_t)B * (H + 2) * (W + 2);
  const int ci = Cin / groups, co = Cout / groups;
  IaViewGemm v{};
  v.A = dyp; v.lda = Cout; v.B = what; v.b_kstrided = 1; v.ldb = 9 * ci; v.C = dxp; v.ldc = Cin;
  v.M = (int)Mp; v.N = ci; v.K = 9 * co;
  v.a_view = -1; v.b_view = 1; v.pw = W + 2; v.lca = v.lcbk = ilog2(co); v.lcbn = ilog2(ci);
  v.a_window = Mp * Cout * 2; v.b_window = (size_t)Cout * 9 * ci * 2;
  v.groups = groups; v.ga = co; v.gb = (long)co * 9 * ci; v.gc = ci;
  return ia_gemm_view(v, stream);
}

// Data gradient through the direct kernel (16 -> 32, 32 -> 64 or 64 -> 64 channels per group): what_t [groups * ci][9 * co] = the
// tap-flipped, transposed filter bank written by ia_conv3x3_flip_weights.  Other shapes: IA_ERR_UNSUPPORTED (ia_conv3x3_padded_bwd_data).
extern "C" int ia_conv3x3_direct_supported(int Cin, int Cout, int groups) {
  return groups > 0 && groups <= 64 && Cin % groups == 0 && Cout % groups == 0 && dconv::pair_ok(Cin / groups, Cout / groups) &&
         dconv::pair_ok(Cout / groups, Cin / groups) && dconv::enabled();
}
extern "C" int ia_conv3x3_flip_weights(const void* what, void* what_t, int Cin, int Cout, int groups, hipStream_t stream) {
  (void)hipGetLastError();
  if (!what || !what_t) return IA_ERR_ARG;
  if (!ia_conv3x3_direct_supported(Cin, Cout, groups)) return IA_ERR_UNSUPPORTED;
  const int ci = Cin / groups, co = Cout / groups, total = groups * co * 9 * ci;
  hipLaunchKernelGGL(dconv::flip_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, (const bf16*)what, (bf16*)what_t, ci, co, total);
  return ia_check_launch();
}
extern "C" int ia_conv3x3_padded_bwd_data_t(const void* dyp, const void* what_t, void* dxp, int B, int H, int W, int Cin, int Cout, int groups,
                                            hipStream_t stream) {
  (void)hipGetLastError();
  const int rc = padded_ok(B, H, W, Cin, Cout, groups);
  if (rc) return rc;
  if (!dyp || !what_t || !dxp) return IA_ERR_ARG;
  if (!ia_conv3x3_direct_supported(Cin, Cout, groups)) return IA_ERR_UNSUPPORTED;
  return dconv::launch(dyp, what_t, nullptr, dxp, B, H, W, Cout / groups, Cin / groups, groups, stream);
}

extern "C" size_t ia_conv3x3_padded_workspace_bytes(int B, int H, int W, int Cin, int Cout, int groups) {
  if (padded_ok(B, H, W, Cin, Cout, groups)) return 0;
  const size_t Mp = (size_t)B * (H + 2) * (W + 2);
  const size_t gw = ia_gemm_view_workspace_bytes(Cout / groups, 9 * (Cin / groups), (int)Mp, groups), cs = ia_colsum_workspace_bytes((int)Mp, Cout);
  const size_t dw = dconv::wgrad_ok(Cin / groups, Cout / groups) ? dconv::wgrad_workspace(Cin / groups, Cout / groups, groups) : 0;
  const size_t m = gw > cs ? gw : cs;
  return m > dw ? m : dw;
}

// dwhat [Cout][9 * Cin/groups] fp32 overwritten, dbias [Cout] accumulated (may be NULL)
extern "C" int ia_conv3x3_padded_bwd_weight(const void* xp, const void* dyp, float* dwhat, float* dbias, int B, int H, int W, int Cin, int Cout,
                                            int groups, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  int rc = padded_ok(B, H, W, Cin, Cout, groups);
  if (rc) return rc;
  if (!xp || !dyp || !dwhat) return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_conv3x3_padded_workspace_bytes(B, H, W, Cin, Cout, groups)) return IA_ERR_WORKSPACE;
  const size_t Mp = (size_t)B * (H + 2) * (W + 2);
  const int ci = Cin / groups, co = Cout / groups;
  // (small maps keep the split-K GEMM: with a handful of tiles per workgroup the per-workgroup banks and their fold cost more than they
  // save -- 25 x 25: 0.049 against 0.036 ms at 32 images)
  if (dconv::wgrad_ok(ci, co) && groups <= 64 && (size_t)B * H * W >= dconv::wgrad_min_pixels() && dconv::enabled() && dconv::wgrad_enabled())
    return dconv::wgrad(xp, dyp, dwhat, dbias, B, H, W, ci, co, groups, workspace, stream);
  IaViewGemm v{};
  v.A = dyp; v.a_kstrided = 1; v.lda = Cout; v.B = xp; v.b_kstrided = 1; v.ldb = Cin; v.C = dwhat; v.c_is_f32 = 1; v.ldc = 9 * ci;
  v.M = co; v.N = 9 * ci; v.K = (int)Mp; v.workspace = workspace; v.workspace_bytes = workspace_bytes;
  v.b_view = 2; v.pw = W + 2; v.lca = v.lcbk = 6; v.lcbn = ilog2(ci);
  v.a_window = Mp * Cout * 2; v.b_window = Mp * Cin * 2;
  v.groups = groups; v.ga = co; v.gb = ci; v.gc = (long)co * 9 * ci;
  v.rsum_out = dbias;       // bias gradient = row sums of dy^T, taken inside the same GEMM (dyp's border rows are zero)
  return ia_gemm_view(v, stream);
}

// y = silu(x) * scale moving between the compact [B,H,W,C] and the zero-bordered [B,H+2,W+2,C] layouts (flags per side)
// ---------------------------------------------------------------------------------------------- 3x3 / stride 2 on the bordered domain
// xp [B, H + 2, W + 2, Cin] (zero border) -> yp [B, Ho + 2, Wo + 2, Cout] (border not written; y_compact: [B, Ho, Wo, Cout], and so
// is dyp), Ho = (H - 1) / 2 + 1: the strided
// convolutions of the NF-Net stage transitions (64 -> 64 channels per group) and of its stem (64 -> 128: output halves over one shared input
// slice) without a patch matrix -- forward by dconv::conv3x3_s2_kernel, weight gradient by four launches of the direct weight-gradient
// kernel on the parity views of x; the data gradient keeps the GEMM + gather form, between bordered layouts.  IA_CONV_S2_DIRECT=0 reports
// every shape as unsupported (callers fall back to ia_conv_nhwc_*).
static bool s2_enabled() {
  const char* e = getenv("IA_CONV_S2_DIRECT");
  return !e || atoi(e) != 0;
}
// virtual groups of 64 output channels; *shared = every one of them reads input channels 0 .. 63
static int s2_groups(int Cin, int Cout, int groups, int* shared) {
  *shared = 0;
  if (groups >= 1 && groups <= 64 && Cin == 64 * groups && Cout == 64 * groups) return groups;
  if (groups == 1 && Cin == 64 && Cout > 64 && Cout % 64 == 0 && Cout <= 64 * 64) { *shared = 1; return Cout / 64; }
  return 0;
}
extern "C" int ia_conv3x3_s2_supported(int Cin, int Cout, int groups) {
  int shared;
  return dconv::enabled() && dconv::wgrad_enabled() && s2_enabled() && s2_groups(Cin, Cout, groups, &shared) > 0;
}
static int s2_ok(int B, int H, int W, int Cin, int Cout, int groups, int* vg, int* shared) {
  if (B <= 0 || H <= 0 || W <= 0 || groups <= 0 || Cin <= 0 || Cout <= 0) return IA_ERR_ARG;
  *vg = s2_groups(Cin, Cout, groups, shared);
  if (!*vg) return IA_ERR_UNSUPPORTED;
  if ((size_t)B * (H + 2) * (W + 2) >= 0x7FFFFFFFull) return IA_ERR_ARG;
  return IA_OK;
}
extern "C" size_t ia_conv3x3_s2_padded_workspace_bytes(int B, int H, int W, int Cin, int Cout, int groups) {
  int vg, shared;
  if (s2_ok(B, H, W, Cin, Cout, groups, &vg, &shared)) return 0;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t dcols = (size_t)B * (Ho + 2) * (Wo + 2) * 9 * Cin * 2, wg = dconv::wgrad_s2_workspace(vg);
  return dcols > wg ? dcols : wg;
}
extern "C" int ia_conv3x3_s2_padded_fwd(const void* xp, const void* what, const float* bias, void* yp, int B, int H, int W, int Cin, int Cout,
                                        int groups, int y_compact, hipStream_t stream) {
  (void)hipGetLastError();
  int vg, shared;
  const int rc = s2_ok(B, H, W, Cin, Cout, groups, &vg, &shared);
  if (rc) return rc;
  if (!xp || !what || !yp) return IA_ERR_ARG;
  return dconv::launch_s2(xp, what, bias, yp, B, H, W, Cin, Cout, vg, shared, y_compact, stream);
}
// dwhat [Cout][9 * Cin / groups] fp32 (overwritten), dbias [Cout] (+=, may be NULL); dyp needs no particular border
extern "C" int ia_conv3x3_s2_padded_bwd_weight(const void* xp, const void* dyp, float* dwhat, float* dbias, int B, int H, int W, int Cin, int Cout,
                                               int groups, int y_compact, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  int vg, shared;
  const int rc = s2_ok(B, H, W, Cin, Cout, groups, &vg, &shared);
  if (rc) return rc;
  if (!xp || !dyp || !dwhat) return IA_ERR_ARG;
  if (!workspace || workspace_bytes < dconv::wgrad_s2_workspace(vg)) return IA_ERR_WORKSPACE;
  return dconv::wgrad_s2(xp, dyp, dwhat, dbias, B, H, W, Cin, Cout, vg, shared, y_compact, workspace, stream);
}
// dxp [B, H + 2, W + 2, Cin] (interior written) from dyp [B, Ho + 2, Wo + 2, Cout] (border rows may hold anything finite)
extern "C" int ia_conv3x3_s2_padded_bwd_data(const void* dyp, const void* what, void* dxp, int B, int H, int W, int Cin, int Cout, int groups,
                                             int y_compact, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  int vg, shared;
  int rc = s2_ok(B, H, W, Cin, Cout, groups, &vg, &shared);
  if (rc) return rc;
  if (!dyp || !what || !dxp) return IA_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, Cg = Cin / groups, Ng = Cout / groups, K = 9 * Cg;
  const size_t Mp = y_compact ? (size_t)B * Ho * Wo : (size_t)B * (Ho + 2) * (Wo + 2);
  if (!workspace || workspace_bytes < Mp * 9 * Cin * 2) return IA_ERR_WORKSPACE;
  bf16* dcols = (bf16*)workspace;
  for (int gi = 0; gi < groups && !rc; ++gi)
    rc = ia_gemm_bf16((const bf16*)dyp + gi * Ng, 0, Cout, (const bf16*)what + (size_t)gi * Ng * K, 1, K, dcols + (size_t)gi * K, 0, 9 * Cin, (int)Mp, K,
                      Ng, IA_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, stream);
  if (rc) return rc;
  const size_t total = (size_t)B * H * W * (Cin >> 3);
  hipLaunchKernelGGL(col2im3_s2_padded_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, dcols, (bf16*)dxp, H, W, Cin, Cg, Ho, Wo, y_compact, total);
  return ia_check_launch();
}

// The data gradient without the patch-matrix detour (ia_conv3x3_s2_dgrad_supported): Cin = Cout = 64 * groups (the stage transitions), and
// groups = 1, Cin = 64, Cout = 64 n (the stem) as n launches over the 64-channel slices of dy that add up in dx -- there what_t holds n
// banks, ia_conv3x3_flip_weights(what, what_t, Cout, Cout, n).
// what_t = the tap-flipped transposed bank of ia_conv3x3_flip_weights, dyp compact (y_compact) or bordered [B, Ho + 2, Wo + 2, Cout] with a ZERO border, dxp bordered
// [B, H + 2, W + 2, Cin] (interior written)
extern "C" int ia_conv3x3_s2_dgrad_supported(int Cin, int Cout, int groups) {
  const char* e = getenv("IA_CONV_S2_DGRAD");                    // 0: off; 1: 64-channel groups only; default: the sliced 64 -> 64 n form too
  const int mode = e ? atoi(e) : 2;
  int shared;
  const int vg = s2_groups(Cin, Cout, groups, &shared);
  return mode > 0 && dconv::enabled() && s2_enabled() && vg > 0 && (!shared || mode > 1);
}
extern "C" int ia_conv3x3_s2_padded_bwd_data_t(const void* dyp, const void* what_t, void* dxp, int B, int H, int W, int Cin, int Cout, int groups,
                                               int y_compact, hipStream_t stream) {
  (void)hipGetLastError();
  int vg, shared;
  const int rc = s2_ok(B, H, W, Cin, Cout, groups, &vg, &shared);
  if (rc) return rc;
  if (!dyp || !what_t || !dxp) return IA_ERR_ARG;
  return shared ? dconv::launch_s2_dgrad(dyp, what_t, dxp, B, H, W, Cin, Cout, 1, vg, y_compact, stream)
                : dconv::launch_s2_dgrad(dyp, what_t, dxp, B, H, W, Cin, Cout, vg, 1, y_compact, stream);
}

extern "C" int ia_silu_pad_fwd(const void* x, void* y, int B, int H, int W, int C, float scale, int in_padded, int out_padded, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return IA_ERR_ARG;
  const size_t rows = out_padded ? (size_t)B * (H + 2) * (W + 2) : (size_t)B * H * W;
  const size_t total = rows * (C >> 3);
  hipLaunchKernelGGL(silu_pad_fwd_kernel<true>, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, H, W, C, scale, in_padded,
                     out_padded, total);
  return ia_check_launch();
}

// plain copy between the compact [B,H,W,C] and the zero-bordered [B,H+2,W+2,C] layouts (a padded output gets a zero border,
// a padded input's border is ignored)
extern "C" int ia_pad_rows(const void* x, void* y, int B, int H, int W, int C, int in_padded, int out_padded, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return IA_ERR_ARG;
  const size_t rows = out_padded ? (size_t)B * (H + 2) * (W + 2) : (size_t)B * H * W;
  const size_t total = rows * (C >> 3);
  hipLaunchKernelGGL(silu_pad_fwd_kernel<false>, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, H, W, C, 1.f, in_padded,
                     out_padded, total);
  return ia_check_launch();
}

// dx (layout of x: in_padded) = dy (layout of y: out_padded) * scale * silu'(x)
extern "C" int ia_silu_pad_bwd(const void* dy, const void* x, void* dx, int B, int H, int W, int C, float scale, int in_padded, int out_padded,
                               hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !x || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return IA_ERR_ARG;
  const size_t rows = in_padded ? (size_t)B * (H + 2) * (W + 2) : (size_t)B * H * W;
  const size_t total = rows * (C >> 3);
  hipLaunchKernelGGL(silu_pad_bwd_kernel, dim3(blocks_of(total)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, (bf16*)dx, H, W, C, scale,
                     in_padded, out_padded, total);
  return ia_check_launch();
}
