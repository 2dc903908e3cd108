// Row-wise LayerNorm family (HBM-bound tails of the encoder layer), gfx950.
//
//   ia_ln_fwd : z = residual + dropout(x + bias);  y = LN(z) * gamma + beta      (BERT post-LN tail,
//               reference: transformers RobertaSelfOutput / RobertaOutput called from
//               src/models/text.py:1241; with residual = bias = null it is the plain pre-LN of ViT)
//   ia_ln_bwd : dz = LN'(dy) (+ optional residual-path gradient dres), per-block partial sums of
//               dgamma / dbeta / dbias, and the dropout-masked gradient of the GEMM branch
//   ia_colsum : column sums of a [M,N] bf16 matrix (bias gradients) as per-block partials
//   ia_reduce_partials : deterministic second stage, accumulates into the fp32 gradient arena
//
// One 64-lane wave owns one row at a time and keeps it in registers (16 B loads, 8 bf16 per lane
// per 512-column slab); statistics are fp32, two-pass (mean, then centred variance).
#include "common.h"

namespace {

constexpr int MAXV = 8;  // up to 4096 columns

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ bias,
                                                     const bf16* __restrict__ res, bf16* __restrict__ z_out,
                                                     bf16* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int M, int H, float eps,
                                                     uint32_t thr16, float inv_keep, uint32_t seed, uint32_t stream) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = i * 512 + lane * 8;
    if (col < H) {
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + (size_t)row * H + col);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = bf2f(xv[j]);
      if (bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + col);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + col + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[i][j] += b0[j]; v[i][4 + j] += b1[j]; }
      }
      if (thr16) {
        const uint32_t base = (uint32_t)(((size_t)row * H + col) >> 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t r = ia_rng(seed, stream, base + j);
          v[i][2 * j] = ((r & 0xFFFFu) >= thr16) ? v[i][2 * j] * inv_keep : 0.f;
          v[i][2 * j + 1] = ((r >> 16) >= thr16) ? v[i][2 * j + 1] * inv_keep : 0.f;
        }
      }
      if (res) {
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(res + (size_t)row * H + col);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] += bf2f(rv[j]);
      }
      {
        // statistics are taken on the bf16-rounded z so fwd and bwd see the same x-hat -- also when z is not stored (forward-only
        // layers, ia_layer_fwd_infer): evaluation then reproduces the training forward bit for bit
        bf16x8 zv;
#pragma unroll
        for (int j = 0; j < 8; ++j) zv[j] = f2bf(v[i][j]);
        if (z_out) *reinterpret_cast<bf16x8*>(z_out + (size_t)row * H + col) = zv;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = bf2f(zv[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = i * 512 + lane * 8;
    if (col < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = i * 512 + lane * 8;
    if (col < H) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + col), g1 = *reinterpret_cast<const f32x4*>(gamma + col + 4);
      f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
      if (beta) { b0 = *reinterpret_cast<const f32x4*>(beta + col); b1 = *reinterpret_cast<const f32x4*>(beta + col + 4); }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = f2bf((v[i][j] - mean) * rstd * g0[j] + b0[j]);
        o[4 + j] = f2bf((v[i][4 + j] - mean) * rstd * g1[j] + b1[j]);
      }
      *reinterpret_cast<bf16x8*>(y + (size_t)row * H + col) = o;
    }
  }
}

// partial layout: part[blk][3][H] : 0 = dgamma, 1 = dbeta, 2 = dbias (gradient of the GEMM branch)
template <int NV, bool FILTER>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dy, const bf16* dy2, const bf16* __restrict__ dres,
                                                     const bf16* __restrict__ z, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     bf16* dz, bf16* __restrict__ dx, float* __restrict__ part,
                                                     int M, int H, uint32_t thr16, float inv_keep, uint32_t seed,
                                                     uint32_t stream, const uint8_t* __restrict__ live) {
  __shared__ float red[3][4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[NV][8], ab[NV][8], ax[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; ax[i][j] = 0.f; }
  float gm[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = i * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) gm[i][j] = (col < H) ? gamma[col + j] : 0.f;
  }
  // Rows are software-pipelined: the 16-byte loads of the wave's next row are in flight while the current row goes
  // through its two reductions (one row alone leaves HBM idle for the whole reduce -> normalise -> store chain).
  const int row0 = blockIdx.x * 4 + wave, rstep = gridDim.x * 4;
  bf16x8 dv[NV], zv[NV], rv[NV];
  float mu = 0.f, rs = 0.f;
  // live (round 6, may be null): live[row] == 0 marks a row whose incoming gradients are exactly zero by the caller's guarantee (a
  // masked position of an encoder whose heads read none of them, ia_layer_cfg::masked_rows_dead): its outputs are zero rows and it adds
  // nothing to the column sums, so none of its four input streams is read -- the kernel is HBM-bound and 45 % of the bench's rows are
  // padding.  The row's zeros are still written (the GEMMs behind read them).
  // (FILTER is a template parameter: with the test compiled into the unfiltered kernel every launch -- the ViT's too -- ran 7 % slower;
  // the flags of the wave's next 64 rows are fetched by ONE load, lane i holding the flag of its i-th row ahead, and balloted into a
  // scalar mask: a per-row flag load in front of the row's own loads put a global-memory round trip into every iteration)
  bool dead = false;
  uint64_t live_bits = ~0ull;
  int bits_left = 0;
  const int lane_ = threadIdx.x & 63;
  auto next_dead = [&](int row) -> bool {
    if constexpr (!FILTER) return false;
    if (bits_left == 0) {
      const long r = (long)row + (long)lane_ * rstep;
      const bool lv = r < M ? live[r] != 0 : true;
      live_bits = __ballot(lv);
      bits_left = 64;
    }
    const bool dd = (live_bits & 1ull) == 0ull;
    live_bits >>= 1; --bits_left;
    return dd;
  };
  auto fetch = [&](int row, bf16x8 (&d)[NV], bf16x8 (&z_)[NV], bf16x8 (&r)[NV], float& m_, float& s_, bool& dd) {
    dd = next_dead(row);                                  // wave-uniform: one row per wave
    if (FILTER && dd) return;
    m_ = mean[row]; s_ = rstd[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int col = i * 512 + lane * 8;
      if (col < H) {
        d[i] = *reinterpret_cast<const bf16x8*>(dy + (size_t)row * H + col);
        if (dy2) {   // second upstream gradient (the residual path of a post-LN layer): summed in fp32, dz may alias dy2
          const bf16x8 e = *reinterpret_cast<const bf16x8*>(dy2 + (size_t)row * H + col);
#pragma unroll
          for (int j = 0; j < 8; ++j) d[i][j] = f2bf(bf2f(d[i][j]) + bf2f(e[j]));
        }
        z_[i] = *reinterpret_cast<const bf16x8*>(z + (size_t)row * H + col);
        if (dres) r[i] = *reinterpret_cast<const bf16x8*>(dres + (size_t)row * H + col);
      }
    }
  };
  if (row0 < M) fetch(row0, dv, zv, rv, mu, rs, dead);
  for (int row = row0; row < M; row += rstep) {
    bf16x8 ndv[NV], nzv[NV], nrv[NV];
    float nmu = 0.f, nrs = 0.f;
    bool ndead = false;
    if (row + rstep < M) fetch(row + rstep, ndv, nzv, nrv, nmu, nrs, ndead);
    if (FILTER && dead) {
      bf16x8 zero;
#pragma unroll
      for (int j = 0; j < 8; ++j) zero[j] = f2bf(0.f);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int col = i * 512 + lane * 8;
        if (col < H) {
          *reinterpret_cast<bf16x8*>(dz + (size_t)row * H + col) = zero;
          if (thr16) *reinterpret_cast<bf16x8*>(dx + (size_t)row * H + col) = zero;
        }
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) { dv[i] = ndv[i]; zv[i] = nzv[i]; rv[i] = nrv[i]; }
      mu = nmu; rs = nrs; dead = ndead;
      continue;
    }
    float g[NV][8], xh[NV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int col = i * 512 + lane * 8;
      if (col < H) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = bf2f(dv[i][j]);
          const float xhat = (bf2f(zv[i][j]) - mu) * rs;
          xh[i][j] = xhat;
          ag[i][j] += d * xhat;
          ab[i][j] += d;
          g[i][j] = d * gm[i][j];
          s1 += g[i][j];
          s2 += g[i][j] * xhat;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { g[i][j] = 0.f; xh[i][j] = 0.f; }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int col = i * 512 + lane * 8;
      if (col < H) {
        float o[8];
        bf16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (g[i][j] - s1 - xh[i][j] * s2);
        if (dres) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += bf2f(rv[i][j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = f2bf(o[j]);
        *reinterpret_cast<bf16x8*>(dz + (size_t)row * H + col) = ov;
        if (thr16) {
          const uint32_t base = (uint32_t)(((size_t)row * H + col) >> 1);
          bf16x8 xv;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t r = ia_rng(seed, stream, base + j);
            const float a = ((r & 0xFFFFu) >= thr16) ? o[2 * j] * inv_keep : 0.f;
            const float b = ((r >> 16) >= thr16) ? o[2 * j + 1] * inv_keep : 0.f;
            xv[2 * j] = f2bf(a); xv[2 * j + 1] = f2bf(b);
            ax[i][2 * j] += a; ax[i][2 * j + 1] += b;
          }
          *reinterpret_cast<bf16x8*>(dx + (size_t)row * H + col) = xv;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) ax[i][j] += o[j];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) { dv[i] = ndv[i]; zv[i] = nzv[i]; rv[i] = nrv[i]; }
    mu = nmu; rs = nrs; dead = ndead;
  }
  // cross-wave reduction, 512 columns at a time
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[0][wave][lane * 8 + j] = ag[i][j];
      red[1][wave][lane * 8 + j] = ab[i][j];
      red[2][wave][lane * 8 + j] = ax[i][j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 512 * 3; c += 256) {
      const int w = c / 512, cc = c % 512;
      const int col = i * 512 + cc;
      if (col < H) part[((size_t)blockIdx.x * 3 + w) * H + col] = red[w][0][cc] + red[w][1][cc] + red[w][2][cc] + red[w][3][cc];
    }
  }
}

// column sums: block (256 threads) covers 512 columns x a strided set of rows
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, float* __restrict__ part, int M, int N, int ld) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 512 + lane * 8;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    for (int row = blockIdx.y * 4 + wave; row < M; row += gridDim.y * 4) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (size_t)row * ld + col);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += bf2f(v[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = a[j];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += 256) {
    const int cc = blockIdx.x * 512 + c;
    if (cc < N) part[(size_t)blockIdx.y * N + cc] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
  }
}

// out_w[c] (+)= sum_b part[(b*nw + w)*H + c] for w < nw; 32 columns x 32 row-slices per block so the
// second stage is bandwidth- not latency-bound (a column-per-thread serial sum took 0.7 ms).
struct ReduceOuts { float* p[3]; };
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ part, int nblk, int nw, int H, ReduceOuts outs,
                                                               int accumulate) {
  __shared__ float red[32][33];
  const int cx = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;          // column in [0, nw*H)
  float s = 0.f;
  if (c < nw * H) {
    const int w = c / H, col = c % H;
    // four independent partial sums: four loads in flight per thread (one dependent add chain waited for every load: 16 round trips
    // of ~0.6 us each were the kernel's 15 us; the order of the sum stays fixed, i.e. deterministic)
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = sl;
    for (; b + 96 < nblk; b += 128) {
      s += part[((size_t)b * nw + w) * H + col];
      s1 += part[((size_t)(b + 32) * nw + w) * H + col];
      s2 += part[((size_t)(b + 64) * nw + w) * H + col];
      s3 += part[((size_t)(b + 96) * nw + w) * H + col];
    }
    for (; b < nblk; b += 32) s += part[((size_t)b * nw + w) * H + col];
    s = (s + s1) + (s2 + s3);
  }
  red[sl][cx] = s;
  __syncthreads();
  if (sl == 0 && c < nw * H) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][cx];
    const int w = c / H, col = c % H;
    float* o = outs.p[w];
    if (o) o[col] = accumulate ? o[col] + t : t;
  }
}

int ln_blocks(int M) { int b = (M + 3) / 4; return b < 512 ? b : 512; }

}  // namespace

extern "C" int ia_ln_fwd(const void* x, const float* bias, const void* residual, void* z_out, void* y, float* mean,
                         float* rstd, const float* gamma, const float* beta, int M, int H, float eps, float drop_p,
                         uint32_t seed, uint32_t stream_id, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!x || !y || !mean || !rstd || !gamma || M <= 0 || H <= 0 || (H & 7) || H > 512 * MAXV) return IA_ERR_ARG;
  const uint32_t thr16 = drop_p > 0.f ? (uint32_t)(drop_p * 65536.f + 0.5f) : 0u;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - (float)thr16 / 65536.f) : 1.f;
  const int nv = (H + 511) / 512;
  dim3 grid((M + 3) / 4), blk(256);
#define IA_LN_FWD(NV) hipLaunchKernelGGL((ln_fwd_kernel<NV>), grid, blk, 0, stream, (const bf16*)x, bias, (const bf16*)residual, \
    (bf16*)z_out, (bf16*)y, mean, rstd, gamma, beta, M, H, eps, thr16, inv_keep, seed, stream_id)
  switch (nv) {
    case 1: IA_LN_FWD(1); break;
    case 2: IA_LN_FWD(2); break;
    case 3: IA_LN_FWD(3); break;
    case 4: IA_LN_FWD(4); break;
    default: IA_LN_FWD(8); break;
  }
#undef IA_LN_FWD
  return ia_check_launch();
}

extern "C" size_t ia_ln_bwd_workspace_bytes(int M, int H) { return (size_t)ln_blocks(M) * 3 * H * sizeof(float); }

extern "C" int ia_ln_bwd2_rows(const void* dy, const void* dy2, const void* dres, const void* z, const float* mean, const float* rstd,
                               const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias, int M, int H,
                               float drop_p, uint32_t seed, uint32_t stream_id, const uint8_t* row_live, void* workspace,
                               size_t workspace_bytes, int accumulate, hipStream_t stream);

extern "C" int ia_ln_bwd2(const void* dy, const void* dy2, const void* dres, const void* z, const float* mean, const float* rstd,
                          const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias, int M, int H,
                          float drop_p, uint32_t seed, uint32_t stream_id, void* workspace, size_t workspace_bytes,
                          int accumulate, hipStream_t stream);

// dgamma/dbeta/dbias may be null (skipped); all three accumulate (+=) into fp32 when `accumulate`.
extern "C" int ia_ln_bwd(const void* dy, const void* dres, const void* z, const float* mean, const float* rstd,
                         const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias, int M, int H,
                         float drop_p, uint32_t seed, uint32_t stream_id, void* workspace, size_t workspace_bytes,
                         int accumulate, hipStream_t stream) {
  return ia_ln_bwd2(dy, nullptr, dres, z, mean, rstd, gamma, dz, dx, dgamma, dbeta, dbias, M, H, drop_p, seed, stream_id, workspace,
                    workspace_bytes, accumulate, stream);
}

// The same with a second upstream gradient dy2 (may be null): the LayerNorm output's gradient is dy + dy2.  A post-LN layer's
// output feeds the next sub-block AND its residual connection; taking both gradients here replaces the "+ residual gradient"
// epilogue of the GEMM that produces dy (one more row stream for an HBM-bound kernel instead of an aux operand in a GEMM epilogue).
// dz may alias dy2.
extern "C" int ia_ln_bwd2(const void* dy, const void* dy2, const void* dres, const void* z, const float* mean, const float* rstd,
                          const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias, int M, int H,
                          float drop_p, uint32_t seed, uint32_t stream_id, void* workspace, size_t workspace_bytes,
                          int accumulate, hipStream_t stream) {
  return ia_ln_bwd2_rows(dy, dy2, dres, z, mean, rstd, gamma, dz, dx, dgamma, dbeta, dbias, M, H, drop_p, seed, stream_id, nullptr, workspace,
                         workspace_bytes, accumulate, stream);
}

// ia_ln_bwd2 with a row filter (round 6): row_live [M] uint8 or NULL; row_live[m] == 0 = the caller guarantees dy, dy2 and dres are zero
// in row m (a masked position whose hidden state no head reads): the row's inputs are not fetched, its dz / dx rows are written as
// zeros, it adds nothing to dgamma / dbeta / dbias.  Identical results on such inputs.
extern "C" int ia_ln_bwd2_rows(const void* dy, const void* dy2, const void* dres, const void* z, const float* mean, const float* rstd,
                               const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias, int M, int H,
                               float drop_p, uint32_t seed, uint32_t stream_id, const uint8_t* row_live, void* workspace,
                               size_t workspace_bytes, int accumulate, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!dy || !z || !mean || !rstd || !gamma || !dz || M <= 0 || (H & 7) || H > 512 * MAXV) return IA_ERR_ARG;
  if (workspace_bytes < ia_ln_bwd_workspace_bytes(M, H) || !workspace) return IA_ERR_WORKSPACE;
  const uint32_t thr16 = drop_p > 0.f ? (uint32_t)(drop_p * 65536.f + 0.5f) : 0u;
  if (thr16 && !dx) return IA_ERR_ARG;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - (float)thr16 / 65536.f) : 1.f;
  const int nv = (H + 511) / 512, nb = ln_blocks(M);
  float* part = (float*)workspace;
  dim3 grid(nb), blk(256);
#define IA_LN_BWD_(NV, F) hipLaunchKernelGGL((ln_bwd_kernel<NV, F>), grid, blk, 0, stream, (const bf16*)dy, (const bf16*)dy2, (const bf16*)dres, (const bf16*)z, \
    mean, rstd, gamma, (bf16*)dz, (bf16*)dx, part, M, H, thr16, inv_keep, seed, stream_id, row_live)
#define IA_LN_BWD(NV) do { if (row_live) IA_LN_BWD_(NV, true); else IA_LN_BWD_(NV, false); } while (0)
  switch (nv) {
    case 1: IA_LN_BWD(1); break;
    case 2: IA_LN_BWD(2); break;
    case 3: IA_LN_BWD(3); break;
    case 4: IA_LN_BWD(4); break;
    default: IA_LN_BWD(8); break;
  }
#undef IA_LN_BWD
#undef IA_LN_BWD_
  int rc = ia_check_launch();
  if (rc) return rc;
  ReduceOuts outs{{dgamma, dbeta, dbias}};
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((3 * H + 31) / 32), dim3(1024), 0, stream, part, nb, 3, H, outs, accumulate);
  return ia_check_launch();
}

int ia_sum_rows_f32(const float* part, int nblk, int N, float* out, int accumulate, hipStream_t stream) {
  ReduceOuts outs{{out, nullptr, nullptr}};
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((N + 31) / 32), dim3(1024), 0, stream, part, nblk, 1, N, outs, accumulate);
  return ia_check_launch();
}

// Narrow matrices (N < 512 dividing 512, rows contiguous: conv-tower bias gradients with 16..256 channels) are summed as
// [M*N/512, 512]: every lane of the column-sum kernel stays busy, and the 512/N column groups fold for free because the
// partial buffer [rows][512] read as [rows * 512/N][N] is exactly what the second stage sums over.
static int colsum_fold(int M, int N, int ld) {
  if (ld != N || N >= 512 || (512 % N)) return 1;
  const int f = 512 / N;
  return (M % f) ? 1 : f;
}

extern "C" size_t ia_colsum_workspace_bytes(int M, int N) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  int rb = (M + 3) / 4; if (rb > 256) rb = 256;
  const int W = N < 512 ? 512 : N;
  return (size_t)rb * W * sizeof(float);
}

extern "C" int ia_colsum(const void* x, int ld, int M, int N, float* out, int accumulate, void* workspace,
                         size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!x || !out || M <= 0 || N <= 0 || (N & 7) || (ld & 7)) return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_colsum_workspace_bytes(M, N)) return IA_ERR_WORKSPACE;
  const int fold = colsum_fold(M, N, ld);
  const int Mw = M / fold, Nw = N * fold, ldw = fold > 1 ? Nw : ld;
  int rb = (Mw + 3) / 4; if (rb > 256) rb = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3((Nw + 511) / 512, rb), dim3(256), 0, stream, (const bf16*)x, (float*)workspace, Mw, Nw, ldw);
  ReduceOuts outs{{out, nullptr, nullptr}};
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((N + 31) / 32), dim3(1024), 0, stream, (const float*)workspace, rb * fold, 1, N, outs, accumulate);
  return ia_check_launch();
}
