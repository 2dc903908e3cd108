// Input-side kernels of the towers (all HBM-bound gathers / layout changes), gfx950.
//
//   ia_embed_ln_fwd/bwd   word + token-type + position gather, add, LayerNorm(eps), dropout
//                         (reference src/models/base.py:238-279 RobertaEmbeddings.forward; the
//                          image splice of :530-546 and the PKGM splice of :422-432 enter through
//                          the `extra` row table + per-token redirect index)
//   ia_im2col_patch16     NCHW fp32 image -> [B*np, C*P*P] bf16 patch rows (patch-embed conv of the
//                         ViT tower == GEMM on these rows; reference multimodal.py:811 -> timm PatchEmbed)
//   ia_vit_tokens_fwd/bwd cls token concat + position embedding add
//   ia_gather_rows_fwd/bwd pick token rows (CLS) out of a hidden-state matrix as fp32, with dropout
#include "common.h"

namespace {

template <int NV>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ tts,
                                                           const int64_t* __restrict__ pids, const int32_t* __restrict__ xidx,
                                                           const float* __restrict__ word, const float* __restrict__ type,
                                                           const float* __restrict__ pos, const float* __restrict__ extra,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16* __restrict__ z_out, bf16* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, int M, int H, float eps, uint32_t thr16,
                                                           float inv_keep, uint32_t seed, uint32_t stream) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int xi = xidx ? xidx[row] : -1;
  const float* wrow = xi >= 0 ? extra + (size_t)xi * H : word + (size_t)ids[row] * H;
  const float* trow = type + (size_t)tts[row] * H;
  const float* prow = pos + (size_t)pids[row] * H;
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = i * 512 + lane * 8;
    if (col < H) {
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wrow + col + 4 * hlf);
        const f32x4 b = *reinterpret_cast<const f32x4*>(trow + col + 4 * hlf);
        const f32x4 c = *reinterpret_cast<const f32x4*>(prow + col + 4 * hlf);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][4 * hlf + j] = (a[j] + b[j]) + c[j];
      }
      bf16x8 zv;
#pragma unroll
      for (int j = 0; j < 8; ++j) { zv[j] = f2bf(v[i][j]); v[i][j] = bf2f(zv[j]); s += v[i][j]; }
      *reinterpret_cast<bf16x8*>(z_out + (size_t)row * H + col) = zv;
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
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * gamma[col + j] + beta[col + j];
      if (thr16) {
        const uint32_t base = (uint32_t)(((size_t)row * H + col) >> 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t r = ia_rng(seed, stream, base + j);
          o[2 * j] = ((r & 0xFFFFu) >= thr16) ? o[2 * j] * inv_keep : 0.f;
          o[2 * j + 1] = ((r >> 16) >= thr16) ? o[2 * j + 1] * inv_keep : 0.f;
        }
      }
      bf16x8 ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) ov[j] = f2bf(o[j]);
      *reinterpret_cast<bf16x8*>(y + (size_t)row * H + col) = ov;
    }
  }
}

// dy -> (dropout mask) -> LN backward -> scatter-add into the fp32 gradient tables.
// part[blk][2][H]: dgamma, dbeta partials (second stage = ia_reduce_partials).
template <int NV>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const int64_t* __restrict__ ids,
                                                           const int64_t* __restrict__ tts, const int64_t* __restrict__ pids,
                                                           const int32_t* __restrict__ xidx, const int32_t* __restrict__ order,
                                                           float* __restrict__ dword,
                                                           float* __restrict__ dtype, float* __restrict__ dpos,
                                                           float* __restrict__ dextra, float* __restrict__ part, int M, int H, int L,
                                                           int word_pad, int pos_pad, uint32_t thr16, float inv_keep, uint32_t seed,
                                                           uint32_t stream) {
  __shared__ float red[2][4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[NV][8], ab[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; }
  // A wave owns one position l of the sequence layout [S, L] and walks (its slice of) the S sequences: the rows it
  // visits share their position id and almost always their token type, so those two gradients are run-length
  // accumulated in registers and flushed with one atomicAdd per column per run (33M contended atomics on 2+512
  // table rows became ~L*G*H); word-row gradients go out per token (random rows, no contention).
  // `order` (rows sorted by position id; unpadded token rows have no [S, L] grid): the wave owns the l-th chunk of S
  // consecutive entries of that list instead of a column of the grid
  const int S = order ? (M + L - 1) / L : M / L;
  const int s_lo = (int)(((long)S * blockIdx.y) / gridDim.y), s_hi = (int)(((long)S * (blockIdx.y + 1)) / gridDim.y);
  float accp[NV][8], acct[NV][8];
  for (int l = blockIdx.x * 4 + wave; l < L; l += gridDim.x * 4) {
  long curp = -1, curt = -1;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { accp[i][j] = 0.f; acct[i][j] = 0.f; }
  // Column mapping of this kernel: register j of lane l stands for column i * 512 + j * 64 + l, so that ONE atomic instruction of
  // the wave covers 64 consecutive floats = two cache lines (round 3 had 8 consecutive columns per lane: every instruction touched
  // 16 lines and the 8 instructions of a chunk the same 16 again -- the table updates, 134 M fp32 atomics per step at the bench
  // size, were 2.3 ms; 8 x fewer line operations now).  The loads are 2-byte loads of the same 128-byte row segments.
  auto flush = [&](float (&acc)[NV][8], float* table, long rowid) {
    if (table && rowid >= 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int col = i * 512 + j * 64 + lane;
          if (col < H) atomicAdd(table + (size_t)rowid * H + col, acc[i][j]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  };
  for (int sq = s_lo; sq <= s_hi; ++sq) {
    if (sq == s_hi || (order && (long)l * S + sq >= M)) { flush(accp, dpos, curp); flush(acct, dtype, curt); break; }
    const int row = order ? order[(long)l * S + sq] : sq * L + l;
    const float mu = mean[row], rs = rstd[row];
    float g[NV][8], xh[NV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = i * 512 + j * 64 + lane;
        if (col < H) {
          float d = bf2f(dy[(size_t)row * H + col]);
          if (thr16) {      // the forward's mask: one 32-bit draw per pair of neighbouring elements, low half = even element
            const size_t e = (size_t)row * H + col;
            const uint32_t r = ia_rng(seed, stream, (uint32_t)(e >> 1));
            d = (((e & 1) ? (r >> 16) : (r & 0xFFFFu)) >= thr16) ? d * inv_keep : 0.f;
          }
          const float xhat = (bf2f(z[(size_t)row * H + col]) - mu) * rs;
          xh[i][j] = xhat;
          ag[i][j] += d * xhat;
          ab[i][j] += d;
          g[i][j] = d * gamma[col];
          s1 += g[i][j];
          s2 += g[i][j] * xhat;
        } else { g[i][j] = 0.f; xh[i][j] = 0.f; }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    const int xi = xidx ? xidx[row] : -1;
    const int64_t wid = ids[row], pid = pids[row];
    float* wdst = xi >= 0 ? (dextra ? dextra + (size_t)xi * H : nullptr)
                          : ((dword && wid != word_pad) ? dword + (size_t)wid * H : nullptr);
    const long tid_now = tts[row];
    const long pid_now = (pid != pos_pad) ? (long)pid : -1;      // padding_idx row gets no gradient
    if (pid_now != curp) { flush(accp, dpos, curp); curp = pid_now; }
    if (tid_now != curt) { flush(acct, dtype, curt); curt = tid_now; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = i * 512 + j * 64 + lane;
        if (col < H) {
          const float o = rs * (g[i][j] - s1 - xh[i][j] * s2);
          if (wdst) atomicAdd(wdst + col, o);
          accp[i][j] += o;
          acct[i][j] += o;
        }
      }
    }
  }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][wave][j * 64 + lane] = ag[i][j]; red[1][wave][j * 64 + lane] = ab[i][j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 512 * 2; c += 256) {
      const int w = c / 512, cc = c % 512;
      const int col = i * 512 + cc;
      if (col < H) part[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + w) * H + col] = red[w][0][cc] + red[w][1][cc] + red[w][2][cc] + red[w][3][cc];
    }
  }
}

// out_w[c] += sum_b part[(b*2 + w)*H + c]; 32 columns x 32 row-slices per block
__global__ __launch_bounds__(1024) void reduce2_kernel(const float* __restrict__ part, int nblk, int H, float* __restrict__ o0,
                                                       float* __restrict__ o1) {
  __shared__ float red[32][33];
  const int cx = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float s = 0.f;
  if (c < 2 * H) {
    const int w = c / H, col = c % H;
    for (int b = sl; b < nblk; b += 32) s += part[((size_t)b * 2 + w) * H + col];
  }
  red[sl][cx] = s;
  __syncthreads();
  if (sl == 0 && c < 2 * H) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][cx];
    float* o = (c / H) ? o1 : o0;
    if (o) o[c % H] += t;
  }
}

// images [B, C, S, S] fp32 NCHW -> patches [B * (S/P)^2, C*P*P] bf16, column = (c, ph, pw); P = 16
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B, int C, int S, int P) {
  const int np1 = S / P;
  const int cols = C * P * P;
  const size_t total = (size_t)B * np1 * np1 * cols / 8;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const size_t e = t * 8;
    const int col = (int)(e % cols);
    const size_t prow = e / cols;
    const int pw = col % P, ph = (col / P) % P, c = col / (P * P);
    const int px = (int)(prow % np1), py = (int)((prow / np1) % np1), b = (int)(prow / ((size_t)np1 * np1));
    const float* src = img + (((size_t)b * C + c) * S + (py * P + ph)) * S + px * P + pw;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), d = *reinterpret_cast<const f32x4*>(src + 4);
    bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(d[0]), f2bf(d[1]), f2bf(d[2]), f2bf(d[3])};
    *reinterpret_cast<bf16x8*>(out + e) = o;
  }
}

// tokens[b, 0] = cls + pos[0]; tokens[b, 1+p] = patch[b, p] + pos[1+p]
__global__ __launch_bounds__(256) void vit_tokens_fwd_kernel(const bf16* __restrict__ patch, const float* __restrict__ cls,
                                                             const float* __restrict__ pos, bf16* __restrict__ tok, int B, int NP, int H) {
  const size_t total = (size_t)B * (NP + 1) * H / 8;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const size_t e = t * 8;
    const int col = (int)(e % H);
    const size_t row = e / H;
    const int n = (int)(row % (NP + 1)), b = (int)(row / (NP + 1));
    float v[8];
    if (n == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = cls[col + j];
    } else {
      const bf16x8 pv = *reinterpret_cast<const bf16x8*>(patch + ((size_t)b * NP + n - 1) * H + col);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = bf2f(pv[j]);
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(v[j] + pos[(size_t)n * H + col + j]);
    *reinterpret_cast<bf16x8*>(tok + e) = o;
  }
}

// dpatch[b,p] = dtok[b,1+p]; dpos[n] (+)= sum_b dtok[b,n]; dcls (+)= sum_b dtok[b,0]
__global__ __launch_bounds__(256) void vit_tokens_bwd_kernel(const bf16* __restrict__ dtok, bf16* __restrict__ dpatch,
                                                             float* __restrict__ dcls, float* __restrict__ dpos, int B, int NP, int H,
                                                             int accumulate) {
  const size_t total = (size_t)(NP + 1) * H / 8;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const size_t e = t * 8;
    const int col = (int)(e % H);
    const int n = (int)(e / H);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b) {
      const bf16x8 g = *reinterpret_cast<const bf16x8*>(dtok + ((size_t)b * (NP + 1) + n) * H + col);
      if (n > 0) *reinterpret_cast<bf16x8*>(dpatch + ((size_t)b * NP + n - 1) * H + col) = g;
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += bf2f(g[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float* p = dpos + (size_t)n * H + col + j;
      *p = accumulate ? *p + s[j] : s[j];
      if (n == 0) { float* c = dcls + col + j; *c = accumulate ? *c + s[j] : s[j]; }
    }
  }
}

// out[b, :] = dropout(src[rows[b], :]) as fp32
__global__ __launch_bounds__(256) void gather_rows_fwd_kernel(const bf16* __restrict__ src, int ld, const int32_t* __restrict__ rows,
                                                              float* __restrict__ out, int B, int H, uint32_t thr16, float inv_keep,
                                                              uint32_t seed, uint32_t stream) {
  const size_t total = (size_t)B * H;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int col = (int)(t % H), b = (int)(t / H);
    float v = bf2f(src[(size_t)rows[b] * ld + col]);
    if (thr16) {
      const uint32_t r = ia_rng(seed, stream, (uint32_t)(t >> 1));
      const uint32_t u = (t & 1) ? (r >> 16) : (r & 0xFFFFu);
      v = u >= thr16 ? v * inv_keep : 0.f;
    }
    out[t] = v;
  }
}

// dsrc[rows[b], :] += dropout_mask * dout[b, :]   (dsrc is bf16; rows are distinct)
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float* __restrict__ dout, int ld, const int32_t* __restrict__ rows,
                                                              bf16* __restrict__ dsrc, int B, int H, uint32_t thr16, float inv_keep,
                                                              uint32_t seed, uint32_t stream, int accumulate) {
  const size_t total = (size_t)B * H;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int col = (int)(t % H), b = (int)(t / H);
    float v = dout[t];
    if (thr16) {
      const uint32_t r = ia_rng(seed, stream, (uint32_t)(t >> 1));
      const uint32_t u = (t & 1) ? (r >> 16) : (r & 0xFFFFu);
      v = u >= thr16 ? v * inv_keep : 0.f;
    }
    bf16* p = dsrc + (size_t)rows[b] * ld + col;
    *p = f2bf(accumulate ? bf2f(*p) + v : v);
  }
}

int ln_blocks(int M) { int b = (M + 3) / 4; return b < 512 ? b : 512; }
void drop_params(float p, uint32_t& thr16, float& inv_keep) {
  thr16 = p > 0.f ? (uint32_t)(p * 65536.f + 0.5f) : 0u;
  inv_keep = p > 0.f ? 1.f / (1.f - (float)thr16 / 65536.f) : 1.f;
}
int grid_for(size_t work_items) { size_t g = (work_items + 255) / 256; return (int)(g < 4096 ? (g ? g : 1) : 4096); }

}  // namespace

extern "C" int ia_embed_ln_fwd(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const int32_t* extra_idx,
                               const float* word, const float* type, const float* pos, const float* extra, const float* gamma,
                               const float* beta, void* z_out, void* y, float* mean, float* rstd, int M, int H, float eps,
                               float drop_p, uint32_t seed, uint32_t stream_id, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!ids || !type_ids || !pos_ids || !word || !type || !pos || !gamma || !beta || !z_out || !y || !mean || !rstd) return IA_ERR_ARG;
  if (M <= 0 || (H & 7) || H > 4096 || (extra_idx && !extra)) return IA_ERR_ARG;
  uint32_t thr16; float inv_keep; drop_params(drop_p, thr16, inv_keep);
  dim3 grid((M + 3) / 4), blk(256);
  const int nv = (H + 511) / 512;
#define IA_E(NV) hipLaunchKernelGGL((embed_ln_fwd_kernel<NV>), grid, blk, 0, stream, ids, type_ids, pos_ids, extra_idx, word, type, pos, \
    extra, gamma, beta, (bf16*)z_out, (bf16*)y, mean, rstd, M, H, eps, thr16, inv_keep, seed, stream_id)
  switch (nv) { case 1: IA_E(1); break; case 2: IA_E(2); break; case 3: IA_E(3); break; case 4: IA_E(4); break; default: IA_E(8); }
#undef IA_E
  return ia_check_launch();
}

// grid of the backward kernel: x = blocks of 4 sequence positions, y = slices of the batch of sequences
static void embed_bwd_grid(int M, int L, int& gx, int& gy) {
  if (L <= 0 || M % L) L = M;
  const int S = M / L;
  gx = (L + 3) / 4; if (gx > 512) gx = 512;
  gy = 8192 / (gx * 4); if (gy < 1) gy = 1; if (gy > S) gy = S;      // up to 2048 workgroups = the rows of the partial-sum workspace
}

extern "C" size_t ia_embed_ln_bwd_workspace_bytes(int M, int H) { return (size_t)2048 * 2 * H * sizeof(float); }

// word_pad / pos_pad: rows of the word / position tables that receive no gradient (nn.Embedding
// padding_idx, reference base.py:213,234-236); pass -1 for none.  All gradients accumulate (+=).
extern "C" int ia_embed_ln_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                               const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const int32_t* extra_idx,
                               const int32_t* row_order, float* dword, float* dtype, float* dpos, float* dextra, float* dgamma, float* dbeta,
                               int M, int H,
                               int L, int word_pad, int pos_pad, float drop_p, uint32_t seed, uint32_t stream_id, void* workspace,
                               size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!dy || !z || !mean || !rstd || !gamma || !ids || !type_ids || !pos_ids) return IA_ERR_ARG;
  if (M <= 0 || (H & 7) || H > 4096) return IA_ERR_ARG;
  if (!workspace || workspace_bytes < ia_embed_ln_bwd_workspace_bytes(M, H)) return IA_ERR_WORKSPACE;
  uint32_t thr16; float inv_keep; drop_params(drop_p, thr16, inv_keep);
  if (L <= 0 || M % L) L = M;      // no sequence structure given: every row is its own position
  if (row_order) L = M < 2048 ? M : 2048;   // chunks of the position-sorted row list, one per wave
  int gx, gy; embed_bwd_grid(M, L, gx, gy);
  if (row_order) gy = 1;
  const int nb = gx * gy, nv = (H + 511) / 512;
  if ((size_t)nb * 2 * H * sizeof(float) > workspace_bytes) return IA_ERR_WORKSPACE;
  float* part = (float*)workspace;
  dim3 grid(gx, gy), blk(256);
#define IA_E(NV) hipLaunchKernelGGL((embed_ln_bwd_kernel<NV>), grid, blk, 0, stream, (const bf16*)dy, (const bf16*)z, mean, rstd, gamma, ids, \
    type_ids, pos_ids, extra_idx, row_order, dword, dtype, dpos, dextra, part, M, H, L, word_pad, pos_pad, thr16, inv_keep, seed, stream_id)
  switch (nv) { case 1: IA_E(1); break; case 2: IA_E(2); break; case 3: IA_E(3); break; case 4: IA_E(4); break; default: IA_E(8); }
#undef IA_E
  hipLaunchKernelGGL(reduce2_kernel, dim3((2 * H + 31) / 32), dim3(1024), 0, stream, part, nb, H, dgamma, dbeta);
  return ia_check_launch();
}

extern "C" int ia_im2col_patch(const float* images, void* patches, int B, int C, int S, int P, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!images || !patches || B <= 0 || C <= 0 || P <= 0 || (P & 7) || S % P) return IA_ERR_ARG;
  const size_t items = (size_t)B * (S / P) * (S / P) * C * P * P / 8;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(items)), dim3(256), 0, stream, images, (bf16*)patches, B, C, S, P);
  return ia_check_launch();
}

extern "C" int ia_vit_tokens_fwd(const void* patch, const float* cls, const float* pos, void* tokens, int B, int NP, int H,
                                 hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!patch || !cls || !pos || !tokens || B <= 0 || NP <= 0 || (H & 7)) return IA_ERR_ARG;
  hipLaunchKernelGGL(vit_tokens_fwd_kernel, dim3(grid_for((size_t)B * (NP + 1) * H / 8)), dim3(256), 0, stream, (const bf16*)patch, cls, pos,
                     (bf16*)tokens, B, NP, H);
  return ia_check_launch();
}

extern "C" int ia_vit_tokens_bwd(const void* dtokens, void* dpatch, float* dcls, float* dpos, int B, int NP, int H, int accumulate,
                                 hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!dtokens || !dpatch || !dcls || !dpos || B <= 0 || NP <= 0 || (H & 7)) return IA_ERR_ARG;
  hipLaunchKernelGGL(vit_tokens_bwd_kernel, dim3(grid_for((size_t)(NP + 1) * H / 8)), dim3(256), 0, stream, (const bf16*)dtokens,
                     (bf16*)dpatch, dcls, dpos, B, NP, H, accumulate);
  return ia_check_launch();
}

extern "C" int ia_gather_rows_fwd(const void* src, int ld, const int32_t* rows, float* out, int B, int H, float drop_p, uint32_t seed,
                                  uint32_t stream_id, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!src || !rows || !out || B <= 0 || H <= 0) return IA_ERR_ARG;
  uint32_t thr16; float inv_keep; drop_params(drop_p, thr16, inv_keep);
  hipLaunchKernelGGL(gather_rows_fwd_kernel, dim3(grid_for((size_t)B * H)), dim3(256), 0, stream, (const bf16*)src, ld, rows, out, B, H,
                     thr16, inv_keep, seed, stream_id);
  return ia_check_launch();
}

extern "C" int ia_gather_rows_bwd(const float* dout, int ld, const int32_t* rows, void* dsrc, int B, int H, float drop_p, uint32_t seed,
                                  uint32_t stream_id, int accumulate, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!dout || !rows || !dsrc || B <= 0 || H <= 0) return IA_ERR_ARG;
  uint32_t thr16; float inv_keep; drop_params(drop_p, thr16, inv_keep);
  hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(grid_for((size_t)B * H)), dim3(256), 0, stream, dout, ld, rows, (bf16*)dsrc, B, H,
                     thr16, inv_keep, seed, stream_id, accumulate);
  return ia_check_launch();
}
