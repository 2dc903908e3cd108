// Pair heads and losses (tiny fp32 work on [B, H] CLS features), gfx950.
//
//   ia_linear_small_fwd/bwd : y = act(x W^T + b), act in {none, tanh}; fp32 master weights are read
//                             directly (reference base.py:139-157 RobertaClassificationHead dense+tanh,
//                             base.py:66-76 VecSim dense+tanh, base.py:530 img2txt)
//   ia_pair_head_ce_fwd/bwd : logits = [x | y] W^T + b, probs = softmax(logits), loss = mean CE
//                             (reference base.py:103-117 TwoTowerClassificationHead + nn.CrossEntropyLoss
//                              text.py:1292,1360; with y = null it is out_proj + softmax + CE of the
//                              one-tower head, base.py:155 + text.py:1463,1473)
// One wave per output element; the batch is tens of rows, so these kernels are latency- not
// bandwidth-bound and are kept simple and deterministic (no atomics).
#include "common.h"

namespace {

enum { ACT_NONE = 0, ACT_TANH = 1 };

// y[b,n] = act(sum_k x[b,k] W[n,k] + bias[n]); grid = ceil(B*N / 4) blocks of 4 waves
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ y, int B, int N,
                                                               int K, int act) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= B * N) return;
  const int b = o / N, n = o % N;
  const float* xr = x + (size_t)b * ldx;
  const float* wr = W + (size_t)n * K;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += xr[k] * wr[k];
  s = wave_sum(s);
  if (lane == 0) {
    if (bias) s += bias[n];
    y[(size_t)b * N + n] = act == ACT_TANH ? tanhf(s) : s;
  }
}

// dpre[b,n] = dy[b,n] * act'(y[b,n]);  dx[b,k] = sum_n dpre[b,n] W[n,k]
__global__ __launch_bounds__(256) void linear_small_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                  const float* __restrict__ W, float* __restrict__ dx, int lddx, int B,
                                                                  int N, int K, int act) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * K) return;
  const int b = t / K, k = t % K;
  float s = 0.f;
  for (int n = 0; n < N; ++n) {
    float g = dy[(size_t)b * N + n];
    if (act == ACT_TANH) { const float yy = y[(size_t)b * N + n]; g *= 1.f - yy * yy; }
    s += g * W[(size_t)n * K + k];
  }
  dx[(size_t)b * lddx + k] = s;
}

// dW[n,k] += sum_b dpre[b,n] x[b,k];  db[n] += sum_b dpre[b,n]
__global__ __launch_bounds__(256) void linear_small_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                  const float* __restrict__ x, int ldx, float* __restrict__ dW,
                                                                  float* __restrict__ db, int B, int N, int K, int act) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= N * K) return;
  const int n = t / K, k = t % K;
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    float g = dy[(size_t)b * N + n];
    if (act == ACT_TANH) { const float yy = y[(size_t)b * N + n]; g *= 1.f - yy * yy; }
    s += g * x[(size_t)b * ldx + k];
    sb += g;
  }
  dW[t] += s;
  if (db && k == 0) db[n] += sb;
}

// ---- the same three products for HUNDREDS of rows (the image-CLS projection of the CoCa model sees one row per image: 512 x 768 ->
// 1024 at the bench size, where the one-thread-per-output kernels above took 0.27 + 0.39 + 0.32 ms): fp32 tiles of 64 x 64 outputs,
// 16 x 16 threads with 4 x 4 outputs each, k in steps of 16 through LDS.  C[i][j] = sum_k A(i,k) B(k,j) with A / B given by row and
// column strides, so one kernel serves x W^T, dpre W and dpre^T x; dpre = dy * tanh'(y) is formed while loading.  Fixed summation
// order (deterministic).  EPI 0: C = act(acc + bias[j]);  1: C = acc;  2: C += acc and db[i] += sum_k A(i,k) (k = batch rows).
template <int EPI>
__global__ __launch_bounds__(256) void linear_tiled_kernel(const float* __restrict__ A, long a_rs, long a_cs, const float* __restrict__ Ay,
                                                           const float* __restrict__ Bm, long b_rs, long b_cs, const float* __restrict__ By,
                                                           float* __restrict__ Cm, long ldc, const float* __restrict__ bias, float* __restrict__ db,
                                                           int Mo, int No, int Kd, int act) {
  __shared__ float sa[16][64 + 1], sb[16][64 + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  float acc[4][4] = {};
  float rsum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < Kd; k0 += 16) {
    // 64 x 16 elements of each operand, 4 per thread; Ay / By (same indexing) carry the tanh outputs the derivative is taken from
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = threadIdx.x + e * 256;
      const int kk = idx & 15, r = idx >> 4;
      const int k = k0 + kk;
      float va = 0.f, vb = 0.f;
      if (k < Kd && i0 + r < Mo) {
        const long o = (long)(i0 + r) * a_rs + (long)k * a_cs;
        va = A[o];
        if (Ay) { const float yy = Ay[o]; va *= 1.f - yy * yy; }
      }
      if (k < Kd && j0 + r < No) {
        const long o = (long)k * b_rs + (long)(j0 + r) * b_cs;
        vb = Bm[o];
        if (By) { const float yy = By[o]; vb *= 1.f - yy * yy; }
      }
      sa[kk][r] = va; sb[kk][r] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = sa[kk][ty * 4 + u]; b[u] = sb[kk][tx * 4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (EPI == 2) rsum[u] += a[u];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + ty * 4 + u;
    if (i >= Mo) continue;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = j0 + tx * 4 + v;
      if (j >= No) continue;
      float c = acc[u][v];
      if (EPI == 0) { if (bias) c += bias[j]; if (act == ACT_TANH) c = tanhf(c); }
      float* dst = Cm + (long)i * ldc + j;
      *dst = EPI == 2 ? *dst + c : c;
    }
    if (EPI == 2 && db && blockIdx.x == 0 && tx == 0) db[i] += rsum[u];
  }
}

// one wave per sample: logits, probs, per-sample loss; then a single-wave mean
__global__ __launch_bounds__(64) void pair_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ yv,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           const int64_t* __restrict__ labels, float* __restrict__ logits,
                                                           float* __restrict__ probs, float* __restrict__ loss_per, int B, int D, int C) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int F = yv ? 2 * D : D;
  float lg[8];
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += x[(size_t)b * D + k] * W[(size_t)c * F + k];
    if (yv) for (int k = lane; k < D; k += 64) s += yv[(size_t)b * D + k] * W[(size_t)c * F + D + k];
    s = wave_sum(s);
    lg[c] = s + (bias ? bias[c] : 0.f);
  }
  if (lane == 0) {
    float mx = lg[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, lg[c]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(lg[c] - mx);
    for (int c = 0; c < C; ++c) { logits[(size_t)b * C + c] = lg[c]; probs[(size_t)b * C + c] = expf(lg[c] - mx) / den; }
    if (labels) loss_per[b] = -(lg[labels[b]] - mx - logf(den));
  }
}

__global__ __launch_bounds__(64) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += v[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

// dlogits[b,c] = dloss * (probs - onehot) / B ; dx = dlogits W[:, :D] ; dy = dlogits W[:, D:]
__global__ __launch_bounds__(256) void pair_head_bwd_dxy_kernel(const float* __restrict__ probs, const int64_t* __restrict__ labels,
                                                                const float* __restrict__ dloss, const float* __restrict__ W,
                                                                float* __restrict__ dx, float* __restrict__ dyv, int B, int D, int C,
                                                                int two) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int F = two ? 2 * D : D;
  if (t >= B * F) return;
  const int b = t / F, k = t % F;
  const float scale = dloss[0] / (float)B;
  float s = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = (probs[(size_t)b * C + c] - (labels[b] == c ? 1.f : 0.f)) * scale;
    s += g * W[(size_t)c * F + k];
  }
  if (k < D) dx[(size_t)b * D + k] = s; else dyv[(size_t)b * D + (k - D)] = s;
}

__global__ __launch_bounds__(256) void pair_head_bwd_dw_kernel(const float* __restrict__ probs, const int64_t* __restrict__ labels,
                                                               const float* __restrict__ dloss, const float* __restrict__ x,
                                                               const float* __restrict__ yv, float* __restrict__ dW, float* __restrict__ db,
                                                               int B, int D, int C, int two) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int F = two ? 2 * D : D;
  if (t >= C * F) return;
  const int c = t / F, k = t % F;
  const float scale = dloss[0] / (float)B;
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = (probs[(size_t)b * C + c] - (labels[b] == c ? 1.f : 0.f)) * scale;
    const float f = k < D ? x[(size_t)b * D + k] : yv[(size_t)b * D + (k - D)];
    s += g * f;
    sb += g;
  }
  dW[t] += s;
  if (db && k == 0) db[c] += sb;
}

// ---- span means of the auxiliary attribute-pair task (reference text.py:66-102 AuxiliaryTaskPair.forward)
// out[s][h] = mean over rows [spans[s][0], spans[s][1]) of seq[row][h]   (an empty span gives NaN, like torch's mean of nothing)
__global__ __launch_bounds__(256) void span_mean_fwd_kernel(const bf16* __restrict__ seq, int ld, const int* __restrict__ spans,
                                                            float* __restrict__ out, int H, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;       // over S * H/8
  if (idx >= total) return;
  const int h8n = H >> 3, h = (idx % h8n) * 8, sp = idx / h8n;
  const int r0 = spans[2 * sp], r1 = spans[2 * sp + 1];
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int r = r0; r < r1; ++r) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(seq + (size_t)r * ld + h);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
  }
  const float inv = 1.f / (float)(r1 - r0);
#pragma unroll
  for (int j = 0; j < 8; ++j) out[(size_t)sp * H + h + j] = acc[j] * inv;
}

// dseq[row][h] = sum over the spans of this row's sample that contain the row of dout[s][h] / len(s)   (gather form, fixed
// order -> deterministic; rows outside every span get zero).  span_ptr[b] .. span_ptr[b+1] = spans of sample b.
__global__ __launch_bounds__(256) void span_mean_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ spans,
                                                            const int* __restrict__ span_ptr, bf16* __restrict__ dseq, int L, int H, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;       // over B*L * H/8
  if (idx >= total) return;
  const int h8n = H >> 3, h = (idx % h8n) * 8, row = idx / h8n, b = row / L;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int sp = span_ptr[b]; sp < span_ptr[b + 1]; ++sp) {
    const int r0 = spans[2 * sp], r1 = spans[2 * sp + 1];
    if (row < r0 || row >= r1) continue;
    const float inv = 1.f / (float)(r1 - r0);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += dout[(size_t)sp * H + h + j] * inv;
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j]);
  *reinterpret_cast<bf16x8*>(dseq + (size_t)row * H + h) = o;
}

}  // namespace

extern "C" int ia_linear_small_fwd(const float* x, int ldx, const float* W, const float* bias, float* y, int B, int N, int K, int act,
                                   hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!x || !W || !y || B <= 0 || N <= 0 || K <= 0 || (act != ACT_NONE && act != ACT_TANH)) return IA_ERR_ARG;
  if (B >= 64)      // y[b][n] = sum_k x[b][k] W[n][k]
    hipLaunchKernelGGL(linear_tiled_kernel<0>, dim3((N + 63) / 64, (B + 63) / 64), dim3(256), 0, stream, x, (long)ldx, 1L, (const float*)nullptr, W, 1L,
                       (long)K, (const float*)nullptr, y, (long)N, bias, (float*)nullptr, B, N, K, act);
  else
    hipLaunchKernelGGL(linear_small_fwd_kernel, dim3((B * N + 3) / 4), dim3(256), 0, stream, x, ldx, W, bias, y, B, N, K, act);
  return ia_check_launch();
}

// dx may be null (input needs no gradient); dW / db accumulate (+=); y = forward output (for tanh').
extern "C" int ia_linear_small_bwd(const float* dy, const float* y, const float* x, int ldx, const float* W, float* dx, int lddx,
                                   float* dW, float* db, int B, int N, int K, int act, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!dy || !x || !W || B <= 0 || N <= 0 || K <= 0 || (act == ACT_TANH && !y)) return IA_ERR_ARG;
  const float* ty_ = act == ACT_TANH ? y : nullptr;
  if (B >= 64) {
    // dx[b][k] = sum_n dpre[b][n] W[n][k];  dW[n][k] += sum_b dpre[b][n] x[b][k], db[n] += sum_b dpre[b][n]
    if (dx) hipLaunchKernelGGL(linear_tiled_kernel<1>, dim3((K + 63) / 64, (B + 63) / 64), dim3(256), 0, stream, dy, (long)N, 1L, ty_, W, (long)K, 1L,
                               (const float*)nullptr, dx, (long)lddx, (const float*)nullptr, (float*)nullptr, B, K, N, act);
    if (dW) hipLaunchKernelGGL(linear_tiled_kernel<2>, dim3((K + 63) / 64, (N + 63) / 64), dim3(256), 0, stream, dy, 1L, (long)N, ty_, x, (long)ldx, 1L,
                               (const float*)nullptr, dW, (long)K, (const float*)nullptr, db, N, K, B, act);
    return ia_check_launch();
  }
  if (dx) hipLaunchKernelGGL(linear_small_bwd_dx_kernel, dim3((B * K + 255) / 256), dim3(256), 0, stream, dy, y, W, dx, lddx, B, N, K, act);
  if (dW) hipLaunchKernelGGL(linear_small_bwd_dw_kernel, dim3((N * K + 255) / 256), dim3(256), 0, stream, dy, y, x, ldx, dW, db, B, N, K, act);
  return ia_check_launch();
}

// x, y: [B, D] fp32 (y may be null: single-feature head); W: [C, 2D] (or [C, D]); labels may be null
// (then loss is not written); loss_per: scratch of B floats.
extern "C" int ia_pair_head_ce_fwd(const float* x, const float* y, const float* W, const float* bias, const int64_t* labels,
                                   float* logits, float* probs, float* loss, float* loss_per, int B, int D, int C, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!x || !W || !logits || !probs || B <= 0 || D <= 0 || C <= 0 || C > 8) return IA_ERR_ARG;
  if (labels && (!loss || !loss_per)) return IA_ERR_ARG;
  hipLaunchKernelGGL(pair_head_fwd_kernel, dim3(B), dim3(64), 0, stream, x, y, W, bias, labels, logits, probs, loss_per, B, D, C);
  if (labels) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, stream, loss_per, B, loss);
  return ia_check_launch();
}

// dloss: device scalar (upstream gradient of the mean loss). dW/db accumulate; dx/dy are overwritten.
extern "C" int ia_pair_head_ce_bwd(const float* probs, const int64_t* labels, const float* dloss, const float* x, const float* y,
                                   const float* W, float* dx, float* dy, float* dW, float* db, int B, int D, int C, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by unrelated runtime calls (e.g. hipEventQuery -> NotReady)
  if (!probs || !labels || !dloss || !x || !W || !dx || B <= 0 || D <= 0 || C <= 0 || C > 8) return IA_ERR_ARG;
  const int two = y != nullptr;
  if (two && !dy) return IA_ERR_ARG;
  const int F = two ? 2 * D : D;
  hipLaunchKernelGGL(pair_head_bwd_dxy_kernel, dim3((B * F + 255) / 256), dim3(256), 0, stream, probs, labels, dloss, W, dx, dy, B, D, C, two);
  if (dW) hipLaunchKernelGGL(pair_head_bwd_dw_kernel, dim3((C * F + 255) / 256), dim3(256), 0, stream, probs, labels, dloss, x, y, dW, db, B, D, C, two);
  return ia_check_launch();
}


// Span means over token rows (auxiliary attribute-pair task, reference text.py:79-86): seq [rows, ld] bf16, spans [S][2] int32
// = absolute (first row, end row) of each span, out [S][H] fp32.
extern "C" int ia_span_mean_fwd(const void* seq, int ld, const int* spans, float* out, int S, int H, hipStream_t stream) {
  (void)hipGetLastError();
  if (!seq || !spans || !out || S <= 0 || H <= 0 || (H & 7) || (ld & 7)) return IA_ERR_ARG;
  const int total = S * (H >> 3);
  hipLaunchKernelGGL(span_mean_fwd_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, (const bf16*)seq, ld, spans, out, H, total);
  return ia_check_launch();
}

// dseq [B*L, H] bf16 (overwritten) from dout [S][H] fp32; span_ptr [B+1] int32: spans of sample b are span_ptr[b]..span_ptr[b+1]
extern "C" int ia_span_mean_bwd(const float* dout, const int* spans, const int* span_ptr, void* dseq, int B, int L, int H, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dout || !spans || !span_ptr || !dseq || B <= 0 || L <= 0 || H <= 0 || (H & 7)) return IA_ERR_ARG;
  const int total = B * L * (H >> 3);
  hipLaunchKernelGGL(span_mean_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, dout, spans, span_ptr, (bf16*)dseq, L, H, total);
  return ia_check_launch();
}
