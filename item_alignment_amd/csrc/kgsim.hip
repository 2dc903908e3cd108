// Small HBM-bound kernels either side of the encoders:
//  * the PKGM knowledge-graph rows (reference src/models/base.py:347-392 RobertaPKGMEmbeddings.kg_embeddings): gather one
//    entity and P relation embeddings per item, sign() of the entity (F.normalize over a size-1 dim, quirk A1), and the
//    triple-query (h + r) / relation-query (M h - r) rows that are spliced into the token sequence;
//  * the vector-similarity head (base.py:10-34 InnerProduct, :75-88 VecSimClassificationHead.forward): cosine / l1 / l2 /
//    inner product of two [B, D] feature matrices and the probability each maps to.
// One wave per output row, 16-byte accesses, fp32 throughout (these tensors are [B, <=61, 1024]).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ PKGM
// ids: [B, ld_ids] int64; entity id at column ent_col, P relation ids from column rel_lo.
// h_sign [B, Dk] = sign(ent_table[e]);  r [B*P, Dk] = rel_table[r_p]
__global__ __launch_bounds__(256) void kg_gather_fwd_kernel(const float* __restrict__ ent, const float* __restrict__ rel,
                                                            const int64_t* __restrict__ ids, int ld_ids, int ent_col, int rel_lo,
                                                            float* __restrict__ h_sign, float* __restrict__ r, int B, int P, int Dk) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * (P + 1)) return;
  const int b = row / (P + 1), j = row % (P + 1);          // j == 0: entity, else relation j-1
  if (j == 0) {
    const float* s = ent + (size_t)ids[(size_t)b * ld_ids + ent_col] * Dk;
    for (int c = lane * 4; c < Dk; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(s + c);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? 1.f : (v[k] < 0.f ? -1.f : 0.f);   // x / max(|x|, 1e-12), |x| >= 1e-12
      *reinterpret_cast<f32x4*>(h_sign + (size_t)b * Dk + c) = o;
    }
  } else {
    const float* s = rel + (size_t)ids[(size_t)b * ld_ids + rel_lo + j - 1] * Dk;
    float* d = r + ((size_t)b * P + j - 1) * Dk;
    for (int c = lane * 4; c < Dk; c += 256) *reinterpret_cast<f32x4*>(d + c) = *reinterpret_cast<const f32x4*>(s + c);
  }
}

// rel_grad[r_p] += dr[b, p]  (a relation may repeat inside a batch: fp32 atomics)
__global__ __launch_bounds__(256) void kg_gather_bwd_kernel(const float* __restrict__ dr, const int64_t* __restrict__ ids, int ld_ids,
                                                            int rel_lo, float* __restrict__ rel_grad, int B, int P, int Dk) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * P) return;
  const int b = row / P, j = row % P;
  float* g = rel_grad + (size_t)ids[(size_t)b * ld_ids + rel_lo + j] * Dk;
  const float* s = dr + (size_t)row * Dk;
  for (int c = lane; c < Dk; c += 64) atomicAdd(g + c, s[c]);
}

// rows[b, p] = h[b] + r[b, p] ; rows[b, P + p] = hp[b] - r[b, p]      (rows: [B, ld_rows/H rows, H], this side at row0)
__global__ __launch_bounds__(256) void kg_rows_fwd_kernel(const float* __restrict__ h, const float* __restrict__ r,
                                                          const float* __restrict__ hp, float* __restrict__ rows, int rows_per_item,
                                                          int row0, int B, int P, int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * P) return;
  const int b = row / P, p = row % P;
  float* o1 = rows + ((size_t)b * rows_per_item + row0 + p) * H;
  float* o2 = o1 + (size_t)P * H;
  for (int c = lane * 4; c < H; c += 256) {
    const f32x4 hv = *reinterpret_cast<const f32x4*>(h + (size_t)b * H + c);
    const f32x4 pv = *reinterpret_cast<const f32x4*>(hp + (size_t)b * H + c);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(r + (size_t)row * H + c);
    *reinterpret_cast<f32x4*>(o1 + c) = hv + rv;
    *reinterpret_cast<f32x4*>(o2 + c) = pv - rv;
  }
}

// dr[b, p] = g[b, p] - g[b, P + p] ; dh[b] = sum_p g[b, p] ; dhp[b] = sum_p g[b, P + p]   (one block per item)
__global__ __launch_bounds__(256) void kg_rows_bwd_kernel(const float* __restrict__ drows, int rows_per_item, int row0,
                                                          float* __restrict__ dh, float* __restrict__ dr, float* __restrict__ dhp, int P,
                                                          int H) {
  const int b = blockIdx.x;
  const float* g = drows + ((size_t)b * rows_per_item + row0) * H;
  for (int c = threadIdx.x; c < H; c += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < P; ++p) {
      const float a = g[(size_t)p * H + c], q = g[(size_t)(P + p) * H + c];
      s1 += a; s2 += q;
      dr[((size_t)b * P + p) * H + c] = a - q;
    }
    dh[(size_t)b * H + c] = s1;
    dhp[(size_t)b * H + c] = s2;
  }
}

// ----------------------------------------------------------------------------------- similarity head
enum { SIM_INNER = 0, SIM_COSINE = 1, SIM_L1 = 2, SIM_L2 = 3 };
constexpr float COS_EPS = 1e-8f;     // torch.nn.functional.cosine_similarity default
constexpr float DIST_EPS = 1e-6f;    // torch.nn.functional.pairwise_distance adds eps to the difference

__global__ __launch_bounds__(256) void pair_sim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           float* __restrict__ sim, float* __restrict__ probs, int B, int D, int measure) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  const float* a = x + (size_t)row * D;
  const float* c = y + (size_t)row * D;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float u = a[i], v = c[i];
    if (measure == SIM_INNER) s0 += u * v;
    else if (measure == SIM_COSINE) { s0 += u * v; s1 += u * u; s2 += v * v; }
    else { const float d = u - v + DIST_EPS; s0 += measure == SIM_L1 ? fabsf(d) : d * d; }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane) return;
  float s, p;
  if (measure == SIM_INNER) { s = s0; p = 1.f / (1.f + __expf(-s)); }
  else if (measure == SIM_COSINE) { s = s0 * rsqrtf(fmaxf(s1 * s2, COS_EPS * COS_EPS)); p = (s + 1.f) * 0.5f; }
  else { s = measure == SIM_L1 ? s0 : sqrtf(s0); p = __expf(-s); }
  sim[row] = s;
  probs[row] = p;
}

// g = dsim + dprobs * dprobs/dsim ; dx, dy = g * dsim/dx, g * dsim/dy
__global__ __launch_bounds__(256) void pair_sim_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ sim, const float* __restrict__ probs,
                                                           const float* __restrict__ dsim, const float* __restrict__ dprobs,
                                                           float* __restrict__ dx, float* __restrict__ dy, int B, int D, int measure) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  const float* a = x + (size_t)row * D;
  const float* c = y + (size_t)row * D;
  const float s = sim[row], p = probs[row];
  float g = dsim ? dsim[row] : 0.f;
  if (dprobs) g += dprobs[row] * (measure == SIM_INNER ? p * (1.f - p) : measure == SIM_COSINE ? 0.5f : -p);
  float n1 = 0.f, n2 = 0.f;
  if (measure == SIM_COSINE) {
    for (int i = lane; i < D; i += 64) { n1 += a[i] * a[i]; n2 += c[i] * c[i]; }
    n1 = wave_sum(n1); n2 = wave_sum(n2);
  }
  const float inv12 = measure == SIM_COSINE ? rsqrtf(fmaxf(n1 * n2, COS_EPS * COS_EPS)) : 0.f;
  for (int i = lane; i < D; i += 64) {
    const float u = a[i], v = c[i];
    float gu, gv;
    if (measure == SIM_INNER) { gu = v; gv = u; }
    else if (measure == SIM_COSINE) { gu = v * inv12 - s * u / fmaxf(n1, COS_EPS); gv = u * inv12 - s * v / fmaxf(n2, COS_EPS); }
    else {
      const float d = u - v + DIST_EPS;
      gu = measure == SIM_L1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : (s > 0.f ? d / s : 0.f);
      gv = -gu;
    }
    dx[(size_t)row * D + i] = g * gu;
    dy[(size_t)row * D + i] = g * gv;
  }
}

}  // namespace

extern "C" int ia_kg_gather_fwd(const float* ent_table, const float* rel_table, const int64_t* ids, int ld_ids, int ent_col, int rel_lo,
                                float* h_sign, float* r, int B, int P, int Dk, hipStream_t stream) {
  (void)hipGetLastError();
  if (!ent_table || !rel_table || !ids || !h_sign || !r || B <= 0 || P <= 0 || Dk <= 0 || (Dk & 3)) return IA_ERR_ARG;
  hipLaunchKernelGGL(kg_gather_fwd_kernel, dim3((B * (P + 1) + 3) / 4), dim3(256), 0, stream, ent_table, rel_table, ids, ld_ids, ent_col,
                     rel_lo, h_sign, r, B, P, Dk);
  return ia_check_launch();
}

// only the relation table receives a gradient: d sign(x) / dx = 0 (quirk A1)
extern "C" int ia_kg_gather_bwd(const float* dr, const int64_t* ids, int ld_ids, int rel_lo, float* rel_grad, int B, int P, int Dk,
                                hipStream_t stream) {
  (void)hipGetLastError();
  if (!dr || !ids || !rel_grad || B <= 0 || P <= 0 || Dk <= 0) return IA_ERR_ARG;
  hipLaunchKernelGGL(kg_gather_bwd_kernel, dim3((B * P + 3) / 4), dim3(256), 0, stream, dr, ids, ld_ids, rel_lo, rel_grad, B, P, Dk);
  return ia_check_launch();
}

extern "C" int ia_kg_rows_fwd(const float* h, const float* r, const float* hp, float* rows, int rows_per_item, int row0, int B, int P,
                              int H, hipStream_t stream) {
  (void)hipGetLastError();
  if (!h || !r || !hp || !rows || B <= 0 || P <= 0 || H <= 0 || (H & 3) || row0 < 0 || row0 + 2 * P > rows_per_item) return IA_ERR_ARG;
  hipLaunchKernelGGL(kg_rows_fwd_kernel, dim3((B * P + 3) / 4), dim3(256), 0, stream, h, r, hp, rows, rows_per_item, row0, B, P, H);
  return ia_check_launch();
}

extern "C" int ia_kg_rows_bwd(const float* drows, int rows_per_item, int row0, float* dh, float* dr, float* dhp, int B, int P, int H,
                              hipStream_t stream) {
  (void)hipGetLastError();
  if (!drows || !dh || !dr || !dhp || B <= 0 || P <= 0 || H <= 0 || row0 < 0 || row0 + 2 * P > rows_per_item) return IA_ERR_ARG;
  hipLaunchKernelGGL(kg_rows_bwd_kernel, dim3(B), dim3(256), 0, stream, drows, rows_per_item, row0, dh, dr, dhp, P, H);
  return ia_check_launch();
}

extern "C" int ia_pair_sim_fwd(const float* x, const float* y, float* sim, float* probs, int B, int D, int measure, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !y || !sim || !probs || B <= 0 || D <= 0 || measure < 0 || measure > 3) return IA_ERR_ARG;
  hipLaunchKernelGGL(pair_sim_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, x, y, sim, probs, B, D, measure);
  return ia_check_launch();
}

extern "C" int ia_pair_sim_bwd(const float* x, const float* y, const float* sim, const float* probs, const float* dsim, const float* dprobs,
                               float* dx, float* dy, int B, int D, int measure, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !y || !sim || !probs || !dx || !dy || B <= 0 || D <= 0 || measure < 0 || measure > 3) return IA_ERR_ARG;
  hipLaunchKernelGGL(pair_sim_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, x, y, sim, probs, dsim, dprobs, dx, dy, B, D, measure);
  return ia_check_launch();
}
