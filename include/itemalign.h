/* itemalign.h — C ABI of libitemalign_hip.so (MI355X / gfx950 kernels of the item-pair matching engine).
 *
 * The reference (sunzeyeah/item-alignment) has no native or FFI boundary: it is pure PyTorch and its
 * arithmetic is dispatched by ATen.  This header therefore *defines* the boundary:
 * each entry point replaces the ATen/cuBLAS work behind one reference call site, cited per function
 * (paths relative to the reference tree).  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *  - every function returns 0 on success, a negative IA_ERR_* code otherwise (ia_strerror for text);
 *    nothing throws across the boundary.
 *  - all pointers are borrowed DEVICE pointers (tensor.data_ptr()) of contiguous row-major tensors;
 *    nothing is allocated or freed inside; scratch comes in through (workspace, workspace_bytes) with
 *    an ia_*_workspace_bytes query.
 *  - launches are asynchronous on `stream` (pass torch's current HIP stream); no device sync inside.
 *  - "bf16" tensors are passed as void*; fp32 master weights / gradients as float*.
 *  - dropout uses a counter-based generator keyed by (seed, stream_id, element index): forward and
 *    backward of one op must be given the same (drop_p, seed, stream_id).
 */
#ifndef ITEMALIGN_H
#define ITEMALIGN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ia_stream_t; /* hipStream_t */

#define IA_OK 0
#define IA_ERR_ARG (-1)
#define IA_ERR_LAUNCH (-2)
#define IA_ERR_WORKSPACE (-3)
#define IA_ERR_UNSUPPORTED (-4)

const char* ia_strerror(int code);
/* Bumped whenever an entry point is added or the meaning of an argument / output changes (round 2 changed what IA_EPI_BIAS_GELU
 * stores in C2 and what IA_EPI_DGELU expects in aux): a caller built against another header must not run on this library.
 * item_alignment_amd/_lib.py refuses to load a library whose version differs from the one it was written for. */
#define IA_ABI_VERSION 10
int ia_abi_version(void);

/* ---- GEMM: torch.nn.Linear forward / dgrad / wgrad (src/models/text.py:1241 -> RobertaLayer dense
 * layers; src/models/multimodal.py:811 -> timm Block qkv/proj/fc1/fc2; patch-embed conv as GEMM).
 * C[M,N] = A*B with A k-contiguous ([M,K], lda) or k-strided ([K,M], lda); B k-contiguous ([N,K], ldb;
 * a Linear weight) or k-strided ([K,N], ldb).  epilogue: */
#define IA_EPI_NONE 0
#define IA_EPI_BIAS 1      /* + bias[N] (fp32) */
#define IA_EPI_BIAS_GELU 2 /* x = bf16(acc + bias): C = gelu_erf(x), C2 = gelu_erf'(x) (saved for IA_EPI_DGELU) */
#define IA_EPI_ADD 3       /* + aux[M,N] (bf16, ldaux) */
#define IA_EPI_DGELU 4     /* * aux[M,N], the derivative IA_EPI_BIAS_GELU saved in C2 */
#define IA_EPI_BIAS_ADD 5  /* + bias + aux */
#define IA_EPI_DGELU_COLSUM 6 /* IA_EPI_DGELU, and C2 (fp32 [N]) += column sums of C: the bias gradient of the Linear in front of the GELU
                               * (row-major C with ldc == N; workspace >= ia_gemm_colsum_workspace_bytes(M, N)) */
#define IA_EPI_BIAS_GELU_ACT 7 /* forward-only IA_EPI_BIAS_GELU: C = gelu_erf(bf16(acc + bias)), nothing else is evaluated or stored (ABI 5) */
int ia_gemm_bf16(const void* A, int a_kstrided, int lda, const void* B, int b_kstrided, int ldb, void* C, int c_is_f32, int ldc,
                 int M, int N, int K, int epilogue, const float* bias, const void* aux, int ldaux, void* C2, int accumulate,
                 void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* IA_EPI_BIAS GEMM with k-contiguous operands and bf16 output whose first scaled_cols columns (a multiple of 128) leave as
 * (acc + bias) * col_scale: the fused QKV projection of an encoder layer (RobertaSelfAttention.query/key/value, src/models/text.py:1241;
 * timm Attention.qkv) handing q * softmax scale * log2(e) to the ia_attn_*_ps kernels below -- the division by sqrt(d) of the reference
 * (attention_scores / math.sqrt(head size)) moved in front of the one bf16 rounding of q (ABI 6) */
int ia_gemm_bf16_qscale(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                        int scaled_cols, float col_scale, ia_stream_t stream);
/* fp32-output (weight-gradient) GEMMs cut K across workgroups when given this much scratch; partial sums are
 * combined in a fixed order (deterministic).  workspace may be NULL (no split).
 * Weight-gradient form (A and B k-strided, fp32 C, IA_EPI_NONE): a non-NULL C2 (fp32 [M]) += sum_k A[k][m], i.e. the bias
 * gradient of the layer whose dW this GEMM computes, out of the same pass over dy (ScaledStdConv2d bias, timm std_conv.py). */
size_t ia_gemm_workspace_bytes(int M, int N, int K, int c_is_f32);
size_t ia_gemm_colsum_workspace_bytes(int M, int N);

/* per-launch HIP-event timing of one GEMM instantiation (variant = a_kstrided*1000 + b_kstrided*100 + epilogue*10 + c_is_f32),
 * recorded on the launch stream; used by bench.py for the roofline of the dominant kernel. */
int ia_prof_begin(int variant, int max_launches);
int ia_prof_end(double* total_ms, double* total_flops, int* launches);
double ia_prof_bytes(void); /* algorithmic bytes of the recorded launches: A and B read once, C written once */
/* Diagnostics (ABI 7): occupy `workgroups` whole CUs (one 256-thread workgroup holding all 160 KiB of LDS each, so nothing else can
 * co-reside) for `milliseconds` on `stream` -- a stand-in for a communication kernel that holds CUs next to the persistent GEMMs
 * (tools/cu_contention.py: how much a GEMM on another stream slows down with 0 / 8 / 16 / 32 CUs taken, SURVEY 8(e) overlap of the
 * gradient all-reduce with backward).  With the dynamic tile claim on (ia_debug_gemm_dynamic below) the large persistent GEMM
 * launches take their tiles from one counter per XCD, so a workgroup that starts late finds no work instead of holding a 1/256
 * share back. */
int ia_debug_cu_hog(int workgroups, float milliseconds, ia_stream_t stream);
/* Run-time switch of the dynamic tile claim; returns the previous setting (on < 0: query only).  Default off (static order) unless
 * IA_GEMM_DYNAMIC=1: a host that overlaps a communication stream with the GEMMs (world size > 1) switches it on -- 0-1.5 % per large
 * GEMM for not paying ~1.3-1.5 x when CUs are taken (profiles/r05_cu_contention.txt). */
int ia_debug_gemm_dynamic(int on);

/* ---- LayerNorm tails (RobertaSelfOutput / RobertaOutput: dense -> dropout -> +residual -> LayerNorm;
 * timm Block norm1/norm2).  z = residual + dropout(x + bias); y = LN(z).  z_out may alias x. */
int ia_ln_fwd(const void* x, const float* bias, const void* residual, void* z_out, void* y, float* mean, float* rstd,
              const float* gamma, const float* beta, int M, int H, float eps, float drop_p, uint32_t seed, uint32_t stream_id,
              ia_stream_t stream);
size_t ia_ln_bwd_workspace_bytes(int M, int H);
/* dz = LN'(dy) + dres; dx = dropout-masked dz (only when drop_p > 0); dgamma/dbeta/dbias (+)= column sums */
int ia_ln_bwd(const void* dy, const void* dres, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz,
              void* dx, float* dgamma, float* dbeta, float* dbias, int M, int H, float drop_p, uint32_t seed, uint32_t stream_id,
              void* workspace, size_t workspace_bytes, int accumulate, ia_stream_t stream);
/* the same with a second upstream gradient: LayerNorm output gradient = dy + dy2 (dy2 may be NULL; dz may alias dy2).  Replaces the
 * "+ residual gradient" epilogue of the GEMM producing dy in a post-LN layer (transformers RobertaOutput / RobertaSelfOutput backward). */
int ia_ln_bwd2(const void* dy, const void* dy2, const void* dres, const void* z, const float* mean, const float* rstd, const float* gamma,
               void* dz, void* dx, float* dgamma, float* dbeta, float* dbias, int M, int H, float drop_p, uint32_t seed, uint32_t stream_id,
               void* workspace, size_t workspace_bytes, int accumulate, ia_stream_t stream);
/* (round 6, ABI 8) ia_ln_bwd2 with a row filter: row_live [M] uint8 or NULL; row_live[m] == 0 = the caller guarantees that dy, dy2 and dres
 * are zero in row m (a masked position of an encoder whose heads read no masked position, ia_layer_cfg::masked_rows_dead): the row's inputs
 * are not fetched, its dz / dx rows are written as zeros.  Identical results on such inputs; NULL = ia_ln_bwd2. */
int ia_ln_bwd2_rows(const void* dy, const void* dy2, const void* dres, const void* z, const float* mean, const float* rstd,
                    const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias, int M, int H, float drop_p,
                    uint32_t seed, uint32_t stream_id, const uint8_t* row_live, void* workspace, size_t workspace_bytes, int accumulate,
                    ia_stream_t stream);
size_t ia_colsum_workspace_bytes(int M, int N);
int ia_colsum(const void* x, int ld, int M, int N, float* out, int accumulate, void* workspace, size_t workspace_bytes,
              ia_stream_t stream);

/* ---- fused self-attention, head dim 64 (transformers RobertaSelfAttention eager path: QK^T/sqrt(d)
 * + additive key mask -> softmax -> dropout -> PV; timm Attention without mask).  q/k/v point at the
 * first column of head 0 of each operand and share row stride ld_qkv (elements); key_mask is [B,L]
 * uint8 (1 = attend) or NULL; lse2 is [B,nh,L] fp32 (log2-domain log-sum-exp, saved for backward).
 * `delta` of every backward entry point is caller-provided SCRATCH of B*nh*L (general form: B*nh*Lq) floats whose contents after
 * the call are unspecified: the two-kernel path leaves rowsum(dO o O) there, the single-kernel path (32 < L <= 256, Lq == Lk) keeps
 * its attendable-key bit map in it.  Hosts must not read it. */
int ia_attn_fwd(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, void* out, int ld_o, float* lse2,
                int B, int nh, int L, float scale, float drop_p, uint32_t seed, ia_stream_t stream);
int ia_attn_bwd(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out, const void* d_out,
                int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv, int B, int nh, int L,
                float scale, float drop_p, uint32_t seed, ia_stream_t stream);
/* ia_attn_bwd that also returns the bias gradient of the fused QKV projection (reference: the autograd of nn.Linear's bias behind
 * RobertaSelfAttention.query/key/value, src/models/text.py:1241): dbias[3*nh*64] fp32 (q | k | v) += column sums of dq, dk, dv over
 * all tokens, taken in the kernels' epilogues (deterministic two-stage sum); workspace: ia_attn_bwd_bias_workspace_bytes(B, nh, L). */
size_t ia_attn_bwd_bias_workspace_bytes(int B, int nh, int L);
int ia_attn_bwd_bias(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out, const void* d_out,
                     int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv, float* dbias, void* workspace,
                     size_t workspace_bytes, int B, int nh, int L, float scale, float drop_p, uint32_t seed, ia_stream_t stream);

/* The same three on a projection whose q columns were written by ia_gemm_bf16_qscale (q * scale * log2 e, bf16): no kernel scales q again,
 * dq is still the gradient of the unscaled q (what the projection's dgrad / wgrad consume).  `scale` keeps its meaning. (ABI 6) */
int ia_attn_fwd_ps(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, void* out, int ld_o, float* lse2,
                   int B, int nh, int L, float scale, float drop_p, uint32_t seed, ia_stream_t stream);
int ia_attn_bwd_bias_ps(const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out, const void* d_out,
                        int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv, float* dbias, void* workspace,
                        size_t workspace_bytes, int B, int nh, int L, float scale, float drop_p, uint32_t seed, ia_stream_t stream);
/* (round 6, ABI 8) the same with flags.  IA_ATTN_MASKED_ROWS_DEAD: the caller guarantees d_out == 0 at every masked position (an
 * encoder whose masked positions never reach the loss -- reference src/models/text.py:1241: RobertaEncoder under an attention mask,
 * heads reading [CLS] / valid spans): the one-kernel backward (L <= 256) skips 32-query blocks that hold only masked positions;
 * dq / dk / dv / dbias are identical to the unflagged call on such inputs. */
#define IA_ATTN_Q_PRESCALED 1
#define IA_ATTN_MASKED_ROWS_DEAD 2
int ia_attn_bwd_bias_ex(int flags, const void* q, const void* k, const void* v, int ld_qkv, const uint8_t* key_mask, const void* out,
                        const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv, float* dbias,
                        void* workspace, size_t workspace_bytes, int B, int nh, int L, float scale, float drop_p, uint32_t seed,
                        ia_stream_t stream);

/* General form (cross-attention and multi-query attention of the CoCa multimodal layers, src/models/multimodal.py:590-616
 * ParallelTransformerBlock and :665-706 CrossAttention): Lq queries attend to Lk keys per (sequence, head).  q / out /
 * d_out rows are b*Lq + i (strides ld_q, ld_o), k / v rows are b*Lk + j (stride ld_kv), head h at column h*64.  One
 * K/V head shared by all query heads = the nh = 1 case with the query heads folded into rows (q viewed as
 * [B, n*heads, 64], ld_q = 64, Lq = n*heads).  key_mask is [B, Lk] or NULL; lse2 is [B, nh, Lq], delta scratch of the same size
 * (contents unspecified on return, see above). */
int ia_attn_fwd_x(const void* q, int ld_q, const void* k, const void* v, int ld_kv, const uint8_t* key_mask, void* out, int ld_o,
                  float* lse2, int B, int nh, int Lq, int Lk, float scale, float drop_p, uint32_t seed, ia_stream_t stream);
int ia_attn_bwd_x(const void* q, int ld_q, const void* k, const void* v, int ld_kv, const uint8_t* key_mask, const void* out,
                  const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, int ld_dq, void* dk, void* dv, int ld_dkv,
                  int B, int nh, int Lq, int Lk, float scale, float drop_p, uint32_t seed, ia_stream_t stream);

/* ---- CoCa multimodal-layer element-wise ops (src/models/multimodal.py:495-524).
 * rotary_split: src rows [M, ld_src] hold q (nh heads x 64) | k (64) | v (64) from column 0 (the head of the fused
 * projection, :586); position = row % n.  q_out [M, nh*64] = rotary(q); kv_out [M, 128] = rotary(k) | v.  The backward
 * call writes (R^T dq | R^T dk | dv) into dsrc[:, 0 : nh*64 + 128].
 * swiglu: src points at 2F columns (x | gate); out [M, F] = silu(gate) * x. */
int ia_rotary_split_fwd(const void* src, int ld_src, void* q_out, void* kv_out, int M, int n, int nh, ia_stream_t stream);
int ia_rotary_split_bwd(const void* dq, const void* dkv, void* dsrc, int ld_src, int M, int n, int nh, ia_stream_t stream);
int ia_swiglu_fwd(const void* src, int ld_src, void* out, int M, int F, ia_stream_t stream);
int ia_swiglu_bwd(const void* dout, const void* src, int ld_src, void* dsrc, int ld_dsrc, int M, int F, ia_stream_t stream);

/* ---- PKGM knowledge-graph rows (src/models/base.py:347-392 RobertaPKGMEmbeddings.kg_embeddings).  ids: [B, ld_ids] int64
 * with the entity id at column ent_col and P relation ids from column rel_lo.  gather: h_sign [B, Dk] = sign(ent[e])
 * (F.normalize over a size-1 dim), r [B*P, Dk] = rel[r_p]; its backward scatters dr into the relation table's gradient
 * (fp32 atomics; the entity table gets none, d sign = 0).  rows: rows[b, row0 + p] = h[b] + r[b, p],
 * rows[b, row0 + P + p] = hp[b] - r[b, p] inside a [B, rows_per_item, H] buffer (hp = proj_mat(h), computed by the
 * caller with ia_linear_small_fwd). */
int ia_kg_gather_fwd(const float* ent_table, const float* rel_table, const int64_t* ids, int ld_ids, int ent_col, int rel_lo,
                     float* h_sign, float* r, int B, int P, int Dk, ia_stream_t stream);
int ia_kg_gather_bwd(const float* dr, const int64_t* ids, int ld_ids, int rel_lo, float* rel_grad, int B, int P, int Dk,
                     ia_stream_t stream);
int ia_kg_rows_fwd(const float* h, const float* r, const float* hp, float* rows, int rows_per_item, int row0, int B, int P, int H,
                   ia_stream_t stream);
int ia_kg_rows_bwd(const float* drows, int rows_per_item, int row0, float* dh, float* dr, float* dhp, int B, int P, int H,
                   ia_stream_t stream);

/* ---- vector-similarity head (src/models/base.py:10-34 InnerProduct, :75-88 VecSimClassificationHead.forward) on fp32
 * [B, D] features: sim [B] and the probability it maps to.  measure: */
#define IA_SIM_INNER 0   /* sum(x*y); probs = sigmoid(sim) */
#define IA_SIM_COSINE 1  /* F.cosine_similarity (eps 1e-8); probs = (sim + 1) / 2 */
#define IA_SIM_L1 2      /* F.pairwise_distance p=1 (eps 1e-6 added to the difference); probs = exp(-sim) */
#define IA_SIM_L2 3      /* F.pairwise_distance p=2; probs = exp(-sim) */
int ia_pair_sim_fwd(const float* x, const float* y, float* sim, float* probs, int B, int D, int measure, ia_stream_t stream);
/* dsim / dprobs: upstream gradients of the two outputs (either may be NULL) */
int ia_pair_sim_bwd(const float* x, const float* y, const float* sim, const float* probs, const float* dsim, const float* dprobs,
                    float* dx, float* dy, int B, int D, int measure, ia_stream_t stream);

/* ---- NHWC convolution tower (ECA-NFNet, src/models/image.py:191-199,253-257 -> timm NormFreeNet / NormFreeBlock /
 * ScaledStdConv2d / EcaModule / DownsampleAvg).  Activations are [B, H, W, C] bf16 rows; a standardised weight is
 * what[Cout][k*k*(C/groups)] bf16 with the tap index outermost inside a row (tap = ky*3 + kx).  k = 1 (stride 1,
 * groups 1) is a plain GEMM; k = 3 (stride 1 or 2, padding 1) gathers patches into the workspace and runs one GEMM
 * per group.  C/groups and Cout/groups must be multiples of 8 (the 3-channel stem input is zero-padded to 8). */
int ia_nchw_to_nhwc_bf16(const float* in, void* out, int B, int C, int H, int W, int Cp, ia_stream_t stream);
size_t ia_conv_nhwc_workspace_bytes(int B, int H, int W, int C, int Cout, int k, int stride, int groups);
int ia_conv_nhwc_fwd(const void* x, const void* what, const float* bias, void* y, int B, int H, int W, int C, int Cout, int k, int stride,
                     int groups, void* workspace, size_t workspace_bytes, ia_stream_t stream);
int ia_conv_nhwc_bwd_data(const void* dy, const void* what, void* dx, int B, int H, int W, int C, int Cout, int k, int stride, int groups,
                          void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* dwhat [Cout][k*k*C/groups] fp32 is overwritten; dbias [Cout] (may be NULL) is accumulated.  cols_valid != 0: the workspace
 * is the one ia_conv_nhwc_fwd used for the same x / geometry and still holds its patch matrix (skips the gather). */
int ia_conv_nhwc_bwd_weight(const void* x, const void* dy, float* dwhat, float* dbias, int B, int H, int W, int C, int Cout, int k,
                            int stride, int groups, int cols_valid, void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* ScaledStdConv2d weight: what[o][t*Cgp + c] = (w[o][c][t] - mean_o) * rstd_o * gain[o] * scale (statistics over the
 * Cg*kk fan-in, biased variance, eps inside the sqrt; channels Cg..Cgp-1 zero).  bwd accumulates into dw / dgain. */
int ia_ws_conv_weight_fwd(const float* w, const float* gain, void* what, float* mean, float* rstd, int Cout, int Cg, int kk, int Cgp,
                          float scale, float eps, ia_stream_t stream);
int ia_ws_conv_weight_bwd(const float* dwhat, const float* w, const float* gain, const float* mean, const float* rstd, float* dw,
                          float* dgain, int Cout, int Cg, int kk, int Cgp, float scale, ia_stream_t stream);
/* y = silu(x) * scale ; dx = dy * scale * silu'(x) (+ dadd) over n bf16 elements */
int ia_silu_fwd(const void* x, void* y, size_t n, float scale, ia_stream_t stream);
int ia_silu_bwd(const void* dy, const void* x, const void* dadd, void* dx, size_t n, float scale, ia_stream_t stream);
/* the same with a second consumer of y: dx = (dy + dy2) * scale * silu'(x) [+ dadd] -- the SiLU output of a downsampling NormFreeBlock
   feeds conv1 and the projected shortcut (timm nfnet.py NormFreeBlock.forward); replaces autograd's separate gradient add */
int ia_silu_bwd_sum(const void* dy, const void* dy2, const void* x, const void* dadd, void* dx, size_t n, float scale, ia_stream_t stream);
/* AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False) on NHWC */
int ia_avgpool2_fwd(const void* x, void* y, int B, int H, int W, int C, ia_stream_t stream);
int ia_avgpool2_bwd(const void* dy, void* dx, int B, int H, int W, int C, ia_stream_t stream);
/* global average pool [B, HW, C] bf16 -> [B, C] fp32 */
size_t ia_gap_workspace_bytes(int B, int HW, int C);
int ia_gap_fwd(const void* x, float* pooled, int B, int HW, int C, void* workspace, size_t workspace_bytes, ia_stream_t stream);
int ia_gap_bwd(const float* dpooled, void* dx, int B, int HW, int C, ia_stream_t stream);
/* block tail: out = x * sigmoid(conv1d_k(mean_HW x))[b, c] * coef + shortcut (coef = attn_gain * alpha); pooled / gate
 * [B, C] fp32 are saved for ia_eca_bwd, which returns dx (the shortcut's gradient is dout) and accumulates dconv_w [k] */
int ia_eca_fwd(const void* x, const float* conv_w, int k, const void* shortcut, void* out, float* pooled, float* gate, int B, int HW, int C,
               float coef, void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* the same tail with pooled = (mean_HW a) what^T + bias, a [B*HW, Cmid] = the input of the 1x1 convolution (what [C][Cmid] bf16, bias
 * [C] fp32 or NULL) whose output is x: the mean over pixels commutes with the per-pixel linear map, so the reduction reads the narrow
 * tensor (reference src/models/image.py:253-257 -> timm NormFreeBlock.conv3 + attn_last; round 6, ABI 8).  act_out (bf16, same shape as
 * out; may be NULL) = silu(out) * act_scale: the activation the next NormFreeBlock opens with (act1(x) * beta), written by the same
 * pass, bit-identical to ia_silu_fwd(out).  Backward: ia_eca_bwd (+ ia_silu_bwd for the gradient that arrives through act_out). */
size_t ia_eca_fwd_linear_workspace_bytes(int B, int HW, int Cmid);
int ia_eca_fwd_linear(const void* x, const void* a, const void* what, const float* bias, int Cmid, const float* conv_w, int k,
                      const void* shortcut, void* out, void* act_out, float act_scale, float* pooled, float* gate, int B, int HW, int C,
                      float coef, void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* backward of a tail that wrote act_out: dtot = (dact [+ dact2]) * act_scale * silu'(out) [+ dout_direct] (= ia_silu_bwd / ia_silu_bwd_sum,
 * bit for bit) is the whole gradient of out -- returned in dtot, it is also the shortcut's gradient -- and the gate gradient's spatial
 * sums of dtot * x come out of the same pass; dx, dconv_w as ia_eca_bwd.  dact2 / dout_direct may be NULL (round 6, ABI 8). */
int ia_eca_silu_bwd(const void* dact, const void* dact2, const void* out, const void* dout_direct, float act_scale, const void* x,
                    const float* conv_w, int k, const float* pooled, const float* gate, void* dtot, void* dx, float* dconv_w, int B, int HW,
                    int C, float coef, void* workspace, size_t workspace_bytes, ia_stream_t stream);
size_t ia_eca_bwd_workspace_bytes(int B, int HW, int C);
int ia_eca_bwd(const void* dout, const void* x, const float* conv_w, int k, const float* pooled, const float* gate, void* dx,
               float* dconv_w, int B, int HW, int C, float coef, void* workspace, size_t workspace_bytes, ia_stream_t stream);

/* ---- 3x3 stride-1 (grouped) convolution without a patch matrix.  Tensors live on the "padded domain": xp / dxp
 * [B, H+2, W+2, Cin], yp / dyp [B, H+2, W+2, Cout] bf16; inputs (xp, dyp) need a ZERO border, outputs carry garbage in their
 * border rows.  Cin/groups and Cout/groups must be powers of two >= 8.  what [Cout][9 * Cin/groups] (tap-major, as
 * ia_ws_conv_weight_fwd writes it).  Tap t is read as the tensor shifted by (t/3-1)(W+2) + (t%3-1) rows inside the GEMM. */
int ia_conv3x3_padded_fwd(const void* xp, const void* what, const float* bias, void* yp, int B, int H, int W, int Cin, int Cout, int groups,
                          ia_stream_t stream);
int ia_conv3x3_padded_bwd_data(const void* dyp, const void* what, void* dxp, int B, int H, int W, int Cin, int Cout, int groups,
                               ia_stream_t stream);
/* (ABI 7) Direct form for groups of 64 -> 64 channels (every 3x3 convolution of the NF-Net stages) and the stem's 16 -> 32 / 32 -> 64:
 * ia_conv3x3_padded_fwd takes it by itself -- persistent workgroups keep the group's filter bank in LDS and stream 8 x 30-pixel tiles
 * (input with halo staged once by LDS-DMA, the nine taps as LDS row offsets) instead of nine shifted reads per tile through the
 * L2 -> LDS path.  The data gradient is the same kernel on dy with the tap-flipped, transposed bank: ia_conv3x3_flip_weights(what ->
 * what_t [Cin][9 * Cout / groups], the same number of elements), then ia_conv3x3_padded_bwd_data_t.  ia_conv3x3_direct_supported:
 * 1 when forward AND data gradient of the shape take that path (IA_CONV_DIRECT=0 switches it off).  ia_conv3x3_padded_bwd_weight
 * takes its direct form by itself for the same channel pairs when the launch covers >= 10^5 pixels (transpose reads of the dy and x
 * tiles, one fp32 bank per workgroup, fixed-order fold: bit-reproducible; IA_CONV_DIRECT_WGRAD=0 switches it off);
 * ia_conv3x3_padded_workspace_bytes covers both forms. */
int ia_conv3x3_direct_supported(int Cin, int Cout, int groups);
int ia_conv3x3_flip_weights(const void* what, void* what_t, int Cin, int Cout, int groups, ia_stream_t stream);
int ia_conv3x3_padded_bwd_data_t(const void* dyp, const void* what_t, void* dxp, int B, int H, int W, int Cin, int Cout, int groups,
                                 ia_stream_t stream);
size_t ia_conv3x3_padded_workspace_bytes(int B, int H, int W, int Cin, int Cout, int groups);
int ia_conv3x3_padded_bwd_weight(const void* xp, const void* dyp, float* dwhat, float* dbias, int B, int H, int W, int Cin, int Cout,
                                 int groups, void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* 3x3 / stride 2 / padding 1 between bordered layouts: xp [B, H+2, W+2, Cin] (zero border) -> yp [B, Ho+2, Wo+2, Cout], Ho = (H-1)/2 + 1
 * (border of yp not written).  The strided convolutions of timm's NormFreeNet (nfnet.py: conv2 of the first block of stages 2-4,
 * 64 channels per group; stem conv4, 64 -> 128) behind reference src/models/image.py:254-257, without a patch matrix: forward = one
 * kernel over the four parity views of x, weight gradient = the direct weight-gradient kernel on those views, data gradient = GEMM +
 * gather between the bordered layouts.  ia_conv3x3_s2_supported: 1 for Cin = Cout = 64 * groups, or groups = 1, Cin = 64, Cout = 64 n
 * (0 also when IA_CONV_S2_DIRECT=0): other shapes go through ia_conv_nhwc_*.  One workspace size serves both gradient calls. */
int ia_conv3x3_s2_supported(int Cin, int Cout, int groups);
size_t ia_conv3x3_s2_padded_workspace_bytes(int B, int H, int W, int Cin, int Cout, int groups);
/* y_compact != 0: yp / dyp are [B, Ho, Wo, Cout] without a border (the stem's last convolution feeds compact consumers) */
int ia_conv3x3_s2_padded_fwd(const void* xp, const void* what, const float* bias, void* yp, int B, int H, int W, int Cin, int Cout,
                             int groups, int y_compact, ia_stream_t stream);
int ia_conv3x3_s2_padded_bwd_data(const void* dyp, const void* what, void* dxp, int B, int H, int W, int Cin, int Cout, int groups,
                                  int y_compact, void* workspace, size_t workspace_bytes, ia_stream_t stream);
int ia_conv3x3_s2_padded_bwd_weight(const void* xp, const void* dyp, float* dwhat, float* dbias, int B, int H, int W, int Cin, int Cout,
                                    int groups, int y_compact, void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* the data gradient as one kernel over the four parity classes of dx (ia_conv3x3_s2_dgrad_supported): what_t from
 * ia_conv3x3_flip_weights, dyp bordered with a ZERO border (or compact: y_compact), dxp bordered (interior written).  Cin = Cout = 64 * groups;
 * groups = 1, Cin = 64, Cout = 64 n runs as n launches over the 64-channel slices of dy that add up in dx, what_t then holds n banks
 * (ia_conv3x3_flip_weights(what, what_t, Cout, Cout, n)) */
int ia_conv3x3_s2_dgrad_supported(int Cin, int Cout, int groups);
int ia_conv3x3_s2_padded_bwd_data_t(const void* dyp, const void* what_t, void* dxp, int B, int H, int W, int Cin, int Cout, int groups,
                                    int y_compact, ia_stream_t stream);
/* y = silu(x) * scale between the compact [B,H,W,C] and the zero-bordered [B,H+2,W+2,C] layouts (one flag per side); the
 * backward call produces dx in x's layout from dy in y's layout */
int ia_silu_pad_fwd(const void* x, void* y, int B, int H, int W, int C, float scale, int in_padded, int out_padded, ia_stream_t stream);
int ia_pad_rows(const void* x, void* y, int B, int H, int W, int C, int in_padded, int out_padded, ia_stream_t stream);   /* plain copy */
int ia_silu_pad_bwd(const void* dy, const void* x, void* dx, int B, int H, int W, int C, float scale, int in_padded, int out_padded,
                    ia_stream_t stream);

/* ---- pre-activation ResNetV2 tower pieces (reference src/models/image.py:298-378 ResNetTwoTower -> timm 0.6.5 resnetv2.py
 * with norm_layer=BatchNormAct2d; README.md:187-197 `--model_name resnetv2_50`).  NHWC rows [B*H*W, C] bf16 as above. */
/* BatchNorm2d + ReLU.  The rows are `segments` equal consecutive runs, each normalised with its own batch statistics (the
 * reference runs the two towers as two forward calls: image.py:337-341).  training: batch statistics, running_mean/var [C]
 * (NULL allowed) updated with `momentum` segment by segment; else the running statistics are used.  mean / rstd
 * [segments][C] fp32 are saved for the backward call. */
size_t ia_bn_act_workspace_bytes(int rows, int C, int segments);
int ia_bn_act_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y, float* mean,
                  float* rstd, int rows, int C, int segments, float eps, float momentum, int training, int relu, void* workspace,
                  size_t workspace_bytes, ia_stream_t stream);
/* dx (+ extra [rows, C] bf16 when not NULL: a second gradient reaching x), dgamma / dbeta [C] accumulated (NULL allowed) */
int ia_bn_act_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                  const void* extra, void* dx, float* dgamma, float* dbeta, int rows, int C, int segments, int training, int relu,
                  void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* GroupNorm(groups, C, eps) + ReLU: the norm layer of the BiT towers (`--model_name resnetv2_50x3_bitm_in21k`, named by the help text
 * of finetune_image.py:23 and handed to timm at :191; timm 0.6.5 resnetv2.py norm_layer=GroupNormAct(num_groups=32)).  x: `images`
 * images of rows / images pixels each; biased statistics per (image, group of C / groups consecutive channels), the same in training
 * and eval mode.  mean / rstd [images][C] fp32 (the group's value once per channel) are saved for the backward call. */
size_t ia_gn_act_workspace_bytes(int rows, int C, int images);
int ia_gn_act_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int C, int images,
                  int groups, float eps, int relu, void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* dx (+ extra [rows, C] bf16 when not NULL), dgamma / dbeta [C] accumulated (NULL allowed) */
int ia_gn_act_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                  const void* extra, void* dx, float* dgamma, float* dbeta, int rows, int C, int images, int groups, int relu,
                  void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* cols [B*Ho*Wo, Kp] bf16: patch matrix of a k x k / stride / pad convolution read from NCHW fp32 images; column
 * (ky*k + kx)*C + c, columns >= k*k*C zero, Kp % 8 == 0 (the 7x7 stem, timm resnetv2.py create_resnetv2_stem) */
int ia_patches_nchw(const float* images, void* cols, int B, int C, int H, int W, int k, int stride, int pad, int Kp, ia_stream_t stream);
/* MaxPool2d(3, stride 2, padding 1): y [B*Ho*Wo, C] bf16, arg [B*Ho*Wo, C] u8 = window position of the first maximum */
int ia_maxpool3s2_fwd(const void* x, void* y, uint8_t* arg, int B, int H, int W, int C, ia_stream_t stream);
/* pad_zero != 0: window positions outside the image count as zeros -- ConstantPad2d(1, 0.) + MaxPool2d(3, 2, padding 0), the 'fixed'
 * stem of the BiT towers (timm resnetv2.py create_resnetv2_stem); ia_maxpool3s2_bwd serves both forms */
int ia_maxpool3s2_fwd_ex(const void* x, void* y, uint8_t* arg, int B, int H, int W, int C, int pad_zero, ia_stream_t stream);
int ia_maxpool3s2_bwd(const void* dy, const uint8_t* arg, void* dx, int B, int H, int W, int C, ia_stream_t stream);
/* rows read by a strided 1x1 convolution: y [B*Ho*Wo, C] = x[b, oy*stride, ox*stride, :]; the backward call writes
 * dx = base (NULL = zeros; may alias dx) + dy scattered onto the stride grid */
int ia_rows_subsample_fwd(const void* x, void* y, int B, int H, int W, int C, int stride, ia_stream_t stream);
int ia_rows_subsample_bwd(const void* dy, const void* base, void* dx, int B, int H, int W, int C, int stride, ia_stream_t stream);
/* plain convolution weight: what [Cout][ldw] bf16 (tap-major, channels padded to Cgp, row padded with zeros to ldw) from the
 * PyTorch layout w [Cout][Cg][kk] fp32; the grad call does dw += dwhat in the inverse mapping */
int ia_conv_weight_pack(const float* w, void* what, int Cout, int Cg, int kk, int Cgp, int ldw, ia_stream_t stream);
int ia_conv_weight_unpack_grad(const float* dwhat, float* dw, int Cout, int Cg, int kk, int Cgp, int ldw, ia_stream_t stream);

/* ---- image input pipeline on the GPU (src/data/data.py:838-866: timm create_transform(is_training=False) = PIL bicubic
 * resize -> ToTensor -> Normalize).  ia_resize_pass_u8 is one separable pass of Pillow's 8-bit resampler (Resample.c) over
 * a batch of equally sized uint8 RGB frames: horizontal != 0: src [B, other_len, in_len, 3] -> dst [B, other_len, out_len, 3];
 * else src [B, in_len, other_len, 3] -> dst [B, out_len, other_len, 3].  bounds [out_len][2] (first tap, tap count) and
 * coeffs [out_len][ksize] (int32, 8.22 fixed point) are Pillow's precompute_coeffs / normalize_coeffs_8bpc tables, built
 * by the host (item_alignment_amd/data/gpu_preproc.py): results are bit-identical to Image.resize(size, BICUBIC). */
int ia_resize_pass_u8(const uint8_t* src, uint8_t* dst, const int* bounds, const int* coeffs, int ksize, int B, int in_len, int out_len,
                      int other_len, int horizontal, ia_stream_t stream);
/* The same pass with an explicit source layout for the horizontal pass (rows src_pitch_bytes apart, frames src_frame_bytes apart): a
 * RandomResizedCrop window of a decoded frame is the frame's own pitch with src at the window's first pixel and in_len = the window
 * width; out_len may be the CenterCrop sub-range of a resize when bounds / coeffs are the matching slice of the tables (timm
 * create_transform as the reference calls it, src/data/data.py:838-841; torchvision on PIL images -> Pillow Resample.c). */
int ia_resize_pass_u8_ex(const uint8_t* src, size_t src_pitch_bytes, size_t src_frame_bytes, uint8_t* dst, const int* bounds,
                         const int* coeffs, int ksize, int B, int in_len, int out_len, int other_len, int horizontal, ia_stream_t stream);
/* One step of torchvision ColorJitter (= Pillow ImageEnhance.{Brightness,Contrast,Color}: Image.blend(degenerate, image, factor),
 * Blend.c float arithmetic) on uint8 frames [B,H,W,3] in place.  op [B] int32: 0 none, 1 brightness, 2 contrast, 3 saturation;
 * factor [B] fp32; scratch: B x 8 bytes of device memory (contrast means); any_contrast != 0 when some op[b] == 2.  One call per
 * position of the images' random op order (reference --color_jitter, data.py:841). */
int ia_color_jitter_step_u8(uint8_t* frames, const int* op, const float* factor, void* scratch, int B, int H, int W, int any_contrast,
                            ia_stream_t stream);
/* uint8 [B,S0,S1,3] -> fp32 [B,3,S0,S1] = (x/255 - mean)/std with IEEE fp32 division (ToTensor + Normalize), optional
 * per-image horizontal flip (flip: [B] device bytes or NULL); mean3 / std3: HOST arrays of 3 floats */
int ia_u8_to_nchw_normalized(const uint8_t* src, const uint8_t* flip, float* out, int B, int S0, int S1, const float* mean3,
                             const float* std3, ia_stream_t stream);

/* ---- embeddings (src/models/base.py:238-279, :501-556, :394-442) */
int ia_embed_ln_fwd(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const int32_t* extra_idx, const float* word,
                    const float* type, const float* pos, const float* extra, const float* gamma, const float* beta, void* z_out,
                    void* y, float* mean, float* rstd, int M, int H, float eps, float drop_p, uint32_t seed, uint32_t stream_id,
                    ia_stream_t stream);
size_t ia_embed_ln_bwd_workspace_bytes(int M, int H);
int ia_embed_ln_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma, const int64_t* ids,
                    const int64_t* type_ids, const int64_t* pos_ids, const int32_t* extra_idx, const int32_t* row_order, float* dword,
                    float* dtype, float* dpos, float* dextra, float* dgamma, float* dbeta, int M, int H, int L, int word_pad, int pos_pad,
                    float drop_p, uint32_t seed, uint32_t stream_id, void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* row_order (int32 [M], may be NULL): the rows sorted by position id, for unpadded token rows that have no [M/L, L] grid — the
 * position / token-type gradients are then run-length accumulated along that list instead of down the columns of the grid */

/* ---- ViT input side (timm PatchEmbed + cls token + pos_embed; src/models/multimodal.py:811) */
int ia_im2col_patch(const float* images, void* patches, int B, int C, int S, int P, ia_stream_t stream);
int ia_vit_tokens_fwd(const void* patch, const float* cls, const float* pos, void* tokens, int B, int NP, int H, ia_stream_t stream);
int ia_vit_tokens_bwd(const void* dtokens, void* dpatch, float* dcls, float* dpos, int B, int NP, int H, int accumulate,
                      ia_stream_t stream);

/* ---- CLS row pick with dropout (features[:, 0, :] -> dropout, src/models/base.py:104,140-141) */
int ia_gather_rows_fwd(const void* src, int ld, const int32_t* rows, float* out, int B, int H, float drop_p, uint32_t seed,
                       uint32_t stream_id, ia_stream_t stream);
int ia_gather_rows_bwd(const float* dout, int ld, const int32_t* rows, void* dsrc, int B, int H, float drop_p, uint32_t seed,
                       uint32_t stream_id, int accumulate, ia_stream_t stream);

/* ---- heads and loss (src/models/base.py:103-117, :139-157; nn.CrossEntropyLoss text.py:1292) */
#define IA_ACT_NONE 0
#define IA_ACT_TANH 1
int ia_linear_small_fwd(const float* x, int ldx, const float* W, const float* bias, float* y, int B, int N, int K, int act,
                        ia_stream_t stream);
int ia_linear_small_bwd(const float* dy, const float* y, const float* x, int ldx, const float* W, float* dx, int lddx, float* dW,
                        float* db, int B, int N, int K, int act, ia_stream_t stream);
int ia_pair_head_ce_fwd(const float* x, const float* y, const float* W, const float* bias, const int64_t* labels, float* logits,
                        float* probs, float* loss, float* loss_per, int B, int D, int C, ia_stream_t stream);
int ia_pair_head_ce_bwd(const float* probs, const int64_t* labels, const float* dloss, const float* x, const float* y, const float* W,
                        float* dx, float* dy, float* dW, float* db, int B, int D, int C, ia_stream_t stream);

/* span means of the auxiliary attribute-pair task (reference text.py:66-102 AuxiliaryTaskPair: mean of the token rows of a
 * key:value span, for the source and the target item; the pair head + CE above finishes it).  spans [S][2] int32 = absolute
 * (first row, end row) into seq [rows, ld] bf16; out [S][H] fp32.  Backward: dseq [B*L, H] bf16 overwritten; span_ptr [B+1]
 * delimits the spans of each sample. */
int ia_span_mean_fwd(const void* seq, int ld, const int* spans, float* out, int S, int H, ia_stream_t stream);
int ia_span_mean_bwd(const float* dout, const int* spans, const int* span_ptr, void* dseq, int B, int L, int H, ia_stream_t stream);

/* packed ("unpadded") self-attention: rows of all sequences back to back, sequence b = rows cu_seqlens[b] .. cu_seqlens[b+1]
 * (int32 [B+1], device; total_tokens = cu_seqlens[B]); q / k / v share the row stride ld_qkv; no key mask; lse2 / delta stay
 * [B, nh, Lmax] with Lmax the longest sequence.  Same arithmetic as ia_attn_fwd / ia_attn_bwd on the valid tokens. */
int ia_attn_fwd_varlen(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens, void* out, int ld_o,
                       float* lse2, int B, int nh, int Lmax, float scale, float drop_p, uint32_t seed, ia_stream_t stream);
int ia_attn_bwd_varlen(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens, const void* out,
                       const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv, int B, int nh,
                       int Lmax, float scale, float drop_p, uint32_t seed, ia_stream_t stream);

int ia_attn_fwd_varlen_ps(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens, void* out, int ld_o,
                          float* lse2, int B, int nh, int Lmax, float scale, float drop_p, uint32_t seed, ia_stream_t stream);
int ia_attn_bwd_varlen_ps(const void* q, const void* k, const void* v, int ld_qkv, const int* cu_seqlens, int total_tokens, const void* out,
                          const void* d_out, int ld_o, const float* lse2, float* delta, void* dq, void* dk, void* dv, int ld_dqkv, int B, int nh,
                          int Lmax, float scale, float drop_p, uint32_t seed, ia_stream_t stream);

/* ---- optimiser (torch.optim.AdamW, finetune_multimodal.py:296-308,460-468) */
/* dst[offset + c*rows + r] = src[offset + r*cols + c] for each of the n table entries {offset_lo, offset_hi, rows, cols} (uint32 x 4,
 * device memory; element offsets, the same in src and dst): the transposed weight shadows of ia_layer_weights::wt_*.  (ABI 7) */
int ia_transpose_bf16_batched(const void* src, void* dst, const void* table, int n, int max_tiles, ia_stream_t stream);
int ia_adamw_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, const void* chunk_table,
                  int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  ia_stream_t stream);
int ia_cast_f32_to_bf16(const float* src, void* dst, size_t n, ia_stream_t stream);
int ia_cast_bf16_to_f32(const void* src, float* dst, size_t n, ia_stream_t stream);

/* ---- whole-layer drivers: one call = every launch of one encoder layer, in order, on `stream`.
 * Weights: bf16 shadows for the GEMM operands, fp32 masters for bias / LayerNorm vectors. */
typedef struct {
  const void* w_qkv;   /* bf16 [3H, H]  query | key | value stacked (state_dict attention.self.{query,key,value}.weight) */
  const float* b_qkv;  /* fp32 [3H] */
  const void* w_o;     /* bf16 [H, H]   attention.output.dense.weight / timm attn.proj */
  const float* b_o;
  const float* ln1_g;  /* attention.output.LayerNorm (BERT post-LN) / norm1 (ViT pre-LN) */
  const float* ln1_b;
  const void* w_fc1;   /* bf16 [I, H]   intermediate.dense / mlp.fc1 */
  const float* b_fc1;
  const void* w_fc2;   /* bf16 [H, I]   output.dense / mlp.fc2 */
  const float* b_fc2;
  const float* ln2_g;  /* output.LayerNorm / norm2 */
  const float* ln2_b;
  /* (ABI 7) optional transposed bf16 copies, [H, 3H] / [H, H] / [H, I] / [I, H] (ia_transpose_bf16_batched keeps them in step with the
   * shadows above): the backward's data-gradient GEMMs dx = dy W then read W^T k-contiguously -- 2-16 % faster per launch than the
   * k-strided form (transpose reads out of LDS), 5.7 ms of a 434 ms step.  NULL = not provided (k-strided form). */
  const void* wt_qkv;
  const void* wt_o;
  const void* wt_fc1;
  const void* wt_fc2;
} ia_layer_weights;

typedef struct { /* fp32 gradient arena slots, same shapes as above; all accumulate (+=) */
  float* w_qkv; float* b_qkv; float* w_o; float* b_o; float* ln1_g; float* ln1_b;
  float* w_fc1; float* b_fc1; float* w_fc2; float* b_fc2; float* ln2_g; float* ln2_b;
} ia_layer_grads;

typedef struct {
  int B, L, H, I, nh;      /* tokens M = B*L; head dim = H/nh = 64 */
  int pre_ln;              /* 0 = BERT post-LN layer (RobertaLayer), 1 = ViT pre-LN block (timm Block) */
  float eps;               /* 1e-12 BERT / 1e-6 ViT */
  float hidden_drop, attn_drop;
  uint32_t seed;           /* per-step seed; the layer index is mixed in through layer_id */
  uint32_t layer_id;
  /* unpadded ("packed") sequences: when cu_seqlens (int32 [B+1], device) is not NULL the layer runs on total_tokens =
   * cu_seqlens[B] rows, sequence b owning rows cu_seqlens[b] .. cu_seqlens[b+1]; L is then the longest sequence and the
   * key_mask argument is ignored (every token of a sequence is attendable).  NULL / 0: padded [B, L] rows. */
  const int* cu_seqlens;
  int total_tokens;
  /* pre-LN (ViT) stacks, backward only (ABI 5): the b_fc2 gradient of a block is the column sum of its incoming gradient dy, i.e. of
   * the input gradient dx the block ABOVE has just produced.  dx_colsum_out != NULL: this block's last LayerNorm backward also adds
   * the column sums of its dx into that fp32 [H] vector (pass the b_fc2 gradient slot of the block below); dy_colsum_done != 0: the
   * block above did that for this block's dy, so the separate column-sum pass over dy is skipped.  Zero / NULL = round-3 behaviour. */
  float* dx_colsum_out;
  int dy_colsum_done;
  /* (ABI 8) nonzero: no output of a masked position (key_mask == 0) reaches the loss, so the gradient arriving at such rows is exactly
   * zero in every layer -- the attention backward may skip query blocks made of masked positions only (IA_ATTN_MASKED_ROWS_DEAD).
   * Zero = the round-5 behaviour (every row is computed). */
  int masked_rows_dead;
} ia_layer_cfg;

/* per-layer activation stash (saved by forward, read by backward) and shared backward scratch */
size_t ia_layer_stash_bytes(const ia_layer_cfg* cfg);
size_t ia_layer_bwd_scratch_bytes(const ia_layer_cfg* cfg);
/* x [M,H] bf16 -> y [M,H] bf16.  key_mask [B,L] uint8 or NULL.  stash (ia_layer_stash_bytes) receives what ia_layer_bwd needs. */
int ia_layer_fwd(const ia_layer_cfg* cfg, const ia_layer_weights* w, const void* x, const uint8_t* key_mask, void* y, void* stash,
                 ia_stream_t stream);
/* Forward only (evaluation / prediction; reference finetune_multimodal.py:470-563, 661-775 run the model under no_grad): the layer
 * without anything kept for a backward pass -- no gelu' stream, no pre-LayerNorm sums, dropout off whatever cfg says.  `scratch`
 * (>= ia_layer_infer_scratch_bytes) is transient: every layer of a stack may be handed the same buffer.  (ABI 5) */
size_t ia_layer_infer_scratch_bytes(const ia_layer_cfg* cfg);
int ia_layer_fwd_infer(const ia_layer_cfg* cfg, const ia_layer_weights* w, const void* x, const uint8_t* key_mask, void* y, void* scratch,
                       size_t scratch_bytes, ia_stream_t stream);
/* dy [M,H] bf16 -> dx [M,H] bf16 (dx may alias dy); parameter gradients accumulate into g. */
int ia_layer_bwd(const ia_layer_cfg* cfg, const ia_layer_weights* w, const ia_layer_grads* g, const void* x, const uint8_t* key_mask,
                 const void* y, const void* stash, const void* dy, void* dx, void* scratch, size_t scratch_bytes, ia_stream_t stream);
/* Split-residual form for stacks of post-LN layers: the gradient of the layer OUTPUT arrives as dy + dy2 (dy2 may be NULL) and the
 * gradient of the layer INPUT leaves as dx + dx2, dx2 being the residual-path part (the attention sub-block's LayerNorm input
 * gradient) - the next layer down takes (dx, dx2) as its (dy, dy2) and sums them inside its first LayerNorm backward, so no GEMM
 * carries a "+ aux" epilogue.  dx may alias dy, dx2 may alias dy2.  dx2 == NULL: dx holds the whole input gradient (bottom layer).
 * Pre-LN (ViT) layers: dy2 / dx2 must be NULL. */
int ia_layer_bwd2(const ia_layer_cfg* cfg, const ia_layer_weights* w, const ia_layer_grads* g, const void* x, const uint8_t* key_mask,
                  const void* y, const void* stash, const void* dy, const void* dy2, void* dx, void* dx2, void* scratch,
                  size_t scratch_bytes, ia_stream_t stream);

/* ---- data-parallel gradient exchange (pure data parallelism, one process per GPU; reference loop being sharded:
 * finetune_multimodal.py:371-468).  RCCL over xGMI behind four calls, for hosts that bind only this header (the Python host of this
 * repository uses torch.distributed, i.e. the same RCCL).  Rank 0 obtains an id and hands its IA_COMM_ID_BYTES bytes to the other
 * ranks by any out-of-band means; every rank then calls ia_comm_init (collective).  ia_comm_allreduce_bucket sums `count` elements in
 * place across the ranks, asynchronously on `stream` -- the caller launches it per gradient bucket as soon as the bucket's last
 * gradient kernel has been enqueued on that stream (or on a side stream ordered behind it) and divides by the world size in its
 * optimiser step (ia_adamw_flat's grad_scale).  RCCL is bound with dlopen at the first call (a copy already mapped into the process
 * is reused, whatever file name it was loaded under); IA_ERR_UNSUPPORTED = no librccl, a symbol missing, or not the 2.x API --
 * ia_comm_last_error() has the text (per calling thread).  PRECONDITION of ia_comm_init and ia_comm_allreduce_bucket: the calling
 * thread's current HIP device (hipSetDevice) is the GPU this rank owns -- RCCL binds the communicator to it.  (ABI 5) */
#define IA_COMM_ID_BYTES 128
#define IA_COMM_F32 0
#define IA_COMM_BF16 1
int ia_comm_unique_id(void* id_out);
int ia_comm_init(const void* id, int rank, int world_size, void** comm_out);
int ia_comm_allreduce_bucket(void* comm, void* buf, size_t count, int dtype, ia_stream_t stream);
int ia_comm_finalize(void* comm);
const char* ia_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ITEMALIGN_H */
