/* libeditor_hip.so - C ABI of the MI355X-native EDITOR hot path (gfx950 / CDNA4).
 *
 * The reference (924973292/EDITOR) is 100 % Python: its "operator interface" for this path is the
 * sequence of PyTorch ops inside modeling/make_model.py:150-258.  Each entry point below replaces one
 * of those op sequences (file:line given per function); the Python host (editor_amd/) binds them with
 * ctypes exactly as INTEGRATION.md shows.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator); kernels never
 *     allocate or free; workspaces are passed in;
 *   - every launch is asynchronous on `stream` (hipStream_t; pass torch.cuda.current_stream().cuda_stream);
 *   - return value is a hipError_t as int (0 = success); no exceptions, no global mutable state, no environment
 *     lookups, no allocation or synchronisation inside an entry point (measurement hooks live in separate debug
 *     libraries: include/editor_debug.h);
 *   - dtype suffix: _f32 = float activations, _bf16 = bfloat16 activations, _f16 = IEEE half activations (raw
 *     uint16 bits); arguments named `*_bf16` / `dtype` are dtype CODES: 0 = fp32, 1 = bf16, 2 = f16;
 *     statistics, residual stream, losses and parameter gradients are always fp32;
 *   - f16 activations carry LOSS-SCALED gradients (as the reference's amp.GradScaler does, engine/processor.py:94):
 *     the `scale` arguments of the cast / column-sum / LayerNorm-backward entry points apply and remove a
 *     power-of-two factor, so results equal the unscaled computation whenever nothing under- or overflows;
 *   - masks are uint8 (0/1), row-major; token rows are (sample, token) row-major.
 */
#ifndef EDITOR_HIP_H
#define EDITOR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* editor_stream_t;   /* == hipStream_t */

/* GEMM epilogues (applied after alpha / bias / rowscale, before beta*C): */
#define EDITOR_EPI_NONE 0
#define EDITOR_EPI_RESIDUAL 1 /* C = v + aux        aux: fp32 (M,N) residual stream  (x + drop_path(branch), vit_pytorch.py:217-218) */
#define EDITOR_EPI_GELU 2     /* aux = v ; C = gelu(v)   aux: pre-activation in the activation dtype (Mlp.fc1 -> act, :140-141);
                               * 16-bit kernels: aux may be NULL (a no-grad forward saves nothing: one output instead of two) */
#define EDITOR_EPI_GELU_BWD 3 /* C = v * gelu'(aux)      aux: saved pre-activation (backward of the above) */
#define EDITOR_EPI_COLSUM 0x100 /* OR-able (editor_gemm_bf16, bf16 C, M >= 2048, N >= 512, A k-major, splitk 1): also write
                                 * the column sums of every 256-row tile of the ROUNDED C to splitk_ws[(M+255)/256][N] - the bias
                                 * gradient of the layer this gradient feeds, folded by editor_reduce_rows */
#define EDITOR_EPI_AUX_GRAD 0x400 /* OR-able with EDITOR_EPI_GELU / _GELU_BWD (16-bit kernels): aux holds gelu'(pre-activation) instead
                                   * of the pre-activation - the forward saves the derivative (the only thing the backward needs of it), so
                                   * the dgrad epilogue is one multiply instead of an erfc + exponential per element */
#define EDITOR_EPI_FORCE_PP 0x200 /* OR-able: run the 256x256 ping-pong kernel whatever the shape heuristic says (N >= 256,
                                   * whole 64-deep K-tiles); tests use it to reach that kernel's edge cases */
#define EDITOR_EPI_PIPE128 0x800 /* OR-able: prefer the 256x128 three-stage kernel over the 256x256 ping-pong kernel (few token rows:
                                  * twice the tiles fill more of the 256 CUs; the caller's tile-count heuristic decides,
                                  * editor_amd.ops.gemm).  Ignored with EDITOR_EPI_COLSUM / _FORCE_PP / _TILE_ROWS. */
#define EDITOR_EPI_STAGGER(c) (((c) & 63) << 17) /* OR-able (256x256 ping-pong kernel, more than 256 tiles; ignored otherwise): the
                                  * first round's workgroups start spread over c * 2048 shader cycles, so that the CUs leave
                                  * lockstep and one CU's HBM-bound epilogue runs beside the others' K loops.  Same results bit
                                  * for bit.  The caller picks c ~ one tile period (editor_amd.ops.gemm_stagger). */
#define EDITOR_EPI_TILE_ROWS(h) ((((h) / 16) & 15) << 12) /* OR-able, h = 208 | 256 (ping-pong kernel, both operands k-major,
                                   * split-K 1, beta 0): rows per output tile.  M = 49 536 token rows x 768 columns are 582 full
                                   * tiles = 2.27 rounds of the 256 CUs; 208-row tiles make it 2.8 rounds of smaller tiles.
                                   * EDITOR_EPI_COLSUM then writes ceil(M / h) partial rows.  hipErrorInvalidValue when the
                                   * conditions do not hold; the caller picks h (editor_amd.ops.gemm_tile_rows). */

/* ---- token selection (non-differentiable) ---------------------------------------------------- */

/* Frequency.py:65-84 + :42-56 - 4-level Haar DWT of each modality (pytorch_wavelets AFB2D, lowlevel.py:336-347),
 * coefficient mean over modalities, inverse DWT (SFB2D, lowlevel.py:671-680), channel mean, count of >0 pixels
 * per 16x16 window.  rgb/nir/tir: (B,C,H,W) fp32 NCHW (tir may be NULL: two-modality form);
 * counts: (B, H/16*W/16) int32, patches row-major as PatchEmbed flattens them (vit_pytorch.py:457). */
int editor_freq_counts_f32(const float* rgb, const float* nir, const float* tir, int B, int C, int H, int W,
                           int32_t* counts, editor_stream_t stream);

/* the same for 2..4 modalities given explicitly (nmod = 4: the synthetic 4-modal configuration; the mean of
 * Frequency.py:71-74 over one more term) */
int editor_freq_counts_nmod_f32(const float* m0, const float* m1, const float* m2, const float* m3, int nmod, int B, int C,
                                int H, int W, int32_t* counts, editor_stream_t stream);

/* torch.topk(k) -> sort -> scatter_ to a bool row (Frequency.py:58-62, SFTS.py:155-158) with torch's CPU tie
 * order (libstdc++ partial_sort if k*64<=n else nth_element).  vals: (rows,n); `group` consecutive rows OR
 * into one mask row (SFTS.py:159-162: OR over heads); mask: (rows/group, n) uint8, fully overwritten. */
int editor_topk_mask_i32(const int32_t* vals, int rows, int n, int k, int group, uint8_t* mask, editor_stream_t stream);
int editor_topk_mask_f32(const float* vals, int rows, int n, int k, int group, uint8_t* mask, editor_stream_t stream);

/* Part_Attention rollout (SFTS.py:150-153): CLS row of A_{L-1} @ ... @ A_0 without the CLS column.
 * probs: L tensors of (BH,T,ldp) fp32 (rows padded to ldp >= T floats) spaced `layer_stride` floats apart;
 * scores: (BH, T-1) fp32. */
int editor_attn_rollout_f32(const float* probs, int L, int BH, int T, int ldp, long layer_stride, float* scores,
                            editor_stream_t stream);

/* index = a | b | c | d (SFTS.py:185-187); b, c, d may be NULL. */
int editor_mask_or(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, uint8_t* out, long n,
                   editor_stream_t stream);

/* ---- row kernels (HBM-bound) -------------------------------------------------------------------- */

/* nn.LayerNorm over D (vit_pytorch.py:206,211,268-297,519); x fp32 (M,D) residual stream; y fp32 or bf16.
 * rowmask (uint8, may be NULL): rows with mask 0 produce y = 0 (AttentionMask/MlpMasked zero the LN output of
 * unselected tokens, vit_pytorch.py:245,162); mask row = row % mask_period (0 = no wrap).  mean/rstd: (M) fp32. */
int editor_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, long M, int D,
                         const uint8_t* rowmask, int mask_period, void* y, int y_bf16, float* mean, float* rstd,
                         const int* m_live, editor_stream_t stream);
/* Residual add + LayerNorm in one pass (bf16 mode, cfg.MODEL.BRANCH16): x_out[m,:] = x[m,:] + rowscale[m] * branch[m,:] with
 * `branch` the 16-bit (b16 = 1 bf16 / 2 f16) output of the projection / fc2 product written with the plain epilogue
 * (vit_pytorch.py:217-218: x + drop_path(branch)), y = LayerNorm(x_out) in the same 16-bit type, mean / rstd of x_out.  Dense
 * rows, D a multiple of 256 (<= 1024).  rowscale may be NULL. */
int editor_resid_add_layernorm_fwd(const float* x, const void* branch, int b16, const float* rowscale, const float* gamma,
                                   const float* beta, float eps, long M, int D, float* x_out, void* y, float* mean,
                                   float* rstd, editor_stream_t stream);
/* backward: dx_out = (dx_in ? dx_in : 0) + dLN/dx ; dgamma/dbeta: ONE (2,D) fp32 buffer (dbeta == dgamma + D; NULL to
 * skip).  workspace: ws_rows*2*D floats. */
int editor_layernorm_bwd(const void* dy, int dy_bf16, float dy_scale /* dy is multiplied by it on load */,
                         const float* x, const float* gamma, const float* mean,
                         const float* rstd, long M, int D, const uint8_t* rowmask, int mask_period,
                         const float* dx_in, float* dx_out, float* dgamma, float* dbeta, float* workspace,
                         int ws_rows, const int* m_live, editor_stream_t stream);
/* The same (dense 16-bit dy, no row mask) with the consumer's cast fused in: also writes cast_out (dtype of dy) =
 * round(dx_out * cast_rowscale[row] * cast_scale) - the drop-path-scaled, loss-scaled 16-bit gradient the previous linear
 * layer's backward takes (vit_pytorch.py:217-218 backward) - and, when cast_colsum != NULL, cast_colsum[n] =
 * cast_colsum_scale * sum_m cast_out[m,n] (that layer's bias gradient): what editor_cast_rows_colsum would compute in a
 * second pass over dx_out.  workspace: ws_rows*3*D floats. */
int editor_layernorm_bwd_cast(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma,
                              const float* mean, const float* rstd, long M, int D, const float* dx_in, float* dx_out,
                              float* dgamma, float* dbeta, float* workspace, int ws_rows, void* cast_out,
                              const float* cast_rowscale, float cast_scale, float* cast_colsum, float cast_colsum_scale,
                              editor_stream_t stream);
/* "_parts" forms (round 4): the same kernels WITHOUT their second-stage fold - the partial rows stay in the caller's workspace
 * and *nparts (host) receives their count P; the caller folds several such sets with ONE editor_reduce_rows_multi launch (a
 * transformer block's backward: six sets, whose totals nothing needs before the block ends).  Layouts: layernorm: [P][2][D]
 * (dgamma | dbeta rows) at workspace, and - cast form with want_colsum - the cast output's column sums [P][D] at
 * workspace + ws_rows*2*D; colsum / cast_rows_colsum: [P][N] / [P][D] at workspace. */
int editor_layernorm_bwd_parts(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma,
                               const float* mean, const float* rstd, long M, int D, const uint8_t* rowmask, int mask_period,
                               const float* dx_in, float* dx_out, float* workspace, int ws_rows, const int* m_live,
                               int* nparts, editor_stream_t stream);
int editor_layernorm_bwd_cast_parts(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma,
                                    const float* mean, const float* rstd, long M, int D, const float* dx_in, float* dx_out,
                                    float* workspace, int ws_rows, void* cast_out, const float* cast_rowscale,
                                    float cast_scale, int want_colsum, int* nparts, editor_stream_t stream);
int editor_colsum_parts(const void* dy, int dy_bf16, long M, int N, long ld, float* workspace, int ws_rows, int* nparts,
                        editor_stream_t stream);
int editor_cast_rows_colsum_parts(const float* in, const float* rowscale, long M, int D, void* out, int out_bf16,
                                  float* workspace, int ws_rows, float scale, int* nparts, editor_stream_t stream);
/* count <= 8 folds out[j][c] = scale[j] * sum_{p < P[j]} partials[j][p * ncol[j] + c] in one launch; the five arrays are HOST
 * arrays of `count` entries (device pointers inside).  Same fixed summation order as editor_reduce_rows. */
int editor_reduce_rows_multi(int count, const float* const* partials, const int* P, const long* ncol, float* const* out,
                             const float* scale, editor_stream_t stream);
/* out[n] = scale * sum_m dy[m,n]  (bias gradients of every nn.Linear).  workspace: ws_rows*N floats. */
int editor_colsum(const void* dy, int dy_bf16, long M, int N, long ld, float* out, float* workspace, int ws_rows,
                  float scale, editor_stream_t stream);
int editor_reduce_rows(const float* partials, int P, long ncol, float* out, int accumulate, float scale,
                       editor_stream_t stream);
/* nn.GELU() exact erf (vit_pytorch.py:130,141) and its derivative */
int editor_gelu_fwd(const void* a, void* g, long n, int bf16, editor_stream_t stream);
int editor_gelu_bwd(const void* a, const void* dg, void* da, long n, int bf16, editor_stream_t stream);
int editor_cast_f32_to_bf16(const float* in, uint16_t* out, long n, editor_stream_t stream);
int editor_cast_f32_to_f16(const float* in, uint16_t* out, long n, editor_stream_t stream);
int editor_cast_f16_to_f32(const uint16_t* in, float* out, long n, editor_stream_t stream);
/* out[m,:] = in[m,:] * rowscale[m] * scale (rowscale may be NULL): fp32 gradient -> GEMM operand dtype, with the
 * per-sample drop-path factor keep/keep_prob of vit_pytorch.py:52-69 (and the f16 loss scale) folded in. */
int editor_cast_rows(const float* in, const float* rowscale, long M, int D, void* out, int out_bf16,
                     const int* m_live, float scale, editor_stream_t stream);
/* editor_cast_rows plus colsum[d] = sum_m out[m,d] (the rounded values): the bias gradient of the nn.Linear that `out`
 * feeds, without another pass over it.  D a multiple of 256; workspace: ws_rows*D floats. */
int editor_cast_rows_colsum(const float* in, const float* rowscale, long M, int D, void* out, int out_bf16, float* colsum,
                            float* workspace, int ws_rows, float scale, float colsum_scale /* applied to colsum only */,
                            editor_stream_t stream);
int editor_cast_bf16_to_f32(const uint16_t* in, float* out, long n, editor_stream_t stream);

/* split-precision producers (COMPUTE_DTYPE 'f16x2'): the same kernels writing x as the half pair hi = half(x),
 * lo = half(x - hi).  editor_split_f32: a flat fp32 tensor times a power-of-two scale (n % 4 == 0). */
int editor_layernorm_fwd_f16x2(const float* x, const float* gamma, const float* beta, float eps, long M, int D,
                               const uint8_t* rowmask, int mask_period, uint16_t* y_hi, uint16_t* y_lo, float* mean,
                               float* rstd, const int* m_live, editor_stream_t stream);
int editor_im2col16_f16x2(const float* img, int B, int C, int H, int W, uint16_t* out_hi, uint16_t* out_lo,
                          editor_stream_t stream);
int editor_split_f32(const float* in, uint16_t* hi, uint16_t* lo, long n, float scale, editor_stream_t stream);

/* PatchEmbed_overlap with stride == patch == 16 (vit_pytorch.py:449-458): im2col rows (b*N+p), cols (c,i,j). */
int editor_im2col16(const float* img, int B, int C, int H, int W, void* out, int out_bf16, editor_stream_t stream);
/* cls token + pos_embed + SIE_COE * sie_embed[cam] (vit_pytorch.py:627-637).  Btot samples may hold several
 * modalities stacked on the batch axis sharing `cam` (length Bcam): cam index = b % Bcam.  sie may be NULL. */
int editor_embed_assemble(const void* patch, int patch_bf16, const float* cls, const float* pos, const float* sie,
                          const long* cam, int Bcam, float coef, long Btot, int T, int D, float* x,
                          editor_stream_t stream);
/* backward: dpatch (cast, * dpatch_scale), dpos (T,D), dsie (ncam,D; may be NULL).  workspace (required):
 * max(EDITOR_EMBED_POS_SPLITS * T * D, Btot * D) floats (partial rows of the two-stage dpos sum, then the per-sample row sums) */
#define EDITOR_EMBED_POS_SPLITS 8
int editor_embed_assemble_bwd(const float* dx, const long* cam, int Bcam, int ncam, float coef, long Btot, int T, int D,
                              void* dpatch, int dpatch_bf16, float dpatch_scale, float* dpos, float* dsie,
                              float* workspace /* Btot*D */, editor_stream_t stream);

/* SFTS.forward mask application + BCC loss (SFTS.py:208-225).  feat/out: (nmod,B,T,D) fp32; index (B,T-1) uint8;
 * loss (1) fp32 or NULL (eval); workspace: ws_len floats. */
int editor_sfts_apply(const float* feat, const uint8_t* index, int nmod, long B, int T, int D, float* out,
                      float* loss, float* workspace, int ws_len, editor_stream_t stream);
int editor_sfts_apply_bwd(const float* feat, const uint8_t* index, const float* dout, const float* dloss, int nmod,
                          long B, int T, int D, float* dfeat, editor_stream_t stream);
/* cls + masked-mean pooling of the fused tokens (make_model.py:186-203). x (B,nmod*T,D) -> out (nmod,B,2D), num (B) */
int editor_pool_fwd(const float* x, long B, int nmod, int T, int D, float* out, float* num, editor_stream_t stream);
int editor_pool_bwd(const float* dout, const float* num, long B, int nmod, int T, int D, float* dx,
                    editor_stream_t stream);
int editor_rowmask_mul(float* x, const uint8_t* rowmask, int period, long M, int D, editor_stream_t stream);

/* ---- contractions ---------------------------------------------------------------------------------- */

/* C = alpha * opA(A) opB(B) (+bias[n]) (+beta*C) (*rowscale[m]) on the exact-fp32 matrix cores.
 * transA=0: A stored (M,K) row-major, 1: stored (K,M).  transB=0: B stored (N,K) (nn.Linear weight layout),
 * 1: stored (K,N).  Two batch levels (count, element strides).  splitk>1 accumulates with fp32 atomics.
 * Replaces F.linear / torch.matmul in vit_pytorch.py:139-145,184-198,240-258 and make_model.py:162-171,205-209. */
int editor_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb, long ldc,
                    int transA, int transB, int batch1, long sA1, long sB1, long sC1, int batch2, long sA2, long sB2,
                    long sC2, float alpha, float beta, const float* bias, const float* rowscale, int splitk,
                    int epilogue, float* aux, long ldaux, editor_stream_t stream);

/* bf16 MFMA form of the same contraction (fp32 accumulate; C bf16 or fp32 per c_f32).  No batching.
 * Requirements: 16-byte aligned pointers, lda/ldb multiples of 8, N and ldc multiples of 4, the contiguous
 * extent of each operand a multiple of 8.  transA/transB select the LDS transpose-read path
 * (ds_read_b64_tr_b16) so dgrad / wgrad need no materialised transposes.  splitk>1 requires c_f32. */
int editor_gemm_bf16(const uint16_t* A, const uint16_t* B, void* C, int c_f32, int M, int N, int K, long lda, long ldb,
                     long ldc, int transA, int transB, float alpha, float beta, const float* bias,
                     const float* rowscale, int splitk, int epilogue, void* aux, long ldaux,
                     float* splitk_ws /* splitk*M*N floats or NULL (then split-K uses fp32 atomics) */,
                     const int* m_live /* see below */, editor_stream_t stream);

/* Split-precision forward product (COMPUTE_DTYPE 'f16x2'; replaces the fp32 F.linear of vit_pytorch.py:139-145,184-198 at
 * fp32-class accuracy on the half matrix cores): every operand is a PAIR of half matrices x = hi + lo of the same shape
 * and leading dimension (A (M,K), B (N,K) = nn.Linear layout, both k-major);
 *     C = alpha * (A_hi B_hi^T + A_lo B_hi^T + A_hi B_lo^T) (+bias) (*rowscale)   [lo.lo, 2^-22 relative, is dropped]
 * accumulated in ONE fp32 accumulator over three passes of the K loop.  c_f32 != 0: C fp32 (C_lo NULL; epilogue NONE or
 * RESIDUAL with aux = fp32 residual); c_f32 == 0: the result leaves as a pair again, C = half(v), C_lo = half(v - C)
 * (epilogue NONE, or GELU | AUX_GRAD: v = exact-erf gelu of the UNROUNDED pre-activation x, aux = half(gelu'(x)) for the
 * 16-bit backward).  K % 64 == 0, N % 8 == 0, lda / ldb / ldc / ldaux multiples of 8; m_live as below.
 * EDITOR_EPI_TILE_ROWS(208) selects the short tiles.  Half keeps SUBNORMAL low-order parts (pinned by
 * tests/test_gpu_select.py::test_probe_mfma_f16_keeps_subnormal_operands); weights are handed over pre-scaled by a
 * power of two (editor_split_f32 / editor_split_multi) with alpha carrying its inverse. */
int editor_gemm_f16x2(const uint16_t* A_hi, const uint16_t* A_lo, const uint16_t* B_hi, const uint16_t* B_lo, void* C,
                      void* C_lo, int c_f32, int M, int N, int K, long lda, long ldb, long ldc, float alpha,
                      const float* bias, const float* rowscale, int epilogue, void* aux, long ldaux, const int* m_live,
                      editor_stream_t stream);

/* The weight gradients of ONE transformer block in one launch (vit_pytorch.py:139-145,184-198 backward): problem i is
 * dW_i (N_i, K_i) fp32 contiguous = alpha * dy_i^T x_i with dy_i (M, N_i) and x_i (M, K_i) 16-bit row-major over the same M
 * token rows (M % 64 == 0; N_i, K_i multiples of 256; count <= 4).  dy / x / dw / N / K are HOST arrays of `count` entries.
 * grid = all 256x256 output tiles back to back x `splitk` reduction splits (pick splitk so that tiles x splitk fills whole
 * rounds of the 256 CUs); partial tiles go to slabs in ws (splitk * sum N_i K_i floats) and are folded in a fixed order
 * (deterministic).  dtype: 1 = bf16, 2 = f16.  m_live: device scalar, live token rows (rows beyond are zero), or NULL. */
int editor_gemm_wgrad_group(int dtype, int count, const uint16_t* const* dy, const uint16_t* const* x, float* const* dw,
                            const int* N, const int* K, int M, float alpha, int splitk, float* ws, const int* m_live,
                            editor_stream_t stream);
/* The same launch (dense rows: no m_live) with a MEMORY-BOUND ROLE riding in it (round 4, opt-in: cfg / EDITOR_WGRAD_LN=1): the first
 * `nmem` workgroups (8 .. 256, a multiple of 8) do the LayerNorm backward that is independent of, and adjacent to, the block's
 * weight gradients (Block.norm1, vit_pytorch.py:215-220 backward) WITH the 16-bit copy of its result for the block below - exactly
 * editor_layernorm_bwd_cast_parts' arithmetic per row (D = 768 or 1024): dx_out = dLN(ln_dy * ln_dy_scale) + dx_in, cast_out =
 * 16-bit(dx_out * cast_rowscale[row] * cast_scale); partial rows, ONE PER MEMORY WORKGROUP: partials [nmem][2][D] (dgamma, dbeta),
 * cast_partials [nmem][D] (column sums of the rounded copy; NULL to skip) - to be folded by editor_reduce_rows(_multi) with P = nmem.
 * On this runtime an HBM-bound and an MFMA-bound kernel do not overlap across queues; two roles of one launch do. */
int editor_gemm_wgrad_group_ln(int dtype, int count, const uint16_t* const* dy, const uint16_t* const* x, float* const* dw,
                               const int* N, const int* K, int M, float alpha, int splitk, float* ws,
                               const uint16_t* ln_dy, float ln_dy_scale, const float* ln_x, const float* gamma, const float* mean,
                               const float* rstd, long ln_M, int D, const float* dx_in, float* dx_out, float* partials,
                               uint16_t* cast_out, const float* cast_rowscale, float cast_scale, float* cast_partials, int nmem,
                               editor_stream_t stream);

/* GROUPED forward / dgrad products (round 4): `count` <= 4 products C_i = alpha A_i B_i^T (+ bias_i) (* rowscale_i) (+ epilogue) of
 * IDENTICAL shape and epilogue kind, both operands k-major (A_i (M,K), B_i (N,K)), as ONE launch of the 256x256 ping-pong kernel -
 * the three per-modality blocks of BlockMask (vit_pytorch.py:311-317), whose kept tokens fill a third of the chip each.  A, B, C,
 * bias, rowscale, aux: HOST arrays of `count` device pointers (bias / rowscale / aux may be NULL, or hold NULL entries); dtype:
 * 1 = bf16, 2 = f16; epilogue: EDITOR_EPI_NONE / _RESIDUAL / _GELU / _GELU_BWD (| EDITOR_EPI_AUX_GRAD); M >= 256, N % 256 == 0,
 * K % 64 == 0; m_live as editor_gemm_bf16 (shared by the problems).  Bit-identical to `count` calls of editor_gemm_bf16 / _f16. */
int editor_gemm_group(int dtype, int count, const uint16_t* const* A, const uint16_t* const* B, void* const* C, int c_f32, int M,
                      int N, int K, long lda, long ldb, long ldc, float alpha, const float* const* bias,
                      const float* const* rowscale, int epilogue, void* const* aux, long ldaux, const int* m_live,
                      editor_stream_t stream);

/* the same contraction on IEEE-half operands (v_mfma_f32_16x16x32_f16): C half or fp32 */
int editor_gemm_f16(const uint16_t* A, const uint16_t* B, void* C, int c_f32, int M, int N, int K, long lda, long ldb,
                    long ldc, int transA, int transB, float alpha, float beta, const float* bias,
                    const float* rowscale, int splitk, int epilogue, void* aux, long ldaux, float* splitk_ws,
                    const int* m_live, editor_stream_t stream);

/* m_live (device int32 scalar, may be NULL) - compacted HMA without a host round trip: buffers and launches are sized
 * for the worst-case row count, only the first *m_live token rows are live.  Row kernels process rows below
 * roundup64(*m_live) (rows in [*m_live, roundup64) carry mask 0 / zeros), GEMMs skip tiles of dead rows (forward, dgrad)
 * or shorten the reduction (wgrad: transA); dead rows are never read by anyone. */

/* Attention.forward / AttentionMask.forward on packed qkv rows (B*T, 3*heads*hd) (vit_pytorch.py:184-198,240-258).
 * mask (B,T) uint8 or NULL.  out (B*T, heads*hd).  probs (B,heads,T,T) fp32: the softmax output the backbone
 * returns (vit_pytorch.py:638-644); REQUIRED in the f32 form (it is also the score buffer). */
int editor_attention_fwd_f32(const float* qkv, int B, int T, int heads, int hd, float scale, const uint8_t* mask,
                             float* out, float* probs, editor_stream_t stream);
int editor_attention_bwd_f32(const float* qkv, const float* dout, const float* probs, int B, int T, int heads, int hd,
                             float scale, float* dqkv, float* workspace, editor_stream_t stream);

/* One step of the attention rollout (SFTS.py:150-153) WITHOUT materialised probabilities: r_out[bh][k] =
 * sum_q r_in[bh][q] * P_l[q,k], P_l recomputed from layer l's packed qkv (B*T, 3*heads*64) and the forward's lse
 * (heads*B*T).  r_in NULL = one-hot CLS row (first step, last layer).  final_step: r_out is (B*heads, T-1) and receives
 * r[1:] (the CLS->patch scores); otherwise (B*heads, T).  Dense, unmasked sequences (the backbone). */
int editor_attn_rollout_step_bf16(const uint16_t* qkv, const float* lse, const float* r_in, int B, int T, int heads, int hd,
                                  float scale, float* r_out, int final_step, editor_stream_t stream);
int editor_attn_rollout_step_f16(const uint16_t* qkv, const float* lse, const float* r_in, int B, int T, int heads, int hd,
                                 float scale, float* r_out, int final_step, editor_stream_t stream);

/* Fused 16-bit form.  hd = 64 (every shipped configuration), 32 or 96 (the factory's other head widths, vit_pytorch.py:704-727:
 * the same kernels built per width; whole-sequence form up to 416 tokens at hd = 96); any other hd is hipErrorInvalidValue.
 * T <= 608: one workgroup per (sample, head) with the whole key range in LDS.
 * T > 608 (joint HMA block of the 4-modal 512-token configuration): 64 own rows per workgroup, the other side streamed
 * through LDS in 256-row chunks; probs must be NULL there.  probs optional (NULL skips the write).  lse (heads*Mtot fp32, log2 units
 * of the scaled scores, +inf for masked queries) is written by the forward (NULL to skip) and consumed by the backward,
 * which recomputes the probabilities from it; `out` is the forward output (delta = rowsum(dO*O));
 * workspace: heads*Mtot floats.  Variable-length (compacted HMA) form: cu (B+1 int32) gives each sequence's packed row
 * range, T = the longest sequence, Mtot = total packed rows (pad rows are not touched); cu == NULL: dense, Mtot = B*T. */
int editor_attention_fwd_bf16(const uint16_t* qkv, int B, int T, int heads, int hd, float scale, const uint8_t* mask,
                              uint16_t* out, float* probs, int ldp /* row stride of probs, multiple of 4 */, float* lse,
                              const int* cu, long Mtot, editor_stream_t stream);
int editor_attention_bwd_bf16(const uint16_t* qkv, const uint16_t* dout, const uint16_t* out, const float* lse, int B,
                              int T, int heads, int hd, float scale, const uint8_t* mask, uint16_t* dqkv,
                              float* workspace, const int* cu, long Mtot, editor_stream_t stream);
/* The same backward, which ALSO leaves the column sums of dqkv - the qkv bias gradient (Attention.qkv, vit_pytorch.py:177,186) -
 * as one partial row per sequence: colparts[B][3*heads*hd] fp32 (every entry written), taken from the dQ / dK / dV accumulators on
 * their way out (fp32, before the 16-bit rounding) and to be folded by editor_reduce_rows(_multi); replaces an editor_colsum pass
 * over dqkv.  T <= 608 (hd = 96: T <= 160); otherwise hipErrorInvalidValue and the caller sums dqkv itself.  Deterministic. */
int editor_attention_bwd_colsum_bf16(const uint16_t* qkv, const uint16_t* dout, const uint16_t* out, const float* lse, int B,
                                     int T, int heads, int hd, float scale, const uint8_t* mask, uint16_t* dqkv,
                                     float* workspace, const int* cu, long Mtot, float* colparts, editor_stream_t stream);
int editor_attention_bwd_colsum_f16(const uint16_t* qkv, const uint16_t* dout, const uint16_t* out, const float* lse, int B,
                                    int T, int heads, int hd, float scale, const uint8_t* mask, uint16_t* dqkv,
                                    float* workspace, const int* cu, long Mtot, float* colparts, editor_stream_t stream);
/* Split-precision attention forward (COMPUTE_DTYPE 'f16x2'): q, k, v as half pairs qkv = qkv_hi + qkv_lo (both (rows,
 * 3*heads*64), the split output of the qkv product); every contraction is three half MFMAs into one fp32 accumulator
 * (S = Q_lo K_hi + Q_hi K_lo + Q_hi K_hi; O likewise on the 2^12-scaled probability pair), the softmax is the reference's
 * fp32 exp(s - max) / sum (vit_pytorch.py:184-198,240-258).  out_hi / out_lo: the output as a half pair (operand of the proj
 * product); probs (optional, dense form only): fp32 softmax rows for the rollout (:638-644); lse: for
 * editor_attention_bwd_f16, which runs on the hi halves.  Any T (<= 288 tokens: whole sequence in LDS; longer: 128-key
 * chunks); mask / cu / Mtot as editor_attention_fwd_bf16. */
int editor_attention_fwd_f16x2(const uint16_t* qkv_hi, const uint16_t* qkv_lo, int B, int T, int heads, int hd, float scale,
                               const uint8_t* mask, uint16_t* out_hi, uint16_t* out_lo, float* probs, int ldp, float* lse,
                               const int* cu, long Mtot, editor_stream_t stream);
/* One rollout step (SFTS.py:150-153) of the split-precision mode WITHOUT materialised probabilities (round 4): as
 * editor_attn_rollout_step_f16, with the scores formed from layer l's q / k half PAIRS in three MFMA passes (fp32-class, the
 * forward's own S) and P = exp2(S - lse) with the lse editor_attention_fwd_f16x2 wrote - the (L,3B,h,T,T) fp32 probability tensor
 * (vit_pytorch.py:638-644) is then never written.  Dense, unmasked sequences, T <= 608. */
int editor_attn_rollout_step_f16x2(const uint16_t* qkv_hi, const uint16_t* qkv_lo, const float* lse, const float* r_in, int B,
                                   int T, int heads, int hd, float scale, float* r_out, int final_step, editor_stream_t stream);
int editor_attention_fwd_f16(const uint16_t* qkv, int B, int T, int heads, int hd, float scale, const uint8_t* mask,
                             uint16_t* out, float* probs, int ldp, float* lse, const int* cu, long Mtot,
                             editor_stream_t stream);
int editor_attention_bwd_f16(const uint16_t* qkv, const uint16_t* dout, const uint16_t* out, const float* lse, int B,
                             int T, int heads, int hd, float scale, const uint8_t* mask, uint16_t* dqkv,
                             float* workspace, const int* cu, long Mtot, editor_stream_t stream);

/* ---- compacted (variable-length) HMA: packing plan and row movement (csrc/compact.hip) ------------------ */
/* index (B,N) uint8 -> cu (B+1): exclusive prefix sum of L_b = 1 + #selected; tok (>= cu[B] ints): token id (0 = cls,
 * n+1 = patch n) of every packed row, sample-major. */
int editor_compact_plan(const uint8_t* index, int B, int N, int* cu, int* tok, editor_stream_t stream);
/* row maps between the dense (nmod,B,T,D) tokens, layout A (nmod x MA rows, modality-major) and layout B (MB rows,
 * sample-major: [R rows | N rows | T rows] per sample); -1 / mask 0 mark pad rows.  cu3 = nmod * cu. */
int editor_compact_maps(const int* cu, const int* tok, int B, int T, int nmod, long MA, long MB, int* mapA, int* mapB,
                        int* mapCls, uint8_t* maskA, uint8_t* maskB, int* cu3, editor_stream_t stream);
/* rows [*live, roundup64(*live)) of a packed (rows, row_bytes) buffer <- 0 (row_bytes a multiple of 16): what the live-row
 * reductions (64-row K-tiles of the weight gradients, LayerNorm backward) may read beyond the live extent */
int editor_zero_tail_rows(void* buf, long row_bytes, long rows, const int* live, editor_stream_t stream);
int editor_gather_rows(const float* in, const int* src, long R, int D, float* out, const int* r_live /* or NULL */,
                       int live_mul, long live_stride, editor_stream_t stream);
int editor_scatter_rows(const float* dy, const int* src, long R, int D, long rows_out, float* dx, editor_stream_t stream);
/* ... without zero-filling dx first: rows no index names keep whatever the buffer held - for consumers that provably never read
 * them (the layout-A gather's gradient -> editor_sfts_apply_bwd reads selected rows only), or together with
 * editor_zero_tail_rows for the pad rows [live, roundup64(live)) the live-row kernels read. */
int editor_scatter_rows_nofill(const float* dy, const int* src, long R, int D, float* dx, editor_stream_t stream);
/* make_model.py:186-203 on layout B */
int editor_pool_packed_fwd(const float* x, const int* cu, long B, int nmod, int D, float* out, float* num,
                           editor_stream_t stream);
int editor_pool_packed_bwd(const float* dout, const float* num, const int* cu, long B, int nmod, int D, long rows,
                           float* dx, editor_stream_t stream);
/* ... without the zero fill of dx (every live row of every sample is written; the caller zeroes the pad rows with
 * editor_zero_tail_rows) */
int editor_pool_packed_bwd_nofill(const float* dout, const float* num, const int* cu, long B, int nmod, int D, float* dx,
                                  editor_stream_t stream);

/* ---- head kernels ------------------------------------------------------------------------------------ */

/* nn.BatchNorm1d on (B,C) rows with row stride ldx (make_model.py:115,120,140).  training: batch statistics, running
 * stats updated in place (momentum, unbiased variance); eval: running stats.  save_mean/save_invstd: (C), training only. */
int editor_bn1d_fwd(const float* x, long ldx, int B, int C, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, int training, float* y, float* save_mean,
                    float* save_invstd, editor_stream_t stream);
int editor_bn1d_bwd(const float* dy, const float* x, long ldx, int B, int C, const float* gamma, const float* save_mean,
                    const float* save_invstd, float* dx, float* dgamma, float* dbeta, editor_stream_t stream);
/* OCFR.forward for ONE modality (OCFR.py:44-84): L2-normalise feat rows (row stride ldf), per-label batch mean,
 * centers[label] = momentum*mean + (1-momentum)*centers[label] in place, loss (+)= MSE(centers[label_b], fnorm_b).
 * fnorm (B,D), inv_norm (B): saved for backward.  workspace: B floats. */
int editor_ocfr_fwd(const float* feat, long ldf, const long* label, int B, int D, int C, float* centers, float momentum,
                    float* fnorm, float* inv_norm, float* workspace, float* loss, int accumulate, editor_stream_t stream);
int editor_ocfr_bwd(const float* fnorm, const float* inv_norm, const float* centers, const long* label,
                    const float* dloss, int B, int D, float* dfeat, editor_stream_t stream);

/* ---- loss head (layers/make_loss.py:36-56 consumes the hot path's train-mode outputs) ------- */

/* CenterLoss.forward (layers/center_loss.py:30-51): loss = sum over the (B, C) matrix [k == label_i] (|x_i|^2 + |c_k|^2 - 2 x_i.c_k), every
 * entry clamped to [1e-12, 1e12], / B.  x (B,D), centers (C,D) fp32, label (B) int64; dist (B) receives the unclamped own-class
 * distances (saved for the backward's clamp gate), row (B) is scratch.  Backward: dx (B,D) and / or dcenters (C,D) (either may be NULL). */
int editor_center_loss_fwd(const float* x, const float* centers, const long* label, int B, int C, int D, float* dist, float* row,
                           float* loss, editor_stream_t stream);
int editor_center_loss_bwd(const float* x, const float* centers, const long* label, const float* dist, const float* dloss, int B, int C,
                           int D, float* dx, float* dcenters, editor_stream_t stream);
/* CrossEntropyLabelSmooth(eps).forward (layers/softmax_loss.py:21-34): loss (+)= mean_b( -sum_c soft_bc log_softmax_bc ),
 * soft = (1-eps) onehot + eps/C.  row_loss: B floats of scratch.  bwd: dlogits = dloss[0]/B * (softmax - soft). */
int editor_ce_smooth_fwd(const float* logits, const long* target, int B, int C, float eps, float* row_loss, float* loss,
                         int accumulate, editor_stream_t stream);
int editor_ce_smooth_bwd(const float* logits, const long* target, int B, int C, float eps, const float* dloss,
                         float* dlogits, editor_stream_t stream);
/* TripletLoss() without margin (layers/triplet_loss.py:16-33,51-84,121-136): Euclidean distances with the 1e-12
 * clamp, batch-hard positive / negative per anchor, loss (+)= mean softplus(d_ap - d_an).  feat rows have stride ldf.
 * Scratch / saved for backward: gram (9,B,B: the Gram matrix + 8 reduction-chunk slabs), sq (B), idx (2B: positive,
 * negative), coef (3B), row_loss (B). */
int editor_triplet_fwd(const float* feat, long ldf, const long* label, int B, int D, float* gram, float* sq, int* idx,
                       float* coef, float* row_loss, float* loss, int accumulate, editor_stream_t stream);
int editor_triplet_bwd(const float* feat, long ldf, int B, int D, const int* idx, const float* coef, const float* dloss,
                       float* dfeat, editor_stream_t stream);

/* ---- retrieval evaluation (utils/metrics.py consumes the hot path's eval-mode features) ------ */

/* F.normalize(x, p=2, dim=1): y = x / max(||x||, eps) (utils/metrics.py:259-261). */
int editor_l2norm_rows(const float* x, long ldx, long M, int D, float eps, float* y, editor_stream_t stream);
/* euclidean_distance (utils/metrics.py:12-18): dist (Q,G) = qq + gg^T - 2 qf gf^T, fp32.  qq (Q), gg (G): scratch. */
int editor_distmat_f32(const float* qf, long ldq, const float* gf, long ldg, int Q, int G, int D, float* qq, float* gg,
                       float* dist, editor_stream_t stream);
/* np.argsort(distmat, axis=1) (utils/metrics.py:143): order (Q,G) int32, ascending distance, exactly tied distances by
 * ascending gallery index.  P = power of two >= G; keys: Q*P 64-bit words of scratch. */
int editor_rank_sort(const float* dist, int Q, int G, int P, unsigned long long* keys, int* order, editor_stream_t stream);
/* eval_func / eval_func_msrv (utils/metrics.py:151-183, 66-123): per query, drop the gallery entries that share the
 * query's pid AND aux id (aux = camera ids, or scene ids for the MSVR310 protocol); ap (Q) = average precision,
 * first_pos (Q) = 0-based rank of the first match among the kept entries (-1: query has no match, skipped);
 * totals[0] = sum of ap over valid queries, totals[1] = number of valid queries, cmc_counts[r], r < max_rank <= 1024 =
 * number of valid queries matched within rank r.  The caller divides (cmc = counts / valid, mAP = sum / valid). */
int editor_rank_metrics(const int* order, const long* q_pids, const long* g_pids, const long* q_aux, const long* g_aux,
                        int Q, int G, int max_rank, double* ap, int* first_pos, double* totals, int* cmc_counts,
                        editor_stream_t stream);

/* k-reciprocal re-ranking (utils/metrics.py:275-278 -> utils/reranking.py:30-101), dense over ALL N = Q + G images; the caller
 * (editor_amd.metrics.re_ranking) strings the stages together: editor_distmat_f32(feat, feat) -> _normalise -> editor_rank_sort(od)
 * -> _weights -> _expand (k2 > 1) -> _final.  V matrices are N x N IEEE-half bit patterns, exactly as the reference stores them.
 *   _normalise  reranking.py:37-47   colmax[i] = max_k dist[k,i]; od[i,j] = dist[j,i] / colmax[i]
 *   _weights    reranking.py:51-72   V (zero-filled by the call) <- exp(-od[i, R*(i,k1)]) / sum over the expanded k-reciprocal set;
 *                                    k1 + 1 <= 64 <= N, k1_half = int(np.around(k1 / 2)) (the caller rounds as numpy does) + 1 <= 32
 *   _expand     reranking.py:74-79   Vq[i,:] = half(mean over r < k2 of V[rank[i,r],:]) (fp32 accumulation in rank order)
 *   _final      reranking.py:81-101  Vt = scratch for V^T; final_dist (Q, N-Q) fp32 = half(jaccard * half(1 - lambda)) + od * lambda,
 *                                    the Jaccard sums in half arithmetic over ascending columns; one_minus_lambda_f16_bits = the bit
 *                                    pattern of np.float16(1 - lambda), lambda = np.float32(lambda) */
int editor_rerank_normalise(const float* dist, int N, float* colmax, float* od, editor_stream_t stream);
int editor_rerank_weights(const float* od, const int* rank, int N, int k1, int k1_half, uint16_t* V, editor_stream_t stream);
int editor_rerank_expand(const uint16_t* V, const int* rank, int N, int k2, uint16_t* Vq, editor_stream_t stream);
int editor_rerank_final(const uint16_t* V, uint16_t* Vt, const float* od, int N, int Q, int one_minus_lambda_f16_bits,
                        float lambda, float* final_dist, editor_stream_t stream);

/* ---- input transform on device (data/datasets/make_dataloader.py:245-253, 55-146) ------------ */

/* RandomHorizontalFlip -> Pad(pad, 0) -> RandomCrop(H,W) -> ToTensor -> Normalize(mean,std) -> RandomErasing('pixel')
 * of a batch of decoded + resized images.  in: uint8 (B,H,W,3); params: int32 (B,8) = {flip, crop_top, crop_left,
 * erase, e_top, e_left, e_h, e_w} drawn on the host in the reference's order; mean / stdv: HOST pointers to 3 floats;
 * noise: fp32 (B,3,H,W) N(0,1) fill for erased pixels, or NULL to generate it on the device from `seed`;
 * out: fp32 (B,3,H,W). */
int editor_augment_u8(const uint8_t* in, const int* params, int B, int H, int W, int pad, const float* mean,
                      const float* stdv, const float* noise, unsigned long long seed, float* out,
                      editor_stream_t stream);

/* ---- JPEG decode (data/datasets/bases.py:9-41 `Image.open(path).convert('RGB')` + the 256-wide
 * crops of the stitched tri-modal image) -------------------------------------------------------------------------------
 * Split: the HOST parses the markers and Huffman-decodes the scan(s) into quantised DCT coefficient blocks (the only
 * inherently serial part); the DEVICE does dequantisation + inverse DCT + chroma upsampling + YCbCr -> RGB + the crop
 * split for a whole batch per call.  Integer for integer libjpeg's default decompression path (jidctint.c islow,
 * jdsample.c fancy upsampling, jdcolor.c) - the pixels equal Pillow's bit for bit (tests/golden/f14_decode.npz).
 * Supported: 8-bit baseline / extended-sequential AND progressive Huffman (SOF0 / SOF1 / SOF2; the progressive scans refine
 * the same coefficient planes: tests/golden/f15_decode_progressive.npz), 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, restart
 * intervals, interleaved or per-component scans.  A progressive file that does not deliver every coefficient at full
 * precision, a second frame header, an over-subscribed Huffman table, a scan set that misses a component ->
 * EDITOR_JPEG_CORRUPT.  Arithmetic / lossless / 12-bit / 4-component files -> EDITOR_JPEG_UNSUPPORTED (no silent fallback).
 * info (16 ints, host): {W, H, ncomp, hmax, vmax, mcus_x, mcus_y, ycc_transform, blocks_per_image, tq0, tq1, tq2, ...}. */
#define EDITOR_JPEG_CORRUPT 9001
#define EDITOR_JPEG_UNSUPPORTED 9002
/* HOST: headers only (geometry of the coefficient buffer) */
int editor_jpeg_parse(const uint8_t* data, long n, int* info);
/* HOST: coef (host, coef_blocks x 64 int16, natural order; component planes one after another, block grids padded to whole
 * MCUs) and qt (host, 3 x 64 uint16, natural order, per component) of one image */
int editor_jpeg_entropy_decode(const uint8_t* data, long n, int16_t* coef, long coef_blocks, uint16_t* qt, int* info);
/* HOST: bytes of the per-image sample-plane scratch editor_jpeg_reconstruct needs */
int editor_jpeg_planes_bytes(const int* info, long* bytes);
/* DEVICE: B images of ONE geometry (info, host pointer): coef (B x blocks_per_image x 64) and qt (B x 3 x 64) in device
 * memory -> out uint8 (ncrop, B, H, crop_w, 3), ncrop = W / crop_w (crop_w <= 0: one crop of the full width); planes:
 * B x planes_bytes scratch. */
int editor_jpeg_reconstruct(const int16_t* coef, const uint16_t* qt, const int* info, int B, uint8_t* planes, int crop_w,
                            uint8_t* out, editor_stream_t stream);

/* T.Resize(size, interpolation) of decoded uint8 images (make_dataloader.py:246,256; torchvision 0.14.1 ->
 * PIL.Image.resize = Pillow ImagingResample, 8-bit path): horizontal pass then vertical pass with 22-bit fixed-point taps.
 * in (B,Hin,Win,3) -> out (B,Hout,Wout,3); bounds: (n_out,2) int32 {window start, tap count}; k: (n_out, ksize) int32
 * taps (device arrays, built on the host: editor_amd.data.resize_coeffs); tmp: (B,Hin,Wout,3) bytes, needed when both
 * axes change.  Bit-exact with Pillow (integer arithmetic). */
int editor_resize_u8(const uint8_t* in, int B, int Hin, int Win, int Hout, int Wout, const int* xbounds, const int* xk,
                     int xksize, const int* ybounds, const int* yk, int yksize, uint8_t* tmp, uint8_t* out,
                     editor_stream_t stream);

/* ---- training-step kernels (drop-path RNG of vit_pytorch.py:52-69) ----------------------- */

/* torch.optim.SGD(momentum, weight_decay, dampening 0) over many tensors in one launch.  Pointer tables and per-tensor
 * lr / wd live in device memory; chunk c covers elements [chunk_off[c], +editor_sgd_chunk_elems()) of tensor chunk_tensor[c].
 * g_ptrs[t] == NULL skips tensor t.  first != 0: momentum buffers are initialised with the (decayed) gradient.
 * h_ptrs (optional table, entries may be NULL): 16-bit shadow of the updated tensor - the GEMM operand copy - written in
 * the same pass instead of by one cast launch per weight; shadow_dtype: 1 = bf16, 2 = f16.
 * Momentum buffers must start at zero: mu*0 + g' reproduces torch's first-step `buf = g'` exactly (no "first" flag that
 * a captured hipGraph would bake in). */
int editor_sgd_multi(float* const* p_ptrs, const float* const* g_ptrs, float* const* m_ptrs, const int* chunk_tensor,
                     const long* chunk_off, const long* numel, const float* lr, const float* wd, float momentum,
                     long nchunks, uint16_t* const* h_ptrs, int shadow_dtype,
                     int* nonfinite /* device flag, OR-ed with 1 when a gradient element is inf / nan; may be NULL */,
                     const float* inv_scale /* device scalar multiplied into every gradient (1 / loss scale); NULL = 1 */,
                     const int* skip /* device flag: nonzero -> the whole update is skipped (GradScaler.step); NULL = never */,
                     editor_stream_t stream);
/* torch.optim.AdamW over the same tables (solver/make_optimizer.py:23-24): m_ptrs = exp_avg, v_ptrs = exp_avg_sq (fp32, zero at start);
 * step: device scalar holding the step count t as a float, advanced by one by this call BEFORE the update (not when *skip is set);
 * betas / eps by value.  p *= 1 - lr wd; m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g^2; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps). */
int editor_adamw_multi(float* const* p_ptrs, const float* const* g_ptrs, float* const* m_ptrs, float* const* v_ptrs,
                       const int* chunk_tensor, const long* chunk_off, const long* numel, const float* lr, const float* wd,
                       double beta1, double beta2, float eps, float* step, long nchunks, uint16_t* const* h_ptrs, int shadow_dtype,
                       int* nonfinite, const float* inv_scale, const int* skip, editor_stream_t stream);
/* amp.GradScaler's overflow check (engine/processor.py:94-96 -> torch/amp/grad_scaler.py): found[0] |= 1 (and sticky[0] |= 1,
 * optional) when any gradient element of the tensors in the chunk tables is inf / nan.  Run BEFORE editor_sgd_multi with
 * skip = found. */
int editor_grad_check_multi(const float* const* g_ptrs, const int* chunk_tensor, const long* chunk_off, const long* numel,
                            long nchunks, int* found, int* sticky, editor_stream_t stream);
/* amp.GradScaler.update on device-resident state (hipGraph-replay safe): found -> scale *= backoff, tracker = 0; else
 * tracker += 1 and every `interval` clean steps scale *= growth.  Refreshes inv_scale = 1 / scale and clears found. */
int editor_scaler_update(float* scale, float* inv_scale, int* tracker, int* found, float growth, float backoff,
                         int interval, editor_stream_t stream);
/* split-precision weight operands of COMPUTE_DTYPE 'f16x2' for a table of fp32 tensors in one launch (same chunk tables as
 * editor_sgd_multi): hi[t] = half(p[t] * scale), lo[t] = half(p[t] * scale - hi[t]); hi_ptrs[t] == NULL skips tensor t. */
int editor_split_multi(const float* const* p_ptrs, uint16_t* const* hi_ptrs, uint16_t* const* lo_ptrs,
                       const int* chunk_tensor, const long* chunk_off, const long* numel, long nchunks, float scale,
                       editor_stream_t stream);
/* dst[t] (cols x rows) = transpose of src[t] (rows x cols), 16-bit elements, for a table of tensors in ONE launch: the
 * k-major copies W^T of the nn.Linear weights that the dgrad products read (both dims multiples of 64).  Tables are
 * device arrays; tile i is the 64x64 tile (tile_r[i], tile_c[i]) of tensor tile_tensor[i]. */
int editor_transpose_multi(const uint16_t* const* src, uint16_t* const* dst, const int* rows, const int* cols,
                           const int* tile_tensor, const int* tile_r, const int* tile_c, long ntiles, editor_stream_t stream);
/* per-row drop-path scales keep/keep_prob for L blocks x 2 branches x B samples, expanded over T tokens:
 * scales (L,2,B*T) fp32; rates (L) fp32 on device; counter-based RNG keyed by `seed`. */
int editor_droppath_scales(const float* rates, int L, long B, int T, long seed, float* scales, editor_stream_t stream);
/* the same with the seed in DEVICE memory: state[0] keys this draw and is advanced by one afterwards, so a captured
 * hipGraph of the training step draws fresh masks on every replay */
int editor_droppath_scales_dev(const float* rates, int L, long B, int T, long* state, float* scales, editor_stream_t stream);

/* ---- stochastic-depth compaction (round 6) --------------------------------------------------------------------------------
 * The reference evaluates every residual branch and multiplies dropped samples by 0 (vit_pytorch.py:66-68, two draws per block,
 * :217-218).  A dropped (sample, block, branch) contributes x + 0 and receives no gradient through the branch, so the MLP branch
 * (LayerNorm-2, fc1, GELU, fc2 and their backward) runs on the LIVE samples' token rows only: same bits for every live row,
 * exact x for the dropped ones.
 * editor_droppath_plan: from scales (L,2,B*T) (editor_droppath_scales*; scale == 0 <=> dropped) -> for each of the L*2 (block,
 * branch) units  perm (L,2,B*T) int32: slot of token row r in the compacted order (live samples first, order kept; dropped
 * samples behind them), inv (L,2,B*T): token row of slot c, live (L,2): live ROWS = live samples * T. */
int editor_droppath_plan(const float* scales, int L, long B, int T, int* perm, int* inv, int* live, editor_stream_t stream);
/* editor_layernorm_fwd (16-bit y, D % 256 == 0) writing row r's output to row perm[r] of y; a dropped row (rowscale[r] == 0) writes
 * zeros there and copies its x row to copy_out (the block output of a dropped sample is its input).  mean / rstd: original rows. */
int editor_layernorm_fwd_perm(const float* x, const float* gamma, const float* beta, float eps, long M, int D, void* y, int y_bf16,
                              float* mean, float* rstd, const int* perm, const float* rowscale, float* copy_out,
                              editor_stream_t stream);
/* editor_layernorm_bwd_cast_parts with dy on compacted rows (dy_perm: row -> slot, dy_live: device scalar - slots >= *dy_live are
 * dropped rows whose gradient is zero and is not read; both NULL: dense dy) and / or the cast output written to the compacted rows
 * of its consumer branch (cast_perm; NULL: dense). */
int editor_layernorm_bwd_cast_perm_parts(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma,
                                         const float* mean, const float* rstd, long M, int D, const float* dx_in, float* dx_out,
                                         float* workspace, int ws_rows, void* cast_out, const float* cast_rowscale,
                                         float cast_scale, int want_colsum, const int* dy_perm, const int* dy_live,
                                         const int* cast_perm, int* nparts, editor_stream_t stream);
/* editor_cast_rows_colsum_parts writing row r to row perm[r] of out */
int editor_cast_rows_colsum_perm_parts(const float* in, const float* rowscale, long M, int D, void* out, int out_bf16,
                                       float* workspace, int ws_rows, float scale, const int* perm, int* nparts,
                                       editor_stream_t stream);
/* editor_gemm_bf16 / _f16 (dtype 1 / 2; A (M,K), B (N,K) k-major, fp32 C, EDITOR_EPI_RESIDUAL) on compacted rows: output row m is
 * scattered to row rowmap[m] of C, reading aux and rowscale there (rowmap = inv of editor_droppath_plan); m_live as editor_gemm_bf16. */
int editor_gemm_h16_rows(int dtype, const uint16_t* A, const uint16_t* B, void* C, int M, int N, int K, long lda, long ldb, long ldc,
                         float alpha, const float* bias, const float* rowscale, int epilogue, void* aux, long ldaux,
                         const int* m_live, const int* rowmap, editor_stream_t stream);
/* the split-precision ('f16x2') forms of the two: LayerNorm onto compacted rows as the half pair (y_hi, y_lo); editor_gemm_f16x2
 * (fp32 C, EDITOR_EPI_RESIDUAL) with the output row map */
int editor_layernorm_fwd_perm_f16x2(const float* x, const float* gamma, const float* beta, float eps, long M, int D, uint16_t* y_hi,
                                    uint16_t* y_lo, float* mean, float* rstd, const int* perm, const float* rowscale,
                                    float* copy_out, editor_stream_t stream);
int editor_gemm_f16x2_rows(const uint16_t* A_hi, const uint16_t* A_lo, const uint16_t* B_hi, const uint16_t* B_lo, void* C, int M,
                           int N, int K, long lda, long ldb, long ldc, float alpha, const float* bias, const float* rowscale,
                           int epilogue, void* aux, long ldaux, const int* m_live, const int* rowmap, editor_stream_t stream);
/* editor_gemm_wgrad_group with one live-row count PER problem (host array of `count` device scalars; NULL entry = all M rows) */
int editor_gemm_wgrad_group_live(int dtype, int count, const uint16_t* const* dy, const uint16_t* const* x, float* const* dw,
                                 const int* N, const int* K, int M, float alpha, int splitk, float* ws,
                                 const int* const* m_live, editor_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
