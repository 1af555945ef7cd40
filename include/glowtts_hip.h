/*
 * glowtts_hip.h — C-ABI of the MI355X-native (gfx950) Glow-TTS training hot path.
 *
 * Drop-in boundary: every entry point takes plain device pointers + sizes and a HIP
 * stream (passed as void* so this header needs no HIP include), is asynchronous on
 * that stream, allocates nothing, keeps no global state and never throws.
 * Return value: 0 = ok, negative = error (GT_E_*).  The caller (a PyTorch-ROCm
 * shim, see glow-tts_amd/_lib.py, or any other host) owns all memory.
 *
 * Each function names the reference interface it replaces (paths are relative to the
 * reference repository arkiven4/glow-tts).
 */
#ifndef GLOWTTS_HIP_H
#define GLOWTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes */
#define GT_OK            0
#define GT_E_INVAL      -1   /* bad argument (null pointer, negative size, ...)        */
#define GT_E_UNSUPPORTED -2  /* shape outside what the kernel supports (see function)  */
#define GT_E_ALIGN      -3   /* pointer / stride alignment requirement violated        */
#define GT_E_LAUNCH     -4   /* hipLaunchKernel reported an error                      */

/* element types for outputs whose dtype follows the caller's tensor */
#define GT_DT_F32  0
#define GT_DT_I32  1
#define GT_DT_F16  2
#define GT_DT_BF16 3
#define GT_DT_U8   4

/* bits set in *status by gt_mas_f32 */
#define GT_MAS_ST_TX_GT_TY   1  /* some utterance had t_x > t_y (reference: silent OOB read,
                                   monotonic_align/core.pyx:34 with boundscheck off)          */
#define GT_MAS_ST_BAD_LEN    2  /* some utterance had t_x/t_y < 0 or beyond T_x/T_y           */

/* library identification: returns a static string "glowtts_hip <version> gfx950". */
const char* gt_version(void);

/* ------------------------------------------------------------------------------------
 * Monotonic Alignment Search.
 * Replaces: monotonic_align/core.pyx:38-45  maximum_path_c(paths, values, t_xs, t_ys)
 *           (+ :9-35 maximum_path_each) and the host wrapper
 *           monotonic_align/__init__.py:6-21 maximum_path(value, mask).
 *
 *   logp        [B, T_x, T_y] fp32 log-likelihood lattice, last dim contiguous,
 *               element (b,x,y) at logp[b*stride_b + x*stride_x + y].  NOT modified
 *               (the reference mutates its private copy).
 *   mask        optional (may be NULL) fp32 tensor with the same strides; when given the
 *               kernel uses logp*mask exactly like __init__.py:11.
 *   t_x, t_y    [B] int32 valid lengths per utterance (__init__.py:18-19).
 *   path        optional (may be NULL) [B, T_x, T_y] contiguous output of element type
 *               path_dtype (GT_DT_*): 1 on the alignment path, 0 elsewhere — every element
 *               is written (by a second, chip-wide kernel on the same stream).
 *   durations   optional [B, T_x] fp32: row sums of path (models.py:1085 `w`).
 *   frame2token optional [B, T_y] int32: for each frame y < t_y the row x with
 *               path[b,x,y]==1, else -1 (lets the prior expansion models.py:1118-1119
 *               be a gather instead of a matmul with a one-hot matrix).
 *   workspace   device scratch of at least gt_mas_workspace_bytes(B,T_x,T_y) bytes, 4-byte
 *               aligned; after the call it holds int32 [B, T_x+1] row start columns (row x of
 *               utterance b is aligned to frames [ws[b][x], ws[b][x+1]) ).
 *   status      optional device int32, OR-ed with GT_MAS_ST_* bits.  Utterances with
 *               invalid lengths get an all-zero path.
 *
 * Limits: T_x <= 512, and gt_mas_lds_bytes(T_x,T_y) <= 160 KiB, else GT_E_UNSUPPORTED.
 * Bit-exact with the reference for every t_x <= t_y (IEEE fp32, same tie-breaks).
 */
int gt_mas_f32(const float* logp, const float* mask,
               const int32_t* t_x, const int32_t* t_y,
               void* path, int path_dtype,
               float* durations, int32_t* frame2token,
               int B, int T_x, int T_y, int64_t stride_b, int64_t stride_x,
               void* workspace, size_t workspace_bytes,
               int32_t* status, void* stream);

/* Scratch bytes gt_mas_f32 needs for a batch (host helper). */
size_t gt_mas_workspace_bytes(int B, int T_x, int T_y);

/* LDS bytes one workgroup of gt_mas_f32 needs for a [T_x, T_y] lattice (host helper). */
size_t gt_mas_lds_bytes(int T_x, int T_y);

/* Lengths from a [B,T_x,T_y] fp32 mask the way monotonic_align/__init__.py:18-19 does:
 * t_x[b] = sum_x mask[b,x,0], t_y[b] = sum_y mask[b,0,y] (truncated to int32). */
int gt_mas_lengths_from_mask_f32(const float* mask, int32_t* t_x, int32_t* t_y,
                                 int B, int T_x, int T_y, int64_t stride_b, int64_t stride_x,
                                 void* stream);

/* ------------------------------------------------------------------------------------
 * "Rows" activation layout used by every kernel below: [R, C] channels-last (C contiguous),
 * uniform: R = B * Tp, utterance b owns rows [b*Tp, (b+1)*Tp), Tp = T + 2*HALO (HALO = 2); ragged: every entry point
 * that takes Tp also takes `row0` (device int32 [B+1], NULL = uniform): utterance b owns rows [row0[b], row0[b+1]) =
 * its own frames + 2*HALO (the last utterance also owns the rows that round R up), so frames past an utterance's
 * length cost no work, and Tp is only an upper bound on the rows of one utterance (grid sizing).  Frame t of
 * utterance b is row base(b) + HALO + t; halo rows and rows past the utterance length are zero
 * (rowmask[R] is 1 on valid frames, 0 elsewhere).  bf16 tensors are raw uint16 bit patterns.
 */
#define GT_HALO 2

/* 1-D convolution (odd k <= 5, dilation 1) as implicit GEMM on bf16 MFMA, fp32 accumulate.
 * Replaces the conv1d calls of modules.py:152,165 / attentions.py:144,172,232-238,365-371 /
 * modules.py:97 / models.py:710 and, with dgrad-packed weights, their data gradients.
 *   Y[m,n] = epi( sum_tap sum_ci X[m + tap - k/2, ci] * W[tap][n][ci] + bias[n] + cond[u(m)][n] )
 *   cond (may be NULL): [B, ldc] fp32 with u(m) = utterance of row m when B > 0 (WN.cond_layer(g), modules.py:148-156);
 *        [R, ldc] fp32 with u(m) = m when B == 0 (per-frame conditioning: WNP.cond_layer1(pitch), modules.py:316-334).
 *   epi: optional relu, optional dropout (drop_p, counter-based on (m,n)), optional + addend[m,n],
 *        optional * rowmask[m]; output bf16 or fp32.
 *   gate == 1 (WaveNet gate, commons.py:61-68; N = 2*half, packed with the gate interleave):
 *     pre = drop(acc + bias) + cond;  T = tanh(pre[:half]), S = sigmoid(pre[half:]),
 *     Y[m, :half] = T*S (bf16), T and S are saved to gate_t / gate_s ([R, ldts] bf16; ldts%8, ldy%8 == 0).
 *   gate == 2 (backward of that gate fused behind a data-gradient GEMM): d = acc + addend is d(T*S);
 *     gate_t / gate_s are the SAVED T / S (read-only); Y is [R, 2N] bf16: Y[m, n] = d*S*(1-T^2), Y[m, N+n] =
 *     d*T*S*(1-S), both times the forward's dropout mask (drop_p / drop_seed as given to the forward call).
 *   gate == 3 (backward of y = dropout(relu(.)), attentions.py:368-370 / modules.py:97-99, fused behind a data-gradient
 *     GEMM): gate_t is the SAVED y ([R, ldts] bf16, read-only; gate_s unused); Y[m, n] = (acc + addend) / (1 - drop_p)
 *     where y[m, n] != 0, else 0 (bf16; N%8, ldts%8, ldy%8 == 0).  No mask is drawn: drop_p only gives the scale.
 *   Wp: weights packed by gt_pack_conv_weights ([taps][Np][Kp] bf16, zero padded).
 * Dropout masks are a counter-based hash of (seed, row, col), replayed by the backward kernels.
 * seed_dev (here and in every entry point that takes it; may be NULL) is a device uint32 XOR-ed into the
 * host seed at kernel start: a captured HIP graph then draws fresh masks on every replay by bumping that word.
 * tile: 0 = the library's choice from (R, Np, gate); tests force a variant with GT_TILE_64x64 … GT_TILE_256x64
 * (rows x packed channels per workgroup; a variant the shape does not allow returns GT_E_INVAL).
 * Alignment: all pointers 16 B; N%4, Cin%8, ldx%8, ldy%4, Kp%64 == 0. */
#define GT_TILE_AUTO 0
#define GT_TILE_64x64 1
#define GT_TILE_64x128 2
#define GT_TILE_128x64 3
#define GT_TILE_128x128 4
#define GT_TILE_256x64 5
#define GT_TILE_64x64_TAPS 6   /* 64 x 64, all taps of a K slice per pipeline stage (short, deep convs; no gate) */
int gt_conv_gemm_bf16(const void* X, int ldx, const void* Wp, const float* bias,
                      const float* cond, int ldc, const float* rowmask,
                      void* Y, int ldy, int out_f32, const void* addend, int ldadd,
                      void* gate_t, void* gate_s, int ldts,
                      int R, int N, int Cin, int taps, int Tp, int Np, int Kp,
                      int relu, int gate, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev,
                      const int32_t* row0, int B, int tile, void* stream);

/* Weight preparation for gt_conv_gemm_bf16: w = g*v/||v|| when g != NULL (torch weight_norm,
 * dim 0: modules.py:127,132,141, attentions.py:103) else w = v; v is [Cout, Cin, taps] fp32.
 * Writes bf16 pack_fwd [taps][Np_fwd][Kp_fwd] and/or pack_dgrad [taps][Np_dgrad][Kp_dgrad]
 * (roles swapped, taps flipped) and inv_norm[Cout] = 1/||v|| (optional; pack pointers may both be NULL when only
 * the norms are wanted).  Padding entries are not touched: zero the buffers once.
 * `gate` is a bit set: 1 = WaveNet-gate row interleave of the forward image ([32 tanh | 32 sigmoid] per 64 rows),
 * 2 = forward image in MFMA-fragment order [tap][n/32][k/16][lane = n%32 + 32*((k%16)/8)][k%8] (one 1-KB MFMA A-fragment per (n/32, k/16): the fused WaveNet-layer kernels),
 * 4 = the same for the data-gradient image (Np, Kp multiples of 32 / 16 then),
 * 16 = the gate interleave at MFMA-block granularity, [16 tanh | 16 sigmoid] per 32 packed rows (instead of 1): every 32-column
 *     accumulator block then holds both halves of its 16 gate channels in the same lanes (gt_wn_layer_fwd's in-register gate),
 * 8 = bf16x3 ("split") images for a near-fp32 product out of three bf16 MFMA passes: the reduction axis is K-concatenated
 *     as [w_hi ; w_lo ; w_hi] (Kp >= 3*Cin resp. 3*Cout; w_hi = bf16(w), w_lo = bf16(w - w_hi)) and meets activations laid
 *     out as [x_hi | x_hi | x_lo] (gt_rows_split3): x_hi w_hi + x_hi w_lo + x_lo w_hi.  The stochastic predictors' 1x1 convs
 *     use it (their spline flows amplify bf16 rounding of the conditioning chaotically); not with 1, 2, 4;
 *     only through the one-row-per-workgroup packing (group8 == 0). */
int gt_pack_conv_weights(const float* v, const float* g, void* pack_fwd, void* pack_dgrad,
                         float* inv_norm, int Cout, int Cin, int taps,
                         int Np_fwd, int Kp_fwd, int Np_dgrad, int Kp_dgrad, int gate, void* stream);

/* The same for every conv of a model in ONE launch: `descs_device` is a device array of n_convs
 * descriptors sorted by row_start (row_start = prefix sum of Cout), total_rows = sum of Cout.  group8 != 0: every conv has
 * Cout % 8 == 0, Cin % 8 == 0 and Cin*taps <= 2304 — 8 output channels per workgroup, both images written with 16-byte
 * stores (else one workgroup per output channel). */
typedef struct gt_pack_desc {
  const float* v; const float* g; void* pack_fwd; void* pack_dgrad; float* inv_norm;
  int32_t Cout, Cin, taps, Np_fwd, Kp_fwd, Np_dgrad, Kp_dgrad, gate, row_start, pad_;
} gt_pack_desc;
/* group8: 0 one workgroup per output channel; 1 eight output channels per workgroup (every conv: Cout % 8 == 0, Cin % 8 == 0,
 * Cin * taps <= 2304); 2 the same for tables whose every conv has Cin * taps <= 1024 (less LDS and fewer registers per workgroup). */
int gt_pack_conv_weights_multi(const void* descs_device, int n_convs, int total_rows, int group8, void* stream);

/* Weight gradient of the rows-layout convolution: partial sums over S row slabs into
 * workspace [S][taps][Cout][Cin] fp32 (S from gt_conv_wgrad_workspace_bytes), bf16 MFMA with
 * transposing LDS reads.  dW[tap][co][ci] = sum_m dY[m,co] * X[m + tap - k/2, ci].
 * (ATen conv backward-weight in the reference, reached through autograd.) */
size_t gt_conv_wgrad_workspace_bytes(int R, int Cin, int Cout, int taps, int* slabs_out);
/* input channels one workgroup tile of the weight-gradient kernel spans at this tap count (64 at k = 3 and 5, 192 at k = 1): the ci0
 * granularity of gt_wgrad_tile below; 0 for an unsupported tap count. */
int gt_conv_wgrad_ci_tile(int taps);
int gt_conv_wgrad_bf16(const void* X, int ldx, const void* dY, int ldy, int R, int Cin, int Cout,
                       int taps, void* workspace, size_t workspace_bytes, void* stream);

/* Reduce the wgrad workspace and map it onto the parameter gradient(s) in the parameter's own
 * [Cout, Cin, taps] layout: plain conv (g == NULL): dv (+)= dW; weight-normed conv
 * (torch weight_norm dim 0): dg = <dW,v>/||v||, dv = g/||v|| (dW - v <dW,v>/||v||^2).
 * dbias (optional) receives the bias gradient: the wgrad kernel also leaves the column sums of dY
 * per slab in the workspace (summed from the MFMA A fragments it reads anyway). */
int gt_weightnorm_bwd(const void* workspace, int R, const float* v, const float* g, const float* inv_norm,
                      float* dv, float* dg, float* dbias, int Cout, int Cin, int taps, int accumulate, void* stream);

/* Batched forms: the weight gradients of a whole network in one launch per tap count, after the
 * data-gradient chain has run (every job's X and dY rows stay resident in HBM until then), and ONE
 * weight-norm backward over all of them.  Tables live in device memory.
 *   job:  dY column c is output channel co_begin + c (co_count columns) of a conv with Cout channels whose
 *         partials are part[S][taps][Cout][Cin] (+ part_bias[S][Cout], may be NULL); slab_rows % 64 == 0.
 *   tile: one workgroup = 128 dY columns from co0 x gt_conv_wgrad_ci_tile(taps) input channels from ci0 x all taps x rows of `slab`;
 *         tiles are ordered taps 5, then 3, then 1.
 *   wnb job: as gt_weightnorm_bwd, rows [row_start, row_start + Cout) of the launch. */
typedef struct gt_wgrad_job {
  const void* X; const void* dY; float* part; float* part_bias;
  int32_t ldx, ldy, R, Cin, Cout, co_begin, co_count, slab_rows;
} gt_wgrad_job;
typedef struct gt_wgrad_tile { int32_t job, co0, ci0, slab; } gt_wgrad_tile;
typedef struct gt_wnb_job {
  const float* part; const float* part_bias; const float* v; const float* g; const float* inv_norm;
  float* dv; float* dg; float* dbias;
  int32_t S, Cout, Cin, taps, row_start, accumulate, pad0_, pad1_;
} gt_wnb_job;
int gt_conv_wgrad_batched(const void* jobs_device, const void* tiles_device, int n_tiles5, int n_tiles3,
                          int n_tiles1, void* stream);
int gt_weightnorm_bwd_batched(const void* jobs_device, int n_jobs, int total_rows, int max_row_elems, void* stream);

/* out[n] += sum_m Y[m,n]  (bias gradients); Y bf16 (is_f32 == 0) or fp32 rows. */
int gt_colsum(const void* Y, int ldy, int is_f32, float* out, int R, int N, void* stream);

/* commons.squeeze / unsqueeze (commons.py:339-364) fused with the [B,C,T] <-> rows layout change.
 * len_sq[b] = valid squeezed frames; Tp = Ty/2 + 2*GT_HALO; C <= 80.  Each is the other's backward. */
int gt_squeeze_rows_f32(const float* y, float* rows, const int32_t* len_sq, int B, int C, int Ty, int Tp,
                        const int32_t* row0, void* stream);
int gt_unsqueeze_rows_f32(const float* rows, float* y, const int32_t* len_sq, int B, int C, int Ty, int Tp,
                          const int32_t* row0, void* stream);

/* ActNorm data-dependent initialisation (ActNorm.initialize, modules.py:607-619) on the rows layout: from the rows x
 * [R, C] fp32 the layer is about to see (rows outside an utterance are zero) and the valid frame counts len[B]:
 *   m = sum x / sum len, v = sum x^2 / sum len - m^2, logs = -0.5 log(max(v, 1e-6)), bias = -m * exp(logs).
 * workspace: 2*C doubles (cleared here).  One-off (first batch of a run with ddi=true, configs/base.json:13). */
int gt_actnorm_ddi(const float* x, const int32_t* len, int B, int R, int C, double* workspace, float* logs, float* bias,
                   void* stream);

/* ActNorm (modules.py:584-599) + InvConvNear (modules.py:635-665) fused, rows layout fp32 [R,C]:
 *   y = (W_4x4 applied per group {2g,2g+1,C/2+2g,C/2+2g+1} to (bias + exp(logs)*x)) * mask
 *   logdet[b] += (sum(logs) + (C/4)*logdet(W)) * len[b]          (when logdet != NULL)
 * gt_flow_scalars precomputes scal[18] = {sum logs, logdet W, W^-T}.  y0_bf16 (optional) receives
 * a bf16 copy of the first C/2 channels (the coupling's start-conv input). */
int gt_flow_scalars(const float* logs, int C, const float* W, float* scal, void* stream);
/* the same for n (ActNorm, InvConvNear) pairs in one launch: logs_ptrs / w_ptrs are DEVICE arrays of n device pointers,
 * scal is [n][18] */
int gt_flow_scalars_multi(const void* logs_ptrs, const void* w_ptrs, int C, float* scal, int n, void* stream);
int gt_actnorm_invconv_fwd(const float* x, float* y, void* y0_bf16, int ld0, const float* logs, const float* bias,
                           const float* W, const float* scal, const float* rowmask, const int32_t* len,
                           float* logdet, int B, int R, int C, void* stream);
/* dlogs/dbias/dW are ACCUMULATED into (zero them first). */
int gt_actnorm_invconv_bwd(const float* x, const float* dy, float* dx, const float* logs, const float* bias,
                           const float* W, const float* scal, const float* rowmask, const int32_t* len,
                           const float* dlogdet, float* dlogs, float* dbias, float* dW, int B, int R, int C, void* stream);

/* Affine coupling (attentions.py:174-186): out = [m|logs] fp32 rows from the `end` conv.
 *   z = [x0 | (m + exp(logs)*x1)*mask],  logdet[b] += sum(logs*mask). */
int gt_coupling_fwd(const float* out, const float* x, float* z, const float* rowmask, float* logdet,
                    int B, int R, int C, int Tp, const int32_t* row0, int sigmoid_scale, void* stream);
int gt_coupling_bwd(const float* out, const float* x, const float* dz, const float* dlogdet, const float* rowmask,
                    float* dx, void* dout_bf16, int B, int R, int C, int Tp, const int32_t* row0, int sigmoid_scale,
                    void* stream);

/* WaveNet gate backward (commons.py:61-68): dpre [R,2*half] bf16 from d(acts), saved T and S;
 * dpre carries the replayed dropout mask, dpre_cond (optional) does not. */
int gt_gate_bwd(const void* dacts, int ldd, const void* T, const void* S, int ldts, void* dpre, int ldp, void* dpre_cond,
                int R, int half, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream);

/* backward of y = rowmask * dropout(relu(c)) given the saved y: dc = (y != 0) ? d/(1-p) : 0 (bf16 rows). */
int gt_relu_drop_bwd(const void* d, int ldd, const void* y, int ldy, void* dc, int ldc, int R, int n, float drop_p, void* stream);

/* small row helpers */
int gt_rows_add_bf16(float* dx, int ldx, const void* add, int lda, int R, int n, void* stream);
int gt_rows_f32_to_bf16(const float* in, int ldi, void* out, int ldo, const float* rowmask, int R, int n, void* stream);

/* Channel LayerNorm of the text encoder / predictors on rows (modules.py:26-44, eps 1e-4), with the
 * surrounding elementwise work fused (attentions.py:79-84, modules.py:95-102, models.py:598-607):
 *   s = a (fp32, optional) + dropout_in(y (bf16, optional));  n = LN(s)*gamma + beta;
 *   o = dropout_out(relu?(n)) * rowmask;  out_f32 / out_bf16 (either optional).  C <= 256.
 * Dropout masks are counter-based (seed) and replayed by the backward. dgamma/dbeta ACCUMULATE.
 * gt_layernorm_bwd: relu bit 0 = the forward's relu flag; bit 1 (value 2) = y is itself a ReLU's output (conv -> relu -> norm,
 * models.py:591-598): dy is zeroed where y == 0, so the ReLU's backward needs no launch of its own. */
int gt_layernorm_fwd(const float* a, const void* y, int ldy, const float* gamma, const float* beta, const float* rowmask,
                     float* out_f32, void* out_bf16, int ldo, float* mean, float* rstd, int R, int C, float eps,
                     float p_in, uint32_t seed_in, float p_out, uint32_t seed_out, int relu, const uint32_t* seed_dev, void* stream);
int gt_layernorm_bwd(const float* a, const void* y, int ldy, const float* gamma, const float* beta, const float* rowmask,
                     const float* mean, const float* rstd, int R, int C, float eps,
                     float p_in, uint32_t seed_in, float p_out, uint32_t seed_out, int relu, const uint32_t* seed_dev,
                     const float* dout_f32, const void* dout_bf16, int lddo,
                     float* da, void* dy, int lddy, float* dgamma, float* dbeta, void* stream);
/* The same backward without atomics: one row per wave, the dgamma | dbeta sums of workgroup w go to row w of partials
 * [gt_layernorm_bwd_partial_rows(R)][2 C]; gt_param_partials_reduce then ADDS the column sums of up to GT_PARTIALS_MAX such
 * buffers [n_rows][Ca + Cb] to their destinations dst_a[Ca] / dst_b[Cb] in one launch (at the end of a module's backward). */
int gt_layernorm_bwd_partial_rows(int R);
int gt_layernorm_bwd_partials(const float* a, const void* y, int ldy, const float* gamma, const float* beta, const float* rowmask,
                              const float* mean, const float* rstd, int R, int C, float eps,
                              float p_in, uint32_t seed_in, float p_out, uint32_t seed_out, int relu, const uint32_t* seed_dev,
                              const float* dout_f32, const void* dout_bf16, int lddo,
                              float* da, void* dy, int lddy, float* partials, void* stream);
#define GT_PARTIALS_MAX 32
typedef struct gt_partials_job { const float* partials; float* dst_a; float* dst_b; int32_t n_rows, Ca, Cb, pad_; } gt_partials_job;
typedef struct gt_partials_args { gt_partials_job job[GT_PARTIALS_MAX]; int32_t n_jobs; } gt_partials_args;
int gt_param_partials_reduce(const gt_partials_args* args, void* stream);

/* Relative-position multi-head self-attention (attentions.py:241-336) in its banded form:
 *   score[i,j] = (q_i.k_j + [|j-i|<=win] q_i.Ek[j-i+win]) / sqrt(D), masked keys/queries -> -1e4,
 *   out_i = sum_j dropout(softmax)[i,j] v_j + sum_{|j-i|<=win} dropout(softmax)[i,j] Ev[j-i+win].
 * q,k,v,out: bf16 rows [B*Tp, H*D]; Ek,Ev: [2*win+1, D] fp32 shared by heads; P: [B,H,T,T] fp32
 * (softmax before dropout, kept for the backward); workspace: gt_attn_bwd_workspace_bytes(B,T,H) bytes of
 * scratch, 16-byte aligned; dEk/dEv ACCUMULATE.  D = 96, win = 4, T <= 256 run on bf16 MFMA. */
int gt_attn_fwd(const void* q, const void* k, const void* v, int ld, const float* Ek, const float* Ev,
                const int32_t* lens, void* out, int ldo, float* P, int B, int T, int Tp, const int32_t* row0, int H, int D, int win,
                float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream);
size_t gt_attn_bwd_workspace_bytes(int B, int T, int H);
int gt_attn_bwd(const void* q, const void* k, const void* v, int ld, const float* Ek, const float* Ev,
                const int32_t* lens, const void* dout, int lddo, const float* P, void* workspace, size_t workspace_bytes,
                void* dq, void* dk, void* dv, int lddq, float* dEk, float* dEv,
                int B, int T, int Tp, const int32_t* row0, int H, int D, int win, float drop_p, uint32_t drop_seed,
                const uint32_t* seed_dev, void* stream);

/* Embedding * scale into rows (models.py:693): fp32 and/or bf16 output, zero halo / padded rows.  emb [n_vocab, C];
 * the rows have stride ld >= C (ld > C: the language embedding of models.py:698-699 fills channels [C, ld) — written by
 * gt_rows_add_cond on that column window); the backward ACCUMULATES into demb. */
int gt_embedding_fwd(const int64_t* ids, const float* emb, const int32_t* lens, float* out_f32, void* out_bf16,
                     int B, int T, int Tp, const int32_t* row0, int R, int C, int ld, float scale, void* stream);
int gt_embedding_bwd(const int64_t* ids, const float* dx, const int32_t* lens, float* demb,
                     int B, int T, int Tp, const int32_t* row0, int R, int C, int ld, float scale, void* stream);

/* Ragged rows layout (see GT_HALO above): from the row offsets row0[B+1] and the lengths, fill the per-row tables the
 * host side keeps next to them — rowbatch[m] (utterance of row m, int64), rowframe[m] (m - row0[b] - GT_HALO) and
 * rowmask[m] (1 on valid frames), rowutt[m] (optional int32 twin of rowbatch) — in one launch (a replayed HIP graph's contexts
 * are refreshed per batch). */
int gt_rows_ctx_fill(const int32_t* row0, const int32_t* lens, int64_t* rowbatch, int32_t* rowframe, float* rowmask, int32_t* rowutt,
                     int B, int R, void* stream);

/* Everything a replayed training step takes from its batch, in ONE launch (train.Trainer's graph replay; the reference's loop does
 * `x.cuda(rank, non_blocking=True)` per tensor, train_ms_emo_lang_pitch.py:286-300): up to GT_STEP_MAX_COPIES padded copies — rows of
 * src_words 4-byte words into rows of dst_words >= src_words words, the tail of every row zeroed — and up to GT_STEP_MAX_CTX ragged row
 * contexts: geo_src = row0[B+1] | lengths[B] as int32 (device or PINNED HOST memory: read directly by the kernel), copied to geo_dst
 * and expanded into gt_rows_ctx_fill's tables.  blk0 is filled in by the call. */
#define GT_STEP_MAX_COPIES 10
#define GT_STEP_MAX_CTX 3
#define GT_STEP_MAX_B 1024
typedef struct gt_step_copy { const void* src; void* dst; int32_t rows, src_words, dst_words, blk0; } gt_step_copy;
typedef struct gt_step_ctx {
  const int32_t* geo_src; int32_t* geo_dst; int64_t* rowbatch; int32_t* rowframe; float* rowmask; int32_t* rowutt;
  int32_t B, R, blk0, pad_;
} gt_step_ctx;
typedef struct gt_step_inputs_args {
  gt_step_copy copy[GT_STEP_MAX_COPIES]; gt_step_ctx ctx[GT_STEP_MAX_CTX]; int32_t n_copy, n_ctx;
} gt_step_inputs_args;
int gt_step_inputs(const gt_step_inputs_args* args, void* stream);
/* The head of a training step: up to GT_ZERO_MAX regions zeroed (16-byte aligned, sizes multiples of 16) and *seed_word += seed_inc
 * (optional), one launch. */
#define GT_ZERO_MAX 4
typedef struct gt_step_zero_args { void* ptr[GT_ZERO_MAX]; uint64_t bytes[GT_ZERO_MAX]; uint32_t* seed_word; uint32_t seed_inc; int32_t n; } gt_step_zero_args;
int gt_step_zero(const gt_step_zero_args* args, void* stream);

/* out[m,:] = (x[m,:] + cond[utterance(m),:]) * rowmask[m]: a per-utterance vector added to every valid row — the
 * speaker conditioning the reference broadcasts over time in attentions.py:66-67 (Encoder.cond_g, before layer index 2)
 * and models.py:587-589 (DurationPredictor.cond).  Source fp32 `x` (wins) or bf16 `xb`; cond [B,C] fp32; fp32 and/or
 * bf16 output (either may be NULL, in place allowed); halo / padded rows are written as zero.  The gradient of cond is
 * the per-utterance row sum of the output gradient (gt_rows_utt_sum). */
int gt_rows_add_cond(const float* x, int ldx, const void* xb, int ldxb, const float* cond, const float* rowmask,
                     float* out, int ldo, void* outb, int ldob, int B, int R, int C, int Tp, const int32_t* row0, void* stream);

/* out[b, :C] (+)= sum over the rows m of utterance b of y[m, :C] * rowmask[m] (rowmask NULL = 1): the gradient of a
 * per-utterance vector broadcast over time — gt_rows_add_cond's cond, and the speaker conditioning g_l that WN adds
 * inside the gate (modules.py:148-156).  y bf16 (is_f32 == 0) or fp32 rows; out [B, ldo] fp32; accumulate != 0 adds to
 * out.  No atomics: one workgroup per (utterance, 64 channels). */
int gt_rows_utt_sum(const void* y, int ldy, int is_f32, const float* rowmask, float* out, int ldo, int accumulate,
                    int B, int R, int C, int Tp, const int32_t* row0, void* stream);

/* log-likelihood lattice (models.py:1076-1082) on exact-fp32 MFMA: x_m, x_logs (NULL = 0): [B,C,Tx],
 * z: [B,C,Ty] -> logp [B,Tx,Ty] fp32. */
int gt_logp_f32(const float* x_m, const float* x_logs, const float* z, float* logp, int B, int C, int Tx, int Ty, void* stream);

/* Prior expansion models.py:1118-1119 as a gather by frame2token (from gt_mas_f32; -1 = padded frame) and its backward as
 * the segment sums over the frames of every token (deterministic: fixed summation order; Tx <= 512). */
int gt_prior_expand(const float* x_m, const int32_t* frame2token, float* z_m, int B, int C, int Tx, int Ty, void* stream);
int gt_prior_expand_bwd(const float* dz_m, const int32_t* frame2token, float* dx_m, int B, int C, int Tx, int Ty, void* stream);

/* mle_loss pieces (commons.py:28-33): gt_mle_sums leaves GT_MLE_PARTS partial pairs (sum(logs), sum(exp(-2 logs)(z-m)^2)) in acc2 — one per
 * workgroup, plain stores, summed by gt_mle_finish;
 * backward: dz = g e^{-2 logs}(z-m), dm = -dz, dlogs = g (1 - e^{-2 logs}(z-m)^2), g = *gscale (/ *gdenom when gdenom != NULL:
 * the loss's denominator stays on the device); dlogdet (optional, [B]) receives -g. */
#define GT_MLE_PARTS 2048      /* acc2 holds this many (sum logs, sum exp(-2 logs)(z-m)^2) partial pairs: 2 * GT_MLE_PARTS floats, all written */
int gt_mle_sums(const float* z, const float* m, const float* logs, float* acc2, size_t n, void* stream);
int gt_mle_bwd(const float* z, const float* m, const float* logs, const float* gscale, float* dz, float* dm, float* dlogs,
               size_t n, const float* gdenom, float* dlogdet, int B, void* stream);
/* commons.sequence_mask as floats: mask[b, t] = t < lengths[b] (lengths int32 or int64), [B, T] in one launch. */
/* dev (bench.py --marks): one lane writes the device wall clock (100 MHz) to *slot — a time mark at this point of `stream`, valid
 * inside a replayed HIP graph, where the profiler's serialisation would hide which branch is the critical one. */
int gt_mark(unsigned long long* slot, void* stream);
int gt_length_mask(const void* lengths, int is_int64, float* mask, int B, int T, void* stream);

/* mle_loss's scalar tail (commons.py:31-33) in one launch: out2[0] = (acc2[0] + 0.5 acc2[1] - sum logdet) / denom + 0.5 log(2 pi),
 * out2[1] = denom = C * sum(mask) (acc2 from gt_mle_sums; mask: the n_mask floats of z_mask [B, 1, T]). */
int gt_mle_finish(const float* acc2, const float* logdet, const float* mask, int n_mask, int B, int C, float* out2, void* stream);
/* Duration loss of the deterministic predictor (models.py:1089-1092): l_length[b] = sum_t (logw[b,t] - log(w[b,t] + 1e-8) * mask)^2
 * / sum(mask), logw / w fp32 [B, Tx] (w = MAS durations), mask from x_lengths; backward: dlogw = g[b] * 2 (logw - logw_) / sum(mask). */
int gt_duration_loss_fwd(const float* logw, const float* w, const int32_t* x_lengths, int B, int Tx, float* l_length, void* stream);
int gt_duration_loss_bwd(const float* logw, const float* w, const int32_t* x_lengths, const float* g, int B, int Tx, float* dlogw, void* stream);

/* [B, C, T] <-> rows at the public boundary (fp32 or bf16 on either side: *_f32 = 1 / 0), one launch each:
 *   gt_rows_from_bct: rows[m, c] = x[b, c, t] for the frame rows of utterance b (row base(b) + HALO + t, 0 <= t < T), 0 elsewhere;
 *   gt_bct_from_rows: x[b, c, t] = rows[base(b) + HALO + t, c] for t < lengths[b], 0 beyond.  row0 == NULL: uniform layout. */
int gt_rows_from_bct(const void* x, int x_f32, void* rows, int rows_f32, const int32_t* row0, int B, int C, int T, int Tp, int R, void* stream);
int gt_bct_from_rows(const void* rows, int rows_f32, void* x, int x_f32, const int32_t* lengths, const int32_t* row0,
                     int B, int C, int T, int Tp, int R, void* stream);

/* Reverse (inference) direction of the flows — models.py:765-785 with reverse=True.
 *   gt_actnorm_invconv_rev: x = ((W^-1 y) * mask - bias) * exp(-logs) * mask  (InvConvNear^-1 then ActNorm^-1,
 *     modules.py:647-652,592-594); scal from gt_flow_scalars (holds W^-T); x0_bf16 (optional) = bf16(x[:, :C/2]).
 *   gt_coupling_rev: x = [z0 | (z1 - m) * exp(-logs) * mask] with [m | logs] = out (attentions.py:178-180). */
int gt_actnorm_invconv_rev(const float* y, float* x, void* x0_bf16, int ld0, const float* logs, const float* bias,
                           const float* scal, const float* rowmask, int R, int C, void* stream);
int gt_coupling_rev(const float* out, const float* z, float* x, const float* rowmask, int R, int C,
                    int sigmoid_scale, void* stream);

/* AdamW over flat fp32 buffers (parameters, gradients, both moments are slices of four buffers of n floats,
 * n % 4 == 0, 16-B aligned) — replaces torch.optim.AdamW.step + commons.clip_grad_value_(params, None)
 * (train_ms_emo_lang_pitch.py:311-312, commons.py:320-336).  hyper (device) = {lr, beta1, beta2, eps,
 * weight_decay, step} with step >= 1 already counting this update; *gnorm_sq (optional, device, caller zeroes
 * it) += sum g^2. */
int gt_adamw_flat(float* p, const float* g, float* m, float* v, size_t n, const float* hyper,
                  float* gnorm_sq, void* stream);

/* ---- A whole WaveNet (all gated layers of modules.WN.forward, modules.py:144-171, without the final skip sum) as ONE kernel
 * (csrc/wn_stack.hip): a workgroup recomputes the 2-row halo every k = 5 layer needs instead of exchanging it — 64 rows computed
 * per layer, gt_wn_stack_rows_per_workgroup(n_layers) = 64 - 4 (n_layers - 1) rows owned and stored.  Same arithmetic, dropout
 * hash and outputs as n_layers calls of gt_wn_layer_fwd (bit-identical).  H = 192, taps = 5, n_layers <= 4; weight images as
 * for gt_wn_layer_fwd; cond: [B, >= 2H n] per utterance (B > 0, row0 / Tp as elsewhere) or [R, >= 2H n] per row (B == 0),
 * layer i reads columns [2H i, 2H (i+1)); dropout seed of layer i is drop_seed + i (^ *seed_dev). */
typedef struct gt_wn_stack_fwd_args {
  const void* x0;                               /* bf16 [R, H]: the WaveNet's input (masked) */
  const void* w_in[4]; const float* b_in[4];    /* in_layer images (flags 2 | 4 | 16) and biases [2H] */
  const void* w_res[4]; const float* b_res[4];  /* residual 1x1 images and biases [H] of layers 0 .. n_layers-2 */
  const float* cond; int ldc; const int32_t* row0; int B; int Tp;
  const float* rowmask;
  void* acts; int ldacts;                       /* out: bf16 [R, >= n_layers * H], layer i in columns [H i, H (i+1)) */
  void* gate_t[4]; void* gate_s[4];             /* out: bf16 [R, H] per layer (saved tanh / sigmoid halves) */
  void* x_out[4];                               /* out: x_out[i] = input of layer i+1, bf16 [R, H], i < n_layers-1 */
  int R, H, taps, n_layers;
  float drop_p; uint32_t drop_seed; const uint32_t* seed_dev;
  unsigned long long* stamps; int stamp_slot; const int32_t* stamp_base;   /* as gt_wn_layer_fwd */
  /* optional, instead of cond: AFFINE per-frame conditioning of modules.WNP (modules.py:316-343, 353-362) formed in the kernel:
   * cond[m, 2H i + c] = aff_b[off + c] + aff_sig[m, par] * aff_w[off + c] with O = H n_layers, par = (2H i) / O, off = (2H i) % O
   * (n_layers even) — aff_w / aff_b: fp32 [O] (cond_layer1's weight-normed weight and bias), aff_sig: fp32 [R, 2] = the contour
   * at frames 2 m and 2 m + 1 of squeezed row m */
  const float* aff_w; const float* aff_b; const float* aff_sig;
} gt_wn_stack_fwd_args;
/* gradient of the affine conditioning's parameters from the d pre rows of the WaveNet's layers (dpre_c of gt_wn_stack_bwd, or dpre
 * where no dropout is applied): dw[off_i + c] += sum_m dpre_i[m, c] aff_sig[m, par_i], db[off_i + c] += sum_m dpre_i[m, c]
 * (par_i, off_i as above; fp32 [H n_layers] accumulators the caller zeroes; dpre_i: bf16 [R, lddp >= 2H]) */
int gt_cond_affine_grads(const void* dpre0, const void* dpre1, const void* dpre2, const void* dpre3, int lddp, const float* sig,
                         float* dw, float* db, int R, int H, int n_layers, void* stream);
int gt_wn_stack_rows_per_workgroup(int n_layers);
int gt_wn_stack_fwd(const gt_wn_stack_fwd_args* args, void* stream);
/* gt_wn_stack_bwd: the data-gradient chain of the same WaveNet in one launch (what n_layers - 1 calls of gt_wn_layer_bwd, the
 * bottom call without a second stage and gt_gate_bwd for the top layer compute; bit-identical):
 *   d pre_{n-1} = gate backward of via_skip[:, H (n-1) ..] (the top layer has no residual output);
 *   for j = n-1 .. 0:  dx[j] = (conv_k5^T(d pre_j; w_in_d[j]) + dx[j+1]) * rowmask;
 *                      j > 0:  d acts_{j-1} = dx[j] @ W_res_{j-1} (w_res_d[j-1]) + via_skip[:, H (j-1) ..];  d pre_{j-1} = gate backward.
 * dx[0] is the gradient at the WaveNet's input; dpre_c[i] (optional) = d pre_i before the dropout mask (gradient of the cond
 * term).  via_skip: bf16 [R, >= n H]; dpre / dpre_c: bf16 [R, 2H]; dx: bf16 [R, H]. */
typedef struct gt_wn_stack_bwd_args {
  const void* via_skip; int ldvs;
  const void* gate_t[4]; const void* gate_s[4];
  const void* w_in_d[4];                        /* data-gradient images of the in_layers (flag 4) */
  const void* w_res_d[4];                       /* data-gradient images of the residual 1x1s, layers 0 .. n_layers-2 */
  const float* rowmask;
  void* dpre[4]; void* dpre_c[4]; void* dx[4];
  int R, H, taps, n_layers;
  float drop_p; uint32_t drop_seed; const uint32_t* seed_dev;
} gt_wn_stack_bwd_args;
int gt_wn_stack_bwd(const gt_wn_stack_bwd_args* args, void* stream);

/* ---- Everything between two WaveNets of the flow decoder as ONE kernel (csrc/wn_boundary.hip): all of it is row-local.
 * Shapes: C = 160 flow channels (n_sqz * 80 mels), H = 192, n_layers = 4 (every reference config); 64 rows per workgroup.
 * Weight images: gt_pack_conv_weights(_multi) in MFMA-fragment order (flags 2 | 4), ks_* = padded K / 16 of each image.
 *
 * gt_wn_boundary_fwd — [tail of block b] when acts != NULL:
 *     wn_out = (acts @ Wskipcat^T + b_skip) * mask       modules.py:168-171 (K-concatenated skip GEMM, bf16 out, kept for wgrad)
 *     [m | logs] = wn_out @ Wend^T + b_end               attentions.py:162-165
 *     z = [y0 | (m + exp(logs) * y1) * mask], logdet[utt] += sum logs * mask      attentions.py:171-184; logs_raw [R, C/2] kept
 *   [head of block b+1] when y_next != NULL (input: the z tile, or x_in [R, C] when there is no tail):
 *     y_next = InvConvNear(ActNorm(z)) * mask, logdet[b] += (sum an_logs + C/4 * logdet W) * len[b]    modules.py:584-599, 635-665
 *     y0_bf16 = bf16(y_next[:, :C/2]);  h_next = (y0 @ Wstart^T + b_start) * mask                      attentions.py:147
 * gt_wn_boundary_bwd — [head of block b+1] when dh != NULL:
 *     d y = [dx_in[:, :C/2] + dh @ Wstart | dx_in[:, C/2:]];  ActNorm / InvConvNear backward against x (their input):
 *     d_an_logs, d_an_bias, d_w_ic accumulated with atomics (+ the log-det terms), d x -> the tile (or dx_out without a tail)
 *   [tail of block b] when dout != NULL (input: that tile, or dz_in [R, C] when there is no head):
 *     dx_out = [d z0 | d z1 * exp(logs) * mask];  dout = bf16 [d m | d logs] (kept for wgrad)
 *     dwn_out = (dout @ Wend) * mask (bf16, kept for wgrad);  via_skip [R, n_layers * H] = dwn_out @ Wskipcat
 * A NULL pointer selects the variant; everything is fp32 rows [R, C] unless said otherwise; rowutt int32 [R]. */
typedef struct gt_boundary_fwd_args {
  /* tail */
  const void* acts; int ldacts;                 /* bf16 [R, >= n_layers*H] gated activations of the block's WN */
  const void* w_skip; const float* b_skip;      /* forward image of the skip-cat GEMM [H, n_layers*H]; bias = sum of skip biases */
  const void* w_end; const float* b_end; int ks_end;
  const float* y;                               /* [R, C]: this block's ActNorm+InvConv output (y0 | y1) */
  void* wn_out;                                 /* out: bf16 [R, H] */
  float* logs_raw;                              /* out: [R, C/2] fp32 (before sigmoid_scale) */
  float* z;                                     /* out: [R, C] */
  float* logdet; const int32_t* rowutt; int sigmoid_scale;
  /* head */
  const float* x_in;                            /* head-only variant: the flow state [R, C] */
  const float* an_logs; const float* an_bias; const float* w_ic; const float* scal; const int32_t* len; int B;
  float* y_next; void* y0_bf16;                 /* out: [R, C] fp32, bf16 [R, C/2] */
  const void* w_start; const float* b_start; int ks_start;
  void* h_next;                                 /* out: bf16 [R, H] */
  const float* rowmask; int R, H, C, n_layers;
  /* optional: commons.squeeze / unsqueeze (commons.py:339-364) folded into the first / last launch of the pass —
   * y_bct (head-only variant, instead of x_in): the decoder's input [B, C/2, T] fp32 (z, if given, then receives the squeezed rows); z_bct (tail-only variant, instead of z): its
   * output [B, C/2, T] fp32, pre-zeroed by the caller; rowbatch (int64) / rowframe (int32): gt_rows_ctx_fill's tables; len as above */
  const float* y_bct; float* z_bct; int T; const int64_t* rowbatch; const int32_t* rowframe;
  /* optional: up to 16 buffers (16-byte aligned, pf_bytes[i] % 16 == 0, pf_ptr[i] == NULL ends the list) that 64 EXTRA workgroups of
   * the launch read once, on CUs the 64-row tiles leave idle — the weight images of the WaveNet launch that follows on the chain,
   * which would otherwise take their first touch (HBM) inside it */
  const void* pf_ptr[16]; uint32_t pf_bytes[16];
} gt_boundary_fwd_args;
typedef struct gt_boundary_bwd_args {
  /* head */
  const void* dh;                               /* bf16 [R, H]: masked gradient at the WN's input */
  const void* w_start_d; int ks_start_d;
  const float* dx_in;                           /* [R, C]: [d z0 | d y1] of this block (dx_out of the previous launch) */
  const float* x;                               /* [R, C]: input of this block's ActNorm */
  const float* an_logs; const float* an_bias; const float* w_ic; const float* scal; const int32_t* len; int B;
  float* d_an_logs; float* d_an_bias; float* d_w_ic;   /* accumulators [C], [C], [16] */
  /* tail */
  const float* dz_in;                           /* tail-only variant: gradient of the decoder's output rows [R, C] */
  const float* logs_raw; const float* y;        /* saved by the forward: [R, C/2], [R, C] */
  const float* dlogdet; const int32_t* rowutt; int sigmoid_scale;
  float* dx_out;                                /* out: [R, C] */
  void* dout;                                   /* out: bf16 [R, C] */
  const void* w_end_d; int ks_end_d;
  void* dwn_out;                                /* out: bf16 [R, H] */
  const void* w_skip_d; int ks_skip_d;
  void* via_skip; int ldvs;                     /* out: bf16 [R, >= n_layers*H] */
  const float* rowmask; int R, H, C, n_layers;
  /* optional, as in the forward: dz_bct (tail-only variant, instead of dz_in) and dx_bct (head-only variant, instead of dx_out;
   * pre-zeroed) are [B, C/2, T] fp32 */
  const float* dz_bct; float* dx_bct; int T; const int64_t* rowbatch; const int32_t* rowframe;
  /* optional (head): one row of gt_boundary_param_partials() floats per workgroup of the launch (ceil(R / 64) rows) — the
   * workgroup's sums for d_an_logs | d_an_bias | d_w_ic are STORED there instead of added to the three accumulators with atomics
   * (152 workgroups on the same 336 addresses serialise at L2: ~10 us of a 33 us launch); gt_boundary_param_reduce adds the rows up */
  float* pg_partial;
  const void* pf_ptr[16]; uint32_t pf_bytes[16];  /* optional prefetch list, as in the forward */
} gt_boundary_bwd_args;
int gt_wn_boundary_fwd(const gt_boundary_fwd_args* args, void* stream);
int gt_wn_boundary_bwd(const gt_boundary_bwd_args* args, void* stream);
/* floats per workgroup row of pg_partial, and the reduction over rows for n_blocks launches at once: partials
 * [n_blocks][n_wg][gt_boundary_param_partials()], dst = DEVICE array of 3 * n_blocks pointers {d_an_logs, d_an_bias, d_w_ic} per block;
 * every destination element gets += the sum over the block's n_wg rows (modules.py:584-599, 635-665 parameter gradients) */
int gt_boundary_param_partials(void);
int gt_boundary_param_reduce(const float* partials, int n_wg, int n_blocks, float* const* dst, void* stream);

/* ---- One WaveNet layer as ONE kernel (modules.WN.forward, one loop iteration, modules.py:151-170; csrc/wn_layer.hip).
 * H = 192 hidden channels, k = 5, dilation 1; a workgroup owns 64 rows and all channels, so the 1x1 residual conv runs on the
 * gated tile the k=5 conv just produced (and, in the backward, the gate backward on the tile the data gradient just produced).
 * Weights are images of gt_pack_conv_weights in MFMA-FRAGMENT order (flags 2 / 4) with Np == N, Kp == K exactly; the in_layer's
 * forward image also carries the block-granular gate interleave (flag 16).  A wave loads its own fragments L2 -> registers.
 *
 * gt_wn_layer_fwd:  x_in = drop(conv_k5(x) + bias_in) + cond;  T = tanh(x_in[:H]), S = sigmoid(x_in[H:]), acts = T*S
 *                   (acts / T / S bf16 rows out, as gt_conv_gemm_bf16 gate == 1 writes them; same dropout hash);
 *                   w_res_frag != NULL:  x_next = (x + acts @ W_res^T + bias_res) * rowmask     (rows [0,H) of res_skip_i)
 * gt_wn_layer_bwd:  dX = (conv_k5^T(dpre_next) + resid) * rowmask  -> dx [R, H]   (data gradient of the NEXT layer's in_layer,
 *                   resid = gradient arriving at that layer's output through the residual path, NULL for the top layer);
 *                   w_res_dgrad_frag != NULL:  d acts = dX @ W_res + via_skip;  dpre = gate backward of (T, S) with the forward's
 *                   dropout replayed, [R, 2H] = [d tanh-half | d sigmoid-half];  dpre_c (optional): the same before the dropout
 *                   mask (the gradient of the conditioning term, which is added after the dropout).  NULL: dx only (bottom layer).
 * stamps (optional, bench.py): device uint64 pairs [2*slot] = min start / [2*slot+1] = max end of the launch in
 * wall_clock64() ticks (100 MHz) — the kernel's duration inside a replayed HIP graph; init to ~0 / 0;
 * slot = stamp_slot + *stamp_base (stamp_base: optional device int32 the step bumps, so every replay fills fresh slots). */
int gt_wn_layer_fwd(const void* x, int ldx, const void* w_in_frag, const float* bias_in,
                    const float* cond, int ldc, const int32_t* row0, int B, int Tp, const float* rowmask,
                    void* acts, int ldacts, void* gate_t, void* gate_s, int ldts,
                    const void* w_res_frag, const float* bias_res, void* x_next, int ldxn,
                    int R, int H, int taps, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev,
                    unsigned long long* stamps, int stamp_slot, const int32_t* stamp_base, void* stream);
int gt_wn_layer_bwd(const void* dpre_next, int lddn, const void* w_in_dgrad_frag, const void* resid, int ldres,
                    const float* rowmask, void* dx, int lddx, const void* w_res_dgrad_frag,
                    const void* via_skip, int ldvs, const void* gate_t, const void* gate_s, int ldts,
                    void* dpre, void* dpre_c, int lddp, int R, int H, int taps, float drop_p, uint32_t drop_seed,
                    const uint32_t* seed_dev, unsigned long long* stamps, int stamp_slot, const int32_t* stamp_base, void* stream);

/* ---- Stochastic duration / pitch / energy predictors (models.py:217-481, modules.py:683-819,
 * transforms.py:12-202) on the rows layout.  C = 192 (filter_channels = in_channels, models.py:223).  `utt` is the
 * int32 utterance index of every row ([R]); rows with rowmask == 0 are zero on input and on output.  Gradient outputs
 * named d<param> ACCUMULATE (atomics); `acc` / `gacc` are per-utterance [B] accumulators of the negative log-likelihood
 * and their incoming gradient.
 *
 * DilatedDepthSeparableConv layer i (modules.py:726-734), dilation = kernel_size^i, kernel_size = 3:
 *   gt_dds_sep_fwd:  a1 = gelu(LayerNorm2(dwconv_d(x) + b))                -> bf16x3 rows [R, 3C], operand of the split 1x1 GEMM
 *   gt_dds_out_fwd:  out = (x + dropout(gelu(LayerNorm2(h2)))) * mask       h2 = 1x1 conv output incl. bias (fp32 rows)
 *   gt_dds_out_bwd:  dy -> d h2 (bf16x3 rows [R, 3C]), d gamma2 / d beta2
 *   gt_dds_sep_bwd:  d a1 (fp32, from the 1x1 data-gradient GEMM) -> d h1, d gamma1 / d beta1   (h1 recomputed from x)
 *   gt_dds_dw_bwd:   dx = (dy + dwconv_d^T(d h1)) * mask, d w [C,3], d b [C] */
/* bf16x3 operand layout for a split GEMM (gt_pack_conv_weights flag 8): out[m] = [hi | hi | lo] of in[m, :C]
 * (in fp32: hi = bf16(x), lo = bf16(x - hi); in bf16: lo = 0), out bf16 [R, ldo >= 3C], rows masked by rowmask if given.
 * gt_dds_sep_fwd's a1 and gt_dds_out_bwd's d h2 are written in this layout directly (lda / row pitch 3C). */
int gt_rows_split3(const void* in, int ldi, int is_f32, void* out, int ldo, const float* rowmask, int R, int C, void* stream);
int gt_dds_sep_fwd(const float* x, int ldx, const float* w, const float* b, const float* gamma, const float* beta,
                   const int32_t* utt, const float* rowmask, void* a1_bf16, int lda, int R, int C, int dilation, float eps, void* stream);
int gt_dds_out_fwd(const float* h2, const float* x, int ldx, const float* gamma, const float* beta, const float* rowmask,
                   float* out, void* out_bf16, int R, int C, float eps, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);
/* The three backward kernels accumulate their parameter gradients with one atomic per address and workgroup — or, with `partials`
 * (then the gradient pointers may be NULL), store the workgroup's sums to row w of partials [gt_dds_bwd_partial_rows(R)][W], W = 2 C
 * (gamma | beta) for out / sep, 4 C (dw[C][3] | db[C]) for dw; gt_param_partials_reduce adds the column sums later. */
int gt_dds_bwd_partial_rows(int R);
int gt_dds_out_bwd(const float* h2, const float* dy, const float* gamma, const float* beta, const float* rowmask,
                   void* dh2_bf16, float* dgamma, float* dbeta, float* partials, int R, int C, float eps, float drop_p, uint32_t seed,
                   const uint32_t* seed_dev, void* stream);
int gt_dds_sep_bwd(const float* x, int ldx, const float* w, const float* b, const float* gamma, const float* beta,
                   const int32_t* utt, const float* rowmask, const float* da1, float* dh1, float* dgamma, float* dbeta,
                   float* partials, int R, int C, int dilation, float eps, void* stream);
int gt_dds_dw_bwd(const float* x, int ldx, const float* dh1, const float* dy, const float* w, const int32_t* utt,
                  const float* rowmask, float* dx, float* dw, float* db, float* partials, int R, int C, int dilation, void* stream);

/* ConvFlow (modules.py:792-819), in_channels = 2, on z rows [R, 2]:
 *   gt_convflow_pre_fwd:    x0 = (w_pre * z[:,0] + b_pre + g1 (+ g2)) * mask          (pre + DDSConv's `x = x + g`)
 *   gt_convflow_pre_bwd:    dx0 -> d w_pre, d b_pre, dz[:,0] += sum_c dx0 w_pre, dg += dx0      (dz / dg may be NULL)
 *   gt_convflow_spline_fwd: params = (Wp h + bp) * mask ([R,32], 29 used); z_out = [z0, RQS(z1)] * mask, channels swapped when
 *                           flip (the torch.flip after every ConvFlow, models.py:317-318); acc[utt] += sign * log|det|
 *   gt_convflow_spline_bwd: dz_out, gacc -> dh, d Wp [29,C], d bp [29], dz_in
 *   gt_convflow_spline_inv: synthesis direction (transforms.py:152-180): z_out = [z0, RQS^-1(z1)] * mask
 * The spline is transforms.piecewise_rational_quadratic_transform with 10 bins, linear tails, tail_bound 5. */
int gt_convflow_pre_fwd(const float* z, int ldz, const float* w_pre, const float* b_pre, const float* g1, const float* g2,
                        const float* rowmask, float* out, int R, int C, void* stream);
int gt_convflow_pre_bwd(const float* dx0, const float* z, int ldz, const float* w_pre, const float* rowmask,
                        float* dw_pre, float* db_pre, float* dz, int lddz, float* dg, int R, int C, void* stream);
int gt_convflow_spline_fwd(const float* h, const float* Wp, const float* bp, const float* z_in, const float* rowmask,
                           const int32_t* utt, float* z_out, float* params, float* acc, float sign, int flip, int R, int C, void* stream);
/* partials (optional; then dWp / dbp may be NULL): row w of [gt_convflow_spline_partial_rows(R)][gt_convflow_spline_partial_width()]
 * receives workgroup w's d Wp [29][C] | d bp [29] sums instead of atomics (gt_param_partials_reduce adds them up). */
int gt_convflow_spline_partial_rows(int R);
int gt_convflow_spline_partial_width(void);
int gt_convflow_spline_bwd(const float* h, const float* Wp, const float* params, const float* z_in, const float* dz_out,
                           const float* gacc, const float* rowmask, const int32_t* utt, float* dh, float* dWp, float* dbp,
                           float* partials, float* dz_in, float sign, int flip, int R, int C, void* stream);
int gt_convflow_spline_inv(const float* h, const float* Wp, const float* bp, const float* z_in, const float* rowmask,
                           float* z_out, int R, int C, void* stream);

/* ElementwiseAffine (modules.py:750-756) on [R,2]: y = (x * exp(log_scale) + translation) * mask (reverse: the inverse),
 * acc[utt] += sign * (log_scale_0 + log_scale_1) per valid row (acc may be NULL). */
int gt_ea_fwd(const float* x, const float* log_scale, const float* translation, const float* rowmask, const int32_t* utt,
              float* y, float* acc, float sign, int reverse, int R, void* stream);
int gt_ea_bwd(const float* x, const float* log_scale, const float* dy, const float* gacc, const float* rowmask, const int32_t* utt,
              float* dx, float* dlog_scale, float* dtranslation, float sign, int R, void* stream);

/* StochasticDurationPredictor between its posterior flows and its flows (models.py:299-311): z_q = [z_u, z_v], durations w [R],
 * noise e_q [R,2] -> z = [log(max(w - sigmoid(z_u), 1e-5)), z_v];  acc[utt] += -0.5 (2 log 2pi + |e_q|^2)
 * - (logsigmoid(z_u) + logsigmoid(-z_u)) + z[:,0].   gt_nll_gauss_*: acc[utt] += 0.5 (2 log 2pi + |z|^2) (models.py:321, 395, 469). */
int gt_sdp_mid_fwd(const float* z_q, const float* w, const float* e_q, const float* rowmask, const int32_t* utt, float* z, float* acc,
                   int R, void* stream);
int gt_sdp_mid_bwd(const float* z_q, const float* w, const float* dz, const float* gacc, const float* rowmask, const int32_t* utt,
                   float* dz_q, int R, void* stream);
int gt_nll_gauss_fwd(const float* z, const float* rowmask, const int32_t* utt, float* acc, int R, void* stream);
int gt_nll_gauss_bwd(const float* z, const float* gacc, const float* rowmask, const int32_t* utt, float* dz, int R, void* stream);

/* x_feature = x @ attn (models.py:1094) for the hard MAS path: frame row m of utterance b takes the token row
 * frame2token[b, t] (bf16 rows, C % 8 == 0); *_x describe the text-side rows, *_f the frame-rate rows. */
int gt_rows_gather_tokens(const void* x_rows, int ldx, const int32_t* frame2token, int Ty, const int32_t* row0_x, int Tp_x,
                          const int32_t* utt_f, const int32_t* row0_f, int Tp_f, const float* rowmask_f, void* out, int R_f, int C,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GLOWTTS_HIP_H */
