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

#ifdef __cplusplus
}
#endif
#endif /* GLOWTTS_HIP_H */
