// Cross-translation-unit helpers that are NOT part of the C-ABI (include/glowtts_hip.h).
#pragma once
#include <stdint.h>

// MFMA attention forward for the configuration every reference config uses (D = 96, window 4) and
// T <= 256; returns 1 when the shape is not handled (caller falls back to the generic kernel).
int gt_attn_fwd_mfma_impl(const void* q, const void* k, const void* v, int ld, const float* Ek, const float* Ev,
                          const int32_t* lens, void* out, int ldo, float* P, int B, int T, int Tp, const int32_t* row0, int H, int Dh, int win,
                          uint32_t drop_thresh, uint32_t drop_seed, float drop_scale, const uint32_t* seed_dev, void* stream);

// MFMA attention backward (same shape limits).  ws: gt_attn_bwd_mfma_ws_bytes(B,T,H) bytes of scratch.
#include <stddef.h>
size_t gt_attn_bwd_mfma_ws_bytes(int B, int T, int H);
int gt_attn_bwd_mfma_impl(const void* q, const void* k, const void* v, int ld, const float* Ek, const float* Ev,
                          const int32_t* lens, const void* dout, int lddo, const float* P, void* ws, size_t ws_bytes,
                          void* dq, void* dk, void* dv, int lddq, float* dEk, float* dEv,
                          int B, int T, int Tp, const int32_t* row0, int H, int Dh, int win, uint32_t drop_thresh, uint32_t drop_seed, float drop_scale,
                          const uint32_t* seed_dev, void* stream);
