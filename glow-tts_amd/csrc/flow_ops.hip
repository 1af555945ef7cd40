// HBM-bound pieces of the flow decoder for gfx950, in the rows layout (see glowtts_hip.h):
//   squeeze / unsqueeze          commons.py:339-364  (+ layout change [B,C,T] <-> rows)
//   ActNorm + InvConvNear fused  modules.py:584-599, 635-665 (forward, backward, log-dets)
//   affine coupling              attentions.py:174-186      (forward, backward)
//   WaveNet gate backward        commons.py:61-68 (autograd of tanh*sigmoid, dropout replay)
// All fp32 math; bf16 only where a tensor feeds an MFMA GEMM.
#include <stdlib.h>
#include "common.h"
#include "../../include/glowtts_hip.h"

namespace {

constexpr int HALO = GT_HALO;

// ------------------------------------------------------------------ squeeze: [B,C,Ty] -> rows [R,2C]
// rows[b*Tp + HALO + t', p*C + c] = y[b, c, 2t'+p] * (t' < len[b]);  halo / padded rows = 0.
__global__ __launch_bounds__(256) void gt_squeeze_rows_kernel(const float* __restrict__ y, float* __restrict__ rows,
                                                              const int32_t* __restrict__ len_sq, int B, int C, int Ty, int Tp,
                                                              const int32_t* __restrict__ row0)
{
  __shared__ float tile[32][2 * 80 + 1];          // [t'][p*C+c], C <= 80
  const int b = blockIdx.y, tp0 = blockIdx.x * 32;                 // tp0 indexes rows of the utterance (incl. halo)
  const int base = gt_row_base(row0, b, Tp), nrows = gt_row_count(row0, b, Tp);
  if (tp0 >= nrows) return;
  const int len = len_sq[b], T2 = Ty / 2;
  // load: for each channel, 64 consecutive frames (= 32 squeezed frames x 2 phases)
  for (int i = threadIdx.x; i < C * 64; i += 256) {
    const int c = i >> 6, f = i & 63;
    const int tq = tp0 - HALO + (f >> 1), p = f & 1;               // squeezed frame, phase
    float v = 0.f;
    if (tq >= 0 && tq < len && tq < T2 && 2 * tq + p < Ty) v = y[((size_t)b * C + c) * Ty + 2 * tq + p];
    tile[f >> 1][p * C + c] = v;
  }
  __syncthreads();
  const int C2 = 2 * C;
  for (int i = threadIdx.x; i < 32 * C2; i += 256) {
    const int rr = i / C2, ch = i - rr * C2;
    if (tp0 + rr < nrows) rows[((size_t)base + tp0 + rr) * C2 + ch] = tile[rr][ch];
  }
}

// unsqueeze: rows [R,2C] -> [B,C,Ty] with y[b,c,2t'+p] = rows[...]*mask; frames >= 2*len (and a
// trailing odd frame) are written as 0.  Also the backward of squeeze (and vice versa).
__global__ __launch_bounds__(256) void gt_unsqueeze_rows_kernel(const float* __restrict__ rows, float* __restrict__ y,
                                                                const int32_t* __restrict__ len_sq, int B, int C, int Ty, int Tp,
                                                                const int32_t* __restrict__ row0)
{
  __shared__ float tile[32][2 * 80 + 1];
  const int b = blockIdx.y, t0 = blockIdx.x * 32;                  // squeezed frame base (no halo offset)
  const int base = gt_row_base(row0, b, Tp);
  const int len = len_sq[b], C2 = 2 * C, T2 = Ty / 2;
  for (int i = threadIdx.x; i < 32 * C2; i += 256) {
    const int rr = i / C2, ch = i - rr * C2;
    const int tq = t0 + rr;
    tile[rr][ch] = (tq < len && tq < T2) ? rows[((size_t)base + HALO + tq) * C2 + ch] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 64; i += 256) {
    const int c = i >> 6, f = i & 63;
    const int t = 2 * t0 + f;
    if (t < Ty) y[((size_t)b * C + c) * Ty + t] = tile[f >> 1][(f & 1) * C + c];
  }
}

// ------------------------------------------------------------------ ActNorm + InvConvNear
// scal[0] = sum(logs), scal[1] = log det W (4x4), scal[2..17] = W^{-T} (row major).
__device__ __forceinline__ void flow_scalars_one(const float* __restrict__ logs, int C, const float* __restrict__ W, float* __restrict__ scal)
{
  float s = 0.f;
  for (int i = threadIdx.x; i < C; i += 64) s += logs[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    scal[0] = s;
    double m[4][4], inv[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { m[i][j] = W[i * 4 + j]; inv[i][j] = (i == j); }
    double det = 1.0;
    for (int c = 0; c < 4; ++c) {                                  // Gauss-Jordan with partial pivoting
      int piv = c; double best = fabs(m[c][c]);
      for (int r = c + 1; r < 4; ++r) if (fabs(m[r][c]) > best) { best = fabs(m[r][c]); piv = r; }
      if (piv != c) { for (int j = 0; j < 4; ++j) { double t = m[c][j]; m[c][j] = m[piv][j]; m[piv][j] = t;
                                                     t = inv[c][j]; inv[c][j] = inv[piv][j]; inv[piv][j] = t; } det = -det; }
      const double d = m[c][c]; det *= d;
      for (int j = 0; j < 4; ++j) { m[c][j] /= d; inv[c][j] /= d; }
      for (int r = 0; r < 4; ++r) if (r != c) { const double f = m[r][c];
        for (int j = 0; j < 4; ++j) { m[r][j] -= f * m[c][j]; inv[r][j] -= f * inv[c][j]; } }
    }
    scal[1] = (float)log(det);                                      // torch.logdet: nan if det < 0
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) scal[2 + i * 4 + j] = (float)inv[j][i];   // W^{-T}
  }
}
__global__ void gt_flow_scalars_kernel(const float* __restrict__ logs, int C, const float* __restrict__ W, float* __restrict__ scal)
{
  flow_scalars_one(logs, C, W, scal);
}
// all flow blocks of a decoder in one launch: block b reads logs_ptrs[b] / w_ptrs[b] (device pointer tables), writes scal[18 b ..]
__global__ void gt_flow_scalars_multi_kernel(const float* const* __restrict__ logs_ptrs, const float* const* __restrict__ w_ptrs, int C,
                                             float* __restrict__ scal)
{
  flow_scalars_one(logs_ptrs[blockIdx.x], C, w_ptrs[blockIdx.x], scal + 18 * blockIdx.x);
}

// thread = (row, group g): members {2g, 2g+1, C/2+2g, C/2+2g+1}.
__global__ __launch_bounds__(256) void gt_actnorm_invconv_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ y, bf16_t* __restrict__ y0_bf16, int ld0,
    const float* __restrict__ logs, const float* __restrict__ bias, const float* __restrict__ W,
    const float* __restrict__ rowmask, int R, int C,
    const float* __restrict__ scal, const int32_t* __restrict__ len, float* __restrict__ logdet, int B)
{
  const int G = C >> 2, half = C >> 1;
  // log-det bookkeeping of this flow pair rides along in workgroup 0 (it used to be a launch of its own on the
  // decoder's dependent chain): logdet[b] += (sum logs + G * logdet W) * len_b
  if (logdet && blockIdx.x == 0) {
    const float per_frame = scal[0] + (float)G * scal[1];
    for (int b = threadIdx.x; b < B; b += 256) logdet[b] += per_frame * (float)len[b];
  }
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * G) return;
  const int m = idx / G, g = idx - m * G;
  const float rm = rowmask[m];
  const float2 xa = *reinterpret_cast<const float2*>(x + (size_t)m * C + 2 * g);
  const float2 xb = *reinterpret_cast<const float2*>(x + (size_t)m * C + half + 2 * g);
  const float a0 = bias[2 * g] + __expf(logs[2 * g]) * xa.x, a1 = bias[2 * g + 1] + __expf(logs[2 * g + 1]) * xa.y;
  const float a2 = bias[half + 2 * g] + __expf(logs[half + 2 * g]) * xb.x, a3 = bias[half + 2 * g + 1] + __expf(logs[half + 2 * g + 1]) * xb.y;
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (W[k * 4] * a0 + W[k * 4 + 1] * a1 + W[k * 4 + 2] * a2 + W[k * 4 + 3] * a3) * rm;
  *reinterpret_cast<float2*>(y + (size_t)m * C + 2 * g) = make_float2(o[0], o[1]);
  *reinterpret_cast<float2*>(y + (size_t)m * C + half + 2 * g) = make_float2(o[2], o[3]);
  if (y0_bf16) *reinterpret_cast<uint32_t*>(y0_bf16 + (size_t)m * ld0 + 2 * g) = pack2bf(o[0], o[1]);
}

// reverse (inference): InvConvNear^-1 then ActNorm^-1 (modules.py:647-652, 592-594):
//   x = ((W^-1 y) * mask - bias) * exp(-logs) * mask;   W^-1[k][j] = scal[2 + 4 j + k]  (scal holds W^-T row major)
__global__ __launch_bounds__(256) void gt_actnorm_invconv_rev_kernel(
    const float* __restrict__ y, float* __restrict__ x, bf16_t* __restrict__ x0_bf16, int ld0,
    const float* __restrict__ logs, const float* __restrict__ bias, const float* __restrict__ scal,
    const float* __restrict__ rowmask, int R, int C)
{
  const int G = C >> 2, half = C >> 1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * G) return;
  const int m = idx / G, g = idx - m * G;
  const float rm = rowmask[m];
  const float2 ya = *reinterpret_cast<const float2*>(y + (size_t)m * C + 2 * g);
  const float2 yb = *reinterpret_cast<const float2*>(y + (size_t)m * C + half + 2 * g);
  const float yv[4] = {ya.x, ya.y, yb.x, yb.y};
  const int ch[4] = {2 * g, 2 * g + 1, half + 2 * g, half + 2 * g + 1};
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) a += scal[2 + 4 * j + k] * yv[j];
    o[k] = (a * rm - bias[ch[k]]) * __expf(-logs[ch[k]]) * rm;
  }
  *reinterpret_cast<float2*>(x + (size_t)m * C + 2 * g) = make_float2(o[0], o[1]);
  *reinterpret_cast<float2*>(x + (size_t)m * C + half + 2 * g) = make_float2(o[2], o[3]);
  if (x0_bf16) *reinterpret_cast<uint32_t*>(x0_bf16 + (size_t)m * ld0 + 2 * g) = pack2bf(o[0], o[1]);
}

// backward: dx = exp(logs) * W^T (dy*mask);  reductions into dlogs[C], dbias[C], dW[16] by atomics.
__global__ __launch_bounds__(256) void gt_actnorm_invconv_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
    const float* __restrict__ logs, const float* __restrict__ bias, const float* __restrict__ W,
    const float* __restrict__ rowmask, float* __restrict__ dlogs, float* __restrict__ dbias, float* __restrict__ dW,
    int R, int C, int rows_per_block,
    const float* __restrict__ scal, const int32_t* __restrict__ len, const float* __restrict__ dlogdet, int B)
{
  // block: 256 threads = 4 row phases x 64 "group lanes" (G = C/4 <= 64); loops over its row slab
  __shared__ float sW[4][16];
  // backward of the log-det bookkeeping, in workgroup 0 (formerly its own launch): with s = sum_b dlogdet[b] * len_b,
  // dlogs[c] += s and dW += (C/4) * s * W^{-T}; atomics, because every workgroup accumulates into dlogs / dW
  if (dlogdet && blockIdx.x == 0) {
    __shared__ float sred[4];
    float sv = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) sv += dlogdet[b] * (float)len[b];
    sv = wave_sum(sv);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = sv;
    __syncthreads();
    sv = sred[0] + sred[1] + sred[2] + sred[3];
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(dlogs + c, sv);
    if (threadIdx.x < 16) atomicAdd(dW + threadIdx.x, (float)(C >> 2) * sv * scal[2 + threadIdx.x]);
  }
  const int G = C >> 2, half = C >> 1;
  const int g = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(R, m0 + rows_per_block);
  float accW[16], accL[4] = {0, 0, 0, 0}, accB[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) accW[i] = 0.f;
  if (g < G) {
    const int ch[4] = {2 * g, 2 * g + 1, half + 2 * g, half + 2 * g + 1};
    float el[4], bs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { el[k] = __expf(logs[ch[k]]); bs[k] = bias[ch[k]]; }
    for (int m = m0 + ph; m < m1; m += 4) {
      const float rm = rowmask[m];
      const float2 xa = *reinterpret_cast<const float2*>(x + (size_t)m * C + 2 * g);
      const float2 xb = *reinterpret_cast<const float2*>(x + (size_t)m * C + half + 2 * g);
      const float2 da = *reinterpret_cast<const float2*>(dy + (size_t)m * C + 2 * g);
      const float2 db = *reinterpret_cast<const float2*>(dy + (size_t)m * C + half + 2 * g);
      const float xv[4] = {xa.x, xa.y, xb.x, xb.y};
      const float dym[4] = {da.x * rm, da.y * rm, db.x * rm, db.y * rm};
      float a[4], d_a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = bs[k] + el[k] * xv[k];
#pragma unroll
      for (int i = 0; i < 4; ++i) d_a[i] = W[i] * dym[0] + W[4 + i] * dym[1] + W[8 + i] * dym[2] + W[12 + i] * dym[3];
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) accW[o * 4 + i] += dym[o] * a[i];
      float dxv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { accB[k] += d_a[k]; accL[k] += d_a[k] * xv[k] * el[k]; dxv[k] = d_a[k] * el[k]; }
      *reinterpret_cast<float2*>(dx + (size_t)m * C + 2 * g) = make_float2(dxv[0], dxv[1]);
      *reinterpret_cast<float2*>(dx + (size_t)m * C + half + 2 * g) = make_float2(dxv[2], dxv[3]);
    }
  }
  // Same-address float atomics serialise at L2 (~50 ns each): fold the 4 row phases in LDS first, so every
  // channel sees ONE add per workgroup (and workgroups cover 128 rows).
  __shared__ float sL[4][64][4], sB[4][64][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { sL[ph][g][k] = accL[k]; sB[ph][g][k] = accB[k]; }
#pragma unroll
  for (int i = 0; i < 16; ++i) accW[i] = wave_sum(accW[i]);
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sW[ph][i] = accW[i];
  }
  __syncthreads();
  if (ph == 0 && g < G) {
    const int ch[4] = {2 * g, 2 * g + 1, half + 2 * g, half + 2 * g + 1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      atomicAdd(dlogs + ch[k], sL[0][g][k] + sL[1][g][k] + sL[2][g][k] + sL[3][g][k]);
      atomicAdd(dbias + ch[k], sB[0][g][k] + sB[1][g][k] + sB[2][g][k] + sB[3][g][k]);
    }
  }
  if (threadIdx.x < 16) atomicAdd(dW + threadIdx.x, sW[0][threadIdx.x] + sW[1][threadIdx.x] + sW[2][threadIdx.x] + sW[3][threadIdx.x]);
}

// ------------------------------------------------------------------ affine coupling
// out = [m | logs] ([R, C] fp32 from the `end` conv), x = [x0 | x1]:
//   z = [x0 | (m + exp(logs) * x1) * mask],  logdet[b] += sum logs*mask.
// One workgroup per (64-row chunk, utterance): a wave walks 16 rows, the log-det of the chunk is
// reduced in registers/LDS and leaves as ONE atomic per workgroup (per-row atomics onto B addresses
// serialise: guide G12).
__global__ __launch_bounds__(256) void gt_coupling_fwd_kernel(const float* __restrict__ out, const float* __restrict__ x,
                                                              float* __restrict__ z, const float* __restrict__ rowmask,
                                                              float* __restrict__ logdet, int R, int C, int Tp, int sigmoid_scale,
                                                              const int32_t* __restrict__ row0)
{
  __shared__ float red[4];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = C >> 1;
  const int base = gt_row_base(row0, b, Tp), nrows = gt_row_count(row0, b, Tp);
  const int t0 = blockIdx.x * 64 + w * 16;
  float s = 0.f;
  for (int tt = 0; tt < 16; ++tt) {
    const int t = t0 + tt;
    if (t >= nrows) break;
    const int m = base + t;
    const float rm = rowmask[m];
    for (int c = lane; c < half; c += 64) {
      const float mm = out[(size_t)m * C + c];
      float lg = out[(size_t)m * C + half + c];
      if (sigmoid_scale) lg = __logf(1e-6f + sigmoidf_(lg + 2.0f));
      const float x1 = x[(size_t)m * C + half + c];
      z[(size_t)m * C + c] = x[(size_t)m * C + c];
      z[(size_t)m * C + half + c] = (mm + __expf(lg) * x1) * rm;
      s += lg * rm;
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) { const float tot = red[0] + red[1] + red[2] + red[3]; if (tot != 0.f) atomicAdd(logdet + b, tot); }
}

// reverse (inference, attentions.py:178-180): x = [z0 | (z1 - m) * exp(-logs) * mask]
__global__ __launch_bounds__(256) void gt_coupling_rev_kernel(const float* __restrict__ out, const float* __restrict__ z,
                                                              float* __restrict__ x, const float* __restrict__ rowmask,
                                                              int R, int C, int sigmoid_scale)
{
  const int idx = blockIdx.x * 256 + threadIdx.x, half = C >> 1;
  if (idx >= R * half) return;
  const int m = idx / half, c = idx - m * half;
  const float mm = out[(size_t)m * C + c];
  float lg = out[(size_t)m * C + half + c];
  if (sigmoid_scale) lg = __logf(1e-6f + sigmoidf_(lg + 2.0f));
  x[(size_t)m * C + c] = z[(size_t)m * C + c];
  x[(size_t)m * C + half + c] = (z[(size_t)m * C + half + c] - mm) * __expf(-lg) * rowmask[m];
}

// backward: dz -> dx (x0 part passed through, the start-conv contribution is added later),
// d_out = [d_m | d_logs] in bf16 for the dgrad/wgrad GEMMs.
__global__ __launch_bounds__(256) void gt_coupling_bwd_kernel(const float* __restrict__ out, const float* __restrict__ x,
                                                              const float* __restrict__ dz, const float* __restrict__ dlogdet,
                                                              const float* __restrict__ rowmask, float* __restrict__ dx,
                                                              bf16_t* __restrict__ dout_bf16, int R, int C, int Tp, int sigmoid_scale,
                                                              const int32_t* __restrict__ row0, int B)
{
  const int idx = blockIdx.x * 256 + threadIdx.x, half = C >> 1;
  if (idx >= R * half) return;
  const int m = idx / half, c = idx - m * half;
  const float rm = rowmask[m];
  const float lraw = out[(size_t)m * C + half + c];
  float lg = lraw, dl_draw = 1.0f;
  if (sigmoid_scale) { const float sg = sigmoidf_(lraw + 2.0f); lg = __logf(1e-6f + sg); dl_draw = sg * (1.0f - sg) / (1e-6f + sg); }
  const float e = __expf(lg), x1 = x[(size_t)m * C + half + c];
  const float dz1 = dz[(size_t)m * C + half + c] * rm;
  const float d_m = dz1;
  const float d_lg = (dz1 * e * x1 + dlogdet[gt_row_batch(row0, B, m, Tp)] * rm) * dl_draw;
  dx[(size_t)m * C + c] = dz[(size_t)m * C + c];
  dx[(size_t)m * C + half + c] = dz1 * e;
  dout_bf16[(size_t)m * C + c] = f2bf(d_m);
  dout_bf16[(size_t)m * C + half + c] = f2bf(d_lg);
}

// dx[:, :n] += add (bf16 rows, e.g. the start-conv data gradient) * mask
__global__ __launch_bounds__(256) void gt_rows_add_bf16_kernel(float* __restrict__ dx, int ldx, const bf16_t* __restrict__ add, int lda,
                                                               int R, int n)
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * n) return;
  const int m = idx / n, c = idx - m * n;
  dx[(size_t)m * ldx + c] += bf2f(add[(size_t)m * lda + c]);
}

// ------------------------------------------------------------------ WaveNet gate backward
// acts = T*S, T = tanh(pre_t), S = sigmoid(pre_s), pre = drop(conv) + cond:
//   d_pre_t = d_acts * S * (1 - T^2),  d_pre_s = d_acts * T * S * (1 - S);  the conv-side gradient
//   additionally carries the replayed dropout mask.  Output [R, 2*half] bf16, natural order.
__global__ __launch_bounds__(256) void gt_gate_bwd_kernel(const bf16_t* __restrict__ dacts, int ldd, const bf16_t* __restrict__ T,
                                                          const bf16_t* __restrict__ S, int ldts, bf16_t* __restrict__ dpre, int ldp,
                                                          bf16_t* __restrict__ dpre_cond, int R, int half,
                                                          uint32_t drop_thresh, uint32_t drop_seed, float drop_scale,
                                                          const uint32_t* __restrict__ seed_dev)
{
  if (seed_dev) drop_seed ^= *seed_dev;
  const int idx = blockIdx.x * 256 + threadIdx.x, q4 = half >> 2;
  if (idx >= R * q4) return;
  const int m = idx / q4, c = (idx - m * q4) * 4;
  const uint2 dv = *reinterpret_cast<const uint2*>(dacts + (size_t)m * ldd + c);
  const uint2 tv = *reinterpret_cast<const uint2*>(T + (size_t)m * ldts + c);
  const uint2 sv = *reinterpret_cast<const uint2*>(S + (size_t)m * ldts + c);
  const float d[4] = {bf2f(dv.x & 0xffff), bf2f(dv.x >> 16), bf2f(dv.y & 0xffff), bf2f(dv.y >> 16)};
  const float t[4] = {bf2f(tv.x & 0xffff), bf2f(tv.x >> 16), bf2f(tv.y & 0xffff), bf2f(tv.y >> 16)};
  const float s[4] = {bf2f(sv.x & 0xffff), bf2f(sv.x >> 16), bf2f(sv.y & 0xffff), bf2f(sv.y >> 16)};
  float gt[4], gs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { gt[i] = d[i] * s[i] * (1.0f - t[i] * t[i]); gs[i] = d[i] * t[i] * s[i] * (1.0f - s[i]); }
  if (dpre_cond) {
    *reinterpret_cast<uint2*>(dpre_cond + (size_t)m * ldp + c) = make_uint2(pack2bf(gt[0], gt[1]), pack2bf(gt[2], gt[3]));
    *reinterpret_cast<uint2*>(dpre_cond + (size_t)m * ldp + half + c) = make_uint2(pack2bf(gs[0], gs[1]), pack2bf(gs[2], gs[3]));
  }
  if (drop_thresh) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool kt, ks;
      drop_keep_gate(drop_seed, m, c + i, drop_thresh16(drop_thresh), kt, ks);
      gt[i] = kt ? gt[i] * drop_scale : 0.0f;
      gs[i] = ks ? gs[i] * drop_scale : 0.0f;
    }
  }
  *reinterpret_cast<uint2*>(dpre + (size_t)m * ldp + c) = make_uint2(pack2bf(gt[0], gt[1]), pack2bf(gt[2], gt[3]));
  *reinterpret_cast<uint2*>(dpre + (size_t)m * ldp + half + c) = make_uint2(pack2bf(gs[0], gs[1]), pack2bf(gs[2], gs[3]));
}

// backward of y = mask * dropout(relu(c)):  dc = (y != 0) ? d * scale : 0   (bf16 rows)
__global__ __launch_bounds__(256) void gt_relu_drop_bwd_kernel(const bf16_t* __restrict__ d, int ldd, const bf16_t* __restrict__ y, int ldy,
                                                               bf16_t* __restrict__ dc, int ldc, int R, int n, float scale)
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * n) return;
  const int m = idx / n, c = idx - m * n;
  const bool on = (y[(size_t)m * ldy + c] & 0x7fff) != 0;
  dc[(size_t)m * ldc + c] = on ? f2bf(bf2f(d[(size_t)m * ldd + c]) * scale) : (bf16_t)0;
}

// generic: out_bf16[m, :n] = in_f32[m, :n] * (rowmask ? rowmask[m] : 1)
__global__ __launch_bounds__(256) void gt_rows_f32_to_bf16_kernel(const float* __restrict__ in, int ldi, bf16_t* __restrict__ out, int ldo,
                                                                  const float* __restrict__ rowmask, int R, int n)
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * n) return;
  const int m = idx / n, c = idx - m * n;
  out[(size_t)m * ldo + c] = f2bf(in[(size_t)m * ldi + c] * (rowmask ? rowmask[m] : 1.0f));
}

// ActNorm data-dependent initialisation (modules.py:607-619): masked per-channel batch statistics of the rows the layer
// is about to see.  Pass 1: per-channel sum / sum of squares over all rows (invalid rows are zero in the rows layout), fp64
// partials folded per workgroup, then atomics.  Pass 2 (one workgroup): logs = -0.5 log(max(var, 1e-6)), bias = -mean * exp(logs).
__global__ __launch_bounds__(256) void gt_actnorm_ddi_stats_kernel(const float* __restrict__ x, int R, int C, int rows_per_block,
                                                                  double* __restrict__ sums)
{
  const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
  for (int c = threadIdx.x; c < C; c += 256) {
    double s = 0.0, q = 0.0;
    for (int r = r0; r < r1; ++r) { const double v = x[(size_t)r * C + c]; s += v; q += v * v; }
    atomicAdd(&sums[c], s); atomicAdd(&sums[C + c], q);
  }
}
__global__ __launch_bounds__(256) void gt_actnorm_ddi_finish_kernel(const double* __restrict__ sums, const int32_t* __restrict__ len,
                                                                   int B, int C, float* __restrict__ logs, float* __restrict__ bias)
{
  __shared__ double denom;
  if (threadIdx.x == 0) { double d = 0.0; for (int b = 0; b < B; ++b) d += (double)len[b]; denom = d; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float m = (float)(sums[c] / denom), msq = (float)(sums[C + c] / denom);
    const float v = msq - m * m;
    const float l = 0.5f * logf(fmaxf(v, 1e-6f));
    bias[c] = -m * expf(-l);
    logs[c] = -l;
  }
}


// ------------------------------------------------------------------ [B, C, T] <-> rows at the public boundary
// One launch instead of the index arithmetic + gather + mask + cast chain of ~8 host-side tensor ops per conversion: a 64-row x
// 64-channel tile goes through LDS so that both sides are coalesced (frames are contiguous in [B, C, T], channels in rows).
__device__ __forceinline__ float ld_any(const void* p, size_t i, int f32) { return f32 ? static_cast<const float*>(p)[i] : bf2f(static_cast<const bf16_t*>(p)[i]); }
__device__ __forceinline__ void st_any(void* p, size_t i, int f32, float v) { if (f32) static_cast<float*>(p)[i] = v; else static_cast<bf16_t*>(p)[i] = f2bf(v); }

// rows[m, c] = x[b(m), c, t(m)] for frame rows (0 <= t < T), 0 for halo / rounding rows
__global__ __launch_bounds__(256) void gt_rows_from_bct_kernel(const void* __restrict__ x, int x_f32, void* __restrict__ rows, int rows_f32,
                                                               const int32_t* __restrict__ row0, int B, int C, int T, int Tp, int R)
{
  __shared__ float tile[64][65];
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int lr = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int m = m0 + lr;
  int b = 0, t = -1;
  if (m < R) { b = gt_row_batch(row0, B, m, Tp); t = m - gt_row_base(row0, b, Tp) - 2; }
  const bool in = t >= 0 && t < T;
#pragma unroll 4
  for (int cc = q; cc < 64; cc += 4) {
    const int c = c0 + cc;
    tile[cc][lr] = (in && c < C) ? ld_any(x, ((size_t)b * C + c) * T + t, x_f32) : 0.f;
  }
  __syncthreads();
#pragma unroll 4
  for (int rr = q; rr < 64; rr += 4) {
    const int mm = m0 + rr, c = c0 + lr;
    if (mm < R && c < C) st_any(rows, (size_t)mm * C + c, rows_f32, tile[lr][rr]);
  }
}
// x[b, c, t] = rows[base(b) + 2 + t, c] for t < len[b], 0 beyond
__global__ __launch_bounds__(256) void gt_bct_from_rows_kernel(const void* __restrict__ rows, int rows_f32, void* __restrict__ x, int x_f32,
                                                               const int32_t* __restrict__ len, const int32_t* __restrict__ row0,
                                                               int B, int C, int T, int Tp, int R)
{
  __shared__ float tile[64][65];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int lr = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int n = min(len[b], T), base = gt_row_base(row0, b, Tp) + 2;
#pragma unroll 4
  for (int rr = q; rr < 64; rr += 4) {
    const int t = t0 + rr, c = c0 + lr, m = base + t;
    tile[rr][lr] = (t < n && c < C && m < R) ? ld_any(rows, (size_t)m * C + c, rows_f32) : 0.f;
  }
  __syncthreads();
#pragma unroll 4
  for (int cc = q; cc < 64; cc += 4) {
    const int c = c0 + cc, t = t0 + lr;
    if (c < C && t < T) st_any(x, ((size_t)b * C + c) * T + t, x_f32, tile[lr][cc]);
  }
}

// Gradient of modules.WNP's affine per-frame conditioning (cond_layer1 has ONE input channel: cond = b + contour * w, squeezed in time,
// see gt_wn_stack_fwd): from the gate backward's d pre rows of the WaveNet's layers (before the dropout mask where dropout is on),
//   d w[off_i + c] += sum_m dpre_i[m, c] * sig[m, par_i],   d b[off_i + c] += sum_m dpre_i[m, c],   c < 2H, layer i.
// grid (row slab, layer); 768 threads = 4 row phases x 192 threads of 2 consecutive channels each (one 4-byte load per row), fp32 sums
// folded over the phases in LDS, one atomic pair per channel pair and workgroup.  (Round 3 first ran 192 threads down 256 rows each:
// 64 dependent load batches, 24 us per launch, 24 launches on the decoder's backward chain of cfg 5.)
constexpr int CAG_PH = 4;
__global__ __launch_bounds__(192 * CAG_PH) void gt_cond_affine_grads_kernel(const bf16_t* __restrict__ d0, const bf16_t* __restrict__ d1, const bf16_t* __restrict__ d2,
                                                                   const bf16_t* __restrict__ d3, int lddp, const float* __restrict__ sig,
                                                                   float* __restrict__ dw, float* __restrict__ db, int R, int H, int n_layers,
                                                                   int rows_per_block)
{
  __shared__ float red[CAG_PH][192][4];
  const int layer = blockIdx.y, O = H * n_layers;
  const bf16_t* d = layer == 0 ? d0 : (layer == 1 ? d1 : (layer == 2 ? d2 : d3));
  const int par = (2 * H * layer) / O, off = 2 * H * layer - par * O;
  const int tc = threadIdx.x % 192, ph = threadIdx.x / 192;
  const int c = 2 * tc;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(R, m0 + rows_per_block);
  float w0 = 0.f, w1 = 0.f, b0 = 0.f, b1 = 0.f;
  if (c < 2 * H) {
#pragma unroll 8
    for (int m = m0 + ph; m < m1; m += CAG_PH) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(d + (size_t)m * lddp + c);
      const float x0 = bf2f(v & 0xffff), x1 = bf2f(v >> 16), p = sig[2 * (size_t)m + par];
      b0 += x0; b1 += x1; w0 += x0 * p; w1 += x1 * p;
    }
  }
  red[ph][tc][0] = w0; red[ph][tc][1] = w1; red[ph][tc][2] = b0; red[ph][tc][3] = b1;
  __syncthreads();
  if (ph == 0 && c < 2 * H) {
#pragma unroll
    for (int q = 1; q < CAG_PH; ++q) { w0 += red[q][tc][0]; w1 += red[q][tc][1]; b0 += red[q][tc][2]; b1 += red[q][tc][3]; }
    atomicAdd(dw + off + c, w0); atomicAdd(dw + off + c + 1, w1);
    atomicAdd(db + off + c, b0); atomicAdd(db + off + c + 1, b1);
  }
}

}  // namespace


#define GT_ST(s) static_cast<hipStream_t>(s)
#define GT_RET() return gt_launch_status(__func__)

extern "C" int gt_squeeze_rows_f32(const float* y, float* rows, const int32_t* len_sq, int B, int C, int Ty, int Tp,
                                   const int32_t* row0, void* stream)
{
  if (!y || !rows || !len_sq || B <= 0 || C <= 0 || C > 80 || Ty <= 0 || Tp <= 2 * HALO) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_squeeze_rows_kernel, dim3((Tp + 31) / 32, B), dim3(256), 0, GT_ST(stream), y, rows, len_sq, B, C, Ty, Tp, row0);
  GT_RET();
}
extern "C" int gt_unsqueeze_rows_f32(const float* rows, float* y, const int32_t* len_sq, int B, int C, int Ty, int Tp,
                                     const int32_t* row0, void* stream)
{
  if (!y || !rows || !len_sq || B <= 0 || C <= 0 || C > 80 || Ty <= 0 || Tp <= 2 * HALO) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_unsqueeze_rows_kernel, dim3((Ty / 2 + 1 + 31) / 32, B), dim3(256), 0, GT_ST(stream), rows, y, len_sq, B, C, Ty, Tp, row0);
  GT_RET();
}
extern "C" int gt_flow_scalars_multi(const void* logs_ptrs, const void* w_ptrs, int C, float* scal, int n, void* stream)
{
  if (!logs_ptrs || !w_ptrs || !scal || C <= 0 || n < 0) return GT_E_INVAL;
  if (n == 0) return GT_OK;
  hipLaunchKernelGGL(gt_flow_scalars_multi_kernel, dim3(n), dim3(64), 0, GT_ST(stream), static_cast<const float* const*>(logs_ptrs),
                     static_cast<const float* const*>(w_ptrs), C, scal);
  GT_RET();
}

extern "C" int gt_flow_scalars(const float* logs, int C, const float* W, float* scal, void* stream)
{
  if (!logs || !W || !scal || C <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_flow_scalars_kernel, dim3(1), dim3(64), 0, GT_ST(stream), logs, C, W, scal);
  GT_RET();
}
extern "C" int gt_actnorm_invconv_fwd(const float* x, float* y, void* y0_bf16, int ld0, const float* logs, const float* bias,
                                      const float* W, const float* scal, const float* rowmask, const int32_t* len,
                                      float* logdet, int B, int R, int C, void* stream)
{
  if (!x || !y || !logs || !bias || !W || !rowmask || R <= 0 || (C & 3) || C > 256) return GT_E_INVAL;
  const int G = C >> 2;
  if (logdet && (!scal || !len || B <= 0)) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_actnorm_invconv_fwd_kernel, dim3((R * G + 255) / 256), dim3(256), 0, GT_ST(stream),
                     x, y, static_cast<bf16_t*>(y0_bf16), ld0, logs, bias, W, rowmask, R, C, scal, len, logdet, B);
  GT_RET();
}
extern "C" int gt_actnorm_invconv_bwd(const float* x, const float* dy, float* dx, const float* logs, const float* bias,
                                      const float* W, const float* scal, const float* rowmask, const int32_t* len,
                                      const float* dlogdet, float* dlogs, float* dbias, float* dW, int B, int R, int C, void* stream)
{
  if (!x || !dy || !dx || !logs || !bias || !W || !rowmask || !dlogs || !dbias || !dW || R <= 0 || (C & 3) || C > 256) return GT_E_INVAL;
  // rows per workgroup: the per-channel partials are folded in LDS before the atomics, so few fat workgroups win;
  // 128 rows down to 32 for short inputs (>= ~64 workgroups)
  const int rows_per_block = R >= 64 * 128 ? 128 : (R >= 64 * 64 ? 64 : 32);
  if (dlogdet && (!scal || !len || B <= 0)) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_actnorm_invconv_bwd_kernel, dim3((R + rows_per_block - 1) / rows_per_block), dim3(256), 0, GT_ST(stream),
                     x, dy, dx, logs, bias, W, rowmask, dlogs, dbias, dW, R, C, rows_per_block, scal, len, dlogdet, B);
  GT_RET();
}
extern "C" int gt_actnorm_invconv_rev(const float* y, float* x, void* x0_bf16, int ld0, const float* logs, const float* bias,
                                      const float* scal, const float* rowmask, int R, int C, void* stream)
{
  if (!y || !x || !logs || !bias || !scal || !rowmask || R <= 0 || (C & 3) || C > 256) return GT_E_INVAL;
  const int G = C >> 2;
  hipLaunchKernelGGL(gt_actnorm_invconv_rev_kernel, dim3((R * G + 255) / 256), dim3(256), 0, GT_ST(stream),
                     y, x, static_cast<bf16_t*>(x0_bf16), ld0, logs, bias, scal, rowmask, R, C);
  GT_RET();
}
extern "C" int gt_coupling_rev(const float* out, const float* z, float* x, const float* rowmask, int R, int C,
                               int sigmoid_scale, void* stream)
{
  if (!out || !z || !x || !rowmask || R <= 0 || (C & 1)) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_coupling_rev_kernel, dim3((R * (C / 2) + 255) / 256), dim3(256), 0, GT_ST(stream),
                     out, z, x, rowmask, R, C, sigmoid_scale);
  GT_RET();
}
extern "C" int gt_coupling_fwd(const float* out, const float* x, float* z, const float* rowmask, float* logdet,
                               int B, int R, int C, int Tp, const int32_t* row0, int sigmoid_scale, void* stream)
{
  if (!out || !x || !z || !rowmask || !logdet || R <= 0 || B <= 0 || (C & 1)) return GT_E_INVAL;
  if (Tp <= 0 || (!row0 && R != B * Tp)) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_coupling_fwd_kernel, dim3((Tp + 63) / 64, B), dim3(256), 0, GT_ST(stream), out, x, z, rowmask, logdet, R, C, Tp, sigmoid_scale, row0);
  GT_RET();
}
extern "C" int gt_coupling_bwd(const float* out, const float* x, const float* dz, const float* dlogdet, const float* rowmask,
                               float* dx, void* dout_bf16, int B, int R, int C, int Tp, const int32_t* row0, int sigmoid_scale,
                               void* stream)
{
  if (!out || !x || !dz || !dlogdet || !rowmask || !dx || !dout_bf16 || R <= 0 || B <= 0 || Tp <= 0 || (C & 1)) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_coupling_bwd_kernel, dim3((R * (C / 2) + 255) / 256), dim3(256), 0, GT_ST(stream),
                     out, x, dz, dlogdet, rowmask, dx, static_cast<bf16_t*>(dout_bf16), R, C, Tp, sigmoid_scale, row0, B);
  GT_RET();
}
extern "C" int gt_rows_add_bf16(float* dx, int ldx, const void* add, int lda, int R, int n, void* stream)
{
  if (!dx || !add || R <= 0 || n <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_rows_add_bf16_kernel, dim3((R * n + 255) / 256), dim3(256), 0, GT_ST(stream), dx, ldx, static_cast<const bf16_t*>(add), lda, R, n);
  GT_RET();
}
extern "C" int gt_gate_bwd(const void* dacts, int ldd, const void* T, const void* S, int ldts, void* dpre, int ldp, void* dpre_cond,
                           int R, int half, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream)
{
  if (!dacts || !T || !S || !dpre || R <= 0 || (half & 3) || (ldd & 3) || (ldts & 3) || (ldp & 3)) return GT_E_INVAL;
  uint32_t th = 0; float sc = 1.0f;
  if (drop_p > 0.f) { th = (uint32_t)((double)drop_p * 4294967296.0); sc = 1.0f / (1.0f - drop_p); }
  hipLaunchKernelGGL(gt_gate_bwd_kernel, dim3((R * (half / 4) + 255) / 256), dim3(256), 0, GT_ST(stream),
                     static_cast<const bf16_t*>(dacts), ldd, static_cast<const bf16_t*>(T), static_cast<const bf16_t*>(S), ldts,
                     static_cast<bf16_t*>(dpre), ldp, static_cast<bf16_t*>(dpre_cond), R, half, th, drop_seed, sc, seed_dev);
  GT_RET();
}
extern "C" int gt_relu_drop_bwd(const void* d, int ldd, const void* y, int ldy, void* dc, int ldc, int R, int n, float drop_p, void* stream)
{
  if (!d || !y || !dc || R <= 0 || n <= 0 || drop_p >= 1.f) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_relu_drop_bwd_kernel, dim3((R * n + 255) / 256), dim3(256), 0, GT_ST(stream), static_cast<const bf16_t*>(d), ldd,
                     static_cast<const bf16_t*>(y), ldy, static_cast<bf16_t*>(dc), ldc, R, n, drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f);
  GT_RET();
}
extern "C" int gt_rows_f32_to_bf16(const float* in, int ldi, void* out, int ldo, const float* rowmask, int R, int n, void* stream)
{
  if (!in || !out || R <= 0 || n <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_rows_f32_to_bf16_kernel, dim3((R * n + 255) / 256), dim3(256), 0, GT_ST(stream), in, ldi, static_cast<bf16_t*>(out), ldo, rowmask, R, n);
  GT_RET();
}

extern "C" int gt_actnorm_ddi(const float* x, const int32_t* len, int B, int R, int C, double* workspace, float* logs, float* bias,
                              void* stream)
{
  if (!x || !len || !workspace || !logs || !bias || B <= 0 || R <= 0 || C <= 0) return GT_E_INVAL;
  if (hipMemsetAsync(workspace, 0, sizeof(double) * 2 * C, GT_ST(stream)) != hipSuccess) return GT_E_LAUNCH;
  const int rpb = 64;
  hipLaunchKernelGGL(gt_actnorm_ddi_stats_kernel, dim3((R + rpb - 1) / rpb), dim3(256), 0, GT_ST(stream), x, R, C, rpb, workspace);
  hipLaunchKernelGGL(gt_actnorm_ddi_finish_kernel, dim3(1), dim3(256), 0, GT_ST(stream), workspace, len, B, C, logs, bias);
  GT_RET();
}

extern "C" int gt_rows_from_bct(const void* x, int x_f32, void* rows, int rows_f32, const int32_t* row0, int B, int C, int T, int Tp, int R,
                                void* stream)
{
  if (!x || !rows || B <= 0 || C <= 0 || T <= 0 || Tp <= 0 || R <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_rows_from_bct_kernel, dim3((R + 63) / 64, (C + 63) / 64), dim3(256), 0, GT_ST(stream), x, x_f32, rows, rows_f32, row0,
                     B, C, T, Tp, R);
  GT_RET();
}
extern "C" int gt_bct_from_rows(const void* rows, int rows_f32, void* x, int x_f32, const int32_t* lengths, const int32_t* row0,
                                int B, int C, int T, int Tp, int R, void* stream)
{
  if (!x || !rows || !lengths || B <= 0 || C <= 0 || T <= 0 || Tp <= 0 || R <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_bct_from_rows_kernel, dim3((T + 63) / 64, (C + 63) / 64, B), dim3(256), 0, GT_ST(stream), rows, rows_f32, x, x_f32,
                     lengths, row0, B, C, T, Tp, R);
  GT_RET();
}

extern "C" int gt_cond_affine_grads(const void* dpre0, const void* dpre1, const void* dpre2, const void* dpre3, int lddp, const float* sig,
                                    float* dw, float* db, int R, int H, int n_layers, void* stream)
{
  if (!dpre0 || !sig || !dw || !db || R < 0 || H != 192 || n_layers < 1 || n_layers > 4 || ((H * n_layers) % (2 * H)) || (lddp & 1)) return GT_E_INVAL;
  if ((n_layers > 1 && !dpre1) || (n_layers > 2 && !dpre2) || (n_layers > 3 && !dpre3)) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  const int rows_per_block = 128;
  hipLaunchKernelGGL(gt_cond_affine_grads_kernel, dim3((R + rows_per_block - 1) / rows_per_block, n_layers), dim3(192 * CAG_PH), 0, GT_ST(stream),
                     static_cast<const bf16_t*>(dpre0), static_cast<const bf16_t*>(dpre1), static_cast<const bf16_t*>(dpre2),
                     static_cast<const bf16_t*>(dpre3), lddp, sig, dw, db, R, H, n_layers, rows_per_block);
  GT_RET();
}
