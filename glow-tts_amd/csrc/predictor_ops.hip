// Stochastic duration / pitch / energy predictors of the fork on the rows layout, gfx950.
//
// Replaces the PyTorch op chains of
//   modules.DilatedDepthSeparableConv.forward   modules.py:718-735   (depthwise dilated conv + LayerNorm2 + GELU, 1x1 conv,
//                                                                      LayerNorm2 + GELU + dropout, residual)
//   modules.ElementwiseAffine.forward           modules.py:750-756
//   modules.ConvFlow.forward                    modules.py:792-819   (pre, DDSConv, proj, rational-quadratic spline)
//   transforms.piecewise_rational_quadratic_transform               transforms.py:12-202 (10 bins, linear tails, bound 5)
//   models.Stochastic{Duration,Pitch,Energy}Predictor.forward        models.py:261-322, 364-396, 438-470 (the glue between)
// and what autograd derives for them.  These chains are HBM/latency-bound row-wise work at C = 192 channels: every kernel
// here is "one wave walks rows, a lane owns C/64 channels" (256-byte coalesced row segments, LayerNorm statistics by
// wave reduction, per-channel parameter gradients summed in registers over the wave's rows and pushed with one atomic per
// lane at the end).  The only dense contraction, the 192x192 1x1 conv, runs on the bf16 MFMA GEMM (gt_conv_gemm_bf16);
// the 29-row proj of a ConvFlow is an exact-fp32 VALU product fused with the spline (its parameters feed a softmax and a
// bin search, so they stay fp32).  All statistics, the spline and the likelihood sums are fp32.
#include "common.h"
#include "../../include/glowtts_hip.h"

namespace {

constexpr int PC = 192;                       // channels of every predictor network (filter_channels = in_channels, models.py:223)
constexpr int NC = PC / 64;                   // channels per lane
#ifndef DDS_RPW
#define DDS_RPW 2                             // (cfg 5 step: 16.2 ms at 8 rows per wave, 15.5 at 4, 15.0-15.2 at 2, 15.1 at 1)
#endif
constexpr int RPW = DDS_RPW;                  // rows per wave per workgroup (forward kernels)
#ifndef DDS_RPB
#define DDS_RPB 2                             // (cfg 5 step, with the partial-row reduce: 18.05 ms at 8 rows per wave, 17.5 at 4; 14.9 at 2 with DDS_RPW 2)
#endif
constexpr int RPB = DDS_RPB;                  // ... of the backward kernels that also accumulate parameter gradients

// parameter-gradient partial of this lane -> ONE atomic per address per workgroup: the four waves' values are folded in LDS
// first (same-address float atomics serialise at L2, ~25-50 ns each; with one per wave they were most of these kernels)
__device__ __forceinline__ void wg_fold_add(float* __restrict__ dst, float v, float* sm, int lane, int wave)
{
  sm[wave * 64 + lane] = v;
  __syncthreads();
  if (wave == 0) {
    const float s = sm[lane] + sm[64 + lane] + sm[128 + lane] + sm[192 + lane];
    if (s != 0.f) atomicAdd(dst, s);
  }
  __syncthreads();
}

// the same fold; the workgroup's sum goes to its row of a partials buffer (plain store: gt_param_partials_reduce adds the column sums of
// all rows to the parameter gradients in one launch per module backward) if there is one, else to dst with one atomic
__device__ __forceinline__ void wg_fold_out(float* __restrict__ dst, float* __restrict__ part, float v, float* sm, int lane, int wave)
{
  sm[wave * 64 + lane] = v;
  __syncthreads();
  if (wave == 0) {
    const float s = sm[lane] + sm[64 + lane] + sm[128 + lane] + sm[192 + lane];
    if (part) *part = s;
    else if (s != 0.f) atomicAdd(dst, s);
  }
  __syncthreads();
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x)
{
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ void ld3(const float* p, int lane, float (&v)[NC])
{
#pragma unroll
  for (int j = 0; j < NC; ++j) v[j] = p[lane + 64 * j];
}
// bf16x3 operand of a split GEMM (gt_pack_conv_weights flag 8): row = [hi | hi | lo], each PC wide
__device__ __forceinline__ void split3_store(bf16_t* row, int c, float v)
{
  const bf16_t hi = f2bf(v);
  row[c] = hi; row[PC + c] = hi; row[2 * PC + c] = f2bf(v - bf2f(hi));
}
// (mean, rstd) of one row held as NC values per lane
__device__ __forceinline__ void ln_stats(const float (&v)[NC], float eps, float& mean, float& rstd)
{
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j) s += v[j];
  mean = wave_sum(s) * (1.0f / PC);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j) { const float d = v[j] - mean; q += d * d; }
  rstd = rsqrtf(wave_sum(q) * (1.0f / PC) + eps);
}
// LayerNorm backward for one row: du (gradient at the affine output) -> gradient at the LayerNorm input
__device__ __forceinline__ void ln_bwd_row(const float (&du)[NC], const float (&xhat)[NC], const float (&gamma)[NC], float rstd, float (&dx)[NC])
{
  float s1 = 0.f, s2 = 0.f, dxh[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) { dxh[j] = du[j] * gamma[j]; s1 += dxh[j]; s2 += dxh[j] * xhat[j]; }
  s1 = wave_sum(s1) * (1.0f / PC); s2 = wave_sum(s2) * (1.0f / PC);
#pragma unroll
  for (int j = 0; j < NC; ++j) dx[j] = rstd * (dxh[j] - s1 - xhat[j] * s2);
}
// a tap of the dilated depthwise conv reads row m + off only inside the same utterance (zero padding, modules.py:709-712)
__device__ __forceinline__ bool tap_ok(const int32_t* utt, const float* rowmask, int m, int off, int R)
{
  const int mm = m + off;
  return mm >= 0 && mm < R && utt[mm] == utt[m] && rowmask[mm] != 0.f;
}

// h1 = dwconv_d(x) + b for row m (x rows are masked: rows outside an utterance's frames are zero)
__device__ __forceinline__ void sep_row(const float* __restrict__ x, int ldx, const float (&w)[3][NC], const float (&b)[NC],
                                        const int32_t* utt, const float* rowmask, int m, int d, int R, int lane, float (&h1)[NC])
{
#pragma unroll
  for (int j = 0; j < NC; ++j) h1[j] = b[j];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int off = (k - 1) * d;
    if (k != 1 && !tap_ok(utt, rowmask, m, off, R)) continue;
    const float* xr = x + (size_t)(m + off) * ldx;
#pragma unroll
    for (int j = 0; j < NC; ++j) h1[j] += w[k][j] * xr[lane + 64 * j];
  }
}

// ------------------------------------------------------------------------------------------------ DDSConv layer
// forward, first half: a1 = gelu(LN1(dwconv(x * mask) + b))  -> bf16 rows (the 1x1 GEMM's operand)
__global__ __launch_bounds__(256) void gt_dds_sep_fwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ gamma, const float* __restrict__ beta, const int32_t* __restrict__ utt,
    const float* __restrict__ rowmask, bf16_t* __restrict__ a1, int lda, int R, int d, float eps)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wk[3][NC], bb[NC], g[NC], be[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = lane + 64 * j;
    wk[0][j] = w[c * 3]; wk[1][j] = w[c * 3 + 1]; wk[2][j] = w[c * 3 + 2];
    bb[j] = b[c]; g[j] = gamma[c]; be[j] = beta[c];
  }
  const int m0 = (blockIdx.x * 4 + wave) * RPW;
  for (int i = 0; i < RPW; ++i) {
    const int m = m0 + i;
    if (m >= R) break;
    bf16_t* o = a1 + (size_t)m * lda;
    if (rowmask[m] == 0.f) {
#pragma unroll
      for (int j = 0; j < NC; ++j) split3_store(o, lane + 64 * j, 0.f);
      continue;
    }
    float h1[NC], mean, rstd;
    sep_row(x, ldx, wk, bb, utt, rowmask, m, d, R, lane, h1);
    ln_stats(h1, eps, mean, rstd);
#pragma unroll
    for (int j = 0; j < NC; ++j) split3_store(o, lane + 64 * j, gelu_f((h1[j] - mean) * rstd * g[j] + be[j]));
  }
}

// forward, second half: x_next = (x + dropout(gelu(LN2(h2)))) * mask   (h2 = 1x1 conv output incl. bias)
__global__ __launch_bounds__(256) void gt_dds_out_fwd_kernel(
    const float* __restrict__ h2, const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ rowmask, float* __restrict__ out, bf16_t* __restrict__ outb,
    int R, float eps, uint32_t thresh, uint32_t seed, const uint32_t* __restrict__ seed_dev, float scale)
{
  if (seed_dev) seed ^= *seed_dev;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g[NC], be[NC];
  ld3(gamma, lane, g); ld3(beta, lane, be);
  const int m0 = (blockIdx.x * 4 + wave) * RPW;
  for (int i = 0; i < RPW; ++i) {
    const int m = m0 + i;
    if (m >= R) break;
    float v[NC] = {};
    if (rowmask[m] != 0.f) {
      float h[NC], mean, rstd;
      ld3(h2 + (size_t)m * PC, lane, h);
      ln_stats(h, eps, mean, rstd);
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        float y = gelu_f((h[j] - mean) * rstd * g[j] + be[j]);
        if (thresh) y = drop_keep(seed, m, lane + 64 * j, thresh) ? y * scale : 0.f;
        v[j] = x[(size_t)m * ldx + lane + 64 * j] + y;
      }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      out[(size_t)m * PC + lane + 64 * j] = v[j];
      if (outb) outb[(size_t)m * PC + lane + 64 * j] = f2bf(v[j]);
    }
  }
}

// backward of the second half: dy (gradient at x_next) -> d h2 (bf16: operand of the 1x1 data / weight gradient GEMMs);
// d gamma2 / d beta2 accumulate
__global__ __launch_bounds__(256) void gt_dds_out_bwd_kernel(
    const float* __restrict__ h2, const float* __restrict__ dy, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ rowmask, bf16_t* __restrict__ dh2, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ partials, int R, float eps, uint32_t thresh, uint32_t seed, const uint32_t* __restrict__ seed_dev, float scale)
{
  if (seed_dev) seed ^= *seed_dev;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g[NC], be[NC], ag[NC] = {}, ab[NC] = {};
  ld3(gamma, lane, g); ld3(beta, lane, be);
  __shared__ float fold_sm[256];
  const int m0 = (blockIdx.x * 4 + wave) * RPB;
  for (int i = 0; i < RPB; ++i) {
    const int m = m0 + i;
    if (m >= R) break;
    float o[NC] = {};
    if (rowmask[m] != 0.f) {
      float h[NC], d[NC], xh[NC], du[NC], mean, rstd;
      ld3(h2 + (size_t)m * PC, lane, h);
      ld3(dy + (size_t)m * PC, lane, d);
      ln_stats(h, eps, mean, rstd);
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        xh[j] = (h[j] - mean) * rstd;
        float da = d[j];
        if (thresh) da = drop_keep(seed, m, lane + 64 * j, thresh) ? da * scale : 0.f;
        du[j] = da * gelu_grad(xh[j] * g[j] + be[j]);
        ag[j] += du[j] * xh[j]; ab[j] += du[j];
      }
      ln_bwd_row(du, xh, g, rstd, o);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) split3_store(dh2 + (size_t)m * 3 * PC, lane + 64 * j, o[j]);
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = lane + 64 * j;
    float* pr = partials ? partials + (size_t)blockIdx.x * 2 * PC : nullptr;            // [gamma | beta]
    wg_fold_out(dgamma + c, pr ? pr + c : nullptr, ag[j], fold_sm, lane, wave);
    wg_fold_out(dbeta + c, pr ? pr + PC + c : nullptr, ab[j], fold_sm, lane, wave);
  }
}

// backward of the first half, row-local part: d a1 (fp32, from the 1x1 data-gradient GEMM) -> d h1 (gradient at the
// depthwise conv's output); d gamma1 / d beta1 accumulate.  h1 and its statistics are recomputed from x (3 taps).
__global__ __launch_bounds__(256) void gt_dds_sep_bwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ gamma, const float* __restrict__ beta, const int32_t* __restrict__ utt,
    const float* __restrict__ rowmask, const float* __restrict__ da1, float* __restrict__ dh1,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ partials, int R, int d, float eps)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wk[3][NC], bb[NC], g[NC], be[NC], ag[NC] = {}, ab[NC] = {};
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = lane + 64 * j;
    wk[0][j] = w[c * 3]; wk[1][j] = w[c * 3 + 1]; wk[2][j] = w[c * 3 + 2];
    bb[j] = b[c]; g[j] = gamma[c]; be[j] = beta[c];
  }
  __shared__ float fold_sm[256];
  const int m0 = (blockIdx.x * 4 + wave) * RPB;
  for (int i = 0; i < RPB; ++i) {
    const int m = m0 + i;
    if (m >= R) break;
    float o[NC] = {};
    if (rowmask[m] != 0.f) {
      float h1[NC], xh[NC], du[NC], da[NC], mean, rstd;
      sep_row(x, ldx, wk, bb, utt, rowmask, m, d, R, lane, h1);
      ln_stats(h1, eps, mean, rstd);
      ld3(da1 + (size_t)m * PC, lane, da);
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        xh[j] = (h1[j] - mean) * rstd;
        du[j] = da[j] * gelu_grad(xh[j] * g[j] + be[j]);
        ag[j] += du[j] * xh[j]; ab[j] += du[j];
      }
      ln_bwd_row(du, xh, g, rstd, o);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) dh1[(size_t)m * PC + lane + 64 * j] = o[j];
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = lane + 64 * j;
    float* pr = partials ? partials + (size_t)blockIdx.x * 2 * PC : nullptr;            // [gamma | beta]
    wg_fold_out(dgamma + c, pr ? pr + c : nullptr, ag[j], fold_sm, lane, wave);
    wg_fold_out(dbeta + c, pr ? pr + PC + c : nullptr, ab[j], fold_sm, lane, wave);
  }
}

// backward of the depthwise conv + the residual: dx = (dy + sum_k w[k] * dh1[m - (k-1)d]) * mask;
// dw[c,k] += sum_m dh1[m,c] x[m + (k-1)d, c],  db[c] += sum_m dh1[m,c]
__global__ __launch_bounds__(256) void gt_dds_dw_bwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ dh1, const float* __restrict__ dy,
    const float* __restrict__ w, const int32_t* __restrict__ utt, const float* __restrict__ rowmask,
    float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, float* __restrict__ partials, int R, int d)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wk[3][NC], aw[3][NC] = {}, ab[NC] = {};
#pragma unroll
  for (int j = 0; j < NC; ++j) { const int c = lane + 64 * j; wk[0][j] = w[c * 3]; wk[1][j] = w[c * 3 + 1]; wk[2][j] = w[c * 3 + 2]; }
  __shared__ float fold_sm[256];
  const int m0 = (blockIdx.x * 4 + wave) * RPB;
  for (int i = 0; i < RPB; ++i) {
    const int m = m0 + i;
    if (m >= R) break;
    float o[NC] = {};
    if (rowmask[m] != 0.f) {
      float g0[NC];
      ld3(dh1 + (size_t)m * PC, lane, g0);
      ld3(dy + (size_t)m * PC, lane, o);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int off = (k - 1) * d;
        const bool ok = k == 1 || tap_ok(utt, rowmask, m, off, R);
        if (ok) {                                              // forward tap k of row m read x[m + off]
          const float* xr = x + (size_t)(m + off) * ldx;
#pragma unroll
          for (int j = 0; j < NC; ++j) aw[k][j] += g0[j] * xr[lane + 64 * j];
        }
        // row m - off used x[m] through tap k
        if (k == 1 || tap_ok(utt, rowmask, m, -off, R)) {
          const float* gr = dh1 + (size_t)(m - off) * PC;
#pragma unroll
          for (int j = 0; j < NC; ++j) o[j] += wk[k][j] * gr[lane + 64 * j];
        }
      }
#pragma unroll
      for (int j = 0; j < NC; ++j) ab[j] += g0[j];
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) dx[(size_t)m * PC + lane + 64 * j] = o[j];
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = lane + 64 * j;
    float* pr = partials ? partials + (size_t)blockIdx.x * 4 * PC : nullptr;            // [dw[C][3] | db[C]]
#pragma unroll
    for (int k = 0; k < 3; ++k) wg_fold_out(dw + c * 3 + k, pr ? pr + c * 3 + k : nullptr, aw[k][j], fold_sm, lane, wave);
    wg_fold_out(db + c, pr ? pr + 3 * PC + c : nullptr, ab[j], fold_sm, lane, wave);
  }
}

// ------------------------------------------------------------------------------------------------ ConvFlow: pre
// x0 = (w_pre * z[:, 0] + b_pre + g1 (+ g2)) * mask      (modules.py:794-795 with DDSConv's `x = x + g`, :724-725)
__global__ __launch_bounds__(256) void gt_convflow_pre_fwd_kernel(
    const float* __restrict__ z, int ldz, const float* __restrict__ wp, const float* __restrict__ bp,
    const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ rowmask, float* __restrict__ out, int R)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w[NC], b[NC];
  ld3(wp, lane, w); ld3(bp, lane, b);
  const int m0 = (blockIdx.x * 4 + wave) * RPW;
  for (int i = 0; i < RPW; ++i) {
    const int m = m0 + i;
    if (m >= R) break;
    const bool on = rowmask[m] != 0.f;
    const float zz = on ? z[(size_t)m * ldz] : 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const size_t q = (size_t)m * PC + lane + 64 * j;
      float v = 0.f;
      if (on) { v = w[j] * zz + b[j]; if (g1) v += g1[q]; if (g2) v += g2[q]; }
      out[q] = v;
    }
  }
}
// dx0 -> d w_pre, d b_pre (accumulate), dz[:, 0] += sum_c dx0 * w_pre, dg += dx0
__global__ __launch_bounds__(256) void gt_convflow_pre_bwd_kernel(
    const float* __restrict__ dx0, const float* __restrict__ z, int ldz, const float* __restrict__ wp,
    const float* __restrict__ rowmask, float* __restrict__ dwp, float* __restrict__ dbp, float* __restrict__ dz, int lddz,
    float* __restrict__ dg, int R)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w[NC], aw[NC] = {}, ab[NC] = {};
  ld3(wp, lane, w);
  __shared__ float fold_sm[256];
  const int m0 = (blockIdx.x * 4 + wave) * RPB;
  for (int i = 0; i < RPB; ++i) {
    const int m = m0 + i;
    if (m >= R) break;
    if (rowmask[m] == 0.f) continue;
    float d[NC], s = 0.f;
    ld3(dx0 + (size_t)m * PC, lane, d);
    const float zz = z[(size_t)m * ldz];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      aw[j] += d[j] * zz; ab[j] += d[j]; s += d[j] * w[j];
      if (dg) dg[(size_t)m * PC + lane + 64 * j] += d[j];
    }
    s = wave_sum(s);
    if (lane == 0 && dz) dz[(size_t)m * lddz] += s;
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) { wg_fold_add(dwp + lane + 64 * j, aw[j], fold_sm, lane, wave); wg_fold_add(dbp + lane + 64 * j, ab[j], fold_sm, lane, wave); }
}

// ------------------------------------------------------------------------------------------------ spline
constexpr int NB = 10;                        // bins (modules.py:777)
constexpr int NPAR = 3 * NB - 1;              // 29 parameters per element
constexpr float TAIL = 5.0f, MINBIN = 1e-3f, MINDER = 1e-3f;

// Everything below indexes its small arrays with compile-time constants only (fully unrolled loops, selects instead of
// gathers), so they live in registers: no private-memory (scratch) arrays — the build's resource audit enforces it.
struct Spline {                               // what the forward and the backward share for one element
  float sw[NB], sh[NB];                       // softmax(unnormalised widths / heights)
  int k; bool inside;
  float cwk, chk;                             // left knot of the element's bin
  float w, h, delta, d0, d1, theta;
  float u0, u1;                               // raw derivative parameters of the bin's two knots (backward)
};
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }

// knots of one axis from the raw parameters: c[0] = -B, c[i] = 2B * cumsum_i - B, c[K] = B (transforms.py:113-134)
__device__ __forceinline__ void spline_knots(const float* p, float inv_sqrt_c, float (&sm)[NB], float (&c)[NB + 1])
{
  float mx = -1e30f;
#pragma unroll
  for (int i = 0; i < NB; ++i) mx = fmaxf(mx, p[i] * inv_sqrt_c);
  float z = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i) { sm[i] = __expf(p[i] * inv_sqrt_c - mx); z += sm[i]; }
  const float iz = 1.f / z;
  float acc = 0.f;
  c[0] = -TAIL;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    sm[i] *= iz;
    acc += MINBIN + (1.f - MINBIN * NB) * sm[i];
    c[i + 1] = 2.f * TAIL * acc - TAIL;
  }
  c[NB] = TAIL;
}

// x: the value the bin is searched with (forward: the input, on the width knots; inverse: the output, on the height knots)
template <bool INVERSE>
__device__ __forceinline__ void spline_setup(const float* p, float x, float inv_sqrt_c, Spline& s)
{
  float cw[NB + 1], chh[NB + 1];
  spline_knots(p, inv_sqrt_c, s.sw, cw);
  spline_knots(p + NB, inv_sqrt_c, s.sh, chh);
  s.inside = x >= -TAIL && x <= TAIL;
  int k = -1;                                   // transforms.searchsorted: #(x >= knot) - 1, last knot + 1e-6
#pragma unroll
  for (int i = 0; i <= NB; ++i) { const float kn = INVERSE ? chh[i] : cw[i]; k += (x >= (i == NB ? kn + 1e-6f : kn)) ? 1 : 0; }
  k = k < 0 ? 0 : (k > NB - 1 ? NB - 1 : k);
  s.k = k;
  float cwk = 0.f, cwk1 = 0.f, chk = 0.f, chk1 = 0.f, u0 = 0.f, u1 = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i == k) {
      cwk = cw[i]; cwk1 = cw[i + 1]; chk = chh[i]; chk1 = chh[i + 1];
      u0 = i >= 1 ? p[2 * NB + (i >= 1 ? i - 1 : 0)] : 0.f;
      u1 = i + 1 <= NB - 1 ? p[2 * NB + (i + 1 <= NB - 1 ? i : 0)] : 0.f;
    }
  // derivatives = min + softplus(.), the two outer ones pinned: min + softplus(log(exp(1 - min) - 1)) == 1 (transforms.py:66-69)
  s.u0 = u0; s.u1 = u1;
  s.d0 = k >= 1 ? MINDER + softplus_f(u0) : 1.0f;
  s.d1 = k + 1 <= NB - 1 ? MINDER + softplus_f(u1) : 1.0f;
  s.cwk = cwk; s.chk = chk;
  s.w = cwk1 - cwk; s.h = chk1 - chk;
  s.delta = s.h / s.w;
  s.theta = (x - cwk) / s.w;
}
__device__ __forceinline__ void spline_eval(const Spline& s, float x, float& y, float& lad)
{
  if (!s.inside) { y = x; lad = 0.f; return; }
  const float th = s.theta, t1 = th * (1.f - th);
  const float num = s.h * (s.delta * th * th + s.d0 * t1);
  const float den = s.delta + (s.d0 + s.d1 - 2.f * s.delta) * t1;
  y = s.chk + num / den;
  const float q2 = s.d1 * th * th + 2.f * s.delta * t1 + s.d0 * (1.f - th) * (1.f - th);
  lad = logf(s.delta * s.delta * q2) - 2.f * logf(den);
}
__device__ __forceinline__ float spline_inverse(const Spline& s, float y)
{
  if (!s.inside) return y;
  const float yy = y - s.chk, sd = s.d0 + s.d1 - 2.f * s.delta;
  const float a = yy * sd + s.h * (s.delta - s.d0), b = s.h * s.d0 - yy * sd, c = -s.delta * yy;
  const float root = (2.f * c) / (-b - sqrtf(b * b - 4.f * a * c));
  return root * s.w + s.cwk;
}
// gradients of (y, lad) w.r.t. the input and the 29 raw parameters, given gy = dL/dy and gl = dL/dlad
__device__ __forceinline__ void spline_grad(const Spline& s, float gy, float gl, float inv_sqrt_c, float& gx, float (&gp)[32])
{
#pragma unroll
  for (int i = 0; i < 32; ++i) gp[i] = 0.f;
  if (!s.inside) { gx = gy; return; }
  const float th = s.theta, t1 = th * (1.f - th), omt = 1.f - th;
  const float sd = s.d0 + s.d1 - 2.f * s.delta;
  const float num = s.h * (s.delta * th * th + s.d0 * t1);
  const float den = s.delta + sd * t1;
  const float q2 = s.d1 * th * th + 2.f * s.delta * t1 + s.d0 * omt * omt;
  const float iden = 1.f / den, iq2 = 1.f / q2;
  // partials of num, den, q2
  const float n_th = s.h * (2.f * s.delta * th + s.d0 * (1.f - 2.f * th)), n_de = s.h * th * th, n_h = s.delta * th * th + s.d0 * t1, n_d0 = s.h * t1;
  const float e_th = sd * (1.f - 2.f * th), e_de = 1.f - 2.f * t1, e_d = t1;
  const float q_th = 2.f * s.d1 * th + 2.f * s.delta * (1.f - 2.f * th) - 2.f * s.d0 * omt, q_de = 2.f * t1, q_d0 = omt * omt, q_d1 = th * th;
  auto dy = [&](float nq, float eq) { return (nq * den - num * eq) * iden * iden; };
  const float G_th = gy * dy(n_th, e_th) + gl * (q_th * iq2 - 2.f * e_th * iden);
  const float G_de = gy * dy(n_de, e_de) + gl * (2.f / s.delta + q_de * iq2 - 2.f * e_de * iden);
  const float G_h = gy * n_h * iden;                                 // explicit h of the numerator only
  const float G_d0 = gy * dy(n_d0, e_d) + gl * (q_d0 * iq2 - 2.f * e_d * iden);
  const float G_d1 = gy * dy(0.f, e_d) + gl * (q_d1 * iq2 - 2.f * e_d * iden);
  const float iw = 1.f / s.w;
  gx = G_th * iw;
  const float g_cw = -G_th * iw, g_w = -G_th * th * iw - G_de * s.delta * iw, g_h = G_h + G_de * iw, g_ch = gy;
  // knots: cw_k, w = cw_{k+1} - cw_k (the end knots are constants); cw_j = 2B * cumsum_j - B for 1 <= j <= K-1
  const int k = s.k;
  const float gk_w = 2.f * TAIL * (g_cw - g_w), gk1_w = 2.f * TAIL * g_w, gk_h = 2.f * TAIL * (g_ch - g_h), gk1_h = 2.f * TAIL * g_h;
  // normalised width i enters every cumsum j > i; then the softmax Jacobian and the 1/sqrt(C) scale (modules.py:801-802)
  float gsw[NB], gsh[NB], dotw = 0.f, doth = 0.f, runw = 0.f, runh = 0.f;
#pragma unroll
  for (int i = NB - 1; i >= 0; --i) {
    const int j = i + 1;                                              // gradient at cumsum_j, j in [1, K-1]
    if (j <= NB - 1) {
      if (j == k) { runw += gk_w; runh += gk_h; }
      if (j == k + 1) { runw += gk1_w; runh += gk1_h; }
    }
    gsw[i] = (1.f - MINBIN * NB) * runw; gsh[i] = (1.f - MINBIN * NB) * runh;
    dotw += gsw[i] * s.sw[i]; doth += gsh[i] * s.sh[i];
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) { gp[i] = s.sw[i] * (gsw[i] - dotw) * inv_sqrt_c; gp[NB + i] = s.sh[i] * (gsh[i] - doth) * inv_sqrt_c; }
  const float g_u0 = k >= 1 ? G_d0 * sigmoidf_(s.u0) : 0.f, g_u1 = k + 1 <= NB - 1 ? G_d1 * sigmoidf_(s.u1) : 0.f;
#pragma unroll
  for (int i = 0; i < NB - 1; ++i) gp[2 * NB + i] = (i == k - 1 ? g_u0 : 0.f) + (i == k ? g_u1 : 0.f);
}

constexpr int SPR = 32;                       // rows per workgroup of the spline kernels
constexpr int HP = PC + 1;                    // LDS pitch of the h tile (conflict-free column walks)
constexpr int SPL_PW = NPAR * PC + NPAR;      // floats per partial row of the spline backward: d Wp [NPAR][PC] | d bp [NPAR]

// add `v` (per lane of wave 0, lanes < SPR) to acc[utt] with one atomic when the rows share an utterance
__device__ __forceinline__ void utt_accumulate(float* acc, int u, bool on, float v)
{
  const int u0 = __shfl(u, 0);
  v = on ? v : 0.f;
  if (__all(u == u0 || !on)) { const float s = wave_sum(v); if ((threadIdx.x & 63) == 0 && s != 0.f) atomicAdd(acc + u0, s); }
  else if (on && v != 0.f) atomicAdd(acc + u, v);
}

// params = (Wp h + bp) * mask; x1' = spline(x1); z_out = [x0, x1'] * mask (channels swapped when flip); acc[utt] += sign * lad
__global__ __launch_bounds__(256) void gt_convflow_spline_fwd_kernel(
    const float* __restrict__ h, const float* __restrict__ Wp, const float* __restrict__ bp, const float* __restrict__ zin,
    const float* __restrict__ rowmask, const int32_t* __restrict__ utt, float* __restrict__ zout, float* __restrict__ par,
    float* __restrict__ acc, float sign, int flip, int R)
{
  __shared__ float Wt[PC][32];                // transposed, zero padded to 32 outputs
  __shared__ float Hs[SPR][HP];
  __shared__ float Ps[SPR][32];
  const int tid = threadIdx.x, m0 = blockIdx.x * SPR;
  for (int q = tid; q < PC * 32; q += 256) { const int k = q >> 5, o = q & 31; Wt[k][o] = o < NPAR ? Wp[o * PC + k] : 0.f; }
  for (int q = tid; q < SPR * PC; q += 256) { const int r = q / PC, k = q - r * PC; Hs[r][k] = (m0 + r < R) ? h[(size_t)(m0 + r) * PC + k] : 0.f; }
  __syncthreads();
  {
    const int r = tid & 31, og = tid >> 5;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k = 0; k < PC; ++k) {
      const float hv = Hs[r][k];
      const float4 wv = *reinterpret_cast<const float4*>(&Wt[k][og * 4]);
      a0 += hv * wv.x; a1 += hv * wv.y; a2 += hv * wv.z; a3 += hv * wv.w;
    }
    const float mk = (m0 + r < R) ? rowmask[m0 + r] : 0.f;
    const float acc4[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int o = og * 4 + j; Ps[r][o] = o < NPAR ? (acc4[j] + bp[o]) * mk : 0.f; }
  }
  __syncthreads();
  for (int q = tid; q < SPR * 32; q += 256) { const int r = q >> 5; if (m0 + r < R) par[(size_t)(m0 + r) * 32 + (q & 31)] = Ps[r][q & 31]; }
  if (tid < 64) {
    const int r = tid, m = m0 + r;
    const bool row = r < SPR && m < R;
    const bool on = row && rowmask[m] != 0.f;
    float lad = 0.f;
    if (row) {
      float y0 = 0.f, y1 = 0.f;
      if (on) {
        Spline s;
        const float x1 = zin[(size_t)m * 2 + 1];
        float p[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) p[i] = Ps[r][i];
        spline_setup<false>(p, x1, rsqrtf((float)PC), s);
        spline_eval(s, x1, y1, lad);
        y0 = zin[(size_t)m * 2];
      }
      zout[(size_t)m * 2 + (flip ? 1 : 0)] = y0;
      zout[(size_t)m * 2 + (flip ? 0 : 1)] = y1;
    }
    utt_accumulate(acc, row ? utt[m] : -1, on, sign * lad);
  }
}

// backward: dz_out, gacc[utt] (gradient at acc) -> dh, dWp / dbp (accumulate), dz_in
__global__ __launch_bounds__(256) void gt_convflow_spline_bwd_kernel(
    const float* __restrict__ h, const float* __restrict__ Wp, const float* __restrict__ par, const float* __restrict__ zin,
    const float* __restrict__ dzout, const float* __restrict__ gacc, const float* __restrict__ rowmask,
    const int32_t* __restrict__ utt, float* __restrict__ dh, float* __restrict__ dWp, float* __restrict__ dbp,
    float* __restrict__ partials, float* __restrict__ dzin, float sign, int flip, int R)
{
  // partials: row blockIdx.x of [gridDim.x][SPL_PW] receives this workgroup's d Wp | d bp sums (plain stores; gt_param_partials_reduce adds
  // them up) instead of NPAR * PC + NPAR atomics per workgroup — 556 workgroups on 5 597 addresses at cfg 5's frame rows: 46 us per launch
  float* prow = partials ? partials + (size_t)blockIdx.x * SPL_PW : nullptr;
  __shared__ float Ws[32][HP];                // [o][k], rows >= NPAR zero
  __shared__ float Hs[SPR][HP];
  __shared__ float Gs[SPR][32];               // d params per row (masked)
  const int tid = threadIdx.x, m0 = blockIdx.x * SPR;
  for (int q = tid; q < 32 * PC; q += 256) { const int o = q / PC, k = q - o * PC; Ws[o][k] = o < NPAR ? Wp[o * PC + k] : 0.f; }
  for (int q = tid; q < SPR * PC; q += 256) { const int r = q / PC, k = q - r * PC; Hs[r][k] = (m0 + r < R) ? h[(size_t)(m0 + r) * PC + k] : 0.f; }
  if (tid < SPR) {
    const int r = tid, m = m0 + r;
    float gp[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) gp[i] = 0.f;
    if (m < R) {
      const bool on = rowmask[m] != 0.f;
      float g0 = 0.f, g1 = 0.f;
      if (on) {
        Spline s;
        const float x1 = zin[(size_t)m * 2 + 1];
        float p[32];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 v4 = *reinterpret_cast<const float4*>(par + (size_t)m * 32 + 4 * i);
          p[4 * i] = v4.x; p[4 * i + 1] = v4.y; p[4 * i + 2] = v4.z; p[4 * i + 3] = v4.w;
        }
        spline_setup<false>(p, x1, rsqrtf((float)PC), s);
        const float gy = dzout[(size_t)m * 2 + (flip ? 0 : 1)];
        spline_grad(s, gy, sign * gacc[utt[m]], rsqrtf((float)PC), g1, gp);
        g0 = dzout[(size_t)m * 2 + (flip ? 1 : 0)];
      }
      dzin[(size_t)m * 2] = g0; dzin[(size_t)m * 2 + 1] = g1;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) Gs[r][i] = gp[i];
  }
  __syncthreads();
  // dh[r][k] = sum_o G[r][o] Wp[o][k]
  for (int q = tid; q < SPR * PC; q += 256) {
    const int r = q / PC, k = q - r * PC;
    if (m0 + r >= R) continue;
    float a = 0.f;
#pragma unroll
    for (int o = 0; o < NPAR; ++o) a += Gs[r][o] * Ws[o][k];
    dh[(size_t)(m0 + r) * PC + k] = a;
  }
  // dWp[o][k] += sum_r G[r][o] h[r][k];  dbp[o] += sum_r G[r][o]
  for (int q = tid; q < NPAR * PC; q += 256) {
    const int o = q / PC, k = q - o * PC;
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < SPR; ++r) a += Gs[r][o] * Hs[r][k];
    if (prow) prow[q] = a;
    else if (a != 0.f) atomicAdd(dWp + q, a);
  }
  if (tid < NPAR) {
    float a = 0.f;
    for (int r = 0; r < SPR; ++r) a += Gs[r][tid];
    if (prow) prow[NPAR * PC + tid] = a;
    else if (a != 0.f) atomicAdd(dbp + tid, a);
  }
}

// inverse spline for synthesis (transforms.py:152-180): params from h as above, x1 = spline^-1(y1); no log-det
__global__ __launch_bounds__(256) void gt_convflow_spline_inv_kernel(
    const float* __restrict__ h, const float* __restrict__ Wp, const float* __restrict__ bp, const float* __restrict__ zin,
    const float* __restrict__ rowmask, float* __restrict__ zout, int R)
{
  __shared__ float Wt[PC][32];
  __shared__ float Hs[SPR][HP];
  __shared__ float Ps[SPR][32];
  const int tid = threadIdx.x, m0 = blockIdx.x * SPR;
  for (int q = tid; q < PC * 32; q += 256) { const int k = q >> 5, o = q & 31; Wt[k][o] = o < NPAR ? Wp[o * PC + k] : 0.f; }
  for (int q = tid; q < SPR * PC; q += 256) { const int r = q / PC, k = q - r * PC; Hs[r][k] = (m0 + r < R) ? h[(size_t)(m0 + r) * PC + k] : 0.f; }
  __syncthreads();
  {
    const int r = tid & 31, og = tid >> 5;
    float a[4] = {};
    for (int k = 0; k < PC; ++k) {
      const float hv = Hs[r][k];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] += hv * Wt[k][og * 4 + j];
    }
    const float mk = (m0 + r < R) ? rowmask[m0 + r] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int o = og * 4 + j; Ps[r][o] = o < NPAR ? (a[j] + bp[o]) * mk : 0.f; }
  }
  __syncthreads();
  if (tid < SPR && m0 + tid < R) {
    const int m = m0 + tid;
    float x0 = 0.f, x1 = 0.f;
    if (rowmask[m] != 0.f) {
      x0 = zin[(size_t)m * 2];
      const float y = zin[(size_t)m * 2 + 1];
      float p[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) p[i] = Ps[tid][i];
      Spline s;
      spline_setup<true>(p, y, rsqrtf((float)PC), s);
      x1 = spline_inverse(s, y);
    }
    zout[(size_t)m * 2] = x0; zout[(size_t)m * 2 + 1] = x1;
  }
}

// ------------------------------------------------------------------------------------------------ small row kernels on [R, 2]
// ElementwiseAffine (modules.py:750-756): y = (x * exp(ls) + t) * mask; acc[utt] += sign * (ls0 + ls1) on valid rows
__global__ __launch_bounds__(256) void gt_ea_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ls, const float* __restrict__ tr,
                                                        const float* __restrict__ rowmask, const int32_t* __restrict__ utt,
                                                        float* __restrict__ y, float* __restrict__ acc, float sign, int reverse, int R)
{
  const int m = blockIdx.x * 256 + threadIdx.x;
  const bool row = m < R;
  const bool on = row && rowmask[m] != 0.f;
  if (row) {
    float y0 = 0.f, y1 = 0.f;
    if (on) {
      if (reverse) { y0 = (x[2 * m] - tr[0]) * __expf(-ls[0]); y1 = (x[2 * m + 1] - tr[1]) * __expf(-ls[1]); }
      else         { y0 = x[2 * m] * __expf(ls[0]) + tr[0];   y1 = x[2 * m + 1] * __expf(ls[1]) + tr[1]; }
    }
    y[2 * m] = y0; y[2 * m + 1] = y1;
  }
  if (acc) utt_accumulate(acc, row ? utt[m] : -1, on, sign * (ls[0] + ls[1]));
}
__global__ __launch_bounds__(256) void gt_ea_bwd_kernel(const float* __restrict__ x, const float* __restrict__ ls, const float* __restrict__ dy,
                                                        const float* __restrict__ gacc, const float* __restrict__ rowmask,
                                                        const int32_t* __restrict__ utt, float* __restrict__ dx, float* __restrict__ dls,
                                                        float* __restrict__ dtr, float sign, int R)
{
  const int m = blockIdx.x * 256 + threadIdx.x;
  float a0 = 0.f, a1 = 0.f, t0 = 0.f, t1 = 0.f;
  if (m < R) {
    float g0 = 0.f, g1 = 0.f;
    if (rowmask[m] != 0.f) {
      const float e0 = __expf(ls[0]), e1 = __expf(ls[1]), ga = sign * gacc[utt[m]];
      g0 = dy[2 * m] * e0; g1 = dy[2 * m + 1] * e1;
      a0 = g0 * x[2 * m] + ga; a1 = g1 * x[2 * m + 1] + ga;
      t0 = dy[2 * m]; t1 = dy[2 * m + 1];
    }
    dx[2 * m] = g0; dx[2 * m + 1] = g1;
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); t0 = wave_sum(t0); t1 = wave_sum(t1);
  __shared__ float sm[4][4];
  if ((threadIdx.x & 63) == 0) { float* q = sm[threadIdx.x >> 6]; q[0] = a0; q[1] = a1; q[2] = t0; q[3] = t1; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float v = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
    if (v != 0.f) atomicAdd((threadIdx.x < 2 ? dls : dtr) + (threadIdx.x & 1), v);
  }
}

// StochasticDurationPredictor, between the posterior flows and the flows (models.py:299-311):
//   u = sigmoid(z_u) * mask, z0 = (w - u) * mask, z = [log(max(z0, 1e-5)) * mask, z_v]
//   acc[utt] += -0.5 (log 2pi + e_q^2) [both noise channels] - (logsigmoid(z_u) + logsigmoid(-z_u)) + log(max(z0, 1e-5))
__global__ __launch_bounds__(256) void gt_sdp_mid_fwd_kernel(const float* __restrict__ zq, const float* __restrict__ w, const float* __restrict__ eq,
                                                             const float* __restrict__ rowmask, const int32_t* __restrict__ utt,
                                                             float* __restrict__ z, float* __restrict__ acc, int R)
{
  const int m = blockIdx.x * 256 + threadIdx.x;
  const bool row = m < R;
  const bool on = row && rowmask[m] != 0.f;
  float a = 0.f;
  if (row) {
    float z0l = 0.f, zv = 0.f;
    if (on) {
      const float zu = zq[2 * m];
      zv = zq[2 * m + 1];
      const float sg = sigmoidf_(zu);
      z0l = logf(fmaxf(w[m] - sg, 1e-5f));
      const float ls_p = -softplus_f(-zu), ls_n = -softplus_f(zu);         // logsigmoid(zu), logsigmoid(-zu)
      a = -0.5f * (2.f * 1.8378770664093453f + eq[2 * m] * eq[2 * m] + eq[2 * m + 1] * eq[2 * m + 1]) - (ls_p + ls_n) + z0l;
    }
    z[2 * m] = z0l; z[2 * m + 1] = zv;
  }
  utt_accumulate(acc, row ? utt[m] : -1, on, a);
}
__global__ __launch_bounds__(256) void gt_sdp_mid_bwd_kernel(const float* __restrict__ zq, const float* __restrict__ w, const float* __restrict__ dz,
                                                             const float* __restrict__ gacc, const float* __restrict__ rowmask,
                                                             const int32_t* __restrict__ utt, float* __restrict__ dzq, int R)
{
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= R) return;
  float gu = 0.f, gv = 0.f;
  if (rowmask[m] != 0.f) {
    const float zu = zq[2 * m], sg = sigmoidf_(zu), z0 = w[m] - sg, ga = gacc[utt[m]];
    const float dz0l = dz[2 * m] + ga;
    const float dz0 = z0 > 1e-5f ? dz0l / z0 : 0.f;
    gu = -dz0 * sg * (1.f - sg) - ga * (1.f - 2.f * sg);
    gv = dz[2 * m + 1];
  }
  dzq[2 * m] = gu; dzq[2 * m + 1] = gv;
}
// acc[utt] += 0.5 (log 2pi + z^2) over both channels; backward dz = gacc * z
__global__ __launch_bounds__(256) void gt_nll_gauss_fwd_kernel(const float* __restrict__ z, const float* __restrict__ rowmask,
                                                               const int32_t* __restrict__ utt, float* __restrict__ acc, int R)
{
  const int m = blockIdx.x * 256 + threadIdx.x;
  const bool row = m < R;
  const bool on = row && rowmask[m] != 0.f;
  float a = 0.f;
  if (on) a = 0.5f * (2.f * 1.8378770664093453f + z[2 * m] * z[2 * m] + z[2 * m + 1] * z[2 * m + 1]);
  utt_accumulate(acc, row ? utt[m] : -1, on, a);
}
__global__ __launch_bounds__(256) void gt_nll_gauss_bwd_kernel(const float* __restrict__ z, const float* __restrict__ gacc,
                                                               const float* __restrict__ rowmask, const int32_t* __restrict__ utt,
                                                               float* __restrict__ dz, int R)
{
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= R) return;
  const float ga = rowmask[m] != 0.f ? gacc[utt[m]] : 0.f;
  dz[2 * m] = ga * z[2 * m]; dz[2 * m + 1] = ga * z[2 * m + 1];
}

// x_feature = x @ attn (models.py:1094) with the hard MAS path as a row gather: frame row m of utterance b <- token row
__global__ __launch_bounds__(256) void gt_rows_gather_tokens_kernel(const bf16_t* __restrict__ xs, int ldx, const int32_t* __restrict__ f2t, int Ty,
                                                                    const int32_t* __restrict__ row0x, int Tpx, const int32_t* __restrict__ utt,
                                                                    const int32_t* __restrict__ row0f, int Tpf, const float* __restrict__ rowmask,
                                                                    bf16_t* __restrict__ out, int R, int C8)
{
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= R * C8) return;
  const int m = q / C8, c = (q - m * C8) * 8;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (rowmask[m] != 0.f) {
    const int b = utt[m];
    const int t = m - gt_row_base(row0f, b, Tpf) - GT_HALO;
    const int tok = f2t[(size_t)b * Ty + t];
    v = *reinterpret_cast<const uint4*>(xs + (size_t)(gt_row_base(row0x, b, Tpx) + GT_HALO + tok) * ldx + c);
  }
  *reinterpret_cast<uint4*>(out + (size_t)m * (C8 * 8) + c) = v;
}

__global__ __launch_bounds__(256) void gt_rows_split3_kernel(const void* __restrict__ in, int ldi, int is_f32, bf16_t* __restrict__ out, int ldo,
                                                             const float* __restrict__ rowmask, int R, int C)
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * C) return;
  const int m = idx / C, c = idx - m * C;
  float v = is_f32 ? static_cast<const float*>(in)[(size_t)m * ldi + c] : bf2f(static_cast<const bf16_t*>(in)[(size_t)m * ldi + c]);
  if (rowmask) v *= rowmask[m];
  const bf16_t hi = f2bf(v);
  bf16_t* o = out + (size_t)m * ldo;
  o[c] = hi; o[C + c] = hi; o[2 * C + c] = f2bf(v - bf2f(hi));
}

#define GT_ST(s) static_cast<hipStream_t>(s)
#define GT_RET() return gt_launch_status(__func__)
inline int wg_rows(int R) { return (R + 4 * RPW - 1) / (4 * RPW); }
inline int wg_rows_b(int R) { return (R + 4 * RPB - 1) / (4 * RPB); }
inline void fill_drop(float p, uint32_t& th, float& sc) { th = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u; sc = p > 0.f ? 1.0f / (1.0f - p) : 1.0f; }

}  // namespace

extern "C" int gt_rows_split3(const void* in, int ldi, int is_f32, void* out, int ldo, const float* rowmask, int R, int C, void* stream)
{
  if (!in || !out || R <= 0 || C <= 0 || ldo < 3 * C) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_rows_split3_kernel, dim3((R * C + 255) / 256), dim3(256), 0, GT_ST(stream), in, ldi, is_f32, static_cast<bf16_t*>(out), ldo, rowmask, R, C);
  GT_RET();
}
extern "C" int gt_dds_sep_fwd(const float* x, int ldx, const float* w, const float* b, const float* gamma, const float* beta,
                              const int32_t* utt, const float* rowmask, void* a1_bf16, int lda, int R, int C, int dilation, float eps, void* stream)
{
  if (!x || !w || !b || !gamma || !beta || !utt || !rowmask || !a1_bf16 || R <= 0 || dilation <= 0 || lda < 3 * C) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_dds_sep_fwd_kernel, dim3(wg_rows(R)), dim3(256), 0, GT_ST(stream), x, ldx, w, b, gamma, beta, utt, rowmask,
                     static_cast<bf16_t*>(a1_bf16), lda, R, dilation, eps);
  GT_RET();
}
extern "C" int gt_dds_out_fwd(const float* h2, const float* x, int ldx, const float* gamma, const float* beta, const float* rowmask,
                              float* out, void* out_bf16, int R, int C, float eps, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream)
{
  if (!h2 || !x || !gamma || !beta || !rowmask || !out || R <= 0 || drop_p < 0.f || drop_p >= 1.f) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  uint32_t th; float sc; fill_drop(drop_p, th, sc);
  hipLaunchKernelGGL(gt_dds_out_fwd_kernel, dim3(wg_rows(R)), dim3(256), 0, GT_ST(stream), h2, x, ldx, gamma, beta, rowmask, out,
                     static_cast<bf16_t*>(out_bf16), R, eps, th, seed, seed_dev, sc);
  GT_RET();
}
extern "C" int gt_dds_bwd_partial_rows(int R) { return R > 0 ? wg_rows_b(R) : 0; }
extern "C" int gt_dds_out_bwd(const float* h2, const float* dy, const float* gamma, const float* beta, const float* rowmask,
                              void* dh2_bf16, float* dgamma, float* dbeta, float* partials, int R, int C, float eps, float drop_p, uint32_t seed,
                              const uint32_t* seed_dev, void* stream)
{
  if (!h2 || !dy || !gamma || !beta || !rowmask || !dh2_bf16 || (!partials && (!dgamma || !dbeta)) || R <= 0 || drop_p < 0.f || drop_p >= 1.f) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  uint32_t th; float sc; fill_drop(drop_p, th, sc);
  hipLaunchKernelGGL(gt_dds_out_bwd_kernel, dim3(wg_rows_b(R)), dim3(256), 0, GT_ST(stream), h2, dy, gamma, beta, rowmask,
                     static_cast<bf16_t*>(dh2_bf16), dgamma, dbeta, partials, R, eps, th, seed, seed_dev, sc);
  GT_RET();
}
extern "C" int gt_dds_sep_bwd(const float* x, int ldx, const float* w, const float* b, const float* gamma, const float* beta,
                              const int32_t* utt, const float* rowmask, const float* da1, float* dh1, float* dgamma, float* dbeta,
                              float* partials, int R, int C, int dilation, float eps, void* stream)
{
  if (!x || !w || !b || !gamma || !beta || !utt || !rowmask || !da1 || !dh1 || (!partials && (!dgamma || !dbeta)) || R <= 0 || dilation <= 0) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_dds_sep_bwd_kernel, dim3(wg_rows_b(R)), dim3(256), 0, GT_ST(stream), x, ldx, w, b, gamma, beta, utt, rowmask, da1, dh1,
                     dgamma, dbeta, partials, R, dilation, eps);
  GT_RET();
}
extern "C" int gt_dds_dw_bwd(const float* x, int ldx, const float* dh1, const float* dy, const float* w, const int32_t* utt,
                             const float* rowmask, float* dx, float* dw, float* db, float* partials, int R, int C, int dilation, void* stream)
{
  if (!x || !dh1 || !dy || !w || !utt || !rowmask || !dx || (!partials && (!dw || !db)) || R <= 0 || dilation <= 0) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_dds_dw_bwd_kernel, dim3(wg_rows_b(R)), dim3(256), 0, GT_ST(stream), x, ldx, dh1, dy, w, utt, rowmask, dx, dw, db, partials, R, dilation);
  GT_RET();
}
extern "C" int gt_convflow_pre_fwd(const float* z, int ldz, const float* w_pre, const float* b_pre, const float* g1, const float* g2,
                                   const float* rowmask, float* out, int R, int C, void* stream)
{
  if (!z || !w_pre || !b_pre || !rowmask || !out || R <= 0 || ldz <= 0) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_convflow_pre_fwd_kernel, dim3(wg_rows(R)), dim3(256), 0, GT_ST(stream), z, ldz, w_pre, b_pre, g1, g2, rowmask, out, R);
  GT_RET();
}
extern "C" int gt_convflow_pre_bwd(const float* dx0, const float* z, int ldz, const float* w_pre, const float* rowmask,
                                   float* dw_pre, float* db_pre, float* dz, int lddz, float* dg, int R, int C, void* stream)
{
  if (!dx0 || !z || !w_pre || !rowmask || !dw_pre || !db_pre || R <= 0) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_convflow_pre_bwd_kernel, dim3(wg_rows_b(R)), dim3(256), 0, GT_ST(stream), dx0, z, ldz, w_pre, rowmask, dw_pre, db_pre,
                     dz, lddz, dg, R);
  GT_RET();
}
extern "C" int gt_convflow_spline_fwd(const float* h, const float* Wp, const float* bp, const float* z_in, const float* rowmask,
                                      const int32_t* utt, float* z_out, float* params, float* acc, float sign, int flip, int R, int C, void* stream)
{
  if (!h || !Wp || !bp || !z_in || !rowmask || !utt || !z_out || !params || !acc || R <= 0) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_convflow_spline_fwd_kernel, dim3((R + SPR - 1) / SPR), dim3(256), 0, GT_ST(stream), h, Wp, bp, z_in, rowmask, utt,
                     z_out, params, acc, sign, flip, R);
  GT_RET();
}
extern "C" int gt_convflow_spline_partial_rows(int R) { return R > 0 ? (R + SPR - 1) / SPR : 0; }
extern "C" int gt_convflow_spline_partial_width(void) { return SPL_PW; }
extern "C" int gt_convflow_spline_bwd(const float* h, const float* Wp, const float* params, const float* z_in, const float* dz_out,
                                      const float* gacc, const float* rowmask, const int32_t* utt, float* dh, float* dWp, float* dbp,
                                      float* partials, float* dz_in, float sign, int flip, int R, int C, void* stream)
{
  if (!h || !Wp || !params || !z_in || !dz_out || !gacc || !rowmask || !utt || !dh || (!partials && (!dWp || !dbp)) || !dz_in || R <= 0) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_convflow_spline_bwd_kernel, dim3((R + SPR - 1) / SPR), dim3(256), 0, GT_ST(stream), h, Wp, params, z_in, dz_out, gacc,
                     rowmask, utt, dh, dWp, dbp, partials, dz_in, sign, flip, R);
  GT_RET();
}
extern "C" int gt_convflow_spline_inv(const float* h, const float* Wp, const float* bp, const float* z_in, const float* rowmask,
                                      float* z_out, int R, int C, void* stream)
{
  if (!h || !Wp || !bp || !z_in || !rowmask || !z_out || R <= 0) return GT_E_INVAL;
  if (C != PC) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_convflow_spline_inv_kernel, dim3((R + SPR - 1) / SPR), dim3(256), 0, GT_ST(stream), h, Wp, bp, z_in, rowmask, z_out, R);
  GT_RET();
}
extern "C" int gt_ea_fwd(const float* x, const float* log_scale, const float* translation, const float* rowmask, const int32_t* utt,
                         float* y, float* acc, float sign, int reverse, int R, void* stream)
{
  if (!x || !log_scale || !translation || !rowmask || !utt || !y || R <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_ea_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, GT_ST(stream), x, log_scale, translation, rowmask, utt, y, acc, sign, reverse, R);
  GT_RET();
}
extern "C" int gt_ea_bwd(const float* x, const float* log_scale, const float* dy, const float* gacc, const float* rowmask, const int32_t* utt,
                         float* dx, float* dlog_scale, float* dtranslation, float sign, int R, void* stream)
{
  if (!x || !log_scale || !dy || !gacc || !rowmask || !utt || !dx || !dlog_scale || !dtranslation || R <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_ea_bwd_kernel, dim3((R + 255) / 256), dim3(256), 0, GT_ST(stream), x, log_scale, dy, gacc, rowmask, utt, dx, dlog_scale,
                     dtranslation, sign, R);
  GT_RET();
}
extern "C" int gt_sdp_mid_fwd(const float* z_q, const float* w, const float* e_q, const float* rowmask, const int32_t* utt, float* z, float* acc,
                              int R, void* stream)
{
  if (!z_q || !w || !e_q || !rowmask || !utt || !z || !acc || R <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_sdp_mid_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, GT_ST(stream), z_q, w, e_q, rowmask, utt, z, acc, R);
  GT_RET();
}
extern "C" int gt_sdp_mid_bwd(const float* z_q, const float* w, const float* dz, const float* gacc, const float* rowmask, const int32_t* utt,
                              float* dz_q, int R, void* stream)
{
  if (!z_q || !w || !dz || !gacc || !rowmask || !utt || !dz_q || R <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_sdp_mid_bwd_kernel, dim3((R + 255) / 256), dim3(256), 0, GT_ST(stream), z_q, w, dz, gacc, rowmask, utt, dz_q, R);
  GT_RET();
}
extern "C" int gt_nll_gauss_fwd(const float* z, const float* rowmask, const int32_t* utt, float* acc, int R, void* stream)
{
  if (!z || !rowmask || !utt || !acc || R <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_nll_gauss_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, GT_ST(stream), z, rowmask, utt, acc, R);
  GT_RET();
}
extern "C" int gt_nll_gauss_bwd(const float* z, const float* gacc, const float* rowmask, const int32_t* utt, float* dz, int R, void* stream)
{
  if (!z || !gacc || !rowmask || !utt || !dz || R <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_nll_gauss_bwd_kernel, dim3((R + 255) / 256), dim3(256), 0, GT_ST(stream), z, gacc, rowmask, utt, dz, R);
  GT_RET();
}
extern "C" int gt_rows_gather_tokens(const void* x_rows, int ldx, const int32_t* frame2token, int Ty, const int32_t* row0_x, int Tp_x,
                                     const int32_t* utt_f, const int32_t* row0_f, int Tp_f, const float* rowmask_f, void* out, int R_f, int C,
                                     void* stream)
{
  if (!x_rows || !frame2token || !utt_f || !rowmask_f || !out || R_f <= 0 || C <= 0 || (C & 7) || (ldx & 7)) return GT_E_INVAL;
  const int C8 = C / 8;
  hipLaunchKernelGGL(gt_rows_gather_tokens_kernel, dim3((R_f * C8 + 255) / 256), dim3(256), 0, GT_ST(stream), static_cast<const bf16_t*>(x_rows), ldx,
                     frame2token, Ty, row0_x, Tp_x, utt_f, row0_f, Tp_f, rowmask_f, static_cast<bf16_t*>(out), R_f, C8);
  GT_RET();
}
