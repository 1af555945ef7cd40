// Text-encoder / glue kernels for gfx950 in the rows layout (see glowtts_hip.h):
//   channel LayerNorm (+residual, dropout, relu)   modules.py:26-44, attentions.py:79-84, modules.py:95-102
//   relative-position multi-head attention          attentions.py:241-336 (9-diagonal band form, SURVEY App. A iii)
//   embedding * sqrt(H)                             models.py:693
//   logp lattice                                    models.py:1076-1082
//   prior expansion (gather) + mle loss             models.py:1118-1119, commons.py:28-33
// fp32 math throughout; bf16 only for tensors that feed MFMA GEMMs.
#include <stdlib.h>
#include "common.h"
#include "../../include/glowtts_hip.h"
#include "internal.h"

namespace {

constexpr int HALO = GT_HALO;

// ------------------------------------------------------------------ LayerNorm over channels
// s = a + drop_in(y);  n = (s-mean)*rstd*gamma + beta;  o = drop_out(relu?(n));
// out_f32 = o*mask, out_bf16 = o*mask.  One wave per row, C <= 256 (4 values per lane).
struct LnArgs {
  const float* a; const bf16_t* y; int ldy;
  const float* gamma; const float* beta;
  const float* rowmask;
  float* out_f32; bf16_t* out_bf16; int ldo;
  float* mean; float* rstd;
  int R, C; float eps;
  uint32_t din_thresh, din_seed; float din_scale;
  uint32_t dout_thresh, dout_seed; float dout_scale;
  int relu;
  const uint32_t* seed_dev;                    // optional device word XOR-ed into both seeds (graph replay)
};

__global__ __launch_bounds__(256) void gt_layernorm_fwd_kernel(LnArgs p)
{
  if (p.seed_dev) { const uint32_t x = *p.seed_dev; p.din_seed ^= x; p.dout_seed ^= x; }
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= p.R) return;
  float s[4]; float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    float v = 0.f;
    if (c < p.C) {
      if (p.a) v = p.a[(size_t)m * p.C + c];
      if (p.y) {
        float yy = bf2f(p.y[(size_t)m * p.ldy + c]);
        if (p.din_thresh) yy = drop_keep(p.din_seed, m, c, p.din_thresh) ? yy * p.din_scale : 0.f;
        v += yy;
      }
      sum += v;
    }
    s[k] = v;
  }
  const float mean = wave_sum(sum) / (float)p.C;
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int c = lane + 64 * k; if (c < p.C) { const float d = s[k] - mean; var += d * d; } }
  const float rstd = rsqrtf(wave_sum(var) / (float)p.C + p.eps);
  if (lane == 0) { p.mean[m] = mean; p.rstd[m] = rstd; }
  const float rm = p.rowmask ? p.rowmask[m] : 1.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    if (c < p.C) {
      float o = (s[k] - mean) * rstd * p.gamma[c] + p.beta[c];
      if (p.relu) o = fmaxf(o, 0.f);
      if (p.dout_thresh) o = drop_keep(p.dout_seed, m, c, p.dout_thresh) ? o * p.dout_scale : 0.f;
      o *= rm;
      if (p.out_f32) p.out_f32[(size_t)m * p.C + c] = o;
      if (p.out_bf16) p.out_bf16[(size_t)m * p.ldo + c] = f2bf(o);
    }
  }
}

struct LnBwdArgs {
  LnArgs f;                                    // forward description (inputs, mean/rstd, flags)
  const float* dout_f32; const bf16_t* dout_bf16; int lddo;
  float* da; bf16_t* dy; int lddy;             // gradients wrt a (fp32) and y (bf16, dropout replayed)
  float* dgamma; float* dbeta;                 // accumulated (atomics)
  float* partials;                             // or: row blockIdx.x of [gridDim.x][2 C] receives this workgroup's sums (no atomics)
  int rows_per_block;
};

// 16 waves, one row per wave at a time; gamma/beta partials are folded in LDS so that each channel gets ONE
// atomic per workgroup (same-address float atomics serialise at L2, ~50 ns each: 32 rows per workgroup keeps both
// the atomic chain and the per-wave row loop short).
template <int LNB_WAVES>
__global__ __launch_bounds__(64 * LNB_WAVES) void gt_layernorm_bwd_kernel(LnBwdArgs q)
{
  if (q.f.seed_dev) { const uint32_t x = *q.f.seed_dev; q.f.din_seed ^= x; q.f.dout_seed ^= x; }
  const LnArgs& p = q.f;
  __shared__ float sg[LNB_WAVES][256], sb[LNB_WAVES][256];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float accg[4] = {0, 0, 0, 0}, accb[4] = {0, 0, 0, 0};
  const int m0 = blockIdx.x * q.rows_per_block, m1 = min(p.R, m0 + q.rows_per_block);
  for (int m = m0 + w; m < m1; m += LNB_WAVES) {
    const float mean = p.mean[m], rstd = p.rstd[m];
    const float rm = p.rowmask ? p.rowmask[m] : 1.0f;
    float xh[4], dn[4]; bool keep_in[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane + 64 * k;
      xh[k] = 0.f; dn[k] = 0.f; keep_in[k] = true;
      if (c < p.C) {
        float v = p.a ? p.a[(size_t)m * p.C + c] : 0.f;
        if (p.y) {
          const bf16_t yraw = p.y[(size_t)m * p.ldy + c];
          float yy = bf2f(yraw);
          if (p.din_thresh) { keep_in[k] = drop_keep(p.din_seed, m, c, p.din_thresh); yy = keep_in[k] ? yy * p.din_scale : 0.f; }
          if ((p.relu & 2) && !(yraw & 0x7fff)) keep_in[k] = false;      // y is a ReLU's output: no gradient through its zeros
          v += yy;
        }
        xh[k] = (v - mean) * rstd;
        float d = 0.f;
        if (q.dout_f32) d += q.dout_f32[(size_t)m * p.C + c];
        if (q.dout_bf16) d += bf2f(q.dout_bf16[(size_t)m * q.lddo + c]);
        d *= rm;
        if (p.dout_thresh) d = drop_keep(p.dout_seed, m, c, p.dout_thresh) ? d * p.dout_scale : 0.f;
        if ((p.relu & 1) && (xh[k] * p.gamma[c] + p.beta[c]) <= 0.f) d = 0.f;
        accg[k] += d * xh[k]; accb[k] += d;
        dn[k] = d * p.gamma[c];
        s1 += dn[k]; s2 += dn[k] * xh[k];
      }
    }
    s1 = wave_sum(s1) / (float)p.C; s2 = wave_sum(s2) / (float)p.C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane + 64 * k;
      if (c < p.C) {
        const float ds = rstd * (dn[k] - s1 - xh[k] * s2);
        if (q.da) q.da[(size_t)m * p.C + c] = ds;
        if (q.dy) q.dy[(size_t)m * q.lddy + c] = f2bf((p.din_thresh || (p.relu & 2)) ? (keep_in[k] ? ds * p.din_scale : 0.f) : ds);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { sg[w][lane + 64 * k] = accg[k]; sb[w][lane + 64 * k] = accb[k]; }
  __syncthreads();
  const int c = threadIdx.x;
  if (c < p.C) {
    float tg = 0.f, tb = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_WAVES; ++i) { tg += sg[i][c]; tb += sb[i][c]; }
#ifdef LNB_EXP
    if (LNB_EXP & 1) { if (tg == 1.2345f) q.dgamma[c] = tb; return; }
#endif
    if (q.partials) {
      q.partials[(size_t)blockIdx.x * 2 * p.C + c] = tg;
      q.partials[(size_t)blockIdx.x * 2 * p.C + p.C + c] = tb;
    } else {
      atomicAdd(q.dgamma + c, tg);
      atomicAdd(q.dbeta + c, tb);
    }
  }
}

// Parameter gradients from the partial rows backward launches left (gt_layernorm_bwd_partials, gt_dds_*_bwd with `partials`): the
// column sums of up to GT_PARTIALS_MAX buffers [n_rows][Ca + Cb] are ADDED to dst_a[Ca] | dst_b[Cb] — one launch at the end of a
// module's backward instead of same-address atomics from every workgroup of every one of them.
// block (x: 64-column group, y: job, z: slice of the rows — PARTIALS_ZS adders per address); thread = (column, row phase)
constexpr int PARTIALS_ZS = 4;
__global__ __launch_bounds__(256) void gt_param_partials_reduce_kernel(gt_partials_args a)
{
  __shared__ float red[4][64];
  const gt_partials_job j = a.job[blockIdx.y];
  const int W = j.Ca + j.Cb;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  if (blockIdx.x * 64 >= W) return;
  const int per = ((j.n_rows + PARTIALS_ZS - 1) / PARTIALS_ZS + 3) & ~3, r0 = blockIdx.z * per, r1 = min(j.n_rows, r0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < W) {
    const float* pp = j.partials + c;
    const size_t st = (size_t)W;
    int r = r0 + ph;
    for (; r + 12 < r1; r += 16) {
      const float v0 = pp[(size_t)r * st], v1 = pp[(size_t)(r + 4) * st], v2 = pp[(size_t)(r + 8) * st], v3 = pp[(size_t)(r + 12) * st];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; r < r1; r += 4) s0 += pp[(size_t)r * st];
  }
  red[ph][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && c < W) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    float* d = c < j.Ca ? j.dst_a + c : j.dst_b + (c - j.Ca);
    atomicAdd(d, t);
  }
}

// ------------------------------------------------------------------ relative-position attention
// q,k,v: rows [R, H*D] bf16 (head h = channels [h*D, (h+1)*D)), D <= 96, window w (2w+1 taps, shared
// by heads).  One workgroup per (query tile of 16 rows, head, utterance); K and V of the
// utterance-head staged in LDS as fp32-convertible bf16 with an odd dword pitch.
//   score[i,j] = (q_i.k_j + [|j-i|<=w] q_i.Ek[j-i+w]) / sqrt(D);  key j >= len -> -1e4
//   p = softmax_j(score);  pd = dropout(p);  out_i = sum_j pd[i,j] v_j + sum_{|j-i|<=w} pd[i,j] Ev[j-i+w]
// P (pre-dropout, fp32) is saved to Pout [B,H,T,T] for the backward pass.
constexpr int AT_QT = 16;
constexpr int AT_MAXD = 96;

__device__ __forceinline__ void at_stage(bf16_t* dst, const bf16_t* src, int ld, size_t rbase, int h, int T, int D, int KP, int tid, int nv1)
{
  for (int i = tid; i < T * (D / 2); i += 256) {
    const int j = i / (D / 2), c2 = i - j * (D / 2);
    *reinterpret_cast<uint32_t*>(dst + (size_t)j * KP + 2 * c2) = *reinterpret_cast<const uint32_t*>(src + (rbase + (j < nv1 ? j : nv1)) * ld + h * D + 2 * c2);
  }
}
__device__ __forceinline__ float at_dot(const float* qv, const bf16_t* row, int D)
{
  float s = 0.f;
  for (int c = 0; c < D; c += 2) {
    const uint32_t kk = *reinterpret_cast<const uint32_t*>(row + c);
    s += qv[c] * bf2f(kk & 0xffff) + qv[c + 1] * bf2f(kk >> 16);
  }
  return s;
}

// LDS: one [T][D+2] bf16 staging buffer (K then V), band tables, per-wave q row, [16][T] score rows.
__global__ __launch_bounds__(256) void gt_attn_fwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int ld,
    const float* __restrict__ Ek, const float* __restrict__ Ev, const int32_t* __restrict__ lens,
    bf16_t* __restrict__ out, int ldo, float* __restrict__ Pout,
    int T, int Tp, const int32_t* row0, int H, int D, int win, uint32_t drop_thresh, uint32_t drop_seed, float drop_scale, const uint32_t* __restrict__ seed_dev)
{
  if (seed_dev) drop_seed ^= *seed_dev;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * AT_QT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int KP = D + 2, NW = 2 * win + 1;                // odd dword pitch
  bf16_t* St = reinterpret_cast<bf16_t*>(smem);          // [T][KP]
  float* Es = reinterpret_cast<float*>(St + (size_t)T * KP + (((size_t)T * KP) & 1));   // Ek [NW][D], Ev [NW][D]
  float* Qs = Es + 2 * NW * D;                           // [4 waves][D]
  float* Ps = Qs + 4 * D;                                // [AT_QT][T]
  const int len = lens[b];
  const size_t rbase = (size_t)gt_row_base(row0, b, Tp) + HALO;
  const int nv1 = gt_row_count(row0, b, Tp) - HALO - 1;          // ragged layout: last own row behind rbase (attn_mfma.hip)
  auto RW = [&](int t) { return rbase + (size_t)(t < nv1 ? t : nv1); };
  at_stage(St, k, ld, rbase, h, T, D, KP, tid, nv1);
  for (int i = tid; i < NW * D; i += 256) { Es[i] = Ek[i]; Es[NW * D + i] = Ev[i]; }
  __syncthreads();
  const float inv_sqrt = rsqrtf((float)D);
  float* qv = Qs + w * D;
  for (int ii = w; ii < AT_QT; ii += 4) {
    const int i = i0 + ii;
    if (i >= T) break;                                   // wave-uniform
    float* pv = Ps + (size_t)ii * T;
    for (int c = lane; c < D; c += 64) qv[c] = bf2f(q[RW(i) * ld + h * D + c]);
    __builtin_amdgcn_wave_barrier();
    float mx = -3.0e38f;
    for (int j = lane; j < T; j += 64) {
      float s = at_dot(qv, St + (size_t)j * KP, D);
      const int rel = j - i + win;
      if (rel >= 0 && rel <= 2 * win) { const float* e = Es + rel * D; float t = 0.f; for (int c = 0; c < D; ++c) t += qv[c] * e[c]; s += t; }
      s *= inv_sqrt;
      if (j >= len || i >= len) s = -1e4f;                // masked_fill(mask == 0, -1e4), attentions.py:260
      pv[j] = s; mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float den = 0.f;
    for (int j = lane; j < T; j += 64) { const float e = __expf(pv[j] - mx); pv[j] = e; den += e; }
    den = wave_sum(den);
    const float rden = 1.0f / den;
    float* prow = Pout + (((size_t)b * H + h) * T + i) * T;
    for (int j = lane; j < T; j += 64) {
      const float pj = pv[j] * rden;
      prow[j] = pj;
      float pd = pj;
      if (drop_thresh) pd = drop_keep(drop_seed, (uint32_t)((b * H + h) * T + i), j, drop_thresh) ? pj * drop_scale : 0.f;
      pv[j] = pd;
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  at_stage(St, v, ld, rbase, h, T, D, KP, tid, nv1);
  __syncthreads();
  for (int ii = w; ii < AT_QT; ii += 4) {
    const int i = i0 + ii;
    if (i >= T) break;
    const float* pv = Ps + (size_t)ii * T;
    for (int c = lane; c < D; c += 64) {
      float acc = 0.f;
      for (int j = 0; j < T; ++j) acc += pv[j] * bf2f(St[(size_t)j * KP + c]);
      for (int rel = 0; rel <= 2 * win; ++rel) { const int j = i + rel - win; if (j >= 0 && j < T) acc += pv[j] * Es[NW * D + rel * D + c]; }
      if (i <= nv1) out[(rbase + i) * ldo + h * D + c] = f2bf(acc);
    }
  }
}

// backward, pass 1 (per query row): dS row -> dSout [B,H,T,T] fp32 (scaled by 1/sqrt(D)), dQ, dEk, dEv.
__global__ __launch_bounds__(256) void gt_attn_bwd_q_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int ld,
    const float* __restrict__ Ek, const float* __restrict__ Ev, const int32_t* __restrict__ lens,
    const bf16_t* __restrict__ dout, int lddo, const float* __restrict__ P, float* __restrict__ dS,
    bf16_t* __restrict__ dq, int lddq, float* __restrict__ dEk, float* __restrict__ dEv,
    int T, int Tp, const int32_t* row0, int H, int D, int win, uint32_t drop_thresh, uint32_t drop_seed, float drop_scale, const uint32_t* __restrict__ seed_dev)
{
  if (seed_dev) drop_seed ^= *seed_dev;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * AT_QT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int KP = D + 2, NW = 2 * win + 1;
  bf16_t* St = reinterpret_cast<bf16_t*>(smem);
  float* Es = reinterpret_cast<float*>(St + (size_t)T * KP + (((size_t)T * KP) & 1));
  float* Acc = Es + 2 * NW * D;                          // block-local dEk | dEv accumulators [2][NW][D]
  float* Qs = Acc + 2 * NW * D;                          // [AT_QT][2*D]: q row and dO row
  float* Ps = Qs + AT_QT * 2 * D;                        // [AT_QT][T]
  const int len = lens[b];
  const size_t rbase = (size_t)gt_row_base(row0, b, Tp) + HALO;
  const int nv1 = gt_row_count(row0, b, Tp) - HALO - 1;          // ragged layout: last own row behind rbase (attn_mfma.hip)
  auto RW = [&](int t) { return rbase + (size_t)(t < nv1 ? t : nv1); };
  at_stage(St, v, ld, rbase, h, T, D, KP, tid, nv1);
  for (int i = tid; i < NW * D; i += 256) { Es[i] = Ek[i]; Es[NW * D + i] = Ev[i]; Acc[i] = 0.f; Acc[NW * D + i] = 0.f; }
  __syncthreads();
  const float inv_sqrt = rsqrtf((float)D);
  for (int ii = w; ii < AT_QT; ii += 4) {
    const int i = i0 + ii;
    if (i >= T) break;
    float* qv = Qs + ii * 2 * D; float* dov = qv + D;
    float* pv = Ps + (size_t)ii * T;
    for (int c = lane; c < D; c += 64) { qv[c] = bf2f(q[RW(i) * ld + h * D + c]); dov[c] = bf2f(dout[RW(i) * lddo + h * D + c]); }
    __builtin_amdgcn_wave_barrier();
    const float* prow = P + (((size_t)b * H + h) * T + i) * T;
    // dPd_j = dO.V_j + [band] dO.Ev[rel];  dP_j = dropout'(dPd_j);  Dsum = sum_j dP_j P_j
    float dsum = 0.f;
    for (int j = lane; j < T; j += 64) {
      float s = at_dot(dov, St + (size_t)j * KP, D);
      const int rel = j - i + win;
      if (rel >= 0 && rel <= 2 * win) { const float* e = Es + NW * D + rel * D; float t = 0.f; for (int c = 0; c < D; ++c) t += dov[c] * e[c]; s += t; }
      if (drop_thresh) s = drop_keep(drop_seed, (uint32_t)((b * H + h) * T + i), j, drop_thresh) ? s * drop_scale : 0.f;
      pv[j] = s;
      dsum += s * prow[j];
    }
    dsum = wave_sum(dsum);
    float* dsrow = dS + (((size_t)b * H + h) * T + i) * T;
    for (int j = lane; j < T; j += 64) {
      float ds = prow[j] * (pv[j] - dsum) * inv_sqrt;
      if (j >= len || i >= len) ds = 0.f;                 // masked_fill blocks the gradient
      pv[j] = ds; dsrow[j] = ds;
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  at_stage(St, k, ld, rbase, h, T, D, KP, tid, nv1);
  __syncthreads();
  for (int ii = w; ii < AT_QT; ii += 4) {
    const int i = i0 + ii;
    if (i >= T) break;
    const float* qv = Qs + ii * 2 * D; const float* dov = qv + D;
    const float* pv = Ps + (size_t)ii * T;
    const float* prow = P + (((size_t)b * H + h) * T + i) * T;
    for (int c = lane; c < D; c += 64) {
      float acc = 0.f;
      for (int j = 0; j < T; ++j) acc += pv[j] * bf2f(St[(size_t)j * KP + c]);
      for (int rel = 0; rel <= 2 * win; ++rel) {
        const int j = i + rel - win;
        if (j >= 0 && j < T) {
          acc += pv[j] * Es[rel * D + c];
          if (pv[j] != 0.f) atomicAdd(Acc + rel * D + c, pv[j] * qv[c]);
          float pd = prow[j];                              // dEv[rel] += dropout(P)[i,j] * dO_i
          if (drop_thresh) pd = drop_keep(drop_seed, (uint32_t)((b * H + h) * T + i), j, drop_thresh) ? pd * drop_scale : 0.f;
          if (pd != 0.f && i < len) atomicAdd(Acc + NW * D + rel * D + c, pd * dov[c]);
        }
      }
      if (i <= nv1) dq[(rbase + i) * lddq + h * D + c] = f2bf(acc);
    }
  }
  __syncthreads();
  for (int i = tid; i < NW * D; i += 256) {
    if (Acc[i] != 0.f) atomicAdd(dEk + i, Acc[i]);
    if (Acc[NW * D + i] != 0.f) atomicAdd(dEv + i, Acc[NW * D + i]);
  }
}

// backward, pass 2 (per key row j): dK_j = sum_i dS[i,j] q_i ;  dV_j = sum_i pd[i,j] dO_i.
__global__ __launch_bounds__(256) void gt_attn_bwd_kv_kernel(
    const bf16_t* __restrict__ q, int ld, const bf16_t* __restrict__ dout, int lddo, const float* __restrict__ P,
    const float* __restrict__ dS, const int32_t* __restrict__ lens, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int lddk,
    int T, int Tp, const int32_t* row0, int H, int D, uint32_t drop_thresh, uint32_t drop_seed, float drop_scale, const uint32_t* __restrict__ seed_dev)
{
  if (seed_dev) drop_seed ^= *seed_dev;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.z, h = blockIdx.y, j0 = blockIdx.x * AT_QT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int KP = D + 2;
  bf16_t* St = reinterpret_cast<bf16_t*>(smem);          // [T][KP] q rows, then dO rows
  float* Cs = reinterpret_cast<float*>(St + (size_t)T * KP + (((size_t)T * KP) & 1));   // [AT_QT][2*T] dS column | pd column
  const size_t rbase = (size_t)gt_row_base(row0, b, Tp) + HALO;
  const int nv1 = gt_row_count(row0, b, Tp) - HALO - 1;          // ragged layout: last own row behind rbase (attn_mfma.hip)
  auto RW = [&](int t) { return rbase + (size_t)(t < nv1 ? t : nv1); };
  const int len = lens[b];
  at_stage(St, q, ld, rbase, h, T, D, KP, tid, nv1);
  for (int jj = w; jj < AT_QT; jj += 4) {
    const int j = j0 + jj;
    if (j >= T) break;
    float* dsc = Cs + (size_t)jj * 2 * T; float* pdc = dsc + T;
    for (int i = lane; i < T; i += 64) {
      const size_t o = (((size_t)b * H + h) * T + i) * T + j;
      dsc[i] = dS[o];
      float pd = (i < len) ? P[o] : 0.f;                  // query rows >= len carry no upstream gradient
      if (drop_thresh) pd = drop_keep(drop_seed, (uint32_t)((b * H + h) * T + i), j, drop_thresh) ? pd * drop_scale : 0.f;
      pdc[i] = pd;
    }
  }
  __syncthreads();
  for (int jj = w; jj < AT_QT; jj += 4) {
    const int j = j0 + jj;
    if (j >= T) break;
    const float* dsc = Cs + (size_t)jj * 2 * T;
    for (int c = lane; c < D; c += 64) {
      float ak = 0.f;
      for (int i = 0; i < T; ++i) ak += dsc[i] * bf2f(St[(size_t)i * KP + c]);
      if (j <= nv1) dk[(rbase + j) * lddk + h * D + c] = f2bf(ak);
    }
  }
  __syncthreads();
  at_stage(St, dout, lddo, rbase, h, T, D, KP, tid, nv1);
  __syncthreads();
  for (int jj = w; jj < AT_QT; jj += 4) {
    const int j = j0 + jj;
    if (j >= T) break;
    const float* pdc = Cs + (size_t)jj * 2 * T + T;
    for (int c = lane; c < D; c += 64) {
      float av = 0.f;
      for (int i = 0; i < T; ++i) av += pdc[i] * bf2f(St[(size_t)i * KP + c]);
      if (j <= nv1) dv[(rbase + j) * lddk + h * D + c] = f2bf(av);
    }
  }
}

// ------------------------------------------------------------------ embedding
// rows[b*Tp+HALO+t, :] = emb[ids[b,t], :] * scale * (t < len[b]);  fp32 + bf16 copies; halos zero.
__global__ __launch_bounds__(256) void gt_embedding_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ emb,
                                                               const int32_t* __restrict__ lens, float* __restrict__ out_f32,
                                                               bf16_t* __restrict__ out_bf16, int B, int T, int Tp, int C, float scale,
                                                               const int32_t* __restrict__ row0, int R, int ld)
{
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= R) return;
  const int b = gt_row_batch(row0, B, m, Tp), t = m - gt_row_base(row0, b, Tp) - HALO;
  const bool valid = t >= 0 && t < T && t < lens[b];
  const int64_t id = valid ? ids[(size_t)b * T + t] : 0;
  for (int c = lane; c < C; c += 64) {
    const float v = valid ? emb[(size_t)id * C + c] * scale : 0.f;
    if (out_f32) out_f32[(size_t)m * ld + c] = v;
    if (out_bf16) out_bf16[(size_t)m * ld + c] = f2bf(v);
  }
}
__global__ __launch_bounds__(256) void gt_embedding_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dx,
                                                               const int32_t* __restrict__ lens, float* __restrict__ demb,
                                                               int B, int T, int Tp, int C, float scale,
                                                               const int32_t* __restrict__ row0, int R, int ld)
{
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= R) return;
  const int b = gt_row_batch(row0, B, m, Tp), t = m - gt_row_base(row0, b, Tp) - HALO;
  if (t < 0 || t >= T || t >= lens[b]) return;
  const int64_t id = ids[(size_t)b * T + t];
  for (int c = lane; c < C; c += 64) atomicAdd(demb + (size_t)id * C + c, dx[(size_t)m * ld + c] * scale);
}

// ------------------------------------------------------------------ per-utterance vector added to every valid row
// out[m,:] = (x[m,:] + cond[batch(m),:]) * rowmask[m]: the speaker vector of attentions.py:66-67 (Encoder.cond_g) and
// models.py:587-589 (DurationPredictor.cond), which the reference broadcasts over time.  Source fp32 (x) or bf16 (xb);
// fp32 and/or bf16 output; halo / padded rows stay zero.
__global__ __launch_bounds__(256) void gt_rows_add_cond_kernel(const float* __restrict__ x, int ldx, const bf16_t* __restrict__ xb, int ldxb,
                                                               const float* __restrict__ cond, const float* __restrict__ rowmask,
                                                               float* __restrict__ out, int ldo, bf16_t* __restrict__ outb, int ldob,
                                                               int B, int R, int C, int Tp, const int32_t* __restrict__ row0)
{
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= R) return;
  const bool valid = rowmask[m] != 0.f;
  const int b = valid ? gt_row_batch(row0, B, m, Tp) : 0;
  for (int c = lane; c < C; c += 64) {
    float v = 0.f;
    if (valid) v = (x ? x[(size_t)m * ldx + c] : bf2f(xb[(size_t)m * ldxb + c])) + cond[(size_t)b * C + c];
    if (out) out[(size_t)m * ldo + c] = v;
    if (outb) outb[(size_t)m * ldob + c] = f2bf(v);
  }
}

// out[b, c] (+)= sum over the rows m of utterance b of y[m, c] * rowmask[m]: the gradient of a per-utterance vector
// (gt_rows_add_cond; the cond input of the WN gate, modules.py:148-156).  One workgroup per (utterance, 64 channels):
// waves stride the rows, lanes own channels, one LDS fold — no atomics.
template <bool F32>
__global__ __launch_bounds__(1024) void gt_rows_utt_sum_kernel(const void* __restrict__ y, int ldy, const float* __restrict__ rowmask,
                                                               float* __restrict__ out, int ldo, int accumulate,
                                                               int B, int C, int Tp, const int32_t* __restrict__ row0)
{
  // latency-bound (a few hundred rows of 128 B per workgroup): 16 waves, 4 independent row loads in flight per lane
  __shared__ float part[16][64];
  const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const int base = gt_row_base(row0, b, Tp), cnt = gt_row_count(row0, b, Tp);
  auto ld1 = [&](int t) -> float {
    if (t >= cnt) return 0.f;
    const size_t m = (size_t)(base + t);
    const float k = rowmask ? rowmask[m] : 1.f;             // both loads independent; masked rows may hold anything
    const float val = F32 ? static_cast<const float*>(y)[m * ldy + c] : bf2f(static_cast<const bf16_t*>(y)[m * ldy + c]);
    return k != 0.f ? k * val : 0.f;
  };
  float acc = 0.f;
  if (c < C)
    for (int t = w; t < cnt; t += 64) {
      const float a0 = ld1(t), a1 = ld1(t + 16), a2 = ld1(t + 32), a3 = ld1(t + 48);
      acc += (a0 + a1) + (a2 + a3);
    }
  part[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += part[i][threadIdx.x];
    float* o = out + (size_t)b * ldo + c;
    *o = accumulate ? *o + s : s;
  }
}

// ragged rows layout: row -> (utterance, frame, valid) tables from the row offsets, one launch (was a dozen host-side ops)
__global__ __launch_bounds__(256) void gt_rows_ctx_fill_kernel(const int32_t* __restrict__ row0, const int32_t* __restrict__ lens,
                                                               int64_t* __restrict__ rowbatch, int32_t* __restrict__ rowframe,
                                                               float* __restrict__ rowmask, int32_t* __restrict__ rowutt, int B, int R)
{
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= R) return;
  const int b = gt_row_batch(row0, B, m, 0);
  const int t = m - row0[b] - HALO;
  rowbatch[m] = b;
  if (rowutt) rowutt[m] = b;
  rowframe[m] = t;
  rowmask[m] = (t >= 0 && t < lens[b]) ? 1.f : 0.f;
}

// One launch for everything a replayed step needs from its batch: the inputs copied into the graph's padded static buffers (rows of
// src_words 4-byte words into rows of dst_words >= src_words, the rest zeroed) and the row tables of up to three ragged contexts rebuilt
// from the row offsets / lengths the host left in pinned memory (read once per workgroup, over PCIe, into LDS).  Was ten launches of
// 4-6 us each between two graph replays: serial time of every step.
__global__ __launch_bounds__(256) void gt_step_inputs_kernel(gt_step_inputs_args a)
{
  __shared__ int32_t geo[2 * GT_STEP_MAX_B + 1];
  const int blk = blockIdx.x;
  for (int j = 0; j < a.n_copy; ++j) {
    const gt_step_copy c = a.copy[j];
    const int nb = (int)(((size_t)c.rows * c.dst_words + 1023) / 1024);
    if (blk >= c.blk0 && blk < c.blk0 + nb) {
      const uint32_t* __restrict__ src = static_cast<const uint32_t*>(c.src);
      uint32_t* __restrict__ dst = static_cast<uint32_t*>(c.dst);
      const size_t total = (size_t)c.rows * c.dst_words;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t w = (size_t)(blk - c.blk0) * 1024 + i * 256 + threadIdx.x;
        if (w < total) {
          const int row = (int)(w / c.dst_words), col = (int)(w - (size_t)row * c.dst_words);
          dst[w] = col < c.src_words ? src[(size_t)row * c.src_words + col] : 0u;
        }
      }
      return;
    }
  }
  for (int j = 0; j < a.n_ctx; ++j) {
    const gt_step_ctx c = a.ctx[j];
    const int nb = (c.R + 255) / 256;
    if (blk >= c.blk0 && blk < c.blk0 + nb) {
      for (int i = threadIdx.x; i < 2 * c.B + 1; i += 256) {
        const int32_t v = c.geo_src[i];
        geo[i] = v;
        if (blk == c.blk0) c.geo_dst[i] = v;            // the device copy later kernels read (row0[B+1] | lengths[B])
      }
      __syncthreads();
      const int m = (blk - c.blk0) * 256 + threadIdx.x;
      if (m >= c.R) return;
      const int b = gt_row_batch(geo, c.B, m, 0);
      const int t = m - geo[b] - HALO;
      c.rowbatch[m] = b;
      if (c.rowutt) c.rowutt[m] = b;
      c.rowframe[m] = t;
      c.rowmask[m] = (t >= 0 && t < geo[c.B + 1 + b]) ? 1.f : 0.f;
      return;
    }
  }
}

// The fills at the head of a training step in one launch: up to GT_ZERO_MAX regions cleared (16-byte aligned, sizes multiples of 16) and
// the dropout seed word advanced (the step's accumulator arena, its pre-zeroed buffer region, the accumulated-gradient slice of the flat
// buffer: three fills and an add before — serial launches in front of both branches of the step).
__global__ __launch_bounds__(256) void gt_step_zero_kernel(gt_step_zero_args a)
{
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.seed_word) *a.seed_word += a.seed_inc;
  size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;               // 16-byte word of the concatenated regions
  for (int j = 0; j < a.n; ++j) {
    const size_t nw = a.bytes[j] >> 4;
    if (w < nw) { reinterpret_cast<uint4*>(a.ptr[j])[w] = make_uint4(0, 0, 0, 0); return; }
    w -= nw;
  }
}

// ------------------------------------------------------------------ logp lattice (models.py:1076-1082)
// logp[b,i,j] = sum_d(-0.5 log 2pi - s_id) + sum_d e^{-2 s_id} (-0.5 z_jd^2) + sum_d m_id e^{-2 s_id} z_jd
//               + sum_d -0.5 m_id^2 e^{-2 s_id}
// x_m, x_logs: [B, C, T_x] fp32; z: [B, C, T_y] fp32.  exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): one wave
// per 32x32 lattice tile, K = 2C interleaved as (e^{-2s}, -0.5 z^2) and (m e^{-2s}, z) pairs.
__global__ __launch_bounds__(256) void gt_logp_kernel(const float* __restrict__ xm, const float* __restrict__ xlogs,
                                                      const float* __restrict__ z, float* __restrict__ logp,
                                                      int B, int C, int Tx, int Ty)
{
  const int b = blockIdx.z;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32, j0 = (blockIdx.x * 4 + w) * 32;
  if (j0 >= Ty) return;
  const int r = lane & 31, kh = lane >> 5;
  const int i = i0 + r, j = j0 + r;
  const float* xmb = xm + (size_t)b * C * Tx;
  const float* xsb = xlogs ? xlogs + (size_t)b * C * Tx : nullptr;
  const float* zb = z + (size_t)b * C * Ty;
  f32x16_t acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float rowc = 0.f;                                       // logp1 + logp4 of row i (lanes kh==0 and kh==1 split d)
  // 8 k-pairs per trip with their loads issued together: one trip = one memory round trip for 16 MFMAs (a load / MFMA pair per
  // trip left the wave waiting on 80 dependent round trips: 33 us for a lattice whose MFMAs take 4)
  constexpr int U = 8;
  for (int d0 = 0; d0 < C; d0 += 2 * U) {
    float m[U], sv[U], zz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int dd = d0 + 2 * u + kh;                     // this lane's k index
      m[u] = 0.f; sv[u] = 0.f; zz[u] = 0.f;
      if (dd < C) {
        if (i < Tx) { m[u] = xmb[(size_t)dd * Tx + i]; if (xsb) sv[u] = xsb[(size_t)dd * Tx + i]; }
        if (j < Ty) zz[u] = zb[(size_t)dd * Ty + j];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = d0 + 2 * u + kh < C;
      const float e2 = xsb ? __expf(-2.0f * sv[u]) : 1.0f;
      if (in) rowc += -0.9189385332046727f - sv[u] - 0.5f * m[u] * m[u] * e2;
      if (d0 + 2 * u < C) {                                 // wave-uniform (C is even): whole k pairs only
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(in ? e2 : 0.f, -0.5f * zz[u] * zz[u], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(in ? m[u] * e2 : 0.f, zz[u], acc, 0, 0, 0);
      }
    }
  }
  rowc += __shfl_xor(rowc, 32);                           // both k halves -> full sum over d, indexed by r = row i0+r
  // C/D layout: col = lane&31 (= j), row = (e&3) + 8*(e>>2) + 4*kh (= i offset)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int ri = (e & 3) + 8 * (e >> 2) + 4 * kh;
    const float rc = __shfl(rowc, ri);                    // lane ri holds row i0+ri's constant
    if (i0 + ri < Tx && j < Ty) logp[((size_t)b * Tx + i0 + ri) * Ty + j] = acc[e] + rc;
  }
}

// ------------------------------------------------------------------ prior expansion + mle loss
// z_m[b,c,j] = x_m[b,c,tok[b,j]] (0 where tok < 0)  — models.py:1118 as a gather.
__global__ __launch_bounds__(256) void gt_prior_expand_kernel(const float* __restrict__ xm, const int32_t* __restrict__ tok,
                                                              float* __restrict__ zm, int B, int C, int Tx, int Ty)
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * C * Ty) return;
  const int j = idx % Ty, bc = idx / Ty, b = bc / C;
  const int t = tok[(size_t)b * Ty + j];
  zm[idx] = t >= 0 ? xm[(size_t)bc * Tx + t] : 0.f;
}
// backward of the gather: dxm[b,c,i] = sum of dzm[b,c,j] over the frames j aligned to token i.  One wave per (b, c) row walks
// the frames 64 at a time (coalesced), sums each run of equal tokens with a segmented wave scan (the MAS path is monotone: runs
// are contiguous) and the run's last lane adds it to the row's token accumulator in LDS — fixed order, no atomics.  (The first
// form — a thread per (b, c, token) looping over its frames — took 74 us on the step's critical path: autograd runs it before the
// decoder's backward.)
__global__ __launch_bounds__(256) void gt_prior_expand_bwd_kernel(const float* __restrict__ dzm, const int32_t* __restrict__ f2t,
                                                                  float* __restrict__ dxm, int B, int C, int Tx, int Ty)
{
  __shared__ float acc[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bc = blockIdx.x * 4 + wave;
  if (bc >= B * C) return;                                  // waves are independent: no workgroup barrier below
  const int b = bc / C;
  float* a = acc[wave];
  for (int i = lane; i < Tx; i += 64) a[i] = 0.f;
  for (int j0 = 0; j0 < Ty; j0 += 64) {
    const int j = j0 + lane;
    int t = -1; float v = 0.f;
    if (j < Ty) { t = f2t[(size_t)b * Ty + j]; v = dzm[(size_t)bc * Ty + j]; }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float vv = __shfl_up(v, off);
      const int tt = __shfl_up(t, off);
      if (lane >= off && tt == t) v += vv;
    }
    const int tn = __shfl_down(t, 1);
    if (t >= 0 && t < Tx && (lane == 63 || tn != t)) a[t] += v;   // a run that crosses the chunk boundary continues in the next chunk
  }
  for (int i = lane; i < Tx; i += 64) dxm[(size_t)bc * Tx + i] = a[i];
}
// mle loss partial sums (commons.py:28-33): acc[0] += sum(logs*?) ... computed over [B,C,T] tensors:
//   acc[0] += sum logs,  acc[1] += sum exp(-2 logs) (z-m)^2   (logs may be NULL = 0)
__global__ __launch_bounds__(256) void gt_mle_sums_kernel(const float* __restrict__ z, const float* __restrict__ m,
                                                          const float* __restrict__ logs, float* __restrict__ acc, size_t n)
{
  __shared__ float red[2][4];
  float a0 = 0.f, a1 = 0.f;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 zz = reinterpret_cast<const float4*>(z)[i], mm = reinterpret_cast<const float4*>(m)[i];
    float4 ll = make_float4(0.f, 0.f, 0.f, 0.f);
    if (logs) ll = reinterpret_cast<const float4*>(logs)[i];
    const float d0 = zz.x - mm.x, d1 = zz.y - mm.y, d2 = zz.z - mm.z, d3 = zz.w - mm.w;
    a0 += ll.x + ll.y + ll.z + ll.w;
    a1 += __expf(-2.0f * ll.x) * d0 * d0 + __expf(-2.0f * ll.y) * d1 * d1 + __expf(-2.0f * ll.z) * d2 * d2 + __expf(-2.0f * ll.w) * d3 * d3;
  }
  if (blockIdx.x == 0) {
    for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      const float l = logs ? logs[i] : 0.f, d = z[i] - m[i];
      a0 += l; a1 += __expf(-2.0f * l) * d * d;
    }
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a0; red[1][threadIdx.x >> 6] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {                                   // one partial pair per workgroup: gt_mle_finish adds the GT_MLE_PARTS of them
    acc[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];          // (512 workgroups x 2 same-address atomics and four
    acc[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];      //  dependent load rounds per thread before: 16.7 us)
  }
}
// dz = g * exp(-2 logs) (z-m);  dm = -dz;  dlogs = g * (1 - exp(-2 logs)(z-m)^2)   with g = *gscale
__global__ __launch_bounds__(256) void gt_mle_bwd_kernel(const float* __restrict__ z, const float* __restrict__ m,
                                                         const float* __restrict__ logs, const float* __restrict__ gscale,
                                                         float* __restrict__ dz, float* __restrict__ dm, float* __restrict__ dlogs, size_t n,
                                                         const float* __restrict__ gdenom, float* __restrict__ dlogdet, int B)
{
  const float g = gdenom ? *gscale / *gdenom : *gscale;
  if (dlogdet && blockIdx.x == 0)
    for (int b = threadIdx.x; b < B; b += 256) dlogdet[b] = -g;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float l = logs ? logs[i] : 0.f, e = __expf(-2.0f * l), d = z[i] - m[i];
    const float v = g * e * d;
    if (dz) dz[i] = v;
    if (dm) dm[i] = -v;
    if (dlogs) dlogs[i] = g * (1.0f - e * d * d);
  }
}

// commons.sequence_mask as floats: mask[b, t] = t < len[b]  (lengths int32 or int64)
__global__ __launch_bounds__(256) void gt_length_mask_kernel(const void* __restrict__ len, int is64, float* __restrict__ mask, int B, int T)
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * T) return;
  const int b = idx / T, t = idx - b * T;
  const long long n = is64 ? static_cast<const long long*>(len)[b] : (long long)static_cast<const int32_t*>(len)[b];
  mask[idx] = t < n ? 1.f : 0.f;
}

// mle_loss's scalar tail (commons.py:31-33) in one launch: out[0] = loss = (acc[0] + 0.5 acc[1] - sum logdet) / denom + 0.5 log 2pi,
// out[1] = denom = C * sum(mask)  (= sum(ones_like(z) * mask))
__global__ __launch_bounds__(1024) void gt_mle_finish_kernel(const float* __restrict__ acc, const float* __restrict__ logdet,
                                                             const float* __restrict__ mask, int n_mask, int B, int C, float* __restrict__ out)
{
  // one workgroup (the result is two scalars), 1024 threads and 16-byte loads: the mask is B x T floats (25 k at cfg 2) and a
  // 256-thread scalar loop over it sat 40 us on the step's critical path
  __shared__ float red[2][16];
  __shared__ float red2[2];
  {                                                          // the GT_MLE_PARTS partial pairs of gt_mle_sums
    float p0 = 0.f, p1 = 0.f;
    for (int i = threadIdx.x; i < GT_MLE_PARTS; i += 1024) { const float2 v = reinterpret_cast<const float2*>(acc)[i]; p0 += v.x; p1 += v.y; }
    p0 = wave_sum(p0); p1 = wave_sum(p1);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = p0; red[1][threadIdx.x >> 6] = p1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      p0 = 0.f; p1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { p0 += red[0][i]; p1 += red[1][i]; }
      red2[0] = p0; red2[1] = p1;
    }
    __syncthreads();
  }
  float sl = 0.f, sn = 0.f;
  for (int b = threadIdx.x; b < B; b += 1024) sl += logdet[b];
  const int n4 = ((reinterpret_cast<uintptr_t>(mask) & 15) == 0) ? n_mask >> 2 : 0;
  for (int i = threadIdx.x; i < n4; i += 1024) { const float4 v = reinterpret_cast<const float4*>(mask)[i]; sn += (v.x + v.y) + (v.z + v.w); }
  for (int i = (n4 << 2) + threadIdx.x; i < n_mask; i += 1024) sn += mask[i];
  sl = wave_sum(sl); sn = wave_sum(sn);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sl; red[1][threadIdx.x >> 6] = sn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sl = 0.f; sn = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { sl += red[0][i]; sn += red[1][i]; }
    const float denom = sn * (float)C;
    out[0] = (red2[0] + 0.5f * red2[1] - sl) / denom + 0.91893853320467274f;
    out[1] = denom;
  }
}

// duration loss of the deterministic predictor (models.py:1089-1092): l[b] = sum_t (logw[b,t] - log(w[b,t] + 1e-8) mask)^2 / sum(mask)
// One wave per utterance; backward: d logw[b,t] = g[b] * 2 (logw - logw_) / sum(mask) on valid tokens (w comes from MAS: no gradient).
__global__ void gt_duration_loss_fwd_kernel(const float* __restrict__ logw, const float* __restrict__ w, const int32_t* __restrict__ len,
                                            int B, int Tx, float* __restrict__ out)
{
  const int b = blockIdx.x, lane = threadIdx.x;
  float tot = 0.f;
  for (int i = lane; i < B; i += 64) tot += (float)len[i];
  tot = wave_sum(tot);
  float a = 0.f;
  const int n = len[b];
  for (int t = lane; t < Tx; t += 64) {
    const float lw = logw[(size_t)b * Tx + t];
    const float ref = t < n ? __logf(w[(size_t)b * Tx + t] + 1e-8f) : 0.f;
    const float d = lw - ref;
    a += d * d;
  }
  a = wave_sum(a);
  if (lane == 0) out[b] = a / tot;
}
__global__ void gt_duration_loss_bwd_kernel(const float* __restrict__ logw, const float* __restrict__ w, const int32_t* __restrict__ len,
                                            const float* __restrict__ g, int B, int Tx, float* __restrict__ dlogw)
{
  const int b = blockIdx.x, lane = threadIdx.x;
  float tot = 0.f;
  for (int i = lane; i < B; i += 64) tot += (float)len[i];
  tot = wave_sum(tot);
  const int n = len[b];
  const float gb = 2.0f * g[b] / tot;
  for (int t = lane; t < Tx; t += 64) {
    const float lw = logw[(size_t)b * Tx + t];
    const float ref = t < n ? __logf(w[(size_t)b * Tx + t] + 1e-8f) : 0.f;
    dlogw[(size_t)b * Tx + t] = gb * (lw - ref);
  }
}

}  // namespace

#define GT_ST(s) static_cast<hipStream_t>(s)
#define GT_RET() return gt_launch_status(__func__)

static void fill_drop(float p, uint32_t seed, uint32_t& th, uint32_t& sd, float& sc)
{
  th = 0; sd = seed; sc = 1.0f;
  if (p > 0.f) { th = (uint32_t)((double)p * 4294967296.0); sc = 1.0f / (1.0f - p); }
}

static int fill_ln(LnArgs& p, const float* a, const void* y, int ldy, const float* gamma, const float* beta, const float* rowmask,
                   float* out_f32, void* out_bf16, int ldo, float* mean, float* rstd, int R, int C, float eps,
                   float p_in, uint32_t seed_in, float p_out, uint32_t seed_out, int relu, const uint32_t* seed_dev)
{
  if ((!a && !y) || !gamma || !beta || !mean || !rstd || R <= 0 || C <= 0 || C > 256) return GT_E_INVAL;
  if (p_in >= 1.f || p_out >= 1.f) return GT_E_INVAL;
  p.a = a; p.y = static_cast<const bf16_t*>(y); p.ldy = ldy; p.gamma = gamma; p.beta = beta; p.rowmask = rowmask;
  p.out_f32 = out_f32; p.out_bf16 = static_cast<bf16_t*>(out_bf16); p.ldo = ldo; p.mean = mean; p.rstd = rstd;
  p.R = R; p.C = C; p.eps = eps; p.relu = relu; p.seed_dev = seed_dev;
  fill_drop(p_in, seed_in, p.din_thresh, p.din_seed, p.din_scale);
  fill_drop(p_out, seed_out, p.dout_thresh, p.dout_seed, p.dout_scale);
  return GT_OK;
}

extern "C" int gt_layernorm_fwd(const float* a, const void* y, int ldy, const float* gamma, const float* beta, const float* rowmask,
                                float* out_f32, void* out_bf16, int ldo, float* mean, float* rstd, int R, int C, float eps,
                                float p_in, uint32_t seed_in, float p_out, uint32_t seed_out, int relu, const uint32_t* seed_dev, void* stream)
{
  LnArgs p;
  const int rc = fill_ln(p, a, y, ldy, gamma, beta, rowmask, out_f32, out_bf16, ldo, mean, rstd, R, C, eps, p_in, seed_in, p_out, seed_out, relu, seed_dev);
  if (rc) return rc;
  hipLaunchKernelGGL(gt_layernorm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, GT_ST(stream), p);
  GT_RET();
}

extern "C" int gt_layernorm_bwd(const float* a, const void* y, int ldy, const float* gamma, const float* beta, const float* rowmask,
                                const float* mean, const float* rstd, int R, int C, float eps,
                                float p_in, uint32_t seed_in, float p_out, uint32_t seed_out, int relu, const uint32_t* seed_dev,
                                const float* dout_f32, const void* dout_bf16, int lddo,
                                float* da, void* dy, int lddy, float* dgamma, float* dbeta, void* stream)
{
  LnBwdArgs q;
  const int rc = fill_ln(q.f, a, y, ldy, gamma, beta, rowmask, nullptr, nullptr, 0, const_cast<float*>(mean), const_cast<float*>(rstd),
                         R, C, eps, p_in, seed_in, p_out, seed_out, relu, seed_dev);
  if (rc) return rc;
  if ((!dout_f32 && !dout_bf16) || !dgamma || !dbeta) return GT_E_INVAL;
  q.dout_f32 = dout_f32; q.dout_bf16 = static_cast<const bf16_t*>(dout_bf16); q.lddo = lddo;
  q.da = da; q.dy = static_cast<bf16_t*>(dy); q.lddy = lddy; q.dgamma = dgamma; q.dbeta = dbeta; q.partials = nullptr;
  // geometry: 16 waves x 32 rows per workgroup (the gamma / beta partials are folded in LDS before the atomics) while
  // that still gives >= 64 workgroups; short inputs fall back to 4 waves x 16 rows so the chip is not left idle
#ifdef LNB_EXP
  if (LNB_EXP & 2) { q.rows_per_block = 64; hipLaunchKernelGGL(gt_layernorm_bwd_kernel<16>, dim3((R + 63) / 64), dim3(1024), 0, GT_ST(stream), q); GT_RET(); }
  if (LNB_EXP & 4) { q.rows_per_block = 32; hipLaunchKernelGGL(gt_layernorm_bwd_kernel<8>, dim3((R + 31) / 32), dim3(512), 0, GT_ST(stream), q); GT_RET(); }
  if (LNB_EXP & 8) { q.rows_per_block = 16; hipLaunchKernelGGL(gt_layernorm_bwd_kernel<4>, dim3((R + 15) / 16), dim3(256), 0, GT_ST(stream), q); GT_RET(); }
  if (LNB_EXP & 16) { q.rows_per_block = 16; hipLaunchKernelGGL(gt_layernorm_bwd_kernel<16>, dim3((R + 15) / 16), dim3(1024), 0, GT_ST(stream), q); GT_RET(); }
#endif
  if (R >= 64 * 32) {
    q.rows_per_block = 32;
    hipLaunchKernelGGL(gt_layernorm_bwd_kernel<16>, dim3((R + 31) / 32), dim3(1024), 0, GT_ST(stream), q);
  } else {
    q.rows_per_block = 16;
    hipLaunchKernelGGL(gt_layernorm_bwd_kernel<4>, dim3((R + 15) / 16), dim3(256), 0, GT_ST(stream), q);
  }
  GT_RET();
}

// The partials form: ONE row per wave (every load of the launch in flight at once: the saved rows are cold by the time the backward
// reads them), 16 rows per workgroup, the workgroup's dgamma | dbeta sums to row blockIdx.x of `partials` — no atomics, whose
// same-address chains grow with the number of workgroups.
extern "C" int gt_layernorm_bwd_partial_rows(int R) { return R > 0 ? (R + 15) / 16 : 0; }

extern "C" int gt_layernorm_bwd_partials(const float* a, const void* y, int ldy, const float* gamma, const float* beta, const float* rowmask,
                                         const float* mean, const float* rstd, int R, int C, float eps,
                                         float p_in, uint32_t seed_in, float p_out, uint32_t seed_out, int relu, const uint32_t* seed_dev,
                                         const float* dout_f32, const void* dout_bf16, int lddo,
                                         float* da, void* dy, int lddy, float* partials, void* stream)
{
  LnBwdArgs q;
  const int rc = fill_ln(q.f, a, y, ldy, gamma, beta, rowmask, nullptr, nullptr, 0, const_cast<float*>(mean), const_cast<float*>(rstd),
                         R, C, eps, p_in, seed_in, p_out, seed_out, relu, seed_dev);
  if (rc) return rc;
  if ((!dout_f32 && !dout_bf16) || !partials) return GT_E_INVAL;
  q.dout_f32 = dout_f32; q.dout_bf16 = static_cast<const bf16_t*>(dout_bf16); q.lddo = lddo;
  q.da = da; q.dy = static_cast<bf16_t*>(dy); q.lddy = lddy; q.dgamma = nullptr; q.dbeta = nullptr; q.partials = partials;
  q.rows_per_block = 16;
  hipLaunchKernelGGL(gt_layernorm_bwd_kernel<16>, dim3((R + 15) / 16), dim3(1024), 0, GT_ST(stream), q);
  GT_RET();
}

extern "C" int gt_param_partials_reduce(const gt_partials_args* args, void* stream)
{
  if (!args || args->n_jobs <= 0 || args->n_jobs > GT_PARTIALS_MAX) return GT_E_INVAL;
  int maxw = 0;
  for (int j = 0; j < args->n_jobs; ++j) {
    const gt_partials_job& b = args->job[j];
    if (!b.partials || !b.dst_a || b.n_rows <= 0 || b.Ca <= 0 || b.Cb < 0 || (b.Cb > 0 && !b.dst_b)) return GT_E_INVAL;
    maxw = b.Ca + b.Cb > maxw ? b.Ca + b.Cb : maxw;
  }
  hipLaunchKernelGGL(gt_param_partials_reduce_kernel, dim3((maxw + 63) / 64, args->n_jobs, PARTIALS_ZS), dim3(256), 0, GT_ST(stream), *args);
  GT_RET();
}

static size_t attn_lds(int T, int D, int win, size_t extra_floats)
{
  size_t halfs = (size_t)T * (D + 2); halfs += halfs & 1;     // keep the float area 4-byte aligned
  return halfs * 2 + (size_t)2 * (2 * win + 1) * D * 4 + extra_floats * 4;
}

extern "C" int gt_attn_fwd(const void* q, const void* k, const void* v, int ld, const float* Ek, const float* Ev,
                           const int32_t* lens, void* out, int ldo, float* P, int B, int T, int Tp, const int32_t* row0, int H, int D, int win,
                           float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream)
{
  if (!q || !k || !v || !Ek || !Ev || !lens || !out || !P || B <= 0 || T <= 0 || H <= 0) return GT_E_INVAL;
  if (D > AT_MAXD || (D & 1) || win < 0 || drop_p >= 1.f) return GT_E_UNSUPPORTED;
  {
    uint32_t th, sd; float sc; fill_drop(drop_p, drop_seed, th, sd, sc);
    {
      const int rc = gt_attn_fwd_mfma_impl(q, k, v, ld, Ek, Ev, lens, out, ldo, P, B, T, Tp, row0, H, D, win, th, sd, sc, seed_dev, stream);
      if (rc != 1) return rc;                      // handled (or failed loudly) on the MFMA path
    }
  }
  const size_t lds = attn_lds(T, D, win, (size_t)4 * D + (size_t)AT_QT * T);
  if (lds > 160 * 1024) return GT_E_UNSUPPORTED;
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GT_E_LAUNCH; attr = true; }
  uint32_t th, sd; float sc; fill_drop(drop_p, drop_seed, th, sd, sc);
  hipLaunchKernelGGL(gt_attn_fwd_kernel, dim3((T + AT_QT - 1) / AT_QT, H, B), dim3(256), lds, GT_ST(stream),
                     static_cast<const bf16_t*>(q), static_cast<const bf16_t*>(k), static_cast<const bf16_t*>(v), ld, Ek, Ev, lens,
                     static_cast<bf16_t*>(out), ldo, P, T, Tp, row0, H, D, win, th, sd, sc, seed_dev);
  GT_RET();
}

extern "C" size_t gt_attn_bwd_workspace_bytes(int B, int T, int H)
{
  if (B <= 0 || T <= 0 || H <= 0) return 0;
  const size_t generic = (size_t)B * H * T * T * sizeof(float);
  const size_t mfma = gt_attn_bwd_mfma_ws_bytes(B, T, H);
  return (generic > mfma ? generic : mfma) + 256;
}

extern "C" int gt_attn_bwd(const void* q, const void* k, const void* v, int ld, const float* Ek, const float* Ev,
                           const int32_t* lens, const void* dout, int lddo, const float* P, void* workspace, size_t workspace_bytes,
                           void* dq, void* dk, void* dv, int lddq, float* dEk, float* dEv,
                           int B, int T, int Tp, const int32_t* row0, int H, int D, int win, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream)
{
  if (!q || !k || !v || !Ek || !Ev || !lens || !dout || !P || !workspace || !dq || !dk || !dv || !dEk || !dEv) return GT_E_INVAL;
  if (D > AT_MAXD || (D & 1) || win < 0 || drop_p >= 1.f) return GT_E_UNSUPPORTED;
  if (workspace_bytes < gt_attn_bwd_workspace_bytes(B, T, H)) return GT_E_INVAL;
  float* dS_ws = static_cast<float*>(workspace);
  {
    uint32_t th, sd; float sc; fill_drop(drop_p, drop_seed, th, sd, sc);
    {
      const int rc = gt_attn_bwd_mfma_impl(q, k, v, ld, Ek, Ev, lens, dout, lddo, P, workspace, workspace_bytes, dq, dk, dv, lddq,
                                           dEk, dEv, B, T, Tp, row0, H, D, win, th, sd, sc, seed_dev, stream);
      if (rc != 1) return rc;
    }
  }
  const size_t lds1 = attn_lds(T, D, win, (size_t)2 * (2 * win + 1) * D + (size_t)AT_QT * 2 * D + (size_t)AT_QT * T);
  size_t halfs2 = (size_t)T * (D + 2); halfs2 += halfs2 & 1;
  const size_t lds2 = halfs2 * 2 + (size_t)AT_QT * 2 * T * 4;
  if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return GT_E_UNSUPPORTED;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_attn_bwd_q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GT_E_LAUNCH;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_attn_bwd_kv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GT_E_LAUNCH;
    attr = true;
  }
  uint32_t th, sd; float sc; fill_drop(drop_p, drop_seed, th, sd, sc);
  const dim3 grid((T + AT_QT - 1) / AT_QT, H, B);
  hipLaunchKernelGGL(gt_attn_bwd_q_kernel, grid, dim3(256), lds1, GT_ST(stream),
                     static_cast<const bf16_t*>(q), static_cast<const bf16_t*>(k), static_cast<const bf16_t*>(v), ld, Ek, Ev, lens,
                     static_cast<const bf16_t*>(dout), lddo, P, dS_ws, static_cast<bf16_t*>(dq), lddq, dEk, dEv,
                     T, Tp, row0, H, D, win, th, sd, sc, seed_dev);
  hipLaunchKernelGGL(gt_attn_bwd_kv_kernel, grid, dim3(256), lds2, GT_ST(stream),
                     static_cast<const bf16_t*>(q), ld, static_cast<const bf16_t*>(dout), lddo, P, dS_ws, lens,
                     static_cast<bf16_t*>(dk), static_cast<bf16_t*>(dv), lddq, T, Tp, row0, H, D, th, sd, sc, seed_dev);
  GT_RET();
}

extern "C" int gt_embedding_fwd(const int64_t* ids, const float* emb, const int32_t* lens, float* out_f32, void* out_bf16,
                                int B, int T, int Tp, const int32_t* row0, int R, int C, int ld, float scale, void* stream)
{
  if (!ids || !emb || !lens || (!out_f32 && !out_bf16) || B <= 0 || T <= 0 || C <= 0 || R <= 0 || ld < C) return GT_E_INVAL;
  if (!row0 && R != B * Tp) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_embedding_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, GT_ST(stream), ids, emb, lens, out_f32,
                     static_cast<bf16_t*>(out_bf16), B, T, Tp, C, scale, row0, R, ld);
  GT_RET();
}
extern "C" int gt_rows_add_cond(const float* x, int ldx, const void* xb, int ldxb, const float* cond, const float* rowmask,
                               float* out, int ldo, void* outb, int ldob, int B, int R, int C, int Tp, const int32_t* row0, void* stream)
{
  if ((!x && !xb) || !cond || !rowmask || (!out && !outb) || B <= 0 || R <= 0 || C <= 0) return GT_E_INVAL;
  if (!row0 && R != B * Tp) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_rows_add_cond_kernel, dim3((R + 3) / 4), dim3(256), 0, GT_ST(stream), x, ldx, static_cast<const bf16_t*>(xb), ldxb,
                     cond, rowmask, out, ldo, static_cast<bf16_t*>(outb), ldob, B, R, C, Tp, row0);
  GT_RET();
}
extern "C" int gt_rows_utt_sum(const void* y, int ldy, int is_f32, const float* rowmask, float* out, int ldo, int accumulate,
                              int B, int R, int C, int Tp, const int32_t* row0, void* stream)
{
  if (!y || !out || B <= 0 || R <= 0 || C <= 0 || ldy < C || ldo < C) return GT_E_INVAL;
  if (!row0 && R != B * Tp) return GT_E_INVAL;
  const dim3 grid(B, (C + 63) / 64);
  if (is_f32) hipLaunchKernelGGL(gt_rows_utt_sum_kernel<true>, grid, dim3(1024), 0, GT_ST(stream), y, ldy, rowmask, out, ldo, accumulate, B, C, Tp, row0);
  else        hipLaunchKernelGGL(gt_rows_utt_sum_kernel<false>, grid, dim3(1024), 0, GT_ST(stream), y, ldy, rowmask, out, ldo, accumulate, B, C, Tp, row0);
  GT_RET();
}
extern "C" int gt_rows_ctx_fill(const int32_t* row0, const int32_t* lens, int64_t* rowbatch, int32_t* rowframe, float* rowmask, int32_t* rowutt,
                               int B, int R, void* stream)
{
  if (!row0 || !lens || !rowbatch || !rowframe || !rowmask || B <= 0 || R <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_rows_ctx_fill_kernel, dim3((R + 255) / 256), dim3(256), 0, GT_ST(stream), row0, lens, rowbatch, rowframe, rowmask, rowutt, B, R);
  GT_RET();
}
extern "C" int gt_step_zero(const gt_step_zero_args* args, void* stream)
{
  if (!args || args->n < 0 || args->n > GT_ZERO_MAX) return GT_E_INVAL;
  size_t words = 0;
  for (int j = 0; j < args->n; ++j) {
    if (!args->ptr[j] || (args->bytes[j] & 15) || ((uintptr_t)args->ptr[j] & 15)) return GT_E_ALIGN;
    words += args->bytes[j] >> 4;
  }
  if (!words && !args->seed_word) return GT_OK;
  if (words > ((size_t)1 << 31)) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_step_zero_kernel, dim3((unsigned)((words + 255) / 256) + (words ? 0 : 1)), dim3(256), 0, GT_ST(stream), *args);
  GT_RET();
}
extern "C" int gt_step_inputs(const gt_step_inputs_args* args, void* stream)
{
  if (!args || args->n_copy < 0 || args->n_copy > GT_STEP_MAX_COPIES || args->n_ctx < 0 || args->n_ctx > GT_STEP_MAX_CTX) return GT_E_INVAL;
  gt_step_inputs_args a = *args;
  int blocks = 0;
  for (int j = 0; j < a.n_copy; ++j) {
    gt_step_copy& c = a.copy[j];
    if (!c.src || !c.dst || c.rows <= 0 || c.src_words < 0 || c.dst_words < c.src_words || c.dst_words <= 0) return GT_E_INVAL;
    if (((uintptr_t)c.src | (uintptr_t)c.dst) & 3) return GT_E_ALIGN;
    c.blk0 = blocks;
    blocks += (int)(((size_t)c.rows * c.dst_words + 1023) / 1024);
  }
  for (int j = 0; j < a.n_ctx; ++j) {
    gt_step_ctx& c = a.ctx[j];
    if (!c.geo_src || !c.geo_dst || !c.rowbatch || !c.rowframe || !c.rowmask || c.B <= 0 || c.B > GT_STEP_MAX_B || c.R <= 0) return GT_E_INVAL;
    c.blk0 = blocks;
    blocks += (c.R + 255) / 256;
  }
  if (!blocks) return GT_OK;
  hipLaunchKernelGGL(gt_step_inputs_kernel, dim3(blocks), dim3(256), 0, GT_ST(stream), a);
  GT_RET();
}
extern "C" int gt_embedding_bwd(const int64_t* ids, const float* dx, const int32_t* lens, float* demb,
                                int B, int T, int Tp, const int32_t* row0, int R, int C, int ld, float scale, void* stream)
{
  if (!ids || !dx || !lens || !demb || B <= 0 || T <= 0 || C <= 0 || R <= 0 || ld < C) return GT_E_INVAL;
  if (!row0 && R != B * Tp) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_embedding_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, GT_ST(stream), ids, dx, lens, demb, B, T, Tp, C, scale, row0, R, ld);
  GT_RET();
}

extern "C" int gt_logp_f32(const float* x_m, const float* x_logs, const float* z, float* logp, int B, int C, int Tx, int Ty, void* stream)
{
  if (!x_m || !z || !logp || B <= 0 || C <= 0 || (C & 1) || Tx <= 0 || Ty <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_logp_kernel, dim3((Ty + 127) / 128, (Tx + 31) / 32, B), dim3(256), 0, GT_ST(stream), x_m, x_logs, z, logp, B, C, Tx, Ty);
  GT_RET();
}

extern "C" int gt_prior_expand(const float* x_m, const int32_t* frame2token, float* z_m, int B, int C, int Tx, int Ty, void* stream)
{
  if (!x_m || !frame2token || !z_m || B <= 0 || C <= 0 || Tx <= 0 || Ty <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_prior_expand_kernel, dim3(((size_t)B * C * Ty + 255) / 256), dim3(256), 0, GT_ST(stream), x_m, frame2token, z_m, B, C, Tx, Ty);
  GT_RET();
}
extern "C" int gt_prior_expand_bwd(const float* dz_m, const int32_t* frame2token, float* dx_m, int B, int C, int Tx, int Ty, void* stream)
{
  if (!dz_m || !frame2token || !dx_m || B <= 0 || C <= 0 || Tx <= 0 || Ty <= 0) return GT_E_INVAL;
  if (Tx > 512) return GT_E_UNSUPPORTED;                      // the MAS kernel's own limit
  hipLaunchKernelGGL(gt_prior_expand_bwd_kernel, dim3(((size_t)B * C + 3) / 4), dim3(256), 0, GT_ST(stream), dz_m, frame2token, dx_m, B, C, Tx, Ty);
  GT_RET();
}
extern "C" int gt_mle_sums(const float* z, const float* m, const float* logs, float* acc2, size_t n, void* stream)
{
  if (!z || !m || !acc2) return GT_E_INVAL;
  if (n == 0) return GT_OK;
  if (((uintptr_t)z | (uintptr_t)m | (uintptr_t)logs) & 15) return GT_E_ALIGN;
  if ((uintptr_t)acc2 & 7) return GT_E_ALIGN;
  hipLaunchKernelGGL(gt_mle_sums_kernel, dim3(GT_MLE_PARTS), dim3(256), 0, GT_ST(stream), z, m, logs, acc2, n);
  GT_RET();
}
__global__ void gt_mark_kernel(unsigned long long* slot) { *slot = (unsigned long long)wall_clock64(); }

extern "C" int gt_mark(unsigned long long* slot, void* stream)
{
  if (!slot) return -1;
  hipLaunchKernelGGL(gt_mark_kernel, dim3(1), dim3(1), 0, GT_ST(stream), slot);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int gt_length_mask(const void* lengths, int is_int64, float* mask, int B, int T, void* stream)
{
  if (!lengths || !mask || B <= 0 || T <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_length_mask_kernel, dim3(((size_t)B * T + 255) / 256), dim3(256), 0, GT_ST(stream), lengths, is_int64, mask, B, T);
  GT_RET();
}
extern "C" int gt_mle_finish(const float* acc2, const float* logdet, const float* mask, int n_mask, int B, int C, float* out2, void* stream)
{
  if (!acc2 || !logdet || !mask || !out2 || B <= 0 || C <= 0 || n_mask <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_mle_finish_kernel, dim3(1), dim3(1024), 0, GT_ST(stream), acc2, logdet, mask, n_mask, B, C, out2);
  GT_RET();
}
extern "C" int gt_duration_loss_fwd(const float* logw, const float* w, const int32_t* x_lengths, int B, int Tx, float* l_length, void* stream)
{
  if (!logw || !w || !x_lengths || !l_length || B <= 0 || Tx <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_duration_loss_fwd_kernel, dim3(B), dim3(64), 0, GT_ST(stream), logw, w, x_lengths, B, Tx, l_length);
  GT_RET();
}
extern "C" int gt_duration_loss_bwd(const float* logw, const float* w, const int32_t* x_lengths, const float* g, int B, int Tx,
                                    float* dlogw, void* stream)
{
  if (!logw || !w || !x_lengths || !g || !dlogw || B <= 0 || Tx <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_duration_loss_bwd_kernel, dim3(B), dim3(64), 0, GT_ST(stream), logw, w, x_lengths, g, B, Tx, dlogw);
  GT_RET();
}
extern "C" int gt_mle_bwd(const float* z, const float* m, const float* logs, const float* gscale, float* dz, float* dm, float* dlogs,
                          size_t n, const float* gdenom, float* dlogdet, int B, void* stream)
{
  if (!z || !m || !gscale) return GT_E_INVAL;
  if (n == 0) return GT_OK;
  size_t blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gt_mle_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, GT_ST(stream), z, m, logs, gscale, dz, dm, dlogs, n, gdenom, dlogdet, B);
  GT_RET();
}
