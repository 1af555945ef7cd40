// Relative-position multi-head attention forward on bf16 MFMA for gfx950 (reference
// attentions.py:241-336 in its 9-diagonal band form; D = 96, window 4, T <= 256).
//
// One workgroup = 4 waves = 128 query rows of one (utterance, head); K and V of that (utterance,
// head) are staged once in LDS.  Per wave (32 query rows), everything stays in registers:
//   S^T = K Q^T            MFMA A = K rows from LDS (ds_read_b128, pitch 208 B), B = Q^T fragments loaded
//                          straight from HBM; a lane then owns ONE query and 16 keys per 32-key tile, so
//                          the softmax row reductions are in-lane + one xor-32 exchange.
//   QE  = Ek Q^T           the relative-key logits q_i.Ek[r] as one more MFMA chain (Ek padded to 32 rows)
//   P   = softmax((S^T + band(QE)) / sqrt(D)), saved fp32 for the backward, dropout replayable
//   O^T = V^T P^T          "accumulator tile as the next MFMA's operand" (guide §3): P^T registers are
//                          converted pairwise to bf16 and fed as B; V^T comes from LDS through
//                          ds_read_b64_tr_b16 (pitch 192 B) in the permuted k order that map requires.
//   O^T += Ev^T band(P)^T  the relative-value term as one K=16 MFMA per 32 channels.
#include <stdlib.h>
#include "common.h"
#include "../../include/glowtts_hip.h"
#include "internal.h"

namespace {

constexpr int HALO = GT_HALO;
constexpr int D = 96, WIN = 4, NW = 9;
constexpr int KP = 104;            // K / Ek pitch in halfs (208 B): conflict-free ds_read_b128 over 16 rows
constexpr int VP = 96;             // V pitch in halfs (192 B): conflict-free transposing reads

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4_t;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8_t;

__device__ __forceinline__ bf16x8_t tr_frag8(const bf16_t* p0, const bf16_t* p1) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)p1);
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t pack8(const float* f) {
  const uint4 u = make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
  return __builtin_bit_cast(bf16x8_t, u);
}

template <int NT>   // key tiles of 32 (T <= 32*NT)
__global__ __launch_bounds__(256, 1) void gt_attn_fwd_mfma_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int ld,
    const float* __restrict__ Ek, const float* __restrict__ Ev, const int32_t* __restrict__ lens,
    bf16_t* __restrict__ out, int ldo, float* __restrict__ Pout,
    int T, int Tp, const int32_t* row0, int H, uint32_t drop_thresh, uint32_t drop_seed, float drop_scale, const uint32_t* __restrict__ seed_dev)
{
  if (seed_dev) drop_seed ^= *seed_dev;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TPAD = NT * 32;
  bf16_t* Ks  = reinterpret_cast<bf16_t*>(smem);                 // [TPAD][KP]
  bf16_t* Vs  = Ks + TPAD * KP;                                  // [TPAD][VP]
  bf16_t* Eks = Vs + TPAD * VP;                                  // [32][KP]   rows >= 9 are zero
  bf16_t* EvT = Eks + 32 * KP;                                   // [96][16]   EvT[d][r], r >= 9 zero
  float*  QE  = reinterpret_cast<float*>(EvT + D * 16);          // [4 waves][32][NW]
  bf16_t* PB  = reinterpret_cast<bf16_t*>(QE + 4 * 32 * NW);     // [4 waves][32][16]

  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int len = lens[b];
  const size_t rbase = (size_t)gt_row_base(row0, b, Tp) + HALO;
  // rows this utterance owns behind rbase (frames + trailing halo): in the ragged layout frame indices past them belong
  // to the NEXT utterance — reads are clamped onto the (zero) trailing halo row, stores are dropped
  const int nv1 = gt_row_count(row0, b, Tp) - HALO - 1;
  auto RW = [&](int t) { return rbase + (size_t)(t < nv1 ? t : nv1); };

  // ---- stage K, V (zero rows >= T), Ek, Ev^T
  for (int i = tid; i < TPAD * (D / 8); i += 256) {
    const int j = i / (D / 8), c8 = i - j * (D / 8);
    uint4 kk = make_uint4(0, 0, 0, 0), vv = kk;
    if (j < T) {
      kk = *reinterpret_cast<const uint4*>(k + RW(j) * ld + h * D + c8 * 8);
      vv = *reinterpret_cast<const uint4*>(v + RW(j) * ld + h * D + c8 * 8);
    }
    *reinterpret_cast<uint4*>(Ks + j * KP + c8 * 8) = kk;
    *reinterpret_cast<uint4*>(Vs + j * VP + c8 * 8) = vv;
  }
  for (int i = tid; i < 32 * D; i += 256) { const int rr = i / D, c = i - rr * D; Eks[rr * KP + c] = rr < NW ? f2bf(Ek[rr * D + c]) : (bf16_t)0; }
  for (int i = tid; i < D * 16; i += 256) { const int d = i >> 4, rr = i & 15; EvT[i] = rr < NW ? f2bf(Ev[rr * D + d]) : (bf16_t)0; }
  for (int i = tid; i < 4 * 32 * 16; i += 256) PB[i] = 0;
  __syncthreads();

  const int i0 = blockIdx.x * 128 + 32 * w;
  if (i0 >= T) return;                                           // wave-uniform, after the only block barrier
  const int i = i0 + r;                                          // this lane's query
  const int ic = i < T ? i : T - 1;
  float* qe = QE + w * 32 * NW;
  bf16_t* pb = PB + w * 32 * 16;

  // ---- Q^T fragments straight from HBM: lane (query r, k-half hh), 6 k-steps of 16 channels
  bf16x8_t qf[6];
#pragma unroll
  for (int ks = 0; ks < 6; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(q + RW(ic) * ld + h * D + ks * 16 + 8 * hh);

  // ---- QE = Ek Q^T  (rows r' < 9 used)
  {
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(Eks + r * KP + ks * 16 + 8 * hh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf[ks], acc, 0, 0, 0);
    }
    // D layout: column = query (lane&31), row r' = (e&3) + 8*(e>>2) + 4*hh
#pragma unroll
    for (int e = 0; e < 4; ++e) qe[r * NW + e + 4 * hh] = acc[e];
    if (hh == 0) qe[r * NW + 8] = acc[4];
  }

  // ---- S^T = K Q^T
  f32x16_t s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s[t][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(Ks + (32 * t + r) * KP + ks * 16 + 8 * hh);
      s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf[ks], s[t], 0, 0, 0);
    }
  }
  __builtin_amdgcn_wave_barrier();                               // qe written by both lane halves

  // ---- softmax over keys (in-lane + xor 32)
  const float inv_sqrt = rsqrtf((float)D);
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh;
      float sc = s[t][e];
      const int rel = j - i + WIN;
      if ((unsigned)rel <= 2u * WIN) sc += qe[r * NW + rel];
      sc *= inv_sqrt;
      if (j >= T) sc = -3.0e38f;                                 // not a key at all
      else if (j >= len || i >= len) sc = -1e4f;                 // masked_fill(mask == 0, -1e4), attentions.py:260
      s[t][e] = sc;
      mx = fmaxf(mx, sc);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float ex = __expf(s[t][e] - mx); s[t][e] = ex; den += ex; }
  den += __shfl_xor(den, 32);
  const float rden = 1.0f / den;
  float* prow = Pout + (((size_t)b * H + h) * T + ic) * T;
  const uint32_t drow = (uint32_t)((b * H + h) * T + i);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j0 = 32 * t + 8 * g + 4 * hh;
      float p4[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) p4[e2] = s[t][4 * g + e2] * rden;
      if (i < T) {
        if (j0 + 3 < T && (T & 3) == 0) *reinterpret_cast<float4*>(prow + j0) = make_float4(p4[0], p4[1], p4[2], p4[3]);
        else {
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) if (j0 + e2 < T) prow[j0 + e2] = p4[e2];
        }
      }
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const int j = j0 + e2;
        float pd = p4[e2];
        if (drop_thresh) pd = drop_keep(drop_seed, drow, j, drop_thresh) ? pd * drop_scale : 0.f;
        s[t][4 * g + e2] = pd;
        const int rel = j - i + WIN;
        if ((unsigned)rel <= 2u * WIN && j < T) pb[r * 16 + rel] = f2bf(pd);
      }
    }
  __builtin_amdgcn_wave_barrier();

  // ---- O^T = V^T P^T (+ Ev^T band(P)^T)
  const int li = lane & 15, qd = li >> 2, pp = li & 3, colhalf = ((lane >> 4) & 1) * 16;
  f32x16_t o[3];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float f8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f8[e] = s[t][8 * s2 + e];
      const bf16x8_t pf = pack8(f8);                             // k order: row 16*s2 + 8*(e>>2) + 4*hh + (e&3) of the tile
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const bf16_t* va = Vs + (32 * t + 16 * s2 + 4 * hh + qd) * VP + 32 * dt + colhalf + 4 * pp;
        const bf16x8_t af = tr_frag8(va, va + 8 * VP);           // V^T[d][keys 4hh..4hh+3 | 8+4hh..8+4hh+3]
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf, o[dt], 0, 0, 0);
      }
    }
  {
    const bf16x8_t bfp = *reinterpret_cast<const bf16x8_t*>(pb + r * 16 + 8 * hh);      // band(P)^T: k = rel
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(EvT + (32 * dt + r) * 16 + 8 * hh);
      o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfp, o[dt], 0, 0, 0);
    }
  }
  if (i < T && i <= nv1) {
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * dt + 8 * g + 4 * hh;
        *reinterpret_cast<uint2*>(out + (rbase + i) * ldo + h * D + d) =
            make_uint2(pack2bf(o[dt][4 * g], o[dt][4 * g + 1]), pack2bf(o[dt][4 * g + 2], o[dt][4 * g + 3]));
      }
  }
}

template <int NT>
int launch_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ld, const float* Ek, const float* Ev, const int32_t* lens,
               bf16_t* out, int ldo, float* P, int B, int T, int Tp, const int32_t* row0, int H, uint32_t th, uint32_t sd, float sc, const uint32_t* seed_dev, hipStream_t st)
{
  constexpr int TPAD = NT * 32;
  const size_t lds = (size_t)TPAD * KP * 2 + (size_t)TPAD * VP * 2 + 32 * KP * 2 + D * 16 * 2 + 4 * 32 * NW * 4 + 4 * 32 * 16 * 2;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_attn_fwd_mfma_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return GT_E_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(gt_attn_fwd_mfma_kernel<NT>, dim3((T + 127) / 128, H, B), dim3(256), lds, st,
                     q, k, v, ld, Ek, Ev, lens, out, ldo, P, T, Tp, row0, H, th, sd, sc, seed_dev);
  return gt_launch_status(__func__);
}

// -----------------------------------------------------------------------------------------
// Long sequences (256 < T <= 384: configs/base_blank.json interleaves blanks, T_x <= 375).  K and V no longer fit the
// LDS together and NT score tiles no longer fit the register file, so: V stays in LDS (it is read through transposing
// reads), K fragments (plain row reads) come straight from L2 with a one-tile register prefetch, and the wave walks the
// key tiles twice — pass 1 an online max / denominator, pass 2 recomputes each tile's scores, normalises, writes P and
// feeds O^T at once.  Live state: one tile + the O^T accumulators, whatever NT is.
template <int NT>
__global__ __launch_bounds__(256, 1) void gt_attn_fwd_mfma_long_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int ld,
    const float* __restrict__ Ek, const float* __restrict__ Ev, const int32_t* __restrict__ lens,
    bf16_t* __restrict__ out, int ldo, float* __restrict__ Pout,
    int T, int Tp, const int32_t* row0, int H, uint32_t drop_thresh, uint32_t drop_seed, float drop_scale, const uint32_t* __restrict__ seed_dev)
{
  if (seed_dev) drop_seed ^= *seed_dev;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TPAD = NT * 32;
  bf16_t* Vs  = reinterpret_cast<bf16_t*>(smem);                 // [TPAD][VP]
  bf16_t* Eks = Vs + TPAD * VP;                                  // [32][KP]   rows >= 9 are zero
  bf16_t* EvT = Eks + 32 * KP;                                   // [96][16]   EvT[d][r], r >= 9 zero
  float*  QE  = reinterpret_cast<float*>(EvT + D * 16);          // [4 waves][32][NW]
  bf16_t* PB  = reinterpret_cast<bf16_t*>(QE + 4 * 32 * NW);     // [4 waves][32][16]

  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int len = lens[b];
  const size_t rbase = (size_t)gt_row_base(row0, b, Tp) + HALO;
  const int nv1 = gt_row_count(row0, b, Tp) - HALO - 1;          // see gt_attn_fwd_mfma_kernel
  auto RW = [&](int t) { return rbase + (size_t)(t < nv1 ? t : nv1); };

  for (int i = tid; i < TPAD * (D / 8); i += 256) {
    const int j = i / (D / 8), c8 = i - j * (D / 8);
    uint4 vv = make_uint4(0, 0, 0, 0);
    if (j < T) vv = *reinterpret_cast<const uint4*>(v + RW(j) * ld + h * D + c8 * 8);
    *reinterpret_cast<uint4*>(Vs + j * VP + c8 * 8) = vv;
  }
  for (int i = tid; i < 32 * D; i += 256) { const int rr = i / D, c = i - rr * D; Eks[rr * KP + c] = rr < NW ? f2bf(Ek[rr * D + c]) : (bf16_t)0; }
  for (int i = tid; i < D * 16; i += 256) { const int d = i >> 4, rr = i & 15; EvT[i] = rr < NW ? f2bf(Ev[rr * D + d]) : (bf16_t)0; }
  for (int i = tid; i < 4 * 32 * 16; i += 256) PB[i] = 0;
  __syncthreads();

  const int i0 = blockIdx.x * 128 + 32 * w;
  if (i0 >= T) return;                                           // wave-uniform, after the only block barrier
  const int i = i0 + r;
  const int ic = i < T ? i : T - 1;
  float* qe = QE + w * 32 * NW;
  bf16_t* pb = PB + w * 32 * 16;

  bf16x8_t qf[6];
#pragma unroll
  for (int ks = 0; ks < 6; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(q + RW(ic) * ld + h * D + ks * 16 + 8 * hh);
  {
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(Eks + r * KP + ks * 16 + 8 * hh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf[ks], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) qe[r * NW + e + 4 * hh] = acc[e];
    if (hh == 0) qe[r * NW + 8] = acc[4];
  }
  __builtin_amdgcn_wave_barrier();

  const int nt = (T + 31) >> 5;                                  // key tiles that hold keys (wave-uniform)
  const float inv_sqrt = rsqrtf((float)D);
  // K fragments of key tile t: lane (key r of the tile, k-half hh); rows >= T are zero
  auto load_k = [&](int t, bf16x8_t* kf) {
    const int j = 32 * t + r;
    if (j < T) {
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(k + RW(j) * ld + h * D + ks * 16 + 8 * hh);
    } else {
      const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) kf[ks] = __builtin_bit_cast(bf16x8_t, z);
    }
  };
  // masked, scaled scores of one tile (element e <-> key 32t + (e&3) + 8(e>>2) + 4hh)
  auto scores = [&](int t, const bf16x8_t* kf, f32x16_t& st) {
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], st, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh;
      float sc = st[e];
      const int rel = j - i + WIN;
      if ((unsigned)rel <= 2u * WIN) sc += qe[r * NW + rel];
      sc *= inv_sqrt;
      if (j >= T) sc = -3.0e38f;                                 // not a key at all
      else if (j >= len || i >= len) sc = -1e4f;                 // masked_fill(mask == 0, -1e4), attentions.py:260
      st[e] = sc;
    }
  };

  // ---- pass 1: online max / denominator over this lane's keys, then the two lane halves are merged
  float mx = -3.0e38f, den = 0.f;
  bf16x8_t kf[6], kn[6];
  load_k(0, kf);
#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) load_k(t + 1, kn);
    f32x16_t st;
    scores(t, kf, st);
    float tm = st[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) tm = fmaxf(tm, st[e]);
    const float mn = fmaxf(mx, tm);
    float add = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) add += __expf(st[e] - mn);
    den = den * __expf(mx - mn) + add;
    mx = mn;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) kf[ks] = kn[ks];
  }
  {
    const float mo = __shfl_xor(mx, 32), dn = __shfl_xor(den, 32);
    const float mm = fmaxf(mx, mo);
    den = den * __expf(mx - mm) + dn * __expf(mo - mm);
    mx = mm;
  }
  const float rden = 1.0f / den;

  // ---- pass 2: P = softmax, dropout, O^T = V^T P^T (+ Ev^T band(P)^T)
  float* prow = Pout + (((size_t)b * H + h) * T + ic) * T;
  const uint32_t drow = (uint32_t)((b * H + h) * T + i);
  const int li = lane & 15, qd = li >> 2, pp = li & 3, colhalf = ((lane >> 4) & 1) * 16;
  const bool vec = (T & 3) == 0;
  f32x16_t o[3];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  }
  load_k(0, kf);
#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) load_k(t + 1, kn);
    f32x16_t st;
    scores(t, kf, st);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j0 = 32 * t + 8 * g + 4 * hh;
      float p4[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) p4[e2] = __expf(st[4 * g + e2] - mx) * rden;
      if (i < T) {
        if (vec && j0 + 3 < T) *reinterpret_cast<float4*>(prow + j0) = make_float4(p4[0], p4[1], p4[2], p4[3]);
        else {
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) if (j0 + e2 < T) prow[j0 + e2] = p4[e2];
        }
      }
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const int j = j0 + e2;
        float pd = p4[e2];
        if (drop_thresh) pd = drop_keep(drop_seed, drow, j, drop_thresh) ? pd * drop_scale : 0.f;
        st[4 * g + e2] = pd;
        const int rel = j - i + WIN;
        if ((unsigned)rel <= 2u * WIN && j < T) pb[r * 16 + rel] = f2bf(pd);
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float f8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f8[e] = st[8 * s2 + e];
      const bf16x8_t pf = pack8(f8);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const bf16_t* va = Vs + (32 * t + 16 * s2 + 4 * hh + qd) * VP + 32 * dt + colhalf + 4 * pp;
        const bf16x8_t af = tr_frag8(va, va + 8 * VP);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf, o[dt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) kf[ks] = kn[ks];
  }
  __builtin_amdgcn_wave_barrier();
  {
    const bf16x8_t bfp = *reinterpret_cast<const bf16x8_t*>(pb + r * 16 + 8 * hh);      // band(P)^T: k = rel
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(EvT + (32 * dt + r) * 16 + 8 * hh);
      o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfp, o[dt], 0, 0, 0);
    }
  }
  if (i < T && i <= nv1) {
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * dt + 8 * g + 4 * hh;
        *reinterpret_cast<uint2*>(out + (rbase + i) * ldo + h * D + d) =
            make_uint2(pack2bf(o[dt][4 * g], o[dt][4 * g + 1]), pack2bf(o[dt][4 * g + 2], o[dt][4 * g + 3]));
      }
  }
}

template <int NT>
int launch_fwd_long(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ld, const float* Ek, const float* Ev, const int32_t* lens,
                    bf16_t* out, int ldo, float* P, int B, int T, int Tp, const int32_t* row0, int H, uint32_t th, uint32_t sd, float sc, const uint32_t* seed_dev, hipStream_t st)
{
  constexpr int TPAD = NT * 32;
  const size_t lds = (size_t)TPAD * VP * 2 + 32 * KP * 2 + D * 16 * 2 + 4 * 32 * NW * 4 + 4 * 32 * 16 * 2;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_attn_fwd_mfma_long_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return GT_E_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(gt_attn_fwd_mfma_long_kernel<NT>, dim3((T + 127) / 128, H, B), dim3(256), lds, st,
                     q, k, v, ld, Ek, Ev, lens, out, ldo, P, T, Tp, row0, H, th, sd, sc, seed_dev);
  return gt_launch_status(__func__);
}

// =========================================================================================
// Backward.  Pass 1 (per 32-query block, same wave/lane mapping as the forward):
//   dPd^T = V dO^T (+ band: Ev dO^T)      dP = dropout'(dPd)       Dsum_i = sum_j dP P
//   dS^T  = P (dP - Dsum) / sqrt(D), masked entries 0
//   dQ^T  = K^T dS^T + Ek^T band(dS)^T     (accumulator-as-operand, K^T through transposing reads)
//   dEk  += band(dS)^T Q,  dEv += band(dropout(P))^T dO   (MFMA over the 32 queries of the wave)
//   dS^T and dropout(P)^T leave as bf16 [B,H,T,TI] (query index contiguous) for pass 2.
// Pass 2 (per 32-key block): dK^T = Q^T dS, dV^T = dO^T dropout(P) with Q^T / dO^T through
// transposing reads of LDS-staged Q / dO and the B operands straight from the pass-1 buffers.
constexpr int BTP = 40;            // pitch (halfs) of the transposed band tables [16][32 + pad]

template <int NT, int WV, bool ONE_TILE>
__global__ __launch_bounds__(64 * WV, 1) void gt_attn_bwd_q_mfma_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int ld,
    const float* __restrict__ Ek, const float* __restrict__ Ev, const int32_t* __restrict__ lens,
    const bf16_t* __restrict__ dout, int lddo, const float* __restrict__ P,
    bf16_t* __restrict__ dST, bf16_t* __restrict__ PdT, int TI,
    bf16_t* __restrict__ dq, int lddq, float* __restrict__ dEk, float* __restrict__ dEv,
    int T, int Tp, const int32_t* row0, int H, uint32_t drop_thresh, uint32_t drop_seed, float drop_scale, const uint32_t* __restrict__ seed_dev)
{
  if (seed_dev) drop_seed ^= *seed_dev;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TPAD = NT * 32, NTH = 64 * WV;
  constexpr bool VG = NT > 8;                                    // long sequences: V rows (plain row reads) come from L2, not LDS
  bf16_t* Vs  = reinterpret_cast<bf16_t*>(smem);                 // [TPAD][KP]  A of dPd^T (ds_read_b128); absent if VG
  bf16_t* Ks  = Vs + (VG ? 0 : TPAD * KP);                       // [TPAD][VP]  A of dQ^T (transposing reads)
  bf16_t* Evs = Ks + TPAD * VP;                                  // [32][KP]    rows >= 9 zero
  bf16_t* EkT = Evs + 32 * KP;                                   // [96][16]
  float*  Acc = reinterpret_cast<float*>(EkT + D * 16);          // [2][NW][D]  block-local dEk | dEv
  float*  DOE = Acc + 2 * NW * D;                                // [WV][32][NW]
  bf16_t* WB  = reinterpret_cast<bf16_t*>(DOE + WV * 32 * NW);   // per wave: dSB[32][16] | dSBT[16][BTP] | PdBT[16][BTP] | Qw[32][VP] | dOw[32][VP]
  constexpr int WBN = 32 * 16 + 2 * 16 * BTP + 2 * 32 * VP;

  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int len = lens[b];
  const size_t rbase = (size_t)gt_row_base(row0, b, Tp) + HALO;
  // rows this utterance owns behind rbase (frames + trailing halo): in the ragged layout frame indices past them belong
  // to the NEXT utterance — reads are clamped onto the (zero) trailing halo row, stores are dropped
  const int nv1 = gt_row_count(row0, b, Tp) - HALO - 1;
  auto RW = [&](int t) { return rbase + (size_t)(t < nv1 ? t : nv1); };

  for (int i = tid; i < TPAD * (D / 8); i += NTH) {
    const int j = i / (D / 8), c8 = i - j * (D / 8);
    uint4 kk = make_uint4(0, 0, 0, 0), vv = kk;
    if (j < T) {
      kk = *reinterpret_cast<const uint4*>(k + RW(j) * ld + h * D + c8 * 8);
      if (!VG) vv = *reinterpret_cast<const uint4*>(v + RW(j) * ld + h * D + c8 * 8);
    }
    *reinterpret_cast<uint4*>(Ks + j * VP + c8 * 8) = kk;
    if (!VG) *reinterpret_cast<uint4*>(Vs + j * KP + c8 * 8) = vv;
  }
  for (int i = tid; i < 32 * D; i += NTH) { const int rr = i / D, c = i - rr * D; Evs[rr * KP + c] = rr < NW ? f2bf(Ev[rr * D + c]) : (bf16_t)0; }
  for (int i = tid; i < D * 16; i += NTH) { const int d = i >> 4, rr = i & 15; EkT[i] = rr < NW ? f2bf(Ek[rr * D + d]) : (bf16_t)0; }
  for (int i = tid; i < 2 * NW * D; i += NTH) Acc[i] = 0.f;
  for (int i = tid; i < WV * (32 * 16 + 2 * 16 * BTP); i += NTH) {            // band tables start at zero
    const int ww = i / (32 * 16 + 2 * 16 * BTP), o = i - ww * (32 * 16 + 2 * 16 * BTP);
    WB[ww * WBN + o] = 0;
  }
  __syncthreads();

  const int i0 = (blockIdx.x * WV + w) * 32;
  const bool active = i0 < T;                                    // wave-uniform
  const int i = i0 + r;
  const int ic = i < T ? i : T - 1;
  float* doe = DOE + w * 32 * NW;
  bf16_t* dSB = WB + w * WBN;
  bf16_t* dSBT = dSB + 32 * 16;
  bf16_t* PdBT = dSBT + 16 * BTP;
  bf16_t* Qw = PdBT + 16 * BTP;
  bf16_t* dOw = Qw + 32 * VP;
  const int li = lane & 15, qd = li >> 2, pp = li & 3, colhalf = ((lane >> 4) & 1) * 16;

  if (active) {
    // per-wave Q / dO tiles (rows >= T zero) for the dEk / dEv contraction
    for (int c = lane; c < 32 * (D / 8); c += 64) {
      const int rr = c / (D / 8), c8 = c - rr * (D / 8);
      uint4 qq = make_uint4(0, 0, 0, 0), dd = qq;
      if (i0 + rr < T) {
        qq = *reinterpret_cast<const uint4*>(q + RW(i0 + rr) * ld + h * D + c8 * 8);
        dd = *reinterpret_cast<const uint4*>(dout + RW(i0 + rr) * lddo + h * D + c8 * 8);
      }
      *reinterpret_cast<uint4*>(Qw + rr * VP + c8 * 8) = qq;
      *reinterpret_cast<uint4*>(dOw + rr * VP + c8 * 8) = dd;
    }
    bf16x8_t dof[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
      dof[ks] = *reinterpret_cast<const bf16x8_t*>(dout + RW(ic) * lddo + h * D + ks * 16 + 8 * hh);
    {
      f32x16_t acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(Evs + r * KP + ks * 16 + 8 * hh);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dof[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) doe[r * NW + e + 4 * hh] = acc[e];
      if (hh == 0) doe[r * NW + 8] = acc[4];
    }
    const float inv_sqrt = rsqrtf((float)D);
    const float* prow = P + (((size_t)b * H + h) * T + ic) * T;
    const uint32_t drow = (uint32_t)((b * H + h) * T + i);
    const bool vec = (T & 3) == 0;
    bf16_t* dst_base = dST + ((size_t)b * H + h) * T * TI;
    bf16_t* pdt_base = PdT + ((size_t)b * H + h) * T * TI;
    f32x16_t o[3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    }
    if constexpr (ONE_TILE) {
      // 161 <= T <= 256: holding all NT score tiles of a wave (NT*16 accumulators) next to P and the dropout masks does
      // not fit the register file.  One tile at a time instead: pass A walks the key tiles for Dsum only, pass B
      // RECOMPUTES each tile's dPd^T (6 MFMAs, operands already in LDS / registers), turns it into dS^T, stores it and
      // feeds dQ^T straight away — live state is one tile + the dQ^T accumulators, whatever NT is.
      auto dp_tile = [&](int t, f32x16_t& st) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
          bf16x8_t af;
          if (VG) {
            const uint4 z = make_uint4(0, 0, 0, 0);
            af = 32 * t + r < T ? *reinterpret_cast<const bf16x8_t*>(v + RW(32 * t + r) * ld + h * D + ks * 16 + 8 * hh)
                                : __builtin_bit_cast(bf16x8_t, z);
          } else {
            af = *reinterpret_cast<const bf16x8_t*>(Vs + (32 * t + r) * KP + ks * 16 + 8 * hh);
          }
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dof[ks], st, 0, 0, 0);
        }
      };
      auto load_p4 = [&](int j0, float* p4) {
        p4[0] = p4[1] = p4[2] = p4[3] = 0.f;
        if (vec && j0 + 3 < T) { const float4 pv = *reinterpret_cast<const float4*>(prow + j0); p4[0] = pv.x; p4[1] = pv.y; p4[2] = pv.z; p4[3] = pv.w; }
        else {
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) if (j0 + e2 < T) p4[e2] = prow[j0 + e2];
        }
      };
      auto dp_elem = [&](float dp, int j) {
        const int rel = j - i + WIN;
        if ((unsigned)rel <= 2u * WIN) dp += doe[r * NW + rel];
        if (drop_thresh) dp = drop_keep(drop_seed, drow, j, drop_thresh) ? dp * drop_scale : 0.f;
        return j >= T ? 0.f : dp;
      };
      __builtin_amdgcn_wave_barrier();
      float dsum = 0.f;
#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        if (32 * t >= T) break;                                      // wave-uniform
        f32x16_t st;
        dp_tile(t, st);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j0 = 32 * t + 8 * g + 4 * hh;
          float p4[4];
          load_p4(j0, p4);
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) dsum += dp_elem(st[4 * g + e2], j0 + e2) * p4[e2];
        }
      }
      dsum += __shfl_xor(dsum, 32);
#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        if (32 * t >= T) break;
        f32x16_t st;
        dp_tile(t, st);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j0 = 32 * t + 8 * g + 4 * hh;
          float p4[4];
          load_p4(j0, p4);
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const int j = j0 + e2;
            float ds = p4[e2] * (dp_elem(st[4 * g + e2], j) - dsum) * inv_sqrt;
            float pd = p4[e2];
            if (drop_thresh) pd = drop_keep(drop_seed, drow, j, drop_thresh) ? pd * drop_scale : 0.f;
            if (j >= T || j >= len || i >= len || i >= T) ds = 0.f;          // masked_fill blocks the gradient
            if (i >= len || i >= T) pd = 0.f;                                // padded queries carry no upstream gradient
            st[4 * g + e2] = ds;
            if (j < T) {
              dst_base[(size_t)j * TI + i] = f2bf(ds);
              pdt_base[(size_t)j * TI + i] = f2bf(pd);
              const int rel = j - i + WIN;
              if ((unsigned)rel <= 2u * WIN) { const bf16_t db = f2bf(ds); dSB[r * 16 + rel] = db; dSBT[rel * BTP + r] = db; PdBT[rel * BTP + r] = f2bf(pd); }
            }
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float f8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f8[e] = st[8 * s2 + e];
          const bf16x8_t pf = pack8(f8);
#pragma unroll
          for (int dt = 0; dt < 3; ++dt) {
            const bf16_t* ka = Ks + (32 * t + 16 * s2 + 4 * hh + qd) * VP + 32 * dt + colhalf + 4 * pp;
            const bf16x8_t af = tr_frag8(ka, ka + 8 * VP);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf, o[dt], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      f32x16_t s[NT];
  #pragma unroll
      for (int t = 0; t < NT; ++t) {
  #pragma unroll
        for (int e = 0; e < 16; ++e) s[t][e] = 0.f;
  #pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
          const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(Vs + (32 * t + r) * KP + ks * 16 + 8 * hh);
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dof[ks], s[t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();

      // pass A: dP in place, Dsum
      float dsum = 0.f;
  #pragma unroll
      for (int t = 0; t < NT; ++t)
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j0 = 32 * t + 8 * g + 4 * hh;
          float p4[4] = {0.f, 0.f, 0.f, 0.f};
          if (vec && j0 + 3 < T) { const float4 pv = *reinterpret_cast<const float4*>(prow + j0); p4[0] = pv.x; p4[1] = pv.y; p4[2] = pv.z; p4[3] = pv.w; }
          else {
  #pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) if (j0 + e2 < T) p4[e2] = prow[j0 + e2];
          }
  #pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const int j = j0 + e2;
            float dp = s[t][4 * g + e2];
            const int rel = j - i + WIN;
            if ((unsigned)rel <= 2u * WIN) dp += doe[r * NW + rel];
            if (drop_thresh) dp = drop_keep(drop_seed, drow, j, drop_thresh) ? dp * drop_scale : 0.f;
            if (j >= T) dp = 0.f;
            s[t][4 * g + e2] = dp;
            dsum += dp * p4[e2];
          }
        }
      dsum += __shfl_xor(dsum, 32);
      // pass B: dS in place; write dS^T, dropout(P)^T and the band tables
  #pragma unroll
      for (int t = 0; t < NT; ++t)
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j0 = 32 * t + 8 * g + 4 * hh;
          float p4[4] = {0.f, 0.f, 0.f, 0.f};
          if (vec && j0 + 3 < T) { const float4 pv = *reinterpret_cast<const float4*>(prow + j0); p4[0] = pv.x; p4[1] = pv.y; p4[2] = pv.z; p4[3] = pv.w; }
          else {
  #pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) if (j0 + e2 < T) p4[e2] = prow[j0 + e2];
          }
  #pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const int j = j0 + e2;
            float ds = p4[e2] * (s[t][4 * g + e2] - dsum) * inv_sqrt;
            float pd = p4[e2];
            if (drop_thresh) pd = drop_keep(drop_seed, drow, j, drop_thresh) ? pd * drop_scale : 0.f;
            if (j >= T || j >= len || i >= len || i >= T) ds = 0.f;          // masked_fill blocks the gradient
            if (i >= len || i >= T) pd = 0.f;                                // padded queries carry no upstream gradient
            s[t][4 * g + e2] = ds;
            if (j < T) {
              dst_base[(size_t)j * TI + i] = f2bf(ds);
              pdt_base[(size_t)j * TI + i] = f2bf(pd);
              const int rel = j - i + WIN;
              if ((unsigned)rel <= 2u * WIN) { const bf16_t db = f2bf(ds); dSB[r * 16 + rel] = db; dSBT[rel * BTP + r] = db; PdBT[rel * BTP + r] = f2bf(pd); }
            }
          }
        }
      __builtin_amdgcn_wave_barrier();

      // dQ^T = K^T dS^T + Ek^T band(dS)^T
  #pragma unroll
      for (int t = 0; t < NT; ++t)
  #pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float f8[8];
  #pragma unroll
          for (int e = 0; e < 8; ++e) f8[e] = s[t][8 * s2 + e];
          const bf16x8_t pf = pack8(f8);
  #pragma unroll
          for (int dt = 0; dt < 3; ++dt) {
            const bf16_t* ka = Ks + (32 * t + 16 * s2 + 4 * hh + qd) * VP + 32 * dt + colhalf + 4 * pp;
            const bf16x8_t af = tr_frag8(ka, ka + 8 * VP);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf, o[dt], 0, 0, 0);
          }
        }
    }
    {
      const bf16x8_t bfp = *reinterpret_cast<const bf16x8_t*>(dSB + r * 16 + 8 * hh);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(EkT + (32 * dt + r) * 16 + 8 * hh);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfp, o[dt], 0, 0, 0);
      }
    }
    if (i < T && i <= nv1) {
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * dt + 8 * g + 4 * hh;
          *reinterpret_cast<uint2*>(dq + (rbase + i) * lddq + h * D + d) =
              make_uint2(pack2bf(o[dt][4 * g], o[dt][4 * g + 1]), pack2bf(o[dt][4 * g + 2], o[dt][4 * g + 3]));
        }
    }
    // dEk[r'][d] += sum_i dSBT[r'][i] Q[i][d];  dEv[r'][d] += sum_i PdBT[r'][i] dO[i][d]   (K = 32 queries)
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const bf16_t* At = which ? PdBT : dSBT;
      const bf16_t* Bt = which ? dOw : Qw;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        f32x16_t acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(At + (r & 15) * BTP + 16 * ks + 8 * hh);
          if (r >= 16) { const uint4 z = make_uint4(0, 0, 0, 0); af = __builtin_bit_cast(bf16x8_t, z); }
          const bf16_t* ba = Bt + (16 * ks + 8 * hh + qd) * VP + 32 * dt + colhalf + 4 * pp;
          const bf16x8_t bf_ = tr_frag8(ba, ba + 4 * VP);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf_, acc, 0, 0, 0);
        }
        float* dstA = Acc + which * NW * D;
        const int d = 32 * dt + r;                                   // D layout: column = d (lane&31), row r' = (e&3)+8(e>>2)+4hh
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dstA + (e + 4 * hh) * D + d, acc[e]);
        if (hh == 0) atomicAdd(dstA + 8 * D + d, acc[4]);
      }
    }
  }
  __syncthreads();
  for (int x = tid; x < NW * D; x += NTH) {
    if (Acc[x] != 0.f) atomicAdd(dEk + x, Acc[x]);
    if (Acc[NW * D + x] != 0.f) atomicAdd(dEv + x, Acc[NW * D + x]);
  }
}

template <int NT>
__global__ __launch_bounds__(256, 1) void gt_attn_bwd_kv_mfma_kernel(
    const bf16_t* __restrict__ q, int ld, const bf16_t* __restrict__ dout, int lddo,
    const bf16_t* __restrict__ dST, const bf16_t* __restrict__ PdT, int TI,
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int lddk, int T, int Tp, const int32_t* row0, int H)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TPAD = NT * 32;
  bf16_t* Qs  = reinterpret_cast<bf16_t*>(smem);                 // [TPAD][VP]
  bf16_t* dOs = Qs + TPAD * VP;                                  // [TPAD][VP]
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const size_t rbase = (size_t)gt_row_base(row0, b, Tp) + HALO;
  // rows this utterance owns behind rbase (frames + trailing halo): in the ragged layout frame indices past them belong
  // to the NEXT utterance — reads are clamped onto the (zero) trailing halo row, stores are dropped
  const int nv1 = gt_row_count(row0, b, Tp) - HALO - 1;
  auto RW = [&](int t) { return rbase + (size_t)(t < nv1 ? t : nv1); };
  for (int i = tid; i < TPAD * (D / 8); i += 256) {
    const int j = i / (D / 8), c8 = i - j * (D / 8);
    uint4 qq = make_uint4(0, 0, 0, 0), dd = qq;
    if (j < T) {
      qq = *reinterpret_cast<const uint4*>(q + RW(j) * ld + h * D + c8 * 8);
      dd = *reinterpret_cast<const uint4*>(dout + RW(j) * lddo + h * D + c8 * 8);
    }
    *reinterpret_cast<uint4*>(Qs + j * VP + c8 * 8) = qq;
    *reinterpret_cast<uint4*>(dOs + j * VP + c8 * 8) = dd;
  }
  __syncthreads();
  const int j0 = blockIdx.x * 128 + 32 * w;
  if (j0 >= T) return;
  const int j = j0 + r, jc = j < T ? j : T - 1;
  const int li = lane & 15, qd = li >> 2, pp = li & 3, colhalf = ((lane >> 4) & 1) * 16;
  const bf16_t* dsr = dST + (((size_t)b * H + h) * T + jc) * TI;
  const bf16_t* pdr = PdT + (((size_t)b * H + h) * T + jc) * TI;
  f32x16_t ak[3], av[3];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) { ak[dt][e] = 0.f; av[dt][e] = 0.f; }
  }
  const int nks = TI / 16;
  for (int ks = 0; ks < nks; ++ks) {
    const bf16x8_t bds = *reinterpret_cast<const bf16x8_t*>(dsr + 16 * ks + 8 * hh);     // dS[i = 16ks+8hh.., j]
    const bf16x8_t bpd = *reinterpret_cast<const bf16x8_t*>(pdr + 16 * ks + 8 * hh);
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const bf16_t* qa = Qs + (16 * ks + 8 * hh + qd) * VP + 32 * dt + colhalf + 4 * pp;
      const bf16_t* da = dOs + (16 * ks + 8 * hh + qd) * VP + 32 * dt + colhalf + 4 * pp;
      ak[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag8(qa, qa + 4 * VP), bds, ak[dt], 0, 0, 0);
      av[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag8(da, da + 4 * VP), bpd, av[dt], 0, 0, 0);
    }
  }
  if (j < T && j <= nv1) {
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * dt + 8 * g + 4 * hh;
        *reinterpret_cast<uint2*>(dk + RW(j) * lddk + h * D + d) =
            make_uint2(pack2bf(ak[dt][4 * g], ak[dt][4 * g + 1]), pack2bf(ak[dt][4 * g + 2], ak[dt][4 * g + 3]));
        *reinterpret_cast<uint2*>(dv + RW(j) * lddk + h * D + d) =
            make_uint2(pack2bf(av[dt][4 * g], av[dt][4 * g + 1]), pack2bf(av[dt][4 * g + 2], av[dt][4 * g + 3]));
      }
  }
}

template <int NT, int WV, bool ONE_TILE = (NT > 5)>
int launch_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ld, const float* Ek, const float* Ev, const int32_t* lens,
               const bf16_t* dout, int lddo, const float* P, bf16_t* ws, bf16_t* dq, bf16_t* dk, bf16_t* dv, int lddq,
               float* dEk, float* dEv, int B, int T, int Tp, const int32_t* row0, int H, uint32_t th, uint32_t sd, float sc, const uint32_t* seed_dev, hipStream_t st)
{
  constexpr int TPAD = NT * 32;
  constexpr int WBN = 32 * 16 + 2 * 16 * BTP + 2 * 32 * VP;
  const int TI = ((T + 31) / 32) * 32;
  bf16_t* dST = ws;
  bf16_t* PdT = ws + (size_t)B * H * T * TI;
  const size_t lds1 = (NT > 8 ? 0 : (size_t)TPAD * KP * 2) + (size_t)TPAD * VP * 2 + 32 * KP * 2 + D * 16 * 2 + 2 * NW * D * 4 +
                      (size_t)WV * 32 * NW * 4 + (size_t)WV * WBN * 2;
  const size_t lds2 = (size_t)2 * TPAD * VP * 2;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_attn_bwd_q_mfma_kernel<NT, WV, ONE_TILE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GT_E_LAUNCH;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_attn_bwd_kv_mfma_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GT_E_LAUNCH;
    attr = true;
  }
  if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return 1;
  hipLaunchKernelGGL((gt_attn_bwd_q_mfma_kernel<NT, WV, ONE_TILE>), dim3((T + 32 * WV - 1) / (32 * WV), H, B), dim3(64 * WV), lds1, st,
                     q, k, v, ld, Ek, Ev, lens, dout, lddo, P, dST, PdT, TI, dq, lddq, dEk, dEv, T, Tp, row0, H, th, sd, sc, seed_dev);
  hipLaunchKernelGGL(gt_attn_bwd_kv_mfma_kernel<NT>, dim3((T + 127) / 128, H, B), dim3(256), lds2, st,
                     q, ld, dout, lddo, dST, PdT, TI, dk, dv, lddq, T, Tp, row0, H);
  return gt_launch_status(__func__);
}

}  // namespace

size_t gt_attn_bwd_mfma_ws_bytes(int B, int T, int H)
{
  const size_t TI = (size_t)((T + 31) / 32) * 32;
  return (size_t)2 * B * H * T * TI * 2;
}

int gt_attn_bwd_mfma_impl(const void* q, const void* k, const void* v, int ld, const float* Ek, const float* Ev,
                          const int32_t* lens, const void* dout, int lddo, const float* P, void* ws, size_t ws_bytes,
                          void* dq, void* dk, void* dv, int lddq, float* dEk, float* dEv,
                          int B, int T, int Tp, const int32_t* row0, int H, int Dh, int win, uint32_t th, uint32_t sd, float sc, const uint32_t* seed_dev, void* stream)
{
  if (Dh != D || win != WIN || T > 384 || (ld & 7) || (lddo & 7) || (lddq & 3)) return 1;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)ws) & 15) return 1;
  if (ws_bytes < gt_attn_bwd_mfma_ws_bytes(B, T, H)) return 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bf16_t* qq = static_cast<const bf16_t*>(q); const bf16_t* kk = static_cast<const bf16_t*>(k); const bf16_t* vv = static_cast<const bf16_t*>(v);
  const bf16_t* dd = static_cast<const bf16_t*>(dout);
  bf16_t* w16 = static_cast<bf16_t*>(ws);
  bf16_t* dqq = static_cast<bf16_t*>(dq); bf16_t* dkk = static_cast<bf16_t*>(dk); bf16_t* dvv = static_cast<bf16_t*>(dv);
  // (T <= 160 through the one-tile-at-a-time form of the longer sequences — <5, 4, true>: 138 registers instead of 362 — measured 66.6 vs
  //  71.0 us back to back and 13 us of a 4.4 ms step: within the noise, not taken; 5 waves per workgroup, one workgroup per head: 460 us)
  if (T <= 160) return launch_bwd<5, 4>(qq, kk, vv, ld, Ek, Ev, lens, dd, lddo, P, w16, dqq, dkk, dvv, lddq, dEk, dEv, B, T, Tp, row0, H, th, sd, sc, seed_dev, st);
  // 161 <= T <= 256: 8 key tiles, 2 waves per workgroup (153 KB of LDS), one score tile live at a time (recompute form:
  // the first version kept all 8 tiles in registers, needed scratch and faulted — see DESIGN.md §4.5)
  if (T <= 256) return launch_bwd<8, 2>(qq, kk, vv, ld, Ek, Ev, lens, dd, lddo, P, w16, dqq, dkk, dvv, lddq, dEk, dEv, B, T, Tp, row0, H, th, sd, sc, seed_dev, st);
  // 257 <= T <= 384 (cfg3): 12 key tiles; K^T (transposing reads) stays in LDS, the V rows come from L2
  return launch_bwd<12, 4>(qq, kk, vv, ld, Ek, Ev, lens, dd, lddo, P, w16, dqq, dkk, dvv, lddq, dEk, dEv, B, T, Tp, row0, H, th, sd, sc, seed_dev, st);
}

// returns 1 if the shape is not handled here (caller falls back to the generic kernel)
int gt_attn_fwd_mfma_impl(const void* q, const void* k, const void* v, int ld, const float* Ek, const float* Ev,
                          const int32_t* lens, void* out, int ldo, float* P, int B, int T, int Tp, const int32_t* row0, int H, int Dh, int win,
                          uint32_t th, uint32_t sd, float sc, const uint32_t* seed_dev, void* stream)
{
  if (Dh != D || win != WIN || T > 384 || (ld & 7) || (ldo & 3)) return 1;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bf16_t* qq = static_cast<const bf16_t*>(q); const bf16_t* kk = static_cast<const bf16_t*>(k); const bf16_t* vv = static_cast<const bf16_t*>(v);
  bf16_t* oo = static_cast<bf16_t*>(out);
  if (T <= 160) return launch_fwd<5>(qq, kk, vv, ld, Ek, Ev, lens, oo, ldo, P, B, T, Tp, row0, H, th, sd, sc, seed_dev, st);
  if (T <= 256) return launch_fwd<8>(qq, kk, vv, ld, Ek, Ev, lens, oo, ldo, P, B, T, Tp, row0, H, th, sd, sc, seed_dev, st);
  return launch_fwd_long<12>(qq, kk, vv, ld, Ek, Ev, lens, oo, ldo, P, B, T, Tp, row0, H, th, sd, sc, seed_dev, st);
}
