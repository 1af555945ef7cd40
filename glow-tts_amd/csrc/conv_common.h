// Shared by the implicit-GEMM conv kernels: launch arguments and the row-chunk epilogues.
#pragma once
#include "common.h"

namespace gtconv {

struct ConvArgs {
  const bf16_t* X; int ldx;                   // [R, >=Cin]
  const bf16_t* W;                            // packed [taps][Np][Kp]
  const float* bias;                          // [N] (gate: [2*half]) or null
  const float* cond; int ldc;                 // [B, ldc] per utterance (B > 0), [R, ldc] per row (B == 0), or null
  const float* rowmask;                       // [R] or null
  void* Y; int ldy;
  const void* addend; int ldadd;              // same dtype as Y, or null
  bf16_t* Tout; bf16_t* Sout; int ldts;       // gate: saved tanh / sigmoid halves
  int R, N, Cin, taps, Tp, Np, Kp;
  const int32_t* row0; int B;                 // ragged rows layout (NULL: uniform Tp rows per utterance): cond row = batch of m
  int out_f32, relu;
  uint32_t drop_thresh, drop_seed; float drop_scale;   // gate dropout (modules.py:153)
  uint32_t gb_thresh;                                  // gatebwd: dropout threshold replayed on the gradient
  const uint32_t* seed_dev;                            // optional device word XOR-ed into drop_seed (graph replay)
  int exp_;                                            // EXPERIMENT bits (dev only)
  int y16;                                             // Y rows allow 16-byte bf16 stores (ldy % 8 == 0, base 16-B aligned)
  int maskbwd;                                         // epilogue = ReLU / dropout backward: Tout is the SAVED activation, Y = acc * scale where it is non-zero
  int gatebwd;                                         // epilogue = WaveNet-gate backward: Tout/Sout are the SAVED tanh/sigmoid, Y = d pre [R, 2N]
};

// Phase 2 of the epilogue (phase 1 = every wave drops its fp32 accumulators into the LDS tile `es`,
// [BM rows][BN packed channels], pitch EP floats): a thread owns (row, 8 consecutive channels) chunks, so
// bias / cond / addend loads and all stores are 16-32 B per lane and whole 128-B lines per row — the MFMA
// layout itself gives only 8 B per lane with a row stride between lanes.

// WaveNet gate (commons.py:61-68): BN packed columns = BN/2 gate channels, [32 tanh | 32 sigmoid] per 64.
template <int NT, int BM, int BN>
__device__ __forceinline__ void epilogue_gate(const ConvArgs& a, const float* es, int EP, int m0, int n0, int tid)
{
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int CPR = BN / 16;                                     // 8-channel chunks per row
  constexpr int NCH = (BM * CPR + NT - 1) / NT;
  const int half = a.N >> 1;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int q = tid + NT * j, row = q / CPR, c = (q % CPR) * 8;   // gate channel within the tile
    const int m = m0 + row;
    if (row >= BM || m >= a.R) continue;
    const int cg = (n0 >> 1) + c;                                  // global gate channel
    const float* et = &es[row * EP + (c >> 5) * 64 + (c & 31)];
    float4 t0 = *reinterpret_cast<const float4*>(et), t1 = *reinterpret_cast<const float4*>(et + 4);
    float4 s0 = *reinterpret_cast<const float4*>(et + 32), s1 = *reinterpret_cast<const float4*>(et + 36);
    float4 bt0 = z4, bt1 = z4, bs0 = z4, bs1 = z4, ct0 = z4, ct1 = z4, cs0 = z4, cs1 = z4;
    if (a.bias) {
      bt0 = *reinterpret_cast<const float4*>(a.bias + cg);        bt1 = *reinterpret_cast<const float4*>(a.bias + cg + 4);
      bs0 = *reinterpret_cast<const float4*>(a.bias + half + cg); bs1 = *reinterpret_cast<const float4*>(a.bias + half + cg + 4);
    }
    if (a.cond) {
      const float* cp = a.cond + (size_t)(a.B > 0 ? gt_row_batch(a.row0, a.B, m, a.Tp) : m) * a.ldc + cg;   // B == 0: per-row cond
      ct0 = *reinterpret_cast<const float4*>(cp);        ct1 = *reinterpret_cast<const float4*>(cp + 4);
      cs0 = *reinterpret_cast<const float4*>(cp + half); cs1 = *reinterpret_cast<const float4*>(cp + half + 4);
    }
    const float pt_[8] = {t0.x + bt0.x, t0.y + bt0.y, t0.z + bt0.z, t0.w + bt0.w, t1.x + bt1.x, t1.y + bt1.y, t1.z + bt1.z, t1.w + bt1.w};
    const float ps_[8] = {s0.x + bs0.x, s0.y + bs0.y, s0.z + bs0.z, s0.w + bs0.w, s1.x + bs1.x, s1.y + bs1.y, s1.z + bs1.z, s1.w + bs1.w};
    const float ct_[8] = {ct0.x, ct0.y, ct0.z, ct0.w, ct1.x, ct1.y, ct1.z, ct1.w};
    const float cs_[8] = {cs0.x, cs0.y, cs0.z, cs0.w, cs1.x, cs1.y, cs1.z, cs1.w};
    float tt[8], ss[8], aa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float pt = pt_[i], ps = ps_[i];
      if (a.drop_thresh) {                                         // x_in = drop(conv(x))
        bool kt, ks;
        drop_keep_gate(a.drop_seed, m, cg + i, drop_thresh16(a.drop_thresh), kt, ks);
        pt = kt ? pt * a.drop_scale : 0.0f;
        ps = ks ? ps * a.drop_scale : 0.0f;
      }
      pt += ct_[i]; ps += cs_[i];
      tt[i] = tanhf_(pt); ss[i] = sigmoidf_(ps); aa[i] = tt[i] * ss[i];
    }
    *reinterpret_cast<uint4*>(a.Tout + (size_t)m * a.ldts + cg) =
        make_uint4(pack2bf(tt[0], tt[1]), pack2bf(tt[2], tt[3]), pack2bf(tt[4], tt[5]), pack2bf(tt[6], tt[7]));
    *reinterpret_cast<uint4*>(a.Sout + (size_t)m * a.ldts + cg) =
        make_uint4(pack2bf(ss[0], ss[1]), pack2bf(ss[2], ss[3]), pack2bf(ss[4], ss[5]), pack2bf(ss[6], ss[7]));
    *reinterpret_cast<uint4*>(static_cast<bf16_t*>(a.Y) + (size_t)m * a.ldy + cg) =
        make_uint4(pack2bf(aa[0], aa[1]), pack2bf(aa[2], aa[3]), pack2bf(aa[4], aa[5]), pack2bf(aa[6], aa[7]));
  }
}

// plain / relu / dropout / addend / mask / gate-backward epilogue
template <int NT, int BM, int BN>
__device__ __forceinline__ void epilogue_plain(const ConvArgs& a, const float* es, int EP, int m0, int n0, int tid)
{
  constexpr int NCH = (BM * (BN / 8) + NT - 1) / NT;               // chunks per thread
  constexpr int CPR = BN / 8;                                      // chunks per row
  // side loads of all chunks first (one exposed latency), then the math and the stores
  uint4 adq[NCH][2], tsq[NCH][2];
  float rmq[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int q = tid + NT * j, row = q / CPR, c = (q % CPR) * 8;
    const int m = m0 + row, n = n0 + c;
    adq[j][0] = make_uint4(0, 0, 0, 0); adq[j][1] = make_uint4(0, 0, 0, 0);
    tsq[j][0] = make_uint4(0, 0, 0, 0); tsq[j][1] = make_uint4(0, 0, 0, 0);
    rmq[j] = 1.0f;
    if (row < BM && m < a.R && n < a.N) {
      if (a.rowmask) rmq[j] = a.rowmask[m];
      if (a.gatebwd) {
        tsq[j][0] = *reinterpret_cast<const uint4*>(a.Tout + (size_t)m * a.ldts + n);
        tsq[j][1] = *reinterpret_cast<const uint4*>(a.Sout + (size_t)m * a.ldts + n);
      }
      if (a.maskbwd) tsq[j][0] = *reinterpret_cast<const uint4*>(a.Tout + (size_t)m * a.ldts + n);
      if (a.addend) {
        if (a.out_f32) {
          const float* ap = static_cast<const float*>(a.addend) + (size_t)m * a.ldadd + n;
          adq[j][0] = *reinterpret_cast<const uint4*>(ap);
          if (n + 4 < a.N) adq[j][1] = *reinterpret_cast<const uint4*>(ap + 4);
        } else {
          const bf16_t* ap = static_cast<const bf16_t*>(a.addend) + (size_t)m * a.ldadd + n;
          const uint2 lo = *reinterpret_cast<const uint2*>(ap);
          uint2 hi = make_uint2(0, 0);
          if (n + 4 < a.N) hi = *reinterpret_cast<const uint2*>(ap + 4);
          adq[j][0] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int q = tid + NT * j, row = q / CPR, c = (q % CPR) * 8;
    const int m = m0 + row, n = n0 + c;
    if (row >= BM || m >= a.R || n >= a.N) continue;                            // N % 4 == 0
    const bool full = n + 4 < a.N;                                 // second 4-channel half valid
    const float* ep = &es[row * EP + c];
    const float4 e0 = *reinterpret_cast<const float4*>(ep), e1 = *reinterpret_cast<const float4*>(ep + 4);
    float v[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    const int n1 = full ? n + 4 : n;                               // clamped address of the second half
    if (a.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(a.bias + n), b1 = *reinterpret_cast<const float4*>(a.bias + n1);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (a.cond) {
      const float* cp = a.cond + (size_t)(a.B > 0 ? gt_row_batch(a.row0, a.B, m, a.Tp) : m) * a.ldc;
      const float4 c0 = *reinterpret_cast<const float4*>(cp + n), c1 = *reinterpret_cast<const float4*>(cp + n1);
      v[0] += c0.x; v[1] += c0.y; v[2] += c0.z; v[3] += c0.w; v[4] += c1.x; v[5] += c1.y; v[6] += c1.z; v[7] += c1.w;
    }
    if (a.relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (a.drop_thresh) {                                           // dropout after the activation (attentions.py:370)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = drop_keep(a.drop_seed, m, n + i, a.drop_thresh) ? v[i] * a.drop_scale : 0.0f;
    }
    float ad[8];
    if (a.out_f32) {
      ad[0] = __uint_as_float(adq[j][0].x); ad[1] = __uint_as_float(adq[j][0].y); ad[2] = __uint_as_float(adq[j][0].z); ad[3] = __uint_as_float(adq[j][0].w);
      ad[4] = __uint_as_float(adq[j][1].x); ad[5] = __uint_as_float(adq[j][1].y); ad[6] = __uint_as_float(adq[j][1].z); ad[7] = __uint_as_float(adq[j][1].w);
    } else {
      const uint4 u = adq[j][0];
      ad[0] = bf2f(u.x & 0xffff); ad[1] = bf2f(u.x >> 16); ad[2] = bf2f(u.y & 0xffff); ad[3] = bf2f(u.y >> 16);
      ad[4] = bf2f(u.z & 0xffff); ad[5] = bf2f(u.z >> 16); ad[6] = bf2f(u.w & 0xffff); ad[7] = bf2f(u.w >> 16);
    }
    const float rm = rmq[j];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (v[i] + ad[i]) * rm;
    if (a.maskbwd) {
      // backward of y = dropout(relu(.)) (attentions.py:368-370, modules.py:97-99) from the saved y: kept and positive <=> y != 0
      const uint32_t yw[4] = {tsq[j][0].x, tsq[j][0].y, tsq[j][0].z, tsq[j][0].w};
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = ((yw[i >> 1] >> (16 * (i & 1))) & 0x7fff) ? v[i] * a.drop_scale : 0.0f;
    }
    if (a.gatebwd) {
      // v = d acts.  d pre_t = d*S*(1-T^2), d pre_s = d*T*S*(1-S), times the replayed dropout mask of the conv
      // output (modules.py:153-156 backward); natural [tanh half | sigmoid half] order, N % 8 == 0.
      const uint32_t tw[4] = {tsq[j][0].x, tsq[j][0].y, tsq[j][0].z, tsq[j][0].w}, sw[4] = {tsq[j][1].x, tsq[j][1].y, tsq[j][1].z, tsq[j][1].w};
      float gt[8], gs[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float t = bf2f((tw[i >> 1] >> (16 * (i & 1))) & 0xffff), sg = bf2f((sw[i >> 1] >> (16 * (i & 1))) & 0xffff);
        gt[i] = v[i] * sg * (1.0f - t * t); gs[i] = v[i] * t * sg * (1.0f - sg);
        if (a.gb_thresh) {
          bool kt, ks;
          drop_keep_gate(a.drop_seed, m, n + i, drop_thresh16(a.gb_thresh), kt, ks);
          gt[i] = kt ? gt[i] * a.drop_scale : 0.0f;
          gs[i] = ks ? gs[i] * a.drop_scale : 0.0f;
        }
      }
      bf16_t* yp = static_cast<bf16_t*>(a.Y) + (size_t)m * a.ldy + n;
      *reinterpret_cast<uint4*>(yp) = make_uint4(pack2bf(gt[0], gt[1]), pack2bf(gt[2], gt[3]), pack2bf(gt[4], gt[5]), pack2bf(gt[6], gt[7]));
      *reinterpret_cast<uint4*>(yp + a.N) = make_uint4(pack2bf(gs[0], gs[1]), pack2bf(gs[2], gs[3]), pack2bf(gs[4], gs[5]), pack2bf(gs[6], gs[7]));
    } else if (a.out_f32) {
      float* yp = static_cast<float*>(a.Y) + (size_t)m * a.ldy + n;
      *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
      if (full) *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      bf16_t* yp = static_cast<bf16_t*>(a.Y) + (size_t)m * a.ldy + n;
      const uint2 lo = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3])), hi = make_uint2(pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
      if (full && a.y16) *reinterpret_cast<uint4*>(yp) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      else { *reinterpret_cast<uint2*>(yp) = lo; if (full) *reinterpret_cast<uint2*>(yp + 4) = hi; }
    }
  }
}

}  // namespace gtconv
