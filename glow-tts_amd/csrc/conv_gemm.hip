// Implicit-GEMM 1-D convolution on bf16 MFMA (v_mfma_f32_32x32x16_bf16) for gfx950.
//
// Replaces the ATen/cuDNN conv1d calls of the reference hot path:
//   modules.py:152 (WN in_layer k=5), :165 (res/skip 1x1), attentions.py:144,172 (start / end 1x1),
//   attentions.py:232-238 (q/k/v/o 1x1), :365-371 (FFN k=3), modules.py:97 (prenet k=5),
//   models.py:710 (proj_m), and — with transposed/flipped packed weights — their data gradients.
//
// Data layout ("rows"): activations are [R, C] channels-last, R = B * Tp rows where every
// utterance owns Tp = T + 2*HALO consecutive rows and its first/last HALO rows (and every row
// past its length) are zero.  A k-tap convolution is then k shifted row-block GEMMs with no
// boundary logic:  Y[m, n] = sum_tap sum_ci X[m + tap - k/2, ci] * W[tap][n][ci].
//
// Mapping: MFMA A = weight tile (rows = output channels n), MFMA B = activation tile
// (columns = rows m), so a lane's 16 accumulators are 4 groups of 4 CONSECUTIVE channels of ONE
// row m -> 8/16-byte channels-last stores, per-channel bias as float4, and the WaveNet gate
// tanh(.)*sigmoid(.) (commons.py:61-68) pairs two accumulator blocks of the same lane when the
// packed weight rows are interleaved [32 tanh | 32 sigmoid] (see gt_pack_conv_weights).
//
// Tile: 128 rows x {128|64} channels per 256-thread workgroup, K slice 64, LDS pitch 144 B
// (conflict-free ds_read_b128 for 16 distinct rows), weights and activations double-buffered
// in LDS with register prefetch (global loads of step i+1 fly under the MFMAs of step i), one
// barrier per (K-slice, tap) step, 2 workgroups per CU.
#include <stdlib.h>
#include "common.h"
#include "../../include/glowtts_hip.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int LDP = 72;                       // halfs per LDS row (64 + 8 pad)
constexpr int MAXTAPS = 5;
constexpr int XROWS = BM + MAXTAPS - 1;

struct ConvArgs {
  const bf16_t* X; int ldx;                   // [R, >=Cin]
  const bf16_t* W;                            // packed [taps][Np][Kp]
  const float* bias;                          // [N] (gate: [2*half]) or null
  const float* cond; int ldc;                 // [B, ldc] or null
  const float* rowmask;                       // [R] or null
  void* Y; int ldy;
  const void* addend; int ldadd;              // same dtype as Y, or null
  bf16_t* Tout; bf16_t* Sout; int ldts;       // gate: saved tanh / sigmoid halves
  int R, N, Cin, taps, Tp, Np, Kp;
  int out_f32, relu;
  uint32_t drop_thresh, drop_seed; float drop_scale;   // gate dropout (modules.py:153)
  uint32_t gb_thresh;                                  // gatebwd: dropout threshold replayed on the gradient
  const uint32_t* seed_dev;                            // optional device word XOR-ed into drop_seed (graph replay)
  int exp_;                                            // EXPERIMENT bits (dev only)
  int y16;                                             // Y rows allow 16-byte bf16 stores (ldy % 8 == 0, base 16-B aligned)
  int gatebwd;                                         // epilogue = WaveNet-gate backward: Tout/Sout are the SAVED tanh/sigmoid, Y = d pre [R, 2N]
};

template <int BN, bool GATE>
__global__ __launch_bounds__(256, 2) void gt_conv_gemm_kernel(ConvArgs a)
{
  constexpr int WN = BN / 64;                 // waves along channels
  constexpr int WM = 4 / WN;                  // waves along rows
  constexpr int MB = BM / (32 * WM);          // 32-row MFMA blocks per wave
  constexpr int WCH = BN / 32;                // weight 16-B chunks per thread per tile

  constexpr int XS_HALFS = XROWS * LDP, WS_HALFS = BN * LDP;
  constexpr int EP = BN + 4;                  // epilogue tile pitch in floats (== 4 mod 64 banks: conflict-free b128 rows)
  constexpr int MAIN_BYTES = 2 * (XS_HALFS + WS_HALFS) * 2, EPI_BYTES = BM * EP * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
  bf16_t (*Xs)[XS_HALFS] = reinterpret_cast<bf16_t (*)[XS_HALFS]>(smem);
  bf16_t (*Ws)[WS_HALFS] = reinterpret_cast<bf16_t (*)[WS_HALFS]>(smem + 2 * XS_HALFS * 2);

  if (a.seed_dev) a.drop_seed ^= *a.seed_dev;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int taps = a.taps, padl = taps >> 1;
  const int NS = a.Kp / BK, NIT = NS * taps;
  const int xrows = BM + taps - 1;

  // staging registers as named scalars (arrays that live across the conditional prefetch end up
  // in scratch memory)
  uint4 w0 = {}, w1 = {}, w2 = {}, w3 = {}, x0 = {}, x1 = {}, x2 = {}, x3 = {}, x4 = {};
  auto ldw1 = [&](int it, int i) -> uint4 {
    const int slice = it / taps, tap = it - slice * taps;
    const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
    return *reinterpret_cast<const uint4*>(a.W + ((size_t)(tap * a.Np + n0 + row) * a.Kp + slice * BK + c8 * 8));
  };
  auto ldx1 = [&](int slice, int i) -> uint4 {
    const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
    int gm = m0 - padl + (row < xrows ? row : xrows - 1);
    gm = gm < 0 ? 0 : (gm >= a.R ? a.R - 1 : gm);
    const int ch = slice * BK + c8 * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ch < a.Cin) v = *reinterpret_cast<const uint4*>(a.X + (size_t)gm * a.ldx + ch);
    return v;
  };
  auto stw1 = [&](int buf, int i, const uint4& v) {
    const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
    *reinterpret_cast<uint4*>(&Ws[buf][row * LDP + c8 * 8]) = v;
  };
  auto stx1 = [&](int buf, int i, const uint4& v) {
    const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
    if (row < xrows) *reinterpret_cast<uint4*>(&Xs[buf][row * LDP + c8 * 8]) = v;
  };
#define load_W(it)  do { w0 = ldw1(it, 0); w1 = ldw1(it, 1); if (WCH > 2) { w2 = ldw1(it, 2); w3 = ldw1(it, 3); } } while (0)
#define store_W(bf) do { stw1(bf, 0, w0); stw1(bf, 1, w1); if (WCH > 2) { stw1(bf, 2, w2); stw1(bf, 3, w3); } } while (0)
#define load_X(sl)  do { x0 = ldx1(sl, 0); x1 = ldx1(sl, 1); x2 = ldx1(sl, 2); x3 = ldx1(sl, 3); x4 = ldx1(sl, 4); } while (0)
#define store_X(bf) do { stx1(bf, 0, x0); stx1(bf, 1, x1); stx1(bf, 2, x2); stx1(bf, 3, x3); stx1(bf, 4, x4); } while (0)

  f32x16_t acc[2][MB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  load_W(0); load_X(0); store_W(0); store_X(0);
  __syncthreads();

  const int mrow0 = wm * (32 * MB);
  for (int it = 0; it < NIT; ++it) {
    const int slice = it / taps, tap = it - slice * taps;
    const int nxt = it + 1;
    const bool has = nxt < NIT;
    const bool newslice = has && (tap == taps - 1);
    if (has && !(a.exp_ & 2)) { load_W(nxt); if (newslice) load_X(slice + 1); }

    const bf16_t* wsb = &Ws[it & 1][(64 * wn + r) * LDP + 8 * h];
    const bf16_t* xsb = &Xs[slice & 1][(mrow0 + r + tap) * LDP + 8 * h];
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t af[2], bfm[MB];
#pragma unroll
      for (int bn = 0; bn < 2; ++bn) af[bn] = *reinterpret_cast<const bf16x8_t*>(wsb + bn * 32 * LDP + ks * 16);
#pragma unroll
      for (int bm = 0; bm < MB; ++bm) bfm[bm] = *reinterpret_cast<const bf16x8_t*>(xsb + bm * 32 * LDP + ks * 16);
#pragma unroll
      for (int bn = 0; bn < 2; ++bn)
#pragma unroll
        for (int bm = 0; bm < MB; ++bm)
          acc[bn][bm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[bn], bfm[bm], acc[bn][bm], 0, 0, 0);
    }

    if (has && !(a.exp_ & 2)) { store_W(nxt & 1); if (newslice) store_X((slice + 1) & 1); }
    __syncthreads();
  }

#undef load_W
#undef store_W
#undef load_X
#undef store_X
  // ------------------------------------------------------------------ epilogue
  // Phase 1: every wave drops its fp32 accumulators into an LDS tile [128 rows][BN channels] (the main loop's
  // last barrier has retired all reads of Xs/Ws).  Phase 2: a thread owns (row, 8 consecutive channels) chunks,
  // so bias / cond / addend loads and all stores are 16-32 B per lane and whole 128-B lines per row — the MFMA
  // layout itself gives only 8 B per lane with a row stride between lanes.
  float* es = reinterpret_cast<float*>(smem);
  if (a.exp_ & 1) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc += acc[i][j][e];
    if (sacc == 123.456f) static_cast<bf16_t*>(a.Y)[tid] = 1;
    return;
  }
#pragma unroll
  for (int bm = 0; bm < MB; ++bm)
#pragma unroll
    for (int bn = 0; bn < 2; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(&es[(mrow0 + 32 * bm + r) * EP + 64 * wn + 32 * bn + 8 * g + 4 * h]) =
            make_float4(acc[bn][bm][4 * g], acc[bn][bm][4 * g + 1], acc[bn][bm][4 * g + 2], acc[bn][bm][4 * g + 3]);
  __syncthreads();

  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (GATE) {
    // 64 gate channels per tile; packed columns: [32 tanh | 32 sigmoid] per 64
    constexpr int NCH = (BM * (BN / 2) / 8) / 256;                   // chunks per thread (4)
    const int half = a.N >> 1;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int q = tid + 256 * j, row = q >> 3, c = (q & 7) * 8;     // gate channel within the tile
      const int m = m0 + row;
      if (m >= a.R) continue;
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
        const float* cp = a.cond + (size_t)(m / a.Tp) * a.ldc + cg;
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
          pt = drop_keep(a.drop_seed, m, cg + i, a.drop_thresh) ? pt * a.drop_scale : 0.0f;
          ps = drop_keep(a.drop_seed, m, half + cg + i, a.drop_thresh) ? ps * a.drop_scale : 0.0f;
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
  } else {
    constexpr int NCH = (BM * BN / 8) / 256;                         // 8 (BN=128) or 4 (BN=64) chunks per thread
    constexpr int CPR = BN / 8;                                      // chunks per row
    // side loads of all chunks first (one exposed latency), then the math and the stores
    uint4 adq[NCH][2], tsq[NCH][2];
    float rmq[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int q = tid + 256 * j, row = q / CPR, c = (q % CPR) * 8;
      const int m = m0 + row, n = n0 + c;
      adq[j][0] = make_uint4(0, 0, 0, 0); adq[j][1] = make_uint4(0, 0, 0, 0);
      tsq[j][0] = make_uint4(0, 0, 0, 0); tsq[j][1] = make_uint4(0, 0, 0, 0);
      rmq[j] = 1.0f;
      if (m < a.R && n < a.N) {
        if (a.rowmask) rmq[j] = a.rowmask[m];
        if (a.gatebwd) {
          tsq[j][0] = *reinterpret_cast<const uint4*>(a.Tout + (size_t)m * a.ldts + n);
          tsq[j][1] = *reinterpret_cast<const uint4*>(a.Sout + (size_t)m * a.ldts + n);
        }
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
      const int q = tid + 256 * j, row = q / CPR, c = (q % CPR) * 8;
      const int m = m0 + row, n = n0 + c;
      if (m >= a.R || n >= a.N) continue;                            // N % 4 == 0
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
        const float* cp = a.cond + (size_t)(m / a.Tp) * a.ldc;
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
            gt[i] = drop_keep(a.drop_seed, m, n + i, a.gb_thresh) ? gt[i] * a.drop_scale : 0.0f;
            gs[i] = drop_keep(a.drop_seed, m, a.N + n + i, a.gb_thresh) ? gs[i] * a.drop_scale : 0.0f;
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
}

// ---------------------------------------------------------------------------------------
// Weight preparation: (optional) weight-norm w = g * v / ||v|| (torch weight_norm dim=0, the
// reference's modules.py:127,132,141 / attentions.py:103), then bf16 packing into
//   fwd   Pf[tap][pn(co)][ci]              (pn = gate interleave or identity)
//   dgrad Pd[taps-1-tap][ci][co]           (data-gradient conv: roles swapped, taps flipped)
// One workgroup per output channel.  Padding entries of Pf/Pd are never written (callers zero
// the buffers once).
__device__ __forceinline__ void pack_one_row(
    const float* __restrict__ v, const float* __restrict__ g, bf16_t* __restrict__ Pf, bf16_t* __restrict__ Pd,
    float* __restrict__ inv_norm, int co, int Cout, int Cin, int taps, int Npf, int Kpf, int Npd, int Kpd, int gate, float* red)
{
  const int tid = threadIdx.x;
  const int n = Cin * taps;
  const float* vr = v + (size_t)co * n;
  float scale = 1.0f;
  if (g) {
    float ss = 0.f;
    for (int i = tid; i < n; i += 256) { const float x = vr[i]; ss += x * x; }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / sqrtf(tot);
    if (tid == 0 && inv_norm) inv_norm[co] = inv;
    scale = g[co] * inv;
  }
  int pn = co;
  if (gate) {                                   // [32 tanh | 32 sigmoid] interleave per 64 packed rows
    const int half = Cout >> 1;
    const int c = co < half ? co : co - half;
    pn = (c >> 5) * 64 + (co < half ? 0 : 32) + (c & 31);
  }
  for (int i = tid; i < n; i += 256) {
    const int ci = i / taps, tap = i - ci * taps;
    const bf16_t w = f2bf(vr[i] * scale);
    if (Pf) Pf[((size_t)tap * Npf + pn) * Kpf + ci] = w;
    if (Pd) Pd[((size_t)(taps - 1 - tap) * Npd + ci) * Kpd + co] = w;
  }
}

__global__ __launch_bounds__(256) void gt_pack_conv_weights_kernel(
    const float* __restrict__ v, const float* __restrict__ g, bf16_t* __restrict__ Pf, bf16_t* __restrict__ Pd,
    float* __restrict__ inv_norm, int Cout, int Cin, int taps, int Npf, int Kpf, int Npd, int Kpd, int gate)
{
  __shared__ float red[4];
  pack_one_row(v, g, Pf, Pd, inv_norm, blockIdx.x, Cout, Cin, taps, Npf, Kpf, Npd, Kpd, gate, red);
}

// every conv of a model in ONE launch: block -> (conv, output channel) through a prefix table
__global__ __launch_bounds__(256) void gt_pack_conv_weights_multi_kernel(const gt_pack_desc* __restrict__ descs, int n)
{
  __shared__ float red[4];
  int lo = 0, hi = n - 1;
  const int blk = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (descs[mid].row_start <= blk) lo = mid; else hi = mid - 1; }
  const gt_pack_desc d = descs[lo];
  pack_one_row(d.v, d.g, static_cast<bf16_t*>(d.pack_fwd), static_cast<bf16_t*>(d.pack_dgrad), d.inv_norm, blk - d.row_start,
               d.Cout, d.Cin, d.taps, d.Np_fwd, d.Kp_fwd, d.Np_dgrad, d.Kp_dgrad, d.gate, red);
}

}  // namespace

extern "C" int gt_conv_gemm_bf16(const void* X, int ldx, const void* Wp, const float* bias,
                                 const float* cond, int ldc, const float* rowmask,
                                 void* Y, int ldy, int out_f32, const void* addend, int ldadd,
                                 void* gate_t, void* gate_s, int ldts,
                                 int R, int N, int Cin, int taps, int Tp, int Np, int Kp,
                                 int relu, int gate, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream)
{
  if (R < 0 || N <= 0 || Cin <= 0) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  if (!X || !Wp || !Y) return GT_E_INVAL;
  if (taps < 1 || taps > MAXTAPS || !(taps & 1)) return GT_E_UNSUPPORTED;
  if ((N & 3) || (Cin & 7) || (ldx & 7) || (ldy & 3) || (Kp % BK) || Kp < Cin) return GT_E_ALIGN;
  if (((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)Y) & 15) return GT_E_ALIGN;
  if (addend && (ldadd & 3)) return GT_E_ALIGN;
  if (cond && Tp <= 0) return GT_E_INVAL;
  ConvArgs a;
  a.X = static_cast<const bf16_t*>(X); a.ldx = ldx; a.W = static_cast<const bf16_t*>(Wp); a.bias = bias;
  a.cond = cond; a.ldc = ldc; a.rowmask = rowmask; a.Y = Y; a.ldy = ldy; a.addend = addend; a.ldadd = ldadd;
  a.Tout = static_cast<bf16_t*>(gate_t); a.Sout = static_cast<bf16_t*>(gate_s); a.ldts = ldts;
  a.R = R; a.N = N; a.Cin = Cin; a.taps = taps; a.Tp = Tp > 0 ? Tp : 1; a.Np = Np; a.Kp = Kp;
  a.out_f32 = out_f32; a.relu = relu;
  a.y16 = !(ldy & 7);
  { static int ex = -1; if (ex < 0) { const char* e = getenv("GT_CONV_EXP"); ex = e ? atoi(e) : 0; } a.exp_ = ex; }
  a.drop_thresh = 0; a.drop_seed = drop_seed; a.drop_scale = 1.0f; a.seed_dev = seed_dev;
  a.gatebwd = (gate == 2); a.gb_thresh = 0;
  if (drop_p > 0.0f) {
    if (drop_p >= 1.0f) return GT_E_UNSUPPORTED;
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.drop_scale = 1.0f / (1.0f - drop_p);
  }
  if (a.gatebwd) {                            // the dropout belongs to the gate's forward: replay it on the gradient only
    if (!gate_t || !gate_s || out_f32 || relu || (N & 7) || (ldts & 7) || (ldy & 7)) return GT_E_INVAL;
    if (((uintptr_t)gate_t | (uintptr_t)gate_s) & 15) return GT_E_ALIGN;
    a.gb_thresh = a.drop_thresh; a.drop_thresh = 0;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 block(256);
  if (gate == 1) {
    if (!gate_t || !gate_s || (N & 127) || Np != N || out_f32) return GT_E_INVAL;
    if ((ldts & 7) || (ldy & 7) || (((uintptr_t)gate_t | (uintptr_t)gate_s) & 15)) return GT_E_ALIGN;
    hipLaunchKernelGGL((gt_conv_gemm_kernel<128, true>), dim3(Np / 128, (R + BM - 1) / BM), block, 0, st, a);
  } else if (Np % 128 == 0) {
    if (Np < N) return GT_E_INVAL;
    hipLaunchKernelGGL((gt_conv_gemm_kernel<128, false>), dim3(Np / 128, (R + BM - 1) / BM), block, 0, st, a);
  } else {
    if (Np % 64 || Np < N) return GT_E_INVAL;
    hipLaunchKernelGGL((gt_conv_gemm_kernel<64, false>), dim3(Np / 64, (R + BM - 1) / BM), block, 0, st, a);
  }
  return gt_launch_status(__func__);
}

extern "C" int gt_pack_conv_weights(const float* v, const float* g, void* pack_fwd, void* pack_dgrad,
                                    float* inv_norm, int Cout, int Cin, int taps,
                                    int Np_fwd, int Kp_fwd, int Np_dgrad, int Kp_dgrad, int gate, void* stream)
{
  if (Cout <= 0 || Cin <= 0 || taps < 1 || taps > MAXTAPS) return GT_E_INVAL;
  if (!v || (!pack_fwd && !pack_dgrad && !(g && inv_norm))) return GT_E_INVAL;
  if (pack_fwd && (Np_fwd < Cout || Kp_fwd < Cin)) return GT_E_INVAL;
  if (pack_dgrad && (Np_dgrad < Cin || Kp_dgrad < Cout)) return GT_E_INVAL;
  if (gate && (Cout % 128)) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_pack_conv_weights_kernel, dim3(Cout), dim3(256), 0, static_cast<hipStream_t>(stream),
                     v, g, static_cast<bf16_t*>(pack_fwd), static_cast<bf16_t*>(pack_dgrad), inv_norm,
                     Cout, Cin, taps, Np_fwd, Kp_fwd, Np_dgrad, Kp_dgrad, gate);
  return gt_launch_status(__func__);
}

extern "C" int gt_pack_conv_weights_multi(const void* descs_device, int n_convs, int total_rows, void* stream)
{
  if (!descs_device || n_convs <= 0 || total_rows <= 0) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_pack_conv_weights_multi_kernel, dim3(total_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const gt_pack_desc*>(descs_device), n_convs);
  return gt_launch_status(__func__);
}
