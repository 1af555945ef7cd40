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
#include "conv_common.h"
#include "../../include/glowtts_hip.h"

namespace {
using gtconv::ConvArgs;

constexpr int BK = 64;
constexpr int LDP = 72;                       // halfs per LDS row (64 + 8 pad)
constexpr int MAXTAPS = 5;


// BM x BN tile: 128 x {128|64} (2 workgroups per CU), or 256 x 64 (1 per CU): the weight tile is re-streamed from L2
// once per ROW tile, so for a wide-N, many-tap conv (in_layer: 737 KB of weights) the tall tile halves the L2->LDS
// fill traffic per output (295 -> 221 KB per 16 K outputs) at the same accumulator count per wave.
template <int BM, int BN, bool GATE>
__global__ __launch_bounds__(256, BM == 256 ? 1 : 2) void gt_conv_gemm_kernel(ConvArgs a)
{
  constexpr int XROWS = BM + MAXTAPS - 1;
  constexpr int XCH = (XROWS * 8 + 255) / 256;                      // activation 16-B chunks per thread per slice (5 or 9)
  // wave tiling: 4 waves; a wave owns NB 32-channel blocks x MB 32-row blocks.  64 x 64 tiles: 2 x 2 waves of 32 x 32
  constexpr int NB = (BM == 64 && BN == 64) ? 1 : 2;   // 32-channel MFMA blocks per wave
  constexpr int WN = BN / (32 * NB);          // waves along channels
  constexpr int WM = 4 / WN;                  // waves along rows
  constexpr int MB = BM / (32 * WM);          // 32-row MFMA blocks per wave
  static_assert(WN * WM == 4 && MB >= 1, "tile does not split over 4 waves");
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
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the column tiles
  // that share one activation row tile are given to ONE XCD (row tile rt lives on XCD rt % 8) — otherwise every
  // activation tile is pulled from HBM/MALL into N/BN different L2s (measured: FETCH_SIZE 4x the algorithmic bytes).
  const int nct = a.Np / BN, nrt = (a.R + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rt = xcd + 8 * (slot / nct), ct = slot - (slot / nct) * nct;
  if (rt >= nrt) return;
  const int n0 = ct * BN, m0 = rt * BM;
  const int taps = a.taps, padl = taps >> 1;
  const int NS = a.Kp / BK, NIT = NS * taps;
  const int xrows = BM + taps - 1;

  // staging registers as named scalars (arrays that live across the conditional prefetch end up
  // in scratch memory)
  uint4 w0 = {}, w1 = {}, w2 = {}, w3 = {}, x0 = {}, x1 = {}, x2 = {}, x3 = {}, x4 = {}, x5 = {}, x6 = {}, x7 = {}, x8 = {};
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
#define load_X(sl)  do { x0 = ldx1(sl, 0); x1 = ldx1(sl, 1); x2 = ldx1(sl, 2); x3 = ldx1(sl, 3); x4 = ldx1(sl, 4); \
                          if (XCH > 5) { x5 = ldx1(sl, 5); x6 = ldx1(sl, 6); x7 = ldx1(sl, 7); x8 = ldx1(sl, 8); } } while (0)
#define store_X(bf) do { stx1(bf, 0, x0); stx1(bf, 1, x1); stx1(bf, 2, x2); stx1(bf, 3, x3); stx1(bf, 4, x4); \
                          if (XCH > 5) { stx1(bf, 5, x5); stx1(bf, 6, x6); stx1(bf, 7, x7); stx1(bf, 8, x8); } } while (0)

  f32x16_t acc[NB][MB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
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
#ifdef GT_DEV_EXPERIMENTS
    const bool ld_ok = !(a.exp_ & 2);
#else
    constexpr bool ld_ok = true;
#endif
    if (has && ld_ok) { load_W(nxt); if (newslice) load_X(slice + 1); }

    const bf16_t* wsb = &Ws[it & 1][(32 * NB * wn + r) * LDP + 8 * h];
    const bf16_t* xsb = &Xs[slice & 1][(mrow0 + r + tap) * LDP + 8 * h];
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t af[NB], bfm[MB];
#pragma unroll
      for (int bn = 0; bn < NB; ++bn) af[bn] = *reinterpret_cast<const bf16x8_t*>(wsb + bn * 32 * LDP + ks * 16);
#pragma unroll
      for (int bm = 0; bm < MB; ++bm) bfm[bm] = *reinterpret_cast<const bf16x8_t*>(xsb + bm * 32 * LDP + ks * 16);
#pragma unroll
      for (int bn = 0; bn < NB; ++bn)
#pragma unroll
        for (int bm = 0; bm < MB; ++bm)
          acc[bn][bm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[bn], bfm[bm], acc[bn][bm], 0, 0, 0);
    }

    if (has && ld_ok) { store_W(nxt & 1); if (newslice) store_X((slice + 1) & 1); }
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
#ifdef GT_DEV_EXPERIMENTS
  if (a.exp_ & 1) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc += acc[i][j][e];
    if (sacc == 123.456f) static_cast<bf16_t*>(a.Y)[tid] = 1;
    return;
  }
#endif
#pragma unroll
  for (int bm = 0; bm < MB; ++bm)
#pragma unroll
    for (int bn = 0; bn < NB; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(&es[(mrow0 + 32 * bm + r) * EP + 32 * NB * wn + 32 * bn + 8 * g + 4 * h]) =
            make_float4(acc[bn][bm][4 * g], acc[bn][bm][4 * g + 1], acc[bn][bm][4 * g + 2], acc[bn][bm][4 * g + 3]);
  __syncthreads();

  if (GATE) gtconv::epilogue_gate<256, BM, BN>(a, es, EP, m0, n0, tid);
  else      gtconv::epilogue_plain<256, BM, BN>(a, es, EP, m0, n0, tid);
}

// ---------------------------------------------------------------------------------------
// Short-and-deep variant: 64 rows x 64 channels, ALL taps of a K slice per pipeline stage.  The text encoder's k = 3 / k = 5
// convs run a few hundred workgroups (at most one or two per CU) through up to 36 (slice, tap) steps of 4 MFMAs each: every
// step then costs one L2 round trip that nothing hides (47 us for the 768 -> 192, k = 3 FFN conv whose MFMAs take 2 us).
// One stage = the activation slice + the AT weight tiles of that slice, so the same loads are in flight AT at a time and
// the step count falls to Kp / 64.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int AT>
__global__ __launch_bounds__(256, AT == 3 ? 2 : 1) void gt_conv_gemm_taps_kernel(ConvArgs a)
{
  constexpr int BM = 64, BN = 64;
  constexpr int XROWS = BM + MAXTAPS - 1;
  constexpr int XCH = (XROWS * 8 + 255) / 256;                      // 3
  constexpr int XS_HALFS = XROWS * LDP, WS_HALFS = BN * LDP;
  constexpr int EP = BN + 4;
  constexpr int MAIN_BYTES = 2 * (XS_HALFS + AT * WS_HALFS) * 2, EPI_BYTES = BM * EP * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
  bf16_t (*Xs)[XS_HALFS] = reinterpret_cast<bf16_t (*)[XS_HALFS]>(smem);
  bf16_t (*Ws)[WS_HALFS] = reinterpret_cast<bf16_t (*)[WS_HALFS]>(smem + 2 * XS_HALFS * 2);       // [stage * AT + tap]

  if (a.seed_dev) a.drop_seed ^= *a.seed_dev;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int r = lane & 31, h = lane >> 5;
  const int nct = a.Np / BN, nrt = (a.R + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rt = xcd + 8 * (slot / nct), ct = slot - (slot / nct) * nct;
  if (rt >= nrt) return;
  const int n0 = ct * BN, m0 = rt * BM;
  const int taps = a.taps, padl = taps >> 1;
  const int NS = a.Kp / BK;
  const int xrows = BM + taps - 1;

  u32x4_t wr[AT][2], xr[XCH];
  auto load_stage = [&](int slice) {
#pragma unroll
    for (int t = 0; t < AT; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
        if (t < taps) wr[t][i] = *reinterpret_cast<const u32x4_t*>(a.W + ((size_t)(t * a.Np + n0 + row) * a.Kp + slice * BK + c8 * 8));
      }
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
      int gm = m0 - padl + (row < xrows ? row : xrows - 1);
      gm = gm < 0 ? 0 : (gm >= a.R ? a.R - 1 : gm);
      const int ch = slice * BK + c8 * 8;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (ch < a.Cin) v = *reinterpret_cast<const u32x4_t*>(a.X + (size_t)gm * a.ldx + ch);
      xr[i] = v;
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int t = 0; t < AT; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
        if (t < taps) *reinterpret_cast<u32x4_t*>(&Ws[buf * AT + t][row * LDP + c8 * 8]) = wr[t][i];
      }
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
      if (row < xrows) *reinterpret_cast<u32x4_t*>(&Xs[buf][row * LDP + c8 * 8]) = xr[i];
    }
  };

  f32x16_t acc[1][1];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.0f;

  load_stage(0); store_stage(0);
  __syncthreads();
  const int mrow0 = wm * 32;
  for (int s = 0; s < NS; ++s) {
    const bool has = s + 1 < NS;
    if (has) load_stage(s + 1);
#pragma unroll
    for (int t = 0; t < AT; ++t) {
      if (t < taps) {
        const bf16_t* wsb = &Ws[(s & 1) * AT + t][(32 * wn + r) * LDP + 8 * h];
        const bf16_t* xsb = &Xs[s & 1][(mrow0 + r + t) * LDP + 8 * h];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks)
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(wsb + ks * 16),
                                                              *reinterpret_cast<const bf16x8_t*>(xsb + ks * 16), acc[0][0], 0, 0, 0);
      }
    }
    if (has) store_stage((s + 1) & 1);
    __syncthreads();
  }
  float* es = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(&es[(mrow0 + r) * EP + 32 * wn + 8 * g + 4 * h]) =
        make_float4(acc[0][0][4 * g], acc[0][0][4 * g + 1], acc[0][0][4 * g + 2], acc[0][0][4 * g + 3]);
  __syncthreads();
  gtconv::epilogue_plain<256, BM, BN>(a, es, EP, m0, n0, tid);
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
  if (gate & 1) {                               // [32 tanh | 32 sigmoid] interleave per 64 packed rows
    const int half = Cout >> 1;
    const int c = co < half ? co : co - half;
    pn = (c >> 5) * 64 + (co < half ? 0 : 32) + (c & 31);
  } else if (gate & 16) {                       // [16 tanh | 16 sigmoid] per 32 packed rows: every MFMA block is self-contained
    const int half = Cout >> 1;
    const int c = co < half ? co : co - half;
    pn = (c >> 4) * 32 + (co < half ? 0 : 16) + (c & 15);
  }
  // flag 2 / 4: forward / data-gradient image in MFMA-fragment order for gt_conv_gemm2_bf16:
  //   [tap][n / 32][k / 16][lane = n % 32 + 32 * ((k % 16) / 8)][k % 8]   (one 1-KB A-fragment per (n/32, k/16))
  const bool ffrag = gate & 2, dfrag = gate & 4, split3 = gate & 8;
  const int NBf = Npf >> 5, KSf = Kpf >> 4, NBd = Npd >> 5, KSd = Kpd >> 4;
  for (int i = tid; i < n; i += 256) {
    const int ci = i / taps, tap = i - ci * taps;
    const float wf = vr[i] * scale;
    const bf16_t w = f2bf(wf);
    if (split3) {
      // bf16x3 GEMM (near-fp32 product from three bf16 MFMA passes): the reduction axis is K-concatenated as
      // [w_hi ; w_lo ; w_hi] against activations [x_hi | x_hi | x_lo]  ->  x_hi w_hi + x_hi w_lo + x_lo w_hi
      const bf16_t wl = f2bf(wf - bf2f(w));
      if (Pf) { bf16_t* r = Pf + ((size_t)tap * Npf + pn) * Kpf; r[ci] = w; r[Cin + ci] = wl; r[2 * Cin + ci] = w; }
      if (Pd) { bf16_t* r = Pd + ((size_t)(taps - 1 - tap) * Npd + ci) * Kpd; r[co] = w; r[Cout + co] = wl; r[2 * Cout + co] = w; }
      continue;
    }
    if (Pf) {
      if (ffrag) Pf[((((size_t)tap * NBf + (pn >> 5)) * KSf + (ci >> 4)) * 64 + (pn & 31) + 32 * ((ci & 15) >> 3)) * 8 + (ci & 7)] = w;
      else       Pf[((size_t)tap * Npf + pn) * Kpf + ci] = w;
    }
    if (Pd) {
      const int tt = taps - 1 - tap;
      if (dfrag) Pd[((((size_t)tt * NBd + (ci >> 5)) * KSd + (co >> 4)) * 64 + (ci & 31) + 32 * ((co & 15) >> 3)) * 8 + (co & 7)] = w;
      else       Pd[((size_t)tt * Npd + ci) * Kpd + co] = w;
    }
  }
}

// Same packing, 8 output channels per workgroup (Cout % 8 == 0, Cin % 8 == 0, Cin*taps <= PK8_MAXN): the scaled bf16
// rows are staged in LDS so that BOTH images leave as 16-byte stores — 8 consecutive input channels of one row for
// the forward image, 8 consecutive output channels of one (tap, ci) for the data-gradient image (one workgroup per
// row writes that image as scattered 2-byte stores: 0.8 TB/s).
constexpr int PK8_MAXN = 2304;
__device__ __forceinline__ size_t pk_index(bool frag, int t, int nrow, int k, int Np, int Kp)
{
  return frag ? ((((size_t)t * (Np >> 5) + (nrow >> 5)) * (Kp >> 4) + (k >> 4)) * 64 + (nrow & 31) + 32 * ((k & 15) >> 3)) * 8 + (k & 7)
              : ((size_t)t * Np + nrow) * Kp + k;
}
#ifndef PK_THREADS
#define PK_THREADS 256
#endif
constexpr int PK_RPW = 8 / (PK_THREADS / 64);       // rows per wave
constexpr int PK8_SMALLN = 1024;                    // the row length the small form holds: 16 KB of LDS and 32 row registers instead of
                                                    // 36 KB and 72, twice the resident workgroups (the decoder's convs: 960; 90 % of the step's weights)
template <int MAXN>
__device__ __forceinline__ void pack_rows8(
    const float* __restrict__ v, const float* __restrict__ g, bf16_t* __restrict__ Pf, bf16_t* __restrict__ Pd,
    float* __restrict__ inv_norm, int co0, int Cout, int Cin, int taps, int Npf, int Kpf, int Npd, int Kpd, int gate,
    bf16_t* tile, float* scl)
{
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = Cin * taps, n4 = n >> 2;                          // n % 8 == 0
  // wave w: rows 2w, 2w+1, each held in registers (<= 9 float4 per lane) from ONE batch of loads that are all in flight
  // together — a load / accumulate loop per row left the launch bound by ~30 dependent HBM latencies per workgroup
  // (171 us for the 28 M weights of the step; the bytes alone are ~40 us)
  float4 x[PK_RPW][MAXN / 256];
  const bool al = ((reinterpret_cast<uintptr_t>(v) | ((size_t)n * 4)) & 15) == 0;
#pragma unroll
  for (int k = 0; k < PK_RPW; ++k) {
    const float* vr = v + (size_t)(co0 + PK_RPW * w + k) * n;
#pragma unroll
    for (int j = 0; j < MAXN / 256; ++j) {
      const int i = lane + 64 * j;
      x[k][j] = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef PK_EXP
      if (PK_EXP & 4) { x[k][j] = make_float4(0.001f * i, 0.5f, 0.25f, 1.f); continue; }
#endif
      if (i < n4) {
        if (al) x[k][j] = *reinterpret_cast<const float4*>(vr + 4 * i);
        else    x[k][j] = make_float4(vr[4 * i], vr[4 * i + 1], vr[4 * i + 2], vr[4 * i + 3]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PK_RPW; ++k) {
    const int rr = PK_RPW * w + k, co = co0 + rr;
    float sc = 1.0f;
    if (g) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < MAXN / 256; ++j) ss += x[k][j].x * x[k][j].x + x[k][j].y * x[k][j].y + x[k][j].z * x[k][j].z + x[k][j].w * x[k][j].w;
      ss = wave_sum(ss);
      const float inv = 1.0f / sqrtf(ss);
      if (lane == 0 && inv_norm) inv_norm[co] = inv;
      sc = g[co] * inv;
    }
#pragma unroll
    for (int j = 0; j < MAXN / 256; ++j) {
      const int i = lane + 64 * j;
      if (i < n4)
        *reinterpret_cast<uint2*>(tile + rr * n + 4 * i) =
            make_uint2((uint32_t)f2bf(x[k][j].x * sc) | ((uint32_t)f2bf(x[k][j].y * sc) << 16),
                       (uint32_t)f2bf(x[k][j].z * sc) | ((uint32_t)f2bf(x[k][j].w * sc) << 16));
    }
  }
  (void)scl;
  __syncthreads();
  const bool ffrag = gate & 2, dfrag = gate & 4;
  const int C8 = Cin >> 3;
#ifdef PK_EXP
  if (PK_EXP & 1) Pf = nullptr;
  if (PK_EXP & 2) Pd = nullptr;
#endif
  if (Pf) {
    for (int q = tid; q < 8 * taps * C8; q += PK_THREADS) {
      // the 8 rows of the group are the fastest index: their 16-byte pieces are neighbours in the packed image (whole 128-byte lines per store)
      const int rr = q & 7, rem = q >> 3, tap = rem / C8, c8 = rem - tap * C8;
      const int co = co0 + rr;
      int pn = co;
      if (gate & 1) { const int half = Cout >> 1, c = co < half ? co : co - half; pn = (c >> 5) * 64 + (co < half ? 0 : 32) + (c & 31); }
      else if (gate & 16) { const int half = Cout >> 1, c = co < half ? co : co - half; pn = (c >> 4) * 32 + (co < half ? 0 : 16) + (c & 15); }
      const bf16_t* t = tile + rr * n + (c8 * 8) * taps + tap;
      uint32_t u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = (uint32_t)t[(2 * j) * taps] | ((uint32_t)t[(2 * j + 1) * taps] << 16);
      *reinterpret_cast<uint4*>(Pf + pk_index(ffrag, tap, pn, c8 * 8, Npf, Kpf)) = make_uint4(u[0], u[1], u[2], u[3]);
    }
  }
  if (Pd) {
    for (int q = tid; q < taps * Cin; q += PK_THREADS) {
      const int tap = q / Cin, ci = q - tap * Cin;
      const bf16_t* t = tile + ci * taps + tap;
      uint32_t u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = (uint32_t)t[(2 * j) * n] | ((uint32_t)t[(2 * j + 1) * n] << 16);
      *reinterpret_cast<uint4*>(Pd + pk_index(dfrag, taps - 1 - tap, ci, co0, Npd, Kpd)) = make_uint4(u[0], u[1], u[2], u[3]);
    }
  }
}

template <int MAXN>
__global__ __launch_bounds__(PK_THREADS) void gt_pack_conv_weights_multi8_kernel(const gt_pack_desc* __restrict__ descs, int n)
{
  __shared__ __attribute__((aligned(16))) bf16_t tile[8 * MAXN];
  __shared__ float scl[8];
  __shared__ int sel;
  const int row = blockIdx.x * 8;
  // the conv this row group belongs to = the last descriptor starting at or before it: every lane tests one descriptor (one load
  // latency instead of a binary search's eight dependent ones)
  if (threadIdx.x == 0) sel = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += PK_THREADS) if (descs[i].row_start <= row) atomicMax(&sel, i);
  __syncthreads();
  const gt_pack_desc d = descs[sel];
  pack_rows8<MAXN>(d.v, d.g, static_cast<bf16_t*>(d.pack_fwd), static_cast<bf16_t*>(d.pack_dgrad), d.inv_norm, row - d.row_start,
             d.Cout, d.Cin, d.taps, d.Np_fwd, d.Kp_fwd, d.Np_dgrad, d.Kp_dgrad, d.gate, tile, scl);
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
                                 int relu, int gate, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev,
                                 const int32_t* row0, int B, int tile, void* stream)
{
  if (R < 0 || N <= 0 || Cin <= 0) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  if (!X || !Wp || !Y) return GT_E_INVAL;
  if (taps < 1 || taps > MAXTAPS || !(taps & 1)) return GT_E_UNSUPPORTED;
  if ((N & 3) || (Cin & 7) || (ldx & 7) || (ldy & 3) || (Kp % BK) || Kp < Cin) return GT_E_ALIGN;
  if (((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)Y) & 15) return GT_E_ALIGN;
  if (addend && (ldadd & 3)) return GT_E_ALIGN;
  if (cond && (Tp <= 0 || (row0 && B <= 0))) return GT_E_INVAL;
  if (tile < GT_TILE_AUTO || tile > GT_TILE_64x64_TAPS) return GT_E_INVAL;
  if (Np % 64 || Np < N) return GT_E_INVAL;
  ConvArgs a;
  a.X = static_cast<const bf16_t*>(X); a.ldx = ldx; a.W = static_cast<const bf16_t*>(Wp); a.bias = bias;
  a.cond = cond; a.ldc = ldc; a.rowmask = rowmask; a.Y = Y; a.ldy = ldy; a.addend = addend; a.ldadd = ldadd;
  a.Tout = static_cast<bf16_t*>(gate_t); a.Sout = static_cast<bf16_t*>(gate_s); a.ldts = ldts;
  a.R = R; a.N = N; a.Cin = Cin; a.taps = taps; a.Tp = Tp > 0 ? Tp : 1; a.Np = Np; a.Kp = Kp;
  a.row0 = row0; a.B = B;
  a.out_f32 = out_f32; a.relu = relu;
  a.y16 = !(ldy & 7);
  a.exp_ = 0;
#ifdef GT_DEV_EXPERIMENTS                    // tools/conv_exp.py builds its own library with this flag; never in the product build
  { const char* e = getenv("GT_CONV_EXP"); a.exp_ = e ? atoi(e) : 0; }
#endif
  a.drop_thresh = 0; a.drop_seed = drop_seed; a.drop_scale = 1.0f; a.seed_dev = seed_dev;
  a.gatebwd = (gate == 2); a.gb_thresh = 0;
  a.maskbwd = (gate == 3);
  if (drop_p > 0.0f) {
    if (drop_p >= 1.0f) return GT_E_UNSUPPORTED;
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.drop_scale = 1.0f / (1.0f - drop_p);
  }
  if (a.gatebwd) {                            // the dropout belongs to the gate's forward: replay it on the gradient only
    if (!gate_t || !gate_s || out_f32 || relu || (N & 7) || (ldts & 7) || (ldy & 7)) return GT_E_INVAL;
    if (((uintptr_t)gate_t | (uintptr_t)gate_s) & 15) return GT_E_ALIGN;
    a.gb_thresh = a.drop_thresh; a.drop_thresh = 0;
  }
  if (a.maskbwd) {                            // the scale 1 / (1 - p) of the forward's dropout is replayed; no fresh mask is drawn
    if (!gate_t || out_f32 || relu || (N & 7) || (ldts & 7) || (ldy & 7)) return GT_E_INVAL;
    if ((uintptr_t)gate_t & 15) return GT_E_ALIGN;
    a.drop_thresh = 0;
  }
  if (gate == 1) {                            // gate: N = Np = 2 * half, [32 tanh | 32 sigmoid] per 64 packed rows, 128-column tiles
    if (!gate_t || !gate_s || (N & 127) || Np != N || out_f32) return GT_E_INVAL;
    if ((ldts & 7) || (ldy & 7) || (((uintptr_t)gate_t | (uintptr_t)gate_s) & 15)) return GT_E_ALIGN;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 block(256);
  // Tile choice (GT_TILE_AUTO).  Short tiles: when 128-row tiles give no more workgroups than the chip has CUs (one wave
  // per SIMD, nothing to hide the LDS / L2 latency behind), 64-row tiles double the resident waves at the price of
  // streaming the weights twice as often (cfg2 step 9.00 -> 8.76 ms; with the threshold at 512 workgroups the gain is
  // gone).  256-row tiles (gate only) never paid on the measured shapes and are reachable only through `tile`.
  constexpr int SHORT_TILE_MAX_WGS = 256;
  const int bnsel = (gate == 1 || Np % 128 == 0) ? 128 : 64;
  if (tile == GT_TILE_AUTO) {
    const bool short_rows = ((R + 127) / 128) * (Np / bnsel) <= SHORT_TILE_MAX_WGS;
    tile = short_rows ? (bnsel == 128 ? GT_TILE_64x128 : GT_TILE_64x64) : (bnsel == 128 ? GT_TILE_128x128 : GT_TILE_128x64);
    // short AND deep (the text encoder's k = 3 FFN / k = 5 pre-net convs): all taps of a K slice per pipeline stage
    if (short_rows && taps > 1 && gate != 1 && ((R + 63) / 64) * (Np / 64) <= 4 * SHORT_TILE_MAX_WGS) tile = GT_TILE_64x64_TAPS;
  }
  if (tile == GT_TILE_64x64_TAPS) {
    if (gate == 1) return GT_E_INVAL;
    const dim3 grid_t(8 * (((R + 63) / 64 + 7) / 8) * (Np / 64));
    if (taps <= 3) hipLaunchKernelGGL(gt_conv_gemm_taps_kernel<3>, grid_t, block, 0, st, a);
    else           hipLaunchKernelGGL(gt_conv_gemm_taps_kernel<5>, grid_t, block, 0, st, a);
    return gt_launch_status(__func__);
  }
  const int bm = (tile == GT_TILE_64x64 || tile == GT_TILE_64x128) ? 64 : (tile == GT_TILE_256x64 ? 256 : 128);
  const int bn = (tile == GT_TILE_64x128 || tile == GT_TILE_128x128) ? 128 : 64;
  if (Np % bn) return GT_E_INVAL;
  if (gate == 1 && bn != 128 && tile != GT_TILE_256x64) return GT_E_INVAL;       // the gate pairs 32 + 32 columns of one wave
  if (tile == GT_TILE_256x64 && gate != 1) return GT_E_INVAL;
  const dim3 grid(8 * (((R + bm - 1) / bm + 7) / 8) * (Np / bn));
  switch (tile) {
    case GT_TILE_64x64:   hipLaunchKernelGGL((gt_conv_gemm_kernel<64, 64, false>), grid, block, 0, st, a); break;
    case GT_TILE_64x128:
      if (gate == 1) hipLaunchKernelGGL((gt_conv_gemm_kernel<64, 128, true>), grid, block, 0, st, a);
      else           hipLaunchKernelGGL((gt_conv_gemm_kernel<64, 128, false>), grid, block, 0, st, a);
      break;
    case GT_TILE_128x64:  hipLaunchKernelGGL((gt_conv_gemm_kernel<128, 64, false>), grid, block, 0, st, a); break;
    case GT_TILE_128x128:
      if (gate == 1) hipLaunchKernelGGL((gt_conv_gemm_kernel<128, 128, true>), grid, block, 0, st, a);
      else           hipLaunchKernelGGL((gt_conv_gemm_kernel<128, 128, false>), grid, block, 0, st, a);
      break;
    default:              hipLaunchKernelGGL((gt_conv_gemm_kernel<256, 64, true>), grid, block, 0, st, a); break;
  }
  return gt_launch_status(__func__);
}

extern "C" int gt_pack_conv_weights(const float* v, const float* g, void* pack_fwd, void* pack_dgrad,
                                    float* inv_norm, int Cout, int Cin, int taps,
                                    int Np_fwd, int Kp_fwd, int Np_dgrad, int Kp_dgrad, int gate, void* stream)
{
  if (Cout <= 0 || Cin <= 0 || taps < 1 || taps > MAXTAPS) return GT_E_INVAL;
  if (!v || (!pack_fwd && !pack_dgrad && !(g && inv_norm))) return GT_E_INVAL;
  const int km = (gate & 8) ? 3 : 1;          // split3: K-concatenated [hi; lo; hi] images
  if ((gate & 8) && (gate & 7)) return GT_E_UNSUPPORTED;
  if (pack_fwd && (Np_fwd < Cout || Kp_fwd < km * Cin)) return GT_E_INVAL;
  if (pack_dgrad && (Np_dgrad < Cin || Kp_dgrad < km * Cout)) return GT_E_INVAL;
  if ((gate & 1) && (Cout % 64)) return GT_E_UNSUPPORTED;
  if ((gate & 16) && ((Cout % 32) || (gate & 1))) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_pack_conv_weights_kernel, dim3(Cout), dim3(256), 0, static_cast<hipStream_t>(stream),
                     v, g, static_cast<bf16_t*>(pack_fwd), static_cast<bf16_t*>(pack_dgrad), inv_norm,
                     Cout, Cin, taps, Np_fwd, Kp_fwd, Np_dgrad, Kp_dgrad, gate);
  return gt_launch_status(__func__);
}

extern "C" int gt_pack_conv_weights_multi(const void* descs_device, int n_convs, int total_rows, int group8, void* stream)
{
  if (!descs_device || n_convs <= 0 || total_rows <= 0) return GT_E_INVAL;
  if (group8) {                                 // 1: rows of <= 2304 elements; 2: the caller vouches for <= 1024 (GT_PACK_SMALL_ROW)
    if (total_rows & 7) return GT_E_INVAL;
    if (group8 == 2)
      hipLaunchKernelGGL(gt_pack_conv_weights_multi8_kernel<PK8_SMALLN>, dim3(total_rows / 8), dim3(PK_THREADS), 0, static_cast<hipStream_t>(stream),
                         static_cast<const gt_pack_desc*>(descs_device), n_convs);
    else
      hipLaunchKernelGGL(gt_pack_conv_weights_multi8_kernel<PK8_MAXN>, dim3(total_rows / 8), dim3(PK_THREADS), 0, static_cast<hipStream_t>(stream),
                         static_cast<const gt_pack_desc*>(descs_device), n_convs);
    return gt_launch_status(__func__);
  }
  hipLaunchKernelGGL(gt_pack_conv_weights_multi_kernel, dim3(total_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const gt_pack_desc*>(descs_device), n_convs);
  return gt_launch_status(__func__);
}
