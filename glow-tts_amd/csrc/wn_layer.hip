// One WaveNet layer of the flow decoder as ONE kernel, forward and backward, gfx950 (bf16 MFMA v_mfma_f32_32x32x16_bf16).
//
// reference modules.WN.forward, one iteration of the loop (modules.py:151-170):
//     x_in = drop(in_layer_i(x));  acts = tanh(x_in[:H] + g[:H]) * sigmoid(x_in[H:] + g[H:]);
//     rs = res_skip_i(acts);  x = (x + rs[:H]) * mask;  output += rs[H:]
// Round 1 ran this as a k=5 gate conv kernel followed by a 1x1 GEMM kernel per layer (and the mirror pair in the
// backward); every launch on the decoder's dependent chain cost ~10 us beyond its arithmetic.  Here a workgroup owns
// 64 rows and ALL channels of them, so the second contraction runs on the tile the first one just produced:
//
//   forward   stage 1: x_in = conv_k5(x)           N = 2H = 384, K = 5 * 192      MFMA, 4 waves x (64 rows x 96 columns)
//             epilogue: bias, dropout, cond, gate -> T, S, acts (HBM, for the backward) and acts -> LDS (bf16)
//             stage 2: res = acts @ W_res^T         N = 192, K = 192               MFMA on the LDS tile
//             epilogue: x_next = (x + res + b_res) * mask
//   backward  stage 1: dX = conv_k5^T(d pre_{i+1}) N = 192, K = 5 * 384           data gradient of the NEXT layer's in_layer
//             epilogue: dX = (dX + residual-path gradient) * mask -> HBM (the res conv's weight gradient reads it) and LDS
//             stage 2: d acts_i = dX @ W_res_i      N = 192, K = 192
//             epilogue: + skip-path gradient, gate backward with the forward's dropout replayed -> d pre_i [R, 2H]
//
// The skip half of res_skip stays what round 1 made it: ONE K = n*H GEMM per WaveNet over the layers' gated activations.
// Weights stream L2 -> registers -> LDS double buffers (the packed images of gt_pack_conv_weights); activations keep the
// rows layout (zero halo rows between utterances, so the 5 taps are 5 shifted row-block GEMMs with no boundary logic).
#include "common.h"
#include "../../include/glowtts_hip.h"

namespace {

constexpr int H = 192;                        // hidden channels (configs/*.json hidden_channels_dec)
constexpr int BM = 64;                        // rows per workgroup
constexpr int BK = 64;                        // K slice
constexpr int LDP = 72;                       // halfs per LDS operand row (64 + 8: conflict-free ds_read_b128 over 16 rows)
constexpr int TAPS = 5;
constexpr int XROWS = BM + TAPS - 1;          // 68 activation rows per slice
constexpr int AP = H + 8;                     // pitch (halfs) of the stage-2 activation tile

struct WnArgs {
  const bf16_t* X; int ldx;                   // stage-1 operand rows (fwd: layer input [R, H]; bwd: d pre of the next layer [R, 2H])
  const bf16_t* W1;                           // stage-1 packed weights [5][N1p][K1p]
  const bf16_t* W2;                           // stage-2 packed weights [H][H(p)] (NULL: no second stage)
  const float* bias1; const float* bias2;     // fwd: in_layer bias [2H], res bias [H]
  const float* cond; int ldc; int B; const int32_t* row0; int Tp;     // fwd: conditioning of the gate (per utterance B > 0, per row B == 0)
  const float* rowmask;
  bf16_t* acts; int ldacts;                   // fwd out: gated activations (window of the WN's [R, n*H] buffer)
  bf16_t* Tt; bf16_t* Ss; int ldts;           // fwd out / bwd in: saved tanh / sigmoid halves
  bf16_t* Xnext; int ldxn;                    // fwd out: next layer's input;  bwd out: dX [R, H]
  const bf16_t* resid; int ldres;             // fwd: x for the residual (== X);  bwd: residual-path gradient (may be NULL)
  const bf16_t* viaskip; int ldvs;            // bwd: skip-path gradient of the layer's acts [R, H]
  bf16_t* dpre; int lddp;                     // bwd out: [R, 2H]
  bf16_t* dpre_c;                             // bwd out (optional): the same BEFORE the dropout mask = gradient of the gate's cond term
  int R, K1p, K2p;
  uint32_t drop_thresh, drop_seed; float drop_scale; const uint32_t* seed_dev;
  // live timing of the dominant kernel inside a captured graph (bench.py): stamps[2*launch] = min start, [2*launch+1] = max end
  unsigned long long* stamps; int stamp_slot;
};

__device__ __forceinline__ void stamp_begin(const WnArgs& a)
{
  if (a.stamps && threadIdx.x == 0) atomicMin(a.stamps + 2 * a.stamp_slot, (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void stamp_end(const WnArgs& a)
{
  if (a.stamps && threadIdx.x == 0) atomicMax(a.stamps + 2 * a.stamp_slot + 1, (unsigned long long)wall_clock64());
}

// One K loop of an implicit GEMM on a 64-row tile: Y[64, N] = sum_tap sum_k Xs[row + tap, k] * W[tap][n][k].
//   NW x MW waves (NW * MW == 4); a wave owns NBW 32-column blocks x MBW 32-row blocks.
//   W image [taps][Np][Kp] row-major; the weight slice of step `it` and the activation slice of K-slice `sl` are fetched
//   L2 -> registers while the MFMAs of the previous step run, then dropped into the other LDS buffer (one barrier per step).
//   XS == true: activations come from global memory (rows m0 - 2 .. m0 + 65 of X, clamped), staged in Xs[2];
//   XS == false: they are already in LDS (`Atile`, pitch AP, the whole K), taps == 1.
template <int N, int NW, int MW, int NBW, int MBW, bool XS, int taps>
__device__ __forceinline__ void gemm_tile(const bf16_t* __restrict__ W, int Np, int Kp,
                                          const bf16_t* __restrict__ X, int ldx, int Kx, int R, int m0,
                                          bf16_t* Ws, bf16_t* Xs, const bf16_t* Atile, f32x16_t (&acc)[NBW][MBW])
{
  static_assert(NW * MW == 4 && NW * NBW * 32 == N && MW * MBW * 32 == BM, "wave tiling");
  constexpr int WCH = N * 8 / 256;            // 16-byte weight chunks per thread per step
  constexpr int XCH = (XROWS * 8 + 255) / 256;
  constexpr int WS_HALFS = N * LDP, XS_HALFS = XROWS * LDP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % NW, wm = wave / NW;
  const int r = lane & 31, h = lane >> 5;
  const int NS = Kp / BK, NIT = NS * taps;
  const int padl = taps >> 1;
  const int xrows = BM + taps - 1;

  uint4 wreg[WCH], xreg[XCH];
  auto load_w = [&](int it) {
    const int slice = it / taps, tap = it - slice * taps;
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
      wreg[i] = *reinterpret_cast<const uint4*>(W + ((size_t)(tap * Np + row) * Kp + slice * BK + c8 * 8));
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
      *reinterpret_cast<uint4*>(Ws + buf * WS_HALFS + row * LDP + c8 * 8) = wreg[i];
    }
  };
  auto load_x = [&](int slice) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
      int gm = m0 - padl + (row < xrows ? row : xrows - 1);
      gm = gm < 0 ? 0 : (gm >= R ? R - 1 : gm);
      const int ch = slice * BK + c8 * 8;
      xreg[i] = make_uint4(0, 0, 0, 0);
      if (ch < Kx) xreg[i] = *reinterpret_cast<const uint4*>(X + (size_t)gm * ldx + ch);
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int chunk = tid + 256 * i, row = chunk >> 3, c8 = chunk & 7;
      if (row < xrows) *reinterpret_cast<uint4*>(Xs + buf * XS_HALFS + row * LDP + c8 * 8) = xreg[i];
    }
  };

#pragma unroll
  for (int i = 0; i < NBW; ++i)
#pragma unroll
    for (int j = 0; j < MBW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  load_w(0); store_w(0);
  if (XS) { load_x(0); store_x(0); }
  __syncthreads();

  // Every prefetch below is unconditional (the last step / slice re-fetches itself into the buffer nobody reads again):
  // staging registers that are defined on one side of a branch only end up in scratch memory.
  for (int slice = 0; slice < NS; ++slice) {
    if (XS) load_x(slice + 1 < NS ? slice + 1 : NS - 1);
#pragma unroll
    for (int tap = 0; tap < taps; ++tap) {
      const int it = slice * taps + tap;
      load_w(it + 1 < NIT ? it + 1 : NIT - 1);

      const bf16_t* wsb = Ws + (it & 1) * WS_HALFS + (32 * NBW * wn + r) * LDP + 8 * h;
      const bf16_t* xsb = XS ? Xs + (slice & 1) * XS_HALFS + (32 * MBW * wm + r + tap) * LDP + 8 * h
                             : Atile + (32 * MBW * wm + r) * AP + slice * BK + 8 * h;
      constexpr int XP = XS ? LDP : AP;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8_t af[NBW], bfm[MBW];
#pragma unroll
        for (int bn = 0; bn < NBW; ++bn) af[bn] = *reinterpret_cast<const bf16x8_t*>(wsb + bn * 32 * LDP + ks * 16);
#pragma unroll
        for (int bm = 0; bm < MBW; ++bm) bfm[bm] = *reinterpret_cast<const bf16x8_t*>(xsb + bm * 32 * XP + ks * 16);
#pragma unroll
        for (int bn = 0; bn < NBW; ++bn)
#pragma unroll
          for (int bm = 0; bm < MBW; ++bm)
            acc[bn][bm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[bn], bfm[bm], acc[bn][bm], 0, 0, 0);
      }
      store_w((it + 1) & 1);
      if (XS && tap == taps - 1) store_x((slice + 1) & 1);
      __syncthreads();
    }
  }
}

// accumulators -> fp32 LDS tile [BM][EP] (a lane's 16 values are 4 groups of 4 consecutive columns of one row)
template <int NW, int NBW, int MBW>
__device__ __forceinline__ void acc_to_lds(float* es, int EP, const f32x16_t (&acc)[NBW][MBW])
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave % NW, wm = wave / NW, r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int bm = 0; bm < MBW; ++bm)
#pragma unroll
    for (int bn = 0; bn < NBW; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(&es[(32 * MBW * wm + 32 * bm + r) * EP + 32 * NBW * wn + 32 * bn + 8 * g + 4 * h]) =
            make_float4(acc[bn][bm][4 * g], acc[bn][bm][4 * g + 1], acc[bn][bm][4 * g + 2], acc[bn][bm][4 * g + 3]);
}

// LDS plan (bytes): stage-1 operands | epilogue tiles alias them once the K loop has drained
constexpr int S1F_BYTES = 2 * (2 * H * LDP + XROWS * LDP) * 2;              // fwd stage 1: W [2][384][72] + X [2][68][72]
constexpr int S1B_BYTES = 2 * (H * LDP + XROWS * LDP) * 2;                  // bwd stage 1: W [2][192][72] + X [2][68][72]
constexpr int EP1 = 2 * H + 4, EP2 = H + 4;
constexpr int ES1_BYTES = BM * EP1 * 4;                                     // [64][388] fp32
constexpr int ES2_BYTES = BM * EP2 * 4;                                     // [64][196] fp32
constexpr int AT_BYTES = BM * AP * 2;                                       // stage-2 activation tile bf16
constexpr int W2_BYTES = 2 * H * LDP * 2;                                   // stage-2 weights [2][192][72]
constexpr int FWD_LDS = (S1F_BYTES > ES1_BYTES + AT_BYTES ? S1F_BYTES : ES1_BYTES + AT_BYTES);
constexpr int BWD_LDS = (S1B_BYTES > ES2_BYTES + AT_BYTES + 2 * 0 ? S1B_BYTES : ES2_BYTES + AT_BYTES) + W2_BYTES;

__device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8])
{
  v[0] = bf2f(u.x & 0xffff); v[1] = bf2f(u.x >> 16); v[2] = bf2f(u.y & 0xffff); v[3] = bf2f(u.y >> 16);
  v[4] = bf2f(u.z & 0xffff); v[5] = bf2f(u.z >> 16); v[6] = bf2f(u.w & 0xffff); v[7] = bf2f(u.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8])
{
  return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}

// ------------------------------------------------------------------------------------------------ forward
template <bool RES>
__global__ __launch_bounds__(256, 1) void gt_wn_layer_fwd_kernel(WnArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  stamp_begin(a);
  if (a.seed_dev) a.drop_seed ^= *a.seed_dev;
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * BM;
  bf16_t* Ws = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Xs = Ws + 2 * (2 * H) * LDP;
  float* es = reinterpret_cast<float*>(smem);
  bf16_t* At = reinterpret_cast<bf16_t*>(smem + ES1_BYTES);

  {
    f32x16_t acc[3][2];
    gemm_tile<2 * H, 4, 1, 3, 2, true, TAPS>(a.W1, 2 * H, a.K1p, a.X, a.ldx, H, a.R, m0, Ws, Xs, nullptr, acc);
    acc_to_lds<4, 3, 2>(es, EP1, acc);
  }
  __syncthreads();

  // gate epilogue: a thread owns (row, 8 gate channels); packed columns are [32 tanh | 32 sigmoid] per 64
  constexpr int CPR = H / 8;                                       // 24 chunks per row
#pragma unroll
  for (int j = 0; j < BM * CPR / 256; ++j) {
    const int q = tid + 256 * j, row = q / CPR, c = (q - row * CPR) * 8;
    const int m = m0 + row;
    const float* et = &es[row * EP1 + (c >> 5) * 64 + (c & 31)];
    const float4 t0 = *reinterpret_cast<const float4*>(et), t1 = *reinterpret_cast<const float4*>(et + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(et + 32), s1 = *reinterpret_cast<const float4*>(et + 36);
    float pt[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w}, ps[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    float aa[8] = {}, tt[8] = {}, ss[8] = {};
    if (m < a.R) {
      const float* cp = nullptr;
      if (a.cond) cp = a.cond + (size_t)(a.B > 0 ? gt_row_batch(a.row0, a.B, m, a.Tp) : m) * a.ldc + c;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float vt = pt[i] + a.bias1[c + i], vs = ps[i] + a.bias1[H + c + i];
        if (a.drop_thresh) {                                         // x_in = drop(conv(x)) (modules.py:153)
          vt = drop_keep(a.drop_seed, m, c + i, a.drop_thresh) ? vt * a.drop_scale : 0.0f;
          vs = drop_keep(a.drop_seed, m, H + c + i, a.drop_thresh) ? vs * a.drop_scale : 0.0f;
        }
        if (cp) { vt += cp[i]; vs += cp[H + i]; }
        tt[i] = tanhf_(vt); ss[i] = sigmoidf_(vs); aa[i] = tt[i] * ss[i];
      }
      *reinterpret_cast<uint4*>(a.Tt + (size_t)m * a.ldts + c) = pack8(tt);
      *reinterpret_cast<uint4*>(a.Ss + (size_t)m * a.ldts + c) = pack8(ss);
      *reinterpret_cast<uint4*>(a.acts + (size_t)m * a.ldacts + c) = pack8(aa);
    }
    if (RES) *reinterpret_cast<uint4*>(At + row * AP + c) = pack8(aa);
  }
  if (!RES) { stamp_end(a); return; }
  __syncthreads();                                                   // es dead, At complete

  {
    bf16_t* W2s = reinterpret_cast<bf16_t*>(smem);                   // aliases es
    f32x16_t acc2[3][1];
    gemm_tile<H, 2, 2, 3, 1, false, 1>(a.W2, H, a.K2p, nullptr, 0, H, a.R, m0, W2s, nullptr, At, acc2);
    float* es2 = reinterpret_cast<float*>(smem);                     // the K loop's last barrier has retired the W2s reads
    acc_to_lds<2, 3, 1>(es2, EP2, acc2);
  }
  __syncthreads();
  {
    const float* es2 = reinterpret_cast<const float*>(smem);
#pragma unroll
    for (int j = 0; j < BM * CPR / 256; ++j) {
      const int q = tid + 256 * j, row = q / CPR, c = (q - row * CPR) * 8;
      const int m = m0 + row;
      if (m >= a.R) continue;
      const float rm = a.rowmask[m];
      const float4 e0 = *reinterpret_cast<const float4*>(&es2[row * EP2 + c]), e1 = *reinterpret_cast<const float4*>(&es2[row * EP2 + c + 4]);
      float v[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w}, xr[8];
      unpack8(*reinterpret_cast<const uint4*>(a.resid + (size_t)m * a.ldres + c), xr);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (v[i] + a.bias2[c + i] + xr[i]) * rm;
      *reinterpret_cast<uint4*>(a.Xnext + (size_t)m * a.ldxn + c) = pack8(v);
    }
  }
  stamp_end(a);
}

// ------------------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(256, 1) void gt_wn_layer_bwd_kernel(WnArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  stamp_begin(a);
  if (a.seed_dev) a.drop_seed ^= *a.seed_dev;
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * BM;
  // stage-2 weights have their own region at the end, so that their first slice can be fetched under stage 1's tail
  bf16_t* Ws = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Xs = Ws + 2 * H * LDP;
  float* es = reinterpret_cast<float*>(smem);
  bf16_t* At = reinterpret_cast<bf16_t*>(smem + ES2_BYTES);
  bf16_t* W2s = reinterpret_cast<bf16_t*>(smem + (BWD_LDS - W2_BYTES));
  constexpr int CPR = H / 8;

  {
    f32x16_t acc[3][1];
    gemm_tile<H, 2, 2, 3, 1, true, TAPS>(a.W1, H, a.K1p, a.X, a.ldx, 2 * H, a.R, m0, Ws, Xs, nullptr, acc);
    acc_to_lds<2, 3, 1>(es, EP2, acc);
  }
  __syncthreads();
  // dX = (conv^T(d pre) + residual-path gradient) * mask -> HBM and the stage-2 tile
#pragma unroll
  for (int j = 0; j < BM * CPR / 256; ++j) {
    const int q = tid + 256 * j, row = q / CPR, c = (q - row * CPR) * 8;
    const int m = m0 + row;
    float v[8] = {};
    if (m < a.R) {
      const float rm = a.rowmask[m];
      const float4 e0 = *reinterpret_cast<const float4*>(&es[row * EP2 + c]), e1 = *reinterpret_cast<const float4*>(&es[row * EP2 + c + 4]);
      float ad[8] = {};
      if (a.resid) unpack8(*reinterpret_cast<const uint4*>(a.resid + (size_t)m * a.ldres + c), ad);
      const float ev[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (ev[i] + ad[i]) * rm;
      *reinterpret_cast<uint4*>(a.Xnext + (size_t)m * a.ldxn + c) = pack8(v);
    }
    *reinterpret_cast<uint4*>(At + row * AP + c) = pack8(v);
  }
  __syncthreads();

  {
    f32x16_t acc2[3][1];
    gemm_tile<H, 2, 2, 3, 1, false, 1>(a.W2, H, a.K2p, nullptr, 0, H, a.R, m0, W2s, nullptr, At, acc2);
    __syncthreads();
    acc_to_lds<2, 3, 1>(es, EP2, acc2);                              // At no longer needed: es may overlap nothing live
  }
  __syncthreads();
  // d acts = dX W_res + skip-path gradient; d pre_t = d S (1 - T^2), d pre_s = d T S (1 - S), times the forward's dropout mask
#pragma unroll
  for (int j = 0; j < BM * CPR / 256; ++j) {
    const int q = tid + 256 * j, row = q / CPR, c = (q - row * CPR) * 8;
    const int m = m0 + row;
    if (m >= a.R) continue;
    const float4 e0 = *reinterpret_cast<const float4*>(&es[row * EP2 + c]), e1 = *reinterpret_cast<const float4*>(&es[row * EP2 + c + 4]);
    float d[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w}, vs[8], t[8], s[8], gt[8], gs[8];
    unpack8(*reinterpret_cast<const uint4*>(a.viaskip + (size_t)m * a.ldvs + c), vs);
    unpack8(*reinterpret_cast<const uint4*>(a.Tt + (size_t)m * a.ldts + c), t);
    unpack8(*reinterpret_cast<const uint4*>(a.Ss + (size_t)m * a.ldts + c), s);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // round 1 formed d acts as a bf16 GEMM output before the gate backward; keep fp32 here (strictly more accurate)
      const float dd = d[i] + vs[i];
      gt[i] = dd * s[i] * (1.0f - t[i] * t[i]); gs[i] = dd * t[i] * s[i] * (1.0f - s[i]);
    }
    if (a.dpre_c) {                                                  // cond enters after the dropout (modules.py:153-156)
      bf16_t* cp = a.dpre_c + (size_t)m * a.lddp + c;
      *reinterpret_cast<uint4*>(cp) = pack8(gt);
      *reinterpret_cast<uint4*>(cp + H) = pack8(gs);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (a.drop_thresh) {
        gt[i] = drop_keep(a.drop_seed, m, c + i, a.drop_thresh) ? gt[i] * a.drop_scale : 0.0f;
        gs[i] = drop_keep(a.drop_seed, m, H + c + i, a.drop_thresh) ? gs[i] * a.drop_scale : 0.0f;
      }
    }
    bf16_t* yp = a.dpre + (size_t)m * a.lddp + c;
    *reinterpret_cast<uint4*>(yp) = pack8(gt);
    *reinterpret_cast<uint4*>(yp + H) = pack8(gs);
  }
  stamp_end(a);
}

int fill_drop(WnArgs& a, float drop_p, uint32_t seed, const uint32_t* seed_dev)
{
  a.drop_thresh = 0; a.drop_seed = seed; a.drop_scale = 1.0f; a.seed_dev = seed_dev;
  if (drop_p > 0.0f) {
    if (drop_p >= 1.0f) return GT_E_UNSUPPORTED;
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.drop_scale = 1.0f / (1.0f - drop_p);
  }
  return GT_OK;
}
inline bool al16(const void* p) { return !((uintptr_t)p & 15); }

}  // namespace

extern "C" int gt_wn_layer_fwd(const void* x, int ldx, const void* w_in, int K1p, const float* bias_in,
                               const float* cond, int ldc, const int32_t* row0, int B, int Tp, const float* rowmask,
                               void* acts, int ldacts, void* gate_t, void* gate_s, int ldts,
                               const void* w_res, int K2p, const float* bias_res, void* x_next, int ldxn,
                               int R, int Hc, int taps, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev,
                               unsigned long long* stamps, int stamp_slot, void* stream)
{
  if (R < 0) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  if (Hc != H || taps != TAPS) return GT_E_UNSUPPORTED;
  if (!x || !w_in || !bias_in || !rowmask || !acts || !gate_t || !gate_s) return GT_E_INVAL;
  if (w_res && (!bias_res || !x_next)) return GT_E_INVAL;
  if ((ldx & 7) || (ldacts & 7) || (ldts & 7) || (w_res && (ldxn & 7)) || (K1p % BK) || K1p < H || (w_res && ((K2p % BK) || K2p < H))) return GT_E_ALIGN;
  if (!al16(x) || !al16(w_in) || !al16(acts) || !al16(gate_t) || !al16(gate_s) || !al16(w_res) || !al16(x_next)) return GT_E_ALIGN;
  if (cond && (Tp <= 0 || (row0 && B <= 0))) return GT_E_INVAL;
  WnArgs a = {};
  a.X = static_cast<const bf16_t*>(x); a.ldx = ldx; a.W1 = static_cast<const bf16_t*>(w_in); a.W2 = static_cast<const bf16_t*>(w_res);
  a.bias1 = bias_in; a.bias2 = bias_res; a.cond = cond; a.ldc = ldc; a.B = B; a.row0 = row0; a.Tp = Tp > 0 ? Tp : 1; a.rowmask = rowmask;
  a.acts = static_cast<bf16_t*>(acts); a.ldacts = ldacts; a.Tt = static_cast<bf16_t*>(gate_t); a.Ss = static_cast<bf16_t*>(gate_s); a.ldts = ldts;
  a.Xnext = static_cast<bf16_t*>(x_next); a.ldxn = ldxn; a.resid = a.X; a.ldres = ldx;
  a.R = R; a.K1p = K1p; a.K2p = K2p; a.stamps = stamps; a.stamp_slot = stamp_slot;
  const int rc = fill_drop(a, drop_p, drop_seed, seed_dev);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((R + BM - 1) / BM), block(256);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_wn_layer_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_wn_layer_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS) != hipSuccess)
      return GT_E_LAUNCH;
    attr = true;
  }
  if (w_res) hipLaunchKernelGGL(gt_wn_layer_fwd_kernel<true>, grid, block, FWD_LDS, st, a);
  else       hipLaunchKernelGGL(gt_wn_layer_fwd_kernel<false>, grid, block, FWD_LDS, st, a);
  return gt_launch_status(__func__);
}

extern "C" int gt_wn_layer_bwd(const void* dpre_next, int lddn, const void* w_in_dgrad, int K1p, const void* resid, int ldres,
                               const float* rowmask, void* dx, int lddx, const void* w_res_dgrad, int K2p,
                               const void* via_skip, int ldvs, const void* gate_t, const void* gate_s, int ldts,
                               void* dpre, void* dpre_c, int lddp, int R, int Hc, int taps, float drop_p, uint32_t drop_seed,
                               const uint32_t* seed_dev, unsigned long long* stamps, int stamp_slot, void* stream)
{
  if (R < 0) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  if (Hc != H || taps != TAPS) return GT_E_UNSUPPORTED;
  if (!dpre_next || !w_in_dgrad || !rowmask || !dx || !w_res_dgrad || !via_skip || !gate_t || !gate_s || !dpre) return GT_E_INVAL;
  if ((lddn & 7) || (ldres & 7) || (lddx & 7) || (ldvs & 7) || (ldts & 7) || (lddp & 7) || (K1p % BK) || K1p < 2 * H || (K2p % BK) || K2p < H) return GT_E_ALIGN;
  if (!al16(dpre_next) || !al16(w_in_dgrad) || !al16(resid) || !al16(dx) || !al16(w_res_dgrad) || !al16(via_skip) || !al16(gate_t) || !al16(gate_s) || !al16(dpre) || !al16(dpre_c))
    return GT_E_ALIGN;
  WnArgs a = {};
  a.X = static_cast<const bf16_t*>(dpre_next); a.ldx = lddn; a.W1 = static_cast<const bf16_t*>(w_in_dgrad); a.W2 = static_cast<const bf16_t*>(w_res_dgrad);
  a.rowmask = rowmask; a.Xnext = static_cast<bf16_t*>(dx); a.ldxn = lddx; a.resid = static_cast<const bf16_t*>(resid); a.ldres = ldres;
  a.viaskip = static_cast<const bf16_t*>(via_skip); a.ldvs = ldvs;
  a.Tt = const_cast<bf16_t*>(static_cast<const bf16_t*>(gate_t)); a.Ss = const_cast<bf16_t*>(static_cast<const bf16_t*>(gate_s)); a.ldts = ldts;
  a.dpre = static_cast<bf16_t*>(dpre); a.dpre_c = static_cast<bf16_t*>(dpre_c); a.lddp = lddp; a.R = R; a.K1p = K1p; a.K2p = K2p; a.stamps = stamps; a.stamp_slot = stamp_slot;
  const int rc = fill_drop(a, drop_p, drop_seed, seed_dev);
  if (rc) return rc;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_wn_layer_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS) != hipSuccess) return GT_E_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(gt_wn_layer_bwd_kernel, dim3((R + BM - 1) / BM), dim3(256), BWD_LDS, static_cast<hipStream_t>(stream), a);
  return gt_launch_status(__func__);
}
