// One WaveNet layer of the flow decoder as ONE kernel, forward and backward, gfx950 (bf16 MFMA v_mfma_f32_32x32x16_bf16).
//
// reference modules.WN.forward, one iteration of the loop (modules.py:151-170):
//     x_in = drop(in_layer_i(x));  acts = tanh(x_in[:H] + g[:H]) * sigmoid(x_in[H:] + g[H:]);
//     rs = res_skip_i(acts);  x = (x + rs[:H]) * mask;  output += rs[H:]
// Round 1 ran this as a k=5 gate conv kernel followed by a 1x1 GEMM kernel per layer (and the mirror pair in the
// backward); every launch on the decoder's dependent chain cost ~10 us beyond its arithmetic.  Here a workgroup owns
// 64 rows and ALL channels of them, so the second contraction runs on the tile the first one just produced:
//
//   forward   stage 1: x_in = conv_k5(x)           N = 2H = 384, K = 5 * 192      4 waves x (64 rows x 96 columns)
//             epilogue (in registers): bias, dropout, cond, gate -> T, S, acts (HBM, for the backward); acts -> LDS (bf16)
//             stage 2: res = acts @ W_res^T         N = 192, K = 192               2 x 2 waves x (32 rows x 96 columns)
//             epilogue (in registers): x_next = (x + res + b_res) * mask
//   backward  stage 1: dX = conv_k5^T(d pre_{i+1}) N = 192, K = 5 * 384           2 (columns) x 2 (K halves) waves, partial
//                                                                                  sums exchanged through LDS
//             epilogue: dX = (dX + residual-path gradient) * mask -> HBM (the res conv's weight gradient reads it) and LDS
//             stage 2: d acts_i = dX @ W_res_i      N = 192, K = 192
//             epilogue: + skip-path gradient, gate backward with the forward's dropout replayed -> d pre_i [R, 2H]
//
// Operand paths.  MFMA A = weights: the packed images are in MFMA-FRAGMENT order (gt_pack_conv_weights flags 2 / 4: one
// contiguous 1-KB fragment per (32 output channels, 16 k)), so a wave fetches exactly the fragments of its own columns
// straight from L2 into registers with one fully coalesced 16-byte load per lane — the weights never pass through LDS,
// no workgroup barrier is spent on them, and they are double-buffered one (slice, tap) step ahead in registers.  MFMA B =
// activations: 68-row x 64-channel slices staged in LDS (double buffer, ONE barrier per K slice of 5 taps).  With this
// split a 64-row tile moves each weight byte once per workgroup through the vector-memory path (737 KB per workgroup,
// the same 64 B/clk/CU budget as its 47 MFLOP of MFMA issue), and LDS only carries the small activation operand.
// The gate's packed column order is [16 tanh | 16 sigmoid] per 32-column MFMA block (pack flag 16): both halves of a
// gate channel sit in the same lane's accumulators, so bias / dropout / cond / tanh * sigmoid run in registers.
//
// The skip half of res_skip stays what round 1 made it: ONE K = n*H GEMM per WaveNet over the layers' gated activations.
#include "common.h"
#include "../../include/glowtts_hip.h"

namespace {

constexpr int H = 192;                        // hidden channels (configs/*.json hidden_channels_dec)
constexpr int BM = 64;                        // rows per workgroup
constexpr int BK = 64;                        // K slice (activation channels per LDS stage)
constexpr int LDP = 72;                       // halfs per LDS activation row (64 + 8: conflict-free ds_read_b128 over 16 rows)
constexpr int TAPS = 5;
constexpr int XROWS = BM + TAPS - 1;          // 68 activation rows per slice
constexpr int XCH = (XROWS * 8 + 255) / 256;  // 16-byte activation chunks per thread per slice
constexpr int AP = H + 8;                     // pitch (halfs) of the stage-2 activation tile
#ifndef WN_RING
#define WN_RING 3
#endif
constexpr int RING = WN_RING;                 // weight-fragment register ring: (slice, tap) steps in flight per wave

struct WnArgs {
  const bf16_t* X; int ldx;                   // stage-1 operand rows (fwd: layer input [R, H]; bwd: d pre of the next layer [R, 2H])
  const bf16_t* W1;                           // stage-1 weights, fragment order [5][N/32][K/16][64 lanes][8]
  const bf16_t* W2;                           // stage-2 weights, fragment order [H/32][H/16][64][8] (NULL: no second stage)
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
  int R;
  uint32_t drop_thresh, drop_seed; float drop_scale; const uint32_t* seed_dev;
  // live timing of the dominant kernel inside a captured graph (bench.py): stamps[2*slot] = min start, [2*slot+1] = max end
  unsigned long long* stamps; int stamp_slot; const int32_t* stamp_base;     // slot = stamp_slot + *stamp_base (a per-step device counter)
};

// workgroup 0's start (kept in a register, stored at the end) and an atomicMax of every workgroup's end, issued after its last wait:
// an atomic at the kernel's start sits in front of every later wait for a load (vector-memory operations retire in order)
__device__ __forceinline__ unsigned long long stamp_begin(const WnArgs& a)
{
  return (a.stamps && threadIdx.x == 0 && blockIdx.x == 0) ? (unsigned long long)wall_clock64() : 0ull;
}
__device__ __forceinline__ void stamp_end(const WnArgs& a, unsigned long long t_begin)
{
  if (a.stamps && threadIdx.x == 0) {
    unsigned long long* slot = a.stamps + 2 * (a.stamp_slot + (a.stamp_base ? *a.stamp_base : 0));
    if (blockIdx.x == 0) slot[0] = t_begin;
    atomicMax(slot + 1, (unsigned long long)wall_clock64());
  }
}

__device__ __forceinline__ uint4 ldfrag(const bf16_t* __restrict__ W, int f, int lane)
{
  return *reinterpret_cast<const uint4*>(W + ((size_t)f * 64 + lane) * 8);
}
__device__ __forceinline__ bf16x8_t asfrag(const uint4& u) { return __builtin_bit_cast(bf16x8_t, u); }
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack2bf(a, b), pack2bf(c, d)); }
__device__ __forceinline__ void unpack4(const uint2& u, float (&v)[4])
{
  v[0] = bf2f(u.x & 0xffff); v[1] = bf2f(u.x >> 16); v[2] = bf2f(u.y & 0xffff); v[3] = bf2f(u.y >> 16);
}

// activation slice `slice` (64 channels) of rows m0 - 2 .. m0 + 65 -> registers -> LDS
__device__ __forceinline__ void x_load(const bf16_t* __restrict__ X, int ldx, int Kx, int R, int m0, int slice, uint4 (&xr)[XCH])
{
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const int chunk = threadIdx.x + 256 * i, row = chunk >> 3, c8 = chunk & 7;
    int gm = m0 - (TAPS >> 1) + (row < XROWS ? row : XROWS - 1);
    gm = gm < 0 ? 0 : (gm >= R ? R - 1 : gm);
    const int ch = slice * BK + c8 * 8;
    xr[i] = make_uint4(0, 0, 0, 0);
    if (ch < Kx) xr[i] = *reinterpret_cast<const uint4*>(X + (size_t)gm * ldx + ch);
  }
}
__device__ __forceinline__ void x_store(bf16_t* Xs, const uint4 (&xr)[XCH])
{
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const int chunk = threadIdx.x + 256 * i, row = chunk >> 3, c8 = chunk & 7;
    if (row < XROWS) *reinterpret_cast<uint4*>(Xs + row * LDP + c8 * 8) = xr[i];
  }
}

// 1x1 stage on the LDS tile At [64][AP] (K = H): this wave's 32 rows (wm) x 96 columns (wn); weights W2 in fragment order.
// The first half of the K steps is fetched by `s2_prefetch` (before the caller's epilogue, so the loads fly under it).
constexpr int KK2 = H / 16;                   // 12 k-steps
__device__ __forceinline__ void s2_prefetch(const bf16_t* __restrict__ W2, int wn, int lane, uint4 (&ring)[KK2 / 2][3])
{
#pragma unroll
  for (int kk = 0; kk < KK2 / 2; ++kk)
#pragma unroll
    for (int bn = 0; bn < 3; ++bn) ring[kk][bn] = ldfrag(W2, (3 * wn + bn) * KK2 + kk, lane);
}
__device__ __forceinline__ void s2_gemm(const bf16_t* __restrict__ W2, const bf16_t* At, int wn, int wm, int lane,
                                        uint4 (&ring)[KK2 / 2][3], f32x16_t (&acc)[3])
{
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int bn = 0; bn < 3; ++bn)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[bn][e] = 0.0f;
  const bf16_t* ab = At + (32 * wm + r) * AP + 8 * h;
#pragma unroll
  for (int kk = 0; kk < KK2; ++kk) {
    const bf16x8_t bfm = *reinterpret_cast<const bf16x8_t*>(ab + kk * 16);
#pragma unroll
    for (int bn = 0; bn < 3; ++bn) {
      acc[bn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[kk % (KK2 / 2)][bn]), bfm, acc[bn], 0, 0, 0);
      if (kk + KK2 / 2 < KK2) ring[kk % (KK2 / 2)][bn] = ldfrag(W2, (3 * wn + bn) * KK2 + kk + KK2 / 2, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward
constexpr int FWD_LDS = 2 * XROWS * LDP * 2 + 3 * BM * AP * 2;              // X [2][68][72] + At, Tl, Sl [64][200] = 96 384 B

template <bool RES>
__global__ __launch_bounds__(256) void gt_wn_layer_fwd_kernel(WnArgs a)
{
  __shared__ __attribute__((aligned(16))) unsigned char smem[FWD_LDS];
  const unsigned long long t_begin_ = stamp_begin(a);
  if (a.seed_dev) a.drop_seed ^= *a.seed_dev;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * BM;
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* At = Xs + 2 * XROWS * LDP;
  bf16_t* Tl = At + BM * AP;                                         // saved tanh / sigmoid tiles on their way out
  bf16_t* Sl = Tl + BM * AP;
  constexpr int NS = H / BK, NIT = NS * TAPS, KS = H / 16, NBT = 2 * H / 32;  // 3 slices, 15 steps, 12 k-steps per tap, 12 column blocks

  f32x16_t acc[3][2];
#pragma unroll
  for (int bn = 0; bn < 3; ++bn)
#pragma unroll
    for (int bm = 0; bm < 2; ++bm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[bn][bm][e] = 0.0f;

  // weight fragments of step `it` (slice, tap): 4 k-steps x this wave's 3 column blocks
  uint4 ring[RING][4][3];
  auto w_load = [&](int it, uint4 (&dst)[4][3]) {
#ifdef WN_EXP_NOLOAD
    if (it > RING) return;
#endif
    const int slice = it / TAPS, tap = it - slice * TAPS;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) dst[ks][bn] = ldfrag(a.W1, (tap * NBT + 3 * wave + bn) * KS + slice * 4 + ks, lane);
  };
  uint4 xr[XCH];
  x_load(a.X, a.ldx, H, a.R, m0, 0, xr);
#pragma unroll
  for (int p = 0; p < RING - 1; ++p) w_load(p < NIT ? p : NIT - 1, ring[p]);
  x_store(Xs, xr);
  __syncthreads();

#pragma unroll
  for (int slice = 0; slice < NS; ++slice) {
    x_load(a.X, a.ldx, H, a.R, m0, slice + 1 < NS ? slice + 1 : NS - 1, xr);       // unconditional: no branch-defined staging registers
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int it = slice * TAPS + tap;
      w_load(it + RING - 1 < NIT ? it + RING - 1 : NIT - 1, ring[(it + RING - 1) % RING]);
      __builtin_amdgcn_sched_barrier(0);       // keep the prefetch RING - 1 steps ahead: the scheduler would sink it next to its use
      const bf16_t* xsb = Xs + (slice & 1) * XROWS * LDP + (r + tap) * LDP + 8 * h;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(xsb + ks * 16);
        const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(xsb + 32 * LDP + ks * 16);
#pragma unroll
        for (int bn = 0; bn < 3; ++bn) {
#ifdef WN_EXP_NOMFMA
          acc[bn][0][0] += __uint_as_float(ring[it % RING][ks][bn].x) + b0[0]; acc[bn][1][1] += __uint_as_float(ring[it % RING][ks][bn].w) + b1[1];
#else
          acc[bn][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RING][ks][bn]), b0, acc[bn][0], 0, 0, 0);
          acc[bn][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RING][ks][bn]), b1, acc[bn][1], 0, 0, 0);
#endif
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    x_store(Xs + ((slice + 1) & 1) * XROWS * LDP, xr);
    __syncthreads();
  }

  // second-stage weights start flying now, under the gate epilogue
  const int wn2 = wave & 1, wm2 = wave >> 1;
  uint4 ring2[KK2 / 2][3];
  if (RES) s2_prefetch(a.W2, wn2, lane, ring2);

  // gate epilogue in registers: block (3*wave + bn) holds [16 tanh | 16 sigmoid] of gate channels 16*(3*wave+bn) .. +15;
  // a lane's accumulator e = 4g + j is row r, packed column 8g + 4h + j -> tanh for g < 2, sigmoid (same channel) at e + 8
#pragma unroll
  for (int bn = 0; bn < 3; ++bn)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int c = 16 * (3 * wave + bn) + 8 * g + 4 * h;             // first of 4 consecutive gate channels
      const float4 bt = *reinterpret_cast<const float4*>(a.bias1 + c), bs = *reinterpret_cast<const float4*>(a.bias1 + H + c);
      const float btv[4] = {bt.x, bt.y, bt.z, bt.w}, bsv[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
      for (int bm = 0; bm < 2; ++bm) {
        const int row = 32 * bm + r, m = m0 + row;
        float tt[4] = {}, ss[4] = {}, aa[4] = {};
        if (m < a.R) {
          float ctv[4] = {}, csv[4] = {};
          if (a.cond) {
            const float* cp = a.cond + (size_t)(a.B > 0 ? gt_row_batch(a.row0, a.B, m, a.Tp) : m) * a.ldc + c;
            const float4 ct = *reinterpret_cast<const float4*>(cp), cs = *reinterpret_cast<const float4*>(cp + H);
            ctv[0] = ct.x; ctv[1] = ct.y; ctv[2] = ct.z; ctv[3] = ct.w; csv[0] = cs.x; csv[1] = cs.y; csv[2] = cs.z; csv[3] = cs.w;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float vt = acc[bn][bm][4 * g + j] + btv[j], vs = acc[bn][bm][4 * g + j + 8] + bsv[j];
            if (a.drop_thresh) {                                       // x_in = drop(conv(x)) (modules.py:153)
              bool kt, ks;
              drop_keep_gate(a.drop_seed, m, c + j, drop_thresh16(a.drop_thresh), kt, ks);
              vt = kt ? vt * a.drop_scale : 0.0f;
              vs = ks ? vs * a.drop_scale : 0.0f;
            }
            vt += ctv[j]; vs += csv[j];
#ifdef WN_EXP_NOEPI
            tt[j] = vt; ss[j] = vs; aa[j] = vt + vs;
#else
            tt[j] = tanhf_(vt); ss[j] = sigmoidf_(vs); aa[j] = tt[j] * ss[j];
#endif
          }
        }
        // T, S and acts leave as whole rows through LDS tiles (wn_stack.hip, DESIGN 4.9: the accumulator layout gives a store
        // instruction 16 bytes in each of 32 rows)
        *reinterpret_cast<uint2*>(Tl + row * AP + c) = pack4(tt[0], tt[1], tt[2], tt[3]);
        *reinterpret_cast<uint2*>(Sl + row * AP + c) = pack4(ss[0], ss[1], ss[2], ss[3]);
        *reinterpret_cast<uint2*>(At + row * AP + c) = pack4(aa[0], aa[1], aa[2], aa[3]);
      }
    }
  __syncthreads();                                                   // At, Tl, Sl complete
  {
    constexpr int CPR = H / 8;
#pragma unroll
    for (int i = 0; i < BM * CPR / 256; ++i) {
      const int idx = threadIdx.x + 256 * i, row = idx / CPR, c8 = idx - row * CPR, m = m0 + row;
      if (m < a.R) {
        *reinterpret_cast<uint4*>(a.Tt + (size_t)m * a.ldts + c8 * 8) = *reinterpret_cast<const uint4*>(Tl + row * AP + c8 * 8);
        *reinterpret_cast<uint4*>(a.Ss + (size_t)m * a.ldts + c8 * 8) = *reinterpret_cast<const uint4*>(Sl + row * AP + c8 * 8);
        *reinterpret_cast<uint4*>(a.acts + (size_t)m * a.ldacts + c8 * 8) = *reinterpret_cast<const uint4*>(At + row * AP + c8 * 8);
      }
    }
  }
  if (!RES) { stamp_end(a, t_begin_); return; }

  f32x16_t acc2[3];
  s2_gemm(a.W2, At, wn2, wm2, lane, ring2, acc2);
  // x_next = (x + res + b_res) * mask: accumulator e = 4g + j is row r, channel 32*(3*wn2 + bn) + 8g + 4h + j
  {
    const int m = m0 + 32 * wm2 + r;
    if (m < a.R) {
      const float rm = a.rowmask[m];
#pragma unroll
      for (int bn = 0; bn < 3; ++bn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 32 * (3 * wn2 + bn) + 8 * g + 4 * h;
          const float4 b2 = *reinterpret_cast<const float4*>(a.bias2 + n);
          float xv[4];
          unpack4(*reinterpret_cast<const uint2*>(a.resid + (size_t)m * a.ldres + n), xv);
          *reinterpret_cast<uint2*>(a.Xnext + (size_t)m * a.ldxn + n) =
              pack4((acc2[bn][4 * g] + b2.x + xv[0]) * rm, (acc2[bn][4 * g + 1] + b2.y + xv[1]) * rm,
                    (acc2[bn][4 * g + 2] + b2.z + xv[2]) * rm, (acc2[bn][4 * g + 3] + b2.w + xv[3]) * rm);
        }
    }
  }
  stamp_end(a, t_begin_);
}

// ------------------------------------------------------------------------------------------------ backward
constexpr int EX_BYTES = 4 * 3 * 16 * 64 * 4;                                // partial-sum exchange: [4 waves][3 blocks][16][64 lanes] fp32
constexpr int BWD_XS = 2 * XROWS * LDP * 2;
constexpr int BWD_LDS = (EX_BYTES > BWD_XS ? EX_BYTES : BWD_XS) + BM * AP * 2;   // Ex aliases the activation slices; + At

template <bool S2>
__global__ __launch_bounds__(256) void gt_wn_layer_bwd_kernel(WnArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned long long t_begin_ = stamp_begin(a);
  if (a.seed_dev) a.drop_seed ^= *a.seed_dev;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wn = wave & 1, wk = wave >> 1;     // stage 1: column half, K half;  afterwards wk doubles as the row half
  const int m0 = blockIdx.x * BM;
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);
  float* Ex = reinterpret_cast<float*>(smem);
  bf16_t* At = reinterpret_cast<bf16_t*>(smem + (EX_BYTES > BWD_XS ? EX_BYTES : BWD_XS));
  constexpr int NS = 2 * H / BK, NIT = NS * TAPS, KS = 2 * H / 16, NBT = H / 32;  // 6 slices, 30 steps, 24 k-steps per tap, 6 column blocks

  f32x16_t acc[3][2];
#pragma unroll
  for (int bn = 0; bn < 3; ++bn)
#pragma unroll
    for (int bm = 0; bm < 2; ++bm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[bn][bm][e] = 0.0f;

  // this wave's K half of step `it`: k-steps 2*wk, 2*wk + 1 of the slice, 3 column blocks
  constexpr int RB = 2 * RING - 1;             // half the fragments per step: twice the depth for the same registers
  uint4 ring[RB][2][3];
  auto w_load = [&](int it, uint4 (&dst)[2][3]) {
    const int slice = it / TAPS, tap = it - slice * TAPS;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) dst[k2][bn] = ldfrag(a.W1, (tap * NBT + 3 * wn + bn) * KS + slice * 4 + 2 * wk + k2, lane);
  };
  uint4 xr[XCH];
  x_load(a.X, a.ldx, 2 * H, a.R, m0, 0, xr);
#pragma unroll
  for (int p = 0; p < RB - 1; ++p) w_load(p < NIT ? p : NIT - 1, ring[p]);
  x_store(Xs, xr);
  __syncthreads();

#pragma unroll
  for (int slice = 0; slice < NS; ++slice) {
    x_load(a.X, a.ldx, 2 * H, a.R, m0, slice + 1 < NS ? slice + 1 : NS - 1, xr);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int it = slice * TAPS + tap;
      w_load(it + RB - 1 < NIT ? it + RB - 1 : NIT - 1, ring[(it + RB - 1) % RB]);
      __builtin_amdgcn_sched_barrier(0);
      const bf16_t* xsb = Xs + (slice & 1) * XROWS * LDP + (r + tap) * LDP + 8 * h + 32 * wk;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(xsb + k2 * 16);
        const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(xsb + 32 * LDP + k2 * 16);
#pragma unroll
        for (int bn = 0; bn < 3; ++bn) {
          acc[bn][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RB][k2][bn]), b0, acc[bn][0], 0, 0, 0);
          acc[bn][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RB][k2][bn]), b1, acc[bn][1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    x_store(Xs + ((slice + 1) & 1) * XROWS * LDP, xr);
    __syncthreads();
  }

  // second-stage weights: this wave's 96 columns (wn), rows 32*wk
  uint4 ring2[KK2 / 2][3];
  if (S2) s2_prefetch(a.W2, wn, lane, ring2);

  // K halves meet: a wave keeps row block bm == wk and hands the other one to its partner (same columns, other K half)
  {
    float* mine = Ex + wave * (3 * 16 * 64);
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int e = 0; e < 16; ++e) mine[(bn * 16 + e) * 64 + lane] = wk ? acc[bn][0][e] : acc[bn][1][e];
  }
  __syncthreads();
  f32x16_t sum[3];
  {
    const float* theirs = Ex + (wave ^ 2) * (3 * 16 * 64);
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum[bn][e] = (wk ? acc[bn][1][e] : acc[bn][0][e]) + theirs[(bn * 16 + e) * 64 + lane];
  }
  // dX = (conv^T(d pre) + residual-path gradient) * mask -> HBM and the stage-2 tile
  const int row = 32 * wk + r, m = m0 + row;
  {
    const float rm = m < a.R ? a.rowmask[m] : 0.0f;
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (3 * wn + bn) + 8 * g + 4 * h;
        float ad[4] = {};
        if (a.resid && m < a.R) unpack4(*reinterpret_cast<const uint2*>(a.resid + (size_t)m * a.ldres + n), ad);
        const uint2 v = pack4((sum[bn][4 * g] + ad[0]) * rm, (sum[bn][4 * g + 1] + ad[1]) * rm,
                              (sum[bn][4 * g + 2] + ad[2]) * rm, (sum[bn][4 * g + 3] + ad[3]) * rm);
        if (m < a.R) *reinterpret_cast<uint2*>(a.Xnext + (size_t)m * a.ldxn + n) = v;
        if (S2) *reinterpret_cast<uint2*>(At + row * AP + n) = v;
      }
  }
  if (!S2) { stamp_end(a, t_begin_); return; }                                 // bottom layer: only the data gradient (no gate below it)
  __syncthreads();

  f32x16_t acc2[3];
  s2_gemm(a.W2, At, wn, wk, lane, ring2, acc2);
  // d acts = dX W_res + skip-path gradient; d pre_t = d S (1 - T^2), d pre_s = d T S (1 - S), times the forward's dropout mask
  if (m < a.R) {
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (3 * wn + bn) + 8 * g + 4 * h;
        float vs[4], t[4], s[4], gt[4], gs[4];
        unpack4(*reinterpret_cast<const uint2*>(a.viaskip + (size_t)m * a.ldvs + n), vs);
        unpack4(*reinterpret_cast<const uint2*>(a.Tt + (size_t)m * a.ldts + n), t);
        unpack4(*reinterpret_cast<const uint2*>(a.Ss + (size_t)m * a.ldts + n), s);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // round 1 formed d acts as a bf16 GEMM output before the gate backward; it stays fp32 here
          const float dd = acc2[bn][4 * g + j] + vs[j];
          gt[j] = dd * s[j] * (1.0f - t[j] * t[j]); gs[j] = dd * t[j] * s[j] * (1.0f - s[j]);
        }
        if (a.dpre_c) {                                              // cond enters after the dropout (modules.py:153-156)
          bf16_t* cp = a.dpre_c + (size_t)m * a.lddp + n;
          *reinterpret_cast<uint2*>(cp) = pack4(gt[0], gt[1], gt[2], gt[3]);
          *reinterpret_cast<uint2*>(cp + H) = pack4(gs[0], gs[1], gs[2], gs[3]);
        }
        if (a.drop_thresh) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bool kt, ks;
            drop_keep_gate(a.drop_seed, m, n + j, drop_thresh16(a.drop_thresh), kt, ks);
            gt[j] = kt ? gt[j] * a.drop_scale : 0.0f;
            gs[j] = ks ? gs[j] * a.drop_scale : 0.0f;
          }
        }
        bf16_t* yp = a.dpre + (size_t)m * a.lddp + n;
        *reinterpret_cast<uint2*>(yp) = pack4(gt[0], gt[1], gt[2], gt[3]);
        *reinterpret_cast<uint2*>(yp + H) = pack4(gs[0], gs[1], gs[2], gs[3]);
      }
  }
  stamp_end(a, t_begin_);
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

extern "C" int gt_wn_layer_fwd(const void* x, int ldx, const void* w_in_frag, const float* bias_in,
                               const float* cond, int ldc, const int32_t* row0, int B, int Tp, const float* rowmask,
                               void* acts, int ldacts, void* gate_t, void* gate_s, int ldts,
                               const void* w_res_frag, const float* bias_res, void* x_next, int ldxn,
                               int R, int Hc, int taps, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev,
                               unsigned long long* stamps, int stamp_slot, const int32_t* stamp_base, void* stream)
{
  if (R < 0) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  if (Hc != H || taps != TAPS) return GT_E_UNSUPPORTED;
  if (!x || !w_in_frag || !bias_in || !rowmask || !acts || !gate_t || !gate_s) return GT_E_INVAL;
  if (w_res_frag && (!bias_res || !x_next)) return GT_E_INVAL;
  if ((ldx & 7) || (ldacts & 7) || (ldts & 7) || (w_res_frag && (ldxn & 3)) || (cond && (ldc & 3))) return GT_E_ALIGN;
  if (!al16(x) || !al16(w_in_frag) || !al16(acts) || !al16(gate_t) || !al16(gate_s) || !al16(w_res_frag) || !al16(x_next) ||
      !al16(bias_in) || !al16(bias_res) || !al16(cond)) return GT_E_ALIGN;
  if (cond && (Tp <= 0 || (row0 && B <= 0))) return GT_E_INVAL;
  WnArgs a = {};
  a.X = static_cast<const bf16_t*>(x); a.ldx = ldx; a.W1 = static_cast<const bf16_t*>(w_in_frag); a.W2 = static_cast<const bf16_t*>(w_res_frag);
  a.bias1 = bias_in; a.bias2 = bias_res; a.cond = cond; a.ldc = ldc; a.B = B; a.row0 = row0; a.Tp = Tp > 0 ? Tp : 1; a.rowmask = rowmask;
  a.acts = static_cast<bf16_t*>(acts); a.ldacts = ldacts; a.Tt = static_cast<bf16_t*>(gate_t); a.Ss = static_cast<bf16_t*>(gate_s); a.ldts = ldts;
  a.Xnext = static_cast<bf16_t*>(x_next); a.ldxn = ldxn; a.resid = a.X; a.ldres = ldx;
  a.R = R; a.stamps = stamps; a.stamp_slot = stamp_slot; a.stamp_base = stamp_base;
  const int rc = fill_drop(a, drop_p, drop_seed, seed_dev);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((R + BM - 1) / BM), block(256);
  if (w_res_frag) hipLaunchKernelGGL(gt_wn_layer_fwd_kernel<true>, grid, block, 0, st, a);
  else            hipLaunchKernelGGL(gt_wn_layer_fwd_kernel<false>, grid, block, 0, st, a);
  return gt_launch_status(__func__);
}

extern "C" int gt_wn_layer_bwd(const void* dpre_next, int lddn, const void* w_in_dgrad_frag, const void* resid, int ldres,
                               const float* rowmask, void* dx, int lddx, const void* w_res_dgrad_frag,
                               const void* via_skip, int ldvs, const void* gate_t, const void* gate_s, int ldts,
                               void* dpre, void* dpre_c, int lddp, int R, int Hc, int taps, float drop_p, uint32_t drop_seed,
                               const uint32_t* seed_dev, unsigned long long* stamps, int stamp_slot, const int32_t* stamp_base, void* stream)
{
  if (R < 0) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  if (Hc != H || taps != TAPS) return GT_E_UNSUPPORTED;
  if (!dpre_next || !w_in_dgrad_frag || !rowmask || !dx) return GT_E_INVAL;
  if (w_res_dgrad_frag && (!via_skip || !gate_t || !gate_s || !dpre)) return GT_E_INVAL;
  if ((lddn & 7) || (ldres & 3) || (lddx & 3) || (ldvs & 3) || (ldts & 3) || (lddp & 3)) return GT_E_ALIGN;
  if (!al16(dpre_next) || !al16(w_in_dgrad_frag) || !al16(resid) || !al16(dx) || !al16(w_res_dgrad_frag) || !al16(via_skip) || !al16(gate_t) ||
      !al16(gate_s) || !al16(dpre) || !al16(dpre_c)) return GT_E_ALIGN;
  WnArgs a = {};
  a.X = static_cast<const bf16_t*>(dpre_next); a.ldx = lddn; a.W1 = static_cast<const bf16_t*>(w_in_dgrad_frag);
  a.W2 = static_cast<const bf16_t*>(w_res_dgrad_frag);
  a.rowmask = rowmask; a.Xnext = static_cast<bf16_t*>(dx); a.ldxn = lddx; a.resid = static_cast<const bf16_t*>(resid); a.ldres = ldres;
  a.viaskip = static_cast<const bf16_t*>(via_skip); a.ldvs = ldvs;
  a.Tt = const_cast<bf16_t*>(static_cast<const bf16_t*>(gate_t)); a.Ss = const_cast<bf16_t*>(static_cast<const bf16_t*>(gate_s)); a.ldts = ldts;
  a.dpre = static_cast<bf16_t*>(dpre); a.dpre_c = static_cast<bf16_t*>(dpre_c); a.lddp = lddp; a.R = R; a.stamps = stamps; a.stamp_slot = stamp_slot; a.stamp_base = stamp_base;
  const int rc = fill_drop(a, drop_p, drop_seed, seed_dev);
  if (rc) return rc;
  static bool attr = false;                    // > 64 KB of LDS: opt in once (per process; the attribute is per device function)
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_wn_layer_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_wn_layer_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS) != hipSuccess)
      return GT_E_LAUNCH;
    attr = true;
  }
  const dim3 grid((R + BM - 1) / BM), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (w_res_dgrad_frag) hipLaunchKernelGGL(gt_wn_layer_bwd_kernel<true>, grid, block, BWD_LDS, st, a);
  else                  hipLaunchKernelGGL(gt_wn_layer_bwd_kernel<false>, grid, block, BWD_LDS, st, a);
  return gt_launch_status(__func__);
}
