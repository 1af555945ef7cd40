// A whole WaveNet (all n = 4 gated layers of modules.WN.forward, modules.py:144-171) as ONE kernel, gfx950.
//
// csrc/wn_layer.hip runs one kernel per layer: a workgroup owns 64 rows and all channels, and between two layers every row
// tile needs 2 rows of its neighbours (k = 5), which is why the layers are separate launches (~2.5 us of launch gap, the first
// tile's load latency and the store drain at the kernel boundary per layer, on the decoder's dependent chain).  Here a workgroup
// RECOMPUTES that halo instead of exchanging it: it loads 68 rows of the WaveNet's input, computes layer 0 on 64 rows, layer 1
// is then exact on the inner 60, layer 2 on 56, layer 3 on 52 — the workgroup OWNS those 52 rows and stores only them; tiles
// advance by 52 rows (9 216..9 728 rows of cfg 2 -> 178..188 workgroups instead of 144..152: the extra 23 % of arithmetic runs on
// CUs that were idle).  The layer's input never leaves LDS ([68][192] bf16, ping-pong), the k=5 conv needs no K-slice staging
// and no barrier in its loop, and x_{i+1} is stored (for the weight gradients / the backward) without anyone waiting for it.
// A halo row is computed by two workgroups with the same arithmetic in the same order and the same dropout hash (seed, global
// row, column): what they produce is bit-identical, so the result equals the per-layer kernels' bit for bit.
//
// Per layer (as wn_layer.hip): stage 1  x_in = conv_k5(x): N = 384, K = 960; 4 waves x (64 rows x 96 columns); weights as MFMA-A
// fragments L2 -> registers (3-step ring), activations = MFMA-B from the LDS tile; gate epilogue in registers (bias, dropout,
// cond, tanh * sigmoid; T, S, acts -> HBM for the owned rows, acts -> LDS); stage 2  x_next = (x + acts @ W_res^T + b) * mask.
#include "common.h"
#include "../../include/glowtts_hip.h"

namespace {

constexpr int H = 192, TAPS = 5, NLMAX = 4;
constexpr int BM = 64;                        // rows computed per layer
constexpr int AP = H + 8;                     // LDS pitch (halfs): 400 B = 16 mod 128 -> conflict-free ds_read_b128
constexpr int XR = BM + TAPS - 1;             // 68 input rows
#ifndef WNS_PIN_STEP
#define WNS_PIN_STEP 1
#endif
#ifndef WNS_EXP
#define WNS_EXP 0                             // dev experiments (bit mask), 0 in every build that ships
#endif
#ifndef WNS_RING
#define WNS_RING 3
#endif
constexpr int RING = WNS_RING;
constexpr int KK2 = H / 16;
constexpr int STACK_LDS = (2 * XR + BM) * AP * 2;                       // Xa, Xb [68][200] + At [64][200] = 80 000 B
constexpr int STACK_FWD_LDS = STACK_LDS + 2 * BM * AP * 2;              // forward: + the saved tanh / sigmoid tiles [64][200] each = 131 200 B
constexpr int CPR = H / 8;                                              // 16-byte chunks per row

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// through an explicit GLOBAL pointer: a pointer that went through `pinned` is generic to the compiler, and flat loads count on
// lgkmcnt as well as vmcnt — every LDS wait would then drain the weight prefetch
__device__ __forceinline__ uint4 ldfrag(const bf16_t* __restrict__ W, int f, int lane)
{
  typedef const u32x4_t __attribute__((address_space(1)))* gptr_t;
  const u32x4_t v = *reinterpret_cast<gptr_t>(reinterpret_cast<uintptr_t>(W + ((size_t)f * 64 + lane) * 8));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bf16x8_t asfrag(const uint4& u) { return __builtin_bit_cast(bf16x8_t, u); }
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack2bf(a, b), pack2bf(c, d)); }
__device__ __forceinline__ void unpack4(const uint2& u, float (&v)[4])
{
  v[0] = bf2f(u.x & 0xffff); v[1] = bf2f(u.x >> 16); v[2] = bf2f(u.y & 0xffff); v[3] = bf2f(u.y >> 16);
}
// A pointer the optimiser cannot see through: the 180 fragment addresses of a layer depend only on a kernel argument and the
// lane, so LLVM hoists them to the kernel's entry block — 360 registers that then live (spilled) across every earlier layer.
template <typename T>
__device__ __forceinline__ const T* pinned(const T* p) { asm volatile("" : "+s"(p)); return p; }
// The wave index as a SCALAR that is a fresh value wherever it is asked for: the fragment offsets (functions of the wave's
// column / K share) are then scalar arithmetic next to each load (SGPR base + 32-bit lane offset addressing) instead of ~180
// vector values that common-subexpression elimination keeps alive from one layer to the next.
__device__ __forceinline__ int wave_scalar()
{
  int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  asm volatile("" : "+s"(w));
  return w;
}

template <typename T>
__device__ __forceinline__ T pick(T const (&arr)[NLMAX], int i)        // scalar selects: no dynamically indexed kernarg copy
{
  return i == 0 ? arr[0] : (i == 1 ? arr[1] : (i == 2 ? arr[2] : arr[3]));
}

// The forward keeps a RUNTIME layer loop: the compiler then forms the layer's 180 fragment offsets once (hoisted out of the loop,
// parked in the accumulation registers the MFMAs do not use) — measured 8 % faster than the straight-line, scalar-addressed form
// the backward kernel needs (its register budget has no room for them).
__global__ __launch_bounds__(256) void gt_wn_stack_fwd_kernel(gt_wn_stack_fwd_args a, uint32_t drop_thresh, float drop_scale)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.stamps && threadIdx.x == 0)
    atomicMin(a.stamps + 2 * (a.stamp_slot + (a.stamp_base ? *a.stamp_base : 0)), (unsigned long long)wall_clock64());
  const uint32_t seed_x = a.seed_dev ? *a.seed_dev : 0u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_layers = a.n_layers, R = a.R;
  const int halo = 2 * (n_layers - 1);                               // rows lost per side: 6 for 4 layers
  const int own = BM - 2 * halo;                                     // rows this workgroup owns: 52
  const int s0 = blockIdx.x * own - halo;                            // global row of tile row 0
  bf16_t* Xc = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Xn = Xc + XR * AP;
  bf16_t* At = Xn + XR * AP;
  bf16_t* Tl = At + BM * AP;                                         // saved tanh / sigmoid halves on their way out (see the gate epilogue)
  bf16_t* Sl = Tl + BM * AP;
  constexpr int KS = H / 16, NBT = 2 * H / 32, NIT = 3 * TAPS;       // 12 k-steps per tap, 12 column blocks, 15 steps of 4 k-steps

  // layer-0 input: rows s0 - 2 .. s0 + 65, all 192 channels (rows outside [0, R) read as zero)
  {
    const bf16_t* x0 = static_cast<const bf16_t*>(a.x0);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int chunk = threadIdx.x + 256 * i, u = chunk / 24, c8 = chunk - u * 24, gm = s0 - 2 + u;
      if (u < XR) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gm >= 0 && gm < R) v = *reinterpret_cast<const uint4*>(x0 + (size_t)gm * H + c8 * 8);
        *reinterpret_cast<uint4*>(Xc + u * AP + c8 * 8) = v;
      }
    }
    if (threadIdx.x < 4 * 24) {                                       // the next tile's rows 0, 1, 66, 67 are never produced
      const int q = threadIdx.x / 24, c8 = threadIdx.x - q * 24, u = q < 2 ? q : XR - 4 + q;
      *reinterpret_cast<uint4*>(Xn + u * AP + c8 * 8) = make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();

  for (int layer = 0; layer < n_layers; ++layer) {
    const bool last = layer == n_layers - 1;
    const bf16_t* W1 = static_cast<const bf16_t*>(pick(a.w_in, layer));
    const bf16_t* W2 = static_cast<const bf16_t*>(pick(a.w_res, layer));
    const float* bias1 = pick(a.b_in, layer);
    const float* bias2 = pick(a.b_res, layer);
    bf16_t* Tt = static_cast<bf16_t*>(pick(a.gate_t, layer));
    bf16_t* Ss = static_cast<bf16_t*>(pick(a.gate_s, layer));
    bf16_t* xo = static_cast<bf16_t*>(pick(a.x_out, layer));
    const uint32_t seed = (a.drop_seed + (uint32_t)layer) ^ seed_x;

    f32x16_t acc[3][2];
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int bm = 0; bm < 2; ++bm)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[bn][bm][e] = 0.0f;

    uint4 ring[RING][4][3];
    auto w_load = [&](int it, uint4 (&dst)[4][3]) {
      if ((WNS_EXP & 2) && it > RING) return;
      const int kg = it / TAPS, tap = it - kg * TAPS;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int bn = 0; bn < 3; ++bn) dst[ks][bn] = ldfrag(W1, (tap * NBT + 3 * wave + bn) * KS + kg * 4 + ks, lane);
    };
#pragma unroll
    for (int p = 0; p < RING - 1; ++p) w_load(p, ring[p]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int kg = it / TAPS, tap = it - kg * TAPS;
      w_load(it + RING - 1 < NIT ? it + RING - 1 : NIT - 1, ring[(it + RING - 1) % RING]);
      __builtin_amdgcn_sched_barrier(0);       // keep the prefetch RING - 1 steps ahead
      const bf16_t* xsb = Xc + (r + tap) * AP + 8 * h + kg * 64;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(xsb + ks * 16);
        const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(xsb + 32 * AP + ks * 16);
#pragma unroll
        for (int bn = 0; bn < 3; ++bn) {
          acc[bn][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RING][ks][bn]), b0, acc[bn][0], 0, 0, 0);
          acc[bn][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RING][ks][bn]), b1, acc[bn][1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // second-stage weights start flying now, under the gate epilogue
    const int wn2 = wave & 1, wm2 = wave >> 1;
    uint4 ring2[KK2 / 2][3];
    if (!last) {
#pragma unroll
      for (int kk = 0; kk < KK2 / 2; ++kk)
#pragma unroll
        for (int bn = 0; bn < 3; ++bn) ring2[kk][bn] = ldfrag(W2, (3 * wn2 + bn) * KK2 + kk, lane);
    }

    // gate epilogue in registers (see wn_layer.hip): block 3*wave + bn holds [16 tanh | 16 sigmoid] of channels 16*(3*wave+bn)..+15
    bf16_t* acts = static_cast<bf16_t*>(a.acts) + layer * H;
    const float* cond = a.cond ? a.cond + (size_t)layer * 2 * H : nullptr;
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int c = 16 * (3 * wave + bn) + 8 * g + 4 * h;
        const float4 bt = *reinterpret_cast<const float4*>(bias1 + c), bs = *reinterpret_cast<const float4*>(bias1 + H + c);
        const float btv[4] = {bt.x, bt.y, bt.z, bt.w}, bsv[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
        for (int bm = 0; bm < 2; ++bm) {
          const int t = 32 * bm + r, m = s0 + t;
          const int mc = m < 0 ? 0 : (m >= R ? R - 1 : m);
          float ctv[4] = {}, csv[4] = {};
          if (cond) {
            const float* cp = cond + (size_t)(a.B > 0 ? gt_row_batch(a.row0, a.B, mc, a.Tp) : mc) * a.ldc + c;
            const float4 ct = *reinterpret_cast<const float4*>(cp), cs = *reinterpret_cast<const float4*>(cp + H);
            ctv[0] = ct.x; ctv[1] = ct.y; ctv[2] = ct.z; ctv[3] = ct.w; csv[0] = cs.x; csv[1] = cs.y; csv[2] = cs.z; csv[3] = cs.w;
          }
          float tt[4], ss[4], aa[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float vt = acc[bn][bm][4 * g + j] + btv[j], vs = acc[bn][bm][4 * g + j + 8] + bsv[j];
            if (drop_thresh && !(WNS_EXP & 4)) {                       // x_in = drop(conv(x)) (modules.py:153)
              bool kt, ks;
              drop_keep_gate(seed, m, c + j, drop_thresh16(drop_thresh), kt, ks);
              vt = kt ? vt * drop_scale : 0.0f;
              vs = ks ? vs * drop_scale : 0.0f;
            }
            vt += ctv[j]; vs += csv[j];
            if (WNS_EXP & 4) { tt[j] = vt; ss[j] = vs; aa[j] = vt + vs; } else { tt[j] = tanhf_(vt); ss[j] = sigmoidf_(vs); aa[j] = tt[j] * ss[j]; }
          }
          // T, S and acts leave through LDS: straight from the MFMA layout a store instruction writes 16 bytes to each of 32
          // rows (2.6 M partial-line requests per launch, the "17 us of stores" of DESIGN 4.9); the tiles below go out as whole rows
          *reinterpret_cast<uint2*>(Tl + t * AP + c) = pack4(tt[0], tt[1], tt[2], tt[3]);
          *reinterpret_cast<uint2*>(Sl + t * AP + c) = pack4(ss[0], ss[1], ss[2], ss[3]);
          *reinterpret_cast<uint2*>(At + t * AP + c) = pack4(aa[0], aa[1], aa[2], aa[3]);
        }
      }
    __syncthreads();                                                   // At, Tl, Sl complete
    if (!(WNS_EXP & 1)) {
      // the rows this workgroup owns are ONE contiguous block of T / S (and 384-byte pieces of the acts rows): 16 bytes per lane,
      // consecutive lanes on consecutive addresses
      for (int idx = threadIdx.x; idx < own * CPR; idx += 256) {
        const int row = idx / CPR, c8 = idx - row * CPR, t = halo + row, m = s0 + t;
        if (m < R) {
          const uint4 vt = *reinterpret_cast<const uint4*>(Tl + t * AP + c8 * 8), vs = *reinterpret_cast<const uint4*>(Sl + t * AP + c8 * 8),
                      va = *reinterpret_cast<const uint4*>(At + t * AP + c8 * 8);
          *reinterpret_cast<uint4*>(Tt + (size_t)m * H + c8 * 8) = vt;
          *reinterpret_cast<uint4*>(Ss + (size_t)m * H + c8 * 8) = vs;
          *reinterpret_cast<uint4*>(acts + (size_t)m * a.ldacts + c8 * 8) = va;
        }
      }
    }
    if (last) break;

    // stage 2: x_next = (x + acts @ W_res^T + b_res) * mask -> the next layer's LDS tile (+ HBM for the owned rows)
    f32x16_t acc2[3];
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[bn][e] = 0.0f;
    {
      const bf16_t* ab = At + (32 * wm2 + r) * AP + 8 * h;
#pragma unroll
      for (int kk = 0; kk < KK2; ++kk) {
        const bf16x8_t bfm = *reinterpret_cast<const bf16x8_t*>(ab + kk * 16);
#pragma unroll
        for (int bn = 0; bn < 3; ++bn) {
          acc2[bn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring2[kk % (KK2 / 2)][bn]), bfm, acc2[bn], 0, 0, 0);
          if (kk + KK2 / 2 < KK2) ring2[kk % (KK2 / 2)][bn] = ldfrag(W2, (3 * wn2 + bn) * KK2 + kk + KK2 / 2, lane);
        }
      }
    }
    {
      const int t = 32 * wm2 + r, m = s0 + t;
      const float rm = (m >= 0 && m < R) ? a.rowmask[m] : 0.0f;
#pragma unroll
      for (int bn = 0; bn < 3; ++bn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 32 * (3 * wn2 + bn) + 8 * g + 4 * h;
          const float4 b2 = *reinterpret_cast<const float4*>(bias2 + n);
          float xv[4];
          unpack4(*reinterpret_cast<const uint2*>(Xc + (t + 2) * AP + n), xv);
          const uint2 v = pack4((acc2[bn][4 * g] + b2.x + xv[0]) * rm, (acc2[bn][4 * g + 1] + b2.y + xv[1]) * rm,
                                (acc2[bn][4 * g + 2] + b2.z + xv[2]) * rm, (acc2[bn][4 * g + 3] + b2.w + xv[3]) * rm);
          *reinterpret_cast<uint2*>(Xn + (t + 2) * AP + n) = v;
        }
    }
    __syncthreads();                                                   // the next layer's input is complete; Xc and At are free
    bf16_t* tmp = Xc; Xc = Xn; Xn = tmp;
    if (!(WNS_EXP & 1)) {                                              // x_next of the owned rows, whole rows from the finished tile
      for (int idx = threadIdx.x; idx < own * CPR; idx += 256) {
        const int row = idx / CPR, c8 = idx - row * CPR, t = halo + row, m = s0 + t;
        if (m < R) *reinterpret_cast<uint4*>(xo + (size_t)m * H + c8 * 8) = *reinterpret_cast<const uint4*>(Xc + (t + 2) * AP + c8 * 8);
      }
    }
    if (threadIdx.x < 4 * 24) {                                        // rows 0, 1, 66, 67 of the tile after next
      const int q = threadIdx.x / 24, c8 = threadIdx.x - q * 24, u = q < 2 ? q : XR - 4 + q;
      *reinterpret_cast<uint4*>(Xn + u * AP + c8 * 8) = make_uint4(0, 0, 0, 0);
    }
  }
  if (a.stamps && threadIdx.x == 0)
    atomicMax(a.stamps + 2 * (a.stamp_slot + (a.stamp_base ? *a.stamp_base : 0)) + 1, (unsigned long long)wall_clock64());
}

// ------------------------------------------------------------------------------------------------ backward
// The mirror: the gate backward of the top layer (row-local, from the skip-path gradient), then for j = n-1 .. 0 the k=5 data
// gradient of in_layer_j on the d pre_j tile -> dX_j (+ the residual-path gradient dX_{j+1}, * mask), and below it the
// residual 1x1's data gradient + skip-path gradient + gate backward -> d pre_{j-1}.  The first conv reads an exact 68-row
// tile, each of the other n-1 loses 2 rows per side: the workgroup owns the same 64 - 4 (n-1) rows as the forward.
//   stage 1: N = 192, K = 5 * 384: 2 (column halves) x 2 (K halves) waves x 64 rows, partial sums exchanged through LDS
//   stage 2: N = 192, K = 192 on the dX tile: 2 x 2 waves x (32 rows x 96 columns)
constexpr int DP = 2 * H + 8;                 // d pre tile pitch (halfs): 784 B = 16 mod 128
constexpr int EX_BYTES = 4 * 3 * 16 * 64 * 4; // partial-sum exchange [4 waves][3 blocks][16][64 lanes] fp32: aliases the d pre tile
constexpr int BWD_DT = XR * DP * 2;           // 53 312 B
static_assert(EX_BYTES <= BWD_DT, "the exchange buffer lives in the d pre tile");
constexpr int SBWD_LDS = BWD_DT + 4 * BM * AP * 2;       // d pre tile | dX tile | skip-gradient, tanh, sigmoid staging tiles = 155 712 B

__device__ __forceinline__ void gate_bwd4(const float (&dd)[4], const float (&t)[4], const float (&s)[4], uint32_t seed, int m, int n,
                                          uint32_t thresh, float scale, uint2& pt, uint2& ps, uint2& ct, uint2& cs)
{
  float gt[4], gs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { gt[j] = dd[j] * s[j] * (1.0f - t[j] * t[j]); gs[j] = dd[j] * t[j] * s[j] * (1.0f - s[j]); }
  ct = pack4(gt[0], gt[1], gt[2], gt[3]); cs = pack4(gs[0], gs[1], gs[2], gs[3]);      // before the dropout mask: d cond
  if (thresh) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bool kt, ks;
      drop_keep_gate(seed, m, n + j, drop_thresh16(thresh), kt, ks);
      gt[j] = kt ? gt[j] * scale : 0.0f;
      gs[j] = ks ? gs[j] * scale : 0.0f;
    }
  }
  pt = pack4(gt[0], gt[1], gt[2], gt[3]); ps = pack4(gs[0], gs[1], gs[2], gs[3]);
}

// one step j of the chain (compile-time j: every pointer is a kernel argument, every fragment address is formed where it is used)
template <int J>
__device__ __forceinline__ void bwd_step(const gt_wn_stack_bwd_args& a, uint32_t drop_thresh, float drop_scale, uint32_t seed_x,
                                         bf16_t* Dt, float* Ex, bf16_t* At, int s0, int halo, int lane, int wave)
{
  const int r = lane & 31, h = lane >> 5;
  wave = wave_scalar();
  const int wn = wave & 1, wk = wave >> 1;     // stage 1: column half, K half;  afterwards wk doubles as the row half
  const int n_layers = a.n_layers, R = a.R;
  const bf16_t* via = static_cast<const bf16_t*>(a.via_skip);
  constexpr int NS = 2 * H / 64, NIT = NS * TAPS, KS = 2 * H / 16, NBT = H / 32;  // 6 slices, 30 steps, 24 k-steps per tap, 6 blocks
  const bf16_t* W1 = pinned(static_cast<const bf16_t*>(a.w_in_d[J]));
  f32x16_t acc[3][2];
#pragma unroll
  for (int bn = 0; bn < 3; ++bn)
#pragma unroll
    for (int bm = 0; bm < 2; ++bm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[bn][bm][e] = 0.0f;
  constexpr int RB = 2 * RING - 1;             // half the fragments per step: twice the depth for the same registers
  uint4 ring[RB][2][3];
  auto w_load = [&](int it, uint4 (&dst)[2][3]) {
    const int slice = it / TAPS, tap = it - slice * TAPS;
    const bf16_t* Wt = pinned(W1 + (size_t)((tap * NBT + 3 * wn) * KS + slice * 4 + 2 * wk) * 512);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) dst[k2][bn] = ldfrag(Wt, bn * KS + k2, lane);
  };
#pragma unroll
  for (int p = 0; p < RB - 1; ++p) w_load(p, ring[p]);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int slice = it / TAPS, tap = it - slice * TAPS;
    w_load(it + RB - 1 < NIT ? it + RB - 1 : NIT - 1, ring[(it + RB - 1) % RB]);
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t* xsb = Dt + (r + tap) * DP + 8 * h + slice * 64 + 32 * wk;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(xsb + k2 * 16);
      const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(xsb + 32 * DP + k2 * 16);
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) {
        acc[bn][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RB][k2][bn]), b0, acc[bn][0], 0, 0, 0);
        acc[bn][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RB][k2][bn]), b1, acc[bn][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // second-stage weights (the layer below): this wave's 96 columns (wn), rows 32*wk
  constexpr int JL = J > 0 ? J - 1 : 0;
  const bf16_t* W2 = pinned(static_cast<const bf16_t*>(a.w_res_d[JL]));
  uint4 ring2[KK2 / 2][3];
  if (J > 0) {
#pragma unroll
    for (int kk = 0; kk < KK2 / 2; ++kk)
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) ring2[kk][bn] = ldfrag(W2, (3 * wn + bn) * KK2 + kk, lane);
  }
  __syncthreads();                            // every wave is done with the d pre tile: the exchange buffer may overwrite it

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
  // dX_j = (conv^T(d pre_j) + dX_{j+1}) * mask -> HBM (owned rows) and the stage-2 tile (which still holds dX_{j+1})
  const int t = 32 * wk + r, m = s0 + t;
  const bool mine_row = t >= halo && t < BM - halo && m < R;
  {
    const float rm = (m >= 0 && m < R) ? a.rowmask[m] : 0.0f;
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (3 * wn + bn) + 8 * g + 4 * h;
        float ad[4] = {};
        if (J < n_layers - 1) unpack4(*reinterpret_cast<const uint2*>(At + t * AP + n), ad);
        const uint2 v = pack4((sum[bn][4 * g] + ad[0]) * rm, (sum[bn][4 * g + 1] + ad[1]) * rm,
                              (sum[bn][4 * g + 2] + ad[2]) * rm, (sum[bn][4 * g + 3] + ad[3]) * rm);
        *reinterpret_cast<uint2*>(At + t * AP + n) = v;
      }
  }
  __syncthreads();
  {
    // dX_j of the owned rows leaves as whole rows from the finished tile (the MFMA layout gives a store 16 bytes in each of 32 rows)
    bf16_t* dx = static_cast<bf16_t*>(a.dx[J]);
    const int own = BM - 2 * halo;
    for (int idx = threadIdx.x; idx < own * CPR; idx += 256) {
      const int row = idx / CPR, c8 = idx - row * CPR, tt = halo + row, mm = s0 + tt;
      if (mm < R) *reinterpret_cast<uint4*>(dx + (size_t)mm * H + c8 * 8) = *reinterpret_cast<const uint4*>(At + tt * AP + c8 * 8);
    }
  }
  if (J == 0) return;                                               // bottom: dX_0 is the gradient at the WaveNet's input

  // d acts_{j-1} = dX_j W_res + skip-path gradient -> gate backward -> d pre_{j-1}: the next tile (+ HBM for the owned rows)
  // the gate backward's three row operands (skip-path gradient, saved tanh, saved sigmoid of layer j-1, all 64 rows of the tile) are
  // fetched as whole rows under the GEMM below and handed to the epilogue through LDS — read in the MFMA layout, every load
  // instruction touched 16 bytes in each of 32 rows
  constexpr int JLp = J > 0 ? J - 1 : 0;
  constexpr int NPRE = (BM * CPR + 255) / 256;                       // 6 chunks per thread and operand
  bf16_t* Vl = At + BM * AP; bf16_t* Tl = Vl + BM * AP; bf16_t* Sl = Tl + BM * AP;
  uint4 pre[3][NPRE];
  {
    const bf16_t* Tg = static_cast<const bf16_t*>(a.gate_t[JLp]);
    const bf16_t* Sg = static_cast<const bf16_t*>(a.gate_s[JLp]);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = threadIdx.x + 256 * i, row = idx / CPR, c8 = idx - row * CPR, mm = s0 + row;
      pre[0][i] = pre[1][i] = pre[2][i] = make_uint4(0, 0, 0, 0);
      if (mm >= 0 && mm < R) {
        pre[0][i] = *reinterpret_cast<const uint4*>(via + (size_t)mm * a.ldvs + JLp * H + c8 * 8);
        pre[1][i] = *reinterpret_cast<const uint4*>(Tg + (size_t)mm * H + c8 * 8);
        pre[2][i] = *reinterpret_cast<const uint4*>(Sg + (size_t)mm * H + c8 * 8);
      }
    }
  }
  f32x16_t acc2[3];
#pragma unroll
  for (int bn = 0; bn < 3; ++bn)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[bn][e] = 0.0f;
  {
    const bf16_t* ab = At + (32 * wk + r) * AP + 8 * h;
#pragma unroll
    for (int kk = 0; kk < KK2; ++kk) {
      const bf16x8_t bfm = *reinterpret_cast<const bf16x8_t*>(ab + kk * 16);
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) {
        acc2[bn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring2[kk % (KK2 / 2)][bn]), bfm, acc2[bn], 0, 0, 0);
        if (kk + KK2 / 2 < KK2) ring2[kk % (KK2 / 2)][bn] = ldfrag(W2, (3 * wn + bn) * KK2 + kk + KK2 / 2, lane);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int idx = threadIdx.x + 256 * i, row = idx / CPR, c8 = idx - row * CPR;
    *reinterpret_cast<uint4*>(Vl + row * AP + c8 * 8) = pre[0][i];
    *reinterpret_cast<uint4*>(Tl + row * AP + c8 * 8) = pre[1][i];
    *reinterpret_cast<uint4*>(Sl + row * AP + c8 * 8) = pre[2][i];
  }
  __syncthreads();
  {
    bf16_t* dpre_c = static_cast<bf16_t*>(a.dpre_c[JL]);
    const uint32_t seed = (a.drop_seed + (uint32_t)JL) ^ seed_x;
    const bool in = m >= 0 && m < R;
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (3 * wn + bn) + 8 * g + 4 * h;
        uint2 pt = make_uint2(0, 0), ps = pt, ct = pt, cs = pt;
        if (in) {
          float vs[4], tt[4], sg[4], dd[4];
          unpack4(*reinterpret_cast<const uint2*>(Vl + t * AP + n), vs);
          unpack4(*reinterpret_cast<const uint2*>(Tl + t * AP + n), tt);
          unpack4(*reinterpret_cast<const uint2*>(Sl + t * AP + n), sg);
#pragma unroll
          for (int q = 0; q < 4; ++q) dd[q] = acc2[bn][4 * g + q] + vs[q];
          gate_bwd4(dd, tt, sg, seed, m, n, drop_thresh, drop_scale, pt, ps, ct, cs);
          if (mine_row && dpre_c) {                                  // (speaker-conditioned configs only)
            *reinterpret_cast<uint2*>(dpre_c + (size_t)m * 2 * H + n) = ct;
            *reinterpret_cast<uint2*>(dpre_c + (size_t)m * 2 * H + H + n) = cs;
          }
        }
        *reinterpret_cast<uint2*>(Dt + (t + 2) * DP + n) = pt;
        *reinterpret_cast<uint2*>(Dt + (t + 2) * DP + H + n) = ps;
      }
  }
  __syncthreads();                                                   // the next conv's input tile is complete
  {
    // d pre_{j-1} of the owned rows: whole 768-byte rows from the tile (read-only until the next step's exchange, which follows a barrier)
    bf16_t* dpre = static_cast<bf16_t*>(a.dpre[JL]);
    const int own = BM - 2 * halo;
    constexpr int CPR2 = 2 * H / 8;
    for (int idx = threadIdx.x; idx < own * CPR2; idx += 256) {
      const int row = idx / CPR2, c8 = idx - row * CPR2, tt = halo + row, mm = s0 + tt;
      if (mm < R) *reinterpret_cast<uint4*>(dpre + (size_t)mm * 2 * H + c8 * 8) = *reinterpret_cast<const uint4*>(Dt + (tt + 2) * DP + c8 * 8);
    }
  }
}

__global__ __launch_bounds__(256) void gt_wn_stack_bwd_kernel(gt_wn_stack_bwd_args a, uint32_t drop_thresh, float drop_scale)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t seed_x = a.seed_dev ? *a.seed_dev : 0u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_layers = a.n_layers, R = a.R;
  const int halo = 2 * (n_layers - 1), own = BM - 2 * halo;
  const int s0 = blockIdx.x * own - halo;
  bf16_t* Dt = reinterpret_cast<bf16_t*>(smem);
  float* Ex = reinterpret_cast<float*>(smem);
  bf16_t* At = reinterpret_cast<bf16_t*>(smem + BWD_DT);
  const bf16_t* via = static_cast<const bf16_t*>(a.via_skip);

  // top layer: d acts = the skip-path gradient only -> d pre on all 68 rows of the tile (row-local)
  {
    const int L = n_layers - 1;
    const bf16_t* Tt = static_cast<const bf16_t*>(pick(a.gate_t, L));
    const bf16_t* Ss = static_cast<const bf16_t*>(pick(a.gate_s, L));
    bf16_t* dpre = static_cast<bf16_t*>(pick(a.dpre, L));
    bf16_t* dpre_c = static_cast<bf16_t*>(pick(a.dpre_c, L));
    const uint32_t seed = (a.drop_seed + (uint32_t)L) ^ seed_x;
    for (int item = threadIdx.x; item < XR * (H / 4); item += 256) {
      const int u = item / (H / 4), n = 4 * (item - u * (H / 4)), m = s0 - 2 + u;
      uint2 pt = make_uint2(0, 0), ps = pt, ct = pt, cs = pt;
      if (m >= 0 && m < R) {
        float dd[4], t[4], sg[4];
        unpack4(*reinterpret_cast<const uint2*>(via + (size_t)m * a.ldvs + L * H + n), dd);
        unpack4(*reinterpret_cast<const uint2*>(Tt + (size_t)m * H + n), t);
        unpack4(*reinterpret_cast<const uint2*>(Ss + (size_t)m * H + n), sg);
        gate_bwd4(dd, t, sg, seed, m, n, drop_thresh, drop_scale, pt, ps, ct, cs);
        if (u - 2 >= halo && u - 2 < BM - halo) {
          *reinterpret_cast<uint2*>(dpre + (size_t)m * 2 * H + n) = pt;
          *reinterpret_cast<uint2*>(dpre + (size_t)m * 2 * H + H + n) = ps;
          if (dpre_c) {
            *reinterpret_cast<uint2*>(dpre_c + (size_t)m * 2 * H + n) = ct;
            *reinterpret_cast<uint2*>(dpre_c + (size_t)m * 2 * H + H + n) = cs;
          }
        }
      }
      *reinterpret_cast<uint2*>(Dt + u * DP + n) = pt;
      *reinterpret_cast<uint2*>(Dt + u * DP + H + n) = ps;
    }
  }
  __syncthreads();
  // the chain, top to bottom (workgroup-uniform branches; each step is its own straight-line code)
  if (n_layers > 3) bwd_step<3>(a, drop_thresh, drop_scale, seed_x, Dt, Ex, At, s0, halo, lane, wave);
  if (n_layers > 2) bwd_step<2>(a, drop_thresh, drop_scale, seed_x, Dt, Ex, At, s0, halo, lane, wave);
  if (n_layers > 1) bwd_step<1>(a, drop_thresh, drop_scale, seed_x, Dt, Ex, At, s0, halo, lane, wave);
  bwd_step<0>(a, drop_thresh, drop_scale, seed_x, Dt, Ex, At, s0, halo, lane, wave);
}

inline bool al16(const void* p) { return !((uintptr_t)p & 15); }

}  // namespace

extern "C" int gt_wn_stack_rows_per_workgroup(int n_layers) { return BM - 4 * (n_layers - 1); }

extern "C" int gt_wn_stack_fwd(const gt_wn_stack_fwd_args* args, void* stream)
{
  if (!args) return GT_E_INVAL;
  const gt_wn_stack_fwd_args& a = *args;
  if (a.R < 0) return GT_E_INVAL;
  if (a.R == 0) return GT_OK;
  if (a.H != H || a.taps != TAPS || a.n_layers < 1 || a.n_layers > NLMAX) return GT_E_UNSUPPORTED;
  if (!a.x0 || !a.rowmask || !a.acts || a.ldacts < a.n_layers * H || (a.ldacts & 7) || (a.cond && (a.ldc & 3))) return GT_E_INVAL;
  if (!al16(a.x0) || !al16(a.acts) || !al16(a.cond)) return GT_E_ALIGN;
  for (int i = 0; i < a.n_layers; ++i) {
    if (!a.w_in[i] || !a.b_in[i] || !a.gate_t[i] || !a.gate_s[i]) return GT_E_INVAL;
    if (i < a.n_layers - 1 && (!a.w_res[i] || !a.b_res[i] || !a.x_out[i])) return GT_E_INVAL;
    if (!al16(a.w_in[i]) || !al16(a.b_in[i]) || !al16(a.gate_t[i]) || !al16(a.gate_s[i]) || !al16(a.w_res[i]) || !al16(a.b_res[i]) ||
        !al16(a.x_out[i])) return GT_E_ALIGN;
  }
  if (a.cond && (a.Tp <= 0 || (a.row0 && a.B <= 0))) return GT_E_INVAL;
  uint32_t thresh = 0; float scale = 1.0f;
  if (a.drop_p > 0.0f) {
    if (a.drop_p >= 1.0f) return GT_E_UNSUPPORTED;
    thresh = (uint32_t)((double)a.drop_p * 4294967296.0); scale = 1.0f / (1.0f - a.drop_p);
  }
  static bool attr = false;                    // > 64 KB of LDS: opt in once per process
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_wn_stack_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, STACK_FWD_LDS) != hipSuccess)
      return GT_E_LAUNCH;
    attr = true;
  }
  const int own = BM - 4 * (a.n_layers - 1);
  const dim3 grid((a.R + own - 1) / own), block(256);
  hipLaunchKernelGGL(gt_wn_stack_fwd_kernel, grid, block, STACK_FWD_LDS, static_cast<hipStream_t>(stream), a, thresh, scale);
  return gt_launch_status(__func__);
}

extern "C" int gt_wn_stack_bwd(const gt_wn_stack_bwd_args* args, void* stream)
{
  if (!args) return GT_E_INVAL;
  const gt_wn_stack_bwd_args& a = *args;
  if (a.R < 0) return GT_E_INVAL;
  if (a.R == 0) return GT_OK;
  if (a.H != H || a.taps != TAPS || a.n_layers < 1 || a.n_layers > NLMAX) return GT_E_UNSUPPORTED;
  if (!a.via_skip || !a.rowmask || a.ldvs < a.n_layers * H || (a.ldvs & 7)) return GT_E_INVAL;
  if (!al16(a.via_skip)) return GT_E_ALIGN;
  for (int i = 0; i < a.n_layers; ++i) {
    if (!a.w_in_d[i] || !a.gate_t[i] || !a.gate_s[i] || !a.dpre[i] || !a.dx[i]) return GT_E_INVAL;
    if (i < a.n_layers - 1 && !a.w_res_d[i]) return GT_E_INVAL;
    if (!al16(a.w_in_d[i]) || !al16(a.gate_t[i]) || !al16(a.gate_s[i]) || !al16(a.dpre[i]) || !al16(a.dpre_c[i]) || !al16(a.dx[i]) ||
        !al16(a.w_res_d[i])) return GT_E_ALIGN;
  }
  uint32_t thresh = 0; float scale = 1.0f;
  if (a.drop_p > 0.0f) {
    if (a.drop_p >= 1.0f) return GT_E_UNSUPPORTED;
    thresh = (uint32_t)((double)a.drop_p * 4294967296.0); scale = 1.0f / (1.0f - a.drop_p);
  }
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_wn_stack_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SBWD_LDS) != hipSuccess)
      return GT_E_LAUNCH;
    attr = true;
  }
  const int own = BM - 4 * (a.n_layers - 1);
  const dim3 grid((a.R + own - 1) / own), block(256);
  hipLaunchKernelGGL(gt_wn_stack_bwd_kernel, grid, block, SBWD_LDS, static_cast<hipStream_t>(stream), a, thresh, scale);
  return gt_launch_status(__func__);
}
