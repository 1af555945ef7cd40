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
#ifndef WNS_LDSBAR
#define WNS_LDSBAR 0                          // 1: barriers wait for LDS traffic only (common.h lds_barrier) instead of __syncthreads()'s
                                              // vmcnt(0).  Measured (round 3, back to back): backward 86.6 us either way, forward 71.2 vs
                                              // 67.9 us — the activation stores then sit in front of the next layer's weight loads on
                                              // the in-order vmcnt counter and stall the conv loop instead of the barrier
#endif
#if WNS_LDSBAR
#define WNS_BARRIER() lds_barrier()
#else
#define WNS_BARRIER() __syncthreads()
#endif
#ifndef WNS_FIXA
#define WNS_FIXA 1                            // loads ahead of stores (see the forward kernel)
#endif
constexpr int RING = WNS_RING;
#ifndef WNS_RING1
#define WNS_RING1 WNS_RING                    // weight ring depths of the 32-row form (NBM = 1).  Its spare accumulator registers could pay for
#endif                                        // deeper rings; measured at 5 120 rows (round 3): forward 56.6 / 58.4 / 58.4 us at depth 4 / 5 / 3,
#ifndef WNS_RB1                               // backward 73.7 / 68.2 us at depth 9 / 5 — what the 256 workgroups wait for is the L2 itself
#define WNS_RB1 (2 * WNS_RING - 1)            // (each streams all 3.2 MB of weights: 0.8 GB per launch), not a ring step's round trip
#endif
#ifndef WNS_PHASES
#define WNS_PHASES 0                          // dev: per-phase shader-clock stamps of wave 0 (tools/wn_stack_phases.py), 0 in every build that ships
#endif
#if WNS_PHASES
__device__ unsigned long long g_wns_ph[1024 * 48];
#define PH(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_wns_ph[blockIdx.x * 48 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH(i) do { } while (0)
#endif
constexpr int KK2 = H / 16;
constexpr int STACK_LDS = (2 * XR + BM) * AP * 2;                       // Xa, Xb [68][200] + At [64][200] = 80 000 B
constexpr int FWD_BIAS_FLOATS = NLMAX * 2 * H + (NLMAX - 1) * H;        // every layer's biases, staged once (see the forward kernel)
constexpr int FWD_AFF_FLOATS = 2 * NLMAX * H;                            // affine conditioning (COND == 2): w | b of cond_layer1, H * n each
constexpr int STACK_FWD_LDS = STACK_LDS + 2 * BM * AP * 2 + FWD_BIAS_FLOATS * 4;   // forward: + the saved tanh / sigmoid tiles [64][200] each + biases = 139 648 B
constexpr int STACK_FWD_LDS_AFF = STACK_FWD_LDS + FWD_AFF_FLOATS * 4;              // + 6 144 B
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
// COND / DROP are compile-time: as run-time (uniform) branches around the conditioning loads and the dropout hashes they cost a
// branch per element and a conservative `s_waitcnt vmcnt(0)` at every join.
// COND: 0 none; 1 rows of a cond tensor (per utterance: the speaker vector through cond_layer; or per row); 2 AFFINE per-frame
// conditioning formed here: modules.WNP (modules.py:316-343) conditions on cond_layer1(contour), a conv with ONE input channel —
// w[c] * contour[t] + b[c] — squeezed in time (modules.py:353-362): squeezed row m, column parity * O + c (O = H n) holds
// w[c] * contour[2 m + parity] + b[c], and layer i reads columns [2H i, 2H (i+1)).  Materialised, that is 6 KB of fp32 per row written
// by a host-side op and read back here; formed from the row's two contour values and the 2 x O parameters it is 8 bytes per row.
// NBM: 32-row blocks computed per layer — 2 (64-row tiles, 52 owned with 4 layers) or 1 (32-row tiles, 20 owned) for launches whose
// 64-row tiles would leave most of the chip idle (cfg 4 / cfg 5's 4-5 k rows: 90 workgroups on 256 CUs): 2.6 x the workgroups, each
// with half the MFMA rows but the same weight stream.  The LDS layout is the 64-row one either way; every row's arithmetic (K order,
// dropout hash, epilogue) is the same, so the results are bit-identical between the two forms.
template <int COND, bool DROP, int NBM>
__global__ __launch_bounds__(256) void gt_wn_stack_fwd_kernel(gt_wn_stack_fwd_args a, uint32_t drop_thresh, float drop_scale)
{
  constexpr int BMv = 32 * NBM, XRv = BMv + TAPS - 1;
  constexpr int RINGv = NBM == 1 ? WNS_RING1 : RING;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // bench.py's live timing: workgroup 0's start (the first dispatched) is kept in a register and stored at the end, every workgroup's
  // end goes into one atomicMax AFTER its last wait — an atomicMin here, 188 workgroups on one address, sat in front of every
  // weight fragment's wait (vector-memory operations retire in order)
  unsigned long long t_begin = 0;
  if (a.stamps && threadIdx.x == 0 && blockIdx.x == 0) t_begin = (unsigned long long)wall_clock64();
  PH(0);
  const uint32_t seed_x = a.seed_dev ? *a.seed_dev : 0u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_layers = a.n_layers, R = a.R;
  const int halo = 2 * (n_layers - 1);                               // rows lost per side: 6 for 4 layers
  const int own = BMv - 2 * halo;                                    // rows this workgroup owns: 52 (NBM = 2) or 20
  const int s0 = blockIdx.x * own - halo;                            // global row of tile row 0
  bf16_t* Xc = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Xn = Xc + XR * AP;
  bf16_t* At = Xn + XR * AP;
  bf16_t* Tl = At + BM * AP;                                         // saved tanh / sigmoid halves on their way out (see the gate epilogue)
  bf16_t* Sl = Tl + BM * AP;
  // Every layer's biases live in LDS from the start: read from global memory inside the epilogues (two float4 per 4 gate channels,
  // each waited for where it is used) they cost a full L2 round trip per (column block, channel group) — 5 k of a layer's 12 k
  // epilogue cycles (tools/wn_stack_phases.py).
  float* Bs = reinterpret_cast<float*>(Sl + BM * AP);                // [n][2H] in_layer biases | [n-1][H] residual biases
  constexpr int KS = H / 16, NBT = 2 * H / 32, NIT = 3 * TAPS;       // 12 k-steps per tap, 12 column blocks, 15 steps of 4 k-steps

  // Vector-memory operations retire IN ORDER on one counter (vmcnt): a wait for a weight fragment also waits for every store issued
  // before that load.  So the loads a layer needs first are issued BEFORE the stores that precede their use: the next layer's first
  // ring steps ahead of the T / S / acts / x_next stores, and layer 0's here, ahead of (and overlapping) the input tile's loads.
  uint4 ring[RINGv][4][3];
  float rm2;
  // Fragment addresses are left to the compiler: it forms the layer's per-lane offsets once, outside the layer loop, and every load
  // is then ONE instruction with an immediate offset.  Measured against scalar bases formed next to each load (`pinned`, as the
  // backward needs for its register budget: + 7 us per launch — one wave per SIMD issues in order, so every scalar instruction in
  // front of a load is time the matrix pipe idles), against buffer loads with the descriptor in SGPRs (+ 2.5 us), and against
  // pinning each load between two MFMAs (sched_group_barrier: + 5 us — a vector-memory instruction holds the wave's issue for
  // longer than the MFMA it was meant to hide behind): tools/stack_variants.sh, round 3.
  auto w_frag = [&](const bf16_t* W, int it, int ks, int bn) {
    const int kg = it / TAPS, tap = it - kg * TAPS;
    return ldfrag(W, (tap * NBT + 3 * wave + bn) * KS + kg * 4 + ks, lane);
  };
  auto w_load_l = [&](const bf16_t* W, int it, uint4 (&dst)[4][3]) {
    if ((WNS_EXP & 2) && it > RINGv) return;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) dst[ks][bn] = w_frag(W, it, ks, bn);
  };
  // layer-0 input: rows s0 - 2 .. s0 + 65, all 192 channels (rows outside [0, R) read as zero); every layer's biases; the row
  // mask of the one row this lane finishes in every residual stage (32 * (wave >> 1) + r).  All of these loads are issued first,
  // then layer 0's first ring steps (a wait for the tile must not wait for the weights behind it), then the LDS writes.
  {
    const bf16_t* x0 = static_cast<const bf16_t*>(a.x0);
    constexpr int NX = (XR * CPR + 255) / 256, NB = (FWD_BIAS_FLOATS / 4 + 255) / 256;   // 7 and 3 per thread
    uint4 xin[NX];
    float4 bin[NB];
    float4 ain[(FWD_AFF_FLOATS / 4 + 255) / 256];
    if (COND == 2) {
      const int O = H * n_layers;
#pragma unroll
      for (int i = 0; i < (FWD_AFF_FLOATS / 4 + 255) / 256; ++i) {
        const int idx = threadIdx.x + 256 * i;                       // [w (O) | b (O)] as float4
        ain[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < O / 4) ain[i] = reinterpret_cast<const float4*>(a.aff_w)[idx];
        else if (idx < 2 * O / 4) ain[i] = reinterpret_cast<const float4*>(a.aff_b)[idx - O / 4];
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int chunk = threadIdx.x + 256 * i, u = chunk / CPR, c8 = chunk - u * CPR, gm = s0 - 2 + u;
      xin[i] = make_uint4(0, 0, 0, 0);
      if (u < XRv && gm >= 0 && gm < R) xin[i] = *reinterpret_cast<const uint4*>(x0 + (size_t)gm * H + c8 * 8);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = threadIdx.x + 256 * i;
      bin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < NLMAX * 2 * H / 4) {
        const int l = idx / (2 * H / 4), o = idx - l * (2 * H / 4);
        if (l < n_layers) bin[i] = reinterpret_cast<const float4*>(pick(a.b_in, l))[o];
      } else if (idx < FWD_BIAS_FLOATS / 4) {
        const int q = idx - NLMAX * 2 * H / 4, l = q / (H / 4), o = q - l * (H / 4);
        if (l < n_layers - 1) bin[i] = reinterpret_cast<const float4*>(pick(a.b_res, l))[o];
      }
    }
    {
      const int m = s0 + 32 * (wave >> 1) + r;
      rm2 = (m >= 0 && m < R) ? a.rowmask[m] : 0.0f;
    }
#if WNS_FIXA
    {
      const bf16_t* W10 = static_cast<const bf16_t*>(a.w_in[0]);
#pragma unroll
      for (int p = 0; p < RINGv - 1; ++p) w_load_l(W10, p, ring[p]);
    }
#endif
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int chunk = threadIdx.x + 256 * i, u = chunk / CPR, c8 = chunk - u * CPR;
      if (u < XRv) *reinterpret_cast<uint4*>(Xc + u * AP + c8 * 8) = xin[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = threadIdx.x + 256 * i;
      if (idx < FWD_BIAS_FLOATS / 4) reinterpret_cast<float4*>(Bs)[idx] = bin[i];
    }
    if (COND == 2) {
#pragma unroll
      for (int i = 0; i < (FWD_AFF_FLOATS / 4 + 255) / 256; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < 2 * H * n_layers / 4) reinterpret_cast<float4*>(Bs + FWD_BIAS_FLOATS)[idx] = ain[i];
      }
    }
    if (threadIdx.x < 4 * 24) {                                       // the next tile's rows 0, 1, 66, 67 are never produced
      const int q = threadIdx.x / 24, c8 = threadIdx.x - q * 24, u = q < 2 ? q : XRv - 4 + q;
      *reinterpret_cast<uint4*>(Xn + u * AP + c8 * 8) = make_uint4(0, 0, 0, 0);
    }
  }
  WNS_BARRIER();
  PH(1);

  // per-utterance conditioning: the utterance of this lane's two rows (one binary search each, once per launch)
  int cb[NBM] = {};
  float sg[NBM][2] = {};                          // COND == 2: the two contour values (even / odd frame) of this lane's two rows
  if (COND == 1) {
#pragma unroll
    for (int bm = 0; bm < NBM; ++bm) {
      const int m = s0 + 32 * bm + r, mc = m < 0 ? 0 : (m >= R ? R - 1 : m);
      cb[bm] = a.B > 0 ? gt_row_batch(a.row0, a.B, mc, a.Tp) : mc;
    }
  }
  if (COND == 2) {
#pragma unroll
    for (int bm = 0; bm < NBM; ++bm) {
      const int m = s0 + 32 * bm + r;
      if (m >= 0 && m < R) { const float2 v = *reinterpret_cast<const float2*>(a.aff_sig + 2 * (size_t)m); sg[bm][0] = v.x; sg[bm][1] = v.y; }
    }
  }

  for (int layer = 0; layer < n_layers; ++layer) {
    const bool last = layer == n_layers - 1;
    const bf16_t* W1 = static_cast<const bf16_t*>(pick(a.w_in, layer));
    const bf16_t* W2 = static_cast<const bf16_t*>(pick(a.w_res, layer));
    const float* bias1 = Bs + layer * 2 * H;
    const float* bias2 = Bs + NLMAX * 2 * H + layer * H;
    bf16_t* Tt = static_cast<bf16_t*>(pick(a.gate_t, layer));
    bf16_t* Ss = static_cast<bf16_t*>(pick(a.gate_s, layer));
    bf16_t* xo = static_cast<bf16_t*>(pick(a.x_out, layer));
    const uint32_t seed = (a.drop_seed + (uint32_t)layer) ^ seed_x;

    f32x16_t acc[3][NBM];
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int bm = 0; bm < NBM; ++bm)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[bn][bm][e] = 0.0f;

    auto w_load = [&](int it, uint4 (&dst)[4][3]) { w_load_l(W1, it, dst); };
#if !WNS_FIXA
#pragma unroll
    for (int p = 0; p < RINGv - 1; ++p) w_load(p, ring[p]);
#endif
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int kg = it / TAPS, tap = it - kg * TAPS;
      w_load(it + RINGv - 1 < NIT ? it + RINGv - 1 : NIT - 1, ring[(it + RINGv - 1) % RINGv]);
      __builtin_amdgcn_sched_barrier(0);       // keep the prefetch RINGv - 1 steps ahead
      const bf16_t* xsb = Xc + (r + tap) * AP + 8 * h + kg * 64;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8_t bx[NBM];
#pragma unroll
        for (int bm = 0; bm < NBM; ++bm) bx[bm] = *reinterpret_cast<const bf16x8_t*>(xsb + 32 * bm * AP + ks * 16);
#pragma unroll
        for (int bn = 0; bn < 3; ++bn)
#pragma unroll
          for (int bm = 0; bm < NBM; ++bm)
            acc[bn][bm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RINGv][ks][bn]), bx[bm], acc[bn][bm], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    PH(2 + 6 * layer);
    // the thread index as a value formed HERE: the copy loops' per-thread addresses and the epilogue's 24 per-channel hash bases
    // are otherwise computed in front of the layer loop and carried (spilled) across the K loops
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int r_e = tid_e & 31, h_e = (tid_e >> 5) & 1, wave_e = tid_e >> 6;
    // second-stage weights start flying now, under the gate epilogue
    const int wn2 = wave & 1, wm2 = wave >> 1;
    const bool st2 = wm2 < NBM;                // the residual stage's 2 x 2 waves are (row block, column half): NBM = 1 leaves two of them out
    constexpr int R2 = KK2 / 2;
    uint4 ring2[R2][3];
    if (!last && st2) {
#pragma unroll
      for (int kk = 0; kk < R2; ++kk)
#pragma unroll
        for (int bn = 0; bn < 3; ++bn) ring2[kk][bn] = ldfrag(W2, (3 * wn2 + bn) * KK2 + kk, lane);
    }

    // gate epilogue in registers (see wn_layer.hip): block 3*wave + bn holds [16 tanh | 16 sigmoid] of channels 16*(3*wave+bn)..+15
    bf16_t* acts = static_cast<bf16_t*>(a.acts) + layer * H;
    const float* cond = COND == 1 ? a.cond + (size_t)layer * 2 * H : nullptr;
    // affine conditioning of this layer: column 2H * layer of the squeezed cond_layer1 output = parity par, parameters off ..
    const int aff_O = H * n_layers, aff_par = (2 * H * layer) / aff_O, aff_off = (2 * H * layer) - aff_par * aff_O;
    const float* Aw = Bs + FWD_BIAS_FLOATS + aff_off;
    const float* Ab = Aw + aff_O;
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int c = 16 * (3 * wave_e + bn) + 8 * g + 4 * h_e;
        const float4 bt = *reinterpret_cast<const float4*>(bias1 + c), bs = *reinterpret_cast<const float4*>(bias1 + H + c);
        const float btv[4] = {bt.x, bt.y, bt.z, bt.w}, bsv[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
        for (int bm = 0; bm < NBM; ++bm) {
          const int t = 32 * bm + r_e, m = s0 + t;
          float ctv[4] = {}, csv[4] = {};
          if (COND == 1) {
            const float* cp = cond + (size_t)cb[bm] * a.ldc + c;
            const float4 ct = *reinterpret_cast<const float4*>(cp), cs = *reinterpret_cast<const float4*>(cp + H);
            ctv[0] = ct.x; ctv[1] = ct.y; ctv[2] = ct.z; ctv[3] = ct.w; csv[0] = cs.x; csv[1] = cs.y; csv[2] = cs.z; csv[3] = cs.w;
          }
          if (COND == 2) {
            const float pv = aff_par ? sg[bm][1] : sg[bm][0];
            const float4 wt = *reinterpret_cast<const float4*>(Aw + c), ws = *reinterpret_cast<const float4*>(Aw + H + c);
            const float4 bt2 = *reinterpret_cast<const float4*>(Ab + c), bs2 = *reinterpret_cast<const float4*>(Ab + H + c);
            // torch.addcmul(b, contour, w) of the host-side form: b + contour * w, one rounding each way it is written
            ctv[0] = bt2.x + pv * wt.x; ctv[1] = bt2.y + pv * wt.y; ctv[2] = bt2.z + pv * wt.z; ctv[3] = bt2.w + pv * wt.w;
            csv[0] = bs2.x + pv * ws.x; csv[1] = bs2.y + pv * ws.y; csv[2] = bs2.z + pv * ws.z; csv[3] = bs2.w + pv * ws.w;
          }
          float tt[4], ss[4], aa[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float vt = acc[bn][bm][4 * g + j] + btv[j], vs = acc[bn][bm][4 * g + j + 8] + bsv[j];
            if (DROP && !(WNS_EXP & 4)) {                              // x_in = drop(conv(x)) (modules.py:153)
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
    PH(3 + 6 * layer);
    WNS_BARRIER();                                                   // At, Tl, Sl complete
    PH(4 + 6 * layer);
    auto store_gate_tiles = [&]() {
      if (WNS_EXP & 1) return;
      // the rows this workgroup owns are ONE contiguous block of T / S (and 384-byte pieces of the acts rows): 16 bytes per lane,
      // consecutive lanes on consecutive addresses; all the LDS reads first, then the stores (as a rolled loop every chunk paid its
      // own LDS round trip)
      constexpr int NC = (BM * CPR + 255) / 256, NP = 2;               // <= 6 chunks per thread and tile, in passes of NP
#pragma unroll 1
      for (int i0 = 0; i0 < NC; i0 += NP) {
        uint4 vt[NP], vs[NP], va[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int idx = tid_e + 256 * (i0 + i), row = idx / CPR, c8 = idx - row * CPR, t = halo + row;
          vt[i] = vs[i] = va[i] = make_uint4(0, 0, 0, 0);
          if (idx < own * CPR) {
            vt[i] = *reinterpret_cast<const uint4*>(Tl + t * AP + c8 * 8); vs[i] = *reinterpret_cast<const uint4*>(Sl + t * AP + c8 * 8);
            va[i] = *reinterpret_cast<const uint4*>(At + t * AP + c8 * 8);
          }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int idx = tid_e + 256 * (i0 + i), row = idx / CPR, c8 = idx - row * CPR, m = s0 + halo + row;
          if (idx < own * CPR && m < R) {
            *reinterpret_cast<uint4*>(Tt + (size_t)m * H + c8 * 8) = vt[i];
            *reinterpret_cast<uint4*>(Ss + (size_t)m * H + c8 * 8) = vs[i];
            *reinterpret_cast<uint4*>(acts + (size_t)m * a.ldacts + c8 * 8) = va[i];
          }
        }
      }
    };
    if (!WNS_FIXA || last) store_gate_tiles();
    if (last) break;

    // stage 2: x_next = (x + acts @ W_res^T + b_res) * mask -> the next layer's LDS tile (+ HBM for the owned rows)
    f32x16_t acc2[3];
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[bn][e] = 0.0f;
    if (st2) {
      const bf16_t* ab = At + (32 * wm2 + r) * AP + 8 * h;
#pragma unroll
      for (int kk = 0; kk < KK2; ++kk) {
        const bf16x8_t bfm = *reinterpret_cast<const bf16x8_t*>(ab + kk * 16);
#pragma unroll
        for (int bn = 0; bn < 3; ++bn) {
          acc2[bn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring2[kk % R2][bn]), bfm, acc2[bn], 0, 0, 0);
          if (R2 < KK2 && kk + R2 < KK2) ring2[kk % R2][bn] = ldfrag(W2, (3 * wn2 + bn) * KK2 + kk + R2, lane);
        }
      }
    }
    PH(5 + 6 * layer);
#if WNS_FIXA
    {                                          // the next layer's first ring steps, ahead of the x_next stores
      const bf16_t* W1n = static_cast<const bf16_t*>(pick(a.w_in, layer + 1));
#pragma unroll
      for (int p = 0; p < RINGv - 1; ++p) w_load_l(W1n, p, ring[p]);
    }
    store_gate_tiles();                        // T, S, acts: the tiles are untouched until the next layer's epilogue
#endif
    if (st2) {
      const int t = 32 * wm2 + r;
      const float rm = rm2;
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
    WNS_BARRIER();                                                   // the next layer's input is complete; Xc and At are free
    PH(6 + 6 * layer);
    bf16_t* tmp = Xc; Xc = Xn; Xn = tmp;
    if (!(WNS_EXP & 1)) {                                              // x_next of the owned rows, whole rows from the finished tile
      constexpr int NC = (BM * CPR + 255) / 256, NP = 3;
#pragma unroll 1
      for (int i0 = 0; i0 < NC; i0 += NP) {
        uint4 vx[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int idx = tid_e + 256 * (i0 + i), row = idx / CPR, c8 = idx - row * CPR;
          vx[i] = make_uint4(0, 0, 0, 0);
          if (idx < own * CPR) vx[i] = *reinterpret_cast<const uint4*>(Xc + (halo + row + 2) * AP + c8 * 8);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int idx = tid_e + 256 * (i0 + i), row = idx / CPR, c8 = idx - row * CPR, m = s0 + halo + row;
          if (idx < own * CPR && m < R) *reinterpret_cast<uint4*>(xo + (size_t)m * H + c8 * 8) = vx[i];
        }
      }
    }
    if (threadIdx.x < 4 * 24) {                                        // rows 0, 1, 66, 67 of the tile after next
      const int q = threadIdx.x / 24, c8 = threadIdx.x - q * 24, u = q < 2 ? q : XRv - 4 + q;
      *reinterpret_cast<uint4*>(Xn + u * AP + c8 * 8) = make_uint4(0, 0, 0, 0);
    }
    PH(7 + 6 * layer);
  }
  PH(40);
  if (a.stamps && threadIdx.x == 0) {
    unsigned long long* slot = a.stamps + 2 * (a.stamp_slot + (a.stamp_base ? *a.stamp_base : 0));
    if (blockIdx.x == 0) slot[0] = t_begin;
    atomicMax(slot + 1, (unsigned long long)wall_clock64());
  }
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

template <bool DROP>
__device__ __forceinline__ void gate_bwd4(const float (&dd)[4], const float (&t)[4], const float (&s)[4], uint32_t seed, int m, int n,
                                          uint32_t thresh, float scale, uint2& pt, uint2& ps, uint2& ct, uint2& cs)
{
  float gt[4], gs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { gt[j] = dd[j] * s[j] * (1.0f - t[j] * t[j]); gs[j] = dd[j] * t[j] * s[j] * (1.0f - s[j]); }
  ct = pack4(gt[0], gt[1], gt[2], gt[3]); cs = pack4(gs[0], gs[1], gs[2], gs[3]);      // before the dropout mask: d cond
  if (DROP) {
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

constexpr int RB = 2 * RING - 1;               // backward ring: half the fragments per step, twice the depth for the same registers
constexpr int B_NS = 2 * H / 64, B_NIT = B_NS * TAPS, B_KS = 2 * H / 16, B_NBT = H / 32;  // 6 slices, 30 steps, 24 k-steps per tap, 6 blocks

// step `it` of a data-gradient conv's weight stream: this wave's column half (wn) and K half (wk)
__device__ __forceinline__ void bwd_w_load(const bf16_t* W1, int it, int wn, int wk, int lane, uint4 (&dst)[2][3])
{
  const int slice = it / TAPS, tap = it - slice * TAPS;
  const bf16_t* Wt = pinned(W1 + (size_t)((tap * B_NBT + 3 * wn) * B_KS + slice * 4 + 2 * wk) * 512);
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
    for (int bn = 0; bn < 3; ++bn) dst[k2][bn] = ldfrag(Wt, bn * B_KS + k2, lane);
}
// the first RB - 1 steps: issued by the PREVIOUS step (or the kernel's head) ahead of its stores — vmcnt retires in order, so a
// wait for these fragments would otherwise also wait for every store issued before them (see the forward kernel)
template <int RBv>
__device__ __forceinline__ void bwd_ring_prologue(const void* w, int lane, uint4 (&ring)[RBv][2][3])
{
  const int wave = wave_scalar(), wn = wave & 1, wk = wave >> 1;
  const bf16_t* W1 = pinned(static_cast<const bf16_t*>(w));
#pragma unroll
  for (int p = 0; p < RBv - 1; ++p) bwd_w_load(W1, p, wn, wk, lane, ring[p]);
}

// one step j of the chain (compile-time j: every pointer is a kernel argument, every fragment address is formed where it is used)
template <int J, bool COND, bool DROP, int NBM>
__device__ __forceinline__ void bwd_step(const gt_wn_stack_bwd_args& a, uint32_t drop_thresh, float drop_scale, uint32_t seed_x,
                                         bf16_t* Dt, float* Ex, bf16_t* At, int s0, int halo, int lane, int wave,
                                         uint4 (&ring)[NBM == 1 ? WNS_RB1 : RB][2][3], float rm)
{
  const int r = lane & 31, h = lane >> 5;
  wave = wave_scalar();
  const int wn = wave & 1, wk = wave >> 1;     // stage 1: column half, K half;  afterwards wk doubles as the row half
  constexpr int BMv = 32 * NBM;                // rows of the tile (NBM: see the forward kernel)
  constexpr int RBv = NBM == 1 ? WNS_RB1 : RB;
  const bool act = wk < NBM;                   // NBM = 1: one row block — the waves of K half 1 hand their sums over and sit the rest out
  const int n_layers = a.n_layers, R = a.R;
  const bf16_t* via = static_cast<const bf16_t*>(a.via_skip);
  constexpr int NIT = B_NIT;
  const bf16_t* W1 = pinned(static_cast<const bf16_t*>(a.w_in_d[J]));
  f32x16_t acc[3][NBM];
#pragma unroll
  for (int bn = 0; bn < 3; ++bn)
#pragma unroll
    for (int bm = 0; bm < NBM; ++bm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[bn][bm][e] = 0.0f;
  auto w_load = [&](int it, uint4 (&dst)[2][3]) { bwd_w_load(W1, it, wn, wk, lane, dst); };
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int slice = it / TAPS, tap = it - slice * TAPS;
    w_load(it + RBv - 1 < NIT ? it + RBv - 1 : NIT - 1, ring[(it + RBv - 1) % RBv]);
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t* xsb = Dt + (r + tap) * DP + 8 * h + slice * 64 + 32 * wk;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      bf16x8_t bx[NBM];
#pragma unroll
      for (int bm = 0; bm < NBM; ++bm) bx[bm] = *reinterpret_cast<const bf16x8_t*>(xsb + 32 * bm * DP + k2 * 16);
#pragma unroll
      for (int bn = 0; bn < 3; ++bn)
#pragma unroll
        for (int bm = 0; bm < NBM; ++bm)
          acc[bn][bm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[it % RBv][k2][bn]), bx[bm], acc[bn][bm], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  PH(2 + 8 * (3 - J));
  // second-stage weights (the layer below): this wave's 96 columns (wn), rows 32*wk
  constexpr int JL = J > 0 ? J - 1 : 0;
  const bf16_t* W2 = pinned(static_cast<const bf16_t*>(a.w_res_d[JL]));
  uint4 ring2[KK2][3];                         // all of them now: reloaded inside the 36-MFMA loop they arrived after it needed them
  if (J > 0 && act) {
#pragma unroll
    for (int kk = 0; kk < KK2; ++kk)
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) ring2[kk][bn] = ldfrag(W2, (3 * wn + bn) * KK2 + kk, lane);
  }
  // the gate backward's three row operands (skip-path gradient, saved tanh, saved sigmoid of layer j-1, all 64 rows of the tile) are
  // fetched as whole rows — read in the MFMA layout, every load instruction touched 16 bytes in each of 32 rows — and handed to the
  // epilogue through LDS; they are issued here, ahead of this step's stores
  constexpr int JLp = J > 0 ? J - 1 : 0;
  constexpr int NPRE = (BM * CPR + 255) / 256;                       // 6 chunks per thread and operand
  uint4 pre[3][NPRE];
  if (J > 0) {
    const bf16_t* Tg = static_cast<const bf16_t*>(a.gate_t[JLp]);
    const bf16_t* Sg = static_cast<const bf16_t*>(a.gate_s[JLp]);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = threadIdx.x + 256 * i, row = idx / CPR, c8 = idx - row * CPR, mm = s0 + row;
      pre[0][i] = pre[1][i] = pre[2][i] = make_uint4(0, 0, 0, 0);
      if (row < BMv && mm >= 0 && mm < R) {
        pre[0][i] = *reinterpret_cast<const uint4*>(via + (size_t)mm * a.ldvs + JLp * H + c8 * 8);
        pre[1][i] = *reinterpret_cast<const uint4*>(Tg + (size_t)mm * H + c8 * 8);
        pre[2][i] = *reinterpret_cast<const uint4*>(Sg + (size_t)mm * H + c8 * 8);
      }
    }
  }
  WNS_BARRIER();                            // every wave is done with the d pre tile: the exchange buffer may overwrite it

  // K halves meet: a wave keeps row block bm == wk and hands the other one to its partner (same columns, other K half)
  if (NBM == 2 || wk) {
    float* mine = Ex + wave * (3 * 16 * 64);
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int e = 0; e < 16; ++e) mine[(bn * 16 + e) * 64 + lane] = (NBM == 2 && !wk) ? acc[bn][NBM - 1][e] : acc[bn][0][e];
  }
  WNS_BARRIER();
  f32x16_t sum[3];
  if (act) {
    const float* theirs = Ex + (wave ^ 2) * (3 * 16 * 64);
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum[bn][e] = ((NBM == 2 && wk) ? acc[bn][NBM - 1][e] : acc[bn][0][e]) + theirs[(bn * 16 + e) * 64 + lane];
  }
  PH(3 + 8 * (3 - J));
  // dX_j = (conv^T(d pre_j) + dX_{j+1}) * mask -> HBM (owned rows) and the stage-2 tile (which still holds dX_{j+1})
  const int t = 32 * wk + r, m = s0 + t;
  const bool mine_row = act && t >= halo && t < BMv - halo && m < R;
  if (act) {
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
  WNS_BARRIER();
  PH(4 + 8 * (3 - J));
  auto store_dx = [&]() {
    // dX_j of the owned rows leaves as whole rows from the finished tile (the MFMA layout gives a store 16 bytes in each of 32 rows)
    bf16_t* dx = static_cast<bf16_t*>(a.dx[J]);
    const int own = BMv - 2 * halo;
    constexpr int NC = (BM * CPR + 255) / 256;
    uint4 vx[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int idx = threadIdx.x + 256 * i, row = idx / CPR, c8 = idx - row * CPR;
      vx[i] = make_uint4(0, 0, 0, 0);
      if (idx < own * CPR) vx[i] = *reinterpret_cast<const uint4*>(At + (halo + row) * AP + c8 * 8);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int idx = threadIdx.x + 256 * i, row = idx / CPR, c8 = idx - row * CPR, mm = s0 + halo + row;
      if (idx < own * CPR && mm < R) *reinterpret_cast<uint4*>(dx + (size_t)mm * H + c8 * 8) = vx[i];
    }
  };
  if (J == 0) { store_dx(); return; }                               // bottom: dX_0 is the gradient at the WaveNet's input

  // d acts_{j-1} = dX_j W_res + skip-path gradient -> gate backward -> d pre_{j-1}: the next tile (+ HBM for the owned rows)
  bf16_t* Vl = At + BM * AP; bf16_t* Tl = Vl + BM * AP; bf16_t* Sl = Tl + BM * AP;
  f32x16_t acc2[3];
#pragma unroll
  for (int bn = 0; bn < 3; ++bn)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[bn][e] = 0.0f;
  if (act) {
    const bf16_t* ab = At + (32 * wk + r) * AP + 8 * h;
#pragma unroll
    for (int kk = 0; kk < KK2; ++kk) {
      const bf16x8_t bfm = *reinterpret_cast<const bf16x8_t*>(ab + kk * 16);
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) {
        acc2[bn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring2[kk][bn]), bfm, acc2[bn], 0, 0, 0);
      }
    }
  }
  PH(5 + 8 * (3 - J));
  bwd_ring_prologue(a.w_in_d[JL], lane, ring);                      // the next step's first fragments, then this step's stores
  store_dx();                                                       // (the dX tile stays as it is until the next step's epilogue)
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int idx = threadIdx.x + 256 * i, row = idx / CPR, c8 = idx - row * CPR;
    if (row >= BMv) continue;
    *reinterpret_cast<uint4*>(Vl + row * AP + c8 * 8) = pre[0][i];
    *reinterpret_cast<uint4*>(Tl + row * AP + c8 * 8) = pre[1][i];
    *reinterpret_cast<uint4*>(Sl + row * AP + c8 * 8) = pre[2][i];
  }
  WNS_BARRIER();
  PH(6 + 8 * (3 - J));
  if (act) {
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
          gate_bwd4<DROP>(dd, tt, sg, seed, m, n, drop_thresh, drop_scale, pt, ps, ct, cs);
          if (COND && mine_row) {                                    // (speaker-conditioned configs only)
            *reinterpret_cast<uint2*>(dpre_c + (size_t)m * 2 * H + n) = ct;
            *reinterpret_cast<uint2*>(dpre_c + (size_t)m * 2 * H + H + n) = cs;
          }
        }
        *reinterpret_cast<uint2*>(Dt + (t + 2) * DP + n) = pt;
        *reinterpret_cast<uint2*>(Dt + (t + 2) * DP + H + n) = ps;
      }
  }
  PH(7 + 8 * (3 - J));
  WNS_BARRIER();                                                   // the next conv's input tile is complete
  PH(8 + 8 * (3 - J));
  {
    // d pre_{j-1} of the owned rows: whole 768-byte rows from the tile (read-only until the next step's exchange, which follows a barrier)
    bf16_t* dpre = static_cast<bf16_t*>(a.dpre[JL]);
    const int own = BMv - 2 * halo;
    constexpr int CPR2 = 2 * H / 8, NC = (BM * CPR2 + 255) / 256, NP = 4;      // 12 chunks per thread, in passes of NP
#pragma unroll
    for (int i0 = 0; i0 < NC; i0 += NP) {
      uint4 vx[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int idx = threadIdx.x + 256 * (i0 + i), row = idx / CPR2, c8 = idx - row * CPR2;
        vx[i] = make_uint4(0, 0, 0, 0);
        if (idx < own * CPR2) vx[i] = *reinterpret_cast<const uint4*>(Dt + (halo + row + 2) * DP + c8 * 8);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int idx = threadIdx.x + 256 * (i0 + i), row = idx / CPR2, c8 = idx - row * CPR2, mm = s0 + halo + row;
        if (idx < own * CPR2 && mm < R) *reinterpret_cast<uint4*>(dpre + (size_t)mm * 2 * H + c8 * 8) = vx[i];
      }
    }
  }
  PH(9 + 8 * (3 - J));
}

template <bool COND, bool DROP, int NBM>
__global__ __launch_bounds__(256) void gt_wn_stack_bwd_kernel(gt_wn_stack_bwd_args a, uint32_t drop_thresh, float drop_scale)
{
  constexpr int BMv = 32 * NBM, XRv = BMv + TAPS - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PH(0);
  const uint32_t seed_x = a.seed_dev ? *a.seed_dev : 0u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_layers = a.n_layers, R = a.R;
  const int halo = 2 * (n_layers - 1), own = BMv - 2 * halo;
  const int s0 = blockIdx.x * own - halo;
  bf16_t* Dt = reinterpret_cast<bf16_t*>(smem);
  float* Ex = reinterpret_cast<float*>(smem);
  bf16_t* At = reinterpret_cast<bf16_t*>(smem + BWD_DT);
  const bf16_t* via = static_cast<const bf16_t*>(a.via_skip);
  constexpr int RBv = NBM == 1 ? WNS_RB1 : RB;
  uint4 ring[RBv][2][3];
  bwd_ring_prologue(pick(a.w_in_d, n_layers - 1), lane, ring);      // ahead of the head's d pre stores
  float rm;                                                         // the row mask of the row this lane finishes in every dX epilogue
  {
    const int m = s0 + 32 * (wave >> 1) + (lane & 31);
    rm = (m >= 0 && m < R) ? a.rowmask[m] : 0.0f;
  }

  // top layer: d acts = the skip-path gradient only -> d pre on all 68 rows of the tile (row-local).  Whole rows, 16 bytes per lane,
  // EVERY load issued before the first use: as a loop of 8-byte loads, each waited for where it was used, this head was 13 serial
  // L2 round trips = 30 k of the launch's 189 k cycles (tools/wn_stack_phases.py).
  {
    const int L = n_layers - 1;
    const bf16_t* Tt = static_cast<const bf16_t*>(pick(a.gate_t, L));
    const bf16_t* Ss = static_cast<const bf16_t*>(pick(a.gate_s, L));
    bf16_t* dpre = static_cast<bf16_t*>(pick(a.dpre, L));
    bf16_t* dpre_c = static_cast<bf16_t*>(pick(a.dpre_c, L));
    const uint32_t seed = (a.drop_seed + (uint32_t)L) ^ seed_x;
    constexpr int NH = (XR * CPR + 255) / 256;                       // 7 chunks of 8 channels per thread
    uint4 hv[3][NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int idx = threadIdx.x + 256 * i, u = idx / CPR, c8 = idx - u * CPR, m = s0 - 2 + u;
      hv[0][i] = hv[1][i] = hv[2][i] = make_uint4(0, 0, 0, 0);
      if (u < XRv && m >= 0 && m < R) {
        hv[0][i] = *reinterpret_cast<const uint4*>(via + (size_t)m * a.ldvs + L * H + c8 * 8);
        hv[1][i] = *reinterpret_cast<const uint4*>(Tt + (size_t)m * H + c8 * 8);
        hv[2][i] = *reinterpret_cast<const uint4*>(Ss + (size_t)m * H + c8 * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int idx = threadIdx.x + 256 * i, u = idx / CPR, c8 = idx - u * CPR, m = s0 - 2 + u, n = c8 * 8;
      if (u >= XRv) continue;
      uint2 pt[2], ps[2], ct[2], cs[2];
      pt[0] = pt[1] = ps[0] = ps[1] = ct[0] = ct[1] = cs[0] = cs[1] = make_uint2(0, 0);
      if (m >= 0 && m < R) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float dd[4], t[4], sg[4];
          unpack4(q ? make_uint2(hv[0][i].z, hv[0][i].w) : make_uint2(hv[0][i].x, hv[0][i].y), dd);
          unpack4(q ? make_uint2(hv[1][i].z, hv[1][i].w) : make_uint2(hv[1][i].x, hv[1][i].y), t);
          unpack4(q ? make_uint2(hv[2][i].z, hv[2][i].w) : make_uint2(hv[2][i].x, hv[2][i].y), sg);
          gate_bwd4<DROP>(dd, t, sg, seed, m, n + 4 * q, drop_thresh, drop_scale, pt[q], ps[q], ct[q], cs[q]);
        }
        if (u - 2 >= halo && u - 2 < BMv - halo) {
          *reinterpret_cast<uint4*>(dpre + (size_t)m * 2 * H + n) = make_uint4(pt[0].x, pt[0].y, pt[1].x, pt[1].y);
          *reinterpret_cast<uint4*>(dpre + (size_t)m * 2 * H + H + n) = make_uint4(ps[0].x, ps[0].y, ps[1].x, ps[1].y);
          if (COND) {
            *reinterpret_cast<uint4*>(dpre_c + (size_t)m * 2 * H + n) = make_uint4(ct[0].x, ct[0].y, ct[1].x, ct[1].y);
            *reinterpret_cast<uint4*>(dpre_c + (size_t)m * 2 * H + H + n) = make_uint4(cs[0].x, cs[0].y, cs[1].x, cs[1].y);
          }
        }
      }
      *reinterpret_cast<uint4*>(Dt + u * DP + n) = make_uint4(pt[0].x, pt[0].y, pt[1].x, pt[1].y);
      *reinterpret_cast<uint4*>(Dt + u * DP + H + n) = make_uint4(ps[0].x, ps[0].y, ps[1].x, ps[1].y);
    }
  }
  WNS_BARRIER();
  PH(1);
  // the chain, top to bottom (workgroup-uniform branches; each step is its own straight-line code)
  if (n_layers > 3) bwd_step<3, COND, DROP, NBM>(a, drop_thresh, drop_scale, seed_x, Dt, Ex, At, s0, halo, lane, wave, ring, rm);
  if (n_layers > 2) bwd_step<2, COND, DROP, NBM>(a, drop_thresh, drop_scale, seed_x, Dt, Ex, At, s0, halo, lane, wave, ring, rm);
  if (n_layers > 1) bwd_step<1, COND, DROP, NBM>(a, drop_thresh, drop_scale, seed_x, Dt, Ex, At, s0, halo, lane, wave, ring, rm);
  bwd_step<0, COND, DROP, NBM>(a, drop_thresh, drop_scale, seed_x, Dt, Ex, At, s0, halo, lane, wave, ring, rm);
  PH(40);
}

inline bool al16(const void* p) { return !((uintptr_t)p & 15); }

}  // namespace

#if WNS_PHASES
extern "C" int gt_dev_wns_phases(void* dst, size_t bytes)
{
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wns_ph), bytes < sizeof(g_wns_ph) ? bytes : sizeof(g_wns_ph)) == hipSuccess ? 0 : -1;
}
#endif

#ifndef WNS_SMALL_TILE_MAX_WGS
#define WNS_SMALL_TILE_MAX_WGS 80             // 64-row tiles give at most this many workgroups -> 32-row tiles (needs n_layers <= 4: 20 owned rows)
#endif
// 32-row blocks per tile for a launch over R rows: 2, or 1 when the 64-row tiling would leave two thirds of the 256 CUs idle.  Measured
// back to back (tools/wn_layer_bench.py stack, WN_BENCH_TY=..., round 3; forward / backward us per launch, 64-row -> 32-row tiles):
// 3 072 rows 61.7 / 71.4 -> 51.2 / 59.3;  3 584 rows 62.0 / 72.9 -> 51.7 / 61.0;  5 120 rows 63.2 / 75.6 -> 57.1 / 68.2 (and nothing left
// of it inside the cfg 4 / cfg 5 steps, whose other branch then finds no free CU: the threshold stays below that);  9 728 rows: 2 rounds.
#ifndef WNS_SMALL_TILE_MAX_WGS_FWD
#define WNS_SMALL_TILE_MAX_WGS_FWD 98         // the FORWARD takes the 32-row form up to here (2.6 x 98 = 255 workgroups: one round of the
                                              // chip; measured at 128, which cfg 4 / cfg 5's 85 / 89 tiles are below either way): with cfg 5's final schedule its decoder forward runs
#endif                                        // alone on the machine for 2.6 ms (14.55-14.64 -> 14.24 ms per step); the backward, which always
                                              // shares it with the predictors' backward, at the same threshold: 15.75 ms
static int stack_row_blocks(int R, int n_layers, bool fwd = false)
{
  const int own2 = BM - 4 * (n_layers - 1), own1 = 32 - 4 * (n_layers - 1);
  return (own1 >= 8 && (R + own2 - 1) / own2 <= (fwd ? WNS_SMALL_TILE_MAX_WGS_FWD : WNS_SMALL_TILE_MAX_WGS)) ? 1 : 2;
}
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
  const bool affine = a.aff_w || a.aff_b || a.aff_sig;
  if (affine) {
    if (a.cond || !a.aff_w || !a.aff_b || !a.aff_sig || ((H * a.n_layers) % (2 * H))) return GT_E_INVAL;   // (O must hold whole layers: n even)
    if (!al16(a.aff_w) || !al16(a.aff_b) || ((uintptr_t)a.aff_sig & 7)) return GT_E_ALIGN;
  }
  uint32_t thresh = 0; float scale = 1.0f;
  if (a.drop_p > 0.0f) {
    if (a.drop_p >= 1.0f) return GT_E_UNSUPPORTED;
    thresh = (uint32_t)((double)a.drop_p * 4294967296.0); scale = 1.0f / (1.0f - a.drop_p);
  }
  typedef void (*kern_t)(gt_wn_stack_fwd_args, uint32_t, float);
  static const kern_t kerns[12] = {gt_wn_stack_fwd_kernel<0, false, 2>, gt_wn_stack_fwd_kernel<0, true, 2>, gt_wn_stack_fwd_kernel<1, false, 2>,
                                   gt_wn_stack_fwd_kernel<1, true, 2>, gt_wn_stack_fwd_kernel<2, false, 2>, gt_wn_stack_fwd_kernel<2, true, 2>,
                                   gt_wn_stack_fwd_kernel<0, false, 1>, gt_wn_stack_fwd_kernel<0, true, 1>, gt_wn_stack_fwd_kernel<1, false, 1>,
                                   gt_wn_stack_fwd_kernel<1, true, 1>, gt_wn_stack_fwd_kernel<2, false, 1>, gt_wn_stack_fwd_kernel<2, true, 1>};
  static bool attr = false;                    // > 64 KB of LDS: opt in once per process
  if (!attr) {
    for (int i = 0; i < 12; ++i)
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[i]), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (i % 6) >= 4 ? STACK_FWD_LDS_AFF : STACK_FWD_LDS) != hipSuccess)
        return GT_E_LAUNCH;
    attr = true;
  }
  const int nbm = stack_row_blocks(a.R, a.n_layers, true);
  const int own = 32 * nbm - 4 * (a.n_layers - 1);
  const dim3 grid((a.R + own - 1) / own), block(256);
  const int mode = affine ? 2 : (a.cond ? 1 : 0);
  hipLaunchKernelGGL(kerns[(nbm == 1 ? 6 : 0) + 2 * mode + (thresh ? 1 : 0)], grid, block, affine ? STACK_FWD_LDS_AFF : STACK_FWD_LDS,
                     static_cast<hipStream_t>(stream), a, thresh, scale);
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
  bool cond = false;                           // d cond outputs: all layers or none
  for (int i = 0; i < a.n_layers; ++i) cond = cond || a.dpre_c[i];
  for (int i = 0; i < a.n_layers; ++i) if (cond && !a.dpre_c[i]) return GT_E_INVAL;
  typedef void (*kern_t)(gt_wn_stack_bwd_args, uint32_t, float);
  static const kern_t kerns[8] = {gt_wn_stack_bwd_kernel<false, false, 2>, gt_wn_stack_bwd_kernel<false, true, 2>,
                                  gt_wn_stack_bwd_kernel<true, false, 2>, gt_wn_stack_bwd_kernel<true, true, 2>,
                                  gt_wn_stack_bwd_kernel<false, false, 1>, gt_wn_stack_bwd_kernel<false, true, 1>,
                                  gt_wn_stack_bwd_kernel<true, false, 1>, gt_wn_stack_bwd_kernel<true, true, 1>};
  static bool attr = false;
  if (!attr) {
    for (kern_t k : kerns)
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, SBWD_LDS) != hipSuccess)
        return GT_E_LAUNCH;
    attr = true;
  }
  const int nbm = stack_row_blocks(a.R, a.n_layers);
  const int own = 32 * nbm - 4 * (a.n_layers - 1);
  const dim3 grid((a.R + own - 1) / own), block(256);
  hipLaunchKernelGGL(kerns[(nbm == 1 ? 4 : 0) + (cond ? 2 : 0) + (thresh ? 1 : 0)], grid, block, SBWD_LDS, static_cast<hipStream_t>(stream), a, thresh, scale);
  return gt_launch_status(__func__);
}
