// Implicit-GEMM 1-D convolution, second generation: LDS-DMA rings, fragment-ordered weights.
//
// Same contract as gt_conv_gemm_bf16 (conv_gemm.hip; the conv1d calls of modules.py:152,165,
// attentions.py:144,172,232-238,365-371 and their data gradients), for shapes with Cin % 64 == 0 and
// N % 96 == 0 — the WaveNet in_layer (192 -> 384, k = 5) and its data gradient (384 -> 192), the 1x1
// res / skip / q / k / v / o GEMMs, the FFN convs.  What is different, and why (measured on MI355X, cfg2
// decoder shape R = 12 928 rows):
//
//  * The first-generation loop stages W and X through registers into padded LDS with ONE step of lookahead
//    and a barrier every 256-512 MFMA cycles; a lone workgroup per CU (303 workgroups on 256 CUs) then waits
//    ~0.7 us of L2 latency per step.  Here both operands travel HBM/L2 -> LDS by LDS-DMA
//    (global_load_lds_dwordx4: no VGPR staging, no ds_write) into rings that run NSW-1 = 3 stages ahead of
//    the MFMAs behind counted `s_waitcnt vmcnt(N)` and a raw s_barrier.
//  * Weights are packed in MFMA-fragment order ([tap][n/32][k/16][lane][8 bf16], gt_pack_conv_weights flag 2/4):
//    one DMA instruction moves exactly one 1-KB A-fragment, lane i's 16 bytes land at byte 16*i, and the
//    ds_read_b128 that feeds the MFMA is conflict-free without padding.
//  * Activation rows (128 B per 64-channel slice) are DMA'd with the XOR swizzle applied on the SOURCE side
//    (LDS chunk c of row r holds channel chunk c ^ ((r >> 1) & 7)): the 16 rows a ds_read_b128 lane group touches
//    fall in 16 different bank quads for every tap shift.
//  * Workgroup tile 128 x 192 (8 waves) or 128 x 96 (4 waves), wave tile 32 rows x 96 channels: 202 workgroups for
//    the decoder shapes = one resident workgroup per CU, no second round, the whole 160 KB of LDS for the rings.
//  * Epilogue: shared with the first generation (conv_common.h): accumulators -> fp32 LDS tile -> row chunks.
#include <stdlib.h>
#include "common.h"
#include "conv_common.h"
#include "../../include/glowtts_hip.h"

namespace {
using gtconv::ConvArgs;

struct Ring { int nsw, nxb, wstage, xbuf; };                       // ring depths and bytes per stage / buffer

__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)reinterpret_cast<uintptr_t>(p); }

// one wave-wide LDS-DMA: lane i's 16 bytes at gptr(lane) -> LDS[dst + 16*i].  Inline asm: with the builtin
// hipcc drains vmcnt(0) in front of every ds_read of the rings (it cannot tell the slots apart).
__device__ __forceinline__ void dma16(const void* gptr, unsigned dst_uniform)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(dst_uniform) : "memory");
}

__device__ __forceinline__ void wait_vm(int n)                     // at most n of this wave's loads still in flight
{
  switch (n) {
#define C_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
    C_(17) C_(18) C_(19) C_(20) C_(21) C_(22) C_(23) C_(24) C_(25) C_(26) C_(27) C_(28) C_(29) C_(30) C_(31) C_(32)
    C_(33) C_(34) C_(35) C_(36) C_(37) C_(38) C_(39) C_(40) C_(41) C_(42) C_(43) C_(44) C_(45) C_(46) C_(47) C_(48)
#undef C_
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// BMW x BNW waves, wave tile = 32 rows x 96 channels (3 MFMA blocks)  ->  BM = 32*BMW, BN = 96*BNW
// NLD > 0: NLD extra LOADER waves issue every LDS-DMA (an LDS-DMA piece costs its issuing wave 60-185 cycles,
// MI355X_MICROARCH.md cycle constants) and the MFMA waves never touch the vector-memory pipe; NLD == 0: the MFMA
// waves load for themselves.
template <int BMW, int BNW, bool GATE, int NLD>
__global__ __launch_bounds__(64 * (BMW * BNW + NLD), 1) void gt_conv_gemm2_kernel(ConvArgs a, Ring rg)
{
  constexpr int NWV = BMW * BNW, NT = 64 * (NWV + NLD), BM = 32 * BMW, BN = 96 * BNW;
  constexpr int NWL = NLD ? NLD : NWV;                             // waves that issue DMA
  constexpr int NF = (BN / 32) * 4;                                // 1-KB weight fragments per stage (BN x 64 k)
  constexpr int NWF = NF / NWL;                                    // ... per loading wave
  constexpr int NG = (BM + 4 + 7) / 8;                             // 8-row groups of an activation slice (BM + halo rows)
  constexpr int NXG = (NG + NWL - 1) / NWL;                        // ... per loading wave (the tail re-loads the last group)
  static_assert(NF % NWL == 0, "weight fragments must split evenly over the loading waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  if (a.seed_dev) a.drop_seed ^= *a.seed_dev;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = NLD ? wave >= NWV : true, computer = wave < NWV;
  const int lw = NLD ? wave - NWV : wave;                          // index among the loading waves
  const int wn = wave % BNW, wm = wave / BNW;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int taps = a.taps, padl = taps >> 1;
  const int NS = a.Kp / 64, NIT = NS * taps;
  const int D = rg.nsw - 1;                                        // stages the DMA runs ahead of the MFMAs
  const unsigned wbase = lds_off(smem), xbase = wbase + (unsigned)(rg.nsw * rg.wstage);
  const int KS = a.Kp / 16, NBT = a.Np / 32;                       // fragment grid of the packed weights
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(a.W);
  const unsigned char* Xb = reinterpret_cast<const unsigned char*>(a.X);

  // ---- DMA issue of one stage: stage j = (slice, tap); X slice rides along with the first tap of its slice
  auto issue = [&](int j, int slice, int tap) {
    if (tap == 0) {
      const unsigned xb = xbase + (unsigned)((slice % rg.nxb) * rg.xbuf);
#pragma unroll
      for (int i = 0; i < NXG; ++i) {
        int g = lw + NWL * i;  g = g < NG ? g : NG - 1;
        const int row = g * 8 + (lane >> 3);
        const int cg = (lane & 7) ^ ((row >> 1) & 7);              // swizzle on the SOURCE side
        int gm = m0 - padl + row;  gm = gm < 0 ? 0 : (gm >= a.R ? a.R - 1 : gm);
        dma16(Xb + ((size_t)gm * a.ldx + slice * 64 + cg * 8) * 2, __builtin_amdgcn_readfirstlane(xb + (unsigned)(g * 1024)));
      }
    }
    const unsigned wb = wbase + (unsigned)((j % rg.nsw) * rg.wstage);
#pragma unroll
    for (int i = 0; i < NWF; ++i) {
      const int f = lw * NWF + i;                                  // fragment (nb = f / 4, ks = f % 4) of this stage
      const size_t frag = ((size_t)tap * NBT + (n0 >> 5) + (f >> 2)) * KS + slice * 4 + (f & 3);
      dma16(Wb + frag * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(wb + (unsigned)(f * 1024)));
    }
  };
  auto loads_of = [&](int tap) { return NWF + (tap == 0 ? NXG : 0); };

  f32x16_t acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

  // ---- prologue: stages 0 .. D-1 in flight
  int is = 0, it_slice = 0, it_tap = 0;                            // issue cursor
  for (; is < D && is < NIT; ++is) {
    if (loader) issue(is, it_slice, it_tap);
    if (++it_tap == taps) { it_tap = 0; ++it_slice; }
  }

  int slice = 0, tap = 0;                                          // compute cursor
  for (int it = 0; it < NIT; ++it) {
    // loads allowed to stay in flight: those of stages it+1 .. min(it+D-1, NIT-1)
    if (a.exp_ & 2) wait_vm(0);
    else if (loader) {
      int allow = 0, t = tap;
      for (int j = it + 1; j < it + D && j < NIT; ++j) { if (++t == taps) t = 0; allow += loads_of(t); }
      wait_vm(allow);
    }
    __builtin_amdgcn_s_barrier();                                  // stage `it` of every wave has landed; stage it-1 is retired
    asm volatile("" ::: "memory");
    if (is < NIT && !(a.exp_ & 2)) {
      if (loader) issue(is, it_slice, it_tap);
      if (++it_tap == taps) { it_tap = 0; ++it_slice; }
      ++is;
    }
    if (computer) {
      const unsigned char* ws = smem + (it % rg.nsw) * rg.wstage + (wn * 12) * 1024 + lane * 16;
      const unsigned char* xs = smem + rg.nsw * rg.wstage + (slice % rg.nxb) * rg.xbuf;
      const int row = wm * 32 + r + tap;
      const unsigned char* xr = xs + row * 128;
      const int sw = (row >> 1) & 7;
      // all 16 fragment reads of the stage first, then the 12 MFMAs: hipcc otherwise pairs every ds_read with an
      // lgkmcnt(0) in front of its MFMA (one exposed LDS latency per MFMA)
      bf16x8_t bq[4], aq[4][3];
  #pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bq[ks] = *reinterpret_cast<const bf16x8_t*>(xr + (((2 * ks + h) ^ sw) << 4));
  #pragma unroll
        for (int nb = 0; nb < 3; ++nb) aq[ks][nb] = ((a.exp_ & 8) && ks) ? aq[0][nb] : *reinterpret_cast<const bf16x8_t*>(ws + (nb * 4 + ks) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
      for (int ks = 0; ks < 4; ++ks)
  #pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][nb], bq[ks], acc[nb], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
    if (++tap == taps) { tap = 0; ++slice; }
  }
  wait_vm(0);
  __syncthreads();                                                 // every wave is done with the rings
  if (a.exp_ & 1) {                                                // dev: time the main loop alone
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc += acc[i][e];
    if (sacc == 123.456f) static_cast<bf16_t*>(a.Y)[tid] = 1;
    return;
  }

  // ---- epilogue phase 1: accumulators -> fp32 LDS tile [BM][BN] (pitch == 4 mod 64 banks)
  constexpr int EP = BN + 4;
  float* es = reinterpret_cast<float*>(smem);
  if (computer)
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(&es[(wm * 32 + r) * EP + wn * 96 + 32 * nb + 8 * g + 4 * h]) =
          make_float4(acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]);
  __syncthreads();
  if (GATE) gtconv::epilogue_gate<NT, BM, BN>(a, es, EP, m0, n0, tid);
  else      gtconv::epilogue_plain<NT, BM, BN>(a, es, EP, m0, n0, tid);
}

template <int BMW, int BNW, bool GATE, int NLD>
int launch2(const ConvArgs& a, hipStream_t st)
{
  constexpr int BM = 32 * BMW, BN = 96 * BNW;
  constexpr int NF = (BN / 32) * 4, NG = (BM + 4 + 7) / 8;
  const int NIT = (a.Kp / 64) * a.taps;
  Ring rg;
  rg.nsw = NIT < 4 ? (NIT < 2 ? 2 : NIT) : 4;
  { static int fn = -1; if (fn < 0) { const char* e = getenv("GT_CONV2_NSW"); fn = e ? atoi(e) : 0; } if (fn >= 2 && fn <= 6 && fn <= NIT) rg.nsw = fn; }
  const int D = rg.nsw - 1;
  rg.nxb = 1 + (D + a.taps - 1) / a.taps;
  rg.wstage = NF * 1024;
  rg.xbuf = NG * 1024;
  size_t lds = (size_t)rg.nsw * rg.wstage + (size_t)rg.nxb * rg.xbuf;
  const size_t epi = (size_t)BM * (BN + 4) * 4;
  if (lds < epi) lds = epi;
  if (lds > 160 * 1024) return GT_E_UNSUPPORTED;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_conv_gemm2_kernel<BMW, BNW, GATE, NLD>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GT_E_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((gt_conv_gemm2_kernel<BMW, BNW, GATE, NLD>), dim3(a.Np / BN, (a.R + BM - 1) / BM), dim3(64 * (BMW * BNW + NLD)), lds, st, a, rg);
  return gt_launch_status("gt_conv_gemm2_bf16");
}

}  // namespace

extern "C" int gt_conv_gemm2_supported(int N, int Cin, int taps, int gate)
{
  if (taps != 1 && taps != 3 && taps != 5) return 0;
  if (Cin <= 0 || (Cin % 64) || N <= 0 || (N % 96)) return 0;
  if (gate == 1 && (N % 192)) return 0;
  return 1;
}

extern "C" int gt_conv_gemm2_bf16(const void* X, int ldx, const void* Wp, const float* bias,
                                  const float* cond, int ldc, const float* rowmask,
                                  void* Y, int ldy, int out_f32, const void* addend, int ldadd,
                                  void* gate_t, void* gate_s, int ldts,
                                  int R, int N, int Cin, int taps, int Tp, int Np, int Kp,
                                  int relu, int gate, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev,
                                 const int32_t* row0, int B, void* stream)
{
  if (R < 0 || N <= 0 || Cin <= 0) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  if (!X || !Wp || !Y) return GT_E_INVAL;
  if (!gt_conv_gemm2_supported(N, Cin, taps, gate) || Np != N || Kp != Cin) return GT_E_UNSUPPORTED;
  if ((ldx & 7) || (ldy & 3)) return GT_E_ALIGN;
  if (((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)Y) & 15) return GT_E_ALIGN;
  if (addend && (ldadd & 3)) return GT_E_ALIGN;
  if (cond && (Tp <= 0 || (row0 && B <= 0))) return GT_E_INVAL;
  ConvArgs a;
  a.X = static_cast<const bf16_t*>(X); a.ldx = ldx; a.W = static_cast<const bf16_t*>(Wp); a.bias = bias;
  a.cond = cond; a.ldc = ldc; a.rowmask = rowmask; a.Y = Y; a.ldy = ldy; a.addend = addend; a.ldadd = ldadd;
  a.Tout = static_cast<bf16_t*>(gate_t); a.Sout = static_cast<bf16_t*>(gate_s); a.ldts = ldts;
  a.R = R; a.N = N; a.Cin = Cin; a.taps = taps; a.Tp = Tp > 0 ? Tp : 1; a.Np = Np; a.Kp = Kp;
  a.row0 = row0; a.B = B;
  a.out_f32 = out_f32; a.relu = relu;
  a.y16 = !(ldy & 7);
  { static int ex = -1; if (ex < 0) { const char* e = getenv("GT_CONV_EXP"); ex = e ? atoi(e) : 0; } a.exp_ = ex; }
  a.drop_thresh = 0; a.drop_seed = drop_seed; a.drop_scale = 1.0f; a.seed_dev = seed_dev;
  a.gatebwd = (gate == 2); a.gb_thresh = 0;
  if (drop_p > 0.0f) {
    if (drop_p >= 1.0f) return GT_E_UNSUPPORTED;
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.drop_scale = 1.0f / (1.0f - drop_p);
  }
  if (a.gatebwd) {
    if (!gate_t || !gate_s || out_f32 || relu || (N & 7) || (ldts & 7) || (ldy & 7)) return GT_E_INVAL;
    if (((uintptr_t)gate_t | (uintptr_t)gate_s) & 15) return GT_E_ALIGN;
    a.gb_thresh = a.drop_thresh; a.drop_thresh = 0;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  static int ld = -1;
  if (ld < 0) { const char* e = getenv("GT_CONV2_LOADERS"); ld = e ? atoi(e) : 1; }
  if (gate == 1) {
    if (!gate_t || !gate_s || out_f32) return GT_E_INVAL;
    if ((ldts & 7) || (ldy & 7) || (((uintptr_t)gate_t | (uintptr_t)gate_s) & 15)) return GT_E_ALIGN;
    return ld ? launch2<4, 2, true, 4>(a, st) : launch2<4, 2, true, 0>(a, st);
  }
  // tile choice: the largest tile that still gives ~one workgroup per CU; small problems take 64-row tiles
  const int rt128 = (R + 127) / 128, rt64 = (R + 63) / 64;
  const bool n192 = (N % 192) == 0;
  const char* fe = getenv("GT_CONV2_TILE");                        // dev/test override of the tile choice
  const int force = fe ? atoi(fe) : 0;
  int pick = 0;                                                    // 1: 128x192, 2: 128x96, 3: 64x192, 4: 64x96
  if (force >= 1 && force <= 4 && (n192 || force == 2 || force == 4)) pick = force;
  else if (n192 && rt128 * (N / 192) >= 160) pick = 1;
  else if (rt128 * (N / 96) >= 160) pick = 2;
  else if (n192 && rt64 * (N / 192) >= 160) pick = 3;
  else pick = 4;
  switch (pick) {
    case 1:  return ld ? launch2<4, 2, false, 4>(a, st) : launch2<4, 2, false, 0>(a, st);
    case 2:  return ld ? launch2<4, 1, false, 4>(a, st) : launch2<4, 1, false, 0>(a, st);
    case 3:  return launch2<2, 2, false, 0>(a, st);
    default: return launch2<2, 1, false, 0>(a, st);
  }
}
