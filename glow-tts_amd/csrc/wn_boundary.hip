// Everything BETWEEN two WaveNets of the flow decoder as one kernel, forward and backward, gfx950.
//
// reference, per flow block b (models.py:765-785 runs ActNorm, InvConvNear, CouplingBlock in turn):
//     y      = InvConvNear(ActNorm(x))                                        modules.py:584-599, 635-665
//     h      = start(y0) * mask                                               attentions.py:147
//     out    = WN(h)           = (sum_i skip_i(acts_i)) * mask                modules.py:144-171
//     m|logs = end(out)                                                       attentions.py:162-165
//     z      = [y0 | (m + exp(logs) * y1) * mask],  logdet += sum logs * mask attentions.py:171-184
// All of it except the WaveNet's k=5 layers is ROW-LOCAL (1x1 convs, elementwise maps, 4x4 channel mixing, per-utterance
// sums).  Round 1 ran it as five launches per block forward (skip GEMM, end conv, coupling, ActNorm+InvConv, start conv)
// and five backward, ~13 us each on the decoder's dependent chain for a few microseconds of work.  Here a workgroup owns
// 64 rows and walks the whole chain on them, tile after tile in LDS:
//
//   forward  [tail of block b]    acts [64, 4H] -> skip GEMM (K = 768) -> wn_out -> end conv (K = 192) -> m | logs
//                                 -> coupling -> z (HBM: the flow state, saved for the backward) + per-utterance log-det
//            [head of block b+1]  ActNorm + InvConvNear on the z tile -> y (HBM), y0 (bf16) -> start conv (K = 80) -> h
//   backward [head of block b+1]  d h -> start data gradient (K = 192, N = 80) + identity path -> ActNorm / InvConvNear
//                                 backward (parameter gradients folded over the tile in LDS, then atomics)
//            [tail of block b]    coupling backward -> d[m | logs] -> end data gradient (K = 160) -> d wn_out -> skip data
//                                 gradient (N = 768) -> the skip-path gradient of every layer's gated activations
// The first / last kernels of a pass have only a head or only a tail.  MFMA A = weights in fragment order straight from
// L2 (every workgroup reads the same 0.4 MB: L2-resident), B = the bf16 LDS tile; accumulators e = 4g + j of lane (r, h)
// are row r, column 32 * block + 8g + 4h + j.  What leaves for HBM is what the batched weight gradients and the next
// pass read anyway (wn_out, y0, d out, d wn_out) plus the fp32 flow state.
#include "common.h"
#include "../../include/glowtts_hip.h"

namespace {

constexpr int H = 192, C = 160, HALF = 80, NL = 4, G = C / 4;
constexpr int BM = 64;
constexpr int AP = H + 8;                     // bf16 tile pitch (halfs): 400 B = 16 mod 128 -> conflict-free ds_read_b128
constexpr int XP = 136;                       // y0 tile pitch (halfs): K = 80 -> 5 k-steps; 272 B = 16 mod 128
constexpr int ZP = C + 4;                     // fp32 tile pitch (floats)
#ifndef WNB_EXP
#define WNB_EXP 0                             // dev experiments (bit mask), 0 in every build that ships
#endif
constexpr int RD = 8;                         // weight-fragment ring: k-steps in flight per wave
#ifndef WNB_LDSBAR
#define WNB_LDSBAR 0                          // 1: barriers wait for LDS traffic only (common.h lds_barrier) instead of __syncthreads()'s
                                              // vmcnt(0).  Measured (round 3, back to back): 28.3 vs 28.1 us forward, 28.9 vs 28.7 us backward:
                                              // the drains are not what the phases wait for
#endif
#if WNB_LDSBAR
#define WNB_BARRIER() lds_barrier()
#else
#define WNB_BARRIER() __syncthreads()
#endif
#ifndef WNB_PHASES
#define WNB_PHASES 0                          // dev: per-phase shader-clock stamps of wave 0 (tools/wn_stack_phases.py), 0 in every build that ships
#endif
#if WNB_PHASES
__device__ unsigned long long g_wnb_ph[1024 * 48];
#define PH(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_wnb_ph[blockIdx.x * 48 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH(i) do { } while (0)
#endif

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));   // plain vector: staging arrays of it stay in registers

__device__ __forceinline__ uint4 ldfrag(const bf16_t* __restrict__ W, int f, int lane)
{
  return *reinterpret_cast<const uint4*>(W + ((size_t)f * 64 + lane) * 8);
}
__device__ __forceinline__ bf16x8_t asfrag(const uint4& u) { return __builtin_bit_cast(bf16x8_t, u); }
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack2bf(a, b), pack2bf(c, d)); }

template <int NB>
__device__ __forceinline__ void acc_zero(f32x16_t (&acc)[NB])
{
#pragma unroll
  for (int bn = 0; bn < NB; ++bn)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[bn][e] = 0.0f;
}

// acc[bn] += W[block nb0 + bn][k-steps 0 .. KK) x Bt: W in fragment order with KS k-steps per block row; Brow points at
// (this lane's row, 8h) of the bf16 LDS tile, k-step kk is 16 halfs further.  The first ring of fragments is fetched by
// gemm_prefetch, which callers issue BEFORE the phase that produces the tile (weights do not depend on it), so the L2
// latency of a stage hides under the previous stage's epilogue.
template <int NB, int KK>
struct WRing { uint4 v[(KK < RD ? KK : RD)][NB]; };

template <int NB, int KK>
__device__ __forceinline__ void gemm_prefetch(const bf16_t* __restrict__ W, int KS, int nb0, int lane, WRing<NB, KK>& ring)
{
  constexpr int D = KK < RD ? KK : RD;
#pragma unroll
  for (int p = 0; p < D; ++p)
#pragma unroll
    for (int bn = 0; bn < NB; ++bn) ring.v[p][bn] = ldfrag(W, (nb0 + bn) * KS + p, lane);
}
template <int NB, int KK>
__device__ __forceinline__ void gemm_run(const bf16_t* __restrict__ W, int KS, int nb0, const bf16_t* Brow, int lane, WRing<NB, KK>& ring,
                                         f32x16_t (&acc)[NB])
{
  // blocks of CH k-steps: their MFMAs, then the loads that refill the ring slots those MFMAs have just read.  (A load placed behind
  // every MFMA holds the wave's issue for longer than the MFMA it was meant to hide behind: wn_stack.hip, round 3.)
  constexpr int D = KK < RD ? KK : RD, CH = 4;
#pragma unroll
  for (int k0 = 0; k0 < KK; k0 += CH) {
#pragma unroll
    for (int kk = k0; kk < k0 + CH && kk < KK; ++kk) {
      const bf16x8_t bfm = *reinterpret_cast<const bf16x8_t*>(Brow + kk * 16);
#pragma unroll
      for (int bn = 0; bn < NB; ++bn) acc[bn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring.v[kk % D][bn]), bfm, acc[bn], 0, 0, 0);
    }
    if (k0 + D < KK) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = k0; kk < k0 + CH && kk + D < KK; ++kk)
#pragma unroll
        for (int bn = 0; bn < NB; ++bn) ring.v[kk % D][bn] = ldfrag(W, (nb0 + bn) * KS + kk + D, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// per-utterance sum of 64 per-row values (one wave; rows of an utterance are consecutive): segmented scan, then ONE
// atomic per utterance run of the tile (per-row atomics onto B addresses would serialise at L2)
__device__ __forceinline__ void utt_atomic_add(float* __restrict__ dst, float s, int utt, int lane)
{
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float v = __shfl_up(s, off);
    const int u = __shfl_up(utt, off);
    if (lane >= off && u == utt) s += v;
  }
  const int un = __shfl_down(utt, 1);
  if ((lane == 63 || un != utt) && s != 0.0f) atomicAdd(dst + utt, s);
}

// layer slice L of the skip GEMM (K = 4H in four LDS tiles): registers -> LDS, barrier, 12 k-steps; the weight ring runs
// across the slices (L is a template parameter so that every ring / staging index is a compile-time constant)
template <int L>
__device__ __forceinline__ void skip_slice(const u32x4_t (&xr)[6], bf16_t* As, const bf16_t* __restrict__ Wskip, int wm, int wn, int r, int h,
                                           int lane, uint4 (&ring)[RD][3], f32x16_t (&acc)[3])
{
  constexpr int KK = NL * H / 16, kbase = L * (H / 16);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int chunk = threadIdx.x + 256 * i, row = chunk / 24, c8 = chunk - row * 24;
    *reinterpret_cast<u32x4_t*>(As + (L * BM + row) * AP + c8 * 8) = xr[i];
  }
  WNB_BARRIER();
  const bf16_t* brow = As + (L * BM + 32 * wm + r) * AP + 8 * h;
  constexpr int CH = 4;
#pragma unroll
  for (int k0 = 0; k0 < H / 16; k0 += CH) {
#pragma unroll
    for (int k2 = k0; k2 < k0 + CH; ++k2) {
      const bf16x8_t bfm = *reinterpret_cast<const bf16x8_t*>(brow + k2 * 16);
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) acc[bn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring[(kbase + k2) % RD][bn]), bfm, acc[bn], 0, 0, 0);
    }
    if (kbase + k0 + RD < KK && !(WNB_EXP & 8)) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k2 = k0; k2 < k0 + CH; ++k2)
#pragma unroll
        for (int bn = 0; bn < 3; ++bn)
          if (kbase + k2 + RD < KK) ring[(kbase + k2) % RD][bn] = ldfrag(Wskip, (3 * wn + bn) * KK + kbase + k2 + RD, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// commons.squeeze / unsqueeze (commons.py:339-364, n_sqz = 2) folded into the first / last kernels of a pass: squeezed frame t of
// utterance b, channel p * 80 + c  <->  y[b, c, 2 t + p] of the [B, 80, T] tensor at the decoder's public boundary
__device__ __forceinline__ float4 sq_gather4(const float* __restrict__ y, int b, int t, int c, int T)
{
  const int p = c >= HALF, cc = c - HALF * p;
  const float* src = y + ((size_t)b * HALF + cc) * T + 2 * t + p;
  return make_float4(src[0], src[(size_t)T], src[2 * (size_t)T], src[3 * (size_t)T]);
}
__device__ __forceinline__ void sq_scatter4(float* __restrict__ y, int b, int t, int c, int T, const float4& v)
{
  const int p = c >= HALF, cc = c - HALF * p;
  float* dst = y + ((size_t)b * HALF + cc) * T + 2 * t + p;
  dst[0] = v.x; dst[(size_t)T] = v.y; dst[2 * (size_t)T] = v.z; dst[3 * (size_t)T] = v.w;
}

// A finished [64][AP] bf16 tile -> global rows m0 .. m0 + 63 (row stride ld, 192 channels): 16 bytes per lane, consecutive lanes on
// consecutive addresses.  Stored straight from the MFMA accumulator layout, one instruction writes 16 bytes to each of 32 rows —
// partial lines that the memory pipeline handles one request at a time (wn_stack.hip measured the difference: -5 us per launch).
__device__ __forceinline__ void coop_store_rows(bf16_t* __restrict__ dst, int ld, const bf16_t* __restrict__ tile, int m0, int R)
{
  constexpr int CPR = H / 8;
#pragma unroll
  for (int i = 0; i < BM * CPR / 256; ++i) {
    const int idx = threadIdx.x + 256 * i, row = idx / CPR, c8 = idx - row * CPR, gm = m0 + row;
    if (gm < R) *reinterpret_cast<uint4*>(dst + (size_t)gm * ld + c8 * 8) = *reinterpret_cast<const uint4*>(tile + row * AP + c8 * 8);
  }
}

// Prefetch workgroups.  The launch that FOLLOWS a boundary launch on the decoder's chain is a whole WaveNet, and every one of its
// workgroups streams all 3.2 MB of that WaveNet's weight images — first touched there, they come from HBM (the step's 2 GB of saved
// activations have pushed everything else out of the 256 MB Infinity Cache) and every workgroup waits on the same misses: + 13 us
// on a forward launch, + 9 us on a backward one, all of it gone once the images are back in the cache (tools/wn_layer_bench.py,
// WN_BENCH_COLD).  A boundary launch runs 152 workgroups on 256 CUs: PF_WGS extra workgroups (blockIdx >= the row tiles) read
// those images once, on CUs that were idle, and retire within a few microseconds.
#ifndef WNB_PF_WGS
#define WNB_PF_WGS 64
#endif
constexpr int PF_WGS = WNB_PF_WGS;
__device__ __forceinline__ void prefetch_images(const void* const (&ptr)[16], const uint32_t (&bytes)[16], int wg, uint32_t* sink)
{
  uint32_t acc = 0;
  const int gid = wg * 256 + threadIdx.x, nthr = PF_WGS * 256;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (!ptr[i]) break;
    const uint4* src = static_cast<const uint4*>(ptr[i]);
    const int n16 = (int)(bytes[i] >> 4);
    for (int c = gid; c < n16; c += 4 * nthr) {                       // four independent loads in flight per thread
      const uint4 v0 = src[c];
      const uint4 v1 = c + nthr < n16 ? src[c + nthr] : make_uint4(0, 0, 0, 0);
      const uint4 v2 = c + 2 * nthr < n16 ? src[c + 2 * nthr] : make_uint4(0, 0, 0, 0);
      const uint4 v3 = c + 3 * nthr < n16 ? src[c + 3 * nthr] : make_uint4(0, 0, 0, 0);
      acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
    }
  }
  if (acc == 0x9E3779B9u && sink) *sink = acc;                          // (keeps the loads alive; bf16 weight bits never form this word on every lane)
}

// ------------------------------------------------------------------------------------------------ forward
constexpr int F_AS = 0;                                    // acts slices [4][64][AP] bf16; later At [64][AP] (slice 0) / y0 tile
constexpr int F_O = BM * AP * 2;                           // m | logs tile [64][ZP] fp32 (over slices 1, 2 once they are dead)
constexpr int F_Z = F_O + BM * ZP * 4;                     // z tile [64][ZP] fp32
constexpr int F_RS = (F_Z + BM * ZP * 4 > NL * BM * AP * 2 ? F_Z + BM * ZP * 4 : NL * BM * AP * 2);   // row sums [64] fp32
constexpr int F_BIAS = F_RS + BM * 4;                      // b_skip [192] | b_end [160 (+ 32 zero)] | b_start [192] fp32, staged once
constexpr int FWD_LDS = F_BIAS + 3 * H * 4 + BM * 4;       // + the tile's row mask [64]
static_assert(F_O + BM * ZP * 4 <= 3 * BM * AP * 2, "the m | logs tile must stay clear of acts slice 3");

template <bool TAIL, bool HEAD>
__global__ __launch_bounds__(256) void gt_wn_boundary_fwd_kernel(gt_boundary_fwd_args a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n_tiles = (a.R + BM - 1) / BM;
  if ((int)blockIdx.x >= n_tiles) { prefetch_images(a.pf_ptr, a.pf_bytes, blockIdx.x - n_tiles, reinterpret_cast<uint32_t*>(a.logdet)); return; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, R = a.R;
  bf16_t* As = reinterpret_cast<bf16_t*>(smem + F_AS);
  float* Ot = reinterpret_cast<float*>(smem + F_O);
  float* Zt = reinterpret_cast<float*>(smem + F_Z);
  float* rowsum = reinterpret_cast<float*>(smem + F_RS);
  const int mrow = m0 + 32 * wm + r;                        // this lane's row in the MFMA epilogues
  const float rm_l = mrow < R ? a.rowmask[mrow] : 0.0f;
  // the three convs' biases: read from global memory inside the MFMA epilogues they cost an L2 round trip per (block, group)
  float* Bs = reinterpret_cast<float*>(smem + F_BIAS);
  if (threadIdx.x < 3 * H / 4) {
    const int which = threadIdx.x / (H / 4), o = threadIdx.x - which * (H / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (which == 0) { if (TAIL) v = reinterpret_cast<const float4*>(a.b_skip)[o]; }
    else if (which == 1) { if (TAIL && o < C / 4) v = reinterpret_cast<const float4*>(a.b_end)[o]; }
    else if (HEAD) v = reinterpret_cast<const float4*>(a.b_start)[o];
    reinterpret_cast<float4*>(Bs)[threadIdx.x] = v;
  }
  float* Rm = Bs + 3 * H;                                    // the tile's row mask (a global load inside a loop that also stores waits
  if (threadIdx.x >= 192) {                                  // for those stores: vector-memory operations retire in order)
    const int row = threadIdx.x - 192;
    Rm[row] = m0 + row < R ? a.rowmask[m0 + row] : 0.0f;
  }
  PH(0);

  if (TAIL) {
    const bf16_t* acts = static_cast<const bf16_t*>(a.acts);
    const bf16_t* Wskip = static_cast<const bf16_t*>(a.w_skip);
    // the tile's gated activations, all four layers: 6 x 16 B per thread per layer slice, in flight together
    u32x4_t xr[NL][6];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int chunk = threadIdx.x + 256 * i, row = chunk / 24, c8 = chunk - row * 24;
        const int gm = m0 + row < R ? m0 + row : R - 1;
        if (WNB_EXP & 1) xr[l][i] = u32x4_t{(uint32_t)gm, 0u, 0u, 0u};
        else xr[l][i] = *reinterpret_cast<const u32x4_t*>(acts + (size_t)gm * a.ldacts + l * H + c8 * 8);
      }
    if (threadIdx.x < BM) rowsum[threadIdx.x] = 0.0f;
    // skip GEMM: wn_out = (acts @ Wskip^T + b) * mask;  wave (wm, wn): rows 32 wm .., column blocks 3 wn ..
    f32x16_t acc[3];
    acc_zero<3>(acc);
    constexpr int KK = NL * H / 16;                         // 48 k-steps
    uint4 ring[RD][3];
#pragma unroll
    for (int p = 0; p < RD; ++p)
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) ring[p][bn] = ldfrag(Wskip, (3 * wn + bn) * KK + p, lane);
    PH(1);
    if (!(WNB_EXP & 2)) {
    skip_slice<0>(xr[0], As, Wskip, wm, wn, r, h, lane, ring, acc);
    skip_slice<1>(xr[1], As, Wskip, wm, wn, r, h, lane, ring, acc);
    skip_slice<2>(xr[2], As, Wskip, wm, wn, r, h, lane, ring, acc);
    skip_slice<3>(xr[3], As, Wskip, wm, wn, r, h, lane, ring, acc);
    } else { acc[0][0] = __uint_as_float(xr[0][0].x ^ xr[1][1].x ^ xr[2][2].x ^ xr[3][3].x ^ ring[0][0].x); WNB_BARRIER(); }
    PH(2);
    // the end conv's first weight fragments fly under this epilogue
    WRing<3, H / 16> ring2;
    gemm_prefetch<3, H / 16>(static_cast<const bf16_t*>(a.w_end), a.ks_end, 3 * wn, lane, ring2);
    // every wave is past slices 0..2 (the barrier before slice 3): At = slice 0's region
    bf16_t* wn_out = static_cast<bf16_t*>(a.wn_out);
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (3 * wn + bn) + 8 * g + 4 * h;
        const float4 b4 = *reinterpret_cast<const float4*>(Bs + n);
        const uint2 v = pack4((acc[bn][4 * g] + b4.x) * rm_l, (acc[bn][4 * g + 1] + b4.y) * rm_l,
                              (acc[bn][4 * g + 2] + b4.z) * rm_l, (acc[bn][4 * g + 3] + b4.w) * rm_l);
        *reinterpret_cast<uint2*>(As + (32 * wm + r) * AP + n) = v;
      }
    WNB_BARRIER();
    PH(3);
    if (!(WNB_EXP & 4)) coop_store_rows(wn_out, H, As, m0, R);       // whole rows from the tile (see coop_store_rows)
    // end conv: [m | logs] = wn_out @ Wend^T + b   (N = 160: blocks 0..4, block 5 is the image's zero padding)
    // the coupling's inputs (this block's y, written by the previous launch) fly under the end conv
    float4 cy0[5], cy1[5];
    float crm[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int item = threadIdx.x + 256 * k, row = item / 20, c = 4 * (item - row * 20), gm = m0 + row < R ? m0 + row : R - 1;
      cy0[k] = *reinterpret_cast<const float4*>(a.y + (size_t)gm * C + c);
      cy1[k] = *reinterpret_cast<const float4*>(a.y + (size_t)gm * C + HALF + c);
      crm[k] = a.rowmask[gm];
    }
    f32x16_t acc2[3];
    acc_zero<3>(acc2);
    gemm_run<3, H / 16>(static_cast<const bf16_t*>(a.w_end), a.ks_end, 3 * wn, As + (32 * wm + r) * AP + 8 * h, lane, ring2, acc2);
    PH(4);
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (3 * wn + bn) + 8 * g + 4 * h;
        if (n < C) {
          const float4 b4 = *reinterpret_cast<const float4*>(Bs + H + n);
          *reinterpret_cast<float4*>(Ot + (32 * wm + r) * ZP + n) =
              make_float4(acc2[bn][4 * g] + b4.x, acc2[bn][4 * g + 1] + b4.y, acc2[bn][4 * g + 2] + b4.z, acc2[bn][4 * g + 3] + b4.w);
        }
      }
    WNB_BARRIER();
    PH(5);
    // affine coupling on (row, 4 channels): z = [y0 | (m + exp(logs) y1) mask]
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int item = threadIdx.x + 256 * k, row = item / 20, c = 4 * (item - row * 20), gm = m0 + row;
      float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f), z1 = z0;
      if (gm < R) {
        const float rm = crm[k];
        const float4 mm = *reinterpret_cast<const float4*>(Ot + row * ZP + c);
        const float4 lr = *reinterpret_cast<const float4*>(Ot + row * ZP + HALF + c);
        z0 = cy0[k];
        const float4 y1 = cy1[k];
        float lg[4] = {lr.x, lr.y, lr.z, lr.w};
        if (a.sigmoid_scale) {
#pragma unroll
          for (int j = 0; j < 4; ++j) lg[j] = __logf(1e-6f + sigmoidf_(lg[j] + 2.0f));
        }
        z1 = make_float4((mm.x + __expf(lg[0]) * y1.x) * rm, (mm.y + __expf(lg[1]) * y1.y) * rm,
                         (mm.z + __expf(lg[2]) * y1.z) * rm, (mm.w + __expf(lg[3]) * y1.w) * rm);
        if (!(WNB_EXP & 4)) {
        if (a.z) {
          *reinterpret_cast<float4*>(a.z + (size_t)gm * C + c) = z0;
          *reinterpret_cast<float4*>(a.z + (size_t)gm * C + HALF + c) = z1;
        }
        if (a.z_bct) {                                             // the unsqueeze: straight into [B, 80, T] (pre-zeroed by the caller)
          const int b = (int)a.rowbatch[gm], t = a.rowframe[gm];
          if (t >= 0 && t < a.len[b]) { sq_scatter4(a.z_bct, b, t, c, a.T, z0); sq_scatter4(a.z_bct, b, t, HALF + c, a.T, z1); }
        }
        *reinterpret_cast<float4*>(a.logs_raw + (size_t)gm * HALF + c) = lr;       // the backward needs logs only
        }
        const float s = (lg[0] + lg[1] + lg[2] + lg[3]) * rm;
        if (s != 0.0f) atomicAdd(rowsum + row, s);
      }
      if (HEAD) {
        *reinterpret_cast<float4*>(Zt + row * ZP + c) = z0;
        *reinterpret_cast<float4*>(Zt + row * ZP + HALF + c) = z1;
      }
    }
    WNB_BARRIER();
    PH(6);
    // (the per-utterance log-det atomics are issued at the very end of the kernel: vector-memory operations retire in order, and a
    // float atomic takes microseconds to come back — issued here, every later wait for a load waited for them too)
    if (!HEAD && wave == 0) {
      const int gm = m0 + lane < R ? m0 + lane : R - 1;
      utt_atomic_add(a.logdet, rowsum[lane], a.rowutt[gm], lane);
    }
    PH(7);
  } else {
    // first block: the squeezed mel rows are the flow state
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int item = threadIdx.x + 256 * k, row = item / 40, c = 4 * (item - row * 40), gm = m0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gm < R) {
        if (a.y_bct) {                                             // the squeeze: straight from [B, 80, T]
          const int b = (int)a.rowbatch[gm], t = a.rowframe[gm];
          if (t >= 0 && t < a.len[b]) v = sq_gather4(a.y_bct, b, t, c, a.T);
          if (a.z) *reinterpret_cast<float4*>(a.z + (size_t)gm * C + c) = v;    // the squeezed rows: ActNorm's input, kept for the backward
        } else v = *reinterpret_cast<const float4*>(a.x_in + (size_t)gm * C + c);
      }
      *reinterpret_cast<float4*>(Zt + row * ZP + c) = v;
    }
    WNB_BARRIER();
  }
  if (!HEAD) return;

  // ActNorm + InvConvNear of the next block on (row, channel group g): members {2g, 2g+1, 80+2g, 80+2g+1}
  bf16_t* X0t = reinterpret_cast<bf16_t*>(smem + F_AS);      // At is dead: every wave is past the end conv
  WRing<3, HALF / 16> ring3;                                 // the start conv's weights fly under the ActNorm / InvConvNear phase
  gemm_prefetch<3, HALF / 16>(static_cast<const bf16_t*>(a.w_start), a.ks_start, 3 * wn, lane, ring3);
  {
    // thread = (channel group g, row phase): the group's ActNorm scale / bias are formed ONCE per thread (as items tid + 256 k every
    // item had another group: eight parameter loads and four exponentials per item, 8 k of the launch's 47 k cycles)
    float Wm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) Wm[i] = a.w_ic[i];
    bf16_t* y0b = static_cast<bf16_t*>(a.y0_bf16);
    constexpr int NPH = 256 / G;                               // 6 row phases of 40 threads (16 threads idle)
    const int g = threadIdx.x % G, ph = threadIdx.x / G;
    if (ph < NPH) {
      const float e0 = __expf(a.an_logs[2 * g]), e1 = __expf(a.an_logs[2 * g + 1]);
      const float e2 = __expf(a.an_logs[HALF + 2 * g]), e3 = __expf(a.an_logs[HALF + 2 * g + 1]);
      const float c0 = a.an_bias[2 * g], c1 = a.an_bias[2 * g + 1], c2 = a.an_bias[HALF + 2 * g], c3 = a.an_bias[HALF + 2 * g + 1];
#pragma unroll
      for (int k = 0; k < (BM + NPH - 1) / NPH; ++k) {
        const int row = ph + NPH * k, gm = m0 + row;
        if (row < BM) {
          const float rm = Rm[row];
          const float2 xa = *reinterpret_cast<const float2*>(Zt + row * ZP + 2 * g);
          const float2 xb = *reinterpret_cast<const float2*>(Zt + row * ZP + HALF + 2 * g);
          const float a0 = c0 + e0 * xa.x, a1 = c1 + e1 * xa.y, a2 = c2 + e2 * xb.x, a3 = c3 + e3 * xb.y;
          float o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (Wm[q * 4] * a0 + Wm[q * 4 + 1] * a1 + Wm[q * 4 + 2] * a2 + Wm[q * 4 + 3] * a3) * rm;
          const uint32_t p01 = pack2bf(o[0], o[1]);
          if (gm < R) {
            *reinterpret_cast<float2*>(a.y_next + (size_t)gm * C + 2 * g) = make_float2(o[0], o[1]);
            *reinterpret_cast<float2*>(a.y_next + (size_t)gm * C + HALF + 2 * g) = make_float2(o[2], o[3]);
            *reinterpret_cast<uint32_t*>(y0b + (size_t)gm * HALF + 2 * g) = p01;
          }
          *reinterpret_cast<uint32_t*>(X0t + row * XP + 2 * g) = p01;
        }
      }
    }
  }
  WNB_BARRIER();
  PH(8);
  // start conv: h = (y0 @ Wstart^T + b) * mask   (K = 80: 5 k-steps)
  {
    f32x16_t acc3[3];
    acc_zero<3>(acc3);
    gemm_run<3, HALF / 16>(static_cast<const bf16_t*>(a.w_start), a.ks_start, 3 * wn, X0t + (32 * wm + r) * XP + 8 * h, lane, ring3, acc3);
    PH(9);
    bf16_t* h0 = static_cast<bf16_t*>(a.h_next);
    bf16_t* Hst = reinterpret_cast<bf16_t*>(smem + F_O);             // the m | logs tile is dead: the h tile on its way out
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (3 * wn + bn) + 8 * g + 4 * h;
        const float4 b4 = *reinterpret_cast<const float4*>(Bs + 2 * H + n);
        *reinterpret_cast<uint2*>(Hst + (32 * wm + r) * AP + n) =
            pack4((acc3[bn][4 * g] + b4.x) * rm_l, (acc3[bn][4 * g + 1] + b4.y) * rm_l,
                  (acc3[bn][4 * g + 2] + b4.z) * rm_l, (acc3[bn][4 * g + 3] + b4.w) * rm_l);
      }
    WNB_BARRIER();
    PH(10);
    coop_store_rows(h0, H, Hst, m0, R);
  }
  if (TAIL && wave == 0) {                                   // the coupling's log-det (row sums folded in LDS above)
    const int gm = m0 + lane < R ? m0 + lane : R - 1;
    utt_atomic_add(a.logdet, rowsum[lane], a.rowutt[gm], lane);
  }
  if (blockIdx.x == 0) {                                     // ActNorm + InvConvNear's log-det: (sum logs + (C/4) logdet W) * len_b
    const float per_frame = a.scal[0] + (float)G * a.scal[1];
    for (int b = threadIdx.x; b < a.B; b += 256) atomicAdd(a.logdet + b, per_frame * (float)a.len[b]);
  }
  PH(11);
}

// ------------------------------------------------------------------------------------------------ backward
// ActNorm / InvConvNear parameter gradients of one workgroup: the four row phases' sums (sL, sB [4][64][4], sW [4][16] in LDS) ->
// one atomic per channel
constexpr int PG = 2 * C + 16;                 // a workgroup's partial sums: d logs [C] | d bias [C] | d W [16]

// The same sums as ONE ROW of partials per workgroup (plain stores; gt_boundary_param_reduce adds the rows up after the pass):
// 152 workgroups adding to the same 336 addresses serialise at L2 — ~10 us per launch that either sit in front of every later
// wait (issued mid-kernel) or hold the kernel open (issued at its end).  extra = this launch's log-det bookkeeping term (workgroup 0).
__device__ __forceinline__ void param_grad_partials(const gt_boundary_bwd_args& a, const float* sL, const float* sB, const float* sW, float extra)
{
  float* row = a.pg_partial + (size_t)blockIdx.x * PG;
  for (int t = threadIdx.x; t < PG; t += 256) {              // 336 values, 256 threads
    if (t < 2 * C) {
      const int c = t < C ? t : t - C, hi = c >= HALF, cc = c - HALF * hi, g = cc >> 1, k = 2 * hi + (cc & 1);
      const float* sp = t < C ? sL : sB;
      float v = sp[g * 4 + k] + sp[(64 + g) * 4 + k] + sp[(128 + g) * 4 + k] + sp[(192 + g) * 4 + k];
      if (t < C) v += extra;
      row[t] = v;
    } else {
      const int i = t - 2 * C;
      row[t] = sW[i] + sW[16 + i] + sW[32 + i] + sW[48 + i] + (float)G * extra * a.scal[2 + i];
    }
  }
}

__device__ __forceinline__ void param_grad_atomics(const gt_boundary_bwd_args& a, const float* sL, const float* sB, const float* sW, int ph, int g)
{
  if (ph == 0 && g < G) {
    const int ch[4] = {2 * g, 2 * g + 1, HALF + 2 * g, HALF + 2 * g + 1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      atomicAdd(a.d_an_logs + ch[k], sL[g * 4 + k] + sL[(64 + g) * 4 + k] + sL[(128 + g) * 4 + k] + sL[(192 + g) * 4 + k]);
      atomicAdd(a.d_an_bias + ch[k], sB[g * 4 + k] + sB[(64 + g) * 4 + k] + sB[(128 + g) * 4 + k] + sB[(192 + g) * 4 + k]);
    }
  }
  if (threadIdx.x < 16) atomicAdd(a.d_w_ic + threadIdx.x, sW[threadIdx.x] + sW[16 + threadIdx.x] + sW[32 + threadIdx.x] + sW[48 + threadIdx.x]);
}

constexpr int B_DH = 0;                                    // d h tile [64][AP] bf16; later d wn_out tile
constexpr int B_D = BM * AP * 2;                           // fp32 gradient tile [64][ZP]
constexpr int B_DO = B_D + BM * ZP * 4;                    // d[m | logs] tile [64][AP] bf16
constexpr int B_RED = B_DO + BM * AP * 2;                  // parameter-gradient fold: sL, sB [4][64][4], sW [4][16], sred[4]
constexpr int BWD_LDS = B_RED + 2 * 4 * 64 * 4 * 4 + 4 * 16 * 4 + 16;

template <bool HEADB, bool TAILB>
__global__ __launch_bounds__(256) void gt_wn_boundary_bwd_kernel(gt_boundary_bwd_args a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n_tiles = (a.R + BM - 1) / BM;
  if ((int)blockIdx.x >= n_tiles) { prefetch_images(a.pf_ptr, a.pf_bytes, blockIdx.x - n_tiles, nullptr); return; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, R = a.R;
  bf16_t* Dh = reinterpret_cast<bf16_t*>(smem + B_DH);
  float* Dt = reinterpret_cast<float*>(smem + B_D);
  bf16_t* Dout = reinterpret_cast<bf16_t*>(smem + B_DO);
  const int mrow = m0 + 32 * wm + r;
  const float rm_l = mrow < R ? a.rowmask[mrow] : 0.0f;
  PH(0);

  if (HEADB) {
    // d y0 (start conv part) = d h @ Wstart: N = 80 (blocks 0..2 of the padded image; wave wn takes 2 wn, 2 wn + 1)
    const bf16_t* dh = static_cast<const bf16_t*>(a.dh);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int chunk = threadIdx.x + 256 * i, row = chunk / 24, c8 = chunk - row * 24;
      const int gm = m0 + row < R ? m0 + row : R - 1;
      *reinterpret_cast<uint4*>(Dh + row * AP + c8 * 8) = *reinterpret_cast<const uint4*>(dh + (size_t)gm * H + c8 * 8);
    }
    WRing<2, H / 16> ring1;
    gemm_prefetch<2, H / 16>(static_cast<const bf16_t*>(a.w_start_d), a.ks_start_d, 2 * wn, lane, ring1);
    WNB_BARRIER();
    PH(1);
    f32x16_t acc[2];
    acc_zero<2>(acc);
    gemm_run<2, H / 16>(static_cast<const bf16_t*>(a.w_start_d), a.ks_start_d, 2 * wn, Dh + (32 * wm + r) * AP + 8 * h, lane, ring1, acc);
#pragma unroll
    for (int bn = 0; bn < 2; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (2 * wn + bn) + 8 * g + 4 * h;
        if (n < HALF)
          *reinterpret_cast<float4*>(Dt + (32 * wm + r) * ZP + n) = make_float4(acc[bn][4 * g], acc[bn][4 * g + 1], acc[bn][4 * g + 2], acc[bn][4 * g + 3]);
      }
    WNB_BARRIER();
    PH(2);
    // ActNorm + InvConvNear backward: wave = row phase (rows ph, ph + 4, ..), lane = channel group
    float* sL = reinterpret_cast<float*>(smem + B_RED);
    float* sB = sL + 4 * 64 * 4;
    float* sW = sB + 4 * 64 * 4;
    float* sred = sW + 4 * 16;
    float extra = 0.f;
    if (blockIdx.x == 0) {                                   // backward of the log-det bookkeeping (see flow_ops.hip)
      float sv = 0.f;
      for (int b = threadIdx.x; b < a.B; b += 256) sv += a.dlogdet[b] * (float)a.len[b];
      sv = wave_sum(sv);
      if (lane == 0) sred[wave] = sv;
      WNB_BARRIER();
      sv = sred[0] + sred[1] + sred[2] + sred[3];
      if (a.pg_partial) extra = sv;
      else {
        for (int c = threadIdx.x; c < C; c += 256) atomicAdd(a.d_an_logs + c, sv);
        if (threadIdx.x < 16) atomicAdd(a.d_w_ic + threadIdx.x, (float)G * sv * a.scal[2 + threadIdx.x]);
      }
    }
    const int g = lane, ph = wave;
    float accW[16], accL[4] = {0, 0, 0, 0}, accB[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) accW[i] = 0.f;
    if (g < G) {
      const int ch[4] = {2 * g, 2 * g + 1, HALF + 2 * g, HALF + 2 * g + 1};
      float el[4], bs[4], Wm[16];
#pragma unroll
      for (int k = 0; k < 4; ++k) { el[k] = __expf(a.an_logs[ch[k]]); bs[k] = a.an_bias[ch[k]]; }
#pragma unroll
      for (int i = 0; i < 16; ++i) Wm[i] = a.w_ic[i];
      const float* __restrict__ xp = a.x;
      const float* __restrict__ dip = a.dx_in;
#pragma unroll
      for (int rb = 0; rb < BM / 4; rb += 8) {               // this wave's 16 rows in two batches: the batch's loads fly together
        float2 xa[8], xb[8], da[8], db[8];
        float rmv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = ph + 4 * (rb + u), gm = m0 + row < R ? m0 + row : R - 1;
          xa[u] = *reinterpret_cast<const float2*>(xp + (size_t)gm * C + 2 * g);
          xb[u] = *reinterpret_cast<const float2*>(xp + (size_t)gm * C + HALF + 2 * g);
          da[u] = *reinterpret_cast<const float2*>(dip + (size_t)gm * C + 2 * g);          // identity path of d y0
          db[u] = *reinterpret_cast<const float2*>(dip + (size_t)gm * C + HALF + 2 * g);
          rmv[u] = m0 + row < R ? a.rowmask[gm] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = ph + 4 * (rb + u), gm = m0 + row;
          const float rm = rmv[u];
          const float2 ds = *reinterpret_cast<const float2*>(Dt + row * ZP + 2 * g);
          const float xv[4] = {xa[u].x, xa[u].y, xb[u].x, xb[u].y};
          const float dym[4] = {(da[u].x + ds.x) * rm, (da[u].y + ds.y) * rm, db[u].x * rm, db[u].y * rm};
          float av[4], d_a[4], dxv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) av[k] = bs[k] + el[k] * xv[k];
#pragma unroll
          for (int i = 0; i < 4; ++i) d_a[i] = Wm[i] * dym[0] + Wm[4 + i] * dym[1] + Wm[8 + i] * dym[2] + Wm[12 + i] * dym[3];
#pragma unroll
          for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int i = 0; i < 4; ++i) accW[o * 4 + i] += dym[o] * av[i];
#pragma unroll
          for (int k = 0; k < 4; ++k) { accB[k] += d_a[k]; accL[k] += d_a[k] * xv[k] * el[k]; dxv[k] = d_a[k] * el[k]; }
          if (TAILB) {
            *reinterpret_cast<float2*>(Dt + row * ZP + 2 * g) = make_float2(dxv[0], dxv[1]);
            *reinterpret_cast<float2*>(Dt + row * ZP + HALF + 2 * g) = make_float2(dxv[2], dxv[3]);
          } else if (gm < R) {
            if (a.dx_bct) {                                        // the unsqueeze of the decoder's input gradient
              const int b = (int)a.rowbatch[gm], t = a.rowframe[gm];
              if (t >= 0 && t < a.len[b]) {
                float* d0 = a.dx_bct + ((size_t)b * HALF + 2 * g) * a.T + 2 * t;
                d0[0] = dxv[0]; d0[(size_t)a.T] = dxv[1]; d0[1] = dxv[2]; d0[(size_t)a.T + 1] = dxv[3];
              }
            } else {
              *reinterpret_cast<float2*>(a.dx_out + (size_t)gm * C + 2 * g) = make_float2(dxv[0], dxv[1]);
              *reinterpret_cast<float2*>(a.dx_out + (size_t)gm * C + HALF + 2 * g) = make_float2(dxv[2], dxv[3]);
            }
          }
        }
      }
    }
    PH(3);
    // one atomic per channel per workgroup (same-address float atomics serialise at L2): fold the row phases in LDS
#pragma unroll
    for (int k = 0; k < 4; ++k) { sL[(ph * 64 + g) * 4 + k] = accL[k]; sB[(ph * 64 + g) * 4 + k] = accB[k]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) accW[i] = wave_sum(accW[i]);
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sW[ph * 16 + i] = accW[i];
    }
    WNB_BARRIER();
    // The workgroup's 336 parameter-gradient atomics (152 workgroups add to the same 336 addresses: they serialise at L2 and take
    // microseconds to retire) are issued at the END of the kernel: vector-memory operations retire in order, so issued here every
    // later wait for a load — the coupling's operands, the next weight fragments — waited for them too (10 k of the launch's 58 k
    // cycles, tools/wn_boundary_phases.py).  The folded sums stay in their LDS region (nothing below reuses it).
    PH(4);
    if (a.pg_partial) param_grad_partials(a, sL, sB, sW, extra);     // plain stores: nothing later waits long for them
    if (!TAILB) {
      if (!a.pg_partial) param_grad_atomics(a, sL, sB, sW, wave, lane);
      return;
    }
  } else {
    // last block: the squeezed gradient of the decoder's output is d z
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int item = threadIdx.x + 256 * k, row = item / 40, c = 4 * (item - row * 40), gm = m0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gm < R) {
        if (a.dz_bct) {
          const int b = (int)a.rowbatch[gm], t = a.rowframe[gm];
          if (t >= 0 && t < a.len[b]) v = sq_gather4(a.dz_bct, b, t, c, a.T);
        } else v = *reinterpret_cast<const float4*>(a.dz_in + (size_t)gm * C + c);
      }
      *reinterpret_cast<float4*>(Dt + row * ZP + c) = v;
    }
    WNB_BARRIER();
  }

  // coupling backward on (row, 4 channels): d x = [d z0 | d z1 exp(logs)], d m = d z1, d logs = d z1 exp(logs) y1 + d logdet
  bf16_t* dout = static_cast<bf16_t*>(a.dout);
  WRing<3, C / 16> ring2;                                    // the end conv's data-gradient weights fly under this phase
  gemm_prefetch<3, C / 16>(static_cast<const bf16_t*>(a.w_end_d), a.ks_end_d, 3 * wn, lane, ring2);
  {
    float4 lr4[5], y14[5];
    float rmv[5], dldv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {                            // all of the phase's HBM reads first
      const int item = threadIdx.x + 256 * k, row = item / 20, c = 4 * (item - row * 20), gm = m0 + row < R ? m0 + row : R - 1;
      lr4[k] = *reinterpret_cast<const float4*>(a.logs_raw + (size_t)gm * HALF + c);
      y14[k] = *reinterpret_cast<const float4*>(a.y + (size_t)gm * C + HALF + c);
      rmv[k] = a.rowmask[gm];
      dldv[k] = a.dlogdet[a.rowutt[gm]];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int item = threadIdx.x + 256 * k, row = item / 20, c = 4 * (item - row * 20), gm = m0 + row;
      uint2 pm = make_uint2(0, 0), pl = make_uint2(0, 0);
      if (gm < R) {
        const float rm = rmv[k];
        const float4 dz0 = *reinterpret_cast<const float4*>(Dt + row * ZP + c);
        const float4 dz1r = *reinterpret_cast<const float4*>(Dt + row * ZP + HALF + c);
        const float dld = dldv[k] * rm;
        const float dz1[4] = {dz1r.x * rm, dz1r.y * rm, dz1r.z * rm, dz1r.w * rm};
        const float lraw[4] = {lr4[k].x, lr4[k].y, lr4[k].z, lr4[k].w}, y1v[4] = {y14[k].x, y14[k].y, y14[k].z, y14[k].w};
        float dx1[4], dlg[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float lg = lraw[j], dl_draw = 1.0f;
          if (a.sigmoid_scale) { const float sg = sigmoidf_(lraw[j] + 2.0f); lg = __logf(1e-6f + sg); dl_draw = sg * (1.0f - sg) / (1e-6f + sg); }
          const float e = __expf(lg);
          dx1[j] = dz1[j] * e;
          dlg[j] = (dz1[j] * e * y1v[j] + dld) * dl_draw;
        }
        *reinterpret_cast<float4*>(a.dx_out + (size_t)gm * C + c) = dz0;
        *reinterpret_cast<float4*>(a.dx_out + (size_t)gm * C + HALF + c) = make_float4(dx1[0], dx1[1], dx1[2], dx1[3]);
        pm = pack4(dz1[0], dz1[1], dz1[2], dz1[3]);
        pl = pack4(dlg[0], dlg[1], dlg[2], dlg[3]);
        *reinterpret_cast<uint2*>(dout + (size_t)gm * C + c) = pm;
        *reinterpret_cast<uint2*>(dout + (size_t)gm * C + HALF + c) = pl;
      }
      *reinterpret_cast<uint2*>(Dout + row * AP + c) = pm;
      *reinterpret_cast<uint2*>(Dout + row * AP + HALF + c) = pl;
    }
  }
  WNB_BARRIER();
  PH(5);
  // end conv data gradient: d wn_out = (d out @ Wend) * mask   (K = 160: 10 k-steps)
  bf16_t* At = Dh;                                           // the d h tile is dead
  const bf16_t* Wsd = static_cast<const bf16_t*>(a.w_skip_d);
  constexpr int KS2 = H / 16, NS = NL * KS2;                 // skip data gradient: 4 layer windows x 12 k-steps, ONE weight stream
  uint4 ring3[RD][3];
  {
    f32x16_t acc[3];
    acc_zero<3>(acc);
    gemm_run<3, C / 16>(static_cast<const bf16_t*>(a.w_end_d), a.ks_end_d, 3 * wn, Dout + (32 * wm + r) * AP + 8 * h, lane, ring2, acc);
    PH(6);
#pragma unroll
    for (int p = 0; p < RD; ++p)                             // the skip stage's first fragments fly under this epilogue
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) ring3[p][bn] = ldfrag(Wsd, (6 * (p / KS2) + 3 * wn + bn) * a.ks_skip_d + p % KS2, lane);
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (3 * wn + bn) + 8 * g + 4 * h;
        const uint2 v = pack4(acc[bn][4 * g] * rm_l, acc[bn][4 * g + 1] * rm_l, acc[bn][4 * g + 2] * rm_l, acc[bn][4 * g + 3] * rm_l);
        *reinterpret_cast<uint2*>(At + (32 * wm + r) * AP + n) = v;
      }
  }
  WNB_BARRIER();
  PH(7);
  coop_store_rows(static_cast<bf16_t*>(a.dwn_out), H, At, m0, R);
  PH(8);
  // skip data gradient: d acts_l (skip path) = d wn_out @ Wskip_l, one layer window per pass
  bf16_t* via = static_cast<bf16_t*>(a.via_skip);
  const bf16_t* brow = At + (32 * wm + r) * AP + 8 * h;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    f32x16_t acc[3];
    acc_zero<3>(acc);
#pragma unroll
    for (int k0 = 0; k0 < KS2; k0 += 4) {
#pragma unroll
      for (int kk = k0; kk < k0 + 4; ++kk) {
        const int st = l * KS2 + kk;
        const bf16x8_t bfm = *reinterpret_cast<const bf16x8_t*>(brow + kk * 16);
#pragma unroll
        for (int bn = 0; bn < 3; ++bn) acc[bn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asfrag(ring3[st % RD][bn]), bfm, acc[bn], 0, 0, 0);
      }
      if (l * KS2 + k0 + RD < NS) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = k0; kk < k0 + 4; ++kk) {
          const int st = l * KS2 + kk, nx = st + RD;
#pragma unroll
          for (int bn = 0; bn < 3; ++bn)
            if (nx < NS) ring3[st % RD][bn] = ldfrag(Wsd, (6 * (nx / KS2) + 3 * wn + bn) * a.ks_skip_d + nx % KS2, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // out through one of two dead tiles (the fp32 gradient tile / the d[m | logs] tile), alternating: one barrier per layer window
    bf16_t* Vst = reinterpret_cast<bf16_t*>(smem + ((l & 1) ? B_DO : B_D));
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (3 * wn + bn) + 8 * g + 4 * h;
        *reinterpret_cast<uint2*>(Vst + (32 * wm + r) * AP + n) = pack4(acc[bn][4 * g], acc[bn][4 * g + 1], acc[bn][4 * g + 2], acc[bn][4 * g + 3]);
      }
    WNB_BARRIER();
    coop_store_rows(via + l * H, a.ldvs, Vst, m0, R);
    PH(9 + l);
  }
  if (HEADB && !a.pg_partial) {
    const float* sL = reinterpret_cast<const float*>(smem + B_RED);
    param_grad_atomics(a, sL, sL + 4 * 64 * 4, sL + 2 * 4 * 64 * 4, wave, lane);
  }
}

// rows of per-workgroup partials [n_blocks][n_wg][PG] -> += into every block's d logs / d bias / d W (one writer per address)
__global__ __launch_bounds__(384) void gt_boundary_param_reduce_kernel(const float* __restrict__ pg, int n_wg, float* const* __restrict__ dst)
{
  // grid (block of the decoder, eighth of the partial rows): 8 adders per address (one thread walking all ~150 rows took 38 dependent
  // load rounds: 20 us on the decoder's backward chain)
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= PG) return;
  const int per = (n_wg + gridDim.y - 1) / gridDim.y, w0 = blockIdx.y * per, w1 = min(n_wg, w0 + per);
  const float* p = pg + (size_t)b * n_wg * PG + t;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int w = w0;
  for (; w + 4 <= w1; w += 4) { s0 += p[(size_t)w * PG]; s1 += p[(size_t)(w + 1) * PG]; s2 += p[(size_t)(w + 2) * PG]; s3 += p[(size_t)(w + 3) * PG]; }
  for (; w < w1; ++w) s0 += p[(size_t)w * PG];
  float* d = t < C ? dst[3 * b] + t : (t < 2 * C ? dst[3 * b + 1] + (t - C) : dst[3 * b + 2] + (t - 2 * C));
  atomicAdd(d, (s0 + s1) + (s2 + s3));
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename K>
int opt_in_lds(K kernel, int bytes)
{
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : -1;
}

}  // namespace

#if WNB_PHASES
extern "C" int gt_dev_wnb_phases(void* dst, size_t bytes)
{
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wnb_ph), bytes < sizeof(g_wnb_ph) ? bytes : sizeof(g_wnb_ph)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int gt_wn_boundary_fwd(const gt_boundary_fwd_args* args, void* stream)
{
  if (!args) return GT_E_INVAL;
  const gt_boundary_fwd_args& a = *args;
  if (a.R < 0) return GT_E_INVAL;
  if (a.R == 0) return GT_OK;
  if (a.H != H || a.C != C || a.n_layers != NL) return GT_E_UNSUPPORTED;
  const bool tail = a.acts != nullptr, head = a.y_next != nullptr;
  if (!tail && !head) return GT_E_INVAL;
  if (!a.rowmask) return GT_E_INVAL;
  if (tail) {
    if (!a.w_skip || !a.b_skip || !a.w_end || !a.b_end || !a.y || !a.wn_out || !a.logs_raw || (!a.z && !a.z_bct) || !a.logdet || !a.rowutt) return GT_E_INVAL;
    if (a.z_bct && (head || !a.rowbatch || !a.rowframe || !a.len || a.T <= 0)) return GT_E_INVAL;
    if (a.ldacts < NL * H || (a.ldacts & 7) || a.ks_end < H / 16) return GT_E_INVAL;
    if (!al16(a.acts) || !al16(a.w_skip) || !al16(a.w_end) || !al16(a.b_skip) || !al16(a.b_end) || !al16(a.y) || !al16(a.wn_out) ||
        !al16(a.logs_raw) || !al16(a.z)) return GT_E_ALIGN;
  } else if (a.y_bct ? (!a.rowbatch || !a.rowframe || a.T <= 0) : (!a.x_in || !al16(a.x_in))) return GT_E_INVAL;
  if (head) {
    if (!a.an_logs || !a.an_bias || !a.w_ic || !a.scal || !a.len || a.B <= 0 || !a.logdet || !a.y0_bf16 || !a.w_start || !a.b_start || !a.h_next)
      return GT_E_INVAL;
    if (a.ks_start < HALF / 16) return GT_E_INVAL;
    if (!al16(a.y_next) || !al16(a.y0_bf16) || !al16(a.w_start) || !al16(a.b_start) || !al16(a.h_next)) return GT_E_ALIGN;
  }
  static bool attr = false;                    // > 64 KB of LDS: opt in once per process
  if (!attr) {
    if (opt_in_lds(&gt_wn_boundary_fwd_kernel<true, true>, FWD_LDS) || opt_in_lds(&gt_wn_boundary_fwd_kernel<true, false>, FWD_LDS) ||
        opt_in_lds(&gt_wn_boundary_fwd_kernel<false, true>, FWD_LDS)) return GT_E_LAUNCH;
    attr = true;
  }
  for (int i = 0; i < 16; ++i) if (a.pf_ptr[i] && (!al16(a.pf_ptr[i]) || (a.pf_bytes[i] & 15))) return GT_E_ALIGN;
  const dim3 grid((a.R + BM - 1) / BM + (a.pf_ptr[0] ? PF_WGS : 0)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (tail && head) hipLaunchKernelGGL((gt_wn_boundary_fwd_kernel<true, true>), grid, block, FWD_LDS, st, a);
  else if (tail)    hipLaunchKernelGGL((gt_wn_boundary_fwd_kernel<true, false>), grid, block, FWD_LDS, st, a);
  else              hipLaunchKernelGGL((gt_wn_boundary_fwd_kernel<false, true>), grid, block, FWD_LDS, st, a);
  return gt_launch_status(__func__);
}

extern "C" int gt_boundary_param_partials(void) { return PG; }

extern "C" int gt_boundary_param_reduce(const float* partials, int n_wg, int n_blocks, float* const* dst, void* stream)
{
  if (!partials || !dst || n_wg <= 0 || n_blocks < 0) return GT_E_INVAL;
  if (n_blocks == 0) return GT_OK;
  hipLaunchKernelGGL(gt_boundary_param_reduce_kernel, dim3(n_blocks, 8), dim3(384), 0, static_cast<hipStream_t>(stream), partials, n_wg, dst);
  return gt_launch_status(__func__);
}

extern "C" int gt_wn_boundary_bwd(const gt_boundary_bwd_args* args, void* stream)
{
  if (!args) return GT_E_INVAL;
  const gt_boundary_bwd_args& a = *args;
  if (a.R < 0) return GT_E_INVAL;
  if (a.R == 0) return GT_OK;
  if (a.H != H || a.C != C || a.n_layers != NL) return GT_E_UNSUPPORTED;
  const bool headb = a.dh != nullptr, tailb = a.dout != nullptr;
  if (!headb && !tailb) return GT_E_INVAL;
  if (!a.rowmask || (!a.dx_out && !a.dx_bct) || !al16(a.dx_out)) return GT_E_INVAL;
  if ((a.dx_bct || a.dz_bct) && (!a.rowbatch || !a.rowframe || !a.len || a.T <= 0)) return GT_E_INVAL;
  if (a.dx_bct && tailb) return GT_E_INVAL;
  if (headb) {
    if (!a.w_start_d || !a.dx_in || !a.x || !a.an_logs || !a.an_bias || !a.w_ic || !a.scal || !a.len || !a.dlogdet || a.B <= 0 ||
        !a.d_an_logs || !a.d_an_bias || !a.d_w_ic) return GT_E_INVAL;
    if (a.ks_start_d < H / 16) return GT_E_INVAL;
    if (!al16(a.dh) || !al16(a.w_start_d) || !al16(a.dx_in) || !al16(a.x)) return GT_E_ALIGN;
  } else if (!a.dz_bct && (!a.dz_in || !al16(a.dz_in))) return GT_E_INVAL;
  if (tailb) {
    if (!a.logs_raw || !a.y || !a.dlogdet || !a.rowutt || !a.w_end_d || !a.dwn_out || !a.w_skip_d || !a.via_skip) return GT_E_INVAL;
    if (a.ks_end_d < C / 16 || a.ks_skip_d < H / 16 || a.ldvs < NL * H || (a.ldvs & 7)) return GT_E_INVAL;
    if (!al16(a.logs_raw) || !al16(a.y) || !al16(a.dout) || !al16(a.w_end_d) || !al16(a.dwn_out) || !al16(a.w_skip_d) || !al16(a.via_skip))
      return GT_E_ALIGN;
  }
  static bool attr = false;
  if (!attr) {
    if (opt_in_lds(&gt_wn_boundary_bwd_kernel<true, true>, BWD_LDS) || opt_in_lds(&gt_wn_boundary_bwd_kernel<true, false>, BWD_LDS) ||
        opt_in_lds(&gt_wn_boundary_bwd_kernel<false, true>, BWD_LDS)) return GT_E_LAUNCH;
    attr = true;
  }
  for (int i = 0; i < 16; ++i) if (a.pf_ptr[i] && (!al16(a.pf_ptr[i]) || (a.pf_bytes[i] & 15))) return GT_E_ALIGN;
  const dim3 grid((a.R + BM - 1) / BM + (a.pf_ptr[0] ? PF_WGS : 0)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (headb && tailb) hipLaunchKernelGGL((gt_wn_boundary_bwd_kernel<true, true>), grid, block, BWD_LDS, st, a);
  else if (headb)     hipLaunchKernelGGL((gt_wn_boundary_bwd_kernel<true, false>), grid, block, BWD_LDS, st, a);
  else                hipLaunchKernelGGL((gt_wn_boundary_bwd_kernel<false, true>), grid, block, BWD_LDS, st, a);
  return gt_launch_status(__func__);
}
