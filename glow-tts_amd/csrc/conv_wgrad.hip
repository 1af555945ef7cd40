// Weight gradient of the rows-layout 1-D convolution on bf16 MFMA for gfx950, plus the
// weight-norm backward that consumes it.  (The reference gets these from autograd through
// ATen conv backward + torch weight_norm: modules.py:127-141,152,165; attentions.py:103.)
//
//   dW[tap][co][ci] = sum_m dY[m, co] * X[m + tap - k/2, ci]          (reduction over ALL rows)
//
// MFMA A = dY^T (rows = co, k = row m), MFMA B = X (k = row m + tap shift, cols = ci).  Both
// operands have the reduction index strided in memory (channels-last), so tiles are staged
// row-major in LDS by coalesced 16-byte loads and consumed through ds_read_b64_tr_b16 (the
// hardware transposing read), pitch = row bytes + 64 so that the four rows of a read block hit
// disjoint bank windows.  One workgroup = 128 co x 64 ci x all taps over one slab of rows; slab
// partials go to a workspace [S][taps][Cout][Cin] with plain 128-byte-coalesced stores (float
// atomics would cap at ~1.3 TB/s) and are summed by the weight-norm backward kernel, which then
// maps dW to (dv, dg) or to a plain dw in the parameter's own [Cout, Cin, taps] layout.
#include <stdlib.h>
#include "common.h"
#include "../../include/glowtts_hip.h"

namespace {

// Per tap count: rows per ring stage (KB), ring depth (NST), input-channel groups of 64 per workgroup tile (NJ).
// (measured on the decoder's 48 k = 5 jobs of 8.9 k rows, `profiles/r03_wgrad_variants.txt`: 32-row stages x 5 deep 336-355 us, 64-row
//  stages x 3 deep 310-316 us — half the barriers; reading the fragments of k-step i + 1 ahead of the MFMAs of k-step i changed nothing
//  with two workgroups per CU and was dropped)
#ifndef WG5_KB
#define WG5_KB 64
#endif
#ifndef WG5_NST
#define WG5_NST 3
#endif
#ifndef WG3_KB
#define WG3_KB 64
#endif
#ifndef WG3_NST
#define WG3_NST 3
#endif
#ifndef WG1_KB
#define WG1_KB 32
#endif
#ifndef WG1_NST
#define WG1_NST 3
#endif
#ifndef WG1_NJ
#define WG1_NJ 3
#endif
constexpr int SLAB_Q = 64;                    // slab_rows is a multiple of this (host planners), so only a job's LAST slab ends ragged

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)reinterpret_cast<uintptr_t>(p); }

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* p0, const unsigned char* p1) {
  // two transposing reads: rows kb..kb+3 and kb+4..kb+7 of this lane's column
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)p1);
  typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// LDS-DMA: 16 bytes per lane from `src` to LDS bytes [dst_base + 16 * lane, +16) — no VGPR destination, so the ring below costs no
// registers; in inline asm because hipcc drains vmcnt(0) in front of every LDS read while a builtin glds is in flight.
__device__ __forceinline__ void glds16(const void* src, unsigned dst_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst_base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// Geometry of one ring stage: KB rows of the dY tile (128 columns = 256 B per row) and KB + TAPS - 1 rows of the X tile (64 NJ columns =
// 128 NJ B per row, NJ = 1 or 3), both DENSE — an LDS-DMA instruction writes 64 lanes x 16 B to consecutive LDS bytes, so rows shorter
// than 1 KB cannot be padded.  Bank conflicts of the transposing reads (a 32-lane group reads 4 rows x 64 B in one LDS cycle if they fall
// into four different 64-byte windows of the 256-byte bank period) are avoided by permuting the 16-byte chunks of a row on the SOURCE
// side instead: LDS chunk c' of dY row r holds global chunk c' ^ 4 (r & 3) (rows alias every 256 B), LDS chunk c' of X row r holds global
// chunk c' ^ 4 ((r >> 1) & 1) (rows of 128 B or 384 B alias every second row; rows r, r + 1 sit in different halves of the period anyway).
template <int TAPS, int NJ, int KB_, int NST_> struct WgGeo {
  static constexpr int KB = KB_, NST = NST_;
  static constexpr int XR = KB + TAPS - 1;
  static constexpr int XCH = 8 * NJ;                 // 16-byte chunks per X row
  static constexpr int YB = KB * 256, XB = XR * XCH * 16, STAGE = YB + XB;
  static constexpr int YI = KB / 16;                 // dY LDS-DMA instructions per wave and stage (KB * 16 chunks over 256 lanes)
  static constexpr int NWX = XR * XCH / 4;           // X chunks per wave and stage
  static constexpr int XI = (NWX + 63) / 64;         // ... instructions (the last one partially masked)
  static constexpr int NLD = YI + XI;                // LDS-DMA instructions one wave has in flight per stage
  static constexpr int NB = TAPS * NJ;               // X fragments (= MFMA pairs) per k-step and wave
  static constexpr int LDS = NST * STAGE;
  static_assert(KB % 16 == 0 && SLAB_Q % KB == 0, "stage rows");
  static_assert((XR * XCH) % 4 == 0, "X chunks split evenly over the waves");
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
  static_assert((NST - 2) * NLD <= 63, "vmcnt field");
  static_assert(TAPS == 1 || NJ == 1, "taps or channel groups");
};

// One workgroup: 128 output channels [co0, co0+128) of the dY view (Ncols columns wide; dY column c is output channel co_begin + c of a
// conv with Cout channels) x 64 NJ input channels x all taps, rows of one slab.  Wave (wco, wci) owns 64 output channels x (32 input
// channels of every 64-group) x all taps: per 16-row k-step 2 dY^T fragments + NB X fragments for 2 NB MFMAs.
//
// Round 3: the operands come through an NST-deep LDS ring filled by LDS-DMA.  The round-2 form (64-row stages staged through
// registers, one stage ahead) took 2.0 us per stage whatever the tap count — 0.53 us of MFMA work at k = 5, 0.1 us at k = 1: one
// HBM/L2 round trip per stage with nothing else in flight (serial profile of the step: 480 us for the decoder's k = 5 launch, 322 us
// for its k = 1 launch).  Now NST - 1 stages are in flight per workgroup, none of them holds a register, with <= 256 registers two
// workgroups share a CU, and the k = 1 tile spans 192 input channels (2 + 3 fragments for 6 MFMAs instead of 2 + 1 for 2).
template <int TAPS, int NJ, int KB_, int NST_>
__device__ __forceinline__ void wgrad_tile(
    const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ dY, int ldy,
    float* __restrict__ part, float* __restrict__ part_bias, int R, int Cin, int Cout, int slab_rows,
    int co0, int ci0, int slab, int co_begin, int Ncols)
{
  using G = WgGeo<TAPS, NJ, KB_, NST_>;
  constexpr int KB = G::KB, NST = G::NST, NB = G::NB;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];       // [NST][STAGE]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int padl = TAPS >> 1;
  const int mbeg = slab * slab_rows;
  const int mend = min(R, mbeg + slab_rows);
  const int nst = (mend - mbeg + KB - 1) / KB;
  const int wco = wave >> 1, wci = wave & 1;
  const bool do_bias = (ci0 == 0) && part_bias && wci == 0;     // column sums of dY ride along (VALU adds on the A fragments)

  f32x16_t acc[NB][2];
  float bsum[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.0f;

  // ---- what this lane fetches per stage (element offsets relative to the stage's first row; same chunk <-> lane map in both fill paths)
  int yrow[G::YI], ycol[G::YI], xrow[G::XI], xcol[G::XI];
  bool xact[G::XI];
#pragma unroll
  for (int j = 0; j < G::YI; ++j) {
    const int L = j * 256 + tid, r = L >> 4, c = (L & 15) ^ ((r & 3) << 2);
    yrow[j] = r; ycol[j] = min(co0 + c * 8, Ncols - 8);          // columns past the view: a valid chunk whose products nobody stores
  }
#pragma unroll
  for (int j = 0; j < G::XI; ++j) {
    const int Lw = j * 64 + lane, L = wave * G::NWX + Lw, r = L / G::XCH, c = (L - r * G::XCH) ^ (((r >> 1) & 1) << 2);
    xact[j] = Lw < G::NWX;
    xrow[j] = r - padl; xcol[j] = min(ci0 + c * 8, Cin - 8);
  }
  const unsigned ring0 = lds_off(ring);

  auto issue = [&](int s, int sl_i) {                          // stage s into ring slot sl_i (= s % NST, tracked by the caller)
    const int mb = mbeg + s * KB;
    const unsigned slot = ring0 + (unsigned)sl_i * G::STAGE;
    const bool edge = (mb - padl < 0) || (mb + KB + padl > R) || (mb + KB > mend);       // workgroup-uniform
    if (!edge) {
#pragma unroll
      for (int j = 0; j < G::YI; ++j)
        glds16(dY + (size_t)(mb + yrow[j]) * ldy + ycol[j], slot + (unsigned)(j * 256 + wave * 64) * 16);
#pragma unroll
      for (int j = 0; j < G::XI; ++j)
        if (xact[j]) glds16(X + (ptrdiff_t)(mb + xrow[j]) * ldx + xcol[j], slot + G::YB + (unsigned)(wave * G::NWX + j * 64) * 16);
    } else {
      // first stage of the tensor / last stage of a job: rows outside [0, R) and dY rows past the slab read as zero (plain loads; the
      // compiler's vmcnt(0) in front of the LDS writes also retires every LDS-DMA issued before)
      unsigned char* sl = ring + (size_t)sl_i * G::STAGE;
#pragma unroll
      for (int j = 0; j < G::YI; ++j) {
        const int m = mb + yrow[j];
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (m < mend) v = *reinterpret_cast<const u32x4_t*>(dY + (size_t)m * ldy + ycol[j]);
        *reinterpret_cast<u32x4_t*>(sl + (size_t)(j * 256 + tid) * 16) = v;
      }
#pragma unroll
      for (int j = 0; j < G::XI; ++j) {
        const int m = mb + xrow[j];
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (xact[j]) {
          if (m >= 0 && m < R) v = *reinterpret_cast<const u32x4_t*>(X + (size_t)m * ldx + xcol[j]);
          *reinterpret_cast<u32x4_t*>(sl + G::YB + (size_t)(wave * G::NWX + j * 64 + lane) * 16) = v;
        }
      }
    }
  };

  // lane geometry of the transposing read (see cdna guide T10): within a 16-lane group lane
  // 4q+p supplies the address of block row q, columns 4p..4p+3 and receives column (lane&15).
  const int li = lane & 15, q = li >> 2, p = li & 3;
  const int colhalf = ((lane >> 4) & 1) * 16;          // which 16 columns of the 32-wide MFMA block
  const int h = lane >> 5;                             // k half (rows 8h..8h+7 of a 16-row step)
  constexpr int NOX = TAPS < 4 ? TAPS : 4;             // distinct row phases (q + t) & 3 of the X swizzle
  int offA[2], offX[NOX];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = wco * 8 + i * 4 + (colhalf >> 3) + (p >> 1);
    offA[i] = (8 * h + q) * 256 + ((ch ^ (q << 2)) << 4) + (p & 1) * 8;
  }
#pragma unroll
  for (int t = 0; t < NOX; ++t) {
    const int ch = wci * 4 + (colhalf >> 3) + (p >> 1);
    offX[t] = (8 * h + q) * (G::XCH * 16) + ((ch ^ ((((q + t) >> 1) & 1) << 2)) << 4) + (p & 1) * 8;
  }

  bf16x8_t af[2], bfg[NB];
  auto load_frags = [&](const unsigned char* yb, const unsigned char* xb, int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned char* ya = yb + offA[i] + ks * 16 * 256;
      af[i] = tr_frag(ya, ya + 4 * 256);
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const unsigned char* xa = xb + offX[t & 3] + (ks * 16 + t) * (G::XCH * 16) + j * 128;
        bfg[t * NJ + j] = tr_frag(xa, xa + 4 * (G::XCH * 16));
      }
  };
  auto mfmas = [&]() {
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfg[t], acc[t][i], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4_t w4 = __builtin_bit_cast(u32x4_t, af[i]);
#pragma unroll
        for (int d = 0; d < 4; ++d) bsum[i] += __uint_as_float(w4[d] << 16) + __uint_as_float(w4[d] & 0xffff0000u);
      }
    }
  };

#pragma unroll 1
  for (int s = 0; s < NST - 1 && s < nst; ++s) issue(s, s);
  int cslot = 0, islot = NST - 1;                               // ring slots of the stage computed / the stage issued in this iteration
#pragma unroll 1
  for (int s = 0; s < nst; ++s) {
    // this wave's share of stage s has landed when at most (stages issued after s) x NLD of its LDS-DMAs are still in flight
    const int after = min(nst - 1, s + NST - 2) - s;
    if (after >= NST - 2) wait_vm<(NST - 2) * G::NLD>();
    else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // everyone's share has; everyone is done with stage s - 1
    if (s + NST - 1 < nst) issue(s + NST - 1, islot);                     // ... whose slot the next stage goes to
    const unsigned char* yb = ring + (size_t)cslot * G::STAGE;
    const unsigned char* xb = yb + G::YB;
    cslot = cslot + 1 == NST ? 0 : cslot + 1;
    islot = islot + 1 == NST ? 0 : islot + 1;
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks) {
      load_frags(yb, xb, ks);                  // all 2 + NB fragment reads of the k-step go out back to back; the MFMAs start as they
      __builtin_amdgcn_sched_barrier(0);       // arrive (counted lgkmcnt), the other workgroup of the CU computes meanwhile
      mfmas();
    }
  }

  // ---- slab partial: part[slab][tap][co][ci], lane = ci (128-byte rows per register)
  const int r = lane & 31;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ci = ci0 + j * 64 + wci * 32 + r;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (ci >= Cin) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = co0 + wco * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (co < Ncols) part[(((size_t)slab * TAPS + t) * Cout + co_begin + co) * Cin + ci] = acc[t * NJ + j][i][e];
        }
      }
    }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float tot = bsum[i] + __shfl_xor(bsum[i], 32);            // the two k halves of a column live 32 lanes apart
      const int co = co0 + wco * 64 + i * 32 + r;
      if (h == 0 && co < Ncols) part_bias[(size_t)slab * Cout + co_begin + co] = tot;
    }
  }
}

template <int TAPS> struct WgSel;
template <> struct WgSel<5> { static constexpr int NJ = 1, KB = WG5_KB, NST = WG5_NST; };
template <> struct WgSel<3> { static constexpr int NJ = 1, KB = WG3_KB, NST = WG3_NST; };
template <> struct WgSel<1> { static constexpr int NJ = WG1_NJ, KB = WG1_KB, NST = WG1_NST; };
template <int TAPS> using WgGeoT = WgGeo<TAPS, WgSel<TAPS>::NJ, WgSel<TAPS>::KB, WgSel<TAPS>::NST>;

template <int TAPS>
__global__ __launch_bounds__(256, 2) void gt_conv_wgrad_kernel(
    const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ dY, int ldy,
    float* __restrict__ part, float* __restrict__ part_bias, int R, int Cin, int Cout, int slab_rows)
{
  using S = WgSel<TAPS>;
  wgrad_tile<TAPS, S::NJ, S::KB, S::NST>(X, ldx, dY, ldy, part, part_bias, R, Cin, Cout, slab_rows,
                                         blockIdx.x * 128, blockIdx.y * 64 * S::NJ, blockIdx.z, 0, Cout);
}

// Batched form: every workgroup reads its (job, tile) from device tables, so ONE launch per tap count
// covers the weight gradients of a whole network (the jobs' X / dY rows stay resident until then).
template <int TAPS>
__global__ __launch_bounds__(256, 2) void gt_conv_wgrad_batched_kernel(
    const gt_wgrad_job* __restrict__ jobs, const gt_wgrad_tile* __restrict__ tiles)
{
  using S = WgSel<TAPS>;
  const gt_wgrad_tile t = tiles[blockIdx.x];
  const gt_wgrad_job j = jobs[t.job];
  wgrad_tile<TAPS, S::NJ, S::KB, S::NST>(static_cast<const bf16_t*>(j.X), j.ldx, static_cast<const bf16_t*>(j.dY), j.ldy,
                                         j.part, j.part_bias, j.R, j.Cin, j.Cout, j.slab_rows, t.co0, t.ci0, t.slab, j.co_begin, j.co_count);
}

template <int TAPS> static int wgrad_lds_attr()
{
  static int done = 0;                          // > 64 KiB of dynamic LDS needs the attribute once per kernel (outside graph capture:
  if (!done) {                                  //  run one eager step first)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_conv_wgrad_kernel<TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, WgGeoT<TAPS>::LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_conv_wgrad_batched_kernel<TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, WgGeoT<TAPS>::LDS) != hipSuccess)
      return GT_E_LAUNCH;
    done = 1;
  }
  return 0;
}

// Sum the S slab partials and map to the parameter gradient(s).  One WAVE per output channel co (4 per workgroup):
//   plain conv:   dw[co][ci][tap] (+)= dW
//   weight-norm:  w = g v / ||v||  =>  dg = <dW, v>/||v|| ;  dv = g/||v|| (dW - v <dW,v>/||v||^2)
//   bias:         db[co] (+)= sum of the slab column sums
// (Round 2 ran one 256-thread workgroup per co with two __syncthreads and three dependent global round trips for <= 960 elements:
// 144 us for the decoder's 18 k rows, against ~60 us of HBM time for the partials, v and dv.  A wave needs no workgroup barrier —
// its row goes through its own LDS strip for the [tap][ci] -> [ci][tap] transposition — and 32+ rows are in flight per CU.)
constexpr int WNB_ROWS = 4;
__device__ __forceinline__ void weightnorm_bwd_row(
    const float* __restrict__ part, const float* __restrict__ part_bias, int S, const float* __restrict__ v,
    const float* __restrict__ g, const float* __restrict__ inv_norm, float* __restrict__ dv, float* __restrict__ dg,
    float* __restrict__ dbias, int Cout, int Cin, int taps, int accumulate, int co, float* dws)
{
  const int lane = threadIdx.x & 63, n = Cin * taps;
  if (dbias && lane == 0) {
    float sb = 0.f;
    for (int k = 0; k < S; ++k) sb += part_bias[(size_t)k * Cout + co];
    dbias[co] = accumulate ? dbias[co] + sb : sb;
  }
  // coalesced pass over the partials: consecutive lanes = consecutive ci of one (slab, tap) row
  const size_t sstride = (size_t)taps * Cout * Cin;              // one slab
  for (int tap = 0; tap < taps; ++tap) {
    const float* pt = part + ((size_t)tap * Cout + co) * Cin;
    for (int ci = lane; ci < Cin; ci += 64) {
      const float* pp = pt + ci;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;               // 4 loads in flight per lane, not a dependent chain
      int k = 0;
      for (; k + 4 <= S; k += 4) {
        const float a0 = pp[(size_t)k * sstride], a1 = pp[(size_t)(k + 1) * sstride];
        const float a2 = pp[(size_t)(k + 2) * sstride], a3 = pp[(size_t)(k + 3) * sstride];
        s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      }
      for (; k < S; ++k) s0 += pp[(size_t)k * sstride];
      dws[ci * taps + tap] = (s0 + s1) + (s2 + s3);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float* vr = v + (size_t)co * n;
  float* dr = dv + (size_t)co * n;
  if (!g) {
    for (int i = lane; i < n; i += 64) dr[i] = accumulate ? dr[i] + dws[i] : dws[i];
    return;
  }
  float dot = 0.f;
  for (int i = lane; i < n; i += 64) dot += dws[i] * vr[i];
  const float d = wave_sum(dot);
  const float inv = inv_norm[co], gg = g[co];
  if (lane == 0) dg[co] = accumulate ? dg[co] + d * inv : d * inv;
  const float a = gg * inv, bcoef = gg * d * inv * inv * inv;
  for (int i = lane; i < n; i += 64) {
    const float val = a * dws[i] - bcoef * vr[i];
    dr[i] = accumulate ? dr[i] + val : val;
  }
}

__global__ __launch_bounds__(64 * WNB_ROWS) void gt_weightnorm_bwd_kernel(
    const float* __restrict__ part, const float* __restrict__ part_bias, int S, const float* __restrict__ v,
    const float* __restrict__ g, const float* __restrict__ inv_norm, float* __restrict__ dv, float* __restrict__ dg,
    float* __restrict__ dbias, int Cout, int Cin, int taps, int accumulate)
{
  extern __shared__ float dws[];               // [WNB_ROWS][Cin*taps] summed dW of a co, natural (ci, tap) order
  const int w = threadIdx.x >> 6, co = blockIdx.x * WNB_ROWS + w;
  if (co >= Cout) return;
  weightnorm_bwd_row(part, part_bias, S, v, g, inv_norm, dv, dg, dbias, Cout, Cin, taps, accumulate, co, dws + (size_t)w * Cin * taps);
}

// Batched form: one wave per output channel of ANY conv; the job is found by bisection on row_start.
__global__ __launch_bounds__(64 * WNB_ROWS) void gt_weightnorm_bwd_batched_kernel(const gt_wnb_job* __restrict__ jobs, int n_jobs, int total_rows,
                                                                                int max_row_elems)
{
  extern __shared__ float dws[];
  const int w = threadIdx.x >> 6, row = blockIdx.x * WNB_ROWS + w;
  if (row >= total_rows) return;
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].row_start <= row) lo = mid; else hi = mid - 1;
  }
  const gt_wnb_job j = jobs[lo];
  weightnorm_bwd_row(j.part, j.part_bias, j.S, j.v, j.g, j.inv_norm, j.dv, j.dg, j.dbias, j.Cout, j.Cin, j.taps,
                     j.accumulate, row - j.row_start, dws + (size_t)w * max_row_elems);
}

// Column sums over rows: out[n] (+)= sum_m Y[m, n] (bias gradients).  bf16 or fp32 input.
template <bool F32>
__global__ __launch_bounds__(256) void gt_colsum_kernel(const void* __restrict__ Y, int ldy, float* __restrict__ out,
                                                        int R, int N, int rows_per_block)
{
  // block (x: 64-column group, y: row slab); thread = (column c = tid&63, row phase tid>>6)
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(R, m0 + rows_per_block);
  float s = 0.f;
  if (c < N) {
    for (int m = m0 + ph; m < m1; m += 4)
      s += F32 ? static_cast<const float*>(Y)[(size_t)m * ldy + c] : bf2f(static_cast<const bf16_t*>(Y)[(size_t)m * ldy + c]);
  }
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && c < N) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

extern "C" int gt_conv_wgrad_ci_tile(int taps)
{
  return taps == 1 ? 64 * WgSel<1>::NJ : taps == 3 ? 64 * WgSel<3>::NJ : taps == 5 ? 64 * WgSel<5>::NJ : 0;
}

extern "C" size_t gt_conv_wgrad_workspace_bytes(int R, int Cin, int Cout, int taps, int* slabs_out)
{
  if (R <= 0 || Cin <= 0 || Cout <= 0 || taps <= 0) { if (slabs_out) *slabs_out = 0; return 0; }
  const int cit = gt_conv_wgrad_ci_tile(taps) > 0 ? gt_conv_wgrad_ci_tile(taps) : 64;
  const int tiles = ((Cout + 127) / 128) * ((Cin + cit - 1) / cit);
  // slabs: enough workgroups to cover the chip, but every slab costs one full dW of partial traffic
  // (written here, re-read by gt_weightnorm_bwd) — ~160 workgroups in total, at most 16 slabs
  constexpr int target = 160;
  int S = (target + tiles - 1) / tiles;
  if (S > 16) S = 16;
  const int max_s = (R + SLAB_Q - 1) / SLAB_Q;
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  if (slabs_out) *slabs_out = S;
  return ((size_t)S * taps * Cout * Cin + (size_t)S * Cout) * sizeof(float);     // weight partials | bias partials
}

extern "C" int gt_conv_wgrad_bf16(const void* X, int ldx, const void* dY, int ldy, int R, int Cin, int Cout,
                                  int taps, void* workspace, size_t workspace_bytes, void* stream)
{
  if (R <= 0 || Cin <= 0 || Cout <= 0) return GT_E_INVAL;
  if (!X || !dY || !workspace) return GT_E_INVAL;
  if (taps != 1 && taps != 3 && taps != 5) return GT_E_UNSUPPORTED;
  if ((Cin & 7) || (Cout & 7) || (ldx & 7) || (ldy & 7)) return GT_E_ALIGN;
  if (((uintptr_t)X | (uintptr_t)dY) & 15) return GT_E_ALIGN;
  int S = 0;
  if (workspace_bytes < gt_conv_wgrad_workspace_bytes(R, Cin, Cout, taps, &S)) return GT_E_INVAL;
  const int slab_rows = (((R + S - 1) / S) + SLAB_Q - 1) / SLAB_Q * SLAB_Q;
  const int cit = gt_conv_wgrad_ci_tile(taps);
  const dim3 grid((Cout + 127) / 128, (Cin + cit - 1) / cit, S);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bf16_t* x = static_cast<const bf16_t*>(X); const bf16_t* dy = static_cast<const bf16_t*>(dY);
  float* part = static_cast<float*>(workspace);
  float* pb = part + (size_t)S * taps * Cout * Cin;
  if (wgrad_lds_attr<5>() || wgrad_lds_attr<3>() || wgrad_lds_attr<1>()) return GT_E_LAUNCH;
  if (taps == 5)      hipLaunchKernelGGL(gt_conv_wgrad_kernel<5>, grid, dim3(256), WgGeoT<5>::LDS, st, x, ldx, dy, ldy, part, pb, R, Cin, Cout, slab_rows);
  else if (taps == 3) hipLaunchKernelGGL(gt_conv_wgrad_kernel<3>, grid, dim3(256), WgGeoT<3>::LDS, st, x, ldx, dy, ldy, part, pb, R, Cin, Cout, slab_rows);
  else                hipLaunchKernelGGL(gt_conv_wgrad_kernel<1>, grid, dim3(256), WgGeoT<1>::LDS, st, x, ldx, dy, ldy, part, pb, R, Cin, Cout, slab_rows);
  return gt_launch_status(__func__);
}

extern "C" int gt_weightnorm_bwd(const void* workspace, int R, const float* v, const float* g, const float* inv_norm,
                                 float* dv, float* dg, float* dbias, int Cout, int Cin, int taps, int accumulate, void* stream)
{
  if (!workspace || !v || !dv || Cout <= 0 || Cin <= 0 || taps <= 0) return GT_E_INVAL;
  if (g && (!inv_norm || !dg)) return GT_E_INVAL;
  int S = 0;
  gt_conv_wgrad_workspace_bytes(R, Cin, Cout, taps, &S);
  const size_t lds = (size_t)WNB_ROWS * Cin * taps * sizeof(float);
  if (lds > 60 * 1024) return GT_E_UNSUPPORTED;
  const float* part = static_cast<const float*>(workspace);
  hipLaunchKernelGGL(gt_weightnorm_bwd_kernel, dim3((Cout + WNB_ROWS - 1) / WNB_ROWS), dim3(64 * WNB_ROWS), lds, static_cast<hipStream_t>(stream),
                     part, part + (size_t)S * taps * Cout * Cin, S, v, g, inv_norm, dv, dg, dbias, Cout, Cin, taps, accumulate);
  return gt_launch_status(__func__);
}

extern "C" int gt_conv_wgrad_batched(const void* jobs_device, const void* tiles_device, int n_tiles5, int n_tiles3,
                                     int n_tiles1, void* stream)
{
  if (!jobs_device || !tiles_device || n_tiles5 < 0 || n_tiles3 < 0 || n_tiles1 < 0) return GT_E_INVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const gt_wgrad_job* jobs = static_cast<const gt_wgrad_job*>(jobs_device);
  const gt_wgrad_tile* tiles = static_cast<const gt_wgrad_tile*>(tiles_device);
  if (wgrad_lds_attr<5>() || wgrad_lds_attr<3>() || wgrad_lds_attr<1>()) return GT_E_LAUNCH;
  if (n_tiles5) hipLaunchKernelGGL(gt_conv_wgrad_batched_kernel<5>, dim3(n_tiles5), dim3(256), WgGeoT<5>::LDS, st, jobs, tiles);
  if (n_tiles3) hipLaunchKernelGGL(gt_conv_wgrad_batched_kernel<3>, dim3(n_tiles3), dim3(256), WgGeoT<3>::LDS, st, jobs, tiles + n_tiles5);
  if (n_tiles1) hipLaunchKernelGGL(gt_conv_wgrad_batched_kernel<1>, dim3(n_tiles1), dim3(256), WgGeoT<1>::LDS, st, jobs, tiles + n_tiles5 + n_tiles3);
  return gt_launch_status(__func__);
}

extern "C" int gt_weightnorm_bwd_batched(const void* jobs_device, int n_jobs, int total_rows, int max_row_elems, void* stream)
{
  if (!jobs_device || n_jobs <= 0 || total_rows <= 0 || max_row_elems <= 0) return GT_E_INVAL;
  const size_t lds = (size_t)WNB_ROWS * max_row_elems * sizeof(float);
  if (lds > 60 * 1024) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_weightnorm_bwd_batched_kernel, dim3((total_rows + WNB_ROWS - 1) / WNB_ROWS), dim3(64 * WNB_ROWS), lds,
                     static_cast<hipStream_t>(stream), static_cast<const gt_wnb_job*>(jobs_device), n_jobs, total_rows, max_row_elems);
  return gt_launch_status(__func__);
}

extern "C" int gt_colsum(const void* Y, int ldy, int is_f32, float* out, int R, int N, void* stream)
{
  if (!Y || !out || R < 0 || N <= 0) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  const int rows_per_block = 128;             // (512: 128 dependent loads per thread — 31 us for 17.8 k x 192 fp32 rows)
  const dim3 grid((N + 63) / 64, (R + rows_per_block - 1) / rows_per_block);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (is_f32) hipLaunchKernelGGL(gt_colsum_kernel<true>,  grid, dim3(256), 0, st, Y, ldy, out, R, N, rows_per_block);
  else        hipLaunchKernelGGL(gt_colsum_kernel<false>, grid, dim3(256), 0, st, Y, ldy, out, R, N, rows_per_block);
  return gt_launch_status(__func__);
}
