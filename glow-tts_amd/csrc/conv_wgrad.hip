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

#ifndef WG_KB
#define WG_KB 64
#endif
constexpr int KB = WG_KB;                     // rows per staging step
constexpr int MAXTAPS = 5;
constexpr int YP = 128 + 32;                  // dY tile pitch in halfs (320 B)
constexpr int XP = 64 + 32;                   // X tile pitch in halfs (192 B)
constexpr int XROWS = KB + MAXTAPS - 1;

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4_t;

__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* p0, const bf16_t* p1) {
  // two transposing reads: rows kb..kb+3 and kb+4..kb+7 of this lane's column
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)p1);
  typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// One workgroup: 128 output channels [co0, co0+128) of the dY view (Ncols columns wide; dY column c is
// output channel co_begin + c of a conv with Cout channels) x 64 input channels x all taps, rows of one slab.
template <int TAPS>
__device__ __forceinline__ void wgrad_tile(
    const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ dY, int ldy,
    float* __restrict__ part, float* __restrict__ part_bias, int R, int Cin, int Cout, int slab_rows,
    int co0, int ci0, int slab, int co_begin, int Ncols)
{
  __shared__ __attribute__((aligned(16))) bf16_t Ys[2][KB * YP];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2][XROWS * XP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int padl = TAPS >> 1;
  const int mbeg = slab * slab_rows;
  const int mend = min(R, mbeg + slab_rows);
  const int wco = wave >> 1, wci = wave & 1;                    // this wave's 64 output channels / 32 input channels of the tile
  const bool do_bias = (ci0 == 0) && part_bias && wci == 0;     // column sums of dY ride along as one more MFMA per co block

  f32x16_t acc[TAPS][2], accb[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { accb[0][e] = 0.0f; accb[1][e] = 0.0f; }
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.0f;

  // lane geometry of the transposing read (see cdna guide T10): within a 16-lane group lane
  // 4q+p supplies the address of block row q, columns 4p..4p+3 and receives column (lane&15).
  const int li = lane & 15, q = li >> 2, p = li & 3;
  const int colhalf = ((lane >> 4) & 1) * 16;          // which 16 columns of the 32-wide MFMA block
  const int h = lane >> 5;                             // k half (rows 8h..8h+7 of a 16-row step)
  typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8_t;
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  // register prefetch of the next KB-row step
  constexpr int YCH = KB * 16 / 256, XCH = (XROWS * 8 + 255) / 256;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t yr[YCH], xr[XCH];
#define WG_LOAD(mb) do { \
    _Pragma("unroll") for (int i_ = 0; i_ < YCH; ++i_) { \
      const int chunk = tid + 256 * i_, row = chunk >> 4, c8 = chunk & 15, m = (mb) + row, co = co0 + c8 * 8; \
      u32x4_t v = {0u, 0u, 0u, 0u}; \
      if (m < mend && co < Ncols) v = *reinterpret_cast<const u32x4_t*>(dY + (size_t)m * ldy + co); \
      yr[i_] = v; } \
    _Pragma("unroll") for (int i_ = 0; i_ < XCH; ++i_) { \
      const int chunk = tid + 256 * i_, row = chunk >> 3, c8 = chunk & 7, m = (mb) - padl + row, ci = ci0 + c8 * 8; \
      u32x4_t v = {0u, 0u, 0u, 0u}; \
      if (row < KB + TAPS - 1 && m >= 0 && m < R && ci < Cin) v = *reinterpret_cast<const u32x4_t*>(X + (size_t)m * ldx + ci); \
      xr[i_] = v; } } while (0)
#define WG_STORE(bf) do { \
    _Pragma("unroll") for (int i_ = 0; i_ < YCH; ++i_) { \
      const int chunk = tid + 256 * i_, row = chunk >> 4, c8 = chunk & 15; \
      *reinterpret_cast<u32x4_t*>(&Ys[bf][row * YP + c8 * 8]) = yr[i_]; } \
    _Pragma("unroll") for (int i_ = 0; i_ < XCH; ++i_) { \
      const int chunk = tid + 256 * i_, row = chunk >> 3, c8 = chunk & 7; \
      if (row < KB + TAPS - 1) *reinterpret_cast<u32x4_t*>(&Xs[bf][row * XP + c8 * 8]) = xr[i_]; } } while (0)

  if (mbeg < mend) { WG_LOAD(mbeg); WG_STORE(0); }
  __syncthreads();
  int buf = 0;
  for (int mb = mbeg; mb < mend; mb += KB, buf ^= 1) {
    const bool has = mb + KB < mend;
    if (has) WG_LOAD(mb + KB);
    const bf16_t* ysb = Ys[buf];
    const bf16_t* xsb = Xs[buf];
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks) {
      const int kb = ks * 16 + 8 * h;
      // wave (wco, wci) owns 64 output channels x 32 input channels x all taps: per k-step 2 dY^T fragments + TAPS X fragments for
      // 2 * TAPS MFMAs (the first mapping — 32 co x 64 ci per wave — read 1 + 2 * TAPS fragments for the same MFMAs, and the
      // transposing LDS reads, 2 per fragment, were the kernel's limit: 0.24 of the MFMA peak at k = 5)
      bf16x8_t af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16_t* ya = ysb + (kb + q) * YP + wco * 64 + i * 32 + colhalf + 4 * p;
        af[i] = tr_frag(ya, ya + 4 * YP);
        if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], ones, accb[i], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const bf16_t* xa = xsb + (kb + t + q) * XP + wci * 32 + colhalf + 4 * p;
        const bf16x8_t bfg = tr_frag(xa, xa + 4 * XP);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfg, acc[t][i], 0, 0, 0);
      }
    }
    if (has) WG_STORE(buf ^ 1);
    __syncthreads();
  }
#undef WG_LOAD
#undef WG_STORE

  // ---- slab partial: part[slab][tap][co][ci], lane = ci (128-byte rows per register)
  const int r = lane & 31;
  const int ci = ci0 + wci * 32 + r;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (ci >= Cin) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (co < Ncols) part[(((size_t)slab * TAPS + t) * Cout + co_begin + co) * Cin + ci] = acc[t][i][e];
      }
    }
  if (do_bias && r == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (co < Ncols) part_bias[(size_t)slab * Cout + co_begin + co] = accb[i][e];
      }
  }
}

template <int TAPS>
__global__ __launch_bounds__(256, 1) void gt_conv_wgrad_kernel(
    const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ dY, int ldy,
    float* __restrict__ part, float* __restrict__ part_bias, int R, int Cin, int Cout, int slab_rows)
{
  wgrad_tile<TAPS>(X, ldx, dY, ldy, part, part_bias, R, Cin, Cout, slab_rows,
                   blockIdx.x * 128, blockIdx.y * 64, blockIdx.z, 0, Cout);
}

// Batched form: every workgroup reads its (job, tile) from device tables, so ONE launch per tap count
// covers the weight gradients of a whole network (the jobs' X / dY rows stay resident until then).
template <int TAPS>
__global__ __launch_bounds__(256, 1) void gt_conv_wgrad_batched_kernel(
    const gt_wgrad_job* __restrict__ jobs, const gt_wgrad_tile* __restrict__ tiles)
{
  const gt_wgrad_tile t = tiles[blockIdx.x];
  const gt_wgrad_job j = jobs[t.job];
  wgrad_tile<TAPS>(static_cast<const bf16_t*>(j.X), j.ldx, static_cast<const bf16_t*>(j.dY), j.ldy,
                   j.part, j.part_bias, j.R, j.Cin, j.Cout, j.slab_rows, t.co0, t.ci0, t.slab, j.co_begin, j.co_count);
}

// Sum the S slab partials and map to the parameter gradient(s).  One workgroup per co.
//   plain conv:   dw[co][ci][tap] (+)= dW
//   weight-norm:  w = g v / ||v||  =>  dg = <dW, v>/||v|| ;  dv = g/||v|| (dW - v <dW,v>/||v||^2)
//   bias:         db[co] (+)= sum of the slab column sums
__device__ __forceinline__ void weightnorm_bwd_row(
    const float* __restrict__ part, const float* __restrict__ part_bias, int S, const float* __restrict__ v,
    const float* __restrict__ g, const float* __restrict__ inv_norm, float* __restrict__ dv, float* __restrict__ dg,
    float* __restrict__ dbias, int Cout, int Cin, int taps, int accumulate, int co, float* dws)
{
  __shared__ float red[4];
  const int tid = threadIdx.x, n = Cin * taps;
  if (dbias && tid == 0) {
    float sb = 0.f;
    for (int k = 0; k < S; ++k) sb += part_bias[(size_t)k * Cout + co];
    dbias[co] = accumulate ? dbias[co] + sb : sb;
  }
  // coalesced pass over the partials: consecutive threads = consecutive ci of one (slab, tap) row
  const size_t sstride = (size_t)taps * Cout * Cin;              // one slab
  for (int i = tid; i < n; i += 256) {
    const int tap = i / Cin, ci = i - tap * Cin;
    const float* pp = part + ((size_t)tap * Cout + co) * Cin + ci;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                 // 4 loads in flight per thread, not a dependent chain
    int k = 0;
    for (; k + 4 <= S; k += 4) {
      const float a0 = pp[(size_t)k * sstride], a1 = pp[(size_t)(k + 1) * sstride];
      const float a2 = pp[(size_t)(k + 2) * sstride], a3 = pp[(size_t)(k + 3) * sstride];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; k < S; ++k) s0 += pp[(size_t)k * sstride];
    dws[ci * taps + tap] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (!g) {
    for (int i = tid; i < n; i += 256) {
      const size_t o = (size_t)co * n + i;
      dv[o] = accumulate ? dv[o] + dws[i] : dws[i];
    }
    return;
  }
  float dot = 0.f;
  for (int i = tid; i < n; i += 256) dot += dws[i] * v[(size_t)co * n + i];
  dot = wave_sum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  const float d = red[0] + red[1] + red[2] + red[3];
  const float inv = inv_norm[co], gg = g[co];
  if (tid == 0) dg[co] = accumulate ? dg[co] + d * inv : d * inv;
  const float a = gg * inv, bcoef = gg * d * inv * inv * inv;
  for (int i = tid; i < n; i += 256) {
    const size_t o = (size_t)co * n + i;
    const float val = a * dws[i] - bcoef * v[o];
    dv[o] = accumulate ? dv[o] + val : val;
  }
}

__global__ __launch_bounds__(256) void gt_weightnorm_bwd_kernel(
    const float* __restrict__ part, const float* __restrict__ part_bias, int S, const float* __restrict__ v,
    const float* __restrict__ g, const float* __restrict__ inv_norm, float* __restrict__ dv, float* __restrict__ dg,
    float* __restrict__ dbias, int Cout, int Cin, int taps, int accumulate)
{
  extern __shared__ float dws[];               // [Cin*taps] summed dW of this co, natural (ci, tap) order
  weightnorm_bwd_row(part, part_bias, S, v, g, inv_norm, dv, dg, dbias, Cout, Cin, taps, accumulate, blockIdx.x, dws);
}

// Batched form: one workgroup per output channel of ANY conv; the job is found by bisection on row_start.
__global__ __launch_bounds__(256) void gt_weightnorm_bwd_batched_kernel(const gt_wnb_job* __restrict__ jobs, int n_jobs)
{
  extern __shared__ float dws[];
  const int row = blockIdx.x;
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].row_start <= row) lo = mid; else hi = mid - 1;
  }
  const gt_wnb_job j = jobs[lo];
  weightnorm_bwd_row(j.part, j.part_bias, j.S, j.v, j.g, j.inv_norm, j.dv, j.dg, j.dbias, j.Cout, j.Cin, j.taps,
                     j.accumulate, row - j.row_start, dws);
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

extern "C" size_t gt_conv_wgrad_workspace_bytes(int R, int Cin, int Cout, int taps, int* slabs_out)
{
  if (R <= 0 || Cin <= 0 || Cout <= 0 || taps <= 0) { if (slabs_out) *slabs_out = 0; return 0; }
  const int tiles = ((Cout + 127) / 128) * ((Cin + 63) / 64);
  // slabs: enough workgroups to cover the chip, but every slab costs one full dW of partial traffic
  // (written here, re-read by gt_weightnorm_bwd) — ~160 workgroups in total, at most 16 slabs
  constexpr int target = 160;
  int S = (target + tiles - 1) / tiles;
  if (S > 16) S = 16;
  const int max_s = (R + KB - 1) / KB;
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
  const int slab_rows = (((R + S - 1) / S) + KB - 1) / KB * KB;
  const dim3 grid((Cout + 127) / 128, (Cin + 63) / 64, S);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bf16_t* x = static_cast<const bf16_t*>(X); const bf16_t* dy = static_cast<const bf16_t*>(dY);
  float* part = static_cast<float*>(workspace);
  float* pb = part + (size_t)S * taps * Cout * Cin;
  if (taps == 5)      hipLaunchKernelGGL(gt_conv_wgrad_kernel<5>, grid, dim3(256), 0, st, x, ldx, dy, ldy, part, pb, R, Cin, Cout, slab_rows);
  else if (taps == 3) hipLaunchKernelGGL(gt_conv_wgrad_kernel<3>, grid, dim3(256), 0, st, x, ldx, dy, ldy, part, pb, R, Cin, Cout, slab_rows);
  else                hipLaunchKernelGGL(gt_conv_wgrad_kernel<1>, grid, dim3(256), 0, st, x, ldx, dy, ldy, part, pb, R, Cin, Cout, slab_rows);
  return gt_launch_status(__func__);
}

extern "C" int gt_weightnorm_bwd(const void* workspace, int R, const float* v, const float* g, const float* inv_norm,
                                 float* dv, float* dg, float* dbias, int Cout, int Cin, int taps, int accumulate, void* stream)
{
  if (!workspace || !v || !dv || Cout <= 0 || Cin <= 0 || taps <= 0) return GT_E_INVAL;
  if (g && (!inv_norm || !dg)) return GT_E_INVAL;
  int S = 0;
  gt_conv_wgrad_workspace_bytes(R, Cin, Cout, taps, &S);
  const size_t lds = (size_t)Cin * taps * sizeof(float);
  if (lds > 60 * 1024) return GT_E_UNSUPPORTED;
  const float* part = static_cast<const float*>(workspace);
  hipLaunchKernelGGL(gt_weightnorm_bwd_kernel, dim3(Cout), dim3(256), lds, static_cast<hipStream_t>(stream),
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
  if (n_tiles5) hipLaunchKernelGGL(gt_conv_wgrad_batched_kernel<5>, dim3(n_tiles5), dim3(256), 0, st, jobs, tiles);
  if (n_tiles3) hipLaunchKernelGGL(gt_conv_wgrad_batched_kernel<3>, dim3(n_tiles3), dim3(256), 0, st, jobs, tiles + n_tiles5);
  if (n_tiles1) hipLaunchKernelGGL(gt_conv_wgrad_batched_kernel<1>, dim3(n_tiles1), dim3(256), 0, st, jobs, tiles + n_tiles5 + n_tiles3);
  return gt_launch_status(__func__);
}

extern "C" int gt_weightnorm_bwd_batched(const void* jobs_device, int n_jobs, int total_rows, int max_row_elems, void* stream)
{
  if (!jobs_device || n_jobs <= 0 || total_rows <= 0 || max_row_elems <= 0) return GT_E_INVAL;
  const size_t lds = (size_t)max_row_elems * sizeof(float);
  if (lds > 60 * 1024) return GT_E_UNSUPPORTED;
  hipLaunchKernelGGL(gt_weightnorm_bwd_batched_kernel, dim3(total_rows), dim3(256), lds, static_cast<hipStream_t>(stream),
                     static_cast<const gt_wnb_job*>(jobs_device), n_jobs);
  return gt_launch_status(__func__);
}

extern "C" int gt_colsum(const void* Y, int ldy, int is_f32, float* out, int R, int N, void* stream)
{
  if (!Y || !out || R < 0 || N <= 0) return GT_E_INVAL;
  if (R == 0) return GT_OK;
  const int rows_per_block = 512;
  const dim3 grid((N + 63) / 64, (R + rows_per_block - 1) / rows_per_block);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (is_f32) hipLaunchKernelGGL(gt_colsum_kernel<true>,  grid, dim3(256), 0, st, Y, ldy, out, R, N, rows_per_block);
  else        hipLaunchKernelGGL(gt_colsum_kernel<false>, grid, dim3(256), 0, st, Y, ldy, out, R, N, rows_per_block);
  return gt_launch_status(__func__);
}
