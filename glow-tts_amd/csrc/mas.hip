// Monotonic Alignment Search for gfx950 (MI355X) — hand-written HIP.
//
// Replaces reference monotonic_align/core.pyx:9-45 (serial Cython DP) and the
// copy-heavy wrapper monotonic_align/__init__.py:6-21.  Bit-exact paths.
//
// Design (one workgroup per utterance, W = ceil(T_x/64) wavefronts):
//   * Q[x,y] = max(Q[x,y-1], Q[x-1,y-1]) + v[x,y] only depends on column y-1, so the DP
//     is column-sequential / row-parallel.  Lane l of wave w owns row x = 64w + l; the
//     x-1 operand is a one-lane DPP shift (wave_shr:1).
//   * Rows are split over waves that run skewed by one 32-column chunk ("anti-diagonal
//     wavefront"): at step s wave w processes chunk s-w; the Q values of its last row
//     cross to wave w+1 through a 2-slot LDS ring, one __syncthreads() per chunk.
//   * logp chunks (64 rows x 32 columns per wave) are fetched with coalesced 16-byte
//     loads one chunk ahead (register prefetch) and staged in an XOR-swizzled LDS tile
//     so that the per-lane row reads (ds_read_b128) are bank-conflict free.
//   * max(a,b)+v == max(a+v, b+v) exactly in IEEE arithmetic (rounding is monotone), so
//     the per-column dependency chain is {add_dpp, max}; the comparison a<b that the
//     reference's backtrack re-evaluates (core.pyx:34) is emitted as ONE direction bit
//     per cell and the fp32 lattice is never stored: T_x*T_y bits live in LDS.
//   * Backtrack: one wave walks rows, not columns — per row a count-leading-zeros on the
//     32-column direction word gives the run length (<= T_x + T_y/32 serial steps).
//   * The path is written by all waves with 16-byte stores from the per-row [start,end)
//     column interval; durations and the frame->token map fall out for free.
//
// Cells outside the reference's band x in [max(0,t_x+y-t_y), min(t_x,y+1)) are computed
// too (garbage) but never read by in-band cells nor by the backtrack.
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)   // bit-exact IEEE adds/compares only
#include <stdint.h>
#include "../../include/glowtts_hip.h"

namespace {

constexpr int   CH       = 32;            // columns per chunk
constexpr int   TILE_B   = 64 * CH * 4;   // bytes of one wave's logp tile (8 KiB)
constexpr int   BND_SLOT = 36;            // floats per boundary slot (33 used)
constexpr int   BND_F    = 2 * BND_SLOT + 104;  // + dummy area for lanes != 63 -> 176 floats
constexpr float NEG      = -1e9f;         // reference max_neg_val (core.pyx:38)

struct MasArgs {
  const float* logp; const float* mask;
  const int32_t* t_x; const int32_t* t_y;
  void* path; int path_dtype;
  float* durations; int32_t* frame2token;
  int T_x, T_y; int64_t stride_b, stride_x;
  int32_t* status;
  int nchp;                                // direction words per row (odd)
};

__host__ __device__ inline int mas_nchp(int T_y) { return ((T_y + CH - 1) / CH) | 1; }

__device__ __forceinline__ float dpp_wave_shr1(float old, float src) {
  // lane l <- src[l-1]; lane 0 keeps `old` (bound_ctrl off)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src),
                                                    0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}

// One 32-column chunk of the forward DP for one wave.  DIAG: the chunk may contain the
// diagonal cell x==y of some lane (only chunks 2w, 2w+1 of wave w).  W0: wave 0 (row 0 has
// no x-1 neighbour: v_prev = 0 at y==0, else max_neg_val — core.pyx:23-27).
template <bool DIAG, bool W0>
__device__ __forceinline__ void mas_chunk(const float* __restrict__ tile, const float* __restrict__ bin,
                                          float* __restrict__ bout, int lane, int x, int c,
                                          float& Q, float& carry, unsigned& dir_out)
{
  unsigned dir = 0;
  const int sw = (lane >> 1) & 7;
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) {
    const float4 v4 = *reinterpret_cast<const float4*>(tile + lane * CH + ((q ^ sw) << 2));
    float4 b4;
    if (W0) {
      b4 = make_float4(NEG, NEG, NEG, NEG);
      if (q == 0 && c == 0) b4.x = 0.0f;                       // core.pyx:24-25
    } else {
      b4 = *reinterpret_cast<const float4*>(bin + q * 4);      // broadcast read
      if (q == 0) b4.x = carry;
    }
    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = q * 4 + i;
      const float P = dpp_wave_shr1(bb[i], Q);                 // Q[x-1, y-1]
      float A = Q;                                             // Q[x,   y-1]
      bool d = false;
      if (DIAG) { d = (x == c * CH + j); A = d ? NEG : A; }    // core.pyx:19-20
      const bool lt = (A < P);                                 // core.pyx:34 predicate / :30 select
      const float qa = A + vv[i];
      const float qp = P + vv[i];
      Q = (qp > qa) ? qp : qa;                                 // == max(A,P)+v bit-exactly
      dir |= ((lt || d) ? 1u : 0u) << j;
      bout[j + 1] = Q;                                         // lane 63: boundary row; others: dummy
    }
  }
  if (!W0) carry = bin[CH];                                    // Q[64w-1, last column of chunk]
  dir_out = dir;
}

template <bool VEC>
__global__ __launch_bounds__(1024) void gt_mas_kernel(MasArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b    = blockIdx.x;
  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int w    = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W    = blockDim.x >> 6;
  const int NT   = blockDim.x;
  const int T_x = a.T_x, T_y = a.T_y, NCHP = a.nchp;

  int t_x = a.t_x[b], t_y = a.t_y[b];
  {
    int st = 0;
    if (t_x < 0 || t_y < 0 || t_x > T_x || t_y > T_y) st |= GT_MAS_ST_BAD_LEN;
    else if (t_x > t_y) st |= GT_MAS_ST_TX_GT_TY;
    if (st) { if (tid == 0 && a.status) atomicOr(a.status, st); t_x = 0; t_y = 0; }
    if (t_x == 0 || t_y == 0) { t_x = 0; t_y = 0; }            // empty utterance -> all-zero path
  }

  float*    tile   = reinterpret_cast<float*>(smem) + w * (64 * CH);
  float*    bndall = reinterpret_cast<float*>(smem + (size_t)W * TILE_B);
  unsigned* dirs   = reinterpret_cast<unsigned*>(bndall + W * BND_F);
  int*      starts = reinterpret_cast<int*>(dirs + (size_t)W * 64 * NCHP);   // [W*64 + 1]

  const int nch  = (t_y + CH - 1) / CH;
  const int Wact = (t_x + 63) >> 6;
  const bool wave_active = w < Wact;
  const int x = w * 64 + lane;

  const float* lp = a.logp + (int64_t)b * a.stride_b;
  const float* mp = a.mask ? a.mask + (int64_t)b * a.stride_b : nullptr;

  float4 r[8];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = w * 64 + i * 8 + (lane >> 3);
      const int col = c * CH + (lane & 7) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < T_x) {
        const float* p = lp + (int64_t)row * a.stride_x + col;
        if (VEC) {
          if (col < T_y) {                                     // T_y % 4 == 0 in VEC mode
            v = *reinterpret_cast<const float4*>(p);
            if (mp) {
              const float4 m = *reinterpret_cast<const float4*>(mp + (int64_t)row * a.stride_x + col);
              v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;   // __init__.py:11
            }
          }
        } else {
          const float* pm = mp ? mp + (int64_t)row * a.stride_x + col : nullptr;
          if (col + 0 < T_y) { v.x = p[0]; if (pm) v.x *= pm[0]; }
          if (col + 1 < T_y) { v.y = p[1]; if (pm) v.y *= pm[1]; }
          if (col + 2 < T_y) { v.z = p[2]; if (pm) v.z *= pm[2]; }
          if (col + 3 < T_y) { v.w = p[3]; if (pm) v.w *= pm[3]; }
        }
      }
      r[i] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = i * 8 + (lane >> 3);
      const int q  = lane & 7;
      *reinterpret_cast<float4*>(tile + rr * CH + ((q ^ ((rr >> 1) & 7)) << 2)) = r[i];
    }
  };

  // ---------------- forward DP: direction bits into LDS ----------------
  if (wave_active && 2 * w < nch) { load_chunk(2 * w); store_tile(); }
  __syncthreads();

  float Q = 0.0f, carry = NEG;
  const float* bin_base  = bndall + (w > 0 ? (w - 1) : 0) * BND_F;       // producer = wave w-1
  float*       bout_base = bndall + w * BND_F;
  const int nsteps = (nch > 0) ? nch + Wact - 1 : 0;
  for (int s = 0; s < nsteps; ++s) {
    const int c = s - w;
    if (wave_active && c >= 2 * w && c < nch) {
      if (c + 1 < nch) load_chunk(c + 1);
      const float* bin = bin_base + (c & 1) * BND_SLOT;
      float* bout = (lane == 63) ? (bout_base + (c & 1) * BND_SLOT)
                                 : (bout_base + 2 * BND_SLOT + lane);
      unsigned dir;
      const bool diag = (c >> 1) == w;
      if (w == 0) {
        if (diag) mas_chunk<true,  true >(tile, bin, bout, lane, x, c, Q, carry, dir);
        else      mas_chunk<false, true >(tile, bin, bout, lane, x, c, Q, carry, dir);
      } else {
        if (diag) mas_chunk<true,  false>(tile, bin, bout, lane, x, c, Q, carry, dir);
        else      mas_chunk<false, false>(tile, bin, bout, lane, x, c, Q, carry, dir);
      }
      dirs[(size_t)x * NCHP + c] = dir;
      if (c + 1 < nch) store_tile();
    } else if (wave_active && w > 0 && c == 2 * w - 1) {
      // the chunk before this wave's first: pick up Q[64w-1, 64w-1] as carry-in
      carry = bin_base[(c & 1) * BND_SLOT + CH];
    }
    __syncthreads();
  }

  // ---------------- backtrack (wave 0): rows, not columns ----------------
  if (w == 0) {
    int idx = t_x - 1;
    int y   = t_y - 1;
    while (y >= 0 && idx > 0) {                      // idx==0: no further moves (core.pyx:34 `index != 0`)
      const int c  = y >> 5;
      const int r  = idx - lane;                      // lane l holds the word of row idx-l
      const unsigned wv = (r >= 0) ? dirs[(size_t)r * NCHP + c] : 0u;
      const int base = idx;
      const int ylo  = c << 5;
      while (true) {
        const unsigned word = __builtin_amdgcn_readlane(wv, base - idx);
        const unsigned m = word << (31 - (y & 31));   // bit of column y -> bit 31
        if (m == 0u) { y = ylo - 1; break; }          // stays on this row down to the chunk start
        const int yp = y - __builtin_clz(m);          // first column (going down) with a diagonal move
        if (lane == 0) starts[idx] = yp;              // row idx occupies columns [yp, ...]
        idx -= 1;
        y = yp - 1;
        if (idx == 0 || y < ylo) break;
      }
    }
    if (lane == 0) { if (t_x > 0) starts[0] = 0; starts[t_x] = t_y; }
  }
  __syncthreads();

  // ---------------- outputs ----------------
  if (a.durations) {
    for (int xx = tid; xx < T_x; xx += NT)
      a.durations[(int64_t)b * T_x + xx] = (xx < t_x) ? (float)(starts[xx + 1] - starts[xx]) : 0.0f;
  }
  if (a.frame2token) {
    for (int yy = tid; yy < T_y; yy += NT) {
      int tok = -1;
      if (yy < t_y) {                                 // largest row with starts[row] <= yy
        int lo = 0, hi = t_x - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (starts[mid] <= yy) lo = mid; else hi = mid - 1; }
        tok = lo;
      }
      a.frame2token[(int64_t)b * T_y + yy] = tok;
    }
  }

  const int64_t obase = (int64_t)b * T_x * T_y;
  const int dt = a.path_dtype;
  if (VEC) {
    const int T4 = T_y >> 2;
    for (int row = w; row < T_x; row += W) {
      int s0 = 0, e0 = 0;
      if (row < t_x) { s0 = starts[row]; e0 = starts[row + 1]; }
      for (int q = lane; q < T4; q += 64) {
        const int y0 = q << 2;
        const unsigned b0 = (y0     >= s0 && y0     < e0);
        const unsigned b1 = (y0 + 1 >= s0 && y0 + 1 < e0);
        const unsigned b2 = (y0 + 2 >= s0 && y0 + 2 < e0);
        const unsigned b3 = (y0 + 3 >= s0 && y0 + 3 < e0);
        const int64_t o = obase + (int64_t)row * T_y + y0;
        if (dt == GT_DT_F32) {
          *reinterpret_cast<float4*>(static_cast<float*>(a.path) + o) =
              make_float4((float)b0, (float)b1, (float)b2, (float)b3);
        } else if (dt == GT_DT_I32) {
          *reinterpret_cast<int4*>(static_cast<int32_t*>(a.path) + o) = make_int4(b0, b1, b2, b3);
        } else if (dt == GT_DT_F16) {                 // 1.0h = 0x3C00
          *reinterpret_cast<uint2*>(static_cast<uint16_t*>(a.path) + o) =
              make_uint2((b0 * 0x3C00u) | ((b1 * 0x3C00u) << 16), (b2 * 0x3C00u) | ((b3 * 0x3C00u) << 16));
        } else if (dt == GT_DT_BF16) {                // 1.0bf16 = 0x3F80
          *reinterpret_cast<uint2*>(static_cast<uint16_t*>(a.path) + o) =
              make_uint2((b0 * 0x3F80u) | ((b1 * 0x3F80u) << 16), (b2 * 0x3F80u) | ((b3 * 0x3F80u) << 16));
        } else {                                      // GT_DT_U8
          *reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(a.path) + o) =
              b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
      }
    }
  } else {
    for (int row = w; row < T_x; row += W) {
      int s0 = 0, e0 = 0;
      if (row < t_x) { s0 = starts[row]; e0 = starts[row + 1]; }
      for (int yy = lane; yy < T_y; yy += 64) {
        const unsigned bit = (yy >= s0 && yy < e0);
        const int64_t o = obase + (int64_t)row * T_y + yy;
        if      (dt == GT_DT_F32)  static_cast<float*>(a.path)[o]    = (float)bit;
        else if (dt == GT_DT_I32)  static_cast<int32_t*>(a.path)[o]  = (int32_t)bit;
        else if (dt == GT_DT_F16)  static_cast<uint16_t*>(a.path)[o] = (uint16_t)(bit * 0x3C00u);
        else if (dt == GT_DT_BF16) static_cast<uint16_t*>(a.path)[o] = (uint16_t)(bit * 0x3F80u);
        else                       static_cast<uint8_t*>(a.path)[o]  = (uint8_t)bit;
      }
    }
  }
}

__global__ void gt_mas_lengths_kernel(const float* mask, int32_t* t_x, int32_t* t_y,
                                      int T_x, int T_y, int64_t stride_b, int64_t stride_x)
{
  // one workgroup (256 threads) per utterance; fp32 sums like numpy's mask.sum(...)
  __shared__ float red[2][4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* m = mask + (int64_t)b * stride_b;
  float sx = 0.f, sy = 0.f;
  for (int i = tid; i < T_x; i += blockDim.x) sx += m[(int64_t)i * stride_x];
  for (int j = tid; j < T_y; j += blockDim.x) sy += m[j];
  for (int o = 32; o > 0; o >>= 1) { sx += __shfl_down(sx, o); sy += __shfl_down(sy, o); }
  if (lane == 0) { red[0][w] = sx; red[1][w] = sy; }
  __syncthreads();
  if (tid == 0) {
    t_x[b] = (int32_t)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    t_y[b] = (int32_t)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

}  // namespace

extern "C" size_t gt_mas_lds_bytes(int T_x, int T_y)
{
  if (T_x <= 0 || T_y <= 0) return 0;
  const size_t W = (size_t)(T_x + 63) / 64;
  return W * TILE_B + W * BND_F * 4 + W * 64 * (size_t)mas_nchp(T_y) * 4 + (W * 64 + 1) * 4 + 12;
}

extern "C" int gt_mas_f32(const float* logp, const float* mask,
                          const int32_t* t_x, const int32_t* t_y,
                          void* path, int path_dtype,
                          float* durations, int32_t* frame2token,
                          int B, int T_x, int T_y, int64_t stride_b, int64_t stride_x,
                          int32_t* status, void* stream)
{
  if (B < 0 || T_x < 0 || T_y < 0) return GT_E_INVAL;
  if (B == 0 || T_x == 0 || T_y == 0) return GT_OK;            // nothing to write
  if (!logp || !t_x || !t_y || !path) return GT_E_INVAL;
  if (path_dtype < GT_DT_F32 || path_dtype > GT_DT_U8) return GT_E_INVAL;
  if (stride_x < T_y || stride_b < (int64_t)T_x * stride_x) return GT_E_INVAL;
  if (T_x > 1024) return GT_E_UNSUPPORTED;
  const size_t lds = gt_mas_lds_bytes(T_x, T_y);
  if (lds > 160 * 1024) return GT_E_UNSUPPORTED;

  MasArgs a;
  a.logp = logp; a.mask = mask; a.t_x = t_x; a.t_y = t_y; a.path = path; a.path_dtype = path_dtype;
  a.durations = durations; a.frame2token = frame2token; a.T_x = T_x; a.T_y = T_y;
  a.stride_b = stride_b; a.stride_x = stride_x; a.status = status; a.nchp = mas_nchp(T_y);

  const int W = (T_x + 63) / 64;
  const bool vec = (T_y % 4 == 0) && (stride_x % 4 == 0) && (stride_b % 4 == 0) &&
                   ((uintptr_t)logp % 16 == 0) && (!mask || (uintptr_t)mask % 16 == 0) &&
                   ((uintptr_t)path % 16 == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (vec) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_mas_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return GT_E_LAUNCH;
    hipLaunchKernelGGL(gt_mas_kernel<true>, dim3(B), dim3(W * 64), lds, st, a);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_mas_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return GT_E_LAUNCH;
    hipLaunchKernelGGL(gt_mas_kernel<false>, dim3(B), dim3(W * 64), lds, st, a);
  }
  return hipGetLastError() == hipSuccess ? GT_OK : GT_E_LAUNCH;
}

extern "C" int gt_mas_lengths_from_mask_f32(const float* mask, int32_t* t_x, int32_t* t_y,
                                            int B, int T_x, int T_y, int64_t stride_b, int64_t stride_x,
                                            void* stream)
{
  if (B < 0 || T_x <= 0 || T_y <= 0) return GT_E_INVAL;
  if (B == 0) return GT_OK;
  if (!mask || !t_x || !t_y) return GT_E_INVAL;
  hipLaunchKernelGGL(gt_mas_lengths_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     mask, t_x, t_y, T_x, T_y, stride_b, stride_x);
  return hipGetLastError() == hipSuccess ? GT_OK : GT_E_LAUNCH;
}
