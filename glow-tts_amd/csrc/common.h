// Shared device helpers for the gfx950 kernels (bf16 storage, MFMA fragment types, RNG).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef uint16_t bf16_t;   // raw bf16 bits in memory

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {       // round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack2bf(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

// The reciprocal is the hardware's (v_rcp_f32, 1 ulp): an IEEE-exact `1.0f / x` is a ten-instruction sequence, and the gate
// epilogues evaluate two of them per element (20 us of a WaveNet forward's 84 were this VALU work).  The results
// are rounded to bf16 right away, 16 bits below the ulp in question.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {      // 1 - 2/(e^{2x}+1): exact limits at +-inf
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}

// Counter-based RNG for dropout: one 32-bit hash per element index, replayable in backward.
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__device__ __forceinline__ bool drop_keep(uint32_t seed, uint32_t row, uint32_t col, uint32_t thresh) {
  // keep with probability 1-p, thresh = p * 2^32.  One finalizer round over a (seed, row, col) mix with odd multipliers:
  // half the VALU work of hashing twice, and it is paid per element in the gate conv's epilogue and its backward.
  return hash_u32(seed + row * 0x9E3779B1U + col * 0x85EBCA6BU) >= thresh;
}

// The WaveNet gate drops the tanh half and the sigmoid half of a channel independently (modules.py:153 drops all 2H conv
// outputs).  ONE hash per (row, gate channel) decides both — low 16 bits for the tanh half, high 16 bits for the sigmoid half,
// each against p * 2^16 (p = 0.05 -> 0.0500031): half the integer multiplies (v_mul_lo_u32 is a quarter-rate instruction, and
// the two hashes per gate element were ~10 % of the fused WaveNet kernels) in every kernel that applies or replays that mask.
__device__ __forceinline__ uint32_t drop_thresh16(uint32_t thresh) { return (thresh + 0x8000u) >> 16; }
__device__ __forceinline__ void drop_keep_gate(uint32_t seed, uint32_t row, uint32_t chan, uint32_t thresh16, bool& keep_t, bool& keep_s) {
  const uint32_t hsh = hash_u32(seed + row * 0x9E3779B1U + chan * 0x85EBCA6BU);
  keep_t = (hsh & 0xffffu) >= thresh16; keep_s = (hsh >> 16) >= thresh16;
}

// Workgroup barrier for LDS hazards only.  hipcc's __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: every global
// load AND store the wave has in flight is drained first — at one wave per SIMD that exposes a full HBM round trip per barrier
// (weights prefetched for the next phase, saved activations on their way out).  This one waits for the wave's LDS traffic only; the
// "memory" clobber keeps the compiler from moving LDS accesses across it.  Not for data handed over through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Rows layout geometry.  Uniform: utterance b owns rows [b*Tp, (b+1)*Tp).  Ragged (row0 != NULL, device int32
// [B+1]): utterance b owns rows [row0[b], row0[b+1]) = its frames + 2*HALO (the last one also owns the rows that
// round R up), so padded frames cost nothing; Tp is then only an upper bound on rows per utterance (grid sizing).
__device__ __forceinline__ int gt_row_base(const int32_t* row0, int b, int Tp) { return row0 ? row0[b] : b * Tp; }
__device__ __forceinline__ int gt_row_count(const int32_t* row0, int b, int Tp) { return row0 ? row0[b + 1] - row0[b] : Tp; }
__device__ __forceinline__ int gt_row_batch(const int32_t* row0, int B, int m, int Tp)
{
  if (!row0) return m / Tp;
  int lo = 0, hi = B - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (row0[mid] <= m) lo = mid; else hi = mid - 1; }
  return lo;
}

// Launch check shared by every entry point: reports WHICH call failed and why on stderr (the C-ABI
// itself only returns GT_E_LAUNCH).
#include <stdio.h>
static inline int gt_launch_status(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  fprintf(stderr, "[glowtts_hip] %s: HIP error %d (%s)\n", what, (int)e, hipGetErrorString(e));
  return -4;
}
