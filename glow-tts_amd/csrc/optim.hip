// AdamW over ONE flat fp32 parameter buffer, with the total gradient norm as a by-product.
//
// The reference step (train_ms_emo_lang_pitch.py:309-314) runs torch.optim.AdamW over ~330 separate
// tensors (multi-tensor-apply chunks, ~1.2 TB/s effective) after commons.clip_grad_value_(params, None)
// (commons.py:320-336: ~330 .norm().item() host syncs, no clipping).  Here parameters, gradients and both
// moments are slices of four flat buffers, so one grid-stride pass reads g, p, m, v once and writes p, m, v
// once (7 x 4 B per parameter, HBM-bound), and the sum of squared gradients rides along (one atomic per
// workgroup).  Hyper-parameters live in device memory so a captured HIP graph follows the LR / momentum
// schedule (OneCycleLR cycles both) without re-capture:
//   hyper = {lr, beta1, beta2, eps, weight_decay, step}      (step already incremented for this update)
// Update rule = torch.optim.AdamW (decoupled decay, bias-corrected):
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include "common.h"
#include "../../include/glowtts_hip.h"

namespace {

__global__ __launch_bounds__(256) void gt_adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, size_t n4,
                                                            const float* __restrict__ hyper, float* __restrict__ gnorm_sq)
{
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], t = hyper[5];
  const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
  const float step_size = lr / bc1, decay = 1.0f - lr * wd, ob1 = 1.0f - b1, ob2 = 1.0f - b2;
  float ss = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 g4 = reinterpret_cast<const float4*>(g)[i];
    float4 p4 = reinterpret_cast<float4*>(p)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
    float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ss += gg[k] * gg[k];
      mm[k] = b1 * mm[k] + ob1 * gg[k];
      vv[k] = b2 * vv[k] + ob2 * gg[k] * gg[k];
      pp[k] = pp[k] * decay - step_size * mm[k] / (sqrtf(vv[k]) / bc2s + eps);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
  }
  if (gnorm_sq) {
    __shared__ float red[4];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(gnorm_sq, red[0] + red[1] + red[2] + red[3]);
  }
}

}  // namespace

extern "C" int gt_adamw_flat(float* p, const float* g, float* m, float* v, size_t n, const float* hyper,
                             float* gnorm_sq, void* stream)
{
  if (!p || !g || !m || !v || !hyper) return GT_E_INVAL;
  if (n == 0) return GT_OK;
  if ((n & 3) || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) return GT_E_ALIGN;
  const size_t n4 = n >> 2;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 256 * 4) blocks = 256 * 4;            // 4 workgroups per CU, grid-stride beyond (one same-address atomic each)
  hipLaunchKernelGGL(gt_adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     p, g, m, v, n4, hyper, gnorm_sq);
  return gt_launch_status(__func__);
}
