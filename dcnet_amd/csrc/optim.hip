// Fused multi-tensor RMSprop step (torch.optim.RMSprop with momentum = 0, centered = False — the reference's
// optimiser, train_DCNet.py:528-534): one pass over parameter, gradient and running square average instead of the five
// element-wise passes of the foreach implementation.  HBM-bound: 12 B read + 8 B written per parameter.
#include "common.h"

namespace {

constexpr int RMS_CHUNK = 32;                 // tensors per launch (pointers travel as kernel arguments: no table upload)
struct RmsChunk {
  float* p[RMS_CHUNK]; const float* g[RMS_CHUNK]; float* v[RMS_CHUNK]; long long n[RMS_CHUNK];
};

// (no FMA contraction: the 16-byte path and the scalar path of a misaligned tensor — a view into a flat gradient buffer — must
//  round alike, or a data-parallel run with bound gradients drifts away from the single-GPU run by an ulp per step)
__device__ __forceinline__ void rms_update(float& p, const float g0, float& v, float lr, float alpha, float eps, float wd) {
#pragma clang fp contract(off)
  const float g = wd != 0.f ? g0 + wd * p : g0;          // grad = grad.add(param, alpha=weight_decay)
  v = v * alpha + (1.f - alpha) * g * g;                 // square_avg.mul_(alpha).addcmul_(grad, grad, value=1-alpha)
  p = p - lr * (g / (sqrtf(v) + eps));                   // param.addcdiv_(grad, square_avg.sqrt().add_(eps), value=-lr)
}

__global__ __launch_bounds__(256) void rmsprop_kernel(const RmsChunk c, float lr, const float* __restrict__ lr_dev, float alpha, float eps, float wd) {
  if (lr_dev) lr = lr_dev[0];                 // a captured step (hipGraph) reads its learning rate from device memory
  const int t = blockIdx.y;
  float* __restrict__ p = c.p[t]; const float* __restrict__ g = c.g[t]; float* __restrict__ v = c.v[t];
  const long long n = c.n[t];
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)v) & 15) == 0);
  const long long n4 = vec ? n / 4 : 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) { float a = pp[e], b = vv[e]; rms_update(a, gg[e], b, lr, alpha, eps, wd); pp[e] = a; vv[e] = b; }
    reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    rms_update(p[i], g[i], v[i], lr, alpha, eps, wd);
}

}  // namespace

extern "C" int dcn_rmsprop_step(float* const* params, const float* const* grads, float* const* square_avgs, const int64_t* numel,
                                int count, float lr, const float* lr_dev, float alpha, float eps, float weight_decay, void* stream) {
  DCN_CHECK_ARG(params && grads && square_avgs && numel && count > 0, "rmsprop_step: bad argument");
  for (int base = 0; base < count; base += RMS_CHUNK) {
    RmsChunk c{};
    const int m = count - base < RMS_CHUNK ? count - base : RMS_CHUNK;
    long long biggest = 0;
    for (int i = 0; i < m; ++i) {
      DCN_CHECK_ARG(params[base + i] && grads[base + i] && square_avgs[base + i] && numel[base + i] >= 0, "rmsprop_step: null tensor %d", base + i);
      c.p[i] = params[base + i]; c.g[i] = grads[base + i]; c.v[i] = square_avgs[base + i]; c.n[i] = numel[base + i];
      if (c.n[i] > biggest) biggest = c.n[i];
    }
    long long bx = (biggest / 4 + 255) / 256;
    bx = bx < 1 ? 1 : (bx > 128 ? 128 : bx);
    hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)bx, m), dim3(256), 0, (hipStream_t)stream, c, lr, lr_dev, alpha, eps, weight_decay);
    DCN_CHECK_LAUNCH("rmsprop_step");
  }
  return DCN_OK;
}
