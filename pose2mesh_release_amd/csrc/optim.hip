// Fused Adam / RMSprop over one flat fp32 parameter buffer (all 76 M parameters in a single launch).
// Reference: torch.optim.Adam(model.parameters(), lr) built at lib/funcs_utils.py:92-96 and stepped at
// lib/core/base.py:148 (defaults betas=(0.9,0.999), eps=1e-8, weight_decay=0, amsgrad=False).
// Pure streaming: 16 B read (p,g,m,v) + 12 B written per parameter -> HBM-bound.
#include "p2m_common.h"

namespace p2m {

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, long n4, long n, float lr, float b1, float b2,
                                               float eps, float bc1, float bc2_sqrt, float grad_scale,
                                               const float* __restrict__ hp) {
  if (hp != nullptr) {          // step-dependent scalars from device memory: the launch can sit in a captured hipGraph
    lr = hp[0];
    bc1 = hp[1];
    bc2_sqrt = hp[2];
    grad_scale = hp[3];
  }
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    float4 P = reinterpret_cast<float4*>(p)[i];
    float4 G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i];
    float4 V = reinterpret_cast<float4*>(v)[i];
    float* pp = reinterpret_cast<float*>(&P);
    float* gg = reinterpret_cast<float*>(&G);
    float* mm = reinterpret_cast<float*>(&M);
    float* vv = reinterpret_cast<float*>(&V);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float gr = gg[k] * grad_scale;
      mm[k] = b1 * mm[k] + (1.f - b1) * gr;
      vv[k] = b2 * vv[k] + (1.f - b2) * gr * gr;
      float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
      pp[k] -= (lr / bc1) * (mm[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
  } else if (i == n4) {
    for (long j = n4 * 4; j < n; j++) {  // tail (< 4 elements)
      float gr = g[j] * grad_scale;
      m[j] = b1 * m[j] + (1.f - b1) * gr;
      v[j] = b2 * v[j] + (1.f - b2) * gr * gr;
      p[j] -= (lr / bc1) * (m[j] / (sqrtf(v[j]) / bc2_sqrt + eps));
    }
  }
}

// torch.optim.RMSprop(params, lr) as the reference's yaml recipes build it (lib/funcs_utils.py:87-91: defaults
// alpha = 0.99, eps = 1e-8, weight_decay = 0, momentum = 0, centered = False):
//   v = alpha v + (1 - alpha) g^2 ;  p -= lr * g / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void k_rmsprop(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ v, long n4, long n, float lr, float alpha,
                                                  float eps, float grad_scale, const float* __restrict__ hp) {
  if (hp != nullptr) {
    lr = hp[0];
    grad_scale = hp[3];
  }
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    float4 P = reinterpret_cast<float4*>(p)[i];
    float4 G = reinterpret_cast<const float4*>(g)[i];
    float4 V = reinterpret_cast<float4*>(v)[i];
    float* pp = reinterpret_cast<float*>(&P);
    float* gg = reinterpret_cast<float*>(&G);
    float* vv = reinterpret_cast<float*>(&V);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float gr = gg[k] * grad_scale;
      vv[k] = alpha * vv[k] + (1.f - alpha) * gr * gr;
      pp[k] -= lr * (gr / (sqrtf(vv[k]) + eps));
    }
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(v)[i] = V;
  } else if (i == n4) {
    for (long j = n4 * 4; j < n; j++) {
      const float gr = g[j] * grad_scale;
      v[j] = alpha * v[j] + (1.f - alpha) * gr * gr;
      p[j] -= lr * (gr / (sqrtf(v[j]) + eps));
    }
  }
}

}  // namespace p2m

using namespace p2m;

extern "C" int p2m_rmsprop_step(float* param, const float* grad, float* square_avg, int64_t n, float lr, float alpha,
                                float eps, float grad_scale, void* stream) {
  P2M_CHECK_ARG(param && grad && square_avg && n > 0, "null pointer or empty buffer");
  P2M_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)square_avg) % 16 == 0, "buffers must be 16-byte aligned");
  const long n4 = n / 4;
  hipLaunchKernelGGL(k_rmsprop, dim3(cdiv(n4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, square_avg,
                     n4, (long)n, lr, alpha, eps, grad_scale, (const float*)nullptr);
  return check_launch("rmsprop_step");
}

// The same steps with the step-dependent scalars read from DEVICE memory, hp = {lr, 1 - beta1^t, sqrt(1 - beta2^t),
// grad_scale} (RMSprop uses hp[0] and hp[3]): the launch itself is then the same every step and can be part of a
// captured hipGraph; the host refreshes the four floats before each replay.
extern "C" int p2m_rmsprop_step_dev(float* param, const float* grad, float* square_avg, int64_t n, const float* hp,
                                    float alpha, float eps, void* stream) {
  P2M_CHECK_ARG(param && grad && square_avg && hp && n > 0, "null pointer or empty buffer");
  P2M_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)square_avg) % 16 == 0, "buffers must be 16-byte aligned");
  const long n4 = n / 4;
  hipLaunchKernelGGL(k_rmsprop, dim3(cdiv(n4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, square_avg,
                     n4, (long)n, 0.f, alpha, eps, 1.f, hp);
  return check_launch("rmsprop_step_dev");
}

extern "C" int p2m_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 const float* hp, float beta1, float beta2, float eps, void* stream) {
  P2M_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && hp && n > 0, "null pointer or empty buffer");
  P2M_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                "buffers must be 16-byte aligned");
  const long n4 = n / 4;
  hipLaunchKernelGGL(k_adam, dim3(cdiv(n4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n4, (long)n, 0.f, beta1, beta2, eps, 1.f, 1.f, 1.f, hp);
  return check_launch("adam_step_dev");
}

extern "C" int p2m_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             int64_t step, float lr, float beta1, float beta2, float eps, float grad_scale,
                             void* stream) {
  P2M_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0, "null pointer, empty buffer or step < 1");
  P2M_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                "buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const long n4 = n / 4;
  hipLaunchKernelGGL(k_adam, dim3(cdiv(n4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n4, (long)n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale,
                     (const float*)nullptr);
  return check_launch("adam_step");
}
