// Optimiser step over the flat parameter arena: one grad-norm reduction, one fused
// clip + AdamW pass (torch.optim.AdamW semantics; mmengine OptimWrapper with
// clip_grad=dict(max_norm=10, norm_type=2), configs/detection/mv-det3d_...py:219-223).
// HBM-bound: 4 reads + 3 writes of 4 B per parameter.
#include "common.h"
#include "../../include/es_hip.h"

#define OPT_BLOCKS 2048
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0;
  size_t n4 = n / 4;
  const float4* g4 = (const float4*)g;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = g4[i];
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[n4 * 4 + threadIdx.x]; s += (double)v * v; }
  s = es_wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_norm_final(const double* __restrict__ partial, int nb, float* __restrict__ norm_out) {
  __shared__ double red[4];
  double s = 0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += partial[i];
  s = es_wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm_out[0] = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
}
// data parallel: the sum of squares of ONE reduced bucket (queued right behind its all-reduce), OPT_BLOCKS doubles per bucket
extern "C" int es_sumsq_partial(const float* grad, size_t n, double* partial, void* stream) {
  hipLaunchKernelGGL(k_sumsq, dim3(OPT_BLOCKS), dim3(256), 0, (hipStream_t)stream, grad, n, partial);
  ES_CHECK_LAUNCH();
  return 0;
}
__global__ void k_norm_final_scaled(const double* __restrict__ partial, int nb, float scale, float* __restrict__ norm_out) {
  __shared__ double red[4];
  double s = 0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += partial[i];
  s = es_wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm_out[0] = (float)(sqrt(red[0] + red[1] + red[2] + red[3]) * (double)scale);
}
// norm_out[0] = scale * sqrt(sum of n_partials doubles): the clip norm of the MEAN gradient from the per-bucket sums of the
// SUMMED gradient (scale = 1 / world)
extern "C" int es_norm_from_partials(const double* partial, int n_partials, float scale, float* norm_out, void* stream) {
  hipLaunchKernelGGL(k_norm_final_scaled, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_partials, scale, norm_out);
  ES_CHECK_LAUNCH();
  return 0;
}
// norm_out[0] = ||g||_2 (device scalar).  partial: OPT_BLOCKS doubles.
extern "C" int es_grad_norm(const float* grad, size_t n, double* partial, float* norm_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_sumsq, dim3(OPT_BLOCKS), dim3(256), 0, st, grad, n, partial);
  hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, st, partial, OPT_BLOCKS, norm_out);
  ES_CHECK_LAUNCH();
  return 0;
}

__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                               float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                               float max_norm, const float* __restrict__ norm, float grad_scale) {
  float clip = 1.f;
  if (max_norm > 0.f) {
    float c = max_norm / (norm[0] + 1e-6f);            // torch.nn.utils.clip_grad_norm_
    clip = c < 1.f ? c : 1.f;
  }
  clip *= grad_scale;                                   // 1 / world when g still holds the SUM over ranks
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * clip;
    float pi = p[i] * (1.f - lr * wd);
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2s + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}
extern "C" int es_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float max_norm,
                             const float* grad_norm_dev, float grad_scale, void* stream) {
  if (n == 0) return 0;
  float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(k_adamw, dim3(4096), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2s, max_norm, grad_norm_dev, grad_scale);
  ES_CHECK_LAUNCH();
  return 0;
}
