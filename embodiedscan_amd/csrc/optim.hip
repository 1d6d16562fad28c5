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

// one element of torch.optim.AdamW (decoupled decay, bias corrections folded into bc1 / bc2s); every AdamW kernel of this file goes
// through it, so the flat pass and the table-driven pass below produce the same bits
__device__ __forceinline__ float adamw_one(float p, float g, float& m, float& v, float clip, float lr, float b1, float b2, float eps,
                                           float wd, float bc1, float bc2s) {
  float gi = g * clip;
  float pi = p * (1.f - lr * wd);
  float mi = b1 * m + (1.f - b1) * gi;
  float vi = b2 * v + (1.f - b2) * gi * gi;
  m = mi;
  v = vi;
  float denom = sqrtf(vi) / bc2s + eps;
  return pi - (lr / bc1) * (mi / denom);
}
__device__ __forceinline__ float adamw_clip(float max_norm, const float* norm, float grad_scale) {
  float clip = 1.f;
  if (max_norm > 0.f) {
    float c = max_norm / (norm[0] + 1e-6f);            // torch.nn.utils.clip_grad_norm_
    clip = c < 1.f ? c : 1.f;
  }
  return clip * grad_scale;                             // 1 / world when g still holds the SUM over ranks
}

__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                               float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                               float max_norm, const float* __restrict__ norm, float grad_scale) {
  const float clip = adamw_clip(max_norm, norm, grad_scale);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float mi = m[i], vi = v[i];
    p[i] = adamw_one(p[i], g[i], mi, vi, clip, lr, b1, b2, eps, wd, bc1, bc2s);
    m[i] = mi;
    v[i] = vi;
  }
}

// AdamW over the arena AND the bf16 copies of the convolution kernels in one pass (round 5; VERDICT r4 item 1: the separate
// k_cast_weights_table pass re-read every updated weight: 8 B per parameter, 1.5 ms per occupancy step).  table rows (9 x int64):
//   {offset into the arena (elements), K | length, A, B, natural bf16 copy | 0, transposed bf16 copy, first work item,
//    double bits of lr_mult, double bits of decay_mult}   (lr = (float)(lr0 * lr_mult) in double: the value the host's
//    float(lr * lr_mult) hands es_adamw_step)
// a row with a natural-copy pointer is a convolution kernel [K][A][B]: one workgroup per 64 x 64 tile of one tap updates its
// elements (rows of 64 consecutive floats), writes the natural bf16 copy beside them and the transposed copy through LDS (as
// k_cast_weights_table does); a row without is a plain range of the arena (norm parameters, biases, padding): 4096 elements per
// workgroup.  The work item -> row search is the cast table's.
#define AW_CHUNK 4096
__global__ __launch_bounds__(256) void k_adamw_table(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ M,
                                                     float* __restrict__ V, const long long* __restrict__ table, int n_entries,
                                                     int total_items, double lr0, float b1, float b2, float eps, double wd0, float bc1,
                                                     float bc2s, float max_norm, const float* __restrict__ norm, float grad_scale) {
  __shared__ unsigned short tile[64][66];
  const int item = blockIdx.x;
  if (item >= total_items) return;
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {                                     // last row with first_item <= item
    int mid = (lo + hi + 1) >> 1;
    if (table[(size_t)mid * 9 + 6] <= item) lo = mid; else hi = mid - 1;
  }
  const long long* t = table + (size_t)lo * 9;
  const size_t off = (size_t)t[0];
  const double* td = (const double*)t;                 // (columns 7, 8 hold the bits of two doubles)
  const float lr = (float)(lr0 * td[7]), wd = (float)(wd0 * td[8]);
  const float clip = adamw_clip(max_norm, norm, grad_scale);
  const int local = item - (int)t[6];
  float* p = P + off;
  const float* g = G + off;
  float* m = M + off;
  float* v = V + off;
  if (t[4] == 0) {                                      // plain range
    const size_t len = (size_t)t[1], beg = (size_t)local * AW_CHUNK;
#pragma unroll 4
    for (int j = 0; j < AW_CHUNK / 256; ++j) {
      const size_t i = beg + (size_t)j * 256 + threadIdx.x;
      if (i < len) {
        float mi = m[i], vi = v[i];
        p[i] = adamw_one(p[i], g[i], mi, vi, clip, lr, b1, b2, eps, wd, bc1, bc2s);
        m[i] = mi;
        v[i] = vi;
      }
    }
    return;
  }
  unsigned short* nat = (unsigned short*)t[4];
  unsigned short* tr = (unsigned short*)t[5];
  const int A = (int)t[2], B = (int)t[3];
  const int ta = (A + 63) >> 6, tb = (B + 63) >> 6;
  const int k = local / (ta * tb), r = local % (ta * tb);
  const int a0 = (r / tb) * 64, b0 = (r % tb) * 64;
  const size_t base = (size_t)k * A * B;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {                    // rows a0 + i, columns b0 + tx (coalesced along b)
    const int a = a0 + i, b = b0 + tx;
    unsigned short h = 0;
    if (a < A && b < B) {
      const size_t e = base + (size_t)a * B + b;
      float mi = m[e], vi = v[e];
      const float pn = adamw_one(p[e], g[e], mi, vi, clip, lr, b1, b2, eps, wd, bc1, bc2s);
      p[e] = pn;
      m[e] = mi;
      v[e] = vi;
      h = (unsigned short)(es_pack_bf16(pn, 0.f) & 0xffff);
      nat[e] = h;
    }
    tile[i][tx] = h;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {                    // rows b0 + i of the transposed copy (coalesced along a)
    const int b = b0 + i, a = a0 + tx;
    if (a < A && b < B) tr[base + (size_t)b * A + a] = tile[tx][i];
  }
}
extern "C" int es_adamw_table(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const void* table_dev, int n_entries,
                              int total_items, double lr, float beta1, float beta2, float eps, double weight_decay, int step,
                              float max_norm, const float* grad_norm_dev, float grad_scale, void* stream) {
  if (n_entries <= 0 || total_items <= 0) return 0;
  float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(k_adamw_table, dim3(total_items), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (const long long*)table_dev, n_entries, total_items, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, max_norm,
                     grad_norm_dev, grad_scale);
  ES_CHECK_LAUNCH();
  return 0;
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
