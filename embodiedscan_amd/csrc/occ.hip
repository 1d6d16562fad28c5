// Occupancy supervision and losses:
//   * es_occ_targets : occ_multiscale_supervision (embodiedscan/models/losses/occ_loss.py:7-36) + the MaxPool3d
//     down-sampling of the visibility mask (dense_heads/imvoxel_occ_head.py:163-171);
//   * es_occ_loss_stats / es_occ_loss_coeffs / es_occ_loss_grad : CrossEntropyLoss(ignore_index=255) + sem_scal_loss +
//     geo_scal_loss (occ_loss.py:39-141, imvoxel_occ_head.py:156-181) for one level in three launches instead of the
//     reference's 81-iteration Python loop.  Every class statistic the three losses need is linear in the softmax
//     probabilities:  A_c = sum_mask p_c,  B_c = sum_mask p_c [t == c],  N_c = #(t == c),  n = #mask, so one pass
//     accumulates them (f64), a one-block kernel turns them into the loss values and into the coefficients
//     dL/dA_c, dL/dB_c, and a second pass writes dL/dlogits = softmax-backward of (alpha_c + beta_c [t == c]) plus the
//     cross-entropy term.
#include "common.h"
#include "../../include/es_hip.h"

#define OCC_MAXC 256
#define OCC_PER (OCC_MAXC / 64)

// ------------------------------------------------------------------ targets
// pass 1: the LAST occurrence of a voxel in gt_occ wins (sequential index_put semantics of the CPU reference)
__global__ void k_occ_winner(const int* __restrict__ occ, int n, int ratio, int X, int Y, int Z, int* __restrict__ winner) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x = occ[(size_t)i * 4] / ratio, y = occ[(size_t)i * 4 + 1] / ratio, z = occ[(size_t)i * 4 + 2] / ratio;
  if (x < 0 || x >= X || y < 0 || y >= Y || z < 0 || z >= Z) return;
  atomicMax(&winner[((size_t)x * Y + y) * Z + z], i);
}
// pass 2: label of the winner (0 = empty); voxels whose ratio^3 window of the visibility mask holds no visible voxel
// become 255 (ignore)
__global__ void k_occ_fill(const int* __restrict__ occ, const int* __restrict__ winner, const unsigned char* __restrict__ mask,
                           int ratio, int X, int Y, int Z, int* __restrict__ gt) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= X * Y * Z) return;
  int w = winner[v];
  int label = w >= 0 ? occ[(size_t)w * 4 + 3] : 0;
  if (mask != nullptr) {
    int z = v % Z, y = (v / Z) % Y, x = v / (Z * Y);
    int Y0 = Y * ratio, Z0 = Z * ratio;
    bool vis = false;
    for (int a = 0; a < ratio && !vis; ++a)
      for (int b = 0; b < ratio && !vis; ++b)
        for (int c = 0; c < ratio; ++c)
          if (mask[((size_t)(x * ratio + a) * Y0 + (y * ratio + b)) * Z0 + z * ratio + c]) { vis = true; break; }
    if (!vis) label = 255;
  }
  gt[v] = label;
}
extern "C" int es_occ_targets(const int* gt_occ, int n, int ratio, int X, int Y, int Z, const unsigned char* mask,
                              int* winner_scratch, int* gt, void* stream) {
  if (ratio <= 0 || X <= 0) return -2;
  hipStream_t st = (hipStream_t)stream;
  ES_TRY(hipMemsetAsync(winner_scratch, 0xFF, sizeof(int) * (size_t)X * Y * Z, st));
  if (n > 0) hipLaunchKernelGGL(k_occ_winner, dim3(es_cdiv(n, 256)), dim3(256), 0, st, gt_occ, n, ratio, X, Y, Z, winner_scratch);
  hipLaunchKernelGGL(k_occ_fill, dim3(es_cdiv((long long)X * Y * Z, 256)), dim3(256), 0, st, gt_occ, winner_scratch, mask, ratio,
                     X, Y, Z, gt);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ losses
// stats layout (doubles): [0..C) A, [C..2C) B, [2C..3C) N, [3C] n_mask, [3C+1] CE sum
__device__ inline void row_softmax(const float* __restrict__ row, int C, int lane, float (&p)[OCC_PER], float& lse) {
  float m = -INFINITY;
#pragma unroll
  for (int q = 0; q < OCC_PER; ++q) {
    int c = lane + q * 64;
    p[q] = c < C ? row[c] : -INFINITY;
    m = fmaxf(m, p[q]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < OCC_PER; ++q) {
    int c = lane + q * 64;
    p[q] = c < C ? __expf(p[q] - m) : 0.f;
    s += p[q];
  }
  s = es_wave_sum(s);
  float inv = 1.f / s;
#pragma unroll
  for (int q = 0; q < OCC_PER; ++q) p[q] *= inv;
  lse = m + __logf(s);
}

__global__ __launch_bounds__(256) void k_occ_stats(const float* __restrict__ logits, int ld, const int* __restrict__ gt, int n,
                                                   int C, double* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
  double A[OCC_PER], Bc[OCC_PER], N[OCC_PER];
#pragma unroll
  for (int q = 0; q < OCC_PER; ++q) A[q] = Bc[q] = N[q] = 0.0;
  double ce = 0.0, nm = 0.0;
  for (int i = wave; i < n; i += nwave) {
    int t = gt[i];
    if (t == 255) continue;                                  // ignore_index / unknown voxel
    float p[OCC_PER], lse;
    row_softmax(logits + (size_t)i * ld, C, lane, p, lse);
#pragma unroll
    for (int q = 0; q < OCC_PER; ++q) {
      int c = lane + q * 64;
      A[q] += (double)p[q];
      if (c == t) { Bc[q] += (double)p[q]; N[q] += 1.0; }
    }
    if (lane == 0) {
      nm += 1.0;
      if (t >= 0 && t < C) ce += (double)(lse - logits[(size_t)i * ld + t]);
    }
  }
#pragma unroll
  for (int q = 0; q < OCC_PER; ++q) {
    int c = lane + q * 64;
    if (c < C) {
      if (A[q] != 0.0) unsafeAtomicAdd(&stats[c], A[q]);
      if (Bc[q] != 0.0) unsafeAtomicAdd(&stats[C + c], Bc[q]);
      if (N[q] != 0.0) unsafeAtomicAdd(&stats[2 * C + c], N[q]);
    }
  }
  if (lane == 0 && nm != 0.0) {
    unsafeAtomicAdd(&stats[3 * C], nm);
    unsafeAtomicAdd(&stats[3 * C + 1], ce);
  }
}

// F.binary_cross_entropy(x, ones): value -max(log x, -100); gradient (x - 1) / max((1 - x) x, 1e-12)  (ATen)
__device__ inline double bce1(double x) { return -fmax(log(x), -100.0); }
__device__ inline double bce1_grad(double x) { return (x - 1.0) / fmax((1.0 - x) * x, 1e-12); }

// one block: losses (out[0] CE, out[1] sem_scal, out[2] geo_scal, out[3] their weighted sum) and coeff[c] = alpha_c,
// coeff[C + c] = beta_c (already multiplied by `weight`), coeff[2C] = weight / n_mask (cross-entropy scale)
__global__ void k_occ_coeffs(const double* __restrict__ stats, int C, float weight, float* __restrict__ coeff,
                             float* __restrict__ out, float* __restrict__ total_acc) {
  __shared__ double s_loss[256], s_cnt[256];
  const int c = threadIdx.x;
  const double n = stats[3 * C];
  double alpha = 0.0, beta = 0.0, lc = 0.0, counted = 0.0;
  if (c < C) {
    double A = stats[c], B = stats[C + c], N = stats[2 * C + c];
    if (N > 0.0) {                                           // occ_loss.py:113: classes present in the target only
      counted = 1.0;
      if (A > 0.0) {
        double pr = B / A;
        lc += bce1(pr);
        double g = bce1_grad(pr);
        beta += g / A;
        alpha += -g * B / (A * A);
      }
      double rc = B / N;
      lc += bce1(rc);
      beta += bce1_grad(rc) / N;
      double rest = n - N;
      if (rest > 0.0) {
        double sp = (rest - (A - B)) / rest;
        lc += bce1(sp);
        double g = bce1_grad(sp);
        alpha += -g / rest;
        beta += g / rest;
      }
    }
  }
  s_loss[threadIdx.x] = lc;
  s_cnt[threadIdx.x] = counted;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { s_loss[threadIdx.x] += s_loss[threadIdx.x + o]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o]; }
    __syncthreads();
  }
  const double count = s_cnt[0];
  const double sem = count > 0.0 ? s_loss[0] / count : 0.0;
  if (c < C) {
    double a = count > 0.0 ? alpha / count : 0.0, b = count > 0.0 ? beta / count : 0.0;
    if (c == 0) {                                            // geo_scal_loss: empty (class 0) vs non-empty, eps 1e-6
      const double eps = 1e-6;
      double A0 = stats[0], B0 = stats[C], N0 = stats[2 * C];
      double inter = (n - N0) - (A0 - B0);
      double D = n - A0 + eps, R = n - N0 + eps, S = N0 + eps;
      double P = inter / D, Rc = inter / R, Sp = B0 / S;
      double gP = bce1_grad(P), gR = bce1_grad(Rc), gS = bce1_grad(Sp);
      a += gP * (inter - D) / (D * D) - gR / R;
      b += gP / D + gR / R + gS / S;
      double geo = bce1(P) + bce1(Rc) + bce1(Sp);
      double ce = stats[3 * C + 1] / n;                      // n == 0 -> NaN, like CrossEntropyLoss(reduction='mean')
      out[0] = (float)ce;
      out[1] = (float)sem;
      out[2] = (float)geo;
      float tot = (float)((ce + sem + geo) * (double)weight);
      out[3] = tot;
      if (total_acc != nullptr) total_acc[0] += tot;
      coeff[2 * C] = (float)((double)weight / n);
    }
    coeff[c] = (float)(a * (double)weight);
    coeff[C + c] = (float)(b * (double)weight);
  }
}

__global__ __launch_bounds__(256) void k_occ_grad(const float* __restrict__ logits, int ld, const int* __restrict__ gt, int n,
                                                  int C, const float* __restrict__ coeff, float* __restrict__ dlogits, int ldg) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
  float al[OCC_PER], be[OCC_PER];
#pragma unroll
  for (int q = 0; q < OCC_PER; ++q) {
    int c = lane + q * 64;
    al[q] = c < C ? coeff[c] : 0.f;
    be[q] = c < C ? coeff[C + c] : 0.f;
  }
  const float ce_scale = coeff[2 * C];
  for (int i = wave; i < n; i += nwave) {
    int t = gt[i];
    float* d = dlogits + (size_t)i * ldg;
    if (t == 255) {
#pragma unroll
      for (int q = 0; q < OCC_PER; ++q) {
        int c = lane + q * 64;
        if (c < C) d[c] = 0.f;
      }
      continue;
    }
    float p[OCC_PER], lse;
    row_softmax(logits + (size_t)i * ld, C, lane, p, lse);
    float g[OCC_PER], dot = 0.f;
#pragma unroll
    for (int q = 0; q < OCC_PER; ++q) {
      int c = lane + q * 64;
      g[q] = al[q] + (c == t ? be[q] : 0.f);
      dot += p[q] * g[q];
    }
    dot = es_wave_sum(dot);
#pragma unroll
    for (int q = 0; q < OCC_PER; ++q) {
      int c = lane + q * 64;
      if (c < C) d[c] = p[q] * (g[q] - dot) + ce_scale * (p[q] - (c == t ? 1.f : 0.f));
    }
  }
}

extern "C" int es_occ_loss(const float* logits, int ld, const int* gt, int n, int C, float weight, double* stats,
                           float* coeff, float* dlogits, int ldg, float* loss_out, float* total_acc, void* stream) {
  if (C > OCC_MAXC || C < 1) return -4;
  if (n <= 0) return -2;
  hipStream_t st = (hipStream_t)stream;
  ES_TRY(hipMemsetAsync(stats, 0, sizeof(double) * (size_t)(3 * C + 2), st));
  int blocks = min(es_cdiv(n, 4), 1024);
  hipLaunchKernelGGL(k_occ_stats, dim3(blocks), dim3(256), 0, st, logits, ld, gt, n, C, stats);
  hipLaunchKernelGGL(k_occ_coeffs, dim3(1), dim3(256), 0, st, stats, C, weight, coeff, loss_out, total_acc);
  if (dlogits != nullptr)
    hipLaunchKernelGGL(k_occ_grad, dim3(blocks), dim3(256), 0, st, logits, ld, gt, n, C, coeff, dlogits, ldg);
  ES_CHECK_LAUNCH();
  return 0;
}
