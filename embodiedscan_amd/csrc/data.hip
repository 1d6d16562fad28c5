// Data-side kernels pulled on-device (rows A1-A3, A18):
//   * es_depth_to_points : depth pixel -> camera point (ConvertRGBDToPoints / points_img2cam,
//     embodiedscan/datasets/transforms/points.py:30-81, structures/bbox_3d/utils.py:335-368)
//     -> global frame (AggregateMultiViewPoints, datasets/transforms/multiview.py:139-169)
//     -> RandomFlip3D / GlobalRotScaleTrans (datasets/transforms/augmentation.py:87-139,322-348)
//     for the (view, pixel) pairs chosen by PointSample (points.py:155-213); un-projecting only the
//     sampled pixels is identical to un-projecting all 307 200 and indexing.
//   * es_preprocess_img  : uint8 BGR NCHW -> f32 RGB normalised, channels-last
//     (Det3DDataPreprocessor.preprocess_img, models/data_preprocessors/data_preprocessor.py:249-264)
#include "common.h"
#include "../../include/es_hip.h"

// mats per view: [0..15] inv(pad4(K))  [16..31] inv(global2cam) ; aug: [0..8] rot_mat_T, [9] scale, [10..12] trans,
// [13] hflip, [14] vflip
__global__ void k_depth_to_points(const float* __restrict__ depth, int H, int W, const int* __restrict__ sel_view,
                                  const int* __restrict__ sel_pix, int n, const float* __restrict__ mats,
                                  const float* __restrict__ aug, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int v = sel_view[i], p = sel_pix[i];
  float d = depth[((size_t)v * H) * W + p];
  float u = (float)(p % W), w_ = (float)(p / W);
  const float* Ki = mats + v * 32;
  const float* Ei = Ki + 16;
  float h0 = u * d, h1 = w_ * d, h2 = d, h3 = 1.f;
  // homo @ inv(K)^T  -> camera point (first three components)
  float c[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) c[r] = fmaf(h3, Ki[r * 4 + 3], fmaf(h2, Ki[r * 4 + 2], fmaf(h1, Ki[r * 4 + 1], h0 * Ki[r * 4])));
  float g[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) g[r] = fmaf(1.f, Ei[r * 4 + 3], fmaf(c[2], Ei[r * 4 + 2], fmaf(c[1], Ei[r * 4 + 1], c[0] * Ei[r * 4])));
  if (aug[13] != 0.f) g[0] = -g[0];
  if (aug[14] != 0.f) g[1] = -g[1];
  float q[3];
#pragma unroll
  for (int cidx = 0; cidx < 3; ++cidx) q[cidx] = fmaf(g[2], aug[6 + cidx], fmaf(g[1], aug[3 + cidx], g[0] * aug[cidx]));
#pragma unroll
  for (int cidx = 0; cidx < 3; ++cidx) out[(size_t)i * 3 + cidx] = q[cidx] * aug[9] + aug[10 + cidx];
}
extern "C" int es_depth_to_points(const float* depth, int H, int W, const int* sel_view, const int* sel_pix, int n,
                                  const float* mats, const float* aug, float* out, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_depth_to_points, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, depth, H, W,
                     sel_view, sel_pix, n, mats, aug, out);
  ES_CHECK_LAUNCH();
  return 0;
}

// in: (NI, 3, H, W) u8 BGR ; out: (NI, H, W, 3) f32 RGB normalised (channels-last feeds the conv engine)
__global__ void k_preprocess_img(const unsigned char* __restrict__ in, int NI, int HW, float m0, float m1, float m2,
                                 float s0, float s1, float s2, float* __restrict__ out) {
  size_t tot = (size_t)NI * HW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    size_t im = e / HW, px = e - im * HW;
    const unsigned char* b = in + im * 3 * HW + px;
    float r0 = ((float)b[2 * (size_t)HW] - m0) / s0;     // channel flip [2,1,0]
    float r1 = ((float)b[(size_t)HW] - m1) / s1;
    float r2 = ((float)b[0] - m2) / s2;
    float* o = out + e * 3;
    o[0] = r0; o[1] = r1; o[2] = r2;
  }
}
extern "C" int es_preprocess_img(const unsigned char* img, int n_img, int H, int W, const float* mean,
                                 const float* std, float* out, void* stream) {
  if (n_img <= 0) return 0;
  hipLaunchKernelGGL(k_preprocess_img, dim3(4096), dim3(256), 0, (hipStream_t)stream, img, n_img, H * W, mean[0],
                     mean[1], mean[2], std[0], std[1], std[2], out);
  ES_CHECK_LAUNCH();
  return 0;
}
