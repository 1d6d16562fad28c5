// Inference post-processing of the FCAF3D 9-DoF head:
//   scores = sigmoid(cls) * sigmoid(centerness), 12-d -> 9-DoF box decode, per-class greedy NMS on the rotated
//   bird's-eye-view IoU.  Replaces FCAF3DHeadRotMat._predict_by_feat_single / _single_scene_multiclass_nms
//   (embodiedscan/models/dense_heads/fcaf3d_head.py:1352-1399,1666-1725) and mmcv.ops.nms3d (iou3d_nms3d_forward).
#include "common.h"
#include "../../include/es_hip.h"

__global__ void k_predict_scores(const float* __restrict__ ho, int ldh, int n, int C, float* __restrict__ scores,
                                 float* __restrict__ maxs) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* h = ho + (size_t)row * ldh;
  float sc = 1.f / (1.f + expf(-h[0]));
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) {
    float s = (1.f / (1.f + expf(-h[13 + c]))) * sc;
    scores[(size_t)row * C + c] = s;
    m = fmaxf(m, s);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) maxs[row] = m;
}
extern "C" int es_predict_scores(const float* ho, int ldh, int n, int C, float* scores, float* max_scores, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_predict_scores, dim3(es_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, ho, ldh, n, C, scores,
                     max_scores);
  ES_CHECK_LAUNCH();
  return 0;
}

// rows selected by idx: 12-d prediction + location -> (cx,cy,cz,dx,dy,dz,alpha,beta,gamma)   (fcaf3d_head.py:1454-1525)
__global__ void k_decode_boxes(const float* __restrict__ points, const float* __restrict__ bbox,
                               const int* __restrict__ idx, int m, float* __restrict__ out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m) return;
  int i = idx ? idx[t] : t;
  const float* b = bbox + (size_t)i * 12;
  const float* p = points + (size_t)i * 3;
  float xr[3] = {b[6], b[7], b[8]}, yr[3] = {b[9], b[10], b[11]};
  float ny = sqrtf(yr[0] * yr[0] + yr[1] * yr[1] + yr[2] * yr[2]) + 1e-8f;
  float y[3] = {yr[0] / ny, yr[1] / ny, yr[2] / ny};
  float z[3] = {xr[1] * y[2] - xr[2] * y[1], xr[2] * y[0] - xr[0] * y[2], xr[0] * y[1] - xr[1] * y[0]};
  float nz = sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]) + 1e-8f;
  z[0] /= nz; z[1] /= nz; z[2] /= nz;
  float x[3] = {y[1] * z[2] - y[2] * z[1], y[2] * z[0] - y[0] * z[2], y[0] * z[1] - y[1] * z[0]};
  float e0 = atan2f(-y[0], y[1]), e1 = asinf(y[2]), e2 = atan2f(-x[2], z[2]);
  float ca = cosf(e0), sa = sinf(e0), cb = cosf(e1), sb = sinf(e1), cc = cosf(e2), sc = sinf(e2);
  float R[9] = {ca * cc - sa * sb * sc, -sa * cb, ca * sc + sa * sb * cc, sa * cc + ca * sb * sc, ca * cb,
                sa * sc - ca * sb * cc, -cb * sc, sb, cb * cc};
  float s0 = (b[1] - b[0]) / 2, s1 = (b[3] - b[2]) / 2, s2 = (b[5] - b[4]) / 2;
  float* o = out + (size_t)t * 9;
  o[0] = p[0] + (s0 * R[0] + s1 * R[1] + s2 * R[2]);
  o[1] = p[1] + (s0 * R[3] + s1 * R[4] + s2 * R[5]);
  o[2] = p[2] + (s0 * R[6] + s1 * R[7] + s2 * R[8]);
  o[3] = b[0] + b[1]; o[4] = b[2] + b[3]; o[5] = b[4] + b[5];
  o[6] = e0; o[7] = e1; o[8] = e2;
}
extern "C" int es_decode_boxes(const float* points, const float* bbox, const int* idx, int m, float* out, void* stream) {
  if (m <= 0) return 0;
  hipLaunchKernelGGL(k_decode_boxes, dim3(es_cdiv(m, 128)), dim3(128), 0, (hipStream_t)stream, points, bbox, idx, m, out);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ rotated BEV IoU (float64, see oracle/predict.py)
struct P2 { double x, y; };
__device__ inline double crs(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }
__device__ inline void bev_corners(const float* b, P2* c) {
  double x = b[0], y = b[1], dx = b[3], dy = b[4], ang = b[6];
  double cs = cos(ang), sn = sin(ang);
  double x1 = x - dx / 2, y1 = y - dy / 2, x2 = x + dx / 2, y2 = y + dy / 2;
  double px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
  for (int k = 0; k < 4; ++k) {
    c[k].x = (px[k] - x) * cs + (py[k] - y) * (-sn) + x;
    c[k].y = (px[k] - x) * sn + (py[k] - y) * cs + y;
  }
  c[4] = c[0];
}
__device__ inline bool seg_isect(P2 p1, P2 p0, P2 q1, P2 q0, P2& out) {
  if (!(fmin(p0.x, p1.x) <= fmax(q0.x, q1.x) && fmin(q0.x, q1.x) <= fmax(p0.x, p1.x) &&
        fmin(p0.y, p1.y) <= fmax(q0.y, q1.y) && fmin(q0.y, q1.y) <= fmax(p0.y, p1.y)))
    return false;
  double s1 = crs(q0.x - p0.x, q0.y - p0.y, p1.x - p0.x, p1.y - p0.y);
  double s2 = crs(p1.x - p0.x, p1.y - p0.y, q1.x - p0.x, q1.y - p0.y);
  double s3 = crs(p0.x - q0.x, p0.y - q0.y, q1.x - q0.x, q1.y - q0.y);
  double s4 = crs(q1.x - q0.x, q1.y - q0.y, p1.x - q0.x, p1.y - q0.y);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  double s5 = crs(q1.x - p0.x, q1.y - p0.y, p1.x - p0.x, p1.y - p0.y);
  if (fabs(s5 - s1) > 1e-8) {
    out.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    out.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    double a0 = p0.y - p1.y, a1 = q0.y - q1.y, b0 = p1.x - p0.x, b1 = q1.x - q0.x;
    double c0 = p0.x * p1.y - p1.x * p0.y, c1 = q0.x * q1.y - q1.x * q0.y;
    double D = a0 * b1 - a1 * b0;
    out.x = (b0 * c1 - b1 * c0) / D;
    out.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}
__device__ inline bool in_box(const float* b, P2 p) {
  double cx = b[0], cy = b[1], dx = b[3], dy = b[4], ang = b[6];
  double cs = cos(-ang), sn = sin(-ang);
  double rx = (p.x - cx) * cs + (p.y - cy) * (-sn), ry = (p.x - cx) * sn + (p.y - cy) * cs;
  return fabs(rx) < dx / 2 + 1e-2 && fabs(ry) < dy / 2 + 1e-2;
}
__device__ double iou_bev(const float* a, const float* b) {
  P2 ca[5], cb[5], pts[16];
  bev_corners(a, ca);
  bev_corners(b, cb);
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 o;
      if (seg_isect(ca[i + 1], ca[i], cb[j + 1], cb[j], o)) pts[cnt++] = o;
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box(a, cb[k])) pts[cnt++] = cb[k];
    if (in_box(b, ca[k])) pts[cnt++] = ca[k];
  }
  double so = 0.0;
  if (cnt >= 3) {
    double mx = 0, my = 0;
    for (int k = 0; k < cnt; ++k) { mx += pts[k].x; my += pts[k].y; }
    mx /= cnt; my /= cnt;
    double ang[16];
    for (int k = 0; k < cnt; ++k) ang[k] = atan2(pts[k].y - my, pts[k].x - mx);
    for (int i = 1; i < cnt; ++i) {                    // insertion sort (stable, like the oracle's list.sort)
      P2 p = pts[i]; double g = ang[i];
      int j = i - 1;
      while (j >= 0 && ang[j] > g) { pts[j + 1] = pts[j]; ang[j + 1] = ang[j]; --j; }
      pts[j + 1] = p; ang[j + 1] = g;
    }
    double area = 0;
    for (int k = 0; k < cnt - 1; ++k)
      area += crs(pts[k].x - pts[0].x, pts[k].y - pts[0].y, pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y);
    so = fabs(area) / 2.0;
  }
  double sa = (double)a[3] * (double)a[4], sb = (double)b[3] * (double)b[4];
  return so / fmax(sa + sb - so, 1e-8);
}

#define NMS_CAP 4096
// one workgroup per class: candidates (score > thr) -> bitonic sort by (score desc, index asc) -> greedy suppression.
// keep_idx[c*M + k] = k-th kept box (descending score), keep_cnt[c] = number kept.
__global__ __launch_bounds__(256) void k_nms3d_multiclass(const float* __restrict__ boxes /* (M,9) */,
                                                          const float* __restrict__ scores /* (M,C) */, int M, int C,
                                                          float score_thr, float iou_thr, int* __restrict__ keep_idx,
                                                          int* __restrict__ keep_cnt) {
  __shared__ float s_key[NMS_CAP];
  __shared__ int s_idx[NMS_CAP];
  __shared__ unsigned char s_sup[NMS_CAP];
  __shared__ int s_n, s_keep;
  const int c = blockIdx.x, t = threadIdx.x;
  if (t == 0) { s_n = 0; s_keep = 0; }
  __syncthreads();
  for (int i = t; i < M; i += 256) {
    float s = scores[(size_t)i * C + c];
    if (s > score_thr) {
      int p = atomicAdd(&s_n, 1);
      if (p < NMS_CAP) { s_key[p] = s; s_idx[p] = i; }
    }
  }
  __syncthreads();
  int n = min(s_n, NMS_CAP);
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = n + t; i < np2; i += 256) { s_key[i] = -INFINITY; s_idx[i] = 0x7fffffff; }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < np2; i += 256) {
        int l = i ^ j;
        if (l > i) {
          bool up = ((i & k) == 0);
          float ki = s_key[i], kl = s_key[l];
          int ii = s_idx[i], il = s_idx[l];
          bool i_first = (ki > kl) || (ki == kl && ii < il);     // i should come before l in the final order
          if (up ? !i_first : i_first) { s_key[i] = kl; s_key[l] = ki; s_idx[i] = il; s_idx[l] = ii; }
        }
      }
      __syncthreads();
    }
  for (int i = t; i < n; i += 256) s_sup[i] = 0;
  __syncthreads();
  for (int p = 0; p < n; ++p) {
    if (s_sup[p]) continue;                                      // uniform: LDS value read by all threads
    if (t == 0) { keep_idx[(size_t)c * M + s_keep] = s_idx[p]; s_keep++; }
    const float* bp = boxes + (size_t)s_idx[p] * 9;
    for (int q = p + 1 + t; q < n; q += 256)
      if (!s_sup[q] && iou_bev(bp, boxes + (size_t)s_idx[q] * 9) > (double)iou_thr) s_sup[q] = 1;
    __syncthreads();
  }
  if (t == 0) keep_cnt[c] = s_keep;
}
extern "C" int es_nms3d_multiclass(const float* boxes, const float* scores, int M, int C, float score_thr, float iou_thr,
                                   int* keep_idx, int* keep_cnt, void* stream) {
  if (C <= 0) return 0;
  if (M > NMS_CAP) return -10;
  hipLaunchKernelGGL(k_nms3d_multiclass, dim3(C), dim3(256), 0, (hipStream_t)stream, boxes, scores, M, C, score_thr,
                     iou_thr, keep_idx, keep_cnt);
  ES_CHECK_LAUNCH();
  return 0;
}
