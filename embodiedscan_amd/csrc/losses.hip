// Loss kernels of the FCAF3D 9-DoF head, forward value AND gradient in one pass
// (the train step needs both; the gradients w.r.t. the head outputs are emitted eagerly).
//   * sigmoid focal loss over (N, 284) logits with label -1 == background (row A14),
//     mmcv sigmoid_focal_loss semantics as used by mmdet.FocalLoss at
//     embodiedscan/models/dense_heads/fcaf3d_head.py:1181-1184
//   * on the positive locations: BCE-with-logits centerness (fcaf3d_head.py:1206-1210),
//     the 12-d -> 9-DoF box coder (fcaf3d_head.py:1454-1525,1728-1750) and the 4-group
//     decoupled corner Chamfer loss (fcaf3d_head.py:1215-1281,
//     embodiedscan/models/losses/chamfer_distance.py:13-79,160-285) -- differentiated
//     exactly with forward-mode dual numbers carrying the 12 partials (rows A13/A15)
//   * exp/Scale/clamp of the regression distances (fcaf3d_head.py:1135)
#include "common.h"
#include "../../include/es_hip.h"

// ------------------------------------------------------------------ focal
// one wave per row, lanes stride over the classes: coalesced, no 64-bit index division; gamma == 2 (the shipped value)
// squares instead of calling powf
__global__ __launch_bounds__(256) void k_focal(const float* __restrict__ logits, int ldl,
                                               const int* __restrict__ labels, int N, int C, float gamma,
                                               float alpha, const float* __restrict__ avg_factor, float grad_scale,
                                               float* __restrict__ grad, int ldg, double* __restrict__ partial) {
  __shared__ double red[4];
  const float inv = grad_scale / (avg_factor[0] + 1.1920929e-07f);
  const bool g2 = gamma == 2.f;
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < N; i += gridDim.x * 4) {
    const int lab = labels[i];
    const float* row = logits + (size_t)i * ldl;
    float* grow = grad ? grad + (size_t)i * ldg : nullptr;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
      float x = row[c];
      float p = 1.f / (1.f + expf(-x));
      float l, g;
      if (lab == c) {
        float q = 1.f - p;
        float lp = logf(fmaxf(p, 1.17549435e-38f));
        float w = g2 ? q * q : powf(q, gamma);
        l = -alpha * w * lp;
        g = -alpha * w * (1.f - p - gamma * p * lp);
      } else {
        float ln = logf(fmaxf(1.f - p, 1.17549435e-38f));
        float w = g2 ? p * p : powf(p, gamma);
        l = -(1.f - alpha) * w * ln;
        g = -(1.f - alpha) * w * (gamma * (1.f - p) * ln - p);
      }
      acc += l;
      if (grow) grow[c] = g * inv;
    }
    s += (double)acc;
  }
  s = es_wave_sum_d(s);
  if (lane == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_sum_partials(const double* __restrict__ partial, int n, const float* __restrict__ avg_factor,
                               float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = es_wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = red[0] + red[1] + red[2] + red[3];
    out[0] += avg_factor ? (float)t / (avg_factor[0] + 1.1920929e-07f) : (float)t;
  }
}
#define FOCAL_BLOCKS 2048
// loss_out[0] += sum / (avg_factor + eps);  grad = dloss/dlogit * grad_scale.  partial: FOCAL_BLOCKS doubles.
extern "C" int es_focal_loss(const float* logits, int ldl, const int* labels, int N, int C, float gamma, float alpha,
                             const float* avg_factor_dev, float grad_scale, float* grad, int ldg, double* partial,
                             float* loss_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int g = es_cdiv(N, 4);
  if (g > FOCAL_BLOCKS) g = FOCAL_BLOCKS;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_focal, dim3(g), dim3(256), 0, st, logits, ldl, labels, N, C, gamma, alpha, avg_factor_dev,
                     grad_scale, grad, ldg, partial);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, partial, g, avg_factor_dev, loss_out);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ regression distance decode
// bbox[:, :6] = clamp(exp(scale * reg[:, :6]), 1e-3) ; bbox[:, 6:] = reg[:, 6:]
__global__ void k_reg_decode(const float* __restrict__ reg, int ldr, int n, const float* __restrict__ scale,
                             float* __restrict__ bbox) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * 12) return;
  int i = e / 12, c = e - i * 12;
  float v = reg[(size_t)i * ldr + c];
  bbox[e] = c < 6 ? fmaxf(expf(scale[0] * v), 1e-3f) : v;
}
extern "C" int es_reg_decode_fwd(const float* reg, int ldr, int n, const float* scale, float* bbox, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_reg_decode, dim3(es_cdiv(n * 12, 256)), dim3(256), 0, (hipStream_t)stream, reg, ldr, n, scale,
                     bbox);
  ES_CHECK_LAUNCH();
  return 0;
}
// dreg = dbbox * d(decode)/dreg ; dscale += sum dbbox * bbox * reg  (where not clamped)
__global__ void k_reg_decode_bwd(const float* __restrict__ reg, int ldr, const float* __restrict__ bbox,
                                 const float* __restrict__ dbbox, int n, const float* __restrict__ scale,
                                 float* __restrict__ dreg, int ldg, float* __restrict__ dscale) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  float ds = 0.f;
  if (e < n * 12) {
    int i = e / 12, c = e - i * 12;
    float g = dbbox[e];
    if (c < 6) {
      float b = bbox[e];
      bool live = b > 1e-3f;                     // clamp passes gradient only above the floor
      float gb = live ? g * b : 0.f;
      dreg[(size_t)i * ldg + c] = gb * scale[0];
      ds = gb * reg[(size_t)i * ldr + c];
    } else {
      dreg[(size_t)i * ldg + c] = g;
    }
  }
  ds = es_wave_sum(ds);
  if ((threadIdx.x & 63) == 0 && ds != 0.f) atomicAdd(dscale, ds);
}
extern "C" int es_reg_decode_bwd(const float* reg, int ldr, const float* bbox, const float* dbbox, int n,
                                 const float* scale, float* dreg, int ldg, float* dscale, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_reg_decode_bwd, dim3(es_cdiv(n * 12, 256)), dim3(256), 0, (hipStream_t)stream, reg, ldr, bbox,
                     dbbox, n, scale, dreg, ldg, dscale);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ dual numbers (12 partials)
// Evaluated in f64: only a few hundred positive locations exist per scan, so the cost is nil, and the
// atan2/asin/normalise chain of the 6D-rotation coder is ill-conditioned enough that f32 partials were the
// largest noise source of the whole backward pass (measured: 3e-5 vs 7e-6 for the f32 autograd oracle).
#define ND 12
typedef double real;
struct Dual {
  real v;
  real d[ND];
};
__device__ inline Dual dconst(real v) {
  Dual r; r.v = v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = 0;
  return r;
}
__device__ inline Dual dvar(real v, int i) { Dual r = dconst(v); r.d[i] = 1; return r; }
__device__ inline Dual operator+(const Dual& a, const Dual& b) {
  Dual r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ inline Dual operator-(const Dual& a, const Dual& b) {
  Dual r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ inline Dual operator-(const Dual& a) {
  Dual r; r.v = -a.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = -a.d[i];
  return r;
}
__device__ inline Dual operator*(const Dual& a, const Dual& b) {
  Dual r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ inline Dual operator*(const Dual& a, real s) {
  Dual r; r.v = a.v * s;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * s;
  return r;
}
__device__ inline Dual operator/(const Dual& a, const Dual& b) {
  Dual r; r.v = a.v / b.v;
  real ib = 1.0 / b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
__device__ inline Dual dchain(const Dual& a, real v, real dv) {   // f(a) with f' = dv
  Dual r; r.v = v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * dv;
  return r;
}
__device__ inline Dual dsqrt(const Dual& a) { real s = sqrt(a.v); return dchain(a, s, s > 0 ? 0.5 / s : 0.0); }
__device__ inline Dual dsin(const Dual& a) { return dchain(a, sin(a.v), cos(a.v)); }
__device__ inline Dual dcos(const Dual& a) { return dchain(a, cos(a.v), -sin(a.v)); }
__device__ inline Dual dasin(const Dual& a) { return dchain(a, asin(a.v), 1.0 / sqrt(fmax(1.0 - a.v * a.v, 1e-30))); }
__device__ inline Dual datan2(const Dual& y, const Dual& x) {
  Dual r; r.v = atan2(y.v, x.v);
  real den = x.v * x.v + y.v * y.v;
  real gy = den > 0 ? x.v / den : 0.0, gx = den > 0 ? -y.v / den : 0.0;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = y.d[i] * gy + x.d[i] * gx;
  return r;
}
__device__ inline Dual dabs(const Dual& a) { return a.v < 0 ? -a : (a.v > 0 ? a : dconst(0)); }

struct D3 { Dual x, y, z; };
__device__ inline D3 dcross(const D3& a, const D3& b) {
  D3 r;
  r.x = a.y * b.z - a.z * b.y;
  r.y = a.z * b.x - a.x * b.z;
  r.z = a.x * b.y - a.y * b.x;
  return r;
}
__device__ inline D3 dnormalize(const D3& a) {
  Dual n = dsqrt(a.x * a.x + a.y * a.y + a.z * a.z) + dconst(1e-8);
  D3 r; r.x = a.x / n; r.y = a.y / n; r.z = a.z / n;
  return r;
}
// R = Rz(e0) Rx(e1) Ry(e2), row-major 9
__device__ inline void deuler_to_mat(const Dual* e, Dual* R) {
  Dual ca = dcos(e[0]), sa = dsin(e[0]), cb = dcos(e[1]), sb = dsin(e[1]), cc = dcos(e[2]), sc = dsin(e[2]);
  R[0] = ca * cc - sa * sb * sc; R[1] = -(sa * cb); R[2] = ca * sc + sa * sb * cc;
  R[3] = sa * cc + ca * sb * sc; R[4] = ca * cb;    R[5] = sa * sc - ca * sb * cc;
  R[6] = -(cb * sc);             R[7] = sb;         R[8] = cb * cc;
}
// sum over the 8 source corners of min over target corners of the L1 distance
__device__ inline Dual corner_cd(const Dual* box, const real* tc) {
  Dual R[9];
  deuler_to_mat(box + 6, R);
  Dual total = dconst(0);
  const real SX[8] = {1, 1, 1, 1, -1, -1, -1, -1}, SY[8] = {1, 1, -1, -1, 1, 1, -1, -1},
             SZ[8] = {1, -1, 1, -1, 1, -1, 1, -1};
  Dual hx = box[3] * 0.5, hy = box[4] * 0.5, hz = box[5] * 0.5;
  for (int a = 0; a < 8; ++a) {
    Dual ex = hx * SX[a], ey = hy * SY[a], ez = hz * SZ[a];
    Dual cx = box[0] + (ex * R[0] + ey * R[1] + ez * R[2]);
    Dual cy = box[1] + (ex * R[3] + ey * R[4] + ez * R[5]);
    Dual cz = box[2] + (ex * R[6] + ey * R[7] + ez * R[8]);
    Dual best = dconst(0);
    real bv = INFINITY;
    for (int b = 0; b < 8; ++b) {
      Dual dist = dabs(cx - dconst(tc[b * 3])) + dabs(cy - dconst(tc[b * 3 + 1])) + dabs(cz - dconst(tc[b * 3 + 2]));
      if (dist.v < bv) { bv = dist.v; best = dist; }
    }
    total = total + best;
  }
  return total;
}

// one thread per location of ONE sample (all levels, fine -> coarse); rows with cls_t < 0 exit immediately.
// The head outputs live in per-level buffers, the targets in per-sample arrays: the level table maps between them.
struct PosLevels {
  int n;
  int off[ES_MAX_LEVELS + 1];          // row offsets of the levels inside the per-sample arrays
  const float* ho[ES_MAX_LEVELS];      // this sample's rows of the level's head output (column 0 = centerness), ld = ldh
  const float* bbox[ES_MAX_LEVELS];    // decoded (n,12) boxes
  float* dho[ES_MAX_LEVELS];
  float* dbbox[ES_MAX_LEVELS];
};
__global__ __launch_bounds__(64) void k_pos_losses(const int* __restrict__ cls_t, int n,
                                                   const int* __restrict__ n_pos_dev,
                                                   const float* __restrict__ points, PosLevels LV, int ldc,
                                                   const float* __restrict__ center_t,
                                                   const float* __restrict__ bbox_t,
                                                   const float* __restrict__ avg_factor, float grad_scale,
                                                   float w0, float w1, float w2, float w3,
                                                   float* __restrict__ loss_acc /* [0]=center sum, [1]=bbox sum */) {
  // four consecutive lanes share one location: lane q evaluates decouple group q (its corner-Chamfer term carries the
  // f64 dual numbers), the 13 partial results are then summed over the quad with shuffles
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = tid >> 2, grp = tid & 3;
  float lc = 0.f, lb = 0.f;
  const bool active = (i < n) && (cls_t[i] >= 0);
  if (active) {
    const int P = n_pos_dev[0];
    int lv = 0;
    for (int l = 1; l < LV.n; ++l) lv += (i >= LV.off[l]);
    const int li_ = i - LV.off[lv];                       // row inside this sample's slice of the level
    const float* center_pred = LV.ho[lv] + (size_t)li_ * ldc;
    const float* bbox_pred = LV.bbox[lv] + (size_t)li_ * 12;
    float* dcenter = LV.dho[lv] + (size_t)li_ * ldc;
    float* dbbox = LV.dbbox[lv] + (size_t)li_ * 12;
    // ---- centerness BCE with logits, sum / (avg_factor + eps)
    float x = center_pred[0], t = center_t[i];
    float inv_avg = 1.f / (avg_factor[0] + 1.1920929e-07f);
    if (grp == 0) {
      lc = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      float sg = 1.f / (1.f + expf(-x));
      dcenter[0] = (sg - t) * inv_avg * grad_scale;
    }
    // ---- box coder on dual numbers (the 12 head outputs are the independent variables)
    Dual bp[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) bp[c] = dvar(bbox_pred[c], c);
    D3 xr = {bp[6], bp[7], bp[8]}, yr = {bp[9], bp[10], bp[11]};
    D3 y = dnormalize(yr);
    D3 z = dnormalize(dcross(xr, y));
    D3 xo = dcross(y, z);
    // matrix columns are (x, y, z): M[r][0]=xo_r, M[r][1]=y_r, M[r][2]=z_r
    Dual eul[3];
    eul[0] = datan2(-y.x, y.y);          // atan2(-M01, M11)
    eul[1] = dasin(y.z);                 // asin(M21)
    eul[2] = datan2(-xo.z, z.z);         // atan2(-M20, M22)
    Dual R[9];
    deuler_to_mat(eul, R);
    Dual s0 = (bp[1] - bp[0]) * 0.5, s1 = (bp[3] - bp[2]) * 0.5, s2 = (bp[5] - bp[4]) * 0.5;
    Dual dec[9];
    dec[0] = dconst(points[(size_t)i * 3 + 0]) + (s0 * R[0] + s1 * R[1] + s2 * R[2]);
    dec[1] = dconst(points[(size_t)i * 3 + 1]) + (s0 * R[3] + s1 * R[4] + s2 * R[5]);
    dec[2] = dconst(points[(size_t)i * 3 + 2]) + (s0 * R[6] + s1 * R[7] + s2 * R[8]);
    dec[3] = bp[0] + bp[1]; dec[4] = bp[2] + bp[3]; dec[5] = bp[4] + bp[5];
    dec[6] = eul[0]; dec[7] = eul[1]; dec[8] = eul[2];
    // ---- target corners (constants)
    Dual tb[9];
    real tc[24];
#pragma unroll
    for (int c = 0; c < 9; ++c) tb[c] = dconst(bbox_t[(size_t)i * 9 + c]);
    {
      Dual Rt[9];
      deuler_to_mat(tb + 6, Rt);
      const real SX[8] = {1, 1, 1, 1, -1, -1, -1, -1}, SY[8] = {1, 1, -1, -1, 1, 1, -1, -1},
                 SZ[8] = {1, -1, 1, -1, 1, -1, 1, -1};
      for (int a = 0; a < 8; ++a) {
        real ex = tb[3].v * 0.5 * SX[a], ey = tb[4].v * 0.5 * SY[a], ez = tb[5].v * 0.5 * SZ[a];
        tc[a * 3 + 0] = tb[0].v + (ex * Rt[0].v + ey * Rt[1].v + ez * Rt[2].v);
        tc[a * 3 + 1] = tb[1].v + (ex * Rt[3].v + ey * Rt[4].v + ez * Rt[5].v);
        tc[a * 3 + 2] = tb[2].v + (ex * Rt[6].v + ey * Rt[7].v + ez * Rt[8].v);
      }
    }
    Dual v[9];
    Dual tot;
    const real wg = grp == 0 ? (real)w0 : (grp == 1 ? (real)w1 : (grp == 2 ? (real)w2 : (real)w3));
    if (grp == 3) {
      tot = corner_cd(dec, tc) * wg;
    } else {
      // group 0: predicted centre, 1: predicted size, 2: predicted euler; the other components from the target
      for (int c = 0; c < 9; ++c) v[c] = (c / 3 == grp) ? dec[c] : tb[c];
      tot = corner_cd(v, tc) * wg;
    }
    real inv_mean = 1.0 / ((real)P * 8.0);
    // quad reduction of value + 12 partials (lanes 4j .. 4j+3 are always in the same wave)
    real red[13];
    red[0] = tot.v;
#pragma unroll
    for (int c = 0; c < 12; ++c) red[c + 1] = tot.d[c];
#pragma unroll
    for (int c = 0; c < 13; ++c) {
      red[c] += __shfl_xor(red[c], 1, 64);
      red[c] += __shfl_xor(red[c], 2, 64);
    }
    if (grp == 0) {
      lb = (float)(red[0] * inv_mean);
#pragma unroll
      for (int c = 0; c < 12; ++c) dbbox[c] = (float)(red[c + 1] * inv_mean * (real)grad_scale);
    }
  }
  lc = es_wave_sum(lc);
  lb = es_wave_sum(lb);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(loss_acc + 0, lc);
    atomicAdd(loss_acc + 1, lb);
  }
}
extern "C" int es_pos_losses(const int* cls_t, int n, const int* n_pos_dev, const float* points, int n_levels,
                             const int* level_off_host, const void* const* ho_host, const void* const* bbox_host,
                             void* const* dho_host, void* const* dbbox_host, int ldh, const float* center_t,
                             const float* bbox_t, const float* avg_factor_dev, float grad_scale, const float* group_w,
                             float* loss_acc, void* stream) {
  if (n <= 0) return 0;
  if (n_levels > ES_MAX_LEVELS) return -3;
  PosLevels LV;
  LV.n = n_levels;
  for (int l = 0; l <= n_levels; ++l) LV.off[l] = level_off_host[l];
  for (int l = 0; l < n_levels; ++l) {
    LV.ho[l] = (const float*)ho_host[l];
    LV.bbox[l] = (const float*)bbox_host[l];
    LV.dho[l] = (float*)dho_host[l];
    LV.dbbox[l] = (float*)dbbox_host[l];
  }
  hipLaunchKernelGGL(k_pos_losses, dim3(es_cdiv(n * 4, 64)), dim3(64), 0, (hipStream_t)stream, cls_t, n, n_pos_dev,
                     points, LV, ldh, center_t, bbox_t, avg_factor_dev, grad_scale, group_w[0], group_w[1], group_w[2],
                     group_w[3], loss_acc);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ grounding box loss (GroundingHead.loss_by_feat_single,
// dense_heads/grounding_head.py:750-822): the same four decoupled corner-Chamfer terms, but on DIRECT 9-DoF predictions
// (centre, size, Euler) of the Hungarian-matched queries of a whole batch; mean over (pairs x 8 corners), no avg_factor.
// q2g (B*Q): matched ground-truth index local to the sample or -1; gt rows of sample b start at gt_off[b].
// Four consecutive lanes share one query (one decouple group each), like k_pos_losses.
__global__ __launch_bounds__(64) void k_box_cd_pairs(const float* __restrict__ pred, const int* __restrict__ q2g, int Q,
                                                     int n_rows, const float* __restrict__ gt_boxes,
                                                     const int* __restrict__ gt_off, float inv_mean, float grad_scale,
                                                     float w0, float w1, float w2, float w3, float* __restrict__ dpred,
                                                     float* __restrict__ loss_acc) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = tid >> 2, grp = tid & 3;
  float lb = 0.f;
  const bool active = (i < n_rows) && (q2g[i] >= 0);
  if (active) {
    const int b = i / Q;
    const float* tgt = gt_boxes + (size_t)(gt_off[b] + q2g[i]) * 9;
    Dual dec[9], tb[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) { dec[c] = dvar(pred[(size_t)i * 9 + c], c); tb[c] = dconst(tgt[c]); }
    real tc[24];
    {
      Dual Rt[9];
      deuler_to_mat(tb + 6, Rt);
      const real SX[8] = {1, 1, 1, 1, -1, -1, -1, -1}, SY[8] = {1, 1, -1, -1, 1, 1, -1, -1},
                 SZ[8] = {1, -1, 1, -1, 1, -1, 1, -1};
      for (int a = 0; a < 8; ++a) {
        real ex = tb[3].v * 0.5 * SX[a], ey = tb[4].v * 0.5 * SY[a], ez = tb[5].v * 0.5 * SZ[a];
        tc[a * 3 + 0] = tb[0].v + (ex * Rt[0].v + ey * Rt[1].v + ez * Rt[2].v);
        tc[a * 3 + 1] = tb[1].v + (ex * Rt[3].v + ey * Rt[4].v + ez * Rt[5].v);
        tc[a * 3 + 2] = tb[2].v + (ex * Rt[6].v + ey * Rt[7].v + ez * Rt[8].v);
      }
    }
    const real wg = grp == 0 ? (real)w0 : (grp == 1 ? (real)w1 : (grp == 2 ? (real)w2 : (real)w3));
    Dual tot;
    if (grp == 3) {
      tot = corner_cd(dec, tc) * wg;
    } else {
      Dual v[9];
      for (int c = 0; c < 9; ++c) v[c] = (c / 3 == grp) ? dec[c] : tb[c];
      tot = corner_cd(v, tc) * wg;
    }
    real red[10];
    red[0] = tot.v;
#pragma unroll
    for (int c = 0; c < 9; ++c) red[c + 1] = tot.d[c];
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      red[c] += __shfl_xor(red[c], 1, 64);
      red[c] += __shfl_xor(red[c], 2, 64);
    }
    if (grp == 0) {
      lb = (float)(red[0] * (real)inv_mean);
      if (dpred) {
#pragma unroll
        for (int c = 0; c < 9; ++c) dpred[(size_t)i * 9 + c] = (float)(red[c + 1] * (real)inv_mean * (real)grad_scale);
      }
    }
  }
  lb = es_wave_sum(lb);
  if ((threadIdx.x & 63) == 0 && lb != 0.f) atomicAdd(loss_acc, lb);
}
extern "C" int es_box_cd_pairs(const float* pred, const int* q2g, int B, int Q, const float* gt_boxes, const int* gt_off_dev,
                               int n_pairs, float grad_scale, const float* group_w, float* dpred, float* loss_acc,
                               void* stream) {
  int n = B * Q;
  if (n <= 0 || n_pairs <= 0) return 0;
  hipLaunchKernelGGL(k_box_cd_pairs, dim3(es_cdiv(n * 4, 64)), dim3(64), 0, (hipStream_t)stream, pred, q2g, Q, n, gt_boxes,
                     gt_off_dev, 1.0f / ((float)n_pairs * 8.0f), grad_scale, group_w[0], group_w[1], group_w[2], group_w[3], dpred,
                     loss_acc);
  ES_CHECK_LAUNCH();
  return 0;
}
