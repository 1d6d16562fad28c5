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
// dreg = dbbox * d(decode)/dreg ; dscale += sum dbbox * bbox * reg  (where not clamped).
// The Scale gradient is reduced deterministically (round 3): a fixed grid of grid-striding blocks, wave sums in a fixed
// shuffle tree, the four wave partials of a block added in wave order, the block partials added in block order by a second
// one-wave launch -- no float atomics.
#define RD_BLOCKS 512
__global__ __launch_bounds__(256) void k_reg_decode_bwd(const float* __restrict__ reg, int ldr, const float* __restrict__ bbox,
                                 const float* __restrict__ dbbox, int n, const float* __restrict__ scale,
                                 float* __restrict__ dreg, int ldg, float* __restrict__ partial) {
  __shared__ float wsum[4];
  float ds = 0.f;
  const size_t tot = (size_t)n * 12;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    size_t i = e / 12;
    int c = (int)(e - i * 12);
    float g = dbbox[e];
    if (c < 6) {
      float b = bbox[e];
      bool live = b > 1e-3f;                     // clamp passes gradient only above the floor
      float gb = live ? g * b : 0.f;
      dreg[i * ldg + c] = gb * scale[0];
      ds += gb * reg[i * ldr + c];
    } else {
      dreg[i * ldg + c] = g;
    }
  }
  ds = es_wave_sum(ds);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ds;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}
__global__ void k_sum_partials_add(const float* __restrict__ partial, int nb, float* __restrict__ dst) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
  s = es_wave_sum(s);
  if (threadIdx.x == 0) dst[0] += s;
}
extern "C" int es_reg_decode_bwd(const float* reg, int ldr, const float* bbox, const float* dbbox, int n,
                                 const float* scale, float* dreg, int ldg, float* dscale, float* partial, void* stream) {
  if (n <= 0) return 0;
  int g = es_cdiv((long long)n * 12, 256);
  if (g > RD_BLOCKS) g = RD_BLOCKS;
  hipLaunchKernelGGL(k_reg_decode_bwd, dim3(g), dim3(256), 0, (hipStream_t)stream, reg, ldr, bbox, dbbox, n, scale, dreg, ldg,
                     partial);
  hipLaunchKernelGGL(k_sum_partials_add, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, g, dscale);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ dual numbers (forward-mode, N partials)
// Evaluated in f64: only a few hundred positive locations exist per scan, and the atan2/asin/normalise chain of the
// 6D-rotation coder is ill-conditioned enough that f32 partials were the largest noise source of the whole backward
// pass (measured: 3e-5 vs 7e-6 for the f32 autograd oracle).  The derivative is split where the chain is narrow:
//   6 rotation outputs --Dual<6>--> Euler angles, rotation matrix       (rot_chain)
//   9 box parameters   --Dual<9>--> corner-Chamfer value and gradient   (corner_cd, nearest target corner chosen on
//                                                                        plain values first, one dual distance per corner)
// and the two Jacobians are multiplied by hand (chain_to_outputs) -- a Dual<12> through everything needed 256 VGPRs +
// 256 AGPRs + scratch and 64 dual distances per corner set.
typedef double real;
template <int N>
struct Dual {
  real v;
  real d[N];
};
template <int N>
__device__ inline Dual<N> dconst(real v) {
  Dual<N> r; r.v = v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = 0;
  return r;
}
template <int N>
__device__ inline Dual<N> dvar(real v, int i) { Dual<N> r = dconst<N>(v); r.d[i] = 1; return r; }
template <int N>
__device__ inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template <int N>
__device__ inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template <int N>
__device__ inline Dual<N> operator-(const Dual<N>& a) {
  Dual<N> r; r.v = -a.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
  return r;
}
template <int N>
__device__ inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
template <int N>
__device__ inline Dual<N> operator*(const Dual<N>& a, real s) {
  Dual<N> r; r.v = a.v * s;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s;
  return r;
}
template <int N>
__device__ inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v / b.v;
  real ib = 1.0 / b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
template <int N>
__device__ inline Dual<N> dchain(const Dual<N>& a, real v, real dv) {   // f(a) with f' = dv
  Dual<N> r; r.v = v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * dv;
  return r;
}
template <int N>
__device__ inline Dual<N> dsqrt(const Dual<N>& a) { real s = sqrt(a.v); return dchain(a, s, s > 0 ? 0.5 / s : 0.0); }
template <int N>
__device__ inline Dual<N> dsin(const Dual<N>& a) { return dchain(a, sin(a.v), cos(a.v)); }
template <int N>
__device__ inline Dual<N> dcos(const Dual<N>& a) { return dchain(a, cos(a.v), -sin(a.v)); }
template <int N>
__device__ inline Dual<N> dasin(const Dual<N>& a) {
  return dchain(a, asin(a.v), 1.0 / sqrt(fmax(1.0 - a.v * a.v, 1e-30)));
}
template <int N>
__device__ inline Dual<N> datan2(const Dual<N>& y, const Dual<N>& x) {
  Dual<N> r; r.v = atan2(y.v, x.v);
  real den = x.v * x.v + y.v * y.v;
  real gy = den > 0 ? x.v / den : 0.0, gx = den > 0 ? -y.v / den : 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = y.d[i] * gy + x.d[i] * gx;
  return r;
}
template <int N>
__device__ inline Dual<N> dabs(const Dual<N>& a) { return a.v < 0 ? -a : (a.v > 0 ? a : dconst<N>(0)); }

template <int N>
struct D3 { Dual<N> x, y, z; };
template <int N>
__device__ inline D3<N> dcross(const D3<N>& a, const D3<N>& b) {
  D3<N> r;
  r.x = a.y * b.z - a.z * b.y;
  r.y = a.z * b.x - a.x * b.z;
  r.z = a.x * b.y - a.y * b.x;
  return r;
}
template <int N>
__device__ inline D3<N> dnormalize(const D3<N>& a) {
  Dual<N> n = dsqrt(a.x * a.x + a.y * a.y + a.z * a.z) + dconst<N>(1e-8);
  D3<N> r; r.x = a.x / n; r.y = a.y / n; r.z = a.z / n;
  return r;
}
// R = Rz(e0) Rx(e1) Ry(e2), row-major 9
template <int N>
__device__ inline void deuler_to_mat(const Dual<N>* e, Dual<N>* R) {
  Dual<N> ca = dcos(e[0]), sa = dsin(e[0]), cb = dcos(e[1]), sb = dsin(e[1]), cc = dcos(e[2]), sc = dsin(e[2]);
  R[0] = ca * cc - sa * sb * sc; R[1] = -(sa * cb); R[2] = ca * sc + sa * sb * cc;
  R[3] = sa * cc + ca * sb * sc; R[4] = ca * cb;    R[5] = sa * sc - ca * sb * cc;
  R[6] = -(cb * sc);             R[7] = sb;         R[8] = cb * cc;
}
__device__ inline void euler_to_mat(const real* e, real* R) {
  real ca = cos(e[0]), sa = sin(e[0]), cb = cos(e[1]), sb = sin(e[1]), cc = cos(e[2]), sc = sin(e[2]);
  R[0] = ca * cc - sa * sb * sc; R[1] = -(sa * cb); R[2] = ca * sc + sa * sb * cc;
  R[3] = sa * cc + ca * sb * sc; R[4] = ca * cb;    R[5] = sa * sc - ca * sb * cc;
  R[6] = -(cb * sc);             R[7] = sb;         R[8] = cb * cc;
}
// the 8 corners of a 9-DoF box (plain values): corner a = centre + R * (+-size/2)
__device__ inline void box_corners(const real* box, real* c) {
  real R[9];
  euler_to_mat(box + 6, R);
  const real SX[8] = {1, 1, 1, 1, -1, -1, -1, -1}, SY[8] = {1, 1, -1, -1, 1, 1, -1, -1},
             SZ[8] = {1, -1, 1, -1, 1, -1, 1, -1};
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    real ex = box[3] * 0.5 * SX[a], ey = box[4] * 0.5 * SY[a], ez = box[5] * 0.5 * SZ[a];
    c[a * 3 + 0] = box[0] + (ex * R[0] + ey * R[1] + ez * R[2]);
    c[a * 3 + 1] = box[1] + (ex * R[3] + ey * R[4] + ez * R[5]);
    c[a * 3 + 2] = box[2] + (ex * R[6] + ey * R[7] + ez * R[8]);
  }
}
// sum over the 8 source corners of min over target corners of the L1 distance; value + gradient g[9] w.r.t. the 9 box
// parameters.  corner_a = centre + R(e) (S_a o size/2): only R needs dual numbers (3 Euler partials); the nearest target
// corner is the first minimum (as torch.min), |.|' = sign with sign(0) = 0 (as torch.abs).
__device__ inline real corner_cd(const real* box, const real* tc, real* g) {
  Dual<3> e[3] = {dvar<3>(box[6], 0), dvar<3>(box[7], 1), dvar<3>(box[8], 2)}, R[9];
  deuler_to_mat(e, R);
  const real h[3] = {box[3] * 0.5, box[4] * 0.5, box[5] * 0.5};
#pragma unroll
  for (int c = 0; c < 9; ++c) g[c] = 0;
  real total = 0;
#pragma unroll 1
  for (int a = 0; a < 8; ++a) {
    const real S[3] = {(a & 4) ? -1.0 : 1.0, (a & 2) ? -1.0 : 1.0, (a & 1) ? -1.0 : 1.0};
    real c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      c[r] = box[r] + ((h[0] * S[0]) * R[r * 3].v + (h[1] * S[1]) * R[r * 3 + 1].v + (h[2] * S[2]) * R[r * 3 + 2].v);
    real bv = INFINITY, bt[3] = {0, 0, 0};                // nearest target corner (registers only: static indices)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      real dist = fabs(c[0] - tc[b * 3]) + fabs(c[1] - tc[b * 3 + 1]) + fabs(c[2] - tc[b * 3 + 2]);
      if (dist < bv) { bv = dist; bt[0] = tc[b * 3]; bt[1] = tc[b * 3 + 1]; bt[2] = tc[b * 3 + 2]; }
    }
    total += bv;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      real df = c[r] - bt[r];
      real sg = df > 0 ? 1.0 : (df < 0 ? -1.0 : 0.0);
      g[r] += sg;
#pragma unroll
      for (int k = 0; k < 3; ++k) g[3 + k] += sg * (0.5 * S[k]) * R[r * 3 + k].v;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        g[6 + j] += sg * ((h[0] * S[0]) * R[r * 3].d[j] + (h[1] * S[1]) * R[r * 3 + 1].d[j] + (h[2] * S[2]) * R[r * 3 + 2].d[j]);
    }
  }
  return total;
}
// decouple group g of the box loss: group 0 takes the predicted centre, 1 the predicted size, 2 the predicted Euler
// angles (the other components from the target), 3 the whole prediction.  Returns the weighted value; g9 = weighted
// gradient w.r.t. the PREDICTED box parameters (zero for the components the group takes from the target).
__device__ inline real decoupled_cd(const real* pred, const real* tgt, const real* tc, int grp, real wg, real* g9) {
  real v[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) v[c] = (grp == 3 || c / 3 == grp) ? pred[c] : tgt[c];
  real g[9];
  real tot = corner_cd(v, tc, g);
#pragma unroll
  for (int c = 0; c < 9; ++c) g9[c] = (grp == 3 || c / 3 == grp) ? g[c] * wg : 0.0;
  return tot * wg;
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
// rows with a positive class target, appended to list[1..] (list[0] = count; order = arrival order of the waves)
__global__ void k_pos_compact(const int* __restrict__ cls_t, int n, int* __restrict__ list, int cap) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool v = i < n && cls_t[i] >= 0;
  unsigned long long m = __ballot(v);
  if (m == 0) return;
  int lane = threadIdx.x & 63, base = 0;
  if (lane == 0) base = atomicAdd(list, __popcll(m));
  base = __shfl(base, 0, 64);
  int pos = base + __popcll(m & ((1ull << lane) - 1ull));
  if (v && pos < cap) list[1 + pos] = i;
}

__global__ __launch_bounds__(64) void k_pos_losses(const int* __restrict__ cls_t, int n,
                                                   const int* __restrict__ pos_list, int cap,
                                                   const int* __restrict__ n_pos_dev,
                                                   const float* __restrict__ points, PosLevels LV, int ldc,
                                                   const float* __restrict__ center_t,
                                                   const float* __restrict__ bbox_t,
                                                   const float* __restrict__ avg_factor, float grad_scale,
                                                   float w0, float w1, float w2, float w3,
                                                   double* __restrict__ loss_acc /* [0]=center sum, [1]=bbox sum */) {
  // four consecutive lanes share one location: lane q evaluates decouple group q (its corner-Chamfer term carries the
  // f64 dual numbers), the 13 partial results are then summed over the quad with shuffles
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = tid >> 2, grp = tid & 3;
  float lc = 0.f, lb = 0.f;
  // the launch covers the host's upper bound `cap` on the positives; the compacted list says how many there are
  const bool active = slot < min(pos_list[0], cap);
  if (active) {
    const int i = pos_list[1 + slot];
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
    // ---- box coder: 12 head outputs -> 9-DoF box.  Only the 6 rotation outputs go through a non-trivial chain
    // (Dual<6>); distances enter linearly and the Jacobian is completed by hand below.
    real bp[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) bp[c] = (real)bbox_pred[c];
    Dual<6> eul[3], R[9];
    {
      D3<6> xr = {dvar<6>(bp[6], 0), dvar<6>(bp[7], 1), dvar<6>(bp[8], 2)};
      D3<6> yr = {dvar<6>(bp[9], 3), dvar<6>(bp[10], 4), dvar<6>(bp[11], 5)};
      D3<6> y = dnormalize(yr);
      D3<6> z = dnormalize(dcross(xr, y));
      D3<6> xo = dcross(y, z);
      // matrix columns are (x, y, z): M[r][0]=xo_r, M[r][1]=y_r, M[r][2]=z_r
      eul[0] = datan2(-y.x, y.y);          // atan2(-M01, M11)
      eul[1] = dasin(y.z);                 // asin(M21)
      eul[2] = datan2(-xo.z, z.z);         // atan2(-M20, M22)
    }
    deuler_to_mat(eul, R);
    const real sh[3] = {(bp[1] - bp[0]) * 0.5, (bp[3] - bp[2]) * 0.5, (bp[5] - bp[4]) * 0.5};
    real dec[9], tb[9], tc[24];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      dec[r] = (real)points[(size_t)i * 3 + r] + (sh[0] * R[r * 3].v + sh[1] * R[r * 3 + 1].v + sh[2] * R[r * 3 + 2].v);
    dec[3] = bp[0] + bp[1]; dec[4] = bp[2] + bp[3]; dec[5] = bp[4] + bp[5];
    dec[6] = eul[0].v; dec[7] = eul[1].v; dec[8] = eul[2].v;
#pragma unroll
    for (int c = 0; c < 9; ++c) tb[c] = (real)bbox_t[(size_t)i * 9 + c];
    box_corners(tb, tc);
    const real wg = grp == 0 ? (real)w0 : (grp == 1 ? (real)w1 : (grp == 2 ? (real)w2 : (real)w3));
    real g9[9];
    real red[13];
    red[0] = decoupled_cd(dec, tb, tc, grp, wg, g9);
    // ---- chain to the 12 head outputs: d dec / d bp
    //   centre_r: -+0.5 R[r][k] w.r.t. the distance pair k, sum_k sh_k dR[r][k] w.r.t. rotation output j
    //   size_k  : 1 w.r.t. both distances of pair k        euler_e: d eul_e w.r.t. rotation output j
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      real gc = g9[0] * R[k].v + g9[1] * R[3 + k].v + g9[2] * R[6 + k].v;
      red[1 + 2 * k] = g9[3 + k] - 0.5 * gc;
      red[2 + 2 * k] = g9[3 + k] + 0.5 * gc;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      real acc = g9[6] * eul[0].d[j] + g9[7] * eul[1].d[j] + g9[8] * eul[2].d[j];
#pragma unroll
      for (int r = 0; r < 3; ++r)
        acc += g9[r] * (sh[0] * R[r * 3].d[j] + sh[1] * R[r * 3 + 1].d[j] + sh[2] * R[r * 3 + 2].d[j]);
      red[7 + j] = acc;
    }
    real inv_mean = 1.0 / ((real)P * 8.0);
    // quad reduction of value + 12 partials (lanes 4j .. 4j+3 are always in the same wave)
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
  // loss VALUES: the positives were compacted in slot-grab order, so the sums are taken in f64 (every f32 term is exact in
  // f64 and the f64 rounding is far below the final f32 rounding: the reported losses do not depend on that order)
  double lcd = es_wave_sum_d((double)lc), lbd = es_wave_sum_d((double)lb);
  if ((threadIdx.x & 63) == 0) {
    unsafeAtomicAdd(loss_acc + 0, lcd);
    unsafeAtomicAdd(loss_acc + 1, lbd);
  }
}
extern "C" int es_pos_losses(const int* cls_t, int n, const int* n_pos_dev, int max_pos, int* pos_ws,
                             const float* points, int n_levels,
                             const int* level_off_host, const void* const* ho_host, const void* const* bbox_host,
                             void* const* dho_host, void* const* dbbox_host, int ldh, const float* center_t,
                             const float* bbox_t, const float* avg_factor_dev, float grad_scale, const float* group_w,
                             double* loss_acc, void* stream) {
  if (n <= 0 || max_pos <= 0) return 0;
  if (n_levels > ES_MAX_LEVELS) return -3;
  if (!pos_ws) return -2;
  if (max_pos > n) max_pos = n;
  PosLevels LV;
  LV.n = n_levels;
  for (int l = 0; l <= n_levels; ++l) LV.off[l] = level_off_host[l];
  for (int l = 0; l < n_levels; ++l) {
    LV.ho[l] = (const float*)ho_host[l];
    LV.bbox[l] = (const float*)bbox_host[l];
    LV.dho[l] = (float*)dho_host[l];
    LV.dbbox[l] = (float*)dbbox_host[l];
  }
  hipStream_t st = (hipStream_t)stream;
  // the positives are a few hundred rows out of ~1e5: compact them first so that the register-heavy loss kernel (one
  // wave per SIMD) is launched over the host's bound only
  if (hipMemsetAsync(pos_ws, 0, sizeof(int), st) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_pos_compact, dim3(es_cdiv(n, 256)), dim3(256), 0, st, cls_t, n, pos_ws, max_pos);
  hipLaunchKernelGGL(k_pos_losses, dim3(es_cdiv(max_pos * 4, 64)), dim3(64), 0, st, cls_t, n, pos_ws, max_pos,
                     n_pos_dev, points, LV, ldh, center_t, bbox_t, avg_factor_dev, grad_scale, group_w[0], group_w[1],
                     group_w[2], group_w[3], loss_acc);
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
                                                     double* __restrict__ loss_acc) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = tid >> 2, grp = tid & 3;
  float lb = 0.f;
  const bool active = (i < n_rows) && (q2g[i] >= 0);
  if (active) {
    const int b = i / Q;
    const float* tgt = gt_boxes + (size_t)(gt_off[b] + q2g[i]) * 9;
    real dec[9], tb[9], tc[24];
#pragma unroll
    for (int c = 0; c < 9; ++c) { dec[c] = (real)pred[(size_t)i * 9 + c]; tb[c] = (real)tgt[c]; }
    box_corners(tb, tc);
    const real wg = grp == 0 ? (real)w0 : (grp == 1 ? (real)w1 : (grp == 2 ? (real)w2 : (real)w3));
    real red[10];
    red[0] = decoupled_cd(dec, tb, tc, grp, wg, red + 1);
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
  // loss VALUE only (gradients are per row).  Round 6: accumulated in f64 like k_pos_losses / k_ground_focal -- the f32 atomic made the
  // reported loss_bbox depend on the arrival order of ~190 workgroups (1 ulp from run to run: tools/bisect_determinism.py found every
  // Var and gradient of two grounder builds bit-identical and only these scalars apart); a wave's f32 sum is taken in a fixed order, the
  // f64 sum of those exact f32 terms does not depend on the order after rounding back to f32
  if ((threadIdx.x & 63) == 0 && lb != 0.f) unsafeAtomicAdd(loss_acc, (double)lb);
}
extern "C" int es_box_cd_pairs(const float* pred, const int* q2g, int B, int Q, const float* gt_boxes, const int* gt_off_dev,
                               int n_pairs, float grad_scale, const float* group_w, float* dpred, double* loss_acc,
                               void* stream) {
  int n = B * Q;
  if (n <= 0 || n_pairs <= 0) return 0;
  hipLaunchKernelGGL(k_box_cd_pairs, dim3(es_cdiv(n * 4, 64)), dim3(64), 0, (hipStream_t)stream, pred, q2g, Q, n, gt_boxes,
                     gt_off_dev, 1.0f / ((float)n_pairs * 8.0f), grad_scale, group_w[0], group_w[1], group_w[2], group_w[3], dpred,
                     loss_acc);
  ES_CHECK_LAUNCH();
  return 0;
}
