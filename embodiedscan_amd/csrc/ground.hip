// Grounding-side assignment and classification loss on the device (SURVEY 8a row A19 + 8f row N2):
//   * es_box3d_iou        : exact IoU of two oriented (9-DoF Euler ZXY) boxes -- EulerInstance3DBoxes.overlaps
//                           (structures/bbox_3d/euler_box3d.py:103-135 -> pytorch3d.ops.box3d_overlap), IoU3DCost
//                           (models/losses/match_cost.py:95-113);
//   * es_ground_match     : BinaryFocalLossCost + BBox3DL1Cost + IoU3DCost (match_cost.py:49-75,213-265), nan_to_num and
//                           the rectangular linear-sum-assignment of HungarianAssigner3D.assign
//                           (task_modules/assigners/hungarian_assigner.py:56-138; scipy.optimize.linear_sum_assignment)
//                           for every sample of a decoder layer in ONE launch -- the reference does a D2H copy and a scipy
//                           call per sample per layer;
//   * es_ground_focal     : GroundingHead._get_targets_single label construction + mmdet FocalLoss on the un-padded text
//                           tokens (dense_heads/grounding_head.py:365-425,717-748), value + gradient;
//   * es_topk_sorted      : per-sample top-k indices in descending score order (query selection,
//                           detectors/sparse_featfusion_grounder.py:374-376).
// All geometry and the assignment run in f64 (scipy casts the cost matrix to double as well).
//
// box3d_overlap: pytorch3d (un-vendored) clips the triangulated faces of each box against the other box and sums
// tetrahedra.  Here: the intersection polytope's faces lie in the 12 face planes of the two boxes; each face is that box
// face (a rectangle) clipped by the six half-spaces of the OTHER box (Sutherland-Hodgman in 3-D), and the volume follows
// from the divergence theorem  V = 1/3 sum_f (n_f . x_f) A_f.  Faces of A are clipped inclusively, faces of B exclusively,
// so a pair of coincident faces is counted once (and faces of two boxes that only touch contribute nothing).
#include "common.h"
#include "../../include/es_hip.h"

struct V3 { double x, y, z; };
__device__ inline V3 v3(double x, double y, double z) { V3 r = {x, y, z}; return r; }
__device__ inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ inline double dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 cross3(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

struct OBox { V3 c; V3 ax[3]; double h[3]; };
// R = Rz(a) Rx(b) Ry(g) (pytorch3d euler_angles_to_matrix 'ZXY'); the box axes are R's columns
__device__ inline OBox make_box(const double* b) {
  OBox o;
  o.c = v3(b[0], b[1], b[2]);
  o.h[0] = b[3] * 0.5; o.h[1] = b[4] * 0.5; o.h[2] = b[5] * 0.5;
  double ca = cos(b[6]), sa = sin(b[6]), cb = cos(b[7]), sb = sin(b[7]), cc = cos(b[8]), sc = sin(b[8]);
  o.ax[0] = v3(ca * cc - sa * sb * sc, sa * cc + ca * sb * sc, -(cb * sc));
  o.ax[1] = v3(-(sa * cb), ca * cb, sb);
  o.ax[2] = v3(ca * sc + sa * sb * cc, sa * sc - ca * sb * cc, cb * cc);
  return o;
}
#define MAXV 16
// sum over the 6 faces of P (clipped by Q) of (n . (x - O)) * area
__device__ inline double faces_flux(const OBox& P, const OBox& Q, V3 O, bool inclusive) {
  double total = 0.0;
  for (int f = 0; f < 6; ++f) {
    const int j = f >> 1, k = (j + 1) % 3, l = (j + 2) % 3;
    const double sgn = (f & 1) ? -1.0 : 1.0;
    V3 n = P.ax[j] * sgn;
    V3 fc = P.c + P.ax[j] * (sgn * P.h[j]);
    V3 ek = P.ax[k] * P.h[k], el = P.ax[l] * P.h[l];
    V3 poly[MAXV], tmp[MAXV];
    int np = 4;
    poly[0] = fc + ek + el; poly[1] = fc - ek + el; poly[2] = fc - ek - el; poly[3] = fc + ek - el;
    for (int g = 0; g < 6 && np > 0; ++g) {
      const int jj = g >> 1;
      const double sg = (g & 1) ? -1.0 : 1.0;
      V3 m = Q.ax[jj] * sg;
      const double off = Q.h[jj];
      int nt = 0;
      for (int i = 0; i < np; ++i) {
        V3 a = poly[i], b = poly[(i + 1) % np];
        double da = dot3(m, a - Q.c) - off, db = dot3(m, b - Q.c) - off;
        // a vertex ON the clipping plane (|d| <= tol) is inside only for P = A's faces and only when the two planes face
        // the same way: coincident faces are then counted once, faces of two boxes that merely touch not at all
        const bool same = inclusive && dot3(m, n) > 0.5;
        bool ia = (da < -1e-12) || (same && da <= 1e-12), ib = (db < -1e-12) || (same && db <= 1e-12);
        if (ia) { if (nt < MAXV) tmp[nt++] = a; }
        if (ia != ib) {
          double t = da / (da - db);
          if (nt < MAXV) tmp[nt++] = a + (b - a) * t;
        }
      }
      np = nt;
      for (int i = 0; i < np; ++i) poly[i] = tmp[i];
    }
    if (np < 3) continue;
    V3 av = v3(0, 0, 0);
    for (int i = 1; i + 1 < np; ++i) av = av + cross3(poly[i] - poly[0], poly[i + 1] - poly[0]);
    double area = 0.5 * sqrt(dot3(av, av));
    total += dot3(n, fc - O) * area;
  }
  return total;
}
__device__ inline double box_iou3d(const double* a9, const double* b9) {
  OBox A = make_box(a9), B = make_box(b9);
  double va = a9[3] * a9[4] * a9[5], vb = b9[3] * b9[4] * b9[5];
  // quick reject on bounding spheres
  V3 d = A.c - B.c;
  double ra = sqrt(A.h[0] * A.h[0] + A.h[1] * A.h[1] + A.h[2] * A.h[2]), rb = sqrt(B.h[0] * B.h[0] + B.h[1] * B.h[1] + B.h[2] * B.h[2]);
  if (dot3(d, d) > (ra + rb) * (ra + rb)) return 0.0;
  double v = (faces_flux(A, B, A.c, true) + faces_flux(B, A, A.c, false)) / 3.0;
  if (v < 0.0) v = 0.0;
  return v / (va + vb - v);
}

__global__ void k_box3d_iou(const float* __restrict__ b1, int N, const float* __restrict__ b2, int M, float* __restrict__ iou) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * M) return;
  int i = e / M, j = e % M;
  double a[9], b[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { a[c] = b1[(size_t)i * 9 + c]; b[c] = b2[(size_t)j * 9 + c]; }
  iou[e] = (float)box_iou3d(a, b);
}
extern "C" int es_box3d_iou(const float* boxes1, int N, const float* boxes2, int M, float* iou, void* stream) {
  if (N <= 0 || M <= 0) return 0;
  hipLaunchKernelGGL(k_box3d_iou, dim3(es_cdiv((long long)N * M, 64)), dim3(64), 0, (hipStream_t)stream, boxes1, N, boxes2, M, iou);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ matching costs
// cost[b][g][q] (G rows, Q columns: the orientation scipy's solver works in after its own transpose, nr <= nc)
__global__ void k_ground_cost(const float* __restrict__ logits, int Tout, const float* __restrict__ boxes, int Q,
                              const float* __restrict__ gt_boxes, const unsigned char* __restrict__ pos_map,
                              const int* __restrict__ gt_off, const int* __restrict__ tlen, int T, float w_cls, float w_l1,
                              float w_iou, float alpha, float gamma, float eps, double* __restrict__ cost, int Gmax) {
  const int b = blockIdx.y;
  const int g0 = gt_off[b], G = gt_off[b + 1] - g0;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= G * Q) return;
  const int g = e / Q, q = e % Q;
  const float* lg = logits + ((size_t)b * Q + q) * Tout;
  const unsigned char* pm = pos_map + (size_t)(g0 + g) * T;
  const int tl = min(tlen[b], T);
  // BinaryFocalLossCost on the un-padded tokens (f32 like the reference; summed in double)
  double c_cls = 0.0;
  for (int t = 0; t < tl; ++t) {
    float p = 1.f / (1.f + expf(-lg[t]));
    float neg = -logf(1.f - p + eps) * (1.f - alpha) * powf(p, gamma);
    float pos = -logf(p + eps) * alpha * powf(1.f - p, gamma);
    c_cls += pm[t] ? (double)pos : (double)neg;
  }
  const float* pb = boxes + ((size_t)b * Q + q) * 9;
  const float* gb = gt_boxes + (size_t)(g0 + g) * 9;
  double l1 = 0.0, a9[9], b9[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { a9[c] = pb[c]; b9[c] = gb[c]; l1 += fabs((double)pb[c] - (double)gb[c]); }
  double iou = box_iou3d(a9, b9);
  double c = (double)w_cls * c_cls + (double)w_l1 * (double)(float)l1 + (double)w_iou * -(double)(float)iou;
  if (isnan(c)) c = 100.0;                               // torch.nan_to_num(cost, nan=100, posinf=100, neginf=-100)
  else if (isinf(c)) c = c > 0 ? 100.0 : -100.0;
  cost[((size_t)b * Gmax + g) * Q + q] = c;
}

// ------------------------------------------------------------------ rectangular linear sum assignment
// One thread per sample: the shortest-augmenting-path algorithm scipy.optimize.linear_sum_assignment uses (Crouse 2016),
// including its tie rule (among equal path costs prefer an unassigned column) and its column visiting order, so that
// equal-cost cases resolve the same way.  nr = G rows, nc = Q columns, nr <= nc.  work (doubles): u[G] v[Q] sp[Q];
// iwork (ints): path[Q] col4row[G] row4col[Q] remaining[Q] SR[G] SC[Q].
__global__ void k_lsa(const double* __restrict__ cost, int Gmax, int Q, const int* __restrict__ gt_off, double* __restrict__ work,
                      int* __restrict__ iwork, int* __restrict__ q2g /* (B,Q): matched gt (local index) or -1 */, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nr = gt_off[b + 1] - gt_off[b], nc = Q;
  const double* C = cost + (size_t)b * Gmax * Q;
  double* u = work + (size_t)b * (Gmax + 2 * Q);
  double* v = u + Gmax;
  double* sp = v + Q;
  int* path = iwork + (size_t)b * (4 * Q + 2 * Gmax);
  int* col4row = path + Q;
  int* row4col = col4row + Gmax;
  int* remaining = row4col + Q;
  int* SR = remaining + Q;
  int* SC = SR + Gmax;
  for (int j = 0; j < nc; ++j) { v[j] = 0.0; row4col[j] = -1; q2g[(size_t)b * Q + j] = -1; }
  for (int i = 0; i < nr; ++i) { u[i] = 0.0; col4row[i] = -1; }
  for (int cur = 0; cur < nr; ++cur) {
    double minVal = 0.0;
    int i = cur, num_remaining = nc, sink = -1;
    for (int it = 0; it < nc; ++it) { remaining[it] = nc - it - 1; SC[it] = 0; sp[it] = INFINITY; }
    for (int it = 0; it < nr; ++it) SR[it] = 0;
    while (sink == -1) {
      int index = -1;
      double lowest = INFINITY;
      SR[i] = 1;
      for (int it = 0; it < num_remaining; ++it) {
        int j = remaining[it];
        double r = minVal + C[(size_t)i * Q + j] - u[i] - v[j];
        if (r < sp[j]) { path[j] = i; sp[j] = r; }
        if (sp[j] < lowest || (sp[j] == lowest && row4col[j] == -1)) { lowest = sp[j]; index = it; }
      }
      minVal = lowest;
      if (!(minVal < INFINITY)) { sink = -2; break; }   // infeasible (cannot happen after nan_to_num)
      int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--num_remaining];
    }
    if (sink < 0) break;
    u[cur] += minVal;
    for (int r = 0; r < nr; ++r) if (SR[r] && r != cur) u[r] += minVal - sp[col4row[r]];
    for (int j = 0; j < nc; ++j) if (SC[j]) v[j] -= minVal - sp[j];
    int j = sink;
    while (true) {
      int r = path[j];
      row4col[j] = r;
      int t = col4row[r]; col4row[r] = j; j = t;
      if (r == cur) break;
    }
  }
  for (int r = 0; r < nr; ++r) if (col4row[r] >= 0) q2g[(size_t)b * Q + col4row[r]] = r;
}

// The same algorithm with one WAVE per sample and the solver state in LDS: the column scan of every augmenting step runs
// 64-wide and the argmin is a shuffle reduction that reproduces the sequential tie rule (first column of the minimum
// value, replaced by the LAST unassigned column among the ties).  The single-thread version above spends ~0.5 ms per
// decoder layer in dependent global loads; this one a few tens of microseconds.  Q <= LSA_MAXQ.
#define LSA_MAXQ 1024
__global__ __launch_bounds__(64) void k_lsa_wave(const double* __restrict__ cost, int Gmax, int Q, const int* __restrict__ gt_off,
                                                 int* __restrict__ q2g) {
  __shared__ double u[LSA_MAXQ], v[LSA_MAXQ], sp[LSA_MAXQ];
  __shared__ int path[LSA_MAXQ], col4row[LSA_MAXQ], row4col[LSA_MAXQ], rem[LSA_MAXQ];
  __shared__ unsigned char SR[LSA_MAXQ], SC[LSA_MAXQ];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nr = gt_off[b + 1] - gt_off[b], nc = Q;
  const double* C = cost + (size_t)b * Gmax * Q;
  for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; }
  for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  __syncthreads();
  for (int cur = 0; cur < nr; ++cur) {
    for (int it = lane; it < nc; it += 64) { rem[it] = nc - it - 1; SC[it] = 0; sp[it] = INFINITY; }
    for (int it = lane; it < nr; it += 64) SR[it] = 0;
    __syncthreads();
    double minVal = 0.0;
    int i = cur, num_remaining = nc, sink = -1;
    while (sink == -1) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      double lowest = INFINITY;
      int first = 0x7fffffff, lastU = -1;
      for (int it = lane; it < num_remaining; it += 64) {
        int j = rem[it];
        double r = minVal + C[(size_t)i * Q + j] - ui - v[j];
        if (r < sp[j]) { path[j] = i; sp[j] = r; }
        double val = sp[j];
        bool un = row4col[j] == -1;
        if (val < lowest) { lowest = val; first = it; lastU = un ? it : -1; }
        else if (val == lowest && un) lastU = it;           // `it` grows within a lane: a later tie
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        double ol = __shfl_xor(lowest, o, 64);
        int of = __shfl_xor(first, o, 64), ou = __shfl_xor(lastU, o, 64);
        if (ol < lowest) { lowest = ol; first = of; lastU = ou; }
        else if (ol == lowest) { first = min(first, of); lastU = max(lastU, ou); }
      }
      minVal = lowest;
      if (!(minVal < INFINITY)) { sink = -2; break; }
      const int index = lastU >= 0 ? lastU : first;
      const int j = rem[index];
      const int r4 = row4col[j];
      __syncthreads();                                      // every lane has read rem[] / row4col[] of this step
      if (r4 == -1) sink = j; else i = r4;
      if (lane == 0) { SC[j] = 1; rem[index] = rem[num_remaining - 1]; }
      --num_remaining;
      __syncthreads();
    }
    if (sink < 0) break;
    if (lane == 0) u[cur] += minVal;
    for (int r = lane; r < nr; r += 64) if (SR[r] && r != cur) u[r] += minVal - sp[col4row[r]];
    for (int j = lane; j < nc; j += 64) if (SC[j]) v[j] -= minVal - sp[j];
    __syncthreads();
    if (lane == 0) {
      int j = sink;
      while (true) {
        int r = path[j];
        row4col[j] = r;
        int t = col4row[r]; col4row[r] = j; j = t;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }
  for (int j = lane; j < nc; j += 64) q2g[(size_t)b * Q + j] = -1;
  __syncthreads();
  for (int r = lane; r < nr; r += 64) if (col4row[r] >= 0) q2g[(size_t)b * Q + col4row[r]] = r;
}

extern "C" int es_ground_match(const float* logits, int Tout, const float* boxes, int B, int Q, const float* gt_boxes,
                               const unsigned char* pos_map, const int* gt_off_dev, int Gmax, const int* tlen_dev, int T,
                               float w_cls, float w_l1, float w_iou, double* cost, double* work, int* iwork, int* q2g,
                               void* stream) {
  if (B <= 0 || Q <= 0) return 0;
  if (Gmax > Q) return -5;                               // the solver is written for nr <= nc (more queries than boxes)
  hipStream_t st = (hipStream_t)stream;
  if (Gmax > 0)
    hipLaunchKernelGGL(k_ground_cost, dim3(es_cdiv((long long)Gmax * Q, 64), B), dim3(64), 0, st, logits, Tout, boxes, Q, gt_boxes,
                       pos_map, gt_off_dev, tlen_dev, T, w_cls, w_l1, w_iou, 0.25f, 2.0f, 1e-12f, cost, Gmax);
  if (Q <= LSA_MAXQ)
    hipLaunchKernelGGL(k_lsa_wave, dim3(B), dim3(64), 0, st, cost, Gmax, Q, gt_off_dev, q2g);
  else
    hipLaunchKernelGGL(k_lsa, dim3(es_cdiv(B, 64)), dim3(64), 0, st, cost, Gmax, Q, gt_off_dev, work, iwork, q2g, B);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ labels + focal loss on the valid tokens
// labels[b,q,t] = pos_map[g][t] for the matched box g of query q, else 0; loss = sum over (b, q, t < tlen[b]) of
// BCEwithlogits(x, y) * (alpha y + (1-alpha)(1-y)) * pt^gamma, pt = (1-p) y + p (1-y)   (mmdet py_sigmoid_focal_loss),
// divided by (avg_factor + eps32); gradient written to dlogits (0 at padded tokens).  One wave per (b, q) row.
__global__ __launch_bounds__(256) void k_ground_focal(const float* __restrict__ logits, int Tout, int B, int Q,
                                                      const int* __restrict__ q2g, const unsigned char* __restrict__ pos_map,
                                                      const int* __restrict__ gt_off, const int* __restrict__ tlen, int T,
                                                      float alpha, float gamma, const float* __restrict__ avg_factor,
                                                      float grad_scale, float* __restrict__ dlogits,
                                                      double* __restrict__ loss_sum) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  double acc = 0.0;
  if (row < B * Q) {
    const int b = row / Q;
    const int g = q2g[row];
    const unsigned char* pm = g >= 0 ? pos_map + (size_t)(gt_off[b] + g) * T : nullptr;
    const int tl = min(tlen[b], T);
    const float inv = 1.f / (avg_factor[0] + 1.1920929e-07f);
    for (int t = lane; t < Tout; t += 64) {
      float gout = 0.f;
      if (t < tl) {
        float x = logits[(size_t)row * Tout + t];
        bool y = pm && pm[t];
        float p = 1.f / (1.f + expf(-x));
        float bce = fmaxf(x, 0.f) - (y ? x : 0.f) + log1pf(expf(-fabsf(x)));
        float pt = y ? (1.f - p) : p;
        float fw = (y ? alpha : (1.f - alpha)) * powf(pt, gamma);
        acc += (double)(bce * fw);
        // d/dx [bce * fw]: dbce/dx = p - y ; dfw/dx = w_a * gamma * pt^(gamma-1) * dpt/dx, dpt/dx = (y ? -1 : 1) * p (1-p)
        float dfw = (y ? alpha : (1.f - alpha)) * gamma * powf(pt, gamma - 1.f) * (y ? -1.f : 1.f) * p * (1.f - p);
        gout = ((p - (y ? 1.f : 0.f)) * fw + bce * dfw) * inv * grad_scale;
      }
      if (dlogits) dlogits[(size_t)row * Tout + t] = gout;
    }
  }
  acc = es_wave_sum_d(acc);
  if (lane == 0 && acc != 0.0) unsafeAtomicAdd(loss_sum, acc);
}
extern "C" int es_ground_focal(const float* logits, int Tout, int B, int Q, const int* q2g, const unsigned char* pos_map,
                               const int* gt_off_dev, const int* tlen_dev, int T, float alpha, float gamma,
                               const float* avg_factor_dev, float grad_scale, float* dlogits, double* loss_sum, void* stream) {
  if (B * Q <= 0) return 0;
  hipLaunchKernelGGL(k_ground_focal, dim3(es_cdiv(B * Q, 4)), dim3(256), 0, (hipStream_t)stream, logits, Tout, B, Q, q2g, pos_map,
                     gt_off_dev, tlen_dev, T, alpha, gamma, avg_factor_dev, grad_scale, dlogits, loss_sum);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ per-sample sorted top-k (one workgroup per sample)
// idx[b, 0..k) = rows (local to the sample) of the k largest values in descending order, ties: lower row first.
// Segment length <= 16384 (bitonic sort of (value, row) keys in dynamic LDS: 8 B per key of the next power of two >= L; the
// reference's token lists hold at most 4 levels x pts_prune_threshold = 4000 rows, 128 KB of the CU's 160 KB cover 16384).
#define TK_MAX 16384
__global__ __launch_bounds__(1024) void k_topk_sorted(const float* __restrict__ vals, int L, const int* __restrict__ vlen, int k,
                                                      int* __restrict__ idx) {
  extern __shared__ unsigned long long key[];
  const int b = blockIdx.x;
  const int n = vlen ? min(vlen[b], L) : L;
  int P = 1;
  while (P < n) P <<= 1;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    unsigned long long kk = ~0ull;                      // padding sorts last
    if (i < n) {
      unsigned int u = __float_as_uint(vals[(size_t)b * L + i]);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending order-preserving map
      kk = ((unsigned long long)(~u) << 32) | (unsigned int)i;    // descending value, ascending row
    }
    key[i] = kk;
  }
  __syncthreads();
  for (int sz = 2; sz <= P; sz <<= 1)
    for (int st = sz >> 1; st > 0; st >>= 1) {
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        int j = i ^ st;
        if (j > i) {
          bool up = (i & sz) == 0;
          unsigned long long a = key[i], c = key[j];
          if ((a > c) == up) { key[i] = c; key[j] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < k; i += blockDim.x) idx[(size_t)b * k + i] = i < n ? (int)(key[i] & 0xffffffffu) : -1;
}
extern "C" int es_topk_sorted(const float* vals, int B, int L, const int* vlen_dev, int k, int* idx, void* stream) {
  if (B <= 0 || k <= 0) return 0;
  if (L > TK_MAX) return -4;
  int P2 = 1;
  while (P2 < L) P2 <<= 1;
  const size_t sh = (size_t)P2 * sizeof(unsigned long long);
  if (sh > 64 * 1024) ES_TRY(hipFuncSetAttribute((const void*)k_topk_sorted, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
  hipLaunchKernelGGL(k_topk_sorted, dim3(B), dim3(1024), sh, (hipStream_t)stream, vals, L, vlen_dev, k, idx);
  ES_CHECK_LAUNCH();
  return 0;
}
