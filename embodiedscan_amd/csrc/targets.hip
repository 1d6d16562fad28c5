// Target assignment for the FCAF3D 9-DoF head (row A12), bit-exact labels.
// Restates FCAF3DHeadRotMat.get_targets / _get_face_distances / _get_centerness
// (embodiedscan/models/dense_heads/fcaf3d_head.py:1527-1664) without the ~15 dense
// (N_points, N_boxes[,6]) temporaries of the reference: one pass computes inside flags,
// per-level counts and centerness, a per-box radix select finds the 19th largest
// centerness, a final pass takes the min-volume box per location.
// All float ops are single IEEE roundings in the reference's order (no FMA contraction),
// the box rotations R(-euler) arrive precomputed from the host.
#include "common.h"
#include "../../include/es_hip.h"

#define TG_MAXL 8

struct Levels { int n; int off[TG_MAXL + 1]; };
__device__ inline int level_of(const Levels& L, int i) {
  int g = 0;
  for (int l = 1; l < L.n; ++l) g += (i >= L.off[l]);
  return g;
}

__device__ inline bool face_dist(const float* __restrict__ box, const float* __restrict__ R, float px, float py,
                                 float pz, float* fd) {
  float s0 = __fsub_rn(px, box[0]), s1 = __fsub_rn(py, box[1]), s2 = __fsub_rn(pz, box[2]);
  float mn = INFINITY;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float sh = __fadd_rn(__fadd_rn(__fmul_rn(s0, R[d * 3 + 0]), __fmul_rn(s1, R[d * 3 + 1])),
                         __fmul_rn(s2, R[d * 3 + 2]));
    float cen = __fadd_rn(box[d], sh);
    float half = __fdiv_rn(box[3 + d], 2.f);
    float lo = __fadd_rn(__fsub_rn(cen, box[d]), half);
    float hi = __fsub_rn(__fadd_rn(box[d], half), cen);
    fd[2 * d] = lo;
    fd[2 * d + 1] = hi;
    mn = fminf(mn, fminf(lo, hi));
  }
  return mn > 0.f;
}
__device__ inline float centerness(const float* fd) {
  float v = __fdiv_rn(fminf(fd[0], fd[1]), fmaxf(fd[0], fd[1]));
  v = __fmul_rn(v, fminf(fd[2], fd[3]));
  v = __fdiv_rn(v, fmaxf(fd[2], fd[3]));
  v = __fmul_rn(v, fminf(fd[4], fd[5]));
  v = __fdiv_rn(v, fmaxf(fd[4], fd[5]));
  return sqrtf(v);   // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is the 1-ulp native op
}

// cen[g*N + i] = inside ? centerness : -1 ; npos[l*G + g] += inside
__global__ void k_tg_inside(const float* __restrict__ pts, int N, Levels L, const float* __restrict__ boxes,
                            const float* __restrict__ rot, int G, float* __restrict__ cen, int* __restrict__ npos) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int g = blockIdx.y;
  __shared__ int cnt[TG_MAXL];
  if (threadIdx.x < TG_MAXL) cnt[threadIdx.x] = 0;
  __syncthreads();
  if (i < N) {
    float fd[6];
    bool in = face_dist(boxes + g * 9, rot + g * 9, pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], fd);
    cen[(size_t)g * N + i] = in ? centerness(fd) : -1.f;
    if (in) atomicAdd(&cnt[level_of(L, i)], 1);
  }
  __syncthreads();
  if (threadIdx.x < L.n && cnt[threadIdx.x]) atomicAdd(&npos[threadIdx.x * G + g], cnt[threadIdx.x]);
}

__device__ inline uint32_t f2ord_t(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float ord2f_t(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
#define TG_CAP 8192
// per box: best level (fcaf3d_head.py:1622-1638) then the k-th largest masked centerness
__global__ __launch_bounds__(1024) void k_tg_select(const float* __restrict__ cen, int N, Levels L, int G,
                                                    const int* __restrict__ npos, int assign_thr, int kth,
                                                    int* __restrict__ best_level, float* __restrict__ thr_out) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_remaining;
  __shared__ int s_best;
  int g = blockIdx.x;
  if (threadIdx.x == 0) {
    // lower_limit_mask = n_pos < thr ; lower_index = argmax(mask) - 1 (clamped to 0);
    // all_upper -> n_levels - 1
    int first = -1;
    for (int l = 0; l < L.n; ++l)
      if (npos[l * G + g] < assign_thr) { first = l; break; }
    int best = (first < 0) ? (L.n - 1) : (first - 1 < 0 ? 0 : first - 1);
    s_best = best;
    best_level[g] = best;
    s_prefix = 0;
    s_remaining = (unsigned)min(kth, N);
  }
  __syncthreads();
  int lb = L.off[s_best], le = L.off[s_best + 1];
  const float* cg = cen + (size_t)g * N;
  // Every location outside the best level carries the masked value -1, so the k-th largest of all N values is the
  // k-th largest inside [lb, le) when that range holds at least k values, and -1 otherwise: only the range is scanned.
  if (le - lb < (int)s_remaining) {
    if (threadIdx.x == 0) thr_out[g] = -1.f;
    return;
  }
  // Most locations are outside the box (value -1): they would all hit one histogram bin (serialised LDS atomics), so
  // they are counted separately: with fewer than k candidates (inside points, value >= 0) the threshold is -1,
  // otherwise the k-th largest lies among the candidates and the -1 values can be ignored.
  // Round 4: the candidates of a box are few (the locations inside it: hundreds to a few thousand of up to ~10^5 scanned), so the
  // ONE scan that counts them also collects them in LDS and the four radix passes then run on that array; only a box with more
  // than TG_CAP candidates falls back to re-scanning global memory per pass (this kernel was 0.25 ms per sample: five scans of
  // the level by one workgroup per box).  The result is the k-th largest VALUE: the order of collection does not matter.
  __shared__ unsigned int s_cand, s_live;
  __shared__ float candS[TG_CAP];
  if (threadIdx.x == 0) s_cand = s_live = 0;
  __syncthreads();
  {
    unsigned int c = 0;
    const int lane = threadIdx.x & 63;
    for (int b0 = lb; b0 < le; b0 += blockDim.x) {
      const int i = b0 + threadIdx.x;
      const float v = (i < le) ? cg[i] : -1.f;
      c += (v >= 0.f) ? 1u : 0u;
      const bool live = (i < le) && !(v < 0.f);
      const unsigned long long m = __ballot(live);
      if (m) {
        unsigned int base = 0;
        const int first = __ffsll((long long)m) - 1;
        if (lane == first) base = atomicAdd(&s_live, (unsigned int)__popcll(m));
        base = __shfl(base, first, 64);
        const unsigned int slot = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        if (live && slot < TG_CAP) candS[slot] = v;
      }
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cand, c);
  }
  __syncthreads();
  if (s_cand < s_remaining) {
    if (threadIdx.x == 0) thr_out[g] = -1.f;
    return;
  }
  const bool in_lds = s_live <= TG_CAP;
  const int n_scan = in_lds ? (int)s_live : (le - lb);
  uint32_t pmask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    uint32_t prefix = s_prefix;
    for (int b0 = 0; b0 < n_scan; b0 += blockDim.x) {       // wave-aggregated histogram (see k_topk_mask, rowops.hip)
      const int i = b0 + threadIdx.x;
      const float v = (i < n_scan) ? (in_lds ? candS[i] : cg[lb + i]) : -1.f;
      const uint32_t u = f2ord_t(v);
      const bool live = (i < n_scan) && !(v < 0.f) && ((u & pmask) == prefix);
      const int d = (int)((u >> shift) & 255);
      unsigned long long m = __ballot(live);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        unsigned long long bal = __ballot((d >> b) & 1);
        m &= ((d >> b) & 1) ? bal : ~bal;
      }
      const int lane = threadIdx.x & 63;
      if (live && (m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&hist[d], (unsigned int)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int rem = s_remaining, b = 255;
      for (;; --b) {
        if (hist[b] >= rem) break;
        rem -= hist[b];
        if (b == 0) break;
      }
      s_prefix = prefix | (b << shift);
      s_remaining = rem;
    }
    pmask |= (255u << shift);
    __syncthreads();
  }
  if (threadIdx.x == 0) thr_out[g] = ord2f_t(s_prefix);
}

__global__ void k_tg_assign(const float* __restrict__ cen, int N, Levels L, const float* __restrict__ boxes,
                            const int* __restrict__ labels, int G, const int* __restrict__ best_level,
                            const float* __restrict__ thr, float* __restrict__ center_t, float* __restrict__ bbox_t,
                            int* __restrict__ cls_t, int* __restrict__ box_idx, int* __restrict__ n_pos) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int pos = 0;
  if (i < N) {
    int lv = level_of(L, i);
    float best_vol = INFINITY;
    int best = 0;
    for (int g = 0; g < G; ++g) {
      float c = cen[(size_t)g * N + i];               // -1 unless inside
      bool lev = best_level[g] == lv;
      float cm = lev ? c : -1.f;
      bool ok = (c >= 0.f || c != c) && lev && (cm > thr[g]);
      // `inside` is encoded as c != -1; centerness of an inside point is >= 0
      float vol = ok ? __fmul_rn(__fmul_rn(boxes[g * 9 + 3], boxes[g * 9 + 4]), boxes[g * 9 + 5]) : 1e8f;
      if (vol < best_vol) { best_vol = vol; best = g; }
    }
    float c0 = cen[(size_t)best * N + i];
    center_t[i] = (best_level[best] == lv) ? c0 : -1.f;
    for (int d = 0; d < 9; ++d) bbox_t[(size_t)i * 9 + d] = boxes[best * 9 + d];
    bool none = best_vol == 1e8f;
    cls_t[i] = none ? -1 : labels[best];
    box_idx[i] = none ? -1 : best;
    pos = none ? 0 : 1;
  }
  unsigned long long m = __ballot(pos);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_pos, __popcll(m));
}

// scratch: cen (G*N floats) | npos (L*G ints) | best_level (G ints) | thr (G floats)
extern "C" int es_get_targets(const float* points, int N, const int* level_off, int n_levels, const float* boxes,
                              const float* rot_neg, const int* labels, int G, int assign_thr, int center_thr,
                              float* scratch, float* center_t, float* bbox_t, int* cls_t, int* box_idx,
                              int* n_pos_dev, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_levels > TG_MAXL) return -3;
  ES_TRY(hipMemsetAsync(n_pos_dev, 0, 4, st));
  if (N <= 0) return 0;
  if (G <= 0) {                                   // fcaf3d_head.py:1603-1607
    ES_TRY(hipMemsetAsync(center_t, 0, (size_t)N * 4, st));
    ES_TRY(hipMemsetAsync(bbox_t, 0, (size_t)N * 36, st));
    ES_TRY(hipMemsetAsync(cls_t, 0xFF, (size_t)N * 4, st));
    ES_TRY(hipMemsetAsync(box_idx, 0xFF, (size_t)N * 4, st));
    return 0;
  }
  Levels L;
  L.n = n_levels;
  for (int l = 0; l <= n_levels; ++l) L.off[l] = level_off[l];
  float* cen = scratch;
  int* npos = (int*)(cen + (size_t)G * N);
  int* best = npos + n_levels * G;
  float* thr = (float*)(best + G);
  ES_TRY(hipMemsetAsync(npos, 0, (size_t)n_levels * G * 4, st));
  hipLaunchKernelGGL(k_tg_inside, dim3(es_cdiv(N, 256), G), dim3(256), 0, st, points, N, L, boxes, rot_neg, G, cen,
                     npos);
  hipLaunchKernelGGL(k_tg_select, dim3(G), dim3(1024), 0, st, cen, N, L, G, npos, assign_thr, center_thr + 1, best,
                     thr);
  hipLaunchKernelGGL(k_tg_assign, dim3(es_cdiv(N, 256)), dim3(256), 0, st, cen, N, L, boxes, labels, G, best, thr,
                     center_t, bbox_t, cls_t, box_idx, n_pos_dev);
  ES_CHECK_LAUNCH();
  return 0;
}
