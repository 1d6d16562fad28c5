// Row-matrix operators on (N, C) f32 feature matrices: batch / instance norm (train mode)
// with fused ReLU / ELU / residual, max pooling, row gather / scatter, strided 2-D copies.
// All HBM-bound; channels are the fast axis so every access is coalesced along C.
// Replaces MinkowskiBatchNorm / MinkowskiInstanceNorm / MinkowskiReLU / MinkowskiELU /
// MinkowskiMaxPooling / MinkowskiPruning / ME.cat at
//   embodiedscan/models/backbones/mink_resnet.py:64-69 (+ ME BasicBlock)
//   embodiedscan/models/dense_heads/fcaf3d_head.py:919-947,1113
//   embodiedscan/models/detectors/sparse_featfusion_single_stage.py:210-219
#include "common.h"
#include "../../include/es_hip.h"

// Rows per statistics chunk: 128 for small levels (enough chunks to fill the chip), up to 512 for the 1e5..1e6-row levels so
// that the finalize kernels do not walk thousands of partials (round 2: 2969 partials per channel on the finest level,
// finalize kernels at 15-18 us for a few-hundred-float reduction).
int ES_OPT_NORM_CHUNK = 0;               // es_set_option key 9: rows per statistics chunk (0: the rule below)
static int norm_chunk_rows(int max_rows) {
  if (ES_OPT_NORM_CHUNK > 0) return ES_OPT_NORM_CHUNK;
  return max_rows <= (1 << 17) ? 128 : (max_rows <= (1 << 18) ? 256 : 512);
}
// finalize kernels: a block owns FCH channels x FST chunk stripes (round 2: 64 x 16 -> ONE block for a 64-channel level)
#define FCH 16
#define FST 64

struct Segs { int n; int off[ES_MAX_SEG + 1]; };
__device__ inline int seg_of(const Segs& s, int row) {
  int g = 0;
  for (int i = 1; i < s.n; ++i) g += (row >= s.off[i]);
  return g;
}
static Segs make_segs(const int* seg_off, int nseg) {
  Segs s;
  s.n = nseg;
  for (int i = 0; i <= nseg; ++i) s.off[i] = seg_off[i];
  return s;
}
static int max_seg_rows(const Segs& s) {
  int m = 0;
  for (int i = 0; i < s.n; ++i) m = max(m, s.off[i + 1] - s.off[i]);
  return m;
}

// partial[(seg*nchunk + chunk)*2*C + {0,1}*C + c] = {sum, M2 about the chunk mean}  (Chan et al. merge in the
// finalize kernel: robust against |mean| >> std, unlike E[x^2] - E[x]^2).  One pass: sums are taken about the chunk's
// first row (a shift that is itself a sample), M2 = q - s^2/n.  256 threads = (C/4 column lanes) x (row lanes), float4
// loads, LDS reduction over the row lanes.
__global__ __launch_bounds__(256) void k_norm_stats(const float* __restrict__ x, int ldx, int C, Segs segs, int nchunk,
                                                    int NCH, float* __restrict__ partial) {
  __shared__ float4 red[2][256];
  int seg = blockIdx.y, chunk = blockIdx.x;
  int r0 = segs.off[seg] + chunk * NCH, r1 = min(segs.off[seg + 1], r0 + NCH);
  if (r0 >= r1) return;
  float* out = partial + ((size_t)(seg * nchunk + chunk) * 2) * C;
  const bool vec = ((C & 3) == 0) && ((ldx & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
  if (!vec) {                                              // generic scalar path (C = 3 image rows never get here)
    float inv = 1.f / (float)(r1 - r0);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float s = 0.f;
      for (int r = r0; r < r1; ++r) s += x[(size_t)r * ldx + c];
      float m = s * inv, q = 0.f;
      for (int r = r0; r < r1; ++r) { float d = x[(size_t)r * ldx + c] - m; q += d * d; }
      out[c] = s;
      out[C + c] = q;
    }
    return;
  }
  int C4 = C >> 2;
  int TX = C4 < 64 ? C4 : 64, TY = 256 / TX;
  int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  float nrows = (float)(r1 - r0);
  for (int cb = 0; cb < C4; cb += TX) {
    int c4 = cb + tx;
    float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0), kk = make_float4(0, 0, 0, 0);
    if (c4 < C4 && ty < TY) {
      kk = *(const float4*)(x + (size_t)r0 * ldx + c4 * 4);
      for (int r = r0 + ty; r < r1; r += TY) {
        float4 v = *(const float4*)(x + (size_t)r * ldx + c4 * 4);
        float dx = v.x - kk.x, dy = v.y - kk.y, dz = v.z - kk.z, dw = v.w - kk.w;
        s.x += dx; s.y += dy; s.z += dz; s.w += dw;
        q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
      }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (ty == 0 && c4 < C4) {
      for (int j = 1; j < TY; ++j) {
        float4 a = red[0][j * TX + tx], b = red[1][j * TX + tx];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
      }
      float4 so = make_float4(s.x + nrows * kk.x, s.y + nrows * kk.y, s.z + nrows * kk.z, s.w + nrows * kk.w);
      float4 qo = make_float4(q.x - s.x * s.x / nrows, q.y - s.y * s.y / nrows, q.z - s.z * s.z / nrows,
                              q.w - s.w * s.w / nrows);
      *(float4*)(out + c4 * 4) = so;
      *(float4*)(out + C + c4 * 4) = make_float4(fmaxf(qo.x, 0.f), fmaxf(qo.y, 0.f), fmaxf(qo.z, 0.f), fmaxf(qo.w, 0.f));
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(FCH * FST) void k_norm_finalize(const float* __restrict__ partial, int C, Segs segs,
                                                            int nchunk, int NCH, float eps, float* __restrict__ mean,
                                                            float* __restrict__ invstd, float* running_mean,
                                                            float* running_var, float momentum) {
  __shared__ double red[FST][FCH];
  int seg = blockIdx.y, cl = threadIdx.x % FCH, st = threadIdx.x / FCH;
  int c = blockIdx.x * FCH + cl;
  int r0 = segs.off[seg], n = segs.off[seg + 1] - r0;
  int used = (n + NCH - 1) / NCH;
  // pass 1: total sum -> mean
  double s = 0;
  if (c < C)
    for (int ch = st; ch < used; ch += FST) s += partial[((size_t)(seg * nchunk + ch) * 2) * C + c];
  red[st][cl] = s;
  __syncthreads();
  double tot = 0;
  for (int i = 0; i < FST; ++i) tot += red[i][cl];
  double m = n > 0 ? tot / n : 0.0;
  __syncthreads();
  // pass 2: M2 = sum_c [ M2_c + n_c (m_c - m)^2 ]
  double q = 0;
  if (c < C)
    for (int ch = st; ch < used; ch += FST) {
      const float* p = partial + ((size_t)(seg * nchunk + ch) * 2) * C;
      int nc = min(NCH, n - ch * NCH);
      double d = (double)p[c] / nc - m;
      q += (double)p[C + c] + nc * d * d;
    }
  red[st][cl] = q;
  __syncthreads();
  if (st == 0 && c < C) {
    double m2 = 0;
    for (int i = 0; i < FST; ++i) m2 += red[i][cl];
    double var = n > 0 ? m2 / n : 0.0;
    mean[seg * C + c] = (float)m;
    invstd[seg * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean && n > 1) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * n / (n - 1));
    }
  }
}
__device__ inline float act_fwd(float z, int act) {
  if (act == 1) return z > 0.f ? z : 0.f;
  if (act == 2) return z > 0.f ? z : (expf(z) - 1.f);
  return z;
}
__global__ void k_norm_apply(const float* __restrict__ x, int ldx, int n, int C, Segs segs,
                             const float* __restrict__ mean, const float* __restrict__ invstd,
                             const float* __restrict__ w, const float* __restrict__ b,
                             const float* __restrict__ res, int ldr, int act, float* __restrict__ y, int ldy) {
  size_t tot = (size_t)n * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int r = (int)(e / C), c = (int)(e - (size_t)r * C);
    int sg = seg_of(segs, r);
    float z = (x[(size_t)r * ldx + c] - mean[sg * C + c]) * invstd[sg * C + c] * w[c] + b[c];
    if (res) z += res[(size_t)r * ldr + c];
    y[(size_t)r * ldy + c] = act_fwd(z, act);
  }
}

// float4 version of k_norm_apply (same arithmetic per element).  Two independent row pieces per loop trip: these passes
// are latency-bound unless every thread keeps several 16-B loads in flight.
__global__ __launch_bounds__(256) void k_norm_apply4(const float* __restrict__ x, int ldx, int n, int C, Segs segs,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     const float* __restrict__ res, int ldr, int act,
                                                     float* __restrict__ y, int ldy, unsigned short* __restrict__ yh) {
  const int C4 = C >> 2;
  const size_t tot = (size_t)n * C4, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < tot; e0 += 2 * stride) {
    size_t e1 = e0 + stride;
    bool v1 = e1 < tot;
    int r0 = (int)(e0 / C4), c0 = (int)(e0 - (size_t)r0 * C4) * 4;
    int r1 = v1 ? (int)(e1 / C4) : r0, c1 = v1 ? (int)(e1 - (size_t)r1 * C4) * 4 : c0;
    float4 x0 = *(const float4*)(x + (size_t)r0 * ldx + c0), x1 = *(const float4*)(x + (size_t)r1 * ldx + c1);
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0;
    if (res) { q0 = *(const float4*)(res + (size_t)r0 * ldr + c0); q1 = *(const float4*)(res + (size_t)r1 * ldr + c1); }
#define NA_ONE(xv, qv, r, c)                                                                                   \
    {                                                                                                          \
      int sg = seg_of(segs, r);                                                                                \
      float4 m = *(const float4*)(mean + sg * C + c), is = *(const float4*)(invstd + sg * C + c);             \
      float4 ww = *(const float4*)(w + c), bb = *(const float4*)(b + c);                                       \
      float z0 = (xv.x - m.x) * is.x * ww.x + bb.x, z1 = (xv.y - m.y) * is.y * ww.y + bb.y;                    \
      float z2 = (xv.z - m.z) * is.z * ww.z + bb.z, z3 = (xv.w - m.w) * is.w * ww.w + bb.w;                    \
      if (res) { z0 += qv.x; z1 += qv.y; z2 += qv.z; z3 += qv.w; }                                             \
      z0 = act_fwd(z0, act); z1 = act_fwd(z1, act); z2 = act_fwd(z2, act); z3 = act_fwd(z3, act);             \
      *(float4*)(y + (size_t)r * ldy + c) = make_float4(z0, z1, z2, z3);                                       \
      if (yh) *(uint2*)(yh + (size_t)r * C + c) = make_uint2(es_pack_bf16(z0, z1), es_pack_bf16(z2, z3));      \
    }
    NA_ONE(x0, q0, r0, c0)
    if (v1) NA_ONE(x1, q1, r1, c1)
#undef NA_ONE
  }
}

// ------------------------------------------------------------------ column-block norm for SHORT matrices (round 4)
// The three-launch pipeline above (row-chunk statistics -> finalize -> apply) is built for long matrices; on the deep levels of
// the sparse networks (190 .. 12 000 rows x 256 .. 1024 channels: 29 of the 47 norm layers of an mv-3ddet step) each of its
// launches runs for 4 - 10 us and the two launch boundaries on the step's dependent chain cost more than the kernels.  Here ONE
// workgroup of 512 threads owns 16 channels (4 float4 lanes x 128 row stripes) of ALL rows: per-channel statistics need no
// other workgroup, so statistics, finalize and apply are one launch (backward: statistics + parameter gradients + apply).
// Statistics: per-thread f32 sums about the first row (a shift that is itself a sample), combined over the 1024 threads in f64
// in a fixed order (wave shuffles, then 8 wave partials) -- deterministic; var = Q/n - (S/n)^2 in f64 on shifted sums.
// The apply arithmetic is the k_norm_apply4 / k_norm_bwd_apply4 expression, element for element.
int ES_OPT_ELECT_SAFE = 0;            // es_set_option key 18: 1 = agent-scope release / acquire fences in the last-workgroup elections (common.h)
int ES_OPT_NORM_CB_ROWS = 4096;       // es_set_option key 15: matrices with at most this many rows take the one-launch path (0: off).
                                      // Measured on the mv-3ddet step (profiles/r4j_sweep.txt): 4096 -> 26.6 ms, off -> 26.8, 16384 -> 27.3
                                      // (8 .. 16 k rows x 256 channels are only 16 workgroups: too few), 65536 -> 33.1
int ES_OPT_NORM_CB_BWD = 1;           // es_set_option key 17: the one-launch path for the backward pass too (0: forward only)
#define NCB_TY 128                     // 512 threads: with 1024 the 128-VGPR budget spilled 26 / 38 registers to scratch (resource guard)
#define NCB_WAVES 8
__device__ inline void cb_reduce4(double v[4], double (*sm)[4][4]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tx = threadIdx.x & 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) v[i] += __shfl_xor(v[i], o, 64);
  }
  __syncthreads();                                 // (sm may still be read by the previous reduction)
  if ((lane >> 2) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[wv][tx][i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double tot = 0;
#pragma unroll
    for (int w = 0; w < NCB_WAVES; ++w) tot += sm[w][tx][i];
    v[i] = tot;
  }
}
__global__ __launch_bounds__(512) void k_norm_fwd_cb(const float* __restrict__ x, int ldx, int n, int C, float eps,
                                                      const float* __restrict__ w, const float* __restrict__ b,
                                                      const float* __restrict__ res, int ldr, int act, float* running_mean,
                                                      float* running_var, float momentum, float* __restrict__ mean,
                                                      float* __restrict__ invstd, float* __restrict__ y, int ldy,
                                                      unsigned short* __restrict__ yh) {
  __shared__ double sm[NCB_WAVES][4][4];
  const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2, c = blockIdx.x * 16 + tx * 4;
  const float4 k = *(const float4*)(x + c);                                    // row 0: the shift
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  for (int rb = ty; rb < n; rb += 4 * NCB_TY) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = rb + u * NCB_TY;
      const float4 t = *(const float4*)(x + (size_t)(r < n ? r : rb) * ldx + c);
      v[u].x = r < n ? t.x : k.x; v[u].y = r < n ? t.y : k.y;                   // (a padding row contributes 0; selecting VALUES: a select
      v[u].z = r < n ? t.z : k.z; v[u].w = r < n ? t.w : k.w;                   //  between the load and `k` itself put k into scratch memory)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float dx = v[u].x - k.x, dy = v[u].y - k.y, dz = v[u].z - k.z, dw = v[u].w - k.w;
      s.x += dx; s.y += dy; s.z += dz; s.w += dw;
      q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
    }
  }
  double a[4] = {s.x, s.y, s.z, s.w}, bq[4] = {q.x, q.y, q.z, q.w};
  cb_reduce4(a, sm);
  cb_reduce4(bq, sm);
  const double inv_n = 1.0 / (double)n;
  const float kk[4] = {k.x, k.y, k.z, k.w};
  float m4[4], is4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double ms = a[i] * inv_n, var = bq[i] * inv_n - ms * ms;
    if (var < 0) var = 0;
    m4[i] = (float)((double)kk[i] + ms);
    is4[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (ty == 0) {
      mean[c + i] = m4[i];
      invstd[c + i] = is4[i];
      if (running_mean && n > 1) {
        running_mean[c + i] = (1.f - momentum) * running_mean[c + i] + momentum * m4[i];
        running_var[c + i] = (1.f - momentum) * running_var[c + i] + momentum * (float)(var * n / (n - 1));
      }
    }
  }
  const float4 ww = *(const float4*)(w + c), bb = *(const float4*)(b + c);
  for (int rb = ty; rb < n; rb += 4 * NCB_TY) {
    float4 v[4], qv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int r = rb + u * NCB_TY;
      int rc = r < n ? r : rb;
      v[u] = *(const float4*)(x + (size_t)rc * ldx + c);
      if (res) qv[u] = *(const float4*)(res + (size_t)rc * ldr + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int r = rb + u * NCB_TY;
      if (r >= n) continue;
      float z0 = (v[u].x - m4[0]) * is4[0] * ww.x + bb.x, z1 = (v[u].y - m4[1]) * is4[1] * ww.y + bb.y;
      float z2 = (v[u].z - m4[2]) * is4[2] * ww.z + bb.z, z3 = (v[u].w - m4[3]) * is4[3] * ww.w + bb.w;
      if (res) { z0 += qv[u].x; z1 += qv[u].y; z2 += qv[u].z; z3 += qv[u].w; }
      z0 = act_fwd(z0, act); z1 = act_fwd(z1, act); z2 = act_fwd(z2, act); z3 = act_fwd(z3, act);
      *(float4*)(y + (size_t)r * ldy + c) = make_float4(z0, z1, z2, z3);
      if (yh) *(uint2*)(yh + (size_t)r * C + c) = make_uint2(es_pack_bf16(z0, z1), es_pack_bf16(z2, z3));
    }
  }
}
static bool norm_cb_ok(int n, int C, int nseg) {
  return ES_OPT_NORM_CB_ROWS > 0 && nseg == 1 && n >= 1 && n <= ES_OPT_NORM_CB_ROWS && (C & 15) == 0;
}

// workspace floats: nseg * cdiv(max_seg_rows, 64) * 2 * C
extern "C" int es_norm_fwd(const float* x, int ldx, int n, int C, const int* seg_off, int nseg, float eps,
                           const float* weight, const float* bias, const float* res, int ldr, int act,
                           float* running_mean, float* running_var, float momentum, float* mean, float* invstd,
                           float* workspace, float* y, int ldy, void* y_bf16, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0 || nseg > ES_MAX_SEG) return nseg > ES_MAX_SEG ? -3 : 0;
  Segs s = make_segs(seg_off, nseg);
  const bool vec0 = ((C & 3) == 0) && ((ldx & 3) == 0) && ((ldy & 3) == 0) && (!res || (ldr & 3) == 0) &&
                    (((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res) | ((uintptr_t)weight) | ((uintptr_t)bias) |
                       ((uintptr_t)mean) | ((uintptr_t)invstd)) & 15) == 0);
  if (vec0 && norm_cb_ok(n, C, nseg) && seg_off[0] == 0 && seg_off[1] == n) {      // short matrix: one launch (see k_norm_fwd_cb)
    hipLaunchKernelGGL(k_norm_fwd_cb, dim3(C / 16), dim3(4 * NCB_TY), 0, st, x, ldx, n, C, eps, weight, bias, res, ldr, act,
                       running_mean, running_var, momentum, mean, invstd, y, ldy, (unsigned short*)y_bf16);
    ES_CHECK_LAUNCH();
    return 0;
  }
  const int NCH = norm_chunk_rows(max_seg_rows(s));
  int nchunk = es_cdiv(max_seg_rows(s), NCH);
  if (nchunk < 1) nchunk = 1;
  hipLaunchKernelGGL(k_norm_stats, dim3(nchunk, nseg), dim3(256), 0, st, x, ldx, C, s, nchunk, NCH,
                     workspace);
  hipLaunchKernelGGL(k_norm_finalize, dim3(es_cdiv(C, FCH), nseg), dim3(FCH * FST), 0, st, workspace, C, s, nchunk, NCH, eps,
                     mean, invstd, running_mean, running_var, momentum);
  const bool vec = ((C & 3) == 0) && ((ldx & 3) == 0) && ((ldy & 3) == 0) && (!res || (ldr & 3) == 0) &&
                   (((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res) | ((uintptr_t)weight) | ((uintptr_t)bias) |
                      ((uintptr_t)mean) | ((uintptr_t)invstd)) & 15) == 0);
  if (vec) {
    int g = es_cdiv((long long)n * (C >> 2), 512);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_norm_apply4, dim3(g < 1 ? 1 : g), dim3(256), 0, st, x, ldx, n, C, s, mean, invstd, weight, bias,
                       res, ldr, act, y, ldy, (unsigned short*)y_bf16);
  } else {
    int g = es_cdiv((long long)n * C, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_norm_apply, dim3(g), dim3(256), 0, st, x, ldx, n, C, s, mean, invstd, weight, bias, res,
                       ldr, act, y, ldy);
    if (y_bf16) return es_cast_rows_bf16(y, ldy, n, C, y_bf16, stream);
  }
  ES_CHECK_LAUNCH();
  return 0;
}
extern "C" size_t es_norm_workspace_floats(int n, int C, const int* seg_off, int nseg) {
  int m = 0;
  for (int i = 0; i < nseg; ++i) m = max(m, seg_off[i + 1] - seg_off[i]);
  int nchunk = es_cdiv(m, norm_chunk_rows(m));
  if (nchunk < 1) nchunk = 1;
  return (size_t)nseg * nchunk * 2 * C;
}

// backward pass 1: dz = dy * act'(y) (written in place into dy), partial sums of dz and dz*xhat
// (same 2-D thread layout / float4 accesses as k_norm_stats)
__global__ __launch_bounds__(256) void k_norm_bwd_stats(float* __restrict__ dy, int ldd, const float* __restrict__ y,
                                                        int ldy, const float* __restrict__ x, int ldx, int C, Segs segs,
                                                        int nchunk, int NCH, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, int act,
                                                        float* __restrict__ partial) {
  __shared__ float4 red[2][256];
  int seg = blockIdx.y, chunk = blockIdx.x;
  int r0 = segs.off[seg] + chunk * NCH, r1 = min(segs.off[seg + 1], r0 + NCH);
  float* out = partial + ((size_t)(seg * nchunk + chunk) * 2) * C;
  const bool vec = ((C & 3) == 0) && ((ldx & 3) == 0) && ((ldd & 3) == 0) && ((ldy & 3) == 0) &&
                   ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)dy) & 15) == 0) && ((((uintptr_t)y) & 15) == 0);
  if (!vec) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float s = 0.f, q = 0.f, m = mean[seg * C + c], is = invstd[seg * C + c];
      for (int r = r0; r < r1; ++r) {
        float g = dy[(size_t)r * ldd + c];
        if (act) {
          float yv = y[(size_t)r * ldy + c];
          g = (act == 1) ? (yv > 0.f ? g : 0.f) : (yv > 0.f ? g : g * (yv + 1.f));
          dy[(size_t)r * ldd + c] = g;
        }
        s += g;
        q += g * ((x[(size_t)r * ldx + c] - m) * is);
      }
      out[c] = s;
      out[C + c] = q;
    }
    return;
  }
  int C4 = C >> 2;
  int TX = C4 < 64 ? C4 : 64, TY = 256 / TX;
  int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  for (int cb = 0; cb < C4; cb += TX) {
    int c4 = cb + tx;
    float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
    if (c4 < C4 && ty < TY) {
      float4 m = *(const float4*)(mean + seg * C + c4 * 4), is = *(const float4*)(invstd + seg * C + c4 * 4);
      // four rows per trip, all loads first: the in-place store of dz would otherwise order every row's loads behind
      // the previous row's store (same pointer), leaving one 16-B load in flight per thread
      for (int rb = r0 + ty; rb < r1; rb += 4 * TY) {
        float4 g[4], yv[4], xv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int r = rb + u * TY;
          ok[u] = r < r1;
          int rc = ok[u] ? r : rb;
          g[u] = *(const float4*)(dy + (size_t)rc * ldd + c4 * 4);
          if (act) yv[u] = *(const float4*)(y + (size_t)rc * ldy + c4 * 4);
          xv[u] = *(const float4*)(x + (size_t)rc * ldx + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
          int r = rb + u * TY;
          float4 gg = g[u];
          if (act) {
            float4 yy = yv[u];
            if (act == 1) {
              gg.x = yy.x > 0.f ? gg.x : 0.f; gg.y = yy.y > 0.f ? gg.y : 0.f;
              gg.z = yy.z > 0.f ? gg.z : 0.f; gg.w = yy.w > 0.f ? gg.w : 0.f;
            } else {
              gg.x = yy.x > 0.f ? gg.x : gg.x * (yy.x + 1.f); gg.y = yy.y > 0.f ? gg.y : gg.y * (yy.y + 1.f);
              gg.z = yy.z > 0.f ? gg.z : gg.z * (yy.z + 1.f); gg.w = yy.w > 0.f ? gg.w : gg.w * (yy.w + 1.f);
            }
            *(float4*)(dy + (size_t)r * ldd + c4 * 4) = gg;
          }
          float4 xx = xv[u];
          s.x += gg.x; s.y += gg.y; s.z += gg.z; s.w += gg.w;
          q.x += gg.x * ((xx.x - m.x) * is.x); q.y += gg.y * ((xx.y - m.y) * is.y);
          q.z += gg.z * ((xx.z - m.z) * is.z); q.w += gg.w * ((xx.w - m.w) * is.w);
        }
      }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (ty == 0 && c4 < C4) {
      for (int j = 1; j < TY; ++j) {
        float4 a = red[0][j * TX + tx], b = red[1][j * TX + tx];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
      }
      *(float4*)(out + c4 * 4) = s;
      *(float4*)(out + C + c4 * 4) = q;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(FCH * FST) void k_norm_bwd_finalize(const float* __restrict__ partial, int C, Segs segs,
                                                                int nchunk, int NCH, float* __restrict__ sum_dz,
                                                                float* __restrict__ sum_dzx, float* dweight,
                                                                float* dbias) {
  __shared__ double red[2][FST][FCH];
  int cl = threadIdx.x % FCH, st = threadIdx.x / FCH;
  int c = blockIdx.x * FCH + cl;
  double tw = 0, tb = 0;
  for (int seg = 0; seg < segs.n; ++seg) {
    int n = segs.off[seg + 1] - segs.off[seg];
    int used = (n + NCH - 1) / NCH;
    double s = 0, q = 0;
    if (c < C)
      for (int ch = st; ch < used; ch += FST) {
        const float* p = partial + ((size_t)(seg * nchunk + ch) * 2) * C;
        s += p[c];
        q += p[C + c];
      }
    red[0][st][cl] = s;
    red[1][st][cl] = q;
    __syncthreads();
    if (st == 0 && c < C) {
      double ts = 0, tq = 0;
      for (int i = 0; i < FST; ++i) { ts += red[0][i][cl]; tq += red[1][i][cl]; }
      sum_dz[seg * C + c] = (float)ts;
      sum_dzx[seg * C + c] = (float)tq;
      tb += ts;
      tw += tq;
    }
    __syncthreads();
  }
  if (st == 0 && c < C) {
    if (dweight) dweight[c] += (float)tw;
    if (dbias) dbias[c] += (float)tb;
  }
}
__global__ void k_norm_bwd_apply(const float* __restrict__ dz, int ldd, const float* __restrict__ x, int ldx, int n,
                                 int C, Segs segs, const float* __restrict__ mean, const float* __restrict__ invstd,
                                 const float* __restrict__ w, const float* __restrict__ sum_dz,
                                 const float* __restrict__ sum_dzx, float* __restrict__ dx, int ldo, int accumulate) {
  size_t tot = (size_t)n * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int r = (int)(e / C), c = (int)(e - (size_t)r * C);
    int sg = seg_of(segs, r);
    float inv_n = 1.f / (float)(segs.off[sg + 1] - segs.off[sg]);
    float is = invstd[sg * C + c];
    float xh = (x[(size_t)r * ldx + c] - mean[sg * C + c]) * is;
    float g = w[c] * is * (dz[(size_t)r * ldd + c] - sum_dz[sg * C + c] * inv_n - xh * sum_dzx[sg * C + c] * inv_n);
    float* p = dx + (size_t)r * ldo + c;
    *p = accumulate ? (*p + g) : g;
  }
}
__global__ __launch_bounds__(256) void k_norm_bwd_apply4(const float* __restrict__ dz, int ldd,
                                                         const float* __restrict__ x, int ldx, int n, int C, Segs segs,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ w,
                                                         const float* __restrict__ sum_dz,
                                                         const float* __restrict__ sum_dzx, float* __restrict__ dx,
                                                         int ldo, int accumulate, unsigned short* __restrict__ dxh) {
  const int C4 = C >> 2;
  const size_t tot = (size_t)n * C4, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < tot; e0 += 2 * stride) {
    size_t e1 = e0 + stride;
    bool v1 = e1 < tot;
    int r0 = (int)(e0 / C4), c0 = (int)(e0 - (size_t)r0 * C4) * 4;
    int r1 = v1 ? (int)(e1 / C4) : r0, c1 = v1 ? (int)(e1 - (size_t)r1 * C4) * 4 : c0;
    float4 x0 = *(const float4*)(x + (size_t)r0 * ldx + c0), x1 = *(const float4*)(x + (size_t)r1 * ldx + c1);
    float4 g0 = *(const float4*)(dz + (size_t)r0 * ldd + c0), g1 = *(const float4*)(dz + (size_t)r1 * ldd + c1);
    float4 o0 = make_float4(0, 0, 0, 0), o1 = o0;
    if (accumulate) { o0 = *(const float4*)(dx + (size_t)r0 * ldo + c0); o1 = *(const float4*)(dx + (size_t)r1 * ldo + c1); }
#define NB_ELT(xe, ge, oe, me, ie, we, se, qe) \
    { float xh = ((xe) - (me)) * (ie); float gg = (we) * (ie) * ((ge) - (se) * inv_n - xh * (qe) * inv_n); oe = accumulate ? ((oe) + gg) : gg; }
#define NB_ONE(xv, gv, ov, r, c)                                                                               \
    {                                                                                                          \
      int sg = seg_of(segs, r);                                                                                \
      float inv_n = 1.f / (float)(segs.off[sg + 1] - segs.off[sg]);                                            \
      float4 m = *(const float4*)(mean + sg * C + c), is = *(const float4*)(invstd + sg * C + c);             \
      float4 ww = *(const float4*)(w + c), sd = *(const float4*)(sum_dz + sg * C + c);                         \
      float4 sq = *(const float4*)(sum_dzx + sg * C + c);                                                      \
      NB_ELT(xv.x, gv.x, ov.x, m.x, is.x, ww.x, sd.x, sq.x) NB_ELT(xv.y, gv.y, ov.y, m.y, is.y, ww.y, sd.y, sq.y) \
      NB_ELT(xv.z, gv.z, ov.z, m.z, is.z, ww.z, sd.z, sq.z) NB_ELT(xv.w, gv.w, ov.w, m.w, is.w, ww.w, sd.w, sq.w) \
      *(float4*)(dx + (size_t)r * ldo + c) = ov;                                                               \
      if (dxh) *(uint2*)(dxh + (size_t)r * C + c) = make_uint2(es_pack_bf16(ov.x, ov.y), es_pack_bf16(ov.z, ov.w)); \
    }
    NB_ONE(x0, g0, o0, r0, c0)
    if (v1) NB_ONE(x1, g1, o1, r1, c1)
#undef NB_ONE
#undef NB_ELT
  }
}
// one-launch backward for short matrices (see k_norm_fwd_cb): dz = dy * act'(y) in place, per-channel sums of dz and dz * xhat
// (f32 per thread, f64 across the workgroup, fixed order), parameter gradients (the workgroup is the only writer of its 16
// channels), then the k_norm_bwd_apply4 expression on the rows the same thread has just written.
__global__ __launch_bounds__(512) void k_norm_bwd_cb(float* __restrict__ dy, int ldd, const float* __restrict__ y, int ldy,
                                                      const float* __restrict__ x, int ldx, int n, int C,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ w, int act, float* dweight, float* dbias,
                                                      float* __restrict__ dx, int ldo, int accumulate,
                                                      unsigned short* __restrict__ dxh) {
  __shared__ double sm[NCB_WAVES][4][4];
  const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2, c = blockIdx.x * 16 + tx * 4;
  const float4 m = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  for (int rb = ty; rb < n; rb += 4 * NCB_TY) {
    float4 g[4], yv[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int r = rb + u * NCB_TY;
      int rc = r < n ? r : rb;
      g[u] = *(const float4*)(dy + (size_t)rc * ldd + c);
      if (act) yv[u] = *(const float4*)(y + (size_t)rc * ldy + c);
      xv[u] = *(const float4*)(x + (size_t)rc * ldx + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int r = rb + u * NCB_TY;
      if (r >= n) continue;
      float4 gg = g[u];
      if (act) {
        float4 yy = yv[u];
        if (act == 1) {
          gg.x = yy.x > 0.f ? gg.x : 0.f; gg.y = yy.y > 0.f ? gg.y : 0.f;
          gg.z = yy.z > 0.f ? gg.z : 0.f; gg.w = yy.w > 0.f ? gg.w : 0.f;
        } else {
          gg.x = yy.x > 0.f ? gg.x : gg.x * (yy.x + 1.f); gg.y = yy.y > 0.f ? gg.y : gg.y * (yy.y + 1.f);
          gg.z = yy.z > 0.f ? gg.z : gg.z * (yy.z + 1.f); gg.w = yy.w > 0.f ? gg.w : gg.w * (yy.w + 1.f);
        }
        *(float4*)(dy + (size_t)r * ldd + c) = gg;
      }
      float4 xx = xv[u];
      s.x += gg.x; s.y += gg.y; s.z += gg.z; s.w += gg.w;
      q.x += gg.x * ((xx.x - m.x) * is.x); q.y += gg.y * ((xx.y - m.y) * is.y);
      q.z += gg.z * ((xx.z - m.z) * is.z); q.w += gg.w * ((xx.w - m.w) * is.w);
    }
  }
  double a[4] = {s.x, s.y, s.z, s.w}, bq[4] = {q.x, q.y, q.z, q.w};
  cb_reduce4(a, sm);
  cb_reduce4(bq, sm);
  if (ty == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (dweight) dweight[c + i] += (float)bq[i];
      if (dbias) dbias[c + i] += (float)a[i];
    }
  }
  const float sd[4] = {(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
  const float sq[4] = {(float)bq[0], (float)bq[1], (float)bq[2], (float)bq[3]};
  const float4 ww = *(const float4*)(w + c);
  const float inv_n = 1.f / (float)n;
  for (int rb = ty; rb < n; rb += 4 * NCB_TY) {
    float4 g[4], xv[4], ov[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int r = rb + u * NCB_TY;
      int rc = r < n ? r : rb;
      g[u] = *(const float4*)(dy + (size_t)rc * ldd + c);                       // dz: written above by this very thread
      xv[u] = *(const float4*)(x + (size_t)rc * ldx + c);
      ov[u] = accumulate ? *(const float4*)(dx + (size_t)rc * ldo + c) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int r = rb + u * NCB_TY;
      if (r >= n) continue;
#define NCB_ELT(xe, ge, oe, me, ie, we, se, qe) \
      { float xh = ((xe) - (me)) * (ie); float gg = (we) * (ie) * ((ge) - (se) * inv_n - xh * (qe) * inv_n); oe = accumulate ? ((oe) + gg) : gg; }
      NCB_ELT(xv[u].x, g[u].x, ov[u].x, m.x, is.x, ww.x, sd[0], sq[0]) NCB_ELT(xv[u].y, g[u].y, ov[u].y, m.y, is.y, ww.y, sd[1], sq[1])
      NCB_ELT(xv[u].z, g[u].z, ov[u].z, m.z, is.z, ww.z, sd[2], sq[2]) NCB_ELT(xv[u].w, g[u].w, ov[u].w, m.w, is.w, ww.w, sd[3], sq[3])
#undef NCB_ELT
      *(float4*)(dx + (size_t)r * ldo + c) = ov[u];
      if (dxh) *(uint2*)(dxh + (size_t)r * C + c) = make_uint2(es_pack_bf16(ov[u].x, ov[u].y), es_pack_bf16(ov[u].z, ov[u].w));
    }
  }
}
// dy is overwritten with dz (= gradient w.r.t. the pre-activation, which is also the
// gradient of the residual input).  workspace as in es_norm_fwd plus 2*nseg*C floats.
extern "C" int es_norm_bwd(float* dy, int ldd, const float* y, int ldy, const float* x, int ldx, int n, int C,
                           const int* seg_off, int nseg, const float* mean, const float* invstd,
                           const float* weight, int act, float* dweight, float* dbias, float* workspace, float* dx,
                           int ldo, int accumulate, void* dx_bf16, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0 || nseg > ES_MAX_SEG) return nseg > ES_MAX_SEG ? -3 : 0;
  Segs s = make_segs(seg_off, nseg);
  const int NCH = norm_chunk_rows(max_seg_rows(s));
  int nchunk = es_cdiv(max_seg_rows(s), NCH);
  if (nchunk < 1) nchunk = 1;
  {
    const bool vec0 = ((C & 3) == 0) && ((ldx & 3) == 0) && ((ldd & 3) == 0) && ((ldo & 3) == 0) && ((ldy & 3) == 0) &&
                      (((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)y) | ((uintptr_t)weight) |
                         ((uintptr_t)mean) | ((uintptr_t)invstd)) & 15) == 0);
    if (vec0 && dx != dy && ES_OPT_NORM_CB_BWD && norm_cb_ok(n, C, nseg) && seg_off[0] == 0 && seg_off[1] == n) {
      hipLaunchKernelGGL(k_norm_bwd_cb, dim3(C / 16), dim3(4 * NCB_TY), 0, st, dy, ldd, y, ldy, x, ldx, n, C, mean, invstd, weight, act,
                         dweight, dbias, dx, ldo, accumulate, (unsigned short*)dx_bf16);
      ES_CHECK_LAUNCH();
      return 0;
    }
  }
  float* sums = workspace + (size_t)nseg * nchunk * 2 * C;
  hipLaunchKernelGGL(k_norm_bwd_stats, dim3(nchunk, nseg), dim3(256), 0, st, dy, ldd, y, ldy, x,
                     ldx, C, s, nchunk, NCH, mean, invstd, act, workspace);
  hipLaunchKernelGGL(k_norm_bwd_finalize, dim3(es_cdiv(C, FCH)), dim3(FCH * FST), 0, st, workspace, C, s, nchunk, NCH, sums,
                     sums + (size_t)nseg * C, dweight, dbias);
  const bool vec = ((C & 3) == 0) && ((ldx & 3) == 0) && ((ldd & 3) == 0) && ((ldo & 3) == 0) &&
                   (((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)weight) | ((uintptr_t)mean) |
                      ((uintptr_t)invstd) | ((uintptr_t)sums)) & 15) == 0) && ((((size_t)nseg * C) & 3) == 0);
  if (vec && dx != dy) {
    int g = es_cdiv((long long)n * (C >> 2), 512);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_norm_bwd_apply4, dim3(g < 1 ? 1 : g), dim3(256), 0, st, dy, ldd, x, ldx, n, C, s, mean, invstd,
                       weight, sums, sums + (size_t)nseg * C, dx, ldo, accumulate, (unsigned short*)dx_bf16);
  } else {
    int g = es_cdiv((long long)n * C, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_norm_bwd_apply, dim3(g), dim3(256), 0, st, dy, ldd, x, ldx, n, C, s, mean, invstd, weight,
                       sums, sums + (size_t)nseg * C, dx, ldo, accumulate);
    if (dx_bf16) return es_cast_rows_bf16(dx, ldo, n, C, dx_bf16, stream);
  }
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ max pooling (k=2,s=2)
__global__ void k_maxpool_fwd(const float* __restrict__ x, int ldx, const int* __restrict__ nbr, int n_out, int K,
                              int C, float* __restrict__ y, int* __restrict__ arg) {
  size_t tot = (size_t)n_out * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int j = (int)(e / C), c = (int)(e - (size_t)j * C);
    float best = -INFINITY;
    int bi = -1;
    for (int k = 0; k < K; ++k) {
      int i = nbr[(size_t)j * K + k];
      if (i < 0) continue;
      float v = x[(size_t)i * ldx + c];
      if (bi < 0 || v > best) { best = v; bi = i; }     // first tap wins ties
    }
    y[e] = best;
    arg[e] = bi;
  }
}
extern "C" int es_maxpool_fwd(const float* x, int ldx, const int* nbr, int n_out, int K, int C, float* y, int* arg,
                              void* stream) {
  if (n_out <= 0) return 0;
  int g = es_cdiv((long long)n_out * C, 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_maxpool_fwd, dim3(g), dim3(256), 0, (hipStream_t)stream, x, ldx, nbr, n_out, K, C, y, arg);
  ES_CHECK_LAUNCH();
  return 0;
}
// forward-only max pooling of f32 rows into bf16 rows (the frozen stem of the image backbone: no argmax is kept)
__global__ void k_maxpool_fwd_h(const float* __restrict__ x, int ldx, const int* __restrict__ nbr, int n_out, int K, int C,
                                unsigned short* __restrict__ y) {
  size_t tot = (size_t)n_out * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int j = (int)(e / C), c = (int)(e - (size_t)j * C);
    float best = -INFINITY;
    bool any = false;
    for (int k = 0; k < K; ++k) {
      int i = nbr[(size_t)j * K + k];
      if (i < 0) continue;
      float v = x[(size_t)i * ldx + c];
      if (!any || v > best) { best = v; any = true; }
    }
    float o = any ? best : 0.f;
    // RNE to bf16 (the values are finite)
    uint32_t u = __float_as_uint(o);
    u += 0x7fffu + ((u >> 16) & 1u);
    y[e] = (unsigned short)(u >> 16);
  }
}
extern "C" int es_maxpool_fwd_h(const float* x, int ldx, const int* nbr, int n_out, int K, int C, void* y_bf16, void* stream) {
  if (n_out <= 0) return 0;
  int g = es_cdiv((long long)n_out * C, 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_maxpool_fwd_h, dim3(g), dim3(256), 0, (hipStream_t)stream, x, ldx, nbr, n_out, K, C, (unsigned short*)y_bf16);
  ES_CHECK_LAUNCH();
  return 0;
}
__global__ void k_maxpool_bwd(const float* __restrict__ dy, const int* __restrict__ arg, int n_out, int C,
                              float* __restrict__ dx, int ldo) {
  size_t tot = (size_t)n_out * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(e % C);
    int i = arg[e];
    if (i >= 0) dx[(size_t)i * ldo + c] += dy[e];        // windows are disjoint: no race
  }
}
extern "C" int es_maxpool_bwd(const float* dy, const int* arg, int n_out, int C, float* dx, int ldo, void* stream) {
  if (n_out <= 0) return 0;
  int g = es_cdiv((long long)n_out * C, 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_maxpool_bwd, dim3(g), dim3(256), 0, (hipStream_t)stream, dy, arg, n_out, C, dx, ldo);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ gather / scatter / copies
// mode 0: dst[i] = src[idx[i]]   mode 1: dst[idx[i]] += src[i] (idx unique)   mode 2: dst[idx[i]] = src[i]
__global__ void k_row_move(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds,
                           const int* __restrict__ idx, int n, int C, int mode) {
  size_t tot = (size_t)n * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int i = (int)(e / C), c = (int)(e - (size_t)i * C);
    int r = idx ? idx[i] : i;
    if (r < 0) continue;
    if (mode == 0) dst[(size_t)i * ldd + c] = src[(size_t)r * lds + c];
    else if (mode == 1) dst[(size_t)r * ldd + c] += src[(size_t)i * lds + c];
    else dst[(size_t)r * ldd + c] = src[(size_t)i * lds + c];
  }
}
__global__ void k_row_move4(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds,
                            const int* __restrict__ idx, int n, int C4, int mode) {
  size_t tot = (size_t)n * C4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int i = (int)(e / C4), c = (int)(e - (size_t)i * C4) * 4;
    int r = idx ? idx[i] : i;
    if (r < 0) continue;
    if (mode == 0) {
      *(float4*)(dst + (size_t)i * ldd + c) = *(const float4*)(src + (size_t)r * lds + c);
    } else if (mode == 1) {
      float4 a = *(const float4*)(dst + (size_t)r * ldd + c), b = *(const float4*)(src + (size_t)i * lds + c);
      *(float4*)(dst + (size_t)r * ldd + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    } else {
      *(float4*)(dst + (size_t)r * ldd + c) = *(const float4*)(src + (size_t)i * lds + c);
    }
  }
}
extern "C" int es_row_move(float* dst, int ldd, const float* src, int lds, const int* idx, int n, int C, int mode,
                           void* stream) {
  if (n <= 0 || C <= 0) return 0;
  if ((C % 4 == 0) && (ldd % 4 == 0) && (lds % 4 == 0) && (((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0)) {
    int g4 = es_cdiv((long long)n * (C / 4), 256);
    if (g4 > 8192) g4 = 8192;
    hipLaunchKernelGGL(k_row_move4, dim3(g4), dim3(256), 0, (hipStream_t)stream, dst, ldd, src, lds, idx, n, C / 4, mode);
    ES_CHECK_LAUNCH();
    return 0;
  }
  int g = es_cdiv((long long)n * C, 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_row_move, dim3(g), dim3(256), 0, (hipStream_t)stream, dst, ldd, src, lds, idx, n, C, mode);
  ES_CHECK_LAUNCH();
  return 0;
}
// dst (op)= alpha * src over a strided (n, C) block; op 0 assign, 1 add
__global__ void k_axpy2d(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds, int n, int C,
                         float alpha, int op) {
  size_t tot = (size_t)n * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int i = (int)(e / C), c = (int)(e - (size_t)i * C);
    float v = alpha * src[(size_t)i * lds + c];
    float* p = dst + (size_t)i * ldd + c;
    *p = op ? (*p + v) : v;
  }
}
extern "C" int es_axpy2d(float* dst, int ldd, const float* src, int lds, int n, int C, float alpha, int op,
                         void* stream) {
  if (n <= 0 || C <= 0) return 0;
  int g = es_cdiv((long long)n * C, 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_axpy2d, dim3(g), dim3(256), 0, (hipStream_t)stream, dst, ldd, src, lds, n, C, alpha, op);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ interpolated scores + top-k prune mask
// s[i] = sum_k w[i,k] * score[idx[i,k]]  (fixed k order; absent corner contributes 0)
__global__ void k_interp_scores(const float* __restrict__ score, const int* __restrict__ idx,
                                const float* __restrict__ w, int n, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int r = idx[i * 8 + k];
    float v = (r >= 0) ? score[r] : 0.f;
    s = s + v * ((r >= 0) ? w[i * 8 + k] : 0.f);
  }
  out[i] = s;
}
extern "C" int es_interp_scores(const float* score, const int* idx, const float* w, int n, float* out,
                                void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_interp_scores, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, score, idx, w, n,
                     out);
  ES_CHECK_LAUNCH();
  return 0;
}

__device__ inline uint32_t f2ord(float f) {   // order preserving float -> uint
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// one workgroup per segment: mask[i] = 1 for the k largest values of v[off[s]:off[s+1]]
// (ties at the threshold: lower row first).  Radix select, 4 passes of 8 bits.
__global__ __launch_bounds__(1024) void k_topk_mask(const float* __restrict__ v, Segs segs, int kkeep,
                                                    int* __restrict__ mask) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_remaining, s_carry;
  __shared__ int wsum[16];
  int seg = blockIdx.x;
  int r0 = segs.off[seg], r1 = segs.off[seg + 1], n = r1 - r0;
  if (kkeep >= n) {
    for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) mask[i] = 1;
    return;
  }
  if (threadIdx.x == 0) { s_prefix = 0; s_remaining = kkeep; }
  uint32_t pmask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    uint32_t prefix = s_prefix;
    // wave-aggregated histogram: the scores of a level share their leading digits, so plain per-element LDS atomics all
    // hit two or three bins and serialise (session C: 390 us for 380 k scores, one workgroup per sample, on the head's
    // critical path).  The lanes of a wave that hold the same digit are found with 8 ballots; one lane adds their count.
    for (int b0 = r0; b0 < r1; b0 += blockDim.x) {
      const int i = b0 + threadIdx.x;
      uint32_t u = (i < r1) ? f2ord(v[i]) : 0u;
      const bool live = (i < r1) && ((u & pmask) == prefix);
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
      for (;; --b) {                      // walk buckets from the largest digit
        if (hist[b] >= rem) break;
        rem -= hist[b];
        if (b == 0) break;
      }
      s_prefix = prefix | (b << shift);
      s_remaining = rem;                  // how many to take from the bucket equal to the threshold
    }
    pmask |= (255u << shift);
    __syncthreads();
  }
  uint32_t thr = s_prefix;
  unsigned int take_eq = s_remaining;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int b0 = r0; b0 < r1; b0 += blockDim.x) {     // ordered pass: rank the ties by row
    int i = b0 + threadIdx.x;
    uint32_t u = (i < r1) ? f2ord(v[i]) : 0;
    int eq = (i < r1) && (u == thr);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = eq;
    for (int o = 1; o < 64; o <<= 1) {
      int tt = __shfl_up(inc, o, 64);
      if (lane >= o) inc += tt;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = s_carry, tot = 0;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) {
      if (q < w) base += wsum[q];
      tot += wsum[q];
    }
    if (i < r1) mask[i] = (u > thr) || (eq && (unsigned)(base + inc - 1) < take_eq);
    __syncthreads();
    if (threadIdx.x == 0) s_carry += tot;
    __syncthreads();
  }
}
extern "C" int es_topk_mask(const float* values, const int* seg_off, int nseg, int k, int* mask, void* stream) {
  if (nseg <= 0 || nseg > ES_MAX_SEG) return nseg > ES_MAX_SEG ? -3 : 0;
  Segs s = make_segs(seg_off, nseg);
  hipLaunchKernelGGL(k_topk_mask, dim3(nseg), dim3(1024), 0, (hipStream_t)stream, values, s, k, mask);
  ES_CHECK_LAUNCH();
  return 0;
}

// ---- the same selection spread over many workgroups (round 4).  k_topk_mask walks a segment with ONE workgroup five times
// (four 8-bit radix passes + the ordered tie pass): 0.5 - 0.6 ms for the ~10^5 scores per sample of the head's finest level, a single
// launch on the step's dependent chain.  Here a segment is cut into TKP slices; per radix pass one launch: every workgroup adds the
// digit histogram of its slice (elements that match the prefix found so far) to the segment's global histogram with INTEGER atomics
// (exact, order-free), the last workgroup of the segment to arrive walks the 256 buckets, extends the prefix and clears the
// histogram.  Then the tie pass: per-slice counts of elements equal to the threshold, scanned by the segment's last workgroup,
// and a final launch writes the mask (ties in ascending row order, like k_topk_mask: bit-identical masks).
// Workspace (ints; tickets and histograms zero-initialised once and left at zero by every call).  The layout is FIXED for
// ES_MAX_SEG segments -- [tickets MAX | state MAX x 2 (prefix, remaining) | hist MAX x 256 | eq counts MAX x TKP | eq bases
// MAX x TKP] -- so that calls with different segment counts can share one workspace (a layout that depended on nseg put the
// tickets of a 12-segment call onto the leftover state of a 1-segment call: found as a memory fault in round 4).
#define TKP 32
#define TK_STATE ES_MAX_SEG
#define TK_HIST (3 * ES_MAX_SEG)
#define TK_CNT (TK_HIST + 256 * ES_MAX_SEG)
#define TK_BASE (TK_CNT + TKP * ES_MAX_SEG)
#define TK_INTS (TK_BASE + TKP * ES_MAX_SEG)
__global__ __launch_bounds__(256) void k_topk_hist(const float* __restrict__ v, Segs segs, int kkeep, int shift, uint32_t pmask,
                                                   unsigned int* __restrict__ wsp, int safe) {
  const int seg = blockIdx.y, r0 = segs.off[seg], r1 = segs.off[seg + 1], n = r1 - r0;
  if (kkeep >= n) return;                               // (segment-uniform: every workgroup of the segment leaves)
  unsigned int* tickets = wsp;
  unsigned int* state = wsp + TK_STATE + 2 * seg;
  unsigned int* hist = wsp + TK_HIST + 256 * seg;
  __shared__ unsigned int lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = shift == 24 ? 0u : state[0];
  const int per = (n + TKP - 1) / TKP, b0 = r0 + blockIdx.x * per, b1 = min(r1, b0 + per);
  const int lane = threadIdx.x & 63;
  for (int base = b0; base < b1; base += 256) {         // wave-aggregated histogram (see k_topk_mask)
    const int i = base + threadIdx.x;
    const uint32_t u = (i < b1) ? f2ord(v[i]) : 0u;
    const bool live = (i < b1) && ((u & pmask) == prefix);
    const int d = (int)((u >> shift) & 255);
    unsigned long long m = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      unsigned long long bal = __ballot((d >> b) & 1);
      m &= ((d >> b) & 1) ? bal : ~bal;
    }
    if (live && (m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&lh[d], (unsigned int)__popcll(m));
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
  if (!es_last_block_sel(tickets + seg, gridDim.x, safe)) return;      // (everything the workgroups exchange goes through atomics)
  // the segment's histogram is complete: walk the buckets from the largest digit (one thread; 256 coherent reads)
  if (threadIdx.x == 0) {
    unsigned int rem = shift == 24 ? (unsigned int)kkeep : state[1], b = 255;
    for (;; --b) {
      unsigned int h = es_coh_load_u(&hist[b]);
      if (h >= rem) break;
      rem -= h;
      if (b == 0) break;
    }
    state[0] = prefix | (b << shift);
    state[1] = rem;                                     // how many to take from the bucket equal to the threshold
  }
  __syncthreads();
  es_coh_store_u(&hist[threadIdx.x], 0u);               // ready for the next pass / the next call
}
__global__ __launch_bounds__(256) void k_topk_eqcount(const float* __restrict__ v, Segs segs, int kkeep, unsigned int* __restrict__ wsp, int safe) {
  const int seg = blockIdx.y, r0 = segs.off[seg], r1 = segs.off[seg + 1], n = r1 - r0;
  if (kkeep >= n) return;
  unsigned int* tickets = wsp;
  const uint32_t thr = wsp[TK_STATE + 2 * seg];
  unsigned int* cnt = wsp + TK_CNT + TKP * seg;
  unsigned int* basep = wsp + TK_BASE + TKP * seg;
  const int per = (n + TKP - 1) / TKP, b0 = r0 + blockIdx.x * per, b1 = min(r1, b0 + per);
  unsigned int c = 0;
  for (int i = b0 + threadIdx.x; i < b1; i += 256) c += (f2ord(v[i]) == thr) ? 1u : 0u;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  __shared__ unsigned int wc[4];
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) es_coh_store_u(&cnt[blockIdx.x], wc[0] + wc[1] + wc[2] + wc[3]);
  if (!es_last_block_sel(tickets + seg, gridDim.x, safe)) return;
  if (threadIdx.x == 0) {
    unsigned int run = 0;
    for (int b = 0; b < TKP; ++b) {
      basep[b] = run;
      run += es_coh_load_u(&cnt[b]);
    }
  }
}
__global__ __launch_bounds__(256) void k_topk_write(const float* __restrict__ v, Segs segs, int kkeep, const unsigned int* __restrict__ wsp,
                                                    int* __restrict__ mask) {
  const int seg = blockIdx.y, r0 = segs.off[seg], r1 = segs.off[seg + 1], n = r1 - r0;
  const int per = (n + TKP - 1) / TKP, b0 = r0 + blockIdx.x * per, b1 = min(r1, b0 + per);
  if (kkeep >= n) {
    for (int i = b0 + threadIdx.x; i < b1; i += 256) mask[i] = 1;
    return;
  }
  const uint32_t thr = wsp[TK_STATE + 2 * seg];
  const unsigned int take_eq = wsp[TK_STATE + 2 * seg + 1];
  __shared__ unsigned int s_carry;
  __shared__ int wsum[4];
  if (threadIdx.x == 0) s_carry = wsp[TK_BASE + TKP * seg + blockIdx.x];
  __syncthreads();
  for (int base = b0; base < b1; base += 256) {         // ordered pass: rank the ties by row
    const int i = base + threadIdx.x;
    const uint32_t u = (i < b1) ? f2ord(v[i]) : 0;
    const int eq = (i < b1) && (u == thr);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = eq;
    for (int o = 1; o < 64; o <<= 1) {
      int tt = __shfl_up(inc, o, 64);
      if (lane >= o) inc += tt;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int bs = (int)s_carry, tot = 0;
    for (int q = 0; q < 4; ++q) {
      if (q < w) bs += wsum[q];
      tot += wsum[q];
    }
    if (i < b1) mask[i] = (u > thr) || (eq && (unsigned)(bs + inc - 1) < take_eq);
    __syncthreads();
    if (threadIdx.x == 0) s_carry += tot;
    __syncthreads();
  }
}
extern "C" size_t es_topk_mask_workspace_ints(int nseg) { (void)nseg; return (size_t)TK_INTS; }     // (fixed layout, see above)
extern "C" int es_topk_mask_ws(const float* values, const int* seg_off, int nseg, int k, int* mask, int* workspace,
                               size_t workspace_ints, void* stream) {
  if (nseg <= 0 || nseg > ES_MAX_SEG) return nseg > ES_MAX_SEG ? -3 : 0;
  if (!workspace || workspace_ints < es_topk_mask_workspace_ints(nseg)) return -5;
  Segs s = make_segs(seg_off, nseg);
  hipStream_t st = (hipStream_t)stream;
  bool any = false;
  for (int i = 0; i < nseg; ++i) any = any || (seg_off[i + 1] - seg_off[i] > k);
  unsigned int* w = (unsigned int*)workspace;
  if (any) {
    uint32_t pmask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
      hipLaunchKernelGGL(k_topk_hist, dim3(TKP, nseg), dim3(256), 0, st, values, s, k, shift, pmask, w, ES_OPT_ELECT_SAFE);
      pmask |= (255u << shift);
    }
    hipLaunchKernelGGL(k_topk_eqcount, dim3(TKP, nseg), dim3(256), 0, st, values, s, k, w, ES_OPT_ELECT_SAFE);
  }
  hipLaunchKernelGGL(k_topk_write, dim3(TKP, nseg), dim3(256), 0, st, values, s, k, w, mask);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ column sums (bias gradients), deterministic, one launch
// dst[c] (+)= sum_rows g[r, c].  Rounds 1-3 ran every bias gradient as a 1 x C weight-gradient GEMM against a column of ones (an
// f32 MFMA launch + its slice reduction: 105 launch pairs moving 0.8 GB for 0.4 GFLOP in one grounding step).  Here a workgroup
// sums a chunk of rows (64, or n / 256 for long matrices: at most 256 chunks; threads over columns x 4 row stripes, fixed order), stores its partial row, and the last workgroup
// to arrive (es_last_block_light: the partial rows travel through coherent stores / loads, no cache maintenance) adds the partials
// in chunk order: no float atomics, no second launch.
#define CS_ROWS 64
static int colsum_rows(int n) { int r = es_cdiv(n > 0 ? n : 1, 256); r = (r + 3) / 4 * 4; return r < CS_ROWS ? CS_ROWS : r; }   // <= 256 chunks
// (64 chunks were too few for the head's 4e5-row launches: 64 workgroups, 0.64 ms, profiles/r4_single_stream_kernel_stats.txt)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ g, int ld, int n, int C, float* __restrict__ dst, int accumulate,
                                                float* __restrict__ ws, int rows_per_block, int safe) {
  // round 6: gridDim.y (<= 4: one ticket each) workgroups share the 64-column groups of a row chunk instead of one workgroup walking them one
  // dependent latency after the other (4 groups at 256 columns: 12.6 us per launch on the decoder's 3 072-row matrices, 117 launches per
  // grounding step), and 16 rows of a stripe are in flight instead of 8; the order of the additions -- and so every bit of the result -- is unchanged
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  float* part = ws + ES_TICKET_FLOATS + (size_t)blockIdx.x * C;
  for (int c0 = blockIdx.y * 64; c0 < C; c0 += 64 * gridDim.y) {
    const int c = c0 + tx;
    float s = 0.f;
    if (c < C)
      for (int r = r0 + ty; r < r1; r += 64) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (r + 4 * u) < r1 ? g[(size_t)(r + 4 * u) * ld + c] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
      }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) es_coh_store(part + c, (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]));
    __syncthreads();
  }
  if (!es_last_block_sel((unsigned int*)ws + blockIdx.y, gridDim.x, safe)) return;
  const int ngroups = (C + 63) / 64, mine = (ngroups - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
  for (int idx = threadIdx.x; idx < mine * 64; idx += 256) {
    const int c = ((int)blockIdx.y + (idx >> 6) * (int)gridDim.y) * 64 + (idx & 63);
    if (c < C) {
      float t = es_coh_sum(ws + ES_TICKET_FLOATS + c, (int)gridDim.x, (size_t)C);
      dst[c] = accumulate ? dst[c] + t : t;
    }
  }
}
extern "C" size_t es_colsum_workspace_floats(int n, int C) { return (size_t)ES_TICKET_FLOATS + (size_t)es_cdiv(n > 0 ? n : 1, colsum_rows(n)) * C; }
extern "C" int es_colsum(const float* g, int ld, int n, int C, float* dst, int accumulate, float* workspace, size_t workspace_floats,
                         void* stream) {
  if (C <= 0) return 0;
  if (n <= 0) { if (!accumulate) ES_TRY(hipMemsetAsync(dst, 0, (size_t)C * 4, (hipStream_t)stream)); return 0; }
  if (!workspace || workspace_floats < es_colsum_workspace_floats(n, C)) return -5;
  const int rpb = colsum_rows(n);
  const int ng = es_cdiv(C, 64);
  hipLaunchKernelGGL(k_colsum, dim3(es_cdiv(n, rpb), ng < 4 ? ng : 4), dim3(256), 0, (hipStream_t)stream, g, ld, n, C, dst, accumulate, workspace, rpb, ES_OPT_ELECT_SAFE);
  ES_CHECK_LAUNCH();
  return 0;
}

// row-wise max over channels (prune score = max class logit, fcaf3d_head.py:1131-1134)
__global__ void k_row_max(const float* __restrict__ x, int ldx, int n, int C, float* __restrict__ out) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[(size_t)row * ldx + c]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) out[row] = m;
}
extern "C" int es_row_max(const float* x, int ldx, int n, int C, float* out, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_row_max, dim3(es_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, n, C, out);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ frozen-BN affine + residual + ReLU (2-D backbone)
// y = act(x * scale[c] + shift[c] (+ res));  mmdet.ResNet with norm_eval=True, BN requires_grad=False
// (configs/detection/mv-det3d_...py:24-34): scale = w / sqrt(var + eps), shift = b - mean * scale.
__global__ void k_affine_act(const float* __restrict__ x, const float* __restrict__ scale,
                             const float* __restrict__ shift, const float* __restrict__ res, size_t n, int C, int act,
                             float* __restrict__ y) {
  size_t tot = n * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(e % C);
    float z = x[e] * scale[c] + shift[c];
    if (res) z += res[e];
    y[e] = act == 1 ? fmaxf(z, 0.f) : (act == 2 ? (z > 0.f ? z : expf(z) - 1.f) : z);
  }
}
extern "C" int es_affine_act_fwd(const float* x, const float* scale, const float* shift, const float* res, size_t n,
                                 int C, int act, float* y, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_affine_act, dim3(8192), dim3(256), 0, (hipStream_t)stream, x, scale, shift, res, n, C, act, y);
  ES_CHECK_LAUNCH();
  return 0;
}
// dz = dy * relu'(y) ; dx (op)= dz * scale ; dres (op)= dz
__global__ void k_affine_act_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                 const float* __restrict__ scale, size_t n, int C, int act, float* __restrict__ dx,
                                 int acc_x, float* __restrict__ dres, int acc_r) {
  size_t tot = n * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(e % C);
    float g = dy[e];
    if (act && !(y[e] > 0.f)) g = 0.f;
    if (dx) dx[e] = acc_x ? dx[e] + g * scale[c] : g * scale[c];
    if (dres) dres[e] = acc_r ? dres[e] + g : g;
  }
}
// C % 4 == 0 and 16-byte aligned pointers: four channels per thread
__global__ void k_affine_act_bwd4(const float4* __restrict__ dy, const float4* __restrict__ y,
                                  const float4* __restrict__ scale, size_t n4, int C4, int act,
                                  float4* __restrict__ dx, int acc_x, float4* __restrict__ dres, int acc_r) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
    float4 g = dy[e];
    if (act) {
      float4 v = y[e];
      if (!(v.x > 0.f)) g.x = 0.f;
      if (!(v.y > 0.f)) g.y = 0.f;
      if (!(v.z > 0.f)) g.z = 0.f;
      if (!(v.w > 0.f)) g.w = 0.f;
    }
    if (dx) {
      float4 s = scale[e % C4];
      float4 o = {g.x * s.x, g.y * s.y, g.z * s.z, g.w * s.w};
      if (acc_x) { float4 p = dx[e]; o.x = p.x + o.x; o.y = p.y + o.y; o.z = p.z + o.z; o.w = p.w + o.w; }
      dx[e] = o;
    }
    if (dres) {
      float4 o = g;
      if (acc_r) { float4 p = dres[e]; o.x = p.x + g.x; o.y = p.y + g.y; o.z = p.z + g.z; o.w = p.w + g.w; }
      dres[e] = o;
    }
  }
}
// the same with the activation y stored in bf16 (image backbone, round 3): only its sign is read
__global__ void k_affine_act_bwd4_yh(const float4* __restrict__ dy, const uint2* __restrict__ y,
                                     const float4* __restrict__ scale, size_t n4, int C4, int act,
                                     float4* __restrict__ dx, int acc_x, float4* __restrict__ dres, int acc_r) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
    float4 g = dy[e];
    if (act) {
      uint2 v = y[e];                                    // 4 bf16: positive <=> sign bit clear and magnitude non-zero
      if (!(__uint_as_float(v.x << 16) > 0.f)) g.x = 0.f;
      if (!(__uint_as_float(v.x & 0xffff0000u) > 0.f)) g.y = 0.f;
      if (!(__uint_as_float(v.y << 16) > 0.f)) g.z = 0.f;
      if (!(__uint_as_float(v.y & 0xffff0000u) > 0.f)) g.w = 0.f;
    }
    if (dx) {
      float4 s = scale[e % C4];
      float4 o = {g.x * s.x, g.y * s.y, g.z * s.z, g.w * s.w};
      if (acc_x) { float4 p = dx[e]; o.x = p.x + o.x; o.y = p.y + o.y; o.z = p.z + o.z; o.w = p.w + o.w; }
      dx[e] = o;
    }
    if (dres) {
      float4 o = g;
      if (acc_r) { float4 p = dres[e]; o.x = p.x + g.x; o.y = p.y + g.y; o.z = p.z + g.z; o.w = p.w + g.w; }
      dres[e] = o;
    }
  }
}
extern "C" int es_affine_act_bwd_yh(const float* dy, const void* y_bf16, const float* scale, size_t n, int C, int act,
                                    float* dx, int acc_x, float* dres, int acc_r, void* stream) {
  if (n == 0) return 0;
  if ((C % 4) || ((((uintptr_t)dy) | ((uintptr_t)scale) | ((uintptr_t)dx) | ((uintptr_t)dres)) & 15) || (((uintptr_t)y_bf16) & 7))
    return -7;
  size_t n4 = n * (size_t)(C / 4);
  int g = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(k_affine_act_bwd4_yh, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float4*)dy, (const uint2*)y_bf16,
                     (const float4*)scale, n4, C / 4, act, (float4*)dx, acc_x, (float4*)dres, acc_r);
  ES_CHECK_LAUNCH();
  return 0;
}
extern "C" int es_affine_act_bwd(const float* dy, const float* y, const float* scale, size_t n, int C, int act,
                                 float* dx, int acc_x, float* dres, int acc_r, void* stream) {
  if (n == 0) return 0;
  bool v4 = (C % 4 == 0) && !((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)scale) | ((uintptr_t)dx) |
                               ((uintptr_t)dres)) & 15);
  if (v4) {
    size_t n4 = n * (size_t)(C / 4);
    int g = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_affine_act_bwd4, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float4*)dy,
                       (const float4*)y, (const float4*)scale, n4, C / 4, act, (float4*)dx, acc_x, (float4*)dres, acc_r);
  } else {
    hipLaunchKernelGGL(k_affine_act_bwd, dim3(8192), dim3(256), 0, (hipStream_t)stream, dy, y, scale, n, C, act, dx,
                       acc_x, dres, acc_r);
  }
  ES_CHECK_LAUNCH();
  return 0;
}
// scale/shift of a frozen BN from its four vectors
__global__ void k_bn_fold(const float* w, const float* b, const float* rm, const float* rv, int C, float eps,
                          float* scale, float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = w[c] / sqrtf(rv[c] + eps);
  scale[c] = s;
  shift[c] = b[c] - rm[c] * s;
}
extern "C" int es_bn_fold(const float* w, const float* b, const float* rm, const float* rv, int C, float eps,
                          float* scale, float* shift, void* stream) {
  hipLaunchKernelGGL(k_bn_fold, dim3(es_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, w, b, rm, rv, C, eps, scale,
                     shift);
  ES_CHECK_LAUNCH();
  return 0;
}
// dense image-grid kernel map for a KHxKW conv (stride s, pad p) over NI images in channels-last row order
__global__ void k_image_map(int NI, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad,
                            int* __restrict__ nbr) {
  long long tot = (long long)NI * Ho * Wo * KH * KW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
    int k = (int)(e % (KH * KW));
    long long j = e / (KH * KW);
    int wo = (int)(j % Wo), ho = (int)((j / Wo) % Ho), im = (int)(j / ((long long)Wo * Ho));
    int hi = ho * stride - pad + k / KW, wi = wo * stride - pad + k % KW;
    nbr[e] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? (int)(((long long)im * H + hi) * W + wi) : -1;
  }
}
extern "C" int es_image_map(int n_img, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int* nbr,
                            void* stream) {
  hipLaunchKernelGGL(k_image_map, dim3(4096), dim3(256), 0, (hipStream_t)stream, n_img, H, W, Ho, Wo, KH, KW, stride,
                     pad, nbr);
  ES_CHECK_LAUNCH();
  return 0;
}
