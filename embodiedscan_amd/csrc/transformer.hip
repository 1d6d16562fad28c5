// Grounding-transformer kernels (BASELINE config 4): multi-head attention forward / backward on the
// matrix cores, LayerNorm (+ residual), ReLU, the contrastive text-visual logits and the 9-DoF "baseline" box coder.
//
// Replaces, for embodiedscan/models/layers/ground_transformer/decoder.py:103-179,224-297:
//   mmcv MultiheadAttention -> torch.nn.MultiheadAttention (scaled dot-product core)   -> es_attn_fwd / es_attn_bwd
//   nn.LayerNorm (+ the residual add of mmcv's MultiheadAttention / FFN wrappers)       -> es_layernorm_fwd / _bwd
//   ContrastiveEmbed (dense_heads/grounding_head.py:20-99)                              -> es_contrastive_fwd / _bwd
//   GroundingHead._bbox_pred_to_bbox, box_coder='baseline', 9 outputs (:267-296)        -> es_ground_decode_fwd / _bwd
// The projections (in_proj / out_proj / FFN / reg branch) are row GEMMs on the convolution engine (K = 1).
//
// Attention: head_dim is fixed to 32 (embed 256 / 8 heads, configs/grounding/...py:48-61) = exactly the reduction depth
// of one v_mfma_f32_16x16x32_bf16, so a 16x16 score tile is ONE matrix instruction.  A workgroup (4 waves) owns 64 query
// rows (forward, dQ) or 64 key rows (dK / dV) of one (sample, head); operand tiles are staged k-contiguous in LDS, every
// GEMM of the forward and backward pass (QK^T, PV, dO V^T, dS K, P^T dO, dS^T Q) is the same "16x16 tile = sum_k A[m][k]
// B[n][k]" primitive.  Online softmax in f32; probabilities are rounded to bf16 only as MFMA operands.  BF = false
// selects the exact-f32 matrix instruction (v_mfma_f32_16x16x4_f32) for the f32 parity mode.  Keys are masked by a
// per-sample valid length (the reference's key_padding_mask is always a prefix mask: padded texts / padded point sets).
#include "common.h"
#include "../../include/es_hip.h"
extern int ES_OPT_ELECT_SAFE;           // rowops.hip (es_set_option key 18): fenced last-workgroup elections

typedef __bf16 tbf16x8_t __attribute__((ext_vector_type(8)));
typedef float tf32x4_t __attribute__((ext_vector_type(4)));

#define AT_D 32          // head dim
#define AT_R 64          // rows owned by a workgroup
#define AT_S 32          // rows of the streamed operand per step

template <bool BF> struct AtT;
template <> struct AtT<true> { typedef unsigned short T; static constexpr int LD = 40; };
template <> struct AtT<false> { typedef float T; static constexpr int LD = 33; };

__device__ inline unsigned short f2bf(float f) {          // round to nearest even
  unsigned int u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
template <bool BF> __device__ inline typename AtT<BF>::T at_cvt(float f);
template <> __device__ inline unsigned short at_cvt<true>(float f) { return f2bf(f); }
template <> __device__ inline float at_cvt<false>(float f) { return f; }

// acc (16x16, D layout: row = kq*4 + r, col = li) += sum_{k<32} A[li][k] * B[li][k]; arow / brow point at row li of the
// 16-row A / B sub-tiles (k-contiguous)
template <bool BF>
__device__ inline tf32x4_t tile_mma(const typename AtT<BF>::T* arow, const typename AtT<BF>::T* brow, int kq, tf32x4_t acc) {
  if constexpr (BF) {
    tbf16x8_t a = *(const tbf16x8_t*)(arow + kq * 8);
    tbf16x8_t b = *(const tbf16x8_t*)(brow + kq * 8);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * s + kq], brow[4 * s + kq], acc, 0, 0, 0);
    return acc;
  }
}

// stage `rows` x 32 floats (global rows r0.., leading dim ld, column offset applied by the caller) into LDS [rows][LD],
// rows >= n_valid zero-filled, values multiplied by `mul`
template <bool BF>
__device__ inline void stage_rows(typename AtT<BF>::T* dst, const float* __restrict__ src, int ld, int r0, int n_valid,
                                  int rows, float mul) {
  constexpr int LD = AtT<BF>::LD;
  for (int e = threadIdx.x; e < rows * 8; e += 256) {
    int r = e >> 3, c = (e & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < n_valid) v = *(const float4*)(src + (size_t)(r0 + r) * ld + c);
    typename AtT<BF>::T* d = dst + r * LD + c;
    d[0] = at_cvt<BF>(v.x * mul); d[1] = at_cvt<BF>(v.y * mul); d[2] = at_cvt<BF>(v.z * mul); d[3] = at_cvt<BF>(v.w * mul);
  }
}
// the same tile transposed: dst [32][LD] with dst[c][r] = src[r0 + r][c]   (rows <= 32)
template <bool BF>
__device__ inline void stage_rows_t(typename AtT<BF>::T* dst, const float* __restrict__ src, int ld, int r0, int n_valid,
                                    int rows, float mul) {
  constexpr int LD = AtT<BF>::LD;
  for (int e = threadIdx.x; e < rows * 8; e += 256) {
    int r = e >> 3, c = (e & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < n_valid) v = *(const float4*)(src + (size_t)(r0 + r) * ld + c);
    dst[(c + 0) * LD + r] = at_cvt<BF>(v.x * mul);
    dst[(c + 1) * LD + r] = at_cvt<BF>(v.y * mul);
    dst[(c + 2) * LD + r] = at_cvt<BF>(v.z * mul);
    dst[(c + 3) * LD + r] = at_cvt<BF>(v.w * mul);
  }
}

__device__ inline float group16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------ forward
// grid (ceil(Lq/64), H, B).  Q/K/V/O: row (b*L + i), columns h*32.. of matrices with leading dims ldq/ldk/ldv/ldo.
template <bool BF>
__global__ __launch_bounds__(256) void k_attn_fwd(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                  const float* __restrict__ V, int ldv, int Lq, int Lk,
                                                  const int* __restrict__ klen, float scale, float* __restrict__ O, int ldo,
                                                  float* __restrict__ lse, int H) {
  typedef typename AtT<BF>::T T;
  constexpr int LD = AtT<BF>::LD;
  __shared__ __attribute__((aligned(16))) T Qs[AT_R * LD], Ks[AT_S * LD], Vt[AT_D * LD], Ps[AT_R * LD];
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AT_R;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
  const int kvalid = klen ? min(klen[b], Lk) : Lk;
  const float* Qb = Q + (size_t)b * Lq * ldq + h * AT_D;
  const float* Kb = K + (size_t)b * Lk * ldk + h * AT_D;
  const float* Vb = V + (size_t)b * Lk * ldv + h * AT_D;
  stage_rows<BF>(Qs, Qb, ldq, q0, Lq, AT_R, scale);
  float m[4], l[4];
  tf32x4_t o[2];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
  o[0] = o[1] = (tf32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < kvalid; k0 += AT_S) {
    __syncthreads();                                   // previous step's readers of Ks / Vt / Ps are done (and Qs is staged)
    stage_rows<BF>(Ks, Kb, ldk, k0, kvalid, AT_S, 1.f);
    stage_rows_t<BF>(Vt, Vb, ldv, k0, kvalid, AT_S, 1.f);
    __syncthreads();
    tf32x4_t s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s[t] = (tf32x4_t){0.f, 0.f, 0.f, 0.f};
      s[t] = tile_mma<BF>(Qs + (wv * 16 + li) * LD, Ks + (t * 16 + li) * LD, kq, s[t]);
    }
    const bool ok0 = (k0 + li) < kvalid, ok1 = (k0 + 16 + li) < kvalid;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s0 = ok0 ? s[0][r] : -INFINITY, s1 = ok1 ? s[1][r] : -INFINITY;
      float mx = group16_max(fmaxf(s0, s1));
      float mn = fmaxf(m[r], mx);                      // finite: every step has at least one valid key
      float corr = __expf(m[r] - mn);
      float p0 = __expf(s0 - mn), p1 = __expf(s1 - mn);
      l[r] = l[r] * corr + group16_sum(p0 + p1);
      m[r] = mn;
      o[0][r] *= corr;
      o[1][r] *= corr;
      T* prow = Ps + (wv * 16 + kq * 4 + r) * LD;
      prow[li] = at_cvt<BF>(p0);
      prow[16 + li] = at_cvt<BF>(p1);
    }
    __syncthreads();
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) o[nf] = tile_mma<BF>(Ps + (wv * 16 + li) * LD, Vt + (nf * 16 + li) * LD, kq, o[nf]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int q = q0 + wv * 16 + kq * 4 + r;
    if (q >= Lq) continue;
    float inv = 1.f / l[r];
    float* orow = O + ((size_t)b * Lq + q) * ldo + h * AT_D;
    orow[li] = o[0][r] * inv;
    orow[16 + li] = o[1][r] * inv;
    if (li == 0) lse[((size_t)b * H + h) * Lq + q] = m[r] + __logf(l[r]);
  }
}

// delta[b,h,q] = sum_d dO[q, h*32 + d] * O[q, h*32 + d]
__global__ void k_attn_delta(const float* __restrict__ O, int ldo, const float* __restrict__ dO, int ldd, int B, int H, int Lq,
                             float* __restrict__ delta) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H * Lq) return;
  int q = (int)(i % Lq), h = (int)((i / Lq) % H), b = (int)(i / ((long long)Lq * H));
  const float* o = O + ((size_t)b * Lq + q) * ldo + h * AT_D;
  const float* g = dO + ((size_t)b * Lq + q) * ldd + h * AT_D;
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < AT_D; d += 4) {
    float4 a = *(const float4*)(o + d), c = *(const float4*)(g + d);
    s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
  delta[i] = s;
}

// ------------------------------------------------------------------ backward: dQ
template <bool BF>
__global__ __launch_bounds__(256) void k_attn_bwd_dq(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                     const float* __restrict__ V, int ldv, const float* __restrict__ dO, int ldd,
                                                     const float* __restrict__ lse, const float* __restrict__ delta, int Lq,
                                                     int Lk, const int* __restrict__ klen, float scale, float* __restrict__ dQ,
                                                     int ldg, int accumulate, int H) {
  typedef typename AtT<BF>::T T;
  constexpr int LD = AtT<BF>::LD;
  __shared__ __attribute__((aligned(16))) T Qs[AT_R * LD], dOs[AT_R * LD], Ks[AT_S * LD], Vs[AT_S * LD], Kt[AT_D * LD], dSs[AT_R * LD];
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AT_R;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
  const int kvalid = klen ? min(klen[b], Lk) : Lk;
  const float* Kb = K + (size_t)b * Lk * ldk + h * AT_D;
  const float* Vb = V + (size_t)b * Lk * ldv + h * AT_D;
  stage_rows<BF>(Qs, Q + (size_t)b * Lq * ldq + h * AT_D, ldq, q0, Lq, AT_R, scale);
  stage_rows<BF>(dOs, dO + (size_t)b * Lq * ldd + h * AT_D, ldd, q0, Lq, AT_R, 1.f);
  float ls[4], dl[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int q = q0 + wv * 16 + kq * 4 + r;
    bool in = q < Lq;
    ls[r] = in ? lse[((size_t)b * H + h) * Lq + q] : 0.f;
    dl[r] = in ? delta[((size_t)b * H + h) * Lq + q] : 0.f;
  }
  tf32x4_t dq[2];
  dq[0] = dq[1] = (tf32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < kvalid; k0 += AT_S) {
    __syncthreads();
    stage_rows<BF>(Ks, Kb, ldk, k0, kvalid, AT_S, 1.f);
    stage_rows<BF>(Vs, Vb, ldv, k0, kvalid, AT_S, 1.f);
    stage_rows_t<BF>(Kt, Kb, ldk, k0, kvalid, AT_S, 1.f);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      tf32x4_t s = (tf32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
      s = tile_mma<BF>(Qs + (wv * 16 + li) * LD, Ks + (t * 16 + li) * LD, kq, s);
      dp = tile_mma<BF>(dOs + (wv * 16 + li) * LD, Vs + (t * 16 + li) * LD, kq, dp);
      const bool ok = (k0 + t * 16 + li) < kvalid;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = ok ? __expf(s[r] - ls[r]) : 0.f;
        dSs[(wv * 16 + kq * 4 + r) * LD + t * 16 + li] = at_cvt<BF>(p * (dp[r] - dl[r]));
      }
    }
    __syncthreads();
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) dq[nf] = tile_mma<BF>(dSs + (wv * 16 + li) * LD, Kt + (nf * 16 + li) * LD, kq, dq[nf]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int q = q0 + wv * 16 + kq * 4 + r;
    if (q >= Lq) continue;
    float* g = dQ + ((size_t)b * Lq + q) * ldg + h * AT_D;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      float v = dq[nf][r] * scale;
      g[nf * 16 + li] = accumulate ? g[nf * 16 + li] + v : v;
    }
  }
}

// ------------------------------------------------------------------ backward: dK, dV   (grid (ceil(Lk/64), H, B))
template <bool BF>
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                      const float* __restrict__ V, int ldv, const float* __restrict__ dO,
                                                      int ldd, const float* __restrict__ lse, const float* __restrict__ delta,
                                                      int Lq, int Lk, const int* __restrict__ klen, float scale,
                                                      float* __restrict__ dK, int ldgk, float* __restrict__ dV, int ldgv,
                                                      int accumulate, int H) {
  typedef typename AtT<BF>::T T;
  constexpr int LD = AtT<BF>::LD;
  __shared__ __attribute__((aligned(16))) T Ks[AT_R * LD], Vs[AT_R * LD], Qs[AT_S * LD], dOs[AT_S * LD], Qt[AT_D * LD], dOt[AT_D * LD],
      PTs[AT_R * LD], dSTs[AT_R * LD];
  __shared__ float lseS[AT_S], delS[AT_S];
  const int b = blockIdx.z, h = blockIdx.y, kbase = blockIdx.x * AT_R;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
  const int kvalid = klen ? min(klen[b], Lk) : Lk;
  const float* Qb = Q + (size_t)b * Lq * ldq + h * AT_D;
  const float* dOb = dO + (size_t)b * Lq * ldd + h * AT_D;
  tf32x4_t dk[2], dv[2];
  dk[0] = dk[1] = dv[0] = dv[1] = (tf32x4_t){0.f, 0.f, 0.f, 0.f};
  if (kbase < kvalid) {                                 // workgroup-uniform: tiles made of padding only skip the loop
    stage_rows<BF>(Ks, K + (size_t)b * Lk * ldk + h * AT_D, ldk, kbase, kvalid, AT_R, 1.f);
    stage_rows<BF>(Vs, V + (size_t)b * Lk * ldv + h * AT_D, ldv, kbase, kvalid, AT_R, 1.f);
    for (int q0 = 0; q0 < Lq; q0 += AT_S) {
      __syncthreads();
      stage_rows<BF>(Qs, Qb, ldq, q0, Lq, AT_S, scale);
      stage_rows<BF>(dOs, dOb, ldd, q0, Lq, AT_S, 1.f);
      stage_rows_t<BF>(Qt, Qb, ldq, q0, Lq, AT_S, scale);
      stage_rows_t<BF>(dOt, dOb, ldd, q0, Lq, AT_S, 1.f);
      if (threadIdx.x < AT_S) {
        int q = q0 + threadIdx.x;
        lseS[threadIdx.x] = q < Lq ? lse[((size_t)b * H + h) * Lq + q] : INFINITY;        // exp(s - inf) = 0 for padding rows
        delS[threadIdx.x] = q < Lq ? delta[((size_t)b * H + h) * Lq + q] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 2; ++t) {                      // S^T / dP^T tiles: rows = this wave's 16 keys, cols = 16 queries
        tf32x4_t st = (tf32x4_t){0.f, 0.f, 0.f, 0.f}, dpt = st;
        st = tile_mma<BF>(Ks + (wv * 16 + li) * LD, Qs + (t * 16 + li) * LD, kq, st);
        dpt = tile_mma<BF>(Vs + (wv * 16 + li) * LD, dOs + (t * 16 + li) * LD, kq, dpt);
        const float lq = lseS[t * 16 + li], dq_ = delS[t * 16 + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bool ok = (kbase + wv * 16 + kq * 4 + r) < kvalid;
          float p = ok ? __expf(st[r] - lq) : 0.f;
          PTs[(wv * 16 + kq * 4 + r) * LD + t * 16 + li] = at_cvt<BF>(p);
          dSTs[(wv * 16 + kq * 4 + r) * LD + t * 16 + li] = at_cvt<BF>(p * (dpt[r] - dq_));
        }
      }
      __syncthreads();
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        dv[nf] = tile_mma<BF>(PTs + (wv * 16 + li) * LD, dOt + (nf * 16 + li) * LD, kq, dv[nf]);
        dk[nf] = tile_mma<BF>(dSTs + (wv * 16 + li) * LD, Qt + (nf * 16 + li) * LD, kq, dk[nf]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int k = kbase + wv * 16 + kq * 4 + r;
    if (k >= Lk) continue;                               // padded keys (k >= kvalid) get exact zeros
    float* gk = dK + ((size_t)b * Lk + k) * ldgk + h * AT_D;
    float* gv = dV + ((size_t)b * Lk + k) * ldgv + h * AT_D;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      gk[nf * 16 + li] = accumulate ? gk[nf * 16 + li] + dk[nf][r] : dk[nf][r];     // Qs / Qt already carry `scale`
      gv[nf * 16 + li] = accumulate ? gv[nf * 16 + li] + dv[nf][r] : dv[nf][r];
    }
  }
}

extern "C" int es_attn_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int B, int H, int Lq,
                           int Lk, const int* klen_dev, float* O, int ldo, float* lse, int bf16, void* stream) {
  if (B <= 0 || Lq <= 0 || Lk <= 0) return 0;
  if ((ldq | ldk | ldv | ldo) & 3) return -3;
  dim3 grid(es_cdiv(Lq, AT_R), H, B);
  const float scale = 0.17677669529663687f;              // 1 / sqrt(32)
  if (bf16)
    hipLaunchKernelGGL(k_attn_fwd<true>, grid, dim3(256), 0, (hipStream_t)stream, Q, ldq, K, ldk, V, ldv, Lq, Lk, klen_dev,
                       scale, O, ldo, lse, H);
  else
    hipLaunchKernelGGL(k_attn_fwd<false>, grid, dim3(256), 0, (hipStream_t)stream, Q, ldq, K, ldk, V, ldv, Lq, Lk, klen_dev,
                       scale, O, ldo, lse, H);
  ES_CHECK_LAUNCH();
  return 0;
}

extern "C" int es_attn_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                           const float* dO, int ldd, const float* lse, int B, int H, int Lq, int Lk, const int* klen_dev,
                           float* delta_scratch, float* dQ, int ldgq, float* dK, int ldgk, float* dV, int ldgv, int accumulate,
                           int bf16, void* stream) {
  if (B <= 0 || Lq <= 0 || Lk <= 0) return 0;
  if ((ldq | ldk | ldv | ldo | ldd | ldgq | ldgk | ldgv) & 3) return -3;
  hipStream_t st = (hipStream_t)stream;
  const float scale = 0.17677669529663687f;
  hipLaunchKernelGGL(k_attn_delta, dim3(es_cdiv((long long)B * H * Lq, 256)), dim3(256), 0, st, O, ldo, dO, ldd, B, H, Lq,
                     delta_scratch);
  dim3 gq(es_cdiv(Lq, AT_R), H, B), gk(es_cdiv(Lk, AT_R), H, B);
  if (bf16) {
    hipLaunchKernelGGL(k_attn_bwd_dq<true>, gq, dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, dO, ldd, lse, delta_scratch, Lq, Lk,
                       klen_dev, scale, dQ, ldgq, accumulate, H);
    hipLaunchKernelGGL(k_attn_bwd_dkv<true>, gk, dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, dO, ldd, lse, delta_scratch, Lq, Lk,
                       klen_dev, scale, dK, ldgk, dV, ldgv, accumulate, H);
  } else {
    hipLaunchKernelGGL(k_attn_bwd_dq<false>, gq, dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, dO, ldd, lse, delta_scratch, Lq, Lk,
                       klen_dev, scale, dQ, ldgq, accumulate, H);
    hipLaunchKernelGGL(k_attn_bwd_dkv<false>, gk, dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, dO, ldd, lse, delta_scratch, Lq, Lk,
                       klen_dev, scale, dK, ldgk, dV, ldgv, accumulate, H);
  }
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ LayerNorm over the channel dim of (n, C) rows
// z = x (+ res); y = (z - mean) * rstd * w + b.  One wave per row, C <= 512.  z is written out when res != NULL (backward
// needs the normalised input); mean / rstd saved per row.
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ x, const float* __restrict__ res, int n, int C,
                                                const float* __restrict__ w, const float* __restrict__ bia, float eps,
                                                float* __restrict__ y, float* __restrict__ z, float* __restrict__ mean,
                                                float* __restrict__ rstd) {
  int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  float v[8];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    int c = lane + q * 64;
    v[q] = 0.f;
    if (c < C) {
      v[q] = x[(size_t)i * C + c] + (res ? res[(size_t)i * C + c] : 0.f);
      s += v[q];
    }
  }
  float mu = es_wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    int c = lane + q * 64;
    if (c < C) { float d = v[q] - mu; ss += d * d; }
  }
  float rs = rsqrtf(es_wave_sum(ss) / (float)C + eps);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    int c = lane + q * 64;
    if (c < C) {
      y[(size_t)i * C + c] = (v[q] - mu) * rs * w[c] + bia[c];
      if (z) z[(size_t)i * C + c] = v[q];
    }
  }
  if (lane == 0) { mean[i] = mu; rstd[i] = rs; }
}
extern "C" int es_layernorm_fwd(const float* x, const float* res, int n, int C, const float* w, const float* b, float eps,
                                float* y, float* z, float* mean, float* rstd, void* stream) {
  if (n <= 0) return 0;
  if (C > 512) return -4;
  hipLaunchKernelGGL(k_ln_fwd, dim3(es_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, res, n, C, w, b, eps, y, z, mean, rstd);
  ES_CHECK_LAUNCH();
  return 0;
}
// dz = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)), g = dy * w;  dw += sum_rows dy * xhat, db += sum_rows dy
// Parameter gradients (round 4, deterministic): every workgroup stores the partial sums of its slice of rows in the workspace,
// the last workgroup to arrive (es_last_block_light, common.h) adds them in workgroup order into dw / db -- one writer, fixed order, no float
// atomics (rounds 2-3 used unsafeAtomicAdd here: the grounder's gradients were reproducible to ~1e-6 only).
#define LN_ROWS_PER_BLOCK 32
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ z, int n, int C,
                                                const float* __restrict__ w, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, float* __restrict__ dz, int accumulate,
                                                float* __restrict__ dw, float* __restrict__ db, int rows_per_block,
                                                float* __restrict__ ws, int safe) {
  __shared__ float sw[4][512], sb[4][512];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float aw[8], ab[8], wc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    int c = lane + q * 64;
    aw[q] = ab[q] = 0.f;
    wc[q] = c < C ? w[c] : 0.f;
  }
  int r_end = min(n, (int)(blockIdx.x + 1) * rows_per_block);
  // two rows of this wave per iteration (i and i + 4): their loads are independent, so one memory latency serves both (round 6: a wave walked
  // its 8 rows one dependent latency at a time -- 32 us per launch on the decoder's 3 072 x 256 matrices); sums accumulate in the old row order
  for (int i0 = blockIdx.x * rows_per_block + wv; i0 < r_end; i0 += 8) {
    float d[2][8], zz[2][8], mu[2], rs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = i0 + 4 * h;
      const bool ok = i < r_end;
      mu[h] = ok ? mean[i] : 0.f;
      rs[h] = ok ? rstd[i] : 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = lane + q * 64;
        d[h][q] = zz[h][q] = 0.f;
        if (ok && c < C) { d[h][q] = dy[(size_t)i * C + c]; zz[h][q] = z[(size_t)i * C + c]; }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = i0 + 4 * h;
      if (i >= r_end) break;
      float g[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = lane + q * 64;
        g[q] = xh[q] = 0.f;
        if (c < C) {
          xh[q] = (zz[h][q] - mu[h]) * rs[h];
          g[q] = d[h][q] * wc[q];
          s1 += g[q];
          s2 += g[q] * xh[q];
          aw[q] += d[h][q] * xh[q];
          ab[q] += d[h][q];
        }
      }
      s1 = es_wave_sum(s1) / (float)C;
      s2 = es_wave_sum(s2) / (float)C;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = lane + q * 64;
        if (c < C) {
          const float v = rs[h] * (g[q] - s1 - xh[q] * s2);
          float* p = dz + (size_t)i * C + c;
          *p = accumulate ? *p + v : v;
        }
      }
    }
  }
  if (!dw && !db) return;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    int c = lane + q * 64;
    if (c < C) { sw[wv][c] = aw[q]; sb[wv][c] = ab[q]; }
  }
  __syncthreads();
  float* part = ws + ES_TICKET_FLOATS;                              // [block][2][C]
  for (int c = threadIdx.x; c < C; c += 256) {                       // (coherent stores / loads: see es_last_block_light)
    es_coh_store(part + ((size_t)blockIdx.x * 2) * C + c, sw[0][c] + sw[1][c] + sw[2][c] + sw[3][c]);
    es_coh_store(part + ((size_t)blockIdx.x * 2 + 1) * C + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
  }
  if (!es_last_block_sel((unsigned int*)ws, gridDim.x, safe)) return;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a, bsum;
    es_coh_sum2(part + c, part + C + c, (int)gridDim.x, (size_t)2 * C, a, bsum);
    if (dw) dw[c] += a;
    if (db) db[c] += bsum;
  }
}
extern "C" size_t es_layernorm_bwd_workspace_floats(int n, int C) {
  return (size_t)ES_TICKET_FLOATS + (size_t)es_cdiv(n > 0 ? n : 1, LN_ROWS_PER_BLOCK) * 2 * C;
}
extern "C" int es_layernorm_bwd(const float* dy, const float* z, int n, int C, const float* w, const float* mean,
                                const float* rstd, float* dz, int accumulate, float* dw, float* db, float* workspace,
                                size_t workspace_floats, void* stream) {
  if (n <= 0) return 0;
  if (C > 512) return -4;
  if ((dw || db) && (!workspace || workspace_floats < es_layernorm_bwd_workspace_floats(n, C))) return -5;
  int rpb = LN_ROWS_PER_BLOCK;
  hipLaunchKernelGGL(k_ln_bwd, dim3(es_cdiv(n, rpb)), dim3(256), 0, (hipStream_t)stream, dy, z, n, C, w, mean, rstd, dz, accumulate,
                     dw, db, rpb, workspace, ES_OPT_ELECT_SAFE);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ ReLU (in place) and its backward through the output
__global__ void k_relu_fwd(float* __restrict__ x, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) x[e] = fmaxf(x[e], 0.f);
}
__global__ void k_relu_bwd(float* __restrict__ dy, const float* __restrict__ y, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    if (!(y[e] > 0.f)) dy[e] = 0.f;
}
extern "C" int es_relu_fwd(float* x, size_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_relu_fwd, dim3(min(es_cdiv(n, 256), 4096)), dim3(256), 0, (hipStream_t)stream, x, n);
  ES_CHECK_LAUNCH();
  return 0;
}
extern "C" int es_relu_bwd(float* dy, const float* y, size_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_relu_bwd, dim3(min(es_cdiv(n, 256), 4096)), dim3(256), 0, (hipStream_t)stream, dy, y, n);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ ContrastiveEmbed (grounding_head.py:62-99)
// logits[b, i, t] = <v[b,i,:], text[b,t,:]> / sqrt(C) + bias   for t < tlen[b] (and i < vlen[b]); -inf elsewhere, up to Tmax.
// One wave per visual row; the sample's text block sits in LDS.  rowmax (optional): max_t logits (query selection,
// sparse_featfusion_grounder.py:370-376).
__global__ __launch_bounds__(256) void k_contrastive_fwd(const float* __restrict__ v, int L, const float* __restrict__ text, int T,
                                                         int C, const int* __restrict__ tlen, const int* __restrict__ vlen,
                                                         const float* __restrict__ bias, float* __restrict__ logits, int Tout,
                                                         float* __restrict__ rowmax) {
  extern __shared__ float ts[];                         // T * C
  const int b = blockIdx.y;
  const int tl = min(tlen[b], T);
  for (int e = threadIdx.x; e < tl * C; e += 256) ts[e] = text[(size_t)b * T * C + e];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float inv = 1.f / sqrtf((float)C), bv = bias ? bias[0] : 0.f;
  const int vl = vlen ? min(vlen[b], L) : L;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < L; i += gridDim.x * 4) {
    const float* vr = v + ((size_t)b * L + i) * C;
    float vv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) vv[q] = (lane + q * 64) < C ? vr[lane + q * 64] : 0.f;
    float best = -INFINITY;
    for (int t = 0; t < Tout; ++t) {
      float out = -INFINITY;
      if (t < tl && i < vl) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) if ((lane + q * 64) < C) s += vv[q] * ts[t * C + lane + q * 64];
        out = es_wave_sum(s) * inv + bv;
      }
      best = fmaxf(best, out);
      if (logits && lane == 0) logits[((size_t)b * L + i) * Tout + t] = out;
    }
    if (rowmax && lane == 0) rowmax[(size_t)b * L + i] = best;
  }
}
extern "C" int es_contrastive_fwd(const float* v, int B, int L, const float* text, int T, int C, const int* tlen_dev,
                                  const int* vlen_dev, const float* bias_dev, float* logits, int Tout, float* rowmax,
                                  void* stream) {
  if (B <= 0 || L <= 0) return 0;
  if (C > 512 || (size_t)T * C * 4 > 160 * 1024 - 1024) return -4;
  size_t sh = (size_t)T * C * sizeof(float);
  if (sh > 64 * 1024) ES_TRY(hipFuncSetAttribute((const void*)k_contrastive_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
  hipLaunchKernelGGL(k_contrastive_fwd, dim3(min(es_cdiv(L, 4), 256), B), dim3(256), sh, (hipStream_t)stream, v, L, text, T, C,
                     tlen_dev, vlen_dev, bias_dev, logits, Tout, rowmax);
  ES_CHECK_LAUNCH();
  return 0;
}
// backward: dv[b,i,:] = sum_t dl[b,i,t] text[b,t,:] / sqrt(C);  dtext[b,t,:] += sum_i dl[b,i,t] v[b,i,:] / sqrt(C);
// dbias += sum dl.  dlogits must be 0 at masked positions.  One launch, two kinds of workgroups (round 4, deterministic: no
// float atomics): blockIdx.x < nA -> one wave per visual row for dv (the sample's text block in LDS); blockIdx.x >= nA -> ONE
// workgroup per (sample, text token) walks the sample's rows in ascending order, threads over channels, and is the only
// writer of its dtext row; its sum of dl goes to the workspace and the last workgroup to arrive (es_last_block) adds those B * T
// partials in index order into dbias.
__global__ __launch_bounds__(256) void k_contrastive_bwd(const float* __restrict__ dl, int Tout, const float* __restrict__ v, int L,
                                                         const float* __restrict__ text, int T, int C,
                                                         const int* __restrict__ tlen, float* __restrict__ dv, int acc_v,
                                                         float* __restrict__ dtext, float* __restrict__ dbias, int nA,
                                                         float* __restrict__ ws, int safe) {
  extern __shared__ float sh[];                         // dv workgroups: the sample's text block [tl * C]
  const int b = blockIdx.y;
  const int tl = min(tlen[b], T);
  const float inv = 1.f / sqrtf((float)C);
  const bool reduce = (dtext != nullptr) || (dbias != nullptr);
  if ((int)blockIdx.x < nA) {
    if (dv) {
      float* ts = sh;
      for (int e = threadIdx.x; e < tl * C; e += 256) ts[e] = text[(size_t)b * T * C + e];
      __syncthreads();
      const int lane = threadIdx.x & 63;
      for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < L; i += nA * 4) {
        const float* dr = dl + ((size_t)b * L + i) * Tout;
        float g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = 0.f;
        for (int t = 0; t < tl; ++t) {
          float d = dr[t];
          if (d == 0.f) continue;
          d *= inv;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            int c = lane + q * 64;
            if (c < C) g[q] += d * ts[t * C + c];
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          int c = lane + q * 64;
          if (c < C) { float* p = dv + ((size_t)b * L + i) * C + c; *p = acc_v ? *p + g[q] : g[q]; }
        }
      }
    }
  } else if (reduce) {
    const int t = (int)blockIdx.x - nA;
    float a0 = 0.f, a1 = 0.f, bs = 0.f;
    const int c0 = threadIdx.x, c1 = threadIdx.x + 256;
    if (t < tl) {
      for (int i0 = 0; i0 < L; i0 += 8) {                  // 8 rows in flight, added in ascending row order
        float d[8], x0[8], x1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u;
          const bool in = i < L;
          d[u] = in ? dl[((size_t)b * L + i) * Tout + t] : 0.f;
          const float* vr = v + ((size_t)b * L + (in ? i : 0)) * C;
          x0[u] = (in && c0 < C) ? vr[c0] : 0.f;
          x1[u] = (in && c1 < C) ? vr[c1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (d[u] == 0.f) continue;                       // (uniform across the workgroup)
          bs += d[u];
          const float ds = d[u] * inv;
          a0 += ds * x0[u];
          a1 += ds * x1[u];
        }
      }
      if (dtext) {
        float* o = dtext + ((size_t)b * T + t) * C;
        if (c0 < C) o[c0] += a0;
        if (c1 < C) o[c1] += a1;
      }
    }
    if (threadIdx.x == 0) es_coh_store(ws + ES_TICKET_FLOATS + (size_t)b * T + t, bs);
  }
  if (!reduce) return;
  if (!es_last_block_sel((unsigned int*)ws, gridDim.x * gridDim.y, safe)) return;
  if (dbias && threadIdx.x == 0) {
    dbias[0] += es_coh_sum(ws + ES_TICKET_FLOATS, (int)(gridDim.y * T), 1);
  }
}
extern "C" size_t es_contrastive_bwd_workspace_floats(int B, int T) { return (size_t)ES_TICKET_FLOATS + (size_t)(B > 0 ? B : 1) * T; }
extern "C" int es_contrastive_bwd(const float* dlogits, int Tout, const float* v, int B, int L, const float* text, int T, int C,
                                  const int* tlen_dev, float* dv, int acc_v, float* dtext, float* dbias, float* workspace,
                                  size_t workspace_floats, void* stream) {
  if (B <= 0 || L <= 0) return 0;
  size_t sh = (size_t)T * C * sizeof(float);
  if (C > 512 || sh > 160 * 1024 - 1024) return -4;
  const bool reduce = dtext || dbias;
  if (reduce && (!workspace || workspace_floats < es_contrastive_bwd_workspace_floats(B, T))) return -5;
  if (sh > 64 * 1024) ES_TRY(hipFuncSetAttribute((const void*)k_contrastive_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
  const int nA = dv ? min(es_cdiv(L, 4), 64) : 0;
  const int nx = nA + (reduce ? T : 0);
  if (nx <= 0) return 0;
  hipLaunchKernelGGL(k_contrastive_bwd, dim3(nx, B), dim3(256), sh, (hipStream_t)stream, dlogits, Tout, v, L, text, T, C, tlen_dev,
                     dv, acc_v, dtext, dbias, nA, workspace, ES_OPT_ELECT_SAFE);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ 9-DoF "baseline" box coder (grounding_head.py:286-296)
// box = (pred[:3] + point, clamp(exp(pred[3:6]), 2e-2), pred[6:9])
__global__ void k_ground_decode_fwd(const float* __restrict__ pred, int ldp, const float* __restrict__ pts, int n,
                                    float* __restrict__ box) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = pred + (size_t)i * ldp;
  float* o = box + (size_t)i * 9;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o[c] = p[c] + pts[(size_t)i * 3 + c];
    o[3 + c] = fmaxf(expf(p[3 + c]), 2e-2f);
    o[6 + c] = p[6 + c];
  }
}
extern "C" int es_ground_decode_fwd(const float* pred, int ldp, const float* points, int n, float* box, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_ground_decode_fwd, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, ldp, points, n, box);
  ES_CHECK_LAUNCH();
  return 0;
}
__global__ void k_ground_decode_bwd(const float* __restrict__ pred, int ldp, const float* __restrict__ dbox, int n,
                                    float* __restrict__ dpred, int ldg, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = pred + (size_t)i * ldp;
  const float* g = dbox + (size_t)i * 9;
  float* o = dpred + (size_t)i * ldg;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float e = expf(p[3 + c]);
    float v0 = g[c], v1 = e > 2e-2f ? g[3 + c] * e : 0.f, v2 = g[6 + c];     // clamp(min): gradient passes where exp > min
    o[c] = accumulate ? o[c] + v0 : v0;
    o[3 + c] = accumulate ? o[3 + c] + v1 : v1;
    o[6 + c] = accumulate ? o[6 + c] + v2 : v2;
  }
}
extern "C" int es_ground_decode_bwd(const float* pred, int ldp, const float* dbox, int n, float* dpred, int ldg, int accumulate,
                                    void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_ground_decode_bwd, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, ldp, dbox, n, dpred, ldg,
                     accumulate);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ 9-DoF "FCAF" box coder (grounding_head.py:308-363;
// configs/grounding/mv-grounding_8xb12_embodiedscan-vg-9dof_fcaf-coder.py:64)
// d = clamp(exp(pred[0:6]), 2e-2) (log distances to the six faces); s = ((d1 - d0) / 2, (d3 - d2) / 2, (d5 - d4) / 2);
// box = (point + R(euler) s, (d0 + d1, d2 + d3, d4 + d5), euler), euler = pred[6:9], R = Rz(a) Rx(b) Ry(c) (rotation_3d_in_euler
// multiplies the row vector by R^T).  The reference writes d in place into bbox_pred; nothing reads bbox_pred afterwards, so this
// is a pure function of (pred, point) and autograd differentiates through exp / clamp (gradient passes where exp >= 2e-2).
struct FcafRot { float c0[3], c1[3], c2[3], sa, ca, sb, cb, sc, cc; };
__device__ inline FcafRot fcaf_rot(float a, float b, float c) {
  FcafRot r;
  r.sa = sinf(a); r.ca = cosf(a); r.sb = sinf(b); r.cb = cosf(b); r.sc = sinf(c); r.cc = cosf(c);
  r.c0[0] = r.ca * r.cc - r.sa * r.sb * r.sc; r.c0[1] = r.sa * r.cc + r.ca * r.sb * r.sc; r.c0[2] = -(r.cb * r.sc);
  r.c1[0] = -(r.sa * r.cb);                   r.c1[1] = r.ca * r.cb;                      r.c1[2] = r.sb;
  r.c2[0] = r.ca * r.sc + r.sa * r.sb * r.cc; r.c2[1] = r.sa * r.sc - r.ca * r.sb * r.cc; r.c2[2] = r.cb * r.cc;
  return r;
}
__global__ void k_ground_decode_fcaf_fwd(const float* __restrict__ pred, int ldp, const float* __restrict__ pts, int n,
                                         float* __restrict__ box) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = pred + (size_t)i * ldp;
  float* o = box + (size_t)i * 9;
  float d[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) d[j] = fmaxf(expf(p[j]), 2e-2f);
  const float s0 = (d[1] - d[0]) * 0.5f, s1 = (d[3] - d[2]) * 0.5f, s2 = (d[5] - d[4]) * 0.5f;
  const FcafRot r = fcaf_rot(p[6], p[7], p[8]);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o[c] = pts[(size_t)i * 3 + c] + (r.c0[c] * s0 + r.c1[c] * s1 + r.c2[c] * s2);
    o[3 + c] = d[2 * c] + d[2 * c + 1];
    o[6 + c] = p[6 + c];
  }
}
extern "C" int es_ground_decode_fcaf_fwd(const float* pred, int ldp, const float* points, int n, float* box, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_ground_decode_fcaf_fwd, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, ldp, points, n, box);
  ES_CHECK_LAUNCH();
  return 0;
}
__global__ void k_ground_decode_fcaf_bwd(const float* __restrict__ pred, int ldp, const float* __restrict__ dbox, int n,
                                         float* __restrict__ dpred, int ldg, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = pred + (size_t)i * ldp;
  const float* g = dbox + (size_t)i * 9;
  float* o = dpred + (size_t)i * ldg;
  float e[6], d[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) { e[j] = expf(p[j]); d[j] = fmaxf(e[j], 2e-2f); }
  const float s0 = (d[1] - d[0]) * 0.5f, s1 = (d[3] - d[2]) * 0.5f, s2 = (d[5] - d[4]) * 0.5f;
  const FcafRot r = fcaf_rot(p[6], p[7], p[8]);
  // gradient w.r.t. the shift: R^T g_center
  const float gs0 = r.c0[0] * g[0] + r.c0[1] * g[1] + r.c0[2] * g[2];
  const float gs1 = r.c1[0] * g[0] + r.c1[1] * g[1] + r.c1[2] * g[2];
  const float gs2 = r.c2[0] * g[0] + r.c2[1] * g[1] + r.c2[2] * g[2];
  const float gs[3] = {gs0, gs1, gs2};
  float out[9];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float lo = -0.5f * gs[c] + g[3 + c], hi = 0.5f * gs[c] + g[3 + c];       // d d[2c], d d[2c+1]
    out[2 * c] = e[2 * c] >= 2e-2f ? lo * e[2 * c] : 0.f;
    out[2 * c + 1] = e[2 * c + 1] >= 2e-2f ? hi * e[2 * c + 1] : 0.f;
  }
  // v = R s; dv/da = (-v.y, v.x, 0); dv/dc = col0 * s2 - col2 * s0; dv/db from the element-wise derivative of Rz Rx Ry
  const float v0 = r.c0[0] * s0 + r.c1[0] * s1 + r.c2[0] * s2, v1 = r.c0[1] * s0 + r.c1[1] * s1 + r.c2[1] * s2;
  const float da = g[0] * (-v1) + g[1] * v0;
  const float b0 = (-(r.sa * r.cb * r.sc)) * s0 + (r.sa * r.sb) * s1 + (r.sa * r.cb * r.cc) * s2;
  const float b1 = (r.ca * r.cb * r.sc) * s0 + (-(r.ca * r.sb)) * s1 + (-(r.ca * r.cb * r.cc)) * s2;
  const float b2 = (r.sb * r.sc) * s0 + r.cb * s1 + (-(r.sb * r.cc)) * s2;
  const float db = g[0] * b0 + g[1] * b1 + g[2] * b2;
  const float dc = g[0] * (r.c0[0] * s2 - r.c2[0] * s0) + g[1] * (r.c0[1] * s2 - r.c2[1] * s0) + g[2] * (r.c0[2] * s2 - r.c2[2] * s0);
  out[6] = g[6] + da; out[7] = g[7] + db; out[8] = g[8] + dc;
#pragma unroll
  for (int j = 0; j < 9; ++j) o[j] = accumulate ? o[j] + out[j] : out[j];
}
extern "C" int es_ground_decode_fcaf_bwd(const float* pred, int ldp, const float* dbox, int n, float* dpred, int ldg, int accumulate,
                                         void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_ground_decode_fcaf_bwd, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, ldp, dbox, n, dpred,
                     ldg, accumulate);
  ES_CHECK_LAUNCH();
  return 0;
}
