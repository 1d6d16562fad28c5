// Sparse (and, through an identity map, dense-row) convolution engine for gfx950.
//
// Output-stationary implicit GEMM on the f32 matrix cores (v_mfma_f32_16x16x4_f32:
// exact f32, k-ordered fmaf chain): a workgroup owns 128 output rows x 64 output
// channels, walks the (tap, C_in-chunk) sequence, gathers the neighbour rows named by
// the kernel map into LDS (absent neighbours = 0), streams the weight slice into LDS
// and accumulates in registers.  Every output element is written exactly once -> no
// atomics, deterministic, one launch per convolution.
//   forward :  Y[j]  = sum_k X[nbr[j,k]]  . W[k]        (W  [K][Cin][Cout])
//   dgrad   :  dX[i] = sum_k dY[inv[i,k]] . W[k]^T      (same kernel, TRANS_W)
//   wgrad   :  dW[k] = sum_j X[nbr[j,k]]^T . dY[j]      (split over rows, f32 atomics)
// Replaces MinkowskiConvolution / MinkowskiGenerativeConvolutionTranspose /
// kernel_size=1 matmuls at embodiedscan/models/backbones/mink_resnet.py:58-62,88-120 and
// embodiedscan/models/dense_heads/fcaf3d_head.py:907-984.
#include "common.h"
#include "../../include/es_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 64
#define BK 16
#define LDB (BN + 16)
#define MAXK 27

__device__ inline int a_swz(int row_k) { return (((row_k >> 2) & 3) << 3) ^ ((row_k & 1) << 4); }

template <bool TRANS_W>
__global__ __launch_bounds__(256) void k_spconv(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                const int* __restrict__ nbr, int n_out, int n_in, int K, int Cin,
                                                int Cout, const float* __restrict__ bias, float* __restrict__ Y,
                                                int ldy, int accumulate) {
  __shared__ float As[BK * BM];
  __shared__ float Bs[BK * LDB];
  __shared__ int nbrS[BM * MAXK];
  __shared__ int tapAny[32];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int row0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const bool vecA = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const bool vecB = TRANS_W ? (((Cin & 3) == 0) && ((((uintptr_t)W) & 15) == 0))
                            : (((Cout & 3) == 0) && ((((uintptr_t)W) & 15) == 0));

  if (t < 32) tapAny[t] = 0;
  __syncthreads();
  for (int e = t; e < BM * K; e += 256) {
    int r = e / K, k = e - r * K;
    int j = row0 + r, v = -1;
    if (j < n_out) v = nbr ? nbr[(size_t)j * K + k] : (j < n_in ? j : -1);
    nbrS[e] = v;
    if (v >= 0) tapAny[k] = 1;
  }
  __syncthreads();

  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nC = (Cin + BK - 1) / BK;
  // thread roles for staging
  const int a_r = t >> 2, a_kk = (t & 3) * 4;               // A: rows a_r, a_r+64 ; k offset a_kk..+3
  const int b_kk = t >> 4, b_n4 = (t & 15) * 4;             // B (normal): row b_kk, cols b_n4..+3
  const int bt_n = t >> 2, bt_kk = (t & 3) * 4;             // B (transposed weights): col bt_n, k bt_kk..+3

  float ra[2][4], rb[4];
  auto load_chunk = [&](int k, int c0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int idx = nbrS[(a_r + h * 64) * K + k];
      int c = c0 + a_kk;
      if (idx >= 0 && c < Cin) {
        const float* p = X + (size_t)idx * ldx + c;
        if (vecA && c + 3 < Cin) {
          float4 v = *(const float4*)p;
          ra[h][0] = v.x; ra[h][1] = v.y; ra[h][2] = v.z; ra[h][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[h][e] = (c + e < Cin) ? p[e] : 0.f;
        }
      } else {
        ra[h][0] = ra[h][1] = ra[h][2] = ra[h][3] = 0.f;
      }
    }
    if (!TRANS_W) {
      int c = c0 + b_kk, n = n0 + b_n4;
      if (c < Cin && n < Cout) {
        const float* p = W + ((size_t)k * Cin + c) * Cout + n;
        if (vecB && n + 3 < Cout) {
          float4 v = *(const float4*)p;
          rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rb[e] = (n + e < Cout) ? p[e] : 0.f;
        }
      } else {
        rb[0] = rb[1] = rb[2] = rb[3] = 0.f;
      }
    } else {
      // weight stored [K][Cout(this GEMM's N)][Cin(this GEMM's reduction)]
      int n = n0 + bt_n, c = c0 + bt_kk;
      if (n < Cout && c < Cin) {
        const float* p = W + ((size_t)k * Cout + n) * Cin + c;
        if (vecB && c + 3 < Cin) {
          float4 v = *(const float4*)p;
          rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rb[e] = (c + e < Cin) ? p[e] : 0.f;
        }
      } else {
        rb[0] = rb[1] = rb[2] = rb[3] = 0.f;
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int rk = a_kk + e;
        As[rk * BM + ((a_r + h * 64) ^ a_swz(rk))] = ra[h][e];
      }
    if (!TRANS_W) {
      *(float4*)&Bs[b_kk * LDB + b_n4] = make_float4(rb[0], rb[1], rb[2], rb[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) Bs[(bt_kk + e) * LDB + bt_n] = rb[e];
    }
  };

  // chunk iterator over (tap with any neighbour) x (C_in chunk)
  int k = 0, ci = 0;
  while (k < K && !tapAny[k]) ++k;
  bool have = k < K;
  if (have) load_chunk(k, 0);
  const int li = lane & 15, kq = lane >> 4;
  while (have) {
    store_chunk();
    __syncthreads();
    int nk = k, nci = ci + 1;
    if (nci >= nC) {
      nci = 0;
      ++nk;
      while (nk < K && !tapAny[nk]) ++nk;
    }
    bool nhave = nk < K;
    if (nhave) load_chunk(nk, nci * BK);     // global loads in flight under the MFMAs below
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      int rk = ks * 4 + kq;
      int sw = a_swz(rk);
      float a[2], b[4];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) a[mf] = As[rk * BM + ((wv * 32 + mf * 16 + li) ^ sw)];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) b[nf] = Bs[rk * LDB + nf * 16 + li];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
    }
    __syncthreads();
    k = nk; ci = nci; have = nhave;
  }

  // epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      int col = n0 + nf * 16 + li;
      if (col >= Cout) continue;
      float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + wv * 32 + mf * 16 + kq * 4 + r;
        if (row < n_out) {
          float* p = Y + (size_t)row * ldy + col;
          float v = acc[mf][nf][r] + bv;
          *p = accumulate ? (*p + v) : v;
        }
      }
    }
}

extern "C" int es_spconv_fwd(const float* X, int ldx, const float* W, const int* nbr, int n_out, int n_in, int K,
                             int Cin, int Cout, const float* bias, float* Y, int ldy, int trans_w, int accumulate,
                             void* stream) {
  if (n_out <= 0 || Cout <= 0) return 0;
  if (K > MAXK) return -2;
  dim3 grid(es_cdiv(n_out, BM), es_cdiv(Cout, BN));
  if (trans_w)
    hipLaunchKernelGGL(k_spconv<true>, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, W, nbr, n_out, n_in, K, Cin,
                       Cout, bias, Y, ldy, accumulate);
  else
    hipLaunchKernelGGL(k_spconv<false>, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, W, nbr, n_out, n_in, K,
                       Cin, Cout, bias, Y, ldy, accumulate);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------ wgrad
// dW[k][c][n] += sum_{j in row slice} X[nbr[j,k]][c] * dY[j][n]
#define WM 64
#define WN 64
#define WR 16
#define LDW (64 + 16)
__global__ __launch_bounds__(256) void k_spconv_wgrad(const float* __restrict__ X, int ldx,
                                                      const float* __restrict__ dY, int ldy,
                                                      const int* __restrict__ nbr, int n_out, int n_in, int K,
                                                      int Cin, int Cout, int rows_per_split,
                                                      float* __restrict__ dW) {
  __shared__ float As[WR * LDW];
  __shared__ float Bs[WR * LDW];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nCt = (Cin + WM - 1) / WM;
  const int k = blockIdx.x / nCt, c0 = (blockIdx.x % nCt) * WM;
  const int n0 = blockIdx.y * WN;
  const int rbeg = blockIdx.z * rows_per_split;
  const int rend = min(n_out, rbeg + rows_per_split);
  const bool vecA = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const bool vecB = ((ldy & 3) == 0) && ((((uintptr_t)dY) & 15) == 0);
  const int lr = t >> 4, l4 = (t & 15) * 4;
  const int li = lane & 15, kq = lane >> 4;

  f32x4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float ra[4], rb[4];
  auto load_rows = [&](int r0) {
    int j = r0 + lr;
    int idx = -1;
    if (j < rend) idx = nbr ? nbr[(size_t)j * K + k] : (j < n_in ? j : -1);
    int c = c0 + l4, n = n0 + l4;
    if (idx >= 0 && c < Cin) {
      const float* p = X + (size_t)idx * ldx + c;
      if (vecA && c + 3 < Cin) {
        float4 v = *(const float4*)p;
        ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[e] = (c + e < Cin) ? p[e] : 0.f;
      }
    } else {
      ra[0] = ra[1] = ra[2] = ra[3] = 0.f;
    }
    if (idx >= 0 && n < Cout) {
      const float* p = dY + (size_t)j * ldy + n;
      if (vecB && n + 3 < Cout) {
        float4 v = *(const float4*)p;
        rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[e] = (n + e < Cout) ? p[e] : 0.f;
      }
    } else {
      rb[0] = rb[1] = rb[2] = rb[3] = 0.f;
    }
  };

  if (rbeg < rend) load_rows(rbeg);
  for (int r0 = rbeg; r0 < rend; r0 += WR) {
    *(float4*)&As[lr * LDW + l4] = make_float4(ra[0], ra[1], ra[2], ra[3]);
    *(float4*)&Bs[lr * LDW + l4] = make_float4(rb[0], rb[1], rb[2], rb[3]);
    __syncthreads();
    if (r0 + WR < rend) load_rows(r0 + WR);
#pragma unroll
    for (int ks = 0; ks < WR / 4; ++ks) {
      int rk = ks * 4 + kq;
      float a = As[rk * LDW + wv * 16 + li];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        float b = Bs[rk * LDW + nf * 16 + li];
        acc[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[nf], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    int col = n0 + nf * 16 + li;
    if (col >= Cout) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int c = c0 + wv * 16 + kq * 4 + r;
      if (c < Cin) atomicAdd(dW + ((size_t)k * Cin + c) * Cout + col, acc[nf][r]);
    }
  }
}

extern "C" int es_spconv_wgrad(const float* X, int ldx, const float* dY, int ldy, const int* nbr, int n_out,
                               int n_in, int K, int Cin, int Cout, float* dW, void* stream) {
  if (n_out <= 0 || Cin <= 0 || Cout <= 0) return 0;
  int base = K * es_cdiv(Cin, WM) * es_cdiv(Cout, WN);
  int splits = es_cdiv(2048, base);
  int max_splits = es_cdiv(n_out, 128);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rows_per_split = es_cdiv(es_cdiv(n_out, splits), WR) * WR;
  splits = es_cdiv(n_out, rows_per_split);
  dim3 grid(K * es_cdiv(Cin, WM), es_cdiv(Cout, WN), splits);
  hipLaunchKernelGGL(k_spconv_wgrad, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, dY, ldy, nbr, n_out, n_in, K,
                     Cin, Cout, rows_per_split, dW);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------ bf16 MFMA path
// Same output-stationary structure on v_mfma_f32_16x16x32_bf16 (16x the f32 matrix rate): features stay f32 in HBM
// and are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while being staged into LDS; weights come from a per-step bf16
// copy laid out [K][N][Kr] (reduction index contiguous) so that both operands are read from LDS as 16-byte
// k-contiguous fragments.  Accumulation is f32.  Used for forward (W^T copy) and dgrad (natural copy).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define HBK 32
#define HLD (HBK + 8)      // bf16 elements per LDS row (80 B: keeps 16-B alignment, spreads banks)

__device__ inline uint32_t pack_bf16(float a, float b) {
  f32x2_t x = {a, b};
  bf16x2_t y = __builtin_convertvector(x, bf16x2_t);
  return *(uint32_t*)&y;
}

__global__ __launch_bounds__(256) void k_spconv_bf16(const float* __restrict__ X, int ldx,
                                                     const unsigned short* __restrict__ W /* [K][N][Kr] bf16 */,
                                                     const int* __restrict__ nbr, int n_out, int n_in, int K, int Cin,
                                                     int Cout, const float* __restrict__ bias, float* __restrict__ Y,
                                                     int ldy, int accumulate) {
  __shared__ __attribute__((aligned(16))) unsigned short As[BM * HLD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[BN * HLD];
  __shared__ int nbrS[BM * MAXK];
  __shared__ int tapAny[32];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int row0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const bool vecA = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const bool vecB = ((Cin & 7) == 0) && ((((uintptr_t)W) & 15) == 0);

  if (t < 32) tapAny[t] = 0;
  __syncthreads();
  for (int e = t; e < BM * K; e += 256) {
    int r = e / K, k = e - r * K;
    int j = row0 + r, v = -1;
    if (j < n_out) v = nbr ? nbr[(size_t)j * K + k] : (j < n_in ? j : -1);
    nbrS[e] = v;
    if (v >= 0) tapAny[k] = 1;
  }
  __syncthreads();

  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nC = (Cin + HBK - 1) / HBK;
  // A staging: thread t owns row t>>1 and 16 consecutive channels (one nbr lookup, four 16-byte loads, two 16-byte
  // LDS stores).  (A variant where 8 lanes share one 128-byte line per load instruction measured 20 % slower: four
  // map lookups and four 8-byte LDS stores per thread outweigh the better line utilisation.)
  const int a_r = t >> 1, a_kk = (t & 1) * 16;
  const int b_n = t >> 2, b_kk = (t & 3) * 8;           // B: one output channel, 8 consecutive reduction elements
  uint32_t ra[8];
  uint4 rb;
  auto load_chunk = [&](int k, int c0) {
    int idx = nbrS[a_r * K + k];
    int c = c0 + a_kk;
    if (idx >= 0 && c < Cin) {
      const float* p = X + (size_t)idx * ldx + c;
      if (vecA && c + 15 < Cin) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = ((const float4*)p)[q];
          ra[2 * q] = pack_bf16(v.x, v.y);
          ra[2 * q + 1] = pack_bf16(v.z, v.w);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v0 = (c + 2 * q < Cin) ? p[2 * q] : 0.f, v1 = (c + 2 * q + 1 < Cin) ? p[2 * q + 1] : 0.f;
          ra[q] = pack_bf16(v0, v1);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) ra[q] = 0u;
    }
    int n = n0 + b_n, cb = c0 + b_kk;
    if (n < Cout && cb < Cin) {
      const unsigned short* p = W + ((size_t)k * Cout + n) * Cin + cb;
      if (vecB && cb + 7 < Cin) {
        rb = *(const uint4*)p;
      } else {
        unsigned short h[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = (cb + q < Cin) ? p[q] : (unsigned short)0;
        rb.x = h[0] | ((uint32_t)h[1] << 16); rb.y = h[2] | ((uint32_t)h[3] << 16);
        rb.z = h[4] | ((uint32_t)h[5] << 16); rb.w = h[6] | ((uint32_t)h[7] << 16);
      }
    } else {
      rb = make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_chunk = [&]() {
    uint4* pa = (uint4*)&As[a_r * HLD + a_kk];
    pa[0] = make_uint4(ra[0], ra[1], ra[2], ra[3]);
    pa[1] = make_uint4(ra[4], ra[5], ra[6], ra[7]);
    *(uint4*)&Bs[b_n * HLD + b_kk] = rb;
  };

  int k = 0, ci = 0;
  while (k < K && !tapAny[k]) ++k;
  bool have = k < K;
  if (have) load_chunk(k, 0);
  const int li = lane & 15, kq = lane >> 4;
  while (have) {
    store_chunk();
    __syncthreads();
    int nk = k, nci = ci + 1;
    if (nci >= nC) {
      nci = 0;
      ++nk;
      while (nk < K && !tapAny[nk]) ++nk;
    }
    bool nhave = nk < K;
    if (nhave) load_chunk(nk, nci * HBK);
    bf16x8_t a[2], b[4];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) a[mf] = *(const bf16x8_t*)&As[(wv * 32 + mf * 16 + li) * HLD + kq * 8];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) b[nf] = *(const bf16x8_t*)&Bs[(nf * 16 + li) * HLD + kq * 8];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
    __syncthreads();
    k = nk; ci = nci; have = nhave;
  }
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      int col = n0 + nf * 16 + li;
      if (col >= Cout) continue;
      float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + wv * 32 + mf * 16 + kq * 4 + r;
        if (row < n_out) {
          float* p = Y + (size_t)row * ldy + col;
          float v = acc[mf][nf][r] + bv;
          *p = accumulate ? (*p + v) : v;
        }
      }
    }
}

extern "C" int es_spconv_fwd_bf16(const float* X, int ldx, const void* W_bf16, const int* nbr, int n_out, int n_in,
                                  int K, int Cin, int Cout, const float* bias, float* Y, int ldy, int accumulate,
                                  void* stream) {
  if (n_out <= 0 || Cout <= 0) return 0;
  if (K > MAXK) return -2;
  dim3 grid(es_cdiv(n_out, BM), es_cdiv(Cout, BN));
  hipLaunchKernelGGL(k_spconv_bf16, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, (const unsigned short*)W_bf16,
                     nbr, n_out, n_in, K, Cin, Cout, bias, Y, ldy, accumulate);
  ES_CHECK_LAUNCH();
  return 0;
}

// f32 [K][A][B] -> bf16 natural [K][A][B] and/or bf16 transposed [K][B][A]
__global__ void k_cast_weight(const float* __restrict__ w, int K, int A, int B, unsigned short* __restrict__ nat,
                              unsigned short* __restrict__ tr) {
  size_t tot = (size_t)K * A * B;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int b = (int)(e % B);
    size_t ka = e / B;
    int a = (int)(ka % A), k = (int)(ka / A);
    uint32_t p = pack_bf16(w[e], 0.f);
    unsigned short h = (unsigned short)(p & 0xffff);
    if (nat) nat[e] = h;
    if (tr) tr[((size_t)k * B + b) * A + a] = h;
  }
}
extern "C" int es_cast_weight_bf16(const float* w, int K, int A, int B, void* natural, void* transposed,
                                   void* stream) {
  size_t tot = (size_t)K * A * B;
  if (tot == 0) return 0;
  int g = es_cdiv((long long)tot, 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_cast_weight, dim3(g), dim3(256), 0, (hipStream_t)stream, w, K, A, B,
                     (unsigned short*)natural, (unsigned short*)transposed);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------ bf16 wgrad
// dW[k][c][n] += sum_j bf16(X[nbr[j,k]][c]) * bf16(dY[j][n]), f32 accumulate.  The reduction runs over rows, so both
// operands are staged TRANSPOSED ([channel][row], row-contiguous): each thread converts the same channel of two
// consecutive rows into one packed bf16x2 LDS word, which makes the MFMA fragments 16-byte k-contiguous reads.
#define GR 32                 // rows per chunk (= MFMA K)
#define GLD (GR + 8)
__global__ __launch_bounds__(256) void k_spconv_wgrad_bf16(const float* __restrict__ X, int ldx,
                                                           const float* __restrict__ dY, int ldy,
                                                           const int* __restrict__ nbr, int n_out, int n_in, int K,
                                                           int Cin, int Cout, int rows_per_split,
                                                           float* __restrict__ dW) {
  __shared__ __attribute__((aligned(16))) unsigned short As[WM * GLD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[WN * GLD];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nCt = (Cin + WM - 1) / WM;
  const int k = blockIdx.x / nCt, c0 = (blockIdx.x % nCt) * WM;
  const int n0 = blockIdx.y * WN;
  const int rbeg = blockIdx.z * rows_per_split;
  const int rend = min(n_out, rbeg + rows_per_split);
  const bool vecA = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const bool vecB = ((ldy & 3) == 0) && ((((uintptr_t)dY) & 15) == 0);
  const int rp = t & 15, l4 = (t >> 4) * 4;
  const int li = lane & 15, kq = lane >> 4;

  f32x4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint32_t ra[4], rb[4];
  auto load_rows = [&](int r0) {
    float xa[2][4], xb[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int j = r0 + 2 * rp + h;
      int idx = -1;
      if (j < rend) idx = nbr ? nbr[(size_t)j * K + k] : (j < n_in ? j : -1);
      int c = c0 + l4, n = n0 + l4;
      if (idx >= 0 && c < Cin) {
        const float* p = X + (size_t)idx * ldx + c;
        if (vecA && c + 3 < Cin) {
          float4 v = *(const float4*)p;
          xa[h][0] = v.x; xa[h][1] = v.y; xa[h][2] = v.z; xa[h][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) xa[h][e] = (c + e < Cin) ? p[e] : 0.f;
        }
      } else {
        xa[h][0] = xa[h][1] = xa[h][2] = xa[h][3] = 0.f;
      }
      if (idx >= 0 && n < Cout) {
        const float* p = dY + (size_t)j * ldy + n;
        if (vecB && n + 3 < Cout) {
          float4 v = *(const float4*)p;
          xb[h][0] = v.x; xb[h][1] = v.y; xb[h][2] = v.z; xb[h][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) xb[h][e] = (n + e < Cout) ? p[e] : 0.f;
        }
      } else {
        xb[h][0] = xb[h][1] = xb[h][2] = xb[h][3] = 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ra[e] = pack_bf16(xa[0][e], xa[1][e]);
      rb[e] = pack_bf16(xb[0][e], xb[1][e]);
    }
  };

  if (rbeg < rend) load_rows(rbeg);
  for (int r0 = rbeg; r0 < rend; r0 += GR) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *(uint32_t*)&As[(l4 + e) * GLD + 2 * rp] = ra[e];
      *(uint32_t*)&Bs[(l4 + e) * GLD + 2 * rp] = rb[e];
    }
    __syncthreads();
    if (r0 + GR < rend) load_rows(r0 + GR);
    bf16x8_t a = *(const bf16x8_t*)&As[(wv * 16 + li) * GLD + kq * 8];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      bf16x8_t b = *(const bf16x8_t*)&Bs[(nf * 16 + li) * GLD + kq * 8];
      acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[nf], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    int col = n0 + nf * 16 + li;
    if (col >= Cout) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int c = c0 + wv * 16 + kq * 4 + r;
      if (c < Cin) atomicAdd(dW + ((size_t)k * Cin + c) * Cout + col, acc[nf][r]);
    }
  }
}

extern "C" int es_spconv_wgrad_bf16(const float* X, int ldx, const float* dY, int ldy, const int* nbr, int n_out,
                                    int n_in, int K, int Cin, int Cout, float* dW, void* stream) {
  if (n_out <= 0 || Cin <= 0 || Cout <= 0) return 0;
  int base = K * es_cdiv(Cin, WM) * es_cdiv(Cout, WN);
  int splits = es_cdiv(4096, base);
  int max_splits = es_cdiv(n_out, 256);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rows_per_split = es_cdiv(es_cdiv(n_out, splits), GR) * GR;
  splits = es_cdiv(n_out, rows_per_split);
  dim3 grid(K * es_cdiv(Cin, WM), es_cdiv(Cout, WN), splits);
  hipLaunchKernelGGL(k_spconv_wgrad_bf16, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, dY, ldy, nbr, n_out, n_in,
                     K, Cin, Cout, rows_per_split, dW);
  ES_CHECK_LAUNCH();
  return 0;
}
