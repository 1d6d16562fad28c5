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
