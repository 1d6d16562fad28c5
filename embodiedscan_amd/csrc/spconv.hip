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

// run-time tuning switches (A/B measurements without a rebuild): key 1 = ping-pong LDS in the fast bf16 kernels
static int ES_OPT_PINGPONG = 1;
static int ES_OPT_WGRAD_HUGE = 1;      // 256 x 256 weight-gradient tile for wide layers (both operands bf16 shadows)
static int ES_OPT_ROWGEMM = 1;         // streaming row GEMM for K = 1 on the identity map
static int ES_OPT_WG_BIG_TARGET = 2048;    // workgroups a 128 x 128-tile weight-gradient launch aims for (row slices); round 4: 8192 -> 2048
                                           // (profiles/r4m_sweep.txt: a quarter of the slices = a quarter of the partial-tile traffic, step -0.1 .. -0.35 ms)
static int ES_OPT_WG_BIG_ROWS = 512;       // ... and the fewest rows a slice may have
static int ES_OPT_WG_SMALL_TARGET = 1024;  // the same for the 64 x 64 tile (round 4: 4096 -> 1024; 512 is slower: 26.9 ms)
static int ES_OPT_WG_CAP_MB = 256;         // workspace of partial tiles per launch (weights <= 8 M floats)
static int ES_OPT_FWD_SPLIT_WGS = 384;     // forward / dgrad launches with fewer workgroups split their tap list (sweep, session F:
                                           // 192 -> 384 neutral on mv-3ddet, -0.8 ms on the occupancy step; 96: +3.5 / +4.7 ms)
static int ES_OPT_DMA = 2;                 // LDS-DMA fast kernel (k_spconv_bf16_dma) for bf16 input rows: 0 off, 1 = 32-channel
                                           // chunks, 2 = 64-channel chunks where C_in % 64 == 0 ...
static int ES_OPT_DMA_MIN_CIN = 768;       // ... for layers with at least this many input channels.  Session K (profiles/r3k_*): the
                                           // dense occupancy neck (768 .. 3072 channels) runs 13.5 % faster with 64-channel DMA chunks
                                           // (97.4 -> 84.3 ms per 7 steps, step 57.4 -> 56.1 ms); the 64 .. 256-channel sparse layers of
                                           // mv-3ddet are equal within 2 % either way (LDS activity -62 %, no bank conflicts left, same
                                           // MFMA busy: those launches are bound by the gather through L2, not by LDS)
static int ES_OPT_WGRAD_TR = 1;            // weight-gradient 128 x 128 tile with LDS-DMA staging + ds_read_b64_tr_b16 (k_spconv_wgrad_bf16_tr): run on
                                           // hardware in round 4 (profiles/r4a_*): bit-identical to the register-transposing tile, step -0.2 ms
static int ES_OPT_ROWGEMM2 = 1;            // second-generation row GEMM (swapped MFMA operands, register epilogue with 16-byte accesses)
static int ES_OPT_SPLIT_FOLD = 0;           // tap-split launches: partial tiles added by a second launch (0, default) or by the tile's last workgroup
                                           // inside the launch (1, es_set_option key 16).  MEASURED AND REJECTED as the default (round 4,
                                           // profiles/r4l_sweep.txt): the in-kernel election needs an agent-scope release fence per workgroup
                                           // (buffer_wbl2: the partial tiles must reach memory, their readers sit on other XCDs), and hundreds of
                                           // L2 write-backs inside a streaming kernel cost far more than the 70 reduce launches they replace:
                                           // mv-3ddet step 30.9 ms against 25.9 (with a seq_cst fence, i.e. + buffer_inv: 32.6).  Kept as a tested option.
static int ES_OPT_RG128_MIN_CIN = 0;       // row GEMM (K = 1): 128-column tiles only for layers with at least this many input channels
static int ES_OPT_NARROW_SLICES = 768;      // ... row slices (workgroups) of their weight gradient (key 22): three per CU
static int ES_OPT_LIN_SMALL = 256;         // K = 1 launches with fewer 128-row workgroups than this run the 64 x 64 whole-stage kernel (key 24; 0: off)
static int ES_OPT_EXPAND = 65536;          // C -> 4 C bf16 expansion layers from this many rows on k_expand_bf16 (key 25; 0: off)
static int ES_OPT_EXPAND_WGS = 1024;       // ... its persistent workgroups (key 26)
static int ES_OPT_RG320 = 1;               // row GEMM with 320 output columns as one column tile (key 23)
static int ES_OPT_NARROW = 1;              // 3-channel K = 27 convolutions (MinkResNet.conv1) on the lane-per-output-channel kernels (key 21)
static int ES_OPT_WSHARE = 0;              // tap-split launches in the weight-sharing workgroup order (key 20)
static int ES_OPT_RG128_MIN_WGS = 0;       // ... and only for launches with at least this many 128-column workgroups (key 19; round 6 A/B: no measurable change, off)
extern int ES_OPT_NORM_CB_ROWS;          // rowops.hip: one-launch norm for matrices with at most this many rows (key 15)
extern int ES_OPT_NORM_CB_BWD;           // ... for the backward pass too (key 17)
extern int ES_OPT_ELECT_SAFE;            // rowops.hip: agent-scope fences in the last-workgroup elections (key 18)
extern int ES_OPT_NORM_CHUNK;            // rowops.hip: rows per norm-statistics chunk (key 9)
extern "C" int es_set_option(int key, int value) {
  if (key == 1) { ES_OPT_PINGPONG = value; return 0; }
  if (key == 2) { ES_OPT_WGRAD_HUGE = value; return 0; }
  if (key == 3) { ES_OPT_ROWGEMM = value; return 0; }
  if (key == 4) { ES_OPT_WG_BIG_TARGET = value; return 0; }
  if (key == 5) { ES_OPT_WG_BIG_ROWS = value; return 0; }
  if (key == 6) { ES_OPT_WG_SMALL_TARGET = value; return 0; }
  if (key == 7) { ES_OPT_WG_CAP_MB = value; return 0; }
  if (key == 8) { ES_OPT_FWD_SPLIT_WGS = value; return 0; }
  if (key == 9) { ES_OPT_NORM_CHUNK = value; return 0; }
  if (key == 10) { ES_OPT_DMA = value; return 0; }
  if (key == 11) { ES_OPT_DMA_MIN_CIN = value; return 0; }
  if (key == 12) { ES_OPT_RG128_MIN_CIN = value; return 0; }
  if (key == 13) { ES_OPT_ROWGEMM2 = value; return 0; }
  if (key == 14) { ES_OPT_WGRAD_TR = value; return 0; }
  if (key == 15) { ES_OPT_NORM_CB_ROWS = value; return 0; }
  if (key == 16) { ES_OPT_SPLIT_FOLD = value; return 0; }
  if (key == 17) { ES_OPT_NORM_CB_BWD = value; return 0; }
  if (key == 18) { ES_OPT_ELECT_SAFE = value; return 0; }
  if (key == 19) { ES_OPT_RG128_MIN_WGS = value; return 0; }
  if (key == 20) { ES_OPT_WSHARE = value; return 0; }
  if (key == 21) { ES_OPT_NARROW = value; return 0; }
  if (key == 22) { ES_OPT_NARROW_SLICES = value; return 0; }
  if (key == 23) { ES_OPT_RG320 = value; return 0; }
  if (key == 24) { ES_OPT_LIN_SMALL = value; return 0; }
  if (key == 25) { ES_OPT_EXPAND = value; return 0; }
  if (key == 26) { ES_OPT_EXPAND_WGS = value; return 0; }
  return -2;
}


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

// ------------------------------------------------------------------ narrow-input convolution (round 6)
// MinkResNet.conv1 (mink_resnet.py:131: 3 point-colour channels -> 64, k3 s2) on the exact-f32 path: K * Cin = 81 products per
// output element.  The tiled kernels above pad Cin = 3 to a 16-channel MFMA chunk and walk 27 mostly empty taps per tile (99 us
// forward, 350 us weight gradient on 132 k rows); here a lane owns one output channel with its K * Cin weights (forward) /
// partial sums (weight gradient) in REGISTERS, a wave walks the rows of a 64-row tile, the gathered inputs sit in LDS as
// [row][tap][4] floats read as one broadcast ds_read_b128 per PRESENT tap (a stride-2 map fills ~ 5 of 27), absent taps are
// skipped by a wave-uniform test of the row's tap mask.  Plain f32 FMAs in (tap, channel) order.
#define NW_ROWS 64
#if defined(ES_EMU)
#define NW_UNIFORM(x) (x)
#else
#define NW_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
template <int CIN>
__device__ __forceinline__ void narrow_stage(const float* __restrict__ X, int ldx, const int* __restrict__ nbr, int row0,
                                             int n_out, int n_in, float* xg, int* idxS, int* maskS) {
  const int t = threadIdx.x;
  // two rounds of INDEPENDENT loads (all map entries of this thread, then all input rows): two memory latencies per tile, not 2 x 7
  constexpr int NI = (NW_ROWS * 27 + 255) / 256;
  int v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = t + i * 256;
    v[i] = -1;
    if (e < NW_ROWS * 27 && row0 + e / 27 < n_out) v[i] = nbr[(size_t)row0 * 27 + e];
    if (v[i] >= n_in) v[i] = -1;
  }
  float xv[NI][CIN];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float* p = X + (size_t)(v[i] >= 0 ? v[i] : 0) * ldx;
#pragma unroll
    for (int c = 0; c < CIN; ++c) xv[i][c] = p[c];
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = t + i * 256;
    if (e < NW_ROWS * 27) {
      idxS[e] = v[i];
      const bool ok = v[i] >= 0;
      float4 x4 = make_float4(ok ? xv[i][0] : 0.f, 0.f, 0.f, 0.f);
      if (CIN > 1) x4.y = ok ? xv[i][1] : 0.f;
      if (CIN > 2) x4.z = ok ? xv[i][2] : 0.f;
      if (CIN > 3) x4.w = ok ? xv[i][3] : 0.f;
      *(float4*)&xg[e * 4] = x4;
    }
  }
  __syncthreads();
  if (t < NW_ROWS) {
    int m = 0;
#pragma unroll
    for (int k = 0; k < 27; ++k) m |= (idxS[t * 27 + k] >= 0) ? (1 << k) : 0;
    maskS[t] = m;
  }
  __syncthreads();
}

template <int CIN>
__global__ __launch_bounds__(256) void k_spconv_narrow_fwd(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                           const int* __restrict__ nbr, int n_out, int n_in,
                                                           const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                           int accumulate) {
  __shared__ __attribute__((aligned(16))) float xg[NW_ROWS * 27 * 4];
  __shared__ int idxS[NW_ROWS * 27];
  __shared__ int maskS[NW_ROWS];
  const int t = threadIdx.x, co = t & 63, wv = t >> 6;
  float w[27][CIN];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int c = 0; c < CIN; ++c) w[k][c] = W[(k * CIN + c) * 64 + co];
  const float bv = bias ? bias[co] : 0.f;
  const int tiles = (n_out + NW_ROWS - 1) / NW_ROWS;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int row0 = tile * NW_ROWS;
    narrow_stage<CIN>(X, ldx, nbr, row0, n_out, n_in, xg, idxS, maskS);
    for (int rr = 0; rr < NW_ROWS / 4; ++rr) {
      const int r = wv * (NW_ROWS / 4) + rr;
      if (row0 + r >= n_out) break;
      const int m = NW_UNIFORM(maskS[r]);
      float acc = bv;
#pragma unroll
      for (int k = 0; k < 27; ++k)
        if (m & (1 << k)) {
          const float4 x4 = *(const float4*)&xg[(r * 27 + k) * 4];
          acc = fmaf(x4.x, w[k][0], acc);
          if (CIN > 1) acc = fmaf(x4.y, w[k][1], acc);
          if (CIN > 2) acc = fmaf(x4.z, w[k][2], acc);
          if (CIN > 3) acc = fmaf(x4.w, w[k][3], acc);
        }
      float* p = Y + (size_t)(row0 + r) * ldy + co;
      *p = accumulate ? (*p + acc) : acc;
    }
    __syncthreads();
  }
}
static bool narrow_ok(const int* nbr, int K, int Cin, int Cout) { return ES_OPT_NARROW && nbr != nullptr && K == 27 && Cin == 3 && Cout == 64; }

extern "C" int es_spconv_fwd(const float* X, int ldx, const float* W, const int* nbr, int n_out, int n_in, int K,
                             int Cin, int Cout, const float* bias, float* Y, int ldy, int trans_w, int accumulate,
                             void* stream) {
  if (n_out <= 0 || Cout <= 0) return 0;
  if (K > MAXK) return -2;
  if (!trans_w && narrow_ok(nbr, K, Cin, Cout)) {
    const int tiles = es_cdiv(n_out, NW_ROWS);
    hipLaunchKernelGGL(k_spconv_narrow_fwd<3>, dim3(tiles > 2048 ? 2048 : tiles), dim3(256), 0, (hipStream_t)stream, X, ldx, W, nbr,
                       n_out, n_in, bias, Y, ldy, accumulate);
    ES_CHECK_LAUNCH();
    return 0;
  }
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
// where a weight-gradient workgroup puts one element of its partial tile: slice `bz` of the workspace (plain store), or --
// without a workspace the launch has ONE row slice -- straight into dW (this workgroup is the element's only writer)
__device__ __forceinline__ void wgrad_emit(float* __restrict__ dW, float* __restrict__ ws, int bz, size_t dw_floats,
                                           size_t off, float v, int accumulate) {
  if (ws) ws[(size_t)bz * dw_floats + off] = v;
  else if (accumulate) dW[off] += v;
  else dW[off] = v;                     // first gradient of the step for this weight: a plain store (no read, no wait)
}
// dW[k][c][n] += sum_{j in row slice} X[nbr[j,k]][c] * dY[j][n]
#define WM 64
#define WN 64
#define WR 16
#define LDW (64 + 16)
__global__ __launch_bounds__(256) void k_spconv_wgrad(const float* __restrict__ X, int ldx,
                                                      const float* __restrict__ dY, int ldy,
                                                      const int* __restrict__ nbr, int n_out, int n_in, int K,
                                                      int Cin, int Cout, int rows_per_split,
                                                      float* __restrict__ dW, float* __restrict__ ws, int accumulate) {
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
      if (c < Cin) wgrad_emit(dW, ws, blockIdx.z, (size_t)K * Cin * Cout, ((size_t)k * Cin + c) * Cout + col, acc[nf][r], accumulate);
    }
  }
}

// ---- deterministic split over rows (round 3): every (tap, channel tile, row slice) workgroup owns its partial tile.
// splits == 1: the tile is added straight into dW (sole owner, plain read-modify-write); splits > 1: the partial tiles go to
// a caller-provided workspace laid out [slice][K][Cin][Cout] and k_wgrad_reduce adds them to dW in slice order.  No float
// atomics anywhere: two runs give bit-identical weight gradients (rounds 1-2 used f32 atomics, 1e-6 run-to-run).
#define WGRAD_WS_CAP_FLOATS (64ll << 20)     // 256 MB of partial tiles per launch (measured: a 64 MB cap -- 36 slices for the 128 x 128 x 27
                                            // head kernels instead of 151 -- cost more in lost parallelism, 1.58 -> 1.82 ms per step, than
                                            // it saved in workspace traffic; one slice of a 768^2 x 27 kernel: 2.3 -> 11 ms)
#define WGRAD_WS_CAP_BIG (256ll << 20)       // weights above 8 M floats (the dense occupancy neck): up to 1 GB, i.e. still 4 .. 8 slices
struct WgradPlan { int kind, splits, rows_per_split; };   // kind: 0 exact-f32 64x64, 1 bf16 64x64, 2 bf16 128x128, 3 bf16 256x256
static int cap_splits(int splits, long long dw_floats, bool have_ws) {
  if (!have_ws) return 1;
  long long cap = (dw_floats > (8ll << 20) ? WGRAD_WS_CAP_BIG : ((long long)ES_OPT_WG_CAP_MB << 18)) / (dw_floats > 0 ? dw_floats : 1);
  if (cap < 1) cap = 1;
  if (splits > cap) splits = (int)cap;
  return splits < 1 ? 1 : splits;
}
// narrow-input weight gradient (see k_spconv_narrow_fwd): lane = output channel, wave g owns taps g, g + 4, ...; a workgroup walks
// its row slice in 64-row tiles (gathered inputs + the dY tile in LDS) and leaves its [27][CIN][64] partial sums in the slice's
// workspace block (or, as the only slice, in dW).
template <int CIN>
__global__ __launch_bounds__(256) void k_spconv_narrow_wgrad(const float* __restrict__ X, int ldx, const float* __restrict__ dY,
                                                             int ldy, const int* __restrict__ nbr, int n_out, int n_in,
                                                             int rows_per_split, float* __restrict__ dW, float* __restrict__ ws,
                                                             int accumulate) {
  __shared__ __attribute__((aligned(16))) float xg[NW_ROWS * 27 * 4];
  __shared__ __attribute__((aligned(16))) float dyS[NW_ROWS * 64];
  __shared__ int idxS[NW_ROWS * 27];
  __shared__ int maskS[NW_ROWS];
  const int t = threadIdx.x, co = t & 63, g = t >> 6;
  float acc[7][CIN];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[i][c] = 0.f;
  const int rbeg = blockIdx.x * rows_per_split, rend = min(n_out, rbeg + rows_per_split);
  const bool vec = ((ldy & 3) == 0) && ((((uintptr_t)dY) & 15) == 0);
  for (int row0 = rbeg; row0 < rend; row0 += NW_ROWS) {
    for (int e = t; e < NW_ROWS * 16; e += 256) {             // the dY tile: 16 float4 per row
      const int r = e >> 4, c4 = (e & 15) * 4, j = row0 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < rend) {
        const float* p = dY + (size_t)j * ldy + c4;
        if (vec) v = *(const float4*)p;
        else v = make_float4(p[0], p[1], p[2], p[3]);
      }
      *(float4*)&dyS[r * 64 + c4] = v;
    }
    narrow_stage<CIN>(X, ldx, nbr, row0, rend, n_in, xg, idxS, maskS);
    const int nr = min(NW_ROWS, rend - row0);
    for (int r = 0; r < nr; ++r) {
      const int m = NW_UNIFORM(maskS[r]) >> g;
      if (!(m & 0x1111111)) continue;
      const float dv = dyS[r * 64 + co];
#pragma unroll
      for (int i = 0; i < 7; ++i)
        if (m & (1 << (4 * i))) {                                // tap g + 4 i (bit 27 and up of a mask are never set)
          const float4 x4 = *(const float4*)&xg[(r * 27 + g + 4 * i) * 4];
          acc[i][0] = fmaf(x4.x, dv, acc[i][0]);
          if (CIN > 1) acc[i][1] = fmaf(x4.y, dv, acc[i][1]);
          if (CIN > 2) acc[i][2] = fmaf(x4.z, dv, acc[i][2]);
          if (CIN > 3) acc[i][3] = fmaf(x4.w, dv, acc[i][3]);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int k = g + 4 * i;
    if (k < 27)
#pragma unroll
      for (int c = 0; c < CIN; ++c)
        wgrad_emit(dW, ws, blockIdx.x, (size_t)27 * CIN * 64, ((size_t)k * CIN + c) * 64 + co, acc[i][c], accumulate);
  }
}
static WgradPlan wgrad_plan_f32(int n_out, int K, int Cin, int Cout, bool have_ws) {
  if (ES_OPT_NARROW && K == 27 && Cin == 3 && Cout == 64) {      // (the launcher falls back to one slice of the tiled kernel without a map)
    int splits = cap_splits(es_cdiv(n_out, NW_ROWS) < ES_OPT_NARROW_SLICES ? es_cdiv(n_out, NW_ROWS) : ES_OPT_NARROW_SLICES, (long long)K * Cin * Cout, have_ws);
    int rows_per_split = es_cdiv(es_cdiv(n_out, splits), NW_ROWS) * NW_ROWS;
    return WgradPlan{4, es_cdiv(n_out, rows_per_split), rows_per_split};
  }
  int base = K * es_cdiv(Cin, WM) * es_cdiv(Cout, WN);
  int splits = es_cdiv(2048, base);
  int max_splits = es_cdiv(n_out, 128);
  if (splits > max_splits) splits = max_splits;
  splits = cap_splits(splits, (long long)K * Cin * Cout, have_ws);
  int rows_per_split = es_cdiv(es_cdiv(n_out, splits), WR) * WR;
  splits = es_cdiv(n_out, rows_per_split);
  return WgradPlan{0, splits, rows_per_split};
}
__global__ void k_wgrad_reduce(const float* __restrict__ ws, int splits, size_t n, float* __restrict__ dW, int accumulate) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    float s = ws[e];
    for (int z = 1; z < splits; ++z) s += ws[(size_t)z * n + e];
    dW[e] = accumulate ? dW[e] + s : s;
  }
}
__global__ void k_wgrad_reduce4(const float4* __restrict__ ws, int splits, size_t n4, float4* __restrict__ dW, int accumulate) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
    float4 s = ws[e];
    for (int z = 1; z < splits; ++z) {
      float4 v = ws[(size_t)z * n4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (accumulate) {
      float4 d = dW[e];
      s = make_float4(d.x + s.x, d.y + s.y, d.z + s.z, d.w + s.w);
    }
    dW[e] = s;
  }
}
// Many slices of a SMALL weight (the 1x1 convolutions of the image backbone: a few thousand weights, > 1e6 rows -> thousands of
// slices): one thread per element would walk thousands of slices serially (measured: 1.3 ms for a 64 x 32 weight).  Here
// a 256-thread block owns EB = 256 / R elements and R slice ranges: thread (el, rr) adds its contiguous range of slices in
// order, the R range sums are then added in range order through LDS -- a fixed tree, so still bit-reproducible.
template <int R>
__global__ __launch_bounds__(256) void k_wgrad_reduce_ranges(const float* __restrict__ ws, int splits, size_t n,
                                                             float* __restrict__ dW, int accumulate) {
  constexpr int EB = 256 / R;
  __shared__ float part[R][EB];
  const int el = threadIdx.x % EB, rr = threadIdx.x / EB;
  const size_t e = (size_t)blockIdx.x * EB + el;
  const int per = (splits + R - 1) / R;
  const int z0 = rr * per, z1 = min(splits, z0 + per);
  float s = 0.f;
  if (e < n)
    for (int z = z0; z < z1; ++z) s += ws[(size_t)z * n + e];
  part[rr][el] = s;
  __syncthreads();
  if (rr == 0 && e < n) {
    float t = part[0][el];
#pragma unroll
    for (int r = 1; r < R; ++r) t += part[r][el];
    dW[e] = accumulate ? dW[e] + t : t;
  }
}
static int wgrad_reduce(const float* ws, int splits, size_t n, float* dW, int accumulate, hipStream_t st) {
  if (splits <= 1 || n == 0) return 0;
  if (splits >= 64) {                      // (the choice depends on the slice count only: same launch -> same summation tree)
    if (splits >= 512) hipLaunchKernelGGL(k_wgrad_reduce_ranges<64>, dim3(es_cdiv((long long)n, 4)), dim3(256), 0, st, ws, splits, n, dW, accumulate);
    else hipLaunchKernelGGL(k_wgrad_reduce_ranges<16>, dim3(es_cdiv((long long)n, 16)), dim3(256), 0, st, ws, splits, n, dW, accumulate);
  } else if ((n % 4 == 0) && (((((uintptr_t)ws) | ((uintptr_t)dW)) & 15) == 0)) {
    int g = es_cdiv((long long)(n / 4), 256);
    hipLaunchKernelGGL(k_wgrad_reduce4, dim3(g > 8192 ? 8192 : g), dim3(256), 0, st, (const float4*)ws, splits, n / 4, (float4*)dW, accumulate);
  } else {
    int g = es_cdiv((long long)n, 256);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(g > 8192 ? 8192 : g), dim3(256), 0, st, ws, splits, n, dW, accumulate);
  }
  ES_CHECK_LAUNCH();
  return 0;
}

extern "C" int es_spconv_wgrad(const float* X, int ldx, const float* dY, int ldy, const int* nbr, int n_out,
                               int n_in, int K, int Cin, int Cout, float* dW, int accumulate, float* ws, size_t ws_floats,
                               void* stream) {
  if (n_out <= 0 || Cin <= 0 || Cout <= 0) return 0;
  WgradPlan p = wgrad_plan_f32(n_out, K, Cin, Cout, ws != nullptr);
  const size_t nw = (size_t)K * Cin * Cout;
  if (p.splits > 1 && ws_floats < (size_t)p.splits * nw) return -5;
  if (p.kind == 4 && nbr != nullptr) {
    hipLaunchKernelGGL(k_spconv_narrow_wgrad<3>, dim3(p.splits), dim3(256), 0, (hipStream_t)stream, X, ldx, dY, ldy, nbr, n_out, n_in,
                       p.rows_per_split, dW, p.splits > 1 ? ws : nullptr, accumulate);
    ES_CHECK_LAUNCH();
    return wgrad_reduce(ws, p.splits, nw, dW, accumulate, (hipStream_t)stream);
  }
  dim3 grid(K * es_cdiv(Cin, WM), es_cdiv(Cout, WN), p.splits);
  hipLaunchKernelGGL(k_spconv_wgrad, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, dY, ldy, nbr, n_out, n_in, K,
                     Cin, Cout, p.rows_per_split, dW, p.splits > 1 ? ws : nullptr, accumulate);
  ES_CHECK_LAUNCH();
  return wgrad_reduce(ws, p.splits, nw, dW, accumulate, (hipStream_t)stream);
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
#define HLD (HBK + 16)     // bf16 elements per padded LDS row: 96 B.  (80-B rows, rounds 1-2, put two pieces of every ds_read_b128
                           // lane group on one bank slot -- 50 % conflict cycles in profiles/r1_pmc_sq_v3.txt; 96 B is conflict-free for
                           // the documented lane groups: tools/lds_conflicts.py)

// activation-storage flags of the image backbone (round 3): bit 0: the Y rows are bf16, bit 1: the ep_res rows are bf16
#define ES_IO_Y16 1
#define ES_IO_R16 2
#define ES_IO_WSHARE 0x10000    // tap-split launches: workgroup order in which the row tiles of one (column tile, tap slice) share an XCD
__device__ __forceinline__ float4 bf16x4_to_f32(uint2 v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
__device__ inline uint32_t pack_bf16(float a, float b) {
  f32x2_t x = {a, b};
  bf16x2_t y = __builtin_convertvector(x, bf16x2_t);
  return *(uint32_t*)&y;
}

// BNT = output channels per workgroup (64, or 128 when Cout >= 128: the gathered A tile is then re-used for twice as
// many output channels).  The (tap, C_in-chunk) stream is software-pipelined TWO chunks deep in registers (raw f32 loads
// are only converted to bf16 when they are written to LDS), because the PMC profile of the first version showed the
// waves parked on memory 72 % of their cycles with one chunk in flight.
template <int BNT>
__global__ __launch_bounds__(256) void k_spconv_bf16(const float* __restrict__ X, int ldx,
                                                     const unsigned short* __restrict__ W /* [K][N][Kr] bf16 */,
                                                     const int* __restrict__ nbr, int n_out, int n_in, int K, int Cin,
                                                     int Cout, const float* __restrict__ bias, float* __restrict__ Y,
                                                     int ldy, int accumulate,
                                                     const float* __restrict__ ep_scale,
                                                     const float* __restrict__ ep_shift,
                                                     const float* __restrict__ ep_res, int ep_ldr, int ep_act, int x_half,
                                                     int io) {
  constexpr int NF = BNT / 16;      // 16-wide output fragments per wave
  constexpr int NB = BNT / 64;      // 16-byte weight pieces per thread and chunk
  __shared__ __attribute__((aligned(16))) unsigned short As[BM * HLD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[BNT * HLD];
  __shared__ int nbrS[BM * MAXK];
  __shared__ int tapAny[32];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int row0 = blockIdx.x * BM, n0 = blockIdx.y * BNT;
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

  f32x4 acc[2][NF];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nC = (Cin + HBK - 1) / HBK;
  const int a_r = t >> 1, a_kk = (t & 1) * 16;          // A: one row, 16 consecutive channels
  const int b_n = t >> 2, b_kk = (t & 3) * 8;           // B: output channel b_n (+64), 8 consecutive reduction elements
  struct Regs { float4 a[4]; uint4 b[NB]; };
  struct It { int k, ci; };

  auto load_chunk = [&](Regs& R, It it) {
    int c0 = it.ci * HBK;
    int idx = nbrS[a_r * K + it.k];
    int c = c0 + a_kk;
    if (idx >= 0 && c < Cin && x_half) {                  // bf16 input rows (image backbone): widened exactly, re-rounded as is
      const unsigned short* ph = (const unsigned short*)X + (size_t)idx * ldx + c;
      if (c + 15 < Cin && (ldx & 7) == 0 && ((((uintptr_t)X) & 15) == 0)) {   // 16 channels = two 16-byte loads
        const uint4 u0 = ((const uint4*)ph)[0], u1 = ((const uint4*)ph)[1];
        R.a[0] = make_float4(__uint_as_float(u0.x << 16), __uint_as_float(u0.x & 0xffff0000u), __uint_as_float(u0.y << 16),
                             __uint_as_float(u0.y & 0xffff0000u));
        R.a[1] = make_float4(__uint_as_float(u0.z << 16), __uint_as_float(u0.z & 0xffff0000u), __uint_as_float(u0.w << 16),
                             __uint_as_float(u0.w & 0xffff0000u));
        R.a[2] = make_float4(__uint_as_float(u1.x << 16), __uint_as_float(u1.x & 0xffff0000u), __uint_as_float(u1.y << 16),
                             __uint_as_float(u1.y & 0xffff0000u));
        R.a[3] = make_float4(__uint_as_float(u1.z << 16), __uint_as_float(u1.z & 0xffff0000u), __uint_as_float(u1.w << 16),
                             __uint_as_float(u1.w & 0xffff0000u));
      } else {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (c + q < Cin) ? __uint_as_float((uint32_t)ph[q] << 16) : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) R.a[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
      }
    } else if (idx >= 0 && c < Cin) {
      const float* p = X + (size_t)idx * ldx + c;
      if (vecA && c + 15 < Cin) {
#pragma unroll
        for (int q = 0; q < 4; ++q) R.a[q] = ((const float4*)p)[q];
      } else {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (c + q < Cin) ? p[q] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) R.a[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) R.a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int h = 0; h < NB; ++h) {
      int n = n0 + b_n + h * 64, cb = c0 + b_kk;
      if (n < Cout && cb < Cin) {
        const unsigned short* p = W + ((size_t)it.k * Cout + n) * Cin + cb;
        if (vecB && cb + 7 < Cin) {
          R.b[h] = *(const uint4*)p;
        } else {
          unsigned short hh[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) hh[q] = (cb + q < Cin) ? p[q] : (unsigned short)0;
          R.b[h].x = hh[0] | ((uint32_t)hh[1] << 16); R.b[h].y = hh[2] | ((uint32_t)hh[3] << 16);
          R.b[h].z = hh[4] | ((uint32_t)hh[5] << 16); R.b[h].w = hh[6] | ((uint32_t)hh[7] << 16);
        }
      } else {
        R.b[h] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto store_chunk = [&](const Regs& R) {
    uint4* pa = (uint4*)&As[a_r * HLD + a_kk];
    pa[0] = make_uint4(pack_bf16(R.a[0].x, R.a[0].y), pack_bf16(R.a[0].z, R.a[0].w), pack_bf16(R.a[1].x, R.a[1].y),
                       pack_bf16(R.a[1].z, R.a[1].w));
    pa[1] = make_uint4(pack_bf16(R.a[2].x, R.a[2].y), pack_bf16(R.a[2].z, R.a[2].w), pack_bf16(R.a[3].x, R.a[3].y),
                       pack_bf16(R.a[3].z, R.a[3].w));
#pragma unroll
    for (int h = 0; h < NB; ++h) *(uint4*)&Bs[(b_n + h * 64) * HLD + b_kk] = R.b[h];
  };
  auto advance = [&](It it) {
    It n = it;
    if (n.k >= K) return n;
    if (++n.ci >= nC) {
      n.ci = 0;
      ++n.k;
      while (n.k < K && !tapAny[n.k]) ++n.k;
    }
    return n;
  };
  const int li = lane & 15, kq = lane >> 4;
  auto compute = [&]() {
    bf16x8_t a[2], b[NF];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) a[mf] = *(const bf16x8_t*)&As[(wv * 32 + mf * 16 + li) * HLD + kq * 8];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) b[nf] = *(const bf16x8_t*)&Bs[(nf * 16 + li) * HLD + kq * 8];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
  };

  It i0 = {0, 0};
  while (i0.k < K && !tapAny[i0.k]) ++i0.k;
  It i1 = advance(i0);
  Regs r0, r1;
  if (i0.k < K) load_chunk(r0, i0);
  if (i1.k < K) load_chunk(r1, i1);
  while (i0.k < K) {
    store_chunk(r0);
    __syncthreads();
    It i2 = advance(i1);
    if (i2.k < K) load_chunk(r0, i2);
    compute();
    __syncthreads();
    if (i1.k >= K) break;
    store_chunk(r1);
    __syncthreads();
    It i3 = advance(i2);
    if (i3.k < K) load_chunk(r1, i3);
    compute();
    __syncthreads();
    i0 = i2;
    i1 = i3;
  }
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      int col = n0 + nf * 16 + li;
      if (col >= Cout) continue;
      float bv = bias ? bias[col] : 0.f;
      float sc = ep_scale ? ep_scale[col] : 1.f, sh = ep_shift ? ep_shift[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + wv * 32 + mf * 16 + kq * 4 + r;
        if (row < n_out) {
          float* p = Y + (size_t)row * ldy + col;
          float v = acc[mf][nf][r] + bv;
          if (ep_scale) v = v * sc + sh;                  // same fused frozen-BN epilogue as the fast kernels
          float rv = 0.f;
          if (ep_res) {
            if (io & ES_IO_R16) rv = __uint_as_float((uint32_t)((const unsigned short*)ep_res)[(size_t)row * ep_ldr + col] << 16);
            else rv = ep_res[(size_t)row * ep_ldr + col];
          }
          if (ep_act == 3) {                              // gate: pass v only where ep_res > 0 (fused ReLU backward)
            if (!(rv > 0.f)) v = 0.f;
          } else {
            if (ep_res) v += rv;
            if (ep_act) v = fmaxf(v, 0.f);
          }
          if (io & ES_IO_Y16) {                           // (Cout % 4 == 0: lanes li, li^1 are valid together) one 4-byte store per pair
            float vn = __shfl_xor(v, 1, 64);
            if (!(li & 1)) *(uint32_t*)((unsigned short*)Y + (size_t)row * ldy + col) = pack_bf16(v, vn);
          } else {
            *p = accumulate ? (*p + v) : v;
          }
        }
      }
    }
}

// Fast path of k_spconv_bf16: shapes where no bounds checks are needed (Cin % 32 == 0, Cout % BNT == 0, 16-byte
// aligned rows, 32-bit element offsets).  The PMC profile of the generic kernel showed 5.5 VALU + 3 SALU instructions
// per MFMA (64-bit address multiplies, per-element bounds tests, iterator bookkeeping): it was VALU-issue bound at 16 %
// MFMA utilisation.  Here the per-tap offsets are computed once per tap, the per-chunk work is add + load + cvt + store.
// XH = true: X is a bf16 row matrix (the "shadow" copy made by es_cast_rows_bf16) -> half the gather bytes and no
// conversion instructions; XH = false: f32 rows converted while staged.
// PP = true: ping-pong LDS buffers -- chunk c is computed from buffer c&1 while chunk c+1 is written into the other one, so
// the loop needs ONE workgroup barrier per 16-MFMA chunk instead of two and the LDS stores overlap the matrix
// instructions of the current chunk (46 KB LDS per workgroup, still 3 workgroups per CU).
// Tail of a tap-split launch (round 4: replaces the k_sum_splits launch behind every split convolution -- 70 launches on the
// dependent chain of an mv-3ddet step): the gridDim.z workgroups of an output tile have written their partial tiles to the
// workspace [slice][row][col]; the last one to arrive (es_last_block on the tile's ticket) adds them IN SLICE ORDER into Y --
// the summation order of k_sum_splits4, bit for bit, whichever workgroup happens to be last.  `tickets`: one zero-initialised
// unsigned int per tile (left at zero); the launcher hands it over through the unused ep_shift argument.
#define ES_SPLIT_TICKETS 1024
template <int BNT_>
__device__ inline void split_tail(const float* ws, unsigned int* tickets, int row0, int n0, int n_out, int Cout,
                                  float* __restrict__ Y, int ldy, int accumulate) {
  if (!es_last_block_release_only(tickets + blockIdx.y * gridDim.x + blockIdx.x, gridDim.z)) return;
  const int split = gridDim.z;
  constexpr int C4 = BNT_ / 4;
  const size_t tot = (size_t)n_out * Cout;
  const bool vec = ((ldy & 3) == 0) && ((((uintptr_t)Y) & 15) == 0);
  for (int e = threadIdx.x; e < BM * C4; e += blockDim.x) {
    const int row = row0 + e / C4, col = n0 + (e % C4) * 4;
    if (row >= n_out) break;
    const float* p0 = ws + (size_t)row * Cout + col;
    float4 sum = *(const float4*)p0;
    for (int z = 1; z < split; ++z) {
      float4 v = *(const float4*)(p0 + (size_t)z * tot);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    float* y = Y + (size_t)row * ldy + col;
    if (vec) {
      if (accumulate) { float4 y0 = *(const float4*)y; sum.x = y0.x + sum.x; sum.y = y0.y + sum.y; sum.z = y0.z + sum.z; sum.w = y0.w + sum.w; }
      *(float4*)y = sum;
    } else {
      y[0] = accumulate ? y[0] + sum.x : sum.x; y[1] = accumulate ? y[1] + sum.y : sum.y;
      y[2] = accumulate ? y[2] + sum.z : sum.z; y[3] = accumulate ? y[3] + sum.w : sum.w;
    }
  }
}

template <int BNT, bool XH, bool PP>
__global__ __launch_bounds__(256, 3) void k_spconv_bf16_fast(const void* __restrict__ Xv, int ldx,
                                                          const unsigned short* __restrict__ W,
                                                          const int* __restrict__ nbr, int n_out, int n_in, int K,
                                                          int Cin, int Cout, const float* __restrict__ bias,
                                                          float* __restrict__ Y, int ldy, int accumulate,
                                                          const float* __restrict__ ep_scale,
                                                          const float* __restrict__ ep_shift,
                                                          const float* __restrict__ ep_res, int ep_ldr, int ep_act, int io) {
  constexpr int NF = BNT / 16;
  constexpr int NB = BNT / 64;
  constexpr int NBUF = PP ? 2 : 1;
  // PP tiles are unpadded (64-B rows) with an XOR swizzle of the four 16-B granules of a row by ((row >> 2) & 3): the 16
  // rows a fragment read touches land in 16 distinct 16-B bank slots, and 2 x (8 + 8) KB + the map tile = 46 KB keeps
  // three workgroups on a CU (the padded 80-B rows would cost 55 KB: two)
  constexpr int LDP = PP ? HBK : HLD;
  __shared__ __attribute__((aligned(16))) unsigned short As[NBUF * BM * LDP];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[NBUF * BNT * LDP];
  __shared__ int nbrS[BM * MAXK];
  __shared__ int taps[32];
  __shared__ int nTaps;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  // (An XCD-aware row-tile order -- contiguous eighths of the tiles per XCD -- was measured: no gain, the halo rows
  // already hit in MALL; plain order kept.)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (io & ES_IO_WSHARE) {
    // tap-split launches are the small-row / wide-channel layers: a (column tile, tap slice) pair reads a weight slice of hundreds of
    // KB that EVERY row tile of the pair re-reads, and in launch order (x fastest, workgroup L on XCD L % 8) those row tiles sit on
    // different XCDs -- each private L2 fetches the slice again.  Here XCD c walks pairs c, c + 8, ... and inside a pair all row
    // tiles, so the slice is fetched into one L2 once (pairs past the last full group of eight keep the launch order).
    const int gx = gridDim.x, gy = gridDim.y, L = bx + gx * (by + gy * bz);
    const int full = ((gy * (int)gridDim.z) >> 3) << 3;
    if (L < full * gx) {
      const int j = L >> 3, pair = (j / gx) * 8 + (L & 7);
      bx = j % gx; by = pair % gy; bz = pair / gy;
    }
  }
  const int row0 = bx * BM, n0 = by * BNT;

  __shared__ int tapFlag[32];
  if (t < 32) tapFlag[t] = 0;
  __syncthreads();
  {                                   // kernel-map tile -> LDS: two threads per row, each walks half of the taps
    int r = t >> 1, kh = (K + 1) >> 1, k0 = (t & 1) * kh, k1 = min(K, k0 + kh);
    int j = row0 + r;
    const int* src = nbr ? nbr + (size_t)j * K : nullptr;
    for (int k = k0; k < k1; ++k) {
      int v = -1;
      if (j < n_out) v = src ? src[k] : (j < n_in ? j : -1);
      nbrS[r * K + k] = v;
      if (v >= 0) tapFlag[k] = 1;
    }
  }
  __syncthreads();
  if (t < 64) {                       // compact the taps that have at least one neighbour in this tile (one wave)
    int f = (t < K) ? tapFlag[t] : 0;
    unsigned long long m = __ballot(f);
    if (f) taps[__popcll(m & ((1ull << t) - 1ull))] = t;
    if (t == 0) nTaps = __popcll(m);
  }
  __syncthreads();
  // tap split (gridDim.z > 1): launches with too few tiles to fill the chip share the tap list among gridDim.z
  // workgroups which write their partial tiles to a workspace; the tile's last workgroup adds them in slice order (split_tail)
  const int nTall = nTaps;
  const int tBeg = (int)(((long long)nTall * bz) / gridDim.z);
  const int nT = (int)(((long long)nTall * (bz + 1)) / gridDim.z);

  f32x4 acc[2][NF];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nC = Cin / HBK;
  const int a_r = t >> 1, a_kk = (t & 1) * 16;
  const int b_n = t >> 2, b_kk = (t & 3) * 8;
  const int b_row = (n0 + b_n) * Cin + b_kk;              // element offset of this thread's weight piece inside a tap
  const int w_tap = Cout * Cin;
  const float* X = (const float*)Xv;
  const unsigned short* Xh = (const unsigned short*)Xv;
  struct Regs { float4 a[XH ? 2 : 4]; uint4 b[NB]; int valid; };   // XH: 2 x 8 bf16, else 4 x 4 f32
  struct It { int ti, ci, a_off, b_off, valid; };
  // Loads are UNCONDITIONAL (an absent neighbour reads row 0 and is zeroed when written to LDS, a chunk past the end
  // re-reads the last tap) so that the compiler can keep the second prefetched chunk in flight with a counted
  // s_waitcnt instead of vmcnt(0); the chunk stream is processed in pairs without a mid-loop exit so that the 64
  // accumulator registers stay in place (the first version of this loop moved them through v_accvgpr_mov).
  auto set_tap = [&](It& it) {
    int ti = it.ti < nT ? it.ti : (nT - 1);
    int k = taps[ti];
    int idx = nbrS[a_r * K + k];
    it.valid = (idx >= 0) && (it.ti < nT);
    it.a_off = (idx >= 0 ? idx : 0) * ldx + a_kk;
    it.b_off = k * w_tap + b_row;
  };
  auto advance = [&](It& it) {
    if (++it.ci >= nC) {
      it.ci = 0;
      ++it.ti;
      set_tap(it);
    }
  };
  auto load_chunk = [&](Regs& R, const It& it) {
    int c0 = it.ci * HBK;
    if (XH) {
      const float4* p = (const float4*)(Xh + it.a_off + c0);
      R.a[0] = p[0];
      R.a[1] = p[1];
    } else {
      const float4* p = (const float4*)(X + it.a_off + c0);
#pragma unroll
      for (int q = 0; q < (XH ? 2 : 4); ++q) R.a[q] = p[q];
    }
#pragma unroll
    for (int h = 0; h < NB; ++h) R.b[h] = *(const uint4*)(W + it.b_off + h * 64 * Cin + c0);
    R.valid = it.valid;
  };
  // swizzle key of a tile row: f(row) = (3 * (row >> 2)) & 3, i.e. 0, 3, 2, 1 by row quad.  ds_read_b128 is serviced in four
  // NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS table); the round-2 key (row >> 2) & 3
  // put two of a group's 16-B pieces on every bank slot (2-way conflict on every fragment read), this one none -- checked by
  // enumeration for both the documented groups and contiguous ones (tools/lds_conflicts.py)
  const int a_sw = PP ? (((a_r >> 2) * 3) & 3) : 0, b_sw = PP ? (((b_n >> 2) * 3) & 3) : 0;
  auto store_chunk = [&](const Regs& R, int buf = 0) {
    uint4* pa = (uint4*)&As[buf * BM * LDP + a_r * LDP + (PP ? 0 : a_kk)];
    uint4 v0, v1;
    if (XH) {                         // bit copies (a pointer cast here made the compiler keep R in scratch memory)
      v0 = make_uint4(__float_as_uint(R.a[0].x), __float_as_uint(R.a[0].y), __float_as_uint(R.a[0].z),
                      __float_as_uint(R.a[0].w));
      v1 = make_uint4(__float_as_uint(R.a[1].x), __float_as_uint(R.a[1].y), __float_as_uint(R.a[1].z),
                      __float_as_uint(R.a[1].w));
    } else {
      constexpr int H = XH ? 0 : 2;      // (a[2], a[3] only exist in the f32 layout)
      v0 = make_uint4(pack_bf16(R.a[0].x, R.a[0].y), pack_bf16(R.a[0].z, R.a[0].w), pack_bf16(R.a[1].x, R.a[1].y),
                      pack_bf16(R.a[1].z, R.a[1].w));
      v1 = make_uint4(pack_bf16(R.a[H].x, R.a[H].y), pack_bf16(R.a[H].z, R.a[H].w), pack_bf16(R.a[H + 1].x, R.a[H + 1].y),
                      pack_bf16(R.a[H + 1].z, R.a[H + 1].w));
    }
    if (!R.valid) v0 = v1 = make_uint4(0u, 0u, 0u, 0u);
    if (PP) {
      const int g0 = (t & 1) * 2;
      pa[(g0 ^ a_sw)] = v0;
      pa[((g0 + 1) ^ a_sw)] = v1;
#pragma unroll
      for (int h = 0; h < NB; ++h) *(uint4*)&Bs[buf * BNT * LDP + (b_n + h * 64) * LDP + (((t & 3) ^ b_sw) * 8)] = R.b[h];
    } else {
      pa[0] = v0;
      pa[1] = v1;
#pragma unroll
      for (int h = 0; h < NB; ++h) *(uint4*)&Bs[(b_n + h * 64) * HLD + b_kk] = R.b[h];
    }
  };
  const int li = lane & 15, kq = lane >> 4;
  const int f_sw = PP ? (((li >> 2) * 3) & 3) : 0;        // swizzle key of the fragment rows (row = 16 * j + li)
  const unsigned short* a_base = &As[(wv * 32 + li) * LDP + (kq ^ f_sw) * 8];
  const unsigned short* b_base = &Bs[li * LDP + (kq ^ f_sw) * 8];
  auto compute = [&](int buf = 0) {
    bf16x8_t a[2], b[NF];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) a[mf] = *(const bf16x8_t*)(a_base + buf * BM * LDP + mf * 16 * LDP);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) b[nf] = *(const bf16x8_t*)(b_base + buf * BNT * LDP + nf * 16 * LDP);
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
  };

  if (nT > tBeg) {
    It i0 = {tBeg, 0, 0, 0, 0};
    set_tap(i0);
    It i1 = i0;
    advance(i1);
    Regs r0, r1;
    load_chunk(r0, i0);
    load_chunk(r1, i1);
    const int npairs = ((nT - tBeg) * nC + 1) >> 1;
    if (PP) {
      store_chunk(r0, 0);             // chunk 0 -> buffer 0
      i0 = i1;
      advance(i0);
      load_chunk(r0, i0);             // chunk 2 in flight
      __syncthreads();
      for (int pr = 0; pr < npairs; ++pr) {
        store_chunk(r1, 1);           // chunk 2*pr + 1 -> buffer 1 (its last readers passed the barrier below)
        i1 = i0;
        advance(i1);
        load_chunk(r1, i1);           // chunk 2*pr + 3
        compute(0);                   // chunk 2*pr
        __syncthreads();
        store_chunk(r0, 0);           // chunk 2*pr + 2 -> buffer 0 (stored past the end: never computed)
        i0 = i1;
        advance(i0);
        load_chunk(r0, i0);           // chunk 2*pr + 4
        compute(1);                   // chunk 2*pr + 1 (a zeroed A tile when the chunk count is odd)
        __syncthreads();
      }
    } else {
      for (int pr = 0; pr < npairs; ++pr) {
        store_chunk(r0);
        __syncthreads();
        i0 = i1;
        advance(i0);                    // chunk 2*pr + 2
        load_chunk(r0, i0);
        compute();
        __syncthreads();
        store_chunk(r1);                // (a zeroed A tile when the chunk count is odd)
        __syncthreads();
        i1 = i0;
        advance(i1);                    // chunk 2*pr + 3
        load_chunk(r1, i1);
        compute();
        __syncthreads();
      }
    }
  }
  // epilogue: optional fused frozen-BN affine (+ residual) (+ ReLU) of the 2-D backbone
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      int col = n0 + nf * 16 + li;
      float bv = bias ? bias[col] : 0.f;
      float sc = ep_scale ? ep_scale[col] : 1.f, sh = (ep_shift && gridDim.z == 1) ? ep_shift[col] : 0.f;   // (split: ep_shift = tickets)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + wv * 32 + mf * 16 + kq * 4 + r;
        if (row < n_out) {
          float* p = Y + (size_t)row * ldy + col;
          if (gridDim.z > 1) {
            float v = acc[mf][nf][r] + (bz == 0 ? bv : 0.f);
            // deterministic tap split (the only kind since round 3): partial sums go to the workspace
            const_cast<float*>(ep_res)[((size_t)bz * n_out + row) * Cout + col] = v;
          } else {
            float v = acc[mf][nf][r] + bv;
            if (ep_scale) v = v * sc + sh;
            float rv = 0.f;                               // residual / gate operand: f32 or bf16 rows
            if (ep_res) {
              if (io & ES_IO_R16) rv = __uint_as_float((uint32_t)((const unsigned short*)ep_res)[(size_t)row * ep_ldr + col] << 16);
              else rv = ep_res[(size_t)row * ep_ldr + col];
            }
            if (ep_act == 3) {                            // gate: pass v only where ep_res > 0 (fused ReLU backward)
              if (!(rv > 0.f)) v = 0.f;
            } else {
              if (ep_res) v += rv;
              if (ep_act) v = fmaxf(v, 0.f);
            }
            if (io & ES_IO_Y16) {                         // bf16 activation rows: lanes (li, li^1) share one 4-byte store
              float vn = __shfl_xor(v, 1, 64);
              if (!(li & 1)) *(uint32_t*)((unsigned short*)Y + (size_t)row * ldy + col) = pack_bf16(v, vn);
            } else {
              *p = accumulate ? (*p + v) : v;
            }
          }
        }
      }
    }
  if (gridDim.z > 1 && ep_shift) split_tail<BNT>(ep_res, (unsigned int*)ep_shift, row0, n0, n_out, Cout, Y, ldy, accumulate);
}

// ------------------------------------------------------------------ LDS-DMA variant of the fast kernel (round 3, late)
// Same contract as k_spconv_bf16_fast<BNT, true, *> (bf16 input rows, kernel-map gather, optional tap split and fused
// epilogue).  What changes is how the operands reach the matrix pipes.  The SQ / LDS arithmetic of the ping-pong kernel per
// wave and 16-MFMA chunk: 4 ds_write_b128 (13 LDS cycles each: MI355X_MICROARCH.md, LDS table) + 10 ds_read_b128 (4 each,
// 8 with the 2-way conflict of the old swizzle key) = 92 .. 132 LDS cycles against 80 cycles of MFMA issue -- the kernel was
// LDS-bound, not MFMA- or HBM-bound (24 % MFMA busy in profiles/r3_mfma_util.txt).  Here
//   * both tiles are staged by global_load_lds_dwordx4 (the per-lane SOURCE address does the gather and the swizzle, the
//     destination is wave-base + lane * 16): no staging registers, no conversion or packing VALU work, no ds_write at all;
//     an absent neighbour reads a 16-byte zero granule in global memory;
//   * the four waves tile the 128 x BNT output 2 x 2 (64 x BNT/2 each): 8 fragment reads per 16 MFMAs instead of 10;
//   * the row swizzle key is conflict-free for the documented ds_read_b128 lane groups (tools/lds_conflicts.py);
//   * KB = 2 stages 64 channels per chunk (one barrier per 32 MFMAs; 78 KB of LDS -> 2 workgroups per CU), KB = 1 keeps 32
//     (46 KB -> 3 workgroups per CU).
// One barrier per chunk: the DMA of chunk c + 1 is issued right after the barrier that retires chunk c's DMA and runs under
// chunk c's MFMAs.  All LDS of the kernel is ONE __shared__ array (a second object makes hipcc wait vmcnt(0) before every
// fragment read of a DMA pipeline: cdna_hip_programming.md 5, trap (a)).
__device__ __attribute__((aligned(16))) unsigned short g_zero_granule[8];

// NBUF = 3 (EXPERIMENTAL, option 10 value 3, not yet run on hardware): a ring of three LDS buffers with TWO chunks of DMA in
// flight -- the wait before the barrier is a counted vmcnt (this wave's pieces of the NEXT chunk stay outstanding) and the barrier
// is the raw s_barrier (a __syncthreads() would drain the DMA queue: cdna_hip_programming.md 5, "glds with > 1 tile in flight").
template <int BNT, int KB, int NBUF = 2>
__global__ __launch_bounds__(256, (KB == 1 && NBUF == 2) ? 3 : 2) void k_spconv_bf16_dma(
    const unsigned short* __restrict__ Xh, int ldx, const unsigned short* __restrict__ W, const int* __restrict__ nbr,
    int n_out, int n_in, int K, int Cin, int Cout, const float* __restrict__ bias, float* __restrict__ Y, int ldy,
    int accumulate, const float* __restrict__ ep_scale, const float* __restrict__ ep_shift,
    const float* __restrict__ ep_res, int ep_ldr, int ep_act, int io) {
  constexpr int G = 4 * KB;                      // 16-byte granules per tile row
  constexpr int RB = 64 * KB;                    // bytes per tile row
  constexpr int BKT = 32 * KB;                   // channels per chunk
  constexpr int A_BYTES = BM * RB, B_BYTES = BNT * RB;
  constexpr int NA = BM * G / 256, NBI = BNT * G / 256;    // DMA instructions per thread and chunk (A, B)
  constexpr int NFW = BNT / 32;                  // 16-wide column fragments per wave (wave tile 64 x BNT / 2)
  constexpr int OFF_B = NBUF * A_BYTES, OFF_MAP = OFF_B + NBUF * B_BYTES, OFF_TAPS = OFF_MAP + BM * MAXK * 4,
                OFF_FLAG = OFF_TAPS + 32 * 4, OFF_NT = OFF_FLAG + 32 * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_NT + 16];
  int* const nbrS = (int*)(smem + OFF_MAP);
  int* const taps = (int*)(smem + OFF_TAPS);
  int* const tapFlag = (int*)(smem + OFF_FLAG);
  int* const nTapsP = (int*)(smem + OFF_NT);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int bz = blockIdx.z;
  const int row0 = blockIdx.x * BM, n0 = blockIdx.y * BNT;

  if (t < 32) tapFlag[t] = 0;
  __syncthreads();
  {                                   // kernel-map tile -> LDS (as in k_spconv_bf16_fast)
    int r = t >> 1, kh = (K + 1) >> 1, k0 = (t & 1) * kh, k1 = min(K, k0 + kh);
    int j = row0 + r;
    const int* src = nbr ? nbr + (size_t)j * K : nullptr;
    for (int k = k0; k < k1; ++k) {
      int v = -1;
      if (j < n_out) v = src ? src[k] : (j < n_in ? j : -1);
      nbrS[r * K + k] = v;
      if (v >= 0) tapFlag[k] = 1;
    }
  }
  __syncthreads();
  if (t < 64) {
    int f = (t < K) ? tapFlag[t] : 0;
    unsigned long long m = __ballot(f);
    if (f) taps[__popcll(m & ((1ull << t) - 1ull))] = t;
    if (t == 0) *nTapsP = __popcll(m);
  }
  __syncthreads();
  const int nTall = *nTapsP;
  const int tBeg = (int)(((long long)nTall * bz) / gridDim.z);
  const int nT = (int)(((long long)nTall * (bz + 1)) / gridDim.z);

  f32x4 acc[4][NFW];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NFW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // DMA piece e = (j * 4 + wv) * 64 + lane of a tile lands at byte e * 16: tile row e / G, slot e % G, and holds the
  // row's granule slot ^ key(row)
  auto key = [](int row) { return KB == 1 ? (((row >> 2) * 3) & 3) : ((row >> 1) & 7); };
  int a_row[NA], a_g8[NA], b_off[NBI];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int e = (j * 4 + wv) * 64 + lane, row = e / G, slot = e % G;
    a_row[j] = row;
    a_g8[j] = (slot ^ key(row)) * 8;
  }
#pragma unroll
  for (int j = 0; j < NBI; ++j) {
    int e = (j * 4 + wv) * 64 + lane, row = e / G, slot = e % G;
    b_off[j] = (n0 + row) * Cin + (slot ^ key(row)) * 8;
  }
  const int nC = Cin / BKT;
  const int w_tap = Cout * Cin;
  int it_ti = tBeg, it_c0 = 0, tap_off = 0;
  int a_off[NA];                                  // element offset of this lane's granule in X (< 0: absent neighbour)
  auto set_tap = [&]() {
    int k = taps[it_ti < nT ? it_ti : (nT - 1)];
    tap_off = k * w_tap;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int idx = nbrS[a_row[j] * K + k];
      a_off[j] = idx >= 0 ? idx * ldx + a_g8[j] : -1;
    }
  };
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned short* p = a_off[j] >= 0 ? (Xh + a_off[j] + it_c0) : g_zero_granule;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + buf * A_BYTES + (j * 4 + wv) * 1024),
                                       16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NBI; ++j) {
      const unsigned short* p = W + tap_off + b_off[j] + it_c0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + OFF_B + buf * B_BYTES + (j * 4 + wv) * 1024),
                                       16, 0, 0);
    }
    it_c0 += BKT;                                 // next chunk of the stream
    if (it_c0 >= Cin) {
      it_c0 = 0;
      ++it_ti;
      set_tap();
    }
  };
  const int li = lane & 15, kq = lane >> 4;
  const int f_key = key(li);                      // tile rows of a fragment are 16 * x + li: the key depends on li only
  const unsigned char* a_frag = smem + (wr * 64 + li) * RB;
  const unsigned char* b_frag = smem + OFF_B + (wc * (BNT / 2) + li) * RB;
  auto compute = [&](int buf) {
#pragma unroll
    for (int h = 0; h < KB; ++h) {
      const int so = ((h * 4 + kq) ^ f_key) * 16;
      bf16x8_t a[4], b[NFW];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) a[mf] = *(const bf16x8_t*)(a_frag + buf * A_BYTES + mf * 16 * RB + so);
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) b[nf] = *(const bf16x8_t*)(b_frag + buf * B_BYTES + nf * 16 * RB + so);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
    }
  };

  if (nT > tBeg) {
    const int nch = (nT - tBeg) * nC;
    set_tap();
    issue(0);
    if (NBUF == 2) {
      for (int c = 0; c < nch; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of chunk c have landed ...
        __syncthreads();                                      // ... everybody's have, and chunk c - 1 has been consumed
        if (c + 1 < nch) issue((c + 1) & 1);
        compute(c & 1);
      }
    } else {
      constexpr int PER = NA + NBI;                           // DMA instructions of one chunk per wave
      if (nch > 1) issue(1);
      int b0 = 0, b2 = 2;                                     // buffers of chunk c and of chunk c + 2
      for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");   // chunk c landed, c + 1 may be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // everybody's pieces of chunk c; chunk c - 1 has been consumed
        if (c + 2 < nch) issue(b2);                           // into the buffer chunk c - 1 was read from
        compute(b0);
        b0 = b0 == 2 ? 0 : b0 + 1;
        b2 = b2 == 2 ? 0 : b2 + 1;
      }
    }
  }
  // epilogue: as k_spconv_bf16_fast, for the 2 x 2 wave tiling
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      int col = n0 + wc * (BNT / 2) + nf * 16 + li;
      float bv = bias ? bias[col] : 0.f;
      float sc = ep_scale ? ep_scale[col] : 1.f, sh = (ep_shift && gridDim.z == 1) ? ep_shift[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + wr * 64 + mf * 16 + kq * 4 + r;
        if (row < n_out) {
          float* p = Y + (size_t)row * ldy + col;
          if (gridDim.z > 1) {
            float v = acc[mf][nf][r] + (bz == 0 ? bv : 0.f);
            const_cast<float*>(ep_res)[((size_t)bz * n_out + row) * Cout + col] = v;
          } else {
            float v = acc[mf][nf][r] + bv;
            if (ep_scale) v = v * sc + sh;
            float rv = 0.f;
            if (ep_res) {
              if (io & ES_IO_R16) rv = __uint_as_float((uint32_t)((const unsigned short*)ep_res)[(size_t)row * ep_ldr + col] << 16);
              else rv = ep_res[(size_t)row * ep_ldr + col];
            }
            if (ep_act == 3) {
              if (!(rv > 0.f)) v = 0.f;
            } else {
              if (ep_res) v += rv;
              if (ep_act) v = fmaxf(v, 0.f);
            }
            if (io & ES_IO_Y16) {
              float vn = __shfl_xor(v, 1, 64);
              if (!(li & 1)) *(uint32_t*)((unsigned short*)Y + (size_t)row * ldy + col) = pack_bf16(v, vn);
            } else {
              *p = accumulate ? (*p + v) : v;
            }
          }
        }
      }
    }
  if (gridDim.z > 1 && ep_shift) split_tail<BNT>(ep_res, (unsigned int*)ep_shift, row0, n0, n_out, Cout, Y, ldy, accumulate);
}

// ------------------------------------------------------------------ K = 1 on the identity map: a streaming row GEMM
// Y[r, :] = epilogue(X[r, :] @ W): every 1x1 convolution of the 2-D backbone (with its frozen-BN / residual / ReLU or
// gated-dgrad epilogue), the head's output GEMMs, the 1x1 shortcuts of MinkResNet, every Linear layer and all of their data
// gradients.  These launches are bound by the f32 rows they stream (C_in + C_out (+ residual) floats per row, a few flops per
// byte), so the kernel is built around memory transactions instead of the gather machinery of k_spconv_bf16_fast:
//   * A fragments come straight from global memory in MFMA layout (lane = row, 8 consecutive channels: 32-B pieces, four lanes
//     complete a 128-B line), converted to bf16 in registers, register double-buffered -- no LDS, no map, no tap lists;
//   * the weight slab (NT x 32 bf16) is ping-ponged through LDS: one barrier per k-step;
//   * the epilogue goes through LDS so that global traffic is whole rows: the accumulator layout (lane = column) would
//     write 64-B pieces; staged, every half-wave reads / writes 512 contiguous bytes of Y (and of the residual).
// XH: the input rows are bf16 (activation storage of the image backbone, round 3) -- the A fragment is then one 16-byte load
// with no conversion.  io bit 0: Y rows are bf16 (4 channels = one 8-byte store), bit 1: the ep_res rows are bf16.
template <int NT, bool XH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_rowgemm_bf16(const void* __restrict__ Xv, int ldx,
                                                      const unsigned short* __restrict__ W, int n_out, int n_in, int Cin,
                                                      int Cout, const float* __restrict__ bias, float* __restrict__ Y,
                                                      int ldy, int accumulate, const float* __restrict__ ep_scale,
                                                      const float* __restrict__ ep_shift,
                                                      const float* __restrict__ ep_res, int ep_ldr, int ep_act, int io) {
  const float* X = (const float*)Xv;
  const unsigned short* Xh = (const unsigned short*)Xv;
  constexpr int NF = NT / 16, SLD = NT + 4;
  constexpr int B_BYTES = 2 * NT * HLD * 2, S_BYTES = 4 * 16 * SLD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[B_BYTES > S_BYTES ? B_BYTES : S_BYTES];
  unsigned short* Bs = (unsigned short*)smem;             // [2][NT][HLD]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * BM + wv * 32, n0 = blockIdx.y * NT;
  const int n_rows = min(n_out, n_in);                    // identity map: rows past the input are empty
  f32x4 acc[2][NF];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ns = (Cin + HBK - 1) / HBK;                     // C_in % 8 == 0; a ragged last slab is zero-filled
  // (scalars, not arrays: arrays captured by the loader lambdas ended up in scratch memory)
  float4 a00, a01, a10, a11;
  uint4 bg0, bg1;
  const int rowA0 = row0 + li, rowA1 = row0 + 16 + li;
  const float* pa0 = X + (size_t)min(rowA0, n_rows - 1) * ldx + kq * 8;
  const float* pa1 = X + (size_t)min(rowA1, n_rows - 1) * ldx + kq * 8;
  const unsigned short* ph0 = Xh + (size_t)min(rowA0, n_rows - 1) * ldx + kq * 8;
  const unsigned short* ph1 = Xh + (size_t)min(rowA1, n_rows - 1) * ldx + kq * 8;
  const bool va0 = rowA0 < n_rows, va1 = rowA1 < n_rows;
  const int bc0 = t >> 2, bq = t & 3;                     // weight slab: NT x 32 bf16 = NT*4 16-byte granules
  const unsigned short* pb0 = W + (size_t)(n0 + (bc0 < NT ? bc0 : 0)) * Cin + bq * 8;
  const unsigned short* pb1 = W + (size_t)(n0 + (NT > 64 ? 64 : 0) + bc0) * Cin + bq * 8;
#define RG_LOAD(s_)                                                                    \
  do {                                                                                 \
    const bool ka_ = (s_) * HBK + kq * 8 < Cin, kb_ = (s_) * HBK + bq * 8 < Cin;       \
    a00 = a01 = a10 = a11 = make_float4(0.f, 0.f, 0.f, 0.f);                           \
    bg0 = bg1 = make_uint4(0u, 0u, 0u, 0u);                                            \
    if (XH) {                       /* 8 bf16 = one 16-byte piece, kept as raw bits in a00 / a10 */     \
      if (va0 && ka_) a00 = *(const float4*)(ph0 + (s_) * HBK);                        \
      if (va1 && ka_) a10 = *(const float4*)(ph1 + (s_) * HBK);                        \
    } else {                                                                           \
      if (va0 && ka_) {                                                                \
        const float4* q0_ = (const float4*)(pa0 + (s_) * HBK);                         \
        a00 = q0_[0]; a01 = q0_[1];                                                    \
      }                                                                                \
      if (va1 && ka_) {                                                                \
        const float4* q1_ = (const float4*)(pa1 + (s_) * HBK);                         \
        a10 = q1_[0]; a11 = q1_[1];                                                    \
      }                                                                                \
    }                                                                                  \
    if (kb_ && bc0 < NT) bg0 = *(const uint4*)(pb0 + (s_) * HBK);                      \
    if (NT > 64 && kb_) bg1 = *(const uint4*)(pb1 + (s_) * HBK);                       \
  } while (0)
#define RG_STORE_B(buf_)                                                               \
  do {                                                                                 \
    if (bc0 < NT) *(uint4*)&Bs[((buf_) * NT + bc0) * HLD + bq * 8] = bg0;              \
    if (NT > 64) *(uint4*)&Bs[((buf_) * NT + 64 + bc0) * HLD + bq * 8] = bg1;          \
  } while (0)
  if (n_rows <= 0) return;
  // The epilogue's second operand (residual / gate rows, or Y itself when accumulating) is requested NOW, in the layout the
  // final stores use: these launches move a few hundred bytes per row and are latency-bound unless every thread keeps
  // many 16-B loads in flight (Little: 8 TB/s x ~2 us = 64 KB per CU).
  constexpr int NI = (16 * NT / 4) / 64;
  const float* pre = ep_res ? ep_res : (accumulate ? Y : nullptr);
  const int pre_ld = ep_res ? ep_ldr : ldy;
  float4 pf[2][NI];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int f = i * 64 + lane, rr = f / (NT / 4), c4 = f - rr * (NT / 4);
      int row = row0 + mf * 16 + rr;
      pf[mf][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pre && row < n_out) {
        if (ep_res && (io & ES_IO_R16))
          pf[mf][i] = bf16x4_to_f32(*(const uint2*)((const unsigned short*)ep_res + (size_t)row * pre_ld + n0 + c4 * 4));
        else
          pf[mf][i] = *(const float4*)(pre + (size_t)row * pre_ld + n0 + c4 * 4);
      }
    }
  RG_LOAD(0);
  RG_STORE_B(0);
  __syncthreads();
  for (int s = 0; s < ns; ++s) {
    uint4 pk0, pk1;
    if (XH) {
      pk0 = make_uint4(__float_as_uint(a00.x), __float_as_uint(a00.y), __float_as_uint(a00.z), __float_as_uint(a00.w));
      pk1 = make_uint4(__float_as_uint(a10.x), __float_as_uint(a10.y), __float_as_uint(a10.z), __float_as_uint(a10.w));
    } else {
      pk0 = make_uint4(pack_bf16(a00.x, a00.y), pack_bf16(a00.z, a00.w), pack_bf16(a01.x, a01.y), pack_bf16(a01.z, a01.w));
      pk1 = make_uint4(pack_bf16(a10.x, a10.y), pack_bf16(a10.z, a10.w), pack_bf16(a11.x, a11.y), pack_bf16(a11.z, a11.w));
    }
    bf16x8_t fa0 = __builtin_bit_cast(bf16x8_t, pk0), fa1 = __builtin_bit_cast(bf16x8_t, pk1);
    if (s + 1 < ns) RG_LOAD(s + 1);
    const unsigned short* Bc = Bs + (s & 1) * NT * HLD;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      bf16x8_t b = *(const bf16x8_t*)&Bc[(nf * 16 + li) * HLD + kq * 8];
      acc[0][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, b, acc[0][nf], 0, 0, 0);
      acc[1][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, b, acc[1][nf], 0, 0, 0);
    }
    if (s + 1 < ns) RG_STORE_B((s + 1) & 1);
    __syncthreads();
  }
#undef RG_LOAD
#undef RG_STORE_B
  // epilogue through LDS: 16 rows x NT columns per wave and pass
  float* stage = (float*)smem + wv * 16 * SLD;
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    if (mf) __syncthreads();
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[(kq * 4 + r) * SLD + nf * 16 + li] = acc[mf][nf][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (16 * NT / 4) / 64; ++i) {
      int f = i * 64 + lane, rr = f / (NT / 4), c4 = f - rr * (NT / 4);
      int row = row0 + mf * 16 + rr, col = n0 + c4 * 4;
      if (row >= n_out) continue;
      float4 v = *(const float4*)&stage[rr * SLD + c4 * 4];
      const float4 q = pf[mf][i];
      auto ep1 = [&](float x, float r, int c) -> float {
        x += bias ? bias[c] : 0.f;
        if (ep_scale) x = x * ep_scale[c] + (ep_shift ? ep_shift[c] : 0.f);
        if (ep_act == 3) {                                // gate: pass x only where the residual operand is > 0
          if (!(r > 0.f)) x = 0.f;
        } else {
          if (ep_res) x += r;
          if (ep_act) x = fmaxf(x, 0.f);
        }
        return x;
      };
      float o0 = ep1(v.x, q.x, col), o1 = ep1(v.y, q.y, col + 1), o2 = ep1(v.z, q.z, col + 2), o3 = ep1(v.w, q.w, col + 3);
      if (io & ES_IO_Y16) {                               // bf16 activation rows: 4 channels = 8 bytes (never accumulated)
        *(uint2*)((unsigned short*)Y + (size_t)row * ldy + col) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
        continue;
      }
      float4* py = (float4*)(Y + (size_t)row * ldy + col);
      if (accumulate) {
        float4 y0 = ep_res ? *py : q;                     // (residual AND accumulation: Y was not prefetched)
        o0 += y0.x; o1 += y0.y; o2 += y0.z; o3 += y0.w;
      }
      *py = make_float4(o0, o1, o2, o3);
    }
  }
}

// Second generation of the row GEMM (round 3, late).  tools/bench_rowgemm.py showed the kernel above at 0.8 .. 1.6 TB/s on the
// OUTPUT-heavy launches (the 16->64 / 32->128 / 64->256 expansion convolutions of the image backbone with their bf16
// residual, the head's 128->320 GEMM) where an elementwise pass over the same bytes runs at 5 .. 7 TB/s, and at 2 .. 5 TB/s on
// the input-heavy ones: the time goes into the epilogue (64 ds_write_b32 + 16 ds_read_b128 per thread, three workgroup
// barriers, 8-byte global accesses for bf16 rows, 2 waves per SIMD because 64 registers hold the prefetched residual).
// Here the MFMA operands are SWAPPED -- the weight fragment is the A operand, the input rows the B operand -- so that a lane
// ends up with output CHANNELS (kq * 4 + r) of ONE row (li) instead of rows of one channel; with the weight rows of a
// fragment pair permuted (fragment 2p takes channels 32p + 8q + {0..3}, fragment 2p + 1 channels 32p + 8q + {4..7}) a lane
// owns 8 consecutive channels of its row: one 16-byte store for bf16 rows, two for f32 rows, straight from the accumulators --
// no LDS staging, no barrier after the k loop, the residual is prefetched in the same layout (16 bytes per 8 channels).
// MT (experimental, ES_GEN_FUSED): the 8 taps of a generative transposed convolution in ONE launch -- forward: blockIdx.z = tap
// (its own weight slice and output column block); data gradient: the taps are extra steps of the k loop (tap t reads the
// input columns t * a_tap .. and the weight slice t * w_tap ..).  MT = false compiles to the kernel described above.
struct RowGemmTaps { int z_w, z_y_bytes, taps, a_tap, w_tap; };
// PRE = false (NT = 320: the head's 128 -> 320 output GEMM as ONE column tile, the input rows read once instead of five times): no
// second epilogue operand, so its 16 bytes x 2 NP prefetch registers per row are not allocated.
template <int NT, bool XH, bool MT = false, bool PRE = true>
__global__ __launch_bounds__(256) void k_rowgemm2_bf16(const void* __restrict__ Xv, int ldx,
                                                       const unsigned short* __restrict__ W, int n_out, int n_in, int Cin,
                                                       int Cout, const float* __restrict__ bias, float* __restrict__ Y,
                                                       int ldy, int accumulate, const float* __restrict__ ep_scale,
                                                       const float* __restrict__ ep_shift,
                                                       const float* __restrict__ ep_res, int ep_ldr, int ep_act, int io,
                                                       RowGemmTaps tp) {
  static_assert(NT % 32 == 0, "fragment pairs");
  if (MT) {
    W += (size_t)blockIdx.z * tp.z_w;
    Y = (float*)((char*)Y + (size_t)blockIdx.z * tp.z_y_bytes);
  }
  const float* X = (const float*)Xv;
  const unsigned short* Xh = (const unsigned short*)Xv;
  constexpr int NF = NT / 16, NP = NT / 32, NBP = (NT + 63) / 64, NPF = PRE ? NP : 1;
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * NT * HLD];       // [2][NT][HLD]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * BM + wv * 32, n0 = blockIdx.y * NT;
  const int n_rows = min(n_out, n_in);
  if (n_rows <= 0) return;
  f32x4 acc[2][NF];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ns = (Cin + HBK - 1) / HBK;
  float4 a00, a01, a10, a11;
  uint4 bg[NBP];
  const int rowA0 = row0 + li, rowA1 = row0 + 16 + li;
  const float* pa0 = X + (size_t)min(rowA0, n_rows - 1) * ldx + kq * 8;
  const float* pa1 = X + (size_t)min(rowA1, n_rows - 1) * ldx + kq * 8;
  const unsigned short* ph0 = Xh + (size_t)min(rowA0, n_rows - 1) * ldx + kq * 8;
  const unsigned short* ph1 = Xh + (size_t)min(rowA1, n_rows - 1) * ldx + kq * 8;
  const int abl = io >> 8;                                // dev ablation: 1 no stores, 2 no input-row loads, 4 no residual loads
  const bool va0 = rowA0 < n_rows && !(abl & 2), va1 = rowA1 < n_rows && !(abl & 2);
  const int bc0 = t >> 2, bq = t & 3;
  const unsigned short* pb0 = W + (size_t)(n0 + (bc0 < NT ? bc0 : 0)) * Cin + bq * 8;       // + 64 j rows for pass j
#define RG2_LOAD(s_)                                                                   \
  do {                                                                                 \
    const int tap_ = MT ? (s_) / ns : 0, ss_ = (s_) - tap_ * ns;                        \
    const int ao_ = ss_ * HBK + (MT ? tap_ * tp.a_tap : 0), bo_ = ss_ * HBK + (MT ? tap_ * tp.w_tap : 0); \
    const bool ka_ = ss_ * HBK + kq * 8 < Cin, kb_ = ss_ * HBK + bq * 8 < Cin;         \
    a00 = a01 = a10 = a11 = make_float4(0.f, 0.f, 0.f, 0.f);                           \
    _Pragma("unroll") for (int j_ = 0; j_ < NBP; ++j_) bg[j_] = make_uint4(0u, 0u, 0u, 0u); \
    if (XH) {                                                                          \
      if (va0 && ka_) a00 = *(const float4*)(ph0 + ao_);                                    \
      if (va1 && ka_) a10 = *(const float4*)(ph1 + ao_);                                    \
    } else {                                                                           \
      if (va0 && ka_) {                                                                \
        const float4* q0_ = (const float4*)(pa0 + ao_);                                     \
        a00 = q0_[0]; a01 = q0_[1];                                                    \
      }                                                                                \
      if (va1 && ka_) {                                                                \
        const float4* q1_ = (const float4*)(pa1 + ao_);                                     \
        a10 = q1_[0]; a11 = q1_[1];                                                    \
      }                                                                                \
    }                                                                                  \
    if (kb_ && bc0 < NT) bg[0] = *(const uint4*)(pb0 + bo_);                                \
    _Pragma("unroll") for (int j_ = 1; j_ < NBP; ++j_)                                 \
      if (kb_) bg[j_] = *(const uint4*)(pb0 + (size_t)64 * j_ * Cin + bo_);            \
  } while (0)
#define RG2_STORE_B(buf_)                                                              \
  do {                                                                                 \
    if (bc0 < NT) *(uint4*)&Bs[((buf_) * NT + bc0) * HLD + bq * 8] = bg[0];            \
    _Pragma("unroll") for (int j_ = 1; j_ < NBP; ++j_)                                 \
      *(uint4*)&Bs[((buf_) * NT + 64 * j_ + bc0) * HLD + bq * 8] = bg[j_];             \
  } while (0)
  // this lane's output: row (mf * 16 + li), channels n0 + 32 p + 8 kq + {0 .. 7}
  const int rowE0 = row0 + li, rowE1 = row0 + 16 + li;
  const int colE = n0 + kq * 8;
  // second operand of the epilogue (residual / gate rows, or Y itself when accumulating), requested before the k loop
  const bool r16 = (io & ES_IO_R16) != 0;
  const float* pre = ep_res ? ep_res : (accumulate ? Y : nullptr);
  const int pre_ld = ep_res ? ep_ldr : ldy;
  float4 pf[2][NPF][2];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int p2 = 0; p2 < NPF; ++p2) {
      const int row = mf ? rowE1 : rowE0, col = colE + 32 * p2;
      pf[mf][p2][0] = pf[mf][p2][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (PRE && pre && row < n_out && !((io >> 8) & 4)) {
        if (ep_res && r16) {                              // 8 bf16 = 16 bytes, kept as raw bits in [0]
          pf[mf][p2][0] = *(const float4*)((const unsigned short*)ep_res + (size_t)row * pre_ld + col);
        } else {
          const float4* q = (const float4*)(pre + (size_t)row * pre_ld + col);
          pf[mf][p2][0] = q[0];
          pf[mf][p2][1] = q[1];
        }
      }
    }
  RG2_LOAD(0);
  RG2_STORE_B(0);
  __syncthreads();
  // weight row this lane reads for fragment nf (as the A operand: lane li = output channel within the fragment)
  const int wrow = (li >> 2) * 8 + (li & 3);              // + 32 * (nf >> 1) + 4 * (nf & 1)
  const int nst = MT ? ns * tp.taps : ns;
  for (int s = 0; s < nst; ++s) {
    uint4 pk0, pk1;
    if (XH) {
      pk0 = make_uint4(__float_as_uint(a00.x), __float_as_uint(a00.y), __float_as_uint(a00.z), __float_as_uint(a00.w));
      pk1 = make_uint4(__float_as_uint(a10.x), __float_as_uint(a10.y), __float_as_uint(a10.z), __float_as_uint(a10.w));
    } else {
      pk0 = make_uint4(pack_bf16(a00.x, a00.y), pack_bf16(a00.z, a00.w), pack_bf16(a01.x, a01.y), pack_bf16(a01.z, a01.w));
      pk1 = make_uint4(pack_bf16(a10.x, a10.y), pack_bf16(a10.z, a10.w), pack_bf16(a11.x, a11.y), pack_bf16(a11.z, a11.w));
    }
    bf16x8_t fa0 = __builtin_bit_cast(bf16x8_t, pk0), fa1 = __builtin_bit_cast(bf16x8_t, pk1);
    if (s + 1 < nst) RG2_LOAD(s + 1);
    const unsigned short* Bc = Bs + (s & 1) * NT * HLD;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      bf16x8_t b = *(const bf16x8_t*)&Bc[(32 * (nf >> 1) + 4 * (nf & 1) + wrow) * HLD + kq * 8];
      acc[0][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, fa0, acc[0][nf], 0, 0, 0);
      acc[1][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, fa1, acc[1][nf], 0, 0, 0);
    }
    if (s + 1 < nst) RG2_STORE_B((s + 1) & 1);
    if (s + 1 < nst) __syncthreads();
  }
#undef RG2_LOAD
#undef RG2_STORE_B
  // epilogue straight from the accumulators (no early exits inside the unrolled loops: a `continue` here kept pf[][][] in
  // scratch memory -- 272 bytes per lane, and the kernel at half its speed)
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const int row = mf ? rowE1 : rowE0;
    const bool row_ok = row < n_out;
#pragma unroll
    for (int p2 = 0; p2 < NP; ++p2) {
      const int col = colE + 32 * p2;
      float v[8], q[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[mf][2 * p2][r]; v[4 + r] = acc[mf][2 * p2 + 1][r]; }
      const float4 h0 = pf[mf][PRE ? p2 : 0][0], h1 = pf[mf][PRE ? p2 : 0][1];
      if (ep_res && r16) {
        const uint32_t u[4] = {__float_as_uint(h0.x), __float_as_uint(h0.y), __float_as_uint(h0.z), __float_as_uint(h0.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) { q[2 * e] = __uint_as_float(u[e] << 16); q[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
      } else {
        q[0] = h0.x; q[1] = h0.y; q[2] = h0.z; q[3] = h0.w; q[4] = h1.x; q[5] = h1.y; q[6] = h1.z; q[7] = h1.w;
      }
      // per-channel epilogue constants: 8 consecutive channels = two 16-byte loads each (L1 / L2 hits)
      float bb[8], sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { bb[e] = 0.f; sc[e] = 1.f; sh[e] = 0.f; }
      if (bias) {
        const float4 t0 = *(const float4*)(bias + col), t1 = *(const float4*)(bias + col + 4);
        bb[0] = t0.x; bb[1] = t0.y; bb[2] = t0.z; bb[3] = t0.w; bb[4] = t1.x; bb[5] = t1.y; bb[6] = t1.z; bb[7] = t1.w;
      }
      if (ep_scale) {
        const float4 t0 = *(const float4*)(ep_scale + col), t1 = *(const float4*)(ep_scale + col + 4);
        sc[0] = t0.x; sc[1] = t0.y; sc[2] = t0.z; sc[3] = t0.w; sc[4] = t1.x; sc[5] = t1.y; sc[6] = t1.z; sc[7] = t1.w;
        if (ep_shift) {
          const float4 u0 = *(const float4*)(ep_shift + col), u1 = *(const float4*)(ep_shift + col + 4);
          sh[0] = u0.x; sh[1] = u0.y; sh[2] = u0.z; sh[3] = u0.w; sh[4] = u1.x; sh[5] = u1.y; sh[6] = u1.z; sh[7] = u1.w;
        }
      }
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[e] + bb[e];
        if (ep_scale) x = x * sc[e] + sh[e];
        if (ep_act == 3) {
          if (!(q[e] > 0.f)) x = 0.f;
        } else {
          if (ep_res) x += q[e];
          if (ep_act) x = fmaxf(x, 0.f);
        }
        o[e] = x;
      }
      if (io & ES_IO_Y16) {
        if (row_ok && !((abl & 1) && o[0] != 12345.678f))   // (ablation: value-dependent so that the arithmetic stays)
          *(uint4*)((unsigned short*)Y + (size_t)row * ldy + col) =
              make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
      } else if (row_ok) {
        float4* py = (float4*)(Y + (size_t)row * ldy + col);
        if (accumulate) {
          float4 y0 = h0, y1 = h1;
          if (ep_res) { y0 = py[0]; y1 = py[1]; }         // (residual AND accumulation: Y was not prefetched)
          o[0] += y0.x; o[1] += y0.y; o[2] += y0.z; o[3] += y0.w; o[4] += y1.x; o[5] += y1.y; o[6] += y1.z; o[7] += y1.w;
        }
        py[0] = make_float4(o[0], o[1], o[2], o[3]);
        py[1] = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
  }
}

// ------------------------------------------------------------------ expansion convolutions of the image backbone (round 6)
// Bottleneck.conv3 / downsample (1x1, C -> 4 C with C = 16 / 32 / 64, frozen BN, (+ residual) (+ ReLU), bf16 rows in and out) on 10^5 .. 10^6 pixel rows
// are OUTPUT streams: 32 .. 128 input bytes and 128 .. 512 output (+ as many residual) bytes per row.  On k_rowgemm2_bf16 a 128-row workgroup pays
// three dependent latencies (operands -> LDS -> barrier; epilogue constants; stores) for 36 KB of traffic: 1.4 - 1.6 TB/s (340 us for 16 -> 64 on
// 3.5 M rows; profiles/r6m_slowest_launches_grounding.txt).  Here nothing goes through LDS: a wave keeps its 64 output channels' weight fragments and
// BN constants in registers for the whole launch and walks 16-row tiles -- input fragment and residual of the NEXT tile in flight under the MFMAs,
// epilogue and 16-byte stores of the current one.  Waves of a workgroup split the output channels (C = 32: two groups, 64: four).  Same MFMA sequence
// and epilogue arithmetic as k_rowgemm2_bf16.
template <int CIN>
__global__ __launch_bounds__(256) void k_expand_bf16(const unsigned short* __restrict__ X, int ldx, const unsigned short* __restrict__ W, int n,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const unsigned short* __restrict__ R, int ldr, int act, unsigned short* __restrict__ Y,
                                                     int ldy) {
  constexpr int COUT = 4 * CIN, WS = COUT / 64, RT = 4 / WS;      // channel groups per workgroup, row tiles per workgroup and round
  constexpr int KS = CIN <= 32 ? 1 : CIN / 32;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int cw0 = (wv % WS) * 64;                                // this wave's first output channel
  const int wrow = (li >> 2) * 8 + (li & 3);
  bf16x8_t b[4][KS];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 32 + kq * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (k < CIN) v = *(const uint4*)(W + (size_t)(cw0 + 32 * (nf >> 1) + 4 * (nf & 1) + wrow) * CIN + k);
      b[nf][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  float sc[2][8], sh[2][8];
#pragma unroll
  for (int p2 = 0; p2 < 2; ++p2) {
    const int col = cw0 + 32 * p2 + kq * 8;
    const float4 s0 = *(const float4*)(scale + col), s1 = *(const float4*)(scale + col + 4);
    const float4 h0 = *(const float4*)(shift + col), h1 = *(const float4*)(shift + col + 4);
    sc[p2][0] = s0.x; sc[p2][1] = s0.y; sc[p2][2] = s0.z; sc[p2][3] = s0.w; sc[p2][4] = s1.x; sc[p2][5] = s1.y; sc[p2][6] = s1.z; sc[p2][7] = s1.w;
    sh[p2][0] = h0.x; sh[p2][1] = h0.y; sh[p2][2] = h0.z; sh[p2][3] = h0.w; sh[p2][4] = h1.x; sh[p2][5] = h1.y; sh[p2][6] = h1.z; sh[p2][7] = h1.w;
  }
  const int tiles = (n + 15) >> 4;
  const int stride = gridDim.x * RT;
  uint4 an[KS], rn[2];
  auto load = [&](int tile) {
    const int row = tile * 16 + li;
    const bool ok = tile < tiles && row < n;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 32 + kq * 8;
      an[ks] = (ok && k < CIN) ? *(const uint4*)(X + (size_t)row * ldx + k) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2)
      rn[p2] = (ok && R) ? *(const uint4*)(R + (size_t)row * ldr + cw0 + 32 * p2 + kq * 8) : make_uint4(0u, 0u, 0u, 0u);
  };
  int tile = blockIdx.x * RT + wv / WS;
  load(tile);
  for (; tile < tiles; tile += stride) {
    uint4 ac[KS], rc[2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ac[ks] = an[ks];
    rc[0] = rn[0]; rc[1] = rn[1];
    load(tile + stride);
    f32x4 acc[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t a = __builtin_bit_cast(bf16x8_t, ac[ks]);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nf][ks], a, acc[nf], 0, 0, 0);
    }
    const int row = tile * 16 + li;
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      float o[8];
      const uint32_t u[4] = {rc[p2].x, rc[p2].y, rc[p2].z, rc[p2].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = (e < 4 ? acc[2 * p2][e] : acc[2 * p2 + 1][e - 4]) + 0.f;
        x = x * sc[p2][e] + sh[p2][e];
        if (R) x += (e & 1) ? __uint_as_float(u[e >> 1] & 0xffff0000u) : __uint_as_float(u[e >> 1] << 16);
        if (act) x = fmaxf(x, 0.f);
        o[e] = x;
      }
      if (row < n)
        *(uint4*)(Y + (size_t)row * ldy + cw0 + 32 * p2 + kq * 8) =
            make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
    }
  }
}

// ------------------------------------------------------------------ small-row linear layers (round 6)
// The grounding decoder (ground_transformer/decoder.py:91-179) runs ~ 190 K = 1 GEMMs per step on 3 072 query rows (256 -> 256, 256 <-> 2 048)
// and a few on 396 text rows: 48 workgroups of the 128-row kernel above, each walking its reduction in 32-channel steps with ONE step of
// prefetch -- 8 .. 64 dependent memory latencies per launch (18.6 us for 256 -> 256, 72 us for 2 048 -> 256; profiles/r6m_slowest_launches_grounding.txt).
// Here a workgroup owns 64 rows x 64 output channels and moves a WHOLE 256-channel stage per memory latency: 24 independent 16-byte loads per
// thread (input rows f32, weights bf16) -> bf16 tiles in LDS (row pitch 528 B: the 16 rows of a fragment read sit on 16 distinct bank groups) ->
// 8 x (1 + 4) fragment reads and 32 MFMAs per wave; the next stage's loads are in flight under them.  Operands swapped like k_rowgemm2_bf16 (a
// lane ends up with 4 consecutive channels of one row: 16-byte stores), same accumulation order -> same bits as that kernel.
#define LS_KS 256
#define LS_LD (LS_KS + 8)
__global__ __launch_bounds__(256) void k_lin_small(const float* __restrict__ X, int ldx, const unsigned short* __restrict__ W, int n, int Cin,
                                                   int Cout, const float* __restrict__ bias, float* __restrict__ Y, int ldy, int accumulate) {
  __shared__ __attribute__((aligned(16))) unsigned short As[64 * LS_LD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[64 * LS_LD];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int lr = t >> 2, lk = (t & 3) * 64;                   // this thread's tile row (input row / output channel) and its 64 reduction indices
  const bool rv = row0 + lr < n;
  const float* pa = X + (size_t)(rv ? row0 + lr : 0) * ldx + lk;
  const unsigned short* pb = W + (size_t)(n0 + lr) * Cin + lk;
  float4 ra[16];
  uint4 rb[8];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) ra[i] = (rv && k0 + lk + 4 * i < Cin) ? *(const float4*)(pa + k0 + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = (k0 + lk + 8 * i < Cin) ? *(const uint4*)(pb + k0 + 8 * i) : make_uint4(0u, 0u, 0u, 0u);
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      *(uint4*)&As[lr * LS_LD + lk + 8 * i] = make_uint4(pack_bf16(ra[2 * i].x, ra[2 * i].y), pack_bf16(ra[2 * i].z, ra[2 * i].w),
                                                          pack_bf16(ra[2 * i + 1].x, ra[2 * i + 1].y), pack_bf16(ra[2 * i + 1].z, ra[2 * i + 1].w));
      *(uint4*)&Bs[lr * LS_LD + lk + 8 * i] = rb[i];
    }
  };
  f32x4 acc[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // weight row this lane reads for fragment nf as the FIRST operand: fragment pair p = nf >> 1 takes channels 32 p + 8 q + {0..3} (even nf) /
  // {4..7} (odd nf), so that a lane owns 8 consecutive channels per pair (k_rowgemm2_bf16's permutation)
  const int wrow = (li >> 2) * 8 + (li & 3);
  load(0);
  for (int k0 = 0; k0 < Cin; k0 += LS_KS) {
    store();
    __syncthreads();
    if (k0 + LS_KS < Cin) load(k0 + LS_KS);
    const int ks_n = min(LS_KS, Cin - k0 + 31 & ~31) / 32;
    for (int ks = 0; ks < ks_n; ++ks) {
      const bf16x8_t a = *(const bf16x8_t*)&As[(wv * 16 + li) * LS_LD + ks * 32 + kq * 8];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const bf16x8_t b = *(const bf16x8_t*)&Bs[(32 * (nf >> 1) + 4 * (nf & 1) + wrow) * LS_LD + ks * 32 + kq * 8];
        acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[nf], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const int row = row0 + wv * 16 + li;
  if (row < n) {
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      const int col = n0 + 32 * p2 + kq * 8;
      float o[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { o[r] = acc[2 * p2][r]; o[4 + r] = acc[2 * p2 + 1][r]; }
      if (bias) {
        const float4 b0 = *(const float4*)(bias + col), b1 = *(const float4*)(bias + col + 4);
        o[0] += b0.x; o[1] += b0.y; o[2] += b0.z; o[3] += b0.w; o[4] += b1.x; o[5] += b1.y; o[6] += b1.z; o[7] += b1.w;
      }
      float4* py = (float4*)(Y + (size_t)row * ldy + col);
      if (accumulate) {
        const float4 y0 = py[0], y1 = py[1];
        o[0] += y0.x; o[1] += y0.y; o[2] += y0.z; o[3] += y0.w; o[4] += y1.x; o[5] += y1.y; o[6] += y1.z; o[7] += y1.w;
      }
      py[0] = make_float4(o[0], o[1], o[2], o[3]);
      py[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
}

// 1 if (shape, alignment) is served by the fast kernels -- the host uses it to decide whether a bf16 shadow of X pays
extern "C" int es_spconv_bf16_is_fast(int n_in, int ldx, int K, int Cin, int Cout) {
  return (Cin % HBK == 0) && (ldx % 8 == 0) && (Cout % 64 == 0) && ((long long)n_in * ldx < (1ll << 31)) &&
         ((long long)K * Cout * Cin < (1ll << 31));
}

// tap split of under-filled launches: how many workgroups share one output tile's tap list
static int split_factor(int n_out, int K, int Cout) {
  int wgs = (Cout % 128 == 0) ? es_cdiv(n_out, BM) * (Cout / 128) : es_cdiv(n_out, BM) * (Cout / 64);
  int split = 1;
  while (split < 8 && wgs * split < ES_OPT_FWD_SPLIT_WGS && split * 3 <= K) split *= 2;
  return split;
}
// Y (+)= sum over the split slices of the partial sums, in slice order (bit-reproducible, unlike f32 atomics)
__global__ void k_sum_splits(const float* __restrict__ ws, int split, int n_out, int Cout, float* __restrict__ Y, int ldy,
                             int accumulate) {
  size_t tot = (size_t)n_out * Cout;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    float s = ws[e];
    for (int z = 1; z < split; ++z) s += ws[(size_t)z * tot + e];
    size_t row = e / Cout;
    float* p = Y + row * ldy + (e - row * Cout);
    *p = accumulate ? (*p + s) : s;
  }
}
__global__ void k_sum_splits4(const float4* __restrict__ ws, int split, size_t tot4, int C4, float* __restrict__ Y, int ldy,
                              int accumulate) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot4; e += (size_t)gridDim.x * blockDim.x) {
    float4 s = ws[e];
    for (int z = 1; z < split; ++z) {
      float4 v = ws[(size_t)z * tot4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    size_t row = e / C4;
    float4* p = (float4*)(Y + row * ldy + (e - row * C4) * 4);
    if (accumulate) { float4 y0 = *p; s.x = y0.x + s.x; s.y = y0.y + s.y; s.z = y0.z + s.z; s.w = y0.w + s.w; }
    *p = s;
  }
}
extern "C" size_t es_spconv_split_workspace_floats(int n_out, int K, int Cin, int Cout) {
  if (K <= 1 || Cin % HBK != 0 || Cout % 64 != 0 || n_out <= 0) return 0;
  int split = split_factor(n_out, K, Cout);
  return split > 1 ? (size_t)ES_SPLIT_TICKETS + (size_t)split * n_out * Cout : 0;     // [tile tickets | partial tiles]
}

static int spconv_fwd_bf16_impl(const void* Xv, int x_is_bf16, int ldx, const void* W_bf16, const int* nbr, int n_out,
                                int n_in, int K, int Cin, int Cout, const float* bias, float* Y, int ldy,
                                int accumulate, const float* ep_scale, const float* ep_shift, const float* ep_res,
                                int ep_ldr, int ep_act, void* stream, float* ws = nullptr, size_t ws_floats = 0,
                                int y_half = 0, int r_half = 0) {
  if (n_out <= 0 || Cout <= 0) return 0;
  if (K > MAXK) return -2;
  int io = (y_half ? ES_IO_Y16 : 0) | ((r_half && ep_res) ? ES_IO_R16 : 0) | (y_half & 0xff00);   // (bits 8-15: dev ablation switches of k_rowgemm2_bf16, tools/bench_rowgemm.py)
  if (y_half && (accumulate || (ldy % 4) || (Cout % 4) || (((uintptr_t)Y) & 7))) return -7;   // bf16 rows: 8-byte stores
  if ((io & ES_IO_R16) && ((ep_ldr % 4) || (((uintptr_t)ep_res) & 7))) return -7;
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* Wh = (const unsigned short*)W_bf16;
  const float* X = (const float*)Xv;
  bool fast = (Cin % HBK == 0) && (ldx % (x_is_bf16 ? 8 : 4) == 0) && ((((uintptr_t)Xv) & 15) == 0) &&
              ((((uintptr_t)Wh) & 15) == 0) && ((long long)n_in * ldx < (1ll << 31)) &&
              ((long long)K * Cout * Cin < (1ll << 31)) && (Cout % 64 == 0);
  dim3 g128(es_cdiv(n_out, BM), Cout / 128), g64(es_cdiv(n_out, BM), Cout / 64);
  if (ES_OPT_ROWGEMM && K == 1 && nbr == nullptr && ep_act != 7 && n_in >= n_out && (Cin % 8 == 0) &&
      (Cout % 16 == 0) && (ldx % (x_is_bf16 ? 8 : 4) == 0) && (ldy % 4 == 0) &&
      (((((uintptr_t)Xv) | ((uintptr_t)Wh)) & 15) == 0) && ((((uintptr_t)Y) & (y_half ? 7 : 15)) == 0) &&
      (!ep_res || ((ep_ldr % 4 == 0) && ((((uintptr_t)ep_res) & ((io & ES_IO_R16) ? 7 : 15)) == 0)))) {
    // 128-column tiles only when they still fill the chip: a launch of fewer than ES_OPT_RG128_MIN_WGS such workgroups (the grounder's
    // 256 -> 256 linears on 3 072 query rows: 48) runs 64-column tiles -- twice the workgroups, half the serial k loop per output
    // column (round 6 A/B with the threshold at 256: grounding 53.35 vs 53.03, mv-3ddet 23.62 vs 23.56, occupancy 34.69 vs 34.76: noise; default 0 = off)
    const bool wide = Cout % 128 == 0 && Cin >= ES_OPT_RG128_MIN_CIN &&
                      (long long)es_cdiv(n_out, BM) * (Cout / 128) >= ES_OPT_RG128_MIN_WGS;
    // the image backbone's C -> 4 C expansion layers on bf16 rows: register-resident weights, no LDS (k_expand_bf16)
    if (ES_OPT_EXPAND && x_is_bf16 && y_half && !(y_half & 0xff00) && ep_scale && ep_shift && !bias && !accumulate && (ep_act == 0 || ep_act == 1) &&
        (!ep_res || r_half) && Cout == 4 * Cin && (Cin == 16 || Cin == 32 || Cin == 64) && n_out >= ES_OPT_EXPAND && n_in >= n_out &&
        (ldx % 8 == 0) && (ldy % 8 == 0) && (!ep_res || (ep_ldr % 8 == 0 && ((((uintptr_t)ep_res) & 15) == 0))) && ((((uintptr_t)Y) & 15) == 0) &&
        (((((uintptr_t)ep_scale) | ((uintptr_t)ep_shift)) & 15) == 0) && (long long)n_out * (ldy > ldx ? ldy : ldx) < (1ll << 31)) {
      const int rt = 4 / (Cout / 64), tiles = es_cdiv(n_out, 16), want = es_cdiv(tiles, rt);
      dim3 ge(want < ES_OPT_EXPAND_WGS ? want : ES_OPT_EXPAND_WGS);
#define EX_LAUNCH(C_) hipLaunchKernelGGL((k_expand_bf16<C_>), ge, dim3(256), 0, st, (const unsigned short*)Xv, ldx, Wh, n_out, ep_scale, ep_shift, \
                                         (const unsigned short*)ep_res, ep_ldr, ep_act, (unsigned short*)Y, ldy)
      if (Cin == 16) EX_LAUNCH(16);
      else if (Cin == 32) EX_LAUNCH(32);
      else EX_LAUNCH(64);
#undef EX_LAUNCH
      ES_CHECK_LAUNCH();
      return 0;
    }
    // few rows (the grounding decoder's linears): 64 x 64 tiles moving a 256-channel stage per memory latency
    if (ES_OPT_LIN_SMALL && !x_is_bf16 && !y_half && !ep_scale && !ep_res && !ep_act && (Cin % 8 == 0) && (Cout % 64 == 0) && Cin >= 64 &&
        (long long)es_cdiv(n_out, BM) * es_cdiv(Cout, 128) < ES_OPT_LIN_SMALL && n_in >= n_out && (ldy % 4 == 0) && ((((uintptr_t)Y) & 15) == 0) &&
        (!bias || ((((uintptr_t)bias) & 15) == 0))) {
      hipLaunchKernelGGL(k_lin_small, dim3(es_cdiv(n_out, 64), Cout / 64), dim3(256), 0, st, X, ldx, Wh, n_out, Cin, Cout, bias, Y, ldy, accumulate);
      ES_CHECK_LAUNCH();
      return 0;
    }
    // Cout = 320 (fcaf3d_head.py: the level's class / box / centerness outputs as one GEMM), no second epilogue operand: ONE 320-column tile
    // (profiles/r6j_rows_ab.txt: 264 -> 224 us on 352 k rows, 52 -> 40 on 60 k, 11.5 -> 19 on 7.5 k: one workgroup per CU needs rows)
    const bool whole = ES_OPT_RG320 && Cout == 320 && n_out >= 16384 && !ep_res && !accumulate && !y_half && ES_OPT_ROWGEMM2 && (ldy % 4 == 0) && ((((uintptr_t)Y) & 15) == 0);
    const int nt = whole ? 320 : wide ? 128 : (Cout % 64 == 0) ? 64 : (Cout % 32 == 0) ? 32 : 16;
    dim3 g(es_cdiv(n_out, BM), Cout / nt);
#define RG_LAUNCH(NT_)                                                                                              \
    do {                                                                                                            \
      if (x_is_bf16)                                                                                                \
        hipLaunchKernelGGL((k_rowgemm_bf16<NT_, true>), g, dim3(256), 0, st, Xv, ldx, Wh, n_out, n_in, Cin, Cout, bias, Y, ldy, \
                           accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);                              \
      else                                                                                                          \
        hipLaunchKernelGGL((k_rowgemm_bf16<NT_, false>), g, dim3(256), 0, st, Xv, ldx, Wh, n_out, n_in, Cin, Cout, bias, Y, ldy, \
                           accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);                              \
    } while (0)
#define RG2_LAUNCH(NT_)                                                                                             \
    do {                                                                                                            \
      if (x_is_bf16)                                                                                                \
        hipLaunchKernelGGL((k_rowgemm2_bf16<NT_, true>), g, dim3(256), 0, st, Xv, ldx, Wh, n_out, n_in, Cin, Cout, bias, Y, ldy, \
                           accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io, RowGemmTaps{0, 0, 1, 0, 0});   \
      else                                                                                                          \
        hipLaunchKernelGGL((k_rowgemm2_bf16<NT_, false>), g, dim3(256), 0, st, Xv, ldx, Wh, n_out, n_in, Cin, Cout, bias, Y, ldy, \
                           accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io, RowGemmTaps{0, 0, 1, 0, 0});  \
    } while (0)
    // second-generation kernel: 16-byte epilogue accesses need 8-channel alignment of every row matrix it touches
    const bool g2 = ES_OPT_ROWGEMM2 && nt >= 32 && (ldy % (y_half ? 8 : 4) == 0) && ((((uintptr_t)Y) & 15) == 0) &&
                    (!ep_res || ((ep_ldr % ((io & ES_IO_R16) ? 8 : 4) == 0) && ((((uintptr_t)ep_res) & 15) == 0)));
    if (whole) {
      if (x_is_bf16)
        hipLaunchKernelGGL((k_rowgemm2_bf16<320, true, false, false>), g, dim3(256), 0, st, Xv, ldx, Wh, n_out, n_in, Cin, Cout, bias, Y, ldy,
                           0, ep_scale, ep_shift, (const float*)nullptr, 0, ep_act, io, RowGemmTaps{0, 0, 1, 0, 0});
      else
        hipLaunchKernelGGL((k_rowgemm2_bf16<320, false, false, false>), g, dim3(256), 0, st, Xv, ldx, Wh, n_out, n_in, Cin, Cout, bias, Y, ldy,
                           0, ep_scale, ep_shift, (const float*)nullptr, 0, ep_act, io, RowGemmTaps{0, 0, 1, 0, 0});
    } else if (g2 && nt == 128) RG2_LAUNCH(128);
    else if (g2 && nt == 64) RG2_LAUNCH(64);
    else if (g2) RG2_LAUNCH(32);
    else if (nt == 128) RG_LAUNCH(128);
    else if (nt == 64) RG_LAUNCH(64);
    else if (nt == 32) RG_LAUNCH(32);
    else RG_LAUNCH(16);
#undef RG2_LAUNCH
#undef RG_LAUNCH
    ES_CHECK_LAUNCH();
    return 0;
  }
  int det_split = 0;
  if (fast && !(ep_scale || ep_res || ep_act) && K > 1 && !y_half) {
    // too few workgroups for 256 CUs: split the tap list over gridDim.z (partial sums through the caller's workspace)
    int split = split_factor(n_out, K, Cout);
    const int tiles = (Cout % 128 == 0) ? g128.x * g128.y : g64.x * g64.y;
    if (split > 1 && ws != nullptr && ws_floats >= (size_t)ES_SPLIT_TICKETS + (size_t)split * n_out * Cout && tiles <= ES_SPLIT_TICKETS &&
        ((((uintptr_t)ws) & 15) == 0)) {
      det_split = split;                   // deterministic: partial tiles to the workspace, added in slice order by the tile's last
      ep_shift = ES_OPT_SPLIT_FOLD ? ws : nullptr;   // workgroup (split_tail); the head of the workspace holds the tile tickets (zero on entry / exit)
      ep_res = ws + ES_SPLIT_TICKETS;
      ep_act = 7;
      g128.z = g64.z = split;
      if (ES_OPT_WSHARE && !ES_OPT_SPLIT_FOLD) io |= ES_IO_WSHARE;
    }                                      // (without a workspace the launch keeps one workgroup per tile: no f32 atomics)
  }
  if (fast && ES_OPT_DMA && x_is_bf16 && Cin >= ES_OPT_DMA_MIN_CIN) {
    const unsigned short* Xh = (const unsigned short*)Xv;
    const bool kb2 = (ES_OPT_DMA == 2) && (Cin % 64 == 0);
#define DMA_LAUNCH(BNT_, KB_, grid_)                                                                                  \
    hipLaunchKernelGGL((k_spconv_bf16_dma<BNT_, KB_>), grid_, dim3(256), 0, st, Xh, ldx, Wh, nbr, n_out, n_in, K, Cin, Cout, \
                       bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io)
    if (ES_OPT_DMA == 3) {                                  // experimental three-buffer ring, 32-channel chunks
      if (Cout % 128 == 0)
        hipLaunchKernelGGL((k_spconv_bf16_dma<128, 1, 3>), g128, dim3(256), 0, st, Xh, ldx, Wh, nbr, n_out, n_in, K, Cin, Cout,
                           bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
      else
        hipLaunchKernelGGL((k_spconv_bf16_dma<64, 1, 3>), g64, dim3(256), 0, st, Xh, ldx, Wh, nbr, n_out, n_in, K, Cin, Cout,
                           bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
    } else if (Cout % 128 == 0) { if (kb2) DMA_LAUNCH(128, 2, g128); else DMA_LAUNCH(128, 1, g128); }
    else                 { if (kb2) DMA_LAUNCH(64, 2, g64);   else DMA_LAUNCH(64, 1, g64); }
#undef DMA_LAUNCH
  } else if (fast && ES_OPT_PINGPONG) {
    if (x_is_bf16 && Cout % 128 == 0)
      hipLaunchKernelGGL((k_spconv_bf16_fast<128, true, true>), g128, dim3(256), 0, st, Xv, ldx, Wh, nbr, n_out, n_in, K, Cin,
                         Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
    else if (x_is_bf16)
      hipLaunchKernelGGL((k_spconv_bf16_fast<64, true, true>), g64, dim3(256), 0, st, Xv, ldx, Wh, nbr, n_out, n_in, K, Cin,
                         Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
    else if (Cout % 128 == 0)
      hipLaunchKernelGGL((k_spconv_bf16_fast<128, false, true>), g128, dim3(256), 0, st, Xv, ldx, Wh, nbr, n_out, n_in, K, Cin,
                         Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
    else
      hipLaunchKernelGGL((k_spconv_bf16_fast<64, false, true>), g64, dim3(256), 0, st, Xv, ldx, Wh, nbr, n_out, n_in, K, Cin,
                         Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
  } else if (fast && x_is_bf16 && Cout % 128 == 0) {
    hipLaunchKernelGGL((k_spconv_bf16_fast<128, true, false>), g128, dim3(256), 0, st, Xv, ldx, Wh, nbr, n_out, n_in, K, Cin,
                       Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
  } else if (fast && x_is_bf16) {
    hipLaunchKernelGGL((k_spconv_bf16_fast<64, true, false>), g64, dim3(256), 0, st, Xv, ldx, Wh, nbr, n_out, n_in, K, Cin,
                       Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
  } else if (fast && Cout % 128 == 0) {
    hipLaunchKernelGGL((k_spconv_bf16_fast<128, false, false>), g128, dim3(256), 0, st, Xv, ldx, Wh, nbr, n_out, n_in, K, Cin,
                       Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
  } else if (fast) {
    hipLaunchKernelGGL((k_spconv_bf16_fast<64, false, false>), g64, dim3(256), 0, st, Xv, ldx, Wh, nbr, n_out, n_in, K, Cin,
                       Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io);
  } else if (Cout >= 128) {
    hipLaunchKernelGGL(k_spconv_bf16<128>, dim3(es_cdiv(n_out, BM), es_cdiv(Cout, 128)), dim3(256), 0, st, X, ldx, Wh,
                       nbr, n_out, n_in, K, Cin, Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, x_is_bf16, io);
  } else {
    hipLaunchKernelGGL(k_spconv_bf16<64>, dim3(es_cdiv(n_out, BM), es_cdiv(Cout, 64)), dim3(256), 0, st, X, ldx, Wh,
                       nbr, n_out, n_in, K, Cin, Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, x_is_bf16, io);
  }
  ES_CHECK_LAUNCH();
  if (det_split && !ES_OPT_SPLIT_FOLD) {       // A/B path (es_set_option 16 = 0): the reduction as a second launch
    const float* part = ws + ES_SPLIT_TICKETS;
    if ((Cout % 4 == 0) && (ldy % 4 == 0) && (((((uintptr_t)part) | ((uintptr_t)Y)) & 15) == 0)) {
      size_t tot4 = (size_t)n_out * (Cout / 4);
      int g = es_cdiv((long long)tot4, 256);
      hipLaunchKernelGGL(k_sum_splits4, dim3(g > 8192 ? 8192 : g), dim3(256), 0, st, (const float4*)part, det_split, tot4,
                         Cout / 4, Y, ldy, accumulate);
    } else {
      int g = es_cdiv((long long)n_out * Cout, 256);
      hipLaunchKernelGGL(k_sum_splits, dim3(g > 4096 ? 4096 : g), dim3(256), 0, st, part, det_split, n_out, Cout, Y, ldy, accumulate);
    }
    ES_CHECK_LAUNCH();
  }
  return 0;
}
// the same with a caller-provided workspace (es_spconv_split_workspace_floats): under-filled launches that split their tap
// list then reduce the partial sums in a fixed order instead of f32 atomics -> bit-reproducible forward / dgrad
extern "C" int es_spconv_fwd_bf16_ws(const void* Xv, int x_is_bf16, int ldx, const void* W_bf16, const int* nbr, int n_out,
                                     int n_in, int K, int Cin, int Cout, const float* bias, float* Y, int ldy,
                                     int accumulate, float* ws, size_t ws_floats, void* stream) {
  return spconv_fwd_bf16_impl(Xv, x_is_bf16, ldx, W_bf16, nbr, n_out, n_in, K, Cin, Cout, bias, Y, ldy, accumulate,
                              nullptr, nullptr, nullptr, 0, 0, stream, ws, ws_floats);
}
extern "C" int es_spconv_fwd_bf16(const void* Xv, int x_is_bf16, int ldx, const void* W_bf16, const int* nbr, int n_out,
                                  int n_in, int K, int Cin, int Cout, const float* bias, float* Y, int ldy,
                                  int accumulate, void* stream) {
  return spconv_fwd_bf16_impl(Xv, x_is_bf16, ldx, W_bf16, nbr, n_out, n_in, K, Cin, Cout, bias, Y, ldy, accumulate,
                              nullptr, nullptr, nullptr, 0, 0, stream);
}
// forward with the frozen-BN affine (+ residual) (+ ReLU) fused into the epilogue:  Y = act((X*W) * scale + shift + res)
extern "C" int es_spconv_fwd_bf16_affine(const void* Xv, int ldx, const void* W_bf16, const int* nbr, int n_out,
                                         int n_in, int K, int Cin, int Cout, const float* scale, const float* shift,
                                         const float* res, int ldr, int act, float* Y, int ldy, void* stream) {
  return spconv_fwd_bf16_impl(Xv, 0, ldx, W_bf16, nbr, n_out, n_in, K, Cin, Cout, nullptr, Y, ldy, 0, scale, shift, res,
                              ldr, act, stream);
}

// the fused conv + frozen-BN (+ residual) (+ ReLU) / gated data-gradient launch with per-operand storage kinds: x_half,
// res_half, y_half non-zero -> that row matrix is bf16 (ld in elements).  The image backbone keeps its ACTIVATIONS in bf16
// (round 3): forward launches read and write bf16 rows, the gated data-gradient launches read the bf16 activation as their
// gate and move f32 gradients.  Y bf16 is never accumulated into.
extern "C" int es_spconv_fwd_bf16_io(const void* Xv, int x_half, int ldx, const void* W_bf16, const int* nbr, int n_out,
                                     int n_in, int K, int Cin, int Cout, const float* scale, const float* shift,
                                     const void* res, int res_half, int ldr, int act, void* Y, int y_half, int ldy,
                                     void* stream) {
  return spconv_fwd_bf16_impl(Xv, x_half, ldx, W_bf16, nbr, n_out, n_in, K, Cin, Cout, nullptr, (float*)Y, ldy, 0, scale, shift,
                              (const float*)res, ldr, act, stream, nullptr, 0, y_half, res_half);
}

// MinkowskiGenerativeConvolutionTranspose(k = 2, s = 2) on f32 rows (fcaf3d_head.py up-blocks): y[8 i + t] = x[i] @ w[t].  The
// eight taps used to be eight K = 1 launches (and eight accumulating data-gradient launches) of a few dozen workgroups each
// on the coarse levels; here ONE launch with gridDim.z = 8 (forward) / one launch whose k loop walks the taps (data gradient).
// Return 1: shape / alignment not served (the caller issues the per-tap launches).  Experimental (ES_GEN_FUSED=1).
template <bool DGRAD>
static int gen_transpose_launch(const float* X, int ldx, const unsigned short* Wh, int n, int Kred, int Ncol, float* Y, int ldy,
                                int accumulate, RowGemmTaps tp, int gz, hipStream_t st) {
  if (n <= 0) return 0;
  if ((Ncol % 32) || (Kred % 8) || (ldx % 4) || (ldy % 4) || ((((uintptr_t)X) | ((uintptr_t)Wh) | ((uintptr_t)Y)) & 15) ||
      ((tp.z_y_bytes | (tp.a_tap * 4)) & 15) || ((tp.w_tap | tp.z_w) % 8))
    return 1;
  const int nt = (Ncol % 128 == 0) ? 128 : (Ncol % 64 == 0) ? 64 : 32;
  dim3 g(es_cdiv(n, BM), Ncol / nt, gz);
#define GT_LAUNCH(NT_)                                                                                                   \
  hipLaunchKernelGGL((k_rowgemm2_bf16<NT_, false, true>), g, dim3(256), 0, st, (const void*)X, ldx, Wh, n, n, Kred, Ncol,  \
                     (const float*)nullptr, Y, ldy, accumulate, (const float*)nullptr, (const float*)nullptr,             \
                     (const float*)nullptr, 0, 0, 0, tp)
  if (nt == 128) GT_LAUNCH(128);
  else if (nt == 64) GT_LAUNCH(64);
  else GT_LAUNCH(32);
#undef GT_LAUNCH
  ES_CHECK_LAUNCH();
  return 0;
}
extern "C" int es_gen_transpose_fwd_bf16(const float* X, int ldx, const void* Wt_bf16, int n, int Cin, int Cout, float* Y,
                                         void* stream) {
  // tap t: weights Wt[t] ([Cout][Cin], reduction contiguous), output columns t * Cout .. of the (n, 8 Cout) row matrix
  return gen_transpose_launch<false>(X, ldx, (const unsigned short*)Wt_bf16, n, Cin, Cout, Y, 8 * Cout, 0,
                                     RowGemmTaps{Cout * Cin, Cout * 4, 1, 0, 0}, 8, (hipStream_t)stream);
}
extern "C" int es_gen_transpose_dgrad_bf16(const float* dY, const void* Wn_bf16, int n, int Cin, int Cout, float* dX, int ldx,
                                           int accumulate, void* stream) {
  // dX[i] (+)= sum_t dY[8 i + t] @ w[t]^T: reduction over the Cout columns of tap t (input columns t * Cout ..) with the
  // natural copy Wn[t] ([Cin][Cout]) as the [output column][reduction] operand
  return gen_transpose_launch<true>(dY, 8 * Cout, (const unsigned short*)Wn_bf16, n, Cout, Cin, dX, ldx, accumulate,
                                    RowGemmTaps{0, 0, 8, Cout, Cin * Cout}, 1, (hipStream_t)stream);
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
//
// Only 20-35 % of the (output row, tap) slots of a 3x3x3 map hold a neighbour, so a workgroup (one tap, one row slice)
// first COMPACTS its slice: 256 map entries at a time are filtered by ballot into an LDS ring of (row, neighbour)
// pairs, and the GEMM consumes the ring 32 pairs at a time.  Rows without that neighbour cost one map read, nothing else.
#define GR 32                 // pairs per chunk (= MFMA K)
#define GLD (GR + 16)         // 96-B rows: conflict-free fragment reads (see HLD)
#define QCAP 512              // ring capacity (max live: 63 left over + 256 appended)

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with a private L2.  All
// (tap, channel-tile) workgroups of one row slice re-read the same X / dY rows, so they are mapped to the SAME XCD,
// consecutively: XCD c walks slices c, c+8, c+16, ... and inside a slice all gridDim.x * gridDim.y tiles.  The launch
// pads gridDim.z to a multiple of 8; slices >= n_slices exit.
__device__ __forceinline__ bool xcd_slice_order(int n_slices, int& bx, int& by, int& bz) {
  const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int c = L & 7, s = L >> 3, per = gridDim.x * gridDim.y;
  const int inner = s % per;
  bz = (s / per) * 8 + c;
  bx = inner % gridDim.x;
  by = inner / gridDim.x;
  return bz < n_slices;
}

struct PairRing {
  int* qj; int* qi; int* wcnt;
  int head, tail, nextb;      // wave-uniform
};

// append the valid pairs among rows [nextb, nextb+256) of tap k
__device__ __forceinline__ void ring_refill(PairRing& q, const int* __restrict__ nbr, long long sj, long long koff, int rend, int n_in) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int j = q.nextb + t, idx = -1;
  if (j < rend) idx = nbr ? nbr[j * sj + koff] : (j < n_in ? j : -1);
  bool v = idx >= 0;
  unsigned long long m = __ballot(v);
  int pre = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) q.wcnt[wv] = __popcll(m);
  __syncthreads();
  int c0 = q.wcnt[0], c1 = q.wcnt[1], c2 = q.wcnt[2], c3 = q.wcnt[3];
  int off = (wv > 0 ? c0 : 0) + (wv > 1 ? c1 : 0) + (wv > 2 ? c2 : 0);
  if (v) {
    int pos = (q.tail + off + pre) & (QCAP - 1);
    q.qj[pos] = j; q.qi[pos] = idx;
  }
  q.tail += c0 + c1 + c2 + c3;
  q.nextb += 256;
  __syncthreads();
}
__device__ __forceinline__ void ring_fill(PairRing& q, const int* __restrict__ nbr, long long sj, long long koff, int rend, int n_in,
                                          int need = 2 * GR) {
  while (q.tail - q.head < need && q.nextb < rend) ring_refill(q, nbr, sj, koff, rend, n_in);
}

// 4 consecutive channels of one row as floats; HALF: the row matrix is a bf16 shadow (exact widening)
template <int HALF>
__device__ __forceinline__ void wgrad_load4(const void* __restrict__ base, size_t row, int ld, int c, int C, bool vec,
                                            float (&o)[4]) {
  if (HALF) {
    const unsigned short* p = (const unsigned short*)base + row * ld + c;
    if (vec && c + 3 < C) {
      uint2 v = *(const uint2*)p;
      o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
      o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (c + e < C) ? __uint_as_float((uint32_t)p[e] << 16) : 0.f;
    }
  } else {
    const float* p = (const float*)base + row * ld + c;
    if (vec && c + 3 < C) {
      float4 v = *(const float4*)p;
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (c + e < C) ? p[e] : 0.f;
    }
  }
}

// XH / YH: X / dY are bf16 shadows (n, ld) instead of f32 row matrices -- half the gathered bytes
template <int XH, int YH>
__global__ __launch_bounds__(256) void k_spconv_wgrad_bf16(const void* __restrict__ X, int ldx,
                                                           const void* __restrict__ dY, int ldy,
                                                           const int* __restrict__ nbr, int n_out, int n_in, int K,
                                                           int Cin, int Cout, int rows_per_split,
                                                           int n_slices, float* __restrict__ dW, float* __restrict__ ws,
                                                           int accumulate) {
  __shared__ __attribute__((aligned(16))) unsigned short As[WM * GLD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[WN * GLD];
  __shared__ int s_qj[QCAP], s_qi[QCAP], s_wc[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nCt = (Cin + WM - 1) / WM;
  int bx, by, bz;
  if (!xcd_slice_order(n_slices, bx, by, bz)) return;
  const int k = bx / nCt, c0 = (bx % nCt) * WM;
  const int n0 = by * WN;
  const int rbeg = bz * rows_per_split;
  const int rend = min(n_out, rbeg + rows_per_split);
  const bool vecA = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const bool vecB = ((ldy & 3) == 0) && ((((uintptr_t)dY) & 15) == 0);
  const int rp = t & 15, l4 = (t >> 4) * 4;
  const int li = lane & 15, kq = lane >> 4;
  PairRing q{s_qj, s_qi, s_wc, 0, 0, rbeg};

  f32x4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint32_t ra[4], rb[4];
  auto load_rows = [&]() {
    float xa[2][4], xb[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int qq = q.head + 2 * rp + h;
      int j = -1, idx = -1;
      if (qq < q.tail) { j = q.qj[qq & (QCAP - 1)]; idx = q.qi[qq & (QCAP - 1)]; }
      int c = c0 + l4, n = n0 + l4;
      if (idx >= 0 && c < Cin) wgrad_load4<XH>(X, (size_t)idx, ldx, c, Cin, vecA, xa[h]);
      else xa[h][0] = xa[h][1] = xa[h][2] = xa[h][3] = 0.f;
      if (idx >= 0 && n < Cout) wgrad_load4<YH>(dY, (size_t)j, ldy, n, Cout, vecB, xb[h]);
      else xb[h][0] = xb[h][1] = xb[h][2] = xb[h][3] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ra[e] = pack_bf16(xa[0][e], xa[1][e]);
      rb[e] = pack_bf16(xb[0][e], xb[1][e]);
    }
  };

  ring_fill(q, nbr, K, k, rend, n_in);
  load_rows();
  while (q.head < q.tail) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *(uint32_t*)&As[(l4 + e) * GLD + 2 * rp] = ra[e];
      *(uint32_t*)&Bs[(l4 + e) * GLD + 2 * rp] = rb[e];
    }
    __syncthreads();
    q.head += GR;
    ring_fill(q, nbr, K, k, rend, n_in);
    load_rows();                                          // next chunk (all-zero past the tail)
    bf16x8_t a = *(const bf16x8_t*)&As[(wv * 16 + li) * GLD + kq * 8];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      bf16x8_t b = *(const bf16x8_t*)&Bs[(nf * 16 + li) * GLD + kq * 8];
      acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[nf], 0, 0, 0);
    }
    __syncthreads();
  }
  if (q.tail == 0 && !ws && accumulate) return;           // tap absent from the (only) slice: nothing to add
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    int col = n0 + nf * 16 + li;
    if (col >= Cout) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int c = c0 + wv * 16 + kq * 4 + r;
      if (c < Cin) wgrad_emit(dW, ws, bz, (size_t)K * Cin * Cout, ((size_t)k * Cin + c) * Cout + col, acc[nf][r], accumulate);
    }
  }
}

// Large-tile bf16 wgrad: one workgroup owns a 128 (C_in) x 128 (C_out) block of dW[k]; each wave a 32 x 128 slab
// (16 MFMAs per 32-pair chunk instead of 4).  For C = 128 layers X[nbr] and dY are then each read exactly once per
// valid pair.  Fast path only (Cin % 128 == 0, Cout % 128 == 0, aligned, 32-bit offsets); other shapes use
// k_spconv_wgrad_bf16.
template <int XH, int YH>
__global__ __launch_bounds__(256) void k_spconv_wgrad_bf16_big(const void* __restrict__ Xv, int ldx,
                                                               const void* __restrict__ dYv, int ldy,
                                                               const int* __restrict__ nbr, int n_out, int n_in, int K,
                                                               int Cin, int Cout, int rows_per_split,
                                                               int n_slices, float* __restrict__ dW, float* __restrict__ ws,
                                                               int accumulate) {
  __shared__ __attribute__((aligned(16))) unsigned short As[128 * GLD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[128 * GLD];
  __shared__ int s_qj[QCAP], s_qi[QCAP], s_wc[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nCt = Cin / 128;
  int bx, by, bz;
  if (!xcd_slice_order(n_slices, bx, by, bz)) return;
  const int k = bx / nCt, c0 = (bx % nCt) * 128;
  const int n0 = by * 128;
  const int rbeg = bz * rows_per_split;
  const int rend = min(n_out, rbeg + rows_per_split);
  // staging: thread = (pair-of-pairs rp 0..15, channel group cg 0..15); it converts channels cg*8 .. cg*8+7 of ring
  // entries 2rp, 2rp+1 into 8 packed bf16x2 words (same channel, two consecutive entries) -> As[c][2rp..2rp+1]
  const int rp = t & 15, c8 = (t >> 4) * 8;
  const int li = lane & 15, kq = lane >> 4;
  PairRing q{s_qj, s_qi, s_wc, 0, 0, rbeg};
  const long long sj = K, koff = k;
  f32x4 acc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per ring entry h: 8 channels of X and of dY, as 2 float4 (f32 source) or one uint4 of 8 bf16 (shadow source)
  float4 xa[2][XH ? 1 : 2], xb[2][YH ? 1 : 2];
  uint4 ha[2], hb[2];
  int va[2];
  auto load_rows = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int qq = q.head + 2 * rp + h;
      va[h] = qq < q.tail;
      int j = va[h] ? q.qj[qq & (QCAP - 1)] : rbeg;       // unconditional loads, masked when packed
      int idx = va[h] ? q.qi[qq & (QCAP - 1)] : 0;
      if (XH) {
        ha[h] = *(const uint4*)((const unsigned short*)Xv + idx * ldx + c0 + c8);
      } else {
        const float4* px = (const float4*)((const float*)Xv + idx * ldx + c0 + c8);
        xa[h][0] = px[0]; xa[h][XH ? 0 : 1] = px[1];
      }
      if (YH) {
        hb[h] = *(const uint4*)((const unsigned short*)dYv + j * ldy + n0 + c8);
      } else {
        const float4* py = (const float4*)((const float*)dYv + j * ldy + n0 + c8);
        xb[h][0] = py[0]; xb[h][YH ? 0 : 1] = py[1];
      }
    }
  };
  auto store_rows = [&]() {
    uint32_t wa[8], wb[8];                                // (entry 2rp, entry 2rp+1) of channel c8+e, packed bf16x2
    if (XH) {
      uint32_t m0 = va[0] ? 0xffffffffu : 0u, m1 = va[1] ? 0xffffffffu : 0u;
      uint32_t u0[4] = {ha[0].x & m0, ha[0].y & m0, ha[0].z & m0, ha[0].w & m0};
      uint32_t u1[4] = {ha[1].x & m1, ha[1].y & m1, ha[1].z & m1, ha[1].w & m1};
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        wa[e] = (u0[e >> 1] & 0xffffu) | (u1[e >> 1] << 16);
        wa[e + 1] = (u0[e >> 1] >> 16) | (u1[e >> 1] & 0xffff0000u);
      }
    } else {
      float a0[8] = {xa[0][0].x, xa[0][0].y, xa[0][0].z, xa[0][0].w, xa[0][XH ? 0 : 1].x, xa[0][XH ? 0 : 1].y, xa[0][XH ? 0 : 1].z, xa[0][XH ? 0 : 1].w};
      float a1[8] = {xa[1][0].x, xa[1][0].y, xa[1][0].z, xa[1][0].w, xa[1][XH ? 0 : 1].x, xa[1][XH ? 0 : 1].y, xa[1][XH ? 0 : 1].z, xa[1][XH ? 0 : 1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) wa[e] = pack_bf16(va[0] ? a0[e] : 0.f, va[1] ? a1[e] : 0.f);
    }
    if (YH) {
      uint32_t m0 = va[0] ? 0xffffffffu : 0u, m1 = va[1] ? 0xffffffffu : 0u;
      uint32_t u0[4] = {hb[0].x & m0, hb[0].y & m0, hb[0].z & m0, hb[0].w & m0};
      uint32_t u1[4] = {hb[1].x & m1, hb[1].y & m1, hb[1].z & m1, hb[1].w & m1};
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        wb[e] = (u0[e >> 1] & 0xffffu) | (u1[e >> 1] << 16);
        wb[e + 1] = (u0[e >> 1] >> 16) | (u1[e >> 1] & 0xffff0000u);
      }
    } else {
      float b0[8] = {xb[0][0].x, xb[0][0].y, xb[0][0].z, xb[0][0].w, xb[0][YH ? 0 : 1].x, xb[0][YH ? 0 : 1].y, xb[0][YH ? 0 : 1].z, xb[0][YH ? 0 : 1].w};
      float b1[8] = {xb[1][0].x, xb[1][0].y, xb[1][0].z, xb[1][0].w, xb[1][YH ? 0 : 1].x, xb[1][YH ? 0 : 1].y, xb[1][YH ? 0 : 1].z, xb[1][YH ? 0 : 1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) wb[e] = pack_bf16(va[0] ? b0[e] : 0.f, va[1] ? b1[e] : 0.f);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      *(uint32_t*)&As[(c8 + e) * GLD + 2 * rp] = wa[e];
      *(uint32_t*)&Bs[(c8 + e) * GLD + 2 * rp] = wb[e];
    }
  };
  ring_fill(q, nbr, sj, koff, rend, n_in);
  load_rows();
  while (q.head < q.tail) {
    store_rows();
    __syncthreads();
    q.head += GR;
    ring_fill(q, nbr, sj, koff, rend, n_in);
    load_rows();                                          // entries past the tail are masked (and clamped) inside
    bf16x8_t a[2], b[8];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) a[mf] = *(const bf16x8_t*)&As[(wv * 32 + mf * 16 + li) * GLD + kq * 8];
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) b[nf] = *(const bf16x8_t*)&Bs[(nf * 16 + li) * GLD + kq * 8];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < 8; ++nf)
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
    __syncthreads();
  }
  if (q.tail == 0 && !ws && accumulate) return;
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) {
      int col = n0 + nf * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int c = c0 + wv * 32 + mf * 16 + kq * 4 + r;
        wgrad_emit(dW, ws, bz, (size_t)K * Cin * Cout, ((size_t)k * Cin + c) * Cout + col, acc[mf][nf][r], accumulate);
      }
    }
}

// es_set_option key 14 (default ON since round 4: bit-identical to the register-transposing tile on the GPU,
// tests/test_gpu_experimental.py; -0.2 ms per mv-3ddet step): the 128 x 128 weight-gradient tile with LDS-DMA staging and
// TRANSPOSED LDS reads.
// The kernels above transpose the gathered rows with VALU packing and 16 ds_write_b32 per thread and chunk so that the MFMA
// fragments are k-contiguous 16-byte LDS reads: 64 + 40 LDS cycles per wave and chunk against 80 cycles of MFMA issue, 15 % MFMA
// busy (profiles/r3_mfma_util.txt).  gfx950 has ds_read_b64_tr_b16: rows can stay in their NATURAL layout [pair][channel] in LDS
// -- written by global_load_lds straight from the gathered bf16 rows, no registers, no conversion, no ds_write -- and a 16-lane
// group reads a 4 (pairs) x 16 (channels) block transposed: lane i gets channel i of the 4 pairs.  Two such reads give the
// 8 k-values of a 16x16x32 fragment.  Per wave and chunk: 16 transposed reads (2 LDS cycles each) for 16 MFMAs.
// Semantics of the transposed read, CONFIRMED by tools/probes/tr_read.hip on MI355X (profiles/r4a_tr_read.txt): within a
// 16-lane group, lane i supplies the address of the 8 bytes at block row (i >> 2), block columns 4 (i & 3) ..; it receives
// column i, rows 0 .. 3.
// Tile rows are 256 bytes (128 channels); the 16-byte granule g of pair p is stored at slot g ^ (key(p) << 1), key(p) =
// (p & 3) | ((p >> 3) & 1) << 2: the 8 pair rows a 32-lane read group touches land on 8 distinct 32-byte bank positions.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
template <int GT>      // pairs per chunk: 32 (one MFMA k step behind every barrier) or 64 (two; 64 KB of LDS: round 6 A/B)
__global__ __launch_bounds__(256) void k_spconv_wgrad_bf16_tr(const unsigned short* __restrict__ Xh, int ldx,
                                                              const unsigned short* __restrict__ dYh, int ldy,
                                                              const int* __restrict__ nbr, int n_out, int n_in, int K,
                                                              int Cin, int Cout, int rows_per_split, int n_slices,
                                                              float* __restrict__ dW, float* __restrict__ ws, int accumulate) {
  constexpr int TB = GT * 256;                              // bytes of one operand tile: GT pairs x 128 channels
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TB + 2 * QCAP * 4 + 16];
  int* const s_qj = (int*)(smem + 4 * TB);
  int* const s_qi = s_qj + QCAP;
  int* const s_wc = s_qi + QCAP;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int wr = wv >> 1, wc = wv & 1;
  const int nCt = Cin / 128;
  int bx, by, bz;
  if (!xcd_slice_order(n_slices, bx, by, bz)) return;
  const int k = bx / nCt, c0 = (bx % nCt) * 128, n0 = by * 128;
  const int rbeg = bz * rows_per_split, rend = min(n_out, rbeg + rows_per_split);
  PairRing q{s_qj, s_qi, s_wc, 0, 0, rbeg};
  const long long sj = K, koff = k;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto key = [](int p) { return ((p & 3) | (((p >> 3) & 1) << 2)) << 1; };
  // DMA piece e = (j * 4 + wv) * 64 + lane of a tile: pair e >> 4, slot e & 15 (16 granules per 256-byte row)
  auto issue = [&](int buf, int head) {
#pragma unroll
    for (int j = 0; j < GT / 16; ++j) {
      const int e = (j * 4 + wv) * 64 + lane, pr = e >> 4, g = (e & 15) ^ key(pr);
      const int qq = head + pr;
      const bool v = qq < q.tail;
      const int row = v ? q.qj[qq & (QCAP - 1)] : 0, idx = v ? q.qi[qq & (QCAP - 1)] : 0;
      const unsigned short* px = v ? (Xh + idx * ldx + c0 + g * 8) : g_zero_granule;
      const unsigned short* py = v ? (dYh + row * ldy + n0 + g * 8) : g_zero_granule;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)px,
                                       (__attribute__((address_space(3))) void*)(smem + (buf * 2 + 0) * TB + (j * 4 + wv) * 1024),
                                       16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)py,
                                       (__attribute__((address_space(3))) void*)(smem + (buf * 2 + 1) * TB + (j * 4 + wv) * 1024),
                                       16, 0, 0);
    }
  };
  // transposed fragment of channel block cb (16 channels) of a tile: the 8 pairs kq * 8 .. of channel li
  auto frag = [&](const unsigned char* tile, int cb, int ks) -> bf16x8_t {
    s16x4_t h[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int pr = ks * 32 + kq * 8 + r * 4 + (li >> 2);
      const int g = cb * 2 + ((li & 3) >> 1);
      const unsigned char* a = tile + pr * 256 + ((g ^ key(pr)) * 16) + (li & 1) * 8;
      h[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a);
    }
    s16x8_t v = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  };
  ring_fill(q, nbr, sj, koff, rend, n_in, 2 * GT);
  int buf = 0;
  if (q.head < q.tail) issue(0, q.head);
  while (q.head < q.tail) {
    q.head += GT;
    ring_fill(q, nbr, sj, koff, rend, n_in, 2 * GT);       // the NEXT chunk's pairs are in the ring before its DMA is issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // the current chunk has landed; the other buffer has been consumed
    if (q.head < q.tail) issue(buf ^ 1, q.head);
    const unsigned char* xt = smem + (buf * 2 + 0) * TB;
    const unsigned char* yt = smem + (buf * 2 + 1) * TB;
#pragma unroll
    for (int ks = 0; ks < GT / 32; ++ks) {
      bf16x8_t a[4], b[4];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) a[mf] = frag(xt, wr * 4 + mf, ks);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) b[nf] = frag(yt, wc * 4 + nf, ks);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
    }
    buf ^= 1;
  }
  if (q.tail == 0 && !ws && accumulate) return;
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int col = n0 + (wc * 4 + nf) * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = c0 + (wr * 4 + mf) * 16 + kq * 4 + r;
        wgrad_emit(dW, ws, bz, (size_t)K * Cin * Cout, ((size_t)k * Cin + c) * Cout + col, acc[mf][nf][r], accumulate);
      }
    }
}

// 256 x 256 tile of dW[k] per workgroup of 8 waves (each a 64 x 128 slab: 32 MFMAs per 32-pair chunk), both operands
// from bf16 shadows.  The 128 x 128 tile moves (128 + 128) * 2 B per pair and 16 k outputs = 64 flop/B through L2 -> LDS,
// which is what bounds the weight gradients of the wide (768 .. 3072 channel) dense layers; this tile doubles that.
#define QCAP2 1024            // ring capacity (max live: 63 left over + 512 appended)
__device__ __forceinline__ void ring_refill512(PairRing& q, const int* __restrict__ nbr, long long sj, long long koff,
                                               int rend, int n_in) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int j = q.nextb + t, idx = -1;
  if (j < rend) idx = nbr ? nbr[j * sj + koff] : (j < n_in ? j : -1);
  bool v = idx >= 0;
  unsigned long long m = __ballot(v);
  int pre = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) q.wcnt[wv] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    int c = q.wcnt[w];
    off += (w < wv) ? c : 0;
    tot += c;
  }
  if (v) {
    int pos = (q.tail + off + pre) & (QCAP2 - 1);
    q.qj[pos] = j; q.qi[pos] = idx;
  }
  q.tail += tot;
  q.nextb += 512;
  __syncthreads();
}
__global__ __launch_bounds__(512) void k_spconv_wgrad_bf16_huge(const unsigned short* __restrict__ X, int ldx,
                                                                const unsigned short* __restrict__ dY, int ldy,
                                                                const int* __restrict__ nbr, int n_out, int n_in, int K,
                                                                int Cin, int Cout, int rows_per_split, int n_slices,
                                                                float* __restrict__ dW, float* __restrict__ ws, int accumulate) {
  __shared__ __attribute__((aligned(16))) unsigned short As[256 * GLD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[256 * GLD];
  __shared__ int s_qj[QCAP2], s_qi[QCAP2], s_wc[8];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
  const int nCt = Cin / 256;
  int bx, by, bz;
  if (!xcd_slice_order(n_slices, bx, by, bz)) return;
  const int k = bx / nCt, c0 = (bx % nCt) * 256;
  const int n0 = by * 256;
  const int rbeg = bz * rows_per_split;
  const int rend = min(n_out, rbeg + rows_per_split);
  // staging: thread = (pair-of-pairs rp 0..15, channel group 0..31 of 8 channels)
  const int rp = t & 15, c8 = (t >> 4) * 8;
  const int li = lane & 15, kq = lane >> 4;
  PairRing q{s_qj, s_qi, s_wc, 0, 0, rbeg};
  const long long sj = K, koff = k;
  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 ha[2], hb[2];
  int va[2];
  auto fill = [&]() {
    while (q.tail - q.head < 2 * GR && q.nextb < rend) ring_refill512(q, nbr, sj, koff, rend, n_in);
  };
  auto load_rows = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int qq = q.head + 2 * rp + h;
      va[h] = qq < q.tail;
      int j = va[h] ? q.qj[qq & (QCAP2 - 1)] : rbeg;      // unconditional loads, masked when packed
      int idx = va[h] ? q.qi[qq & (QCAP2 - 1)] : 0;
      ha[h] = *(const uint4*)(X + idx * ldx + c0 + c8);
      hb[h] = *(const uint4*)(dY + j * ldy + n0 + c8);
    }
  };
  auto store_rows = [&]() {
    uint32_t m0 = va[0] ? 0xffffffffu : 0u, m1 = va[1] ? 0xffffffffu : 0u;
    uint32_t a0[4] = {ha[0].x & m0, ha[0].y & m0, ha[0].z & m0, ha[0].w & m0};
    uint32_t a1[4] = {ha[1].x & m1, ha[1].y & m1, ha[1].z & m1, ha[1].w & m1};
    uint32_t b0[4] = {hb[0].x & m0, hb[0].y & m0, hb[0].z & m0, hb[0].w & m0};
    uint32_t b1[4] = {hb[1].x & m1, hb[1].y & m1, hb[1].z & m1, hb[1].w & m1};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      *(uint32_t*)&As[(c8 + e) * GLD + 2 * rp] = (a0[e >> 1] & 0xffffu) | (a1[e >> 1] << 16);
      *(uint32_t*)&As[(c8 + e + 1) * GLD + 2 * rp] = (a0[e >> 1] >> 16) | (a1[e >> 1] & 0xffff0000u);
      *(uint32_t*)&Bs[(c8 + e) * GLD + 2 * rp] = (b0[e >> 1] & 0xffffu) | (b1[e >> 1] << 16);
      *(uint32_t*)&Bs[(c8 + e + 1) * GLD + 2 * rp] = (b0[e >> 1] >> 16) | (b1[e >> 1] & 0xffff0000u);
    }
  };
  fill();
  load_rows();
  while (q.head < q.tail) {
    store_rows();
    __syncthreads();
    q.head += GR;
    fill();
    load_rows();                                          // entries past the tail are masked (and clamped) inside
    bf16x8_t a[4], b[8];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) a[mf] = *(const bf16x8_t*)&As[(wm * 64 + mf * 16 + li) * GLD + kq * 8];
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) b[nf] = *(const bf16x8_t*)&Bs[(wn * 128 + nf * 16 + li) * GLD + kq * 8];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < 8; ++nf)
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
    __syncthreads();
  }
  if (q.tail == 0 && !ws && accumulate) return;
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) {
      int col = n0 + wn * 128 + nf * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int c = c0 + wm * 64 + mf * 16 + kq * 4 + r;
        wgrad_emit(dW, ws, bz, (size_t)K * Cin * Cout, ((size_t)k * Cin + c) * Cout + col, acc[mf][nf][r], accumulate);
      }
    }
}

// which tile and how many row slices a bf16 weight-gradient launch uses (shared by the launch and the workspace query)
// weight gradient of the same layers: dW[ci][co] = sum over rows X[row][ci] dY[row][co], f32 rows rounded to bf16 while staged.  A workgroup owns a
// (64 x 64) tile of dW and a slice of 256 rows: one round of independent loads (a row's 64 + 64 channels per thread), TRANSPOSED 2-byte LDS stores
// ([channel][row], pitch 528 B) so that both operands are read as 16-byte row-contiguous fragments, 32 MFMAs per wave, partial tile to the slice's
// workspace block (added in slice order by the reduction launch every weight-gradient launch ends with).
__global__ __launch_bounds__(256) void k_lin_wgrad_small(const float* __restrict__ X, int ldx, const float* __restrict__ dY, int ldy, int n,
                                                         int Cin, int Cout, int rows_per_split, float* __restrict__ dW,
                                                         float* __restrict__ ws, int accumulate) {
  __shared__ __attribute__((aligned(16))) unsigned short Xt[64 * LS_LD];
  __shared__ __attribute__((aligned(16))) unsigned short Yt[64 * LS_LD];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int c0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int rbeg = blockIdx.z * rows_per_split, rend = min(n, rbeg + rows_per_split);
  f32x4 acc[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int r0 = rbeg; r0 < rend; r0 += LS_KS) {
    const int row = r0 + t;
    float4 rx[16], ry[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      rx[i] = row < rend ? *(const float4*)(X + (size_t)row * ldx + c0 + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
      ry[i] = row < rend ? *(const float4*)(dY + (size_t)row * ldy + n0 + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t x01 = pack_bf16(rx[i].x, rx[i].y), x23 = pack_bf16(rx[i].z, rx[i].w);
      const uint32_t y01 = pack_bf16(ry[i].x, ry[i].y), y23 = pack_bf16(ry[i].z, ry[i].w);
      Xt[(4 * i + 0) * LS_LD + t] = (unsigned short)(x01 & 0xffffu);
      Xt[(4 * i + 1) * LS_LD + t] = (unsigned short)(x01 >> 16);
      Xt[(4 * i + 2) * LS_LD + t] = (unsigned short)(x23 & 0xffffu);
      Xt[(4 * i + 3) * LS_LD + t] = (unsigned short)(x23 >> 16);
      Yt[(4 * i + 0) * LS_LD + t] = (unsigned short)(y01 & 0xffffu);
      Yt[(4 * i + 1) * LS_LD + t] = (unsigned short)(y01 >> 16);
      Yt[(4 * i + 2) * LS_LD + t] = (unsigned short)(y23 & 0xffffu);
      Yt[(4 * i + 3) * LS_LD + t] = (unsigned short)(y23 >> 16);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < LS_KS / 32; ++ks) {
      const bf16x8_t a = *(const bf16x8_t*)&Xt[(wv * 16 + li) * LS_LD + ks * 32 + kq * 8];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const bf16x8_t b = *(const bf16x8_t*)&Yt[(nf * 16 + li) * LS_LD + ks * 32 + kq * 8];
        acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[nf], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      wgrad_emit(dW, ws, blockIdx.z, (size_t)Cin * Cout, (size_t)(c0 + wv * 16 + kq * 4 + r) * Cout + n0 + nf * 16 + li, acc[nf][r], accumulate);
}
static bool lin_wgrad_small_ok(int XH, int YH, const void* X, int ldx, const void* dY, int ldy, int n_out, int n_in, int K, int Cin, int Cout) {
  return ES_OPT_LIN_SMALL && K == 1 && !XH && !YH && n_out == n_in && n_out <= 8192 && (Cin % 64 == 0) && (Cout % 64 == 0) &&
         (long long)Cin * Cout <= 256ll * 256ll && (ldx % 4 == 0) && (ldy % 4 == 0) && ((((uintptr_t)X) | ((uintptr_t)dY)) & 15) == 0;
}
static WgradPlan wgrad_plan_bf16(int XH, int YH, const void* X, int ldx, const void* dY, int ldy, int n_out, int n_in, int K,
                                 int Cin, int Cout, bool have_ws) {
  const long long nw = (long long)K * Cin * Cout;
  if (lin_wgrad_small_ok(XH, YH, X, ldx, dY, ldy, n_out, n_in, K, Cin, Cout)) {     // few rows, narrow layers (identity map: the launcher checks)
    const int splits = cap_splits(es_cdiv(n_out, LS_KS), nw, have_ws);
    const int rows_per_split = es_cdiv(es_cdiv(n_out, splits), LS_KS) * LS_KS;
    return WgradPlan{5, es_cdiv(n_out, rows_per_split), rows_per_split};
  }
  // 16-byte row segments: 4 floats or 8 bf16 per load
  const int ax = XH ? 8 : 4, ay = YH ? 8 : 4;
  bool big = (Cin % 128 == 0) && (Cout % 128 == 0) && (ldx % ax == 0) && (ldy % ay == 0) &&
             ((((uintptr_t)X) & 15) == 0) && ((((uintptr_t)dY) & 15) == 0) && ((long long)n_in * ldx < (1ll << 31)) &&
             ((long long)n_out * ldy < (1ll << 31)) &&
             (n_out >= 512 || (long long)Cin * Cout >= 512ll * 512ll);   // few rows x many channels: dW traffic decides
  bool huge = big && XH && YH && ES_OPT_WGRAD_HUGE && (Cin % 256 == 0) && (Cout % 256 == 0) && (ldx % 8 == 0) && (ldy % 8 == 0);
  if (huge) {
    int base = K * (Cin / 256) * (Cout / 256);
    int splits = es_cdiv(2048, base);
    int max_splits = es_cdiv(n_out, 1024);
    if (splits > max_splits) splits = max_splits;
    splits = cap_splits(splits, nw, have_ws);
    // one 512-thread workgroup per CU: below ~4 workgroups per CU the 128 x 128 tile fills the chip better (measured: the
    // 256 .. 1024-channel levels of mv-3ddet have 190 .. 12 k rows -> 200 .. 400 workgroups, +0.9 ms per step)
    if ((long long)base * splits >= 960) {
      int rows_per_split = es_cdiv(es_cdiv(n_out, splits), GR) * GR;
      return WgradPlan{3, es_cdiv(n_out, rows_per_split), rows_per_split};
    }
  }
  if (big) {
    int base = K * (Cin / 128) * (Cout / 128);
    int splits = es_cdiv(ES_OPT_WG_BIG_TARGET, base);
    int max_splits = es_cdiv(n_out, ES_OPT_WG_BIG_ROWS);
    if (splits > max_splits) splits = max_splits;
    splits = cap_splits(splits, nw, have_ws);
    int rows_per_split = es_cdiv(es_cdiv(n_out, splits), GR) * GR;
    return WgradPlan{2, es_cdiv(n_out, rows_per_split), rows_per_split};
  }
  int base = K * es_cdiv(Cin, WM) * es_cdiv(Cout, WN);
  int splits = es_cdiv(ES_OPT_WG_SMALL_TARGET, base);
  int max_splits = es_cdiv(n_out, 256);
  if (splits > max_splits) splits = max_splits;
  splits = cap_splits(splits, nw, have_ws);
  int rows_per_split = es_cdiv(es_cdiv(n_out, splits), GR) * GR;
  return WgradPlan{1, es_cdiv(n_out, rows_per_split), rows_per_split};
}

template <int XH, int YH>
static int wgrad_bf16_launch(const void* X, int ldx, const void* dY, int ldy, const int* nbr, int n_out, int n_in, int K,
                             int Cin, int Cout, float* dW, int accumulate, float* ws, size_t ws_floats, void* stream) {
  if (n_out <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const WgradPlan p = wgrad_plan_bf16(XH, YH, X, ldx, dY, ldy, n_out, n_in, K, Cin, Cout, ws != nullptr);
  const size_t nw = (size_t)K * Cin * Cout;
  if (p.splits > 1 && ws_floats < (size_t)p.splits * nw) return -5;
  float* wsk = p.splits > 1 ? ws : nullptr;
  const int gz = es_cdiv(p.splits, 8) * 8;                // XCD-aware slice order: slices >= p.splits exit
  hipStream_t st = (hipStream_t)stream;
  if (p.kind == 5 && nbr == nullptr) {
    hipLaunchKernelGGL(k_lin_wgrad_small, dim3(Cin / 64, Cout / 64, p.splits), dim3(256), 0, st, (const float*)X, ldx, (const float*)dY, ldy,
                       n_out, Cin, Cout, p.rows_per_split, dW, wsk, accumulate);
  } else if (p.kind == 3) {
    dim3 grid(K * (Cin / 256), Cout / 256, gz);
    hipLaunchKernelGGL(k_spconv_wgrad_bf16_huge, grid, dim3(512), 0, st, (const unsigned short*)X, ldx,
                       (const unsigned short*)dY, ldy, nbr, n_out, n_in, K, Cin, Cout, p.rows_per_split, p.splits, dW, wsk, accumulate);
  } else if (p.kind == 2 && XH && YH && ES_OPT_WGRAD_TR && (ldx % 8 == 0) && (ldy % 8 == 0)) {
    dim3 grid(K * (Cin / 128), Cout / 128, gz);                // experimental: LDS-DMA staging + transposed LDS reads
    if (ES_OPT_WGRAD_TR == 2)
      hipLaunchKernelGGL(k_spconv_wgrad_bf16_tr<64>, grid, dim3(256), 0, st, (const unsigned short*)X, ldx, (const unsigned short*)dY,
                         ldy, nbr, n_out, n_in, K, Cin, Cout, p.rows_per_split, p.splits, dW, wsk, accumulate);
    else
      hipLaunchKernelGGL(k_spconv_wgrad_bf16_tr<32>, grid, dim3(256), 0, st, (const unsigned short*)X, ldx, (const unsigned short*)dY,
                         ldy, nbr, n_out, n_in, K, Cin, Cout, p.rows_per_split, p.splits, dW, wsk, accumulate);
  } else if (p.kind == 2) {
    dim3 grid(K * (Cin / 128), Cout / 128, gz);
    hipLaunchKernelGGL((k_spconv_wgrad_bf16_big<XH, YH>), grid, dim3(256), 0, st, X, ldx, dY, ldy, nbr, n_out, n_in, K, Cin,
                       Cout, p.rows_per_split, p.splits, dW, wsk, accumulate);
  } else {
    dim3 grid(K * es_cdiv(Cin, WM), es_cdiv(Cout, WN), gz);
    hipLaunchKernelGGL((k_spconv_wgrad_bf16<XH, YH>), grid, dim3(256), 0, st, X, ldx, dY, ldy, nbr, n_out, n_in, K, Cin,
                       Cout, p.rows_per_split, p.splits, dW, wsk, accumulate);
  }
  ES_CHECK_LAUNCH();
  return wgrad_reduce(ws, p.splits, nw, dW, accumulate, st);
}

extern "C" int es_spconv_wgrad_bf16(const float* X, int ldx, const float* dY, int ldy, const int* nbr, int n_out,
                                    int n_in, int K, int Cin, int Cout, float* dW, int accumulate, float* ws, size_t ws_floats,
                                    void* stream) {
  return wgrad_bf16_launch<0, 0>(X, ldx, dY, ldy, nbr, n_out, n_in, K, Cin, Cout, dW, accumulate, ws, ws_floats, stream);
}

extern "C" int es_spconv_wgrad_bf16_src(const void* X, int x_half, int ldx, const void* dY, int dy_half, int ldy,
                                        const int* nbr, int n_out, int n_in, int K, int Cin, int Cout, float* dW,
                                        int accumulate, float* ws, size_t ws_floats, void* stream) {
  if (x_half && dy_half) return wgrad_bf16_launch<1, 1>(X, ldx, dY, ldy, nbr, n_out, n_in, K, Cin, Cout, dW, accumulate, ws, ws_floats, stream);
  if (x_half) return wgrad_bf16_launch<1, 0>(X, ldx, dY, ldy, nbr, n_out, n_in, K, Cin, Cout, dW, accumulate, ws, ws_floats, stream);
  if (dy_half) return wgrad_bf16_launch<0, 1>(X, ldx, dY, ldy, nbr, n_out, n_in, K, Cin, Cout, dW, accumulate, ws, ws_floats, stream);
  return wgrad_bf16_launch<0, 0>(X, ldx, dY, ldy, nbr, n_out, n_in, K, Cin, Cout, dW, accumulate, ws, ws_floats, stream);
}

// floats of workspace the weight-gradient launch of this shape wants for its row split (0: a single slice, none needed).
// bf16 = 0: es_spconv_wgrad (exact f32); 1: es_spconv_wgrad_bf16[_src] with the given operand kinds / strides / pointers
// (alignment decides the tile).  Without a workspace the launches run ONE row slice: still deterministic, under-filled.
extern "C" size_t es_spconv_wgrad_workspace_floats(int bf16, const void* X, int x_half, int ldx, const void* dY, int dy_half,
                                                   int ldy, int n_out, int n_in, int K, int Cin, int Cout) {
  if (n_out <= 0 || Cin <= 0 || Cout <= 0) return 0;
  WgradPlan p = bf16 ? wgrad_plan_bf16(x_half, dy_half, X, ldx, dY, ldy, n_out, n_in, K, Cin, Cout, true)
                     : wgrad_plan_f32(n_out, K, Cin, Cout, true);
  return p.splits > 1 ? (size_t)p.splits * K * Cin * Cout : 0;
}

// f32 row matrix -> contiguous bf16 "shadow" (n, C) used as the gather source of the bf16 kernels
__global__ void k_cast_rows(const float* __restrict__ x, int ldx, size_t n, int C, unsigned short* __restrict__ h) {
  size_t tot = n * (size_t)(C >> 1);
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    size_t r = e / (C >> 1);
    int c2 = (int)(e - r * (C >> 1)) * 2;
    const float* p = x + r * ldx + c2;
    ((uint32_t*)h)[e] = pack_bf16(p[0], p[1]);
  }
}
// 8 channels per thread: two float4 in, one 16-B store
__global__ void k_cast_rows8(const float* __restrict__ x, int ldx, size_t n, int C8, uint4* __restrict__ h) {
  size_t tot = n * (size_t)C8;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    size_t r = e / C8;
    int c = (int)(e - r * C8) * 8;
    const float4* p = (const float4*)(x + r * ldx + c);
    float4 a = p[0], b = p[1];
    h[e] = make_uint4(pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w));
  }
}
extern "C" int es_cast_rows_bf16(const float* x, int ldx, int n, int C, void* h, void* stream) {
  if (n <= 0 || C <= 0) return 0;
  if (C & 1) return -8;
  if ((C % 8 == 0) && (ldx % 4 == 0) && (((((uintptr_t)x) | ((uintptr_t)h)) & 15) == 0)) {
    long long tot8 = (long long)n * (C / 8);
    int g8 = es_cdiv(tot8, 256);
    if (g8 > 8192) g8 = 8192;
    hipLaunchKernelGGL(k_cast_rows8, dim3(g8), dim3(256), 0, (hipStream_t)stream, x, ldx, (size_t)n, C / 8, (uint4*)h);
    ES_CHECK_LAUNCH();
    return 0;
  }
  long long tot = (long long)n * (C >> 1);
  int g = es_cdiv(tot, 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_cast_rows, dim3(g), dim3(256), 0, (hipStream_t)stream, x, ldx, (size_t)n, C, (unsigned short*)h);
  ES_CHECK_LAUNCH();
  return 0;
}

// all conv kernels of the model in ONE launch.  table rows = {src f32 ptr, natural bf16 ptr, transposed bf16 ptr, K, A, B,
// first tile}; one workgroup per 64x64 tile of one tap (found by binary search over the tile prefix), transposed through
// LDS so that both copies are written with coalesced rows.
__global__ __launch_bounds__(256) void k_cast_weights_table(const long long* __restrict__ table, int n_entries,
                                                            int total_tiles) {
  __shared__ unsigned short tile[64][66];
  int tid = blockIdx.x;
  if (tid >= total_tiles) return;
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {                                     // last entry with first_tile <= tid
    int mid = (lo + hi + 1) >> 1;
    if (table[(size_t)mid * 7 + 6] <= tid) lo = mid; else hi = mid - 1;
  }
  const long long* t = table + (size_t)lo * 7;
  const float* w = (const float*)t[0];
  unsigned short* nat = (unsigned short*)t[1];
  unsigned short* tr = (unsigned short*)t[2];
  int A = (int)t[4], B = (int)t[5];
  int local = tid - (int)t[6];
  int ta = (A + 63) >> 6, tb = (B + 63) >> 6;
  int k = local / (ta * tb), r = local % (ta * tb);
  int a0 = (r / tb) * 64, b0 = (r % tb) * 64;
  const size_t base = (size_t)k * A * B;
  int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {                    // read rows a0+i, columns b0+tx (coalesced along b)
    int a = a0 + i, b = b0 + tx;
    unsigned short h = 0;
    if (a < A && b < B) {
      h = (unsigned short)(pack_bf16(w[base + (size_t)a * B + b], 0.f) & 0xffff);
      nat[base + (size_t)a * B + b] = h;
    }
    tile[i][tx] = h;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {                    // write rows b0+i of the transposed copy (coalesced along a)
    int b = b0 + i, a = a0 + tx;
    if (a < A && b < B) tr[base + (size_t)b * A + a] = tile[tx][i];
  }
}
extern "C" int es_cast_weights_table(const void* table_dev, int n_entries, int total_tiles, void* stream) {
  if (n_entries <= 0 || total_tiles <= 0) return 0;
  hipLaunchKernelGGL(k_cast_weights_table, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)table_dev, n_entries, total_tiles);
  ES_CHECK_LAUNCH();
  return 0;
}
