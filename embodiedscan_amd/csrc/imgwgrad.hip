// Weight gradient of the image backbone's 3x3 convolutions on small channel counts (round 6; SURVEY 8a row A7: mmdet.ResNet
// depth 50, base_channels 16 -- configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:24-34 -- Bottleneck.conv2 of
// layer2 / layer3: 32 -> 32 on 120 x 120 and 64 -> 64 on 60 x 60 feature maps, 80 images per mv-3ddet step).
//
// dW[t][ci][co] = sum over (image, y, x) of X[y + ty - 1][x + tx - 1][ci] * dY[y][x][co],  t = ty * 3 + tx, zero padding.
//
// The map kernel (k_spconv_wgrad_bf16<1, 0>, spconv.hip) runs one workgroup per (tap, row slice): it compacts (row, neighbour)
// pairs through a ring, gathers 2 x 4 channels per thread and chunk, stores them transposed to LDS and issues FOUR MFMAs behind
// two barriers -- 4 % MFMA busy, 83 % of the wave cycles waiting (profiles/r6m_mfma_util.txt), 150 us for a 55 MB problem.  On an
// image grid nothing has to be gathered or compacted: the neighbour of pixel x under tap tx is pixel x + tx - 1 of an image row
// that is already in LDS.  Here
//   * a workgroup owns a band of output rows of ONE image and ALL nine taps: wave w holds dW[.][16 w .. 16 w + 16][.] for the nine
//     taps in registers (9 x C / 16 accumulator fragments), so X and dY are each read from memory ONCE (the map kernel: once per tap);
//   * the image rows live in a four-slot LDS ring ((W + 2) pixels, the two pad pixels stay zero: no border test in the loop),
//     filled by LDS-DMA one output row ahead; the dY row (f32 in memory) is prefetched into registers one row ahead, rounded to
//     bf16 and stored to one of two LDS tiles: one barrier per output ROW (72 / 144 MFMAs per wave behind it);
//   * both MFMA operands are read TRANSPOSED from their natural [pixel][channel] layout by ds_read_b64_tr_b16 (as k_dconv_wgrad):
//     A = X^T (16 ci x 32 pixels, shifted by the tap), B = dY (32 pixels x 16 co);
//   * partial dW tensors per workgroup go to a workspace and are added in workgroup order by k_img_wgrad_reduce: bit-reproducible.
// Arithmetic: bf16 operands (dY rounded to nearest even like every gradient shadow), f32 accumulation; the summation order
// differs from the map kernel's, the products do not.
#include "common.h"
#include "../../include/es_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) unsigned short g_iw_zero[8];

// C channels (= Cin = Cout), WP = padded OUTPUT width (multiple of 32, >= Wo), S = stride (1, 2): the input grid is (H, W), the
// output grid (H / S, W / S) (3x3, pad 1: output (oy, ox) under tap (ty, tx) reads input (S oy + ty - 1, S ox + tx - 1))
template <int C, int WP, int S>
__global__ __launch_bounds__(C * 4) void k_img_wgrad9(const unsigned short* __restrict__ Xh, int ldx,
                                                      const float* __restrict__ dY, int ldy, int H, int W, int rows_per_wg,
                                                      int bands, float* __restrict__ out /* dW or ws[wg][9][C][C] */,
                                                      int to_ws, int accumulate) {
  constexpr int NW = C / 16, NT = NW * 64;                 // waves / threads per workgroup
  constexpr int RB = C * 2;                                 // bytes per pixel row of an LDS tile
  constexpr int XP = S * WP + 2;                            // pixels per ring slot: x = -1 .. S * WP (pads and the tail stay zero)
  constexpr int NS = S == 1 ? 4 : 8;                        // ring slots (rows S oy - 1 .. S oy + 1 in use, S more in flight)
  constexpr int X_BYTES = XP * RB, Y_BYTES = WP * RB;
  constexpr int NF = C / 16;
  constexpr int GX = RB / 16;                               // 16-byte granules per pixel row
  constexpr int NPX = S * WP * GX / NT;                     // LDS-DMA pieces per thread and image row
  static_assert(S * WP * GX % NT == 0 && (WP * C / 4) % NT == 0, "whole pieces per row");
  constexpr int NLY = (WP * C / 4 + NT - 1) / NT;           // float4 loads per thread and dY row
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * X_BYTES + 2 * Y_BYTES];
  unsigned char* const ytile = smem + NS * X_BYTES;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int im = blockIdx.x / bands, band = blockIdx.x - im * bands;
  const int Ho = H / S, Wo = W / S;
  const int oy0 = band * rows_per_wg, oy1 = min(Ho, oy0 + rows_per_wg);
  if (oy0 >= oy1) return;

  for (int i = t; i < (NS * X_BYTES + 2 * Y_BYTES) / 16; i += NT) ((uint4*)smem)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();

  f32x4 acc[9][NF];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // image row iy -> ring slot iy & 3, pixels 0 .. W - 1 at LDS pixels 1 .. W (granules past W read a zero granule: the tail stays zero)
  auto issue_x = [&](int iy) {
    if (iy < 0 || iy >= H) return;                          // (uniform)
    const unsigned short* src = Xh + ((size_t)im * H + iy) * W * ldx;
    unsigned char* dst = smem + (iy & (NS - 1)) * X_BYTES + RB;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const int e = (j * NW + wv) * 64 + lane;              // granule of the row: pixel e / GX, granule e % GX
      const int px = e / GX, g = e - px * GX;
      const unsigned short* p = px < W ? (src + (size_t)px * ldx + g * 8) : g_iw_zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(dst + (j * NW + wv) * 1024), 16, 0, 0);
    }
  };
  float4 yreg[NLY];
  auto load_y = [&](int oy) {                                // dY row oy -> registers (f32)
    const float* src = dY + ((size_t)im * Ho + oy) * Wo * ldy;
#pragma unroll
    for (int j = 0; j < NLY; ++j) {
      const int e = j * NT + t;                             // float4 index: pixel e / (C / 4), channel quad e % (C / 4)
      const int px = e / (C / 4), c4 = e - px * (C / 4);
      yreg[j] = (oy < oy1 && px < Wo) ? *(const float4*)(src + (size_t)px * ldy + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_y = [&](int buf) {                              // registers -> bf16 tile (pixels >= W: zeros)
#pragma unroll
    for (int j = 0; j < NLY; ++j) {
      const int e = j * NT + t;
      const int px = e / (C / 4), c4 = e - px * (C / 4);
      if (px < WP) {
        uint2 v;
        v.x = es_pack_bf16(yreg[j].x, yreg[j].y);
        v.y = es_pack_bf16(yreg[j].z, yreg[j].w);
        *(uint2*)(ytile + buf * Y_BYTES + px * RB + c4 * 8) = v;
      }
    }
  };
  // transposed fragment: 8 consecutive pixels (p0 + kq * 8 ..) of channel cb * 16 + li from a [pixel][channel] tile
  auto frag = [&](const unsigned char* tile, int p0, int cb, int step, int off) -> bf16x8_t {      // LDS pixel = step * pixel + off
    s16x4_t h[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int px = (p0 + kq * 8 + r * 4 + (li >> 2)) * step + off;
      const unsigned char* a = tile + px * RB + cb * 32 + (li & 3) * 8;
      h[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a);
    }
    s16x8_t v = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  };

  // prologue: the image rows of the first output row (S oy0 - 1 .. S oy0 + 1), the rows the second one adds, and the first dY row
  for (int r = -1; r <= 1 + (S - 1); ++r) issue_x(S * oy0 + r);
  load_y(oy0);
  for (int oy = oy0; oy < oy1; ++oy) {
    const int buf = (oy - oy0) & 1;
    store_y(buf);                                            // (the tile was last read two rows ago: a barrier has passed since)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of the image rows of output row oy have landed ...
    __syncthreads();                                          // ... everybody's; the dY tile is complete; output row oy - 1 has been consumed
    // the rows output row oy + 1 adds (S = 1: oy + 2; S = 2: 2 oy + 3 and -- a row further ahead -- 2 oy + 4): their ring slots held
    // rows S oy - 2 .. that nobody reads any more
    if (S == 1) issue_x(oy + 2);
    else { issue_x(2 * oy + 3); issue_x(2 * oy + 4); }
    load_y(oy + 1);
    const unsigned char* yt = ytile + buf * Y_BYTES;
#pragma unroll 1
    for (int p0 = 0; p0 < WP; p0 += 32) {
      if (p0 >= Wo) break;
      bf16x8_t b[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) b[nf] = frag(yt, p0, nf, 1, 0);
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int iy = S * oy + ty - 1;
        if (iy < 0 || iy >= H) continue;                    // (uniform: a row outside the image contributes nothing)
        const unsigned char* xt = smem + (iy & (NS - 1)) * X_BYTES;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const bf16x8_t a = frag(xt, p0, wv, S, tx);        // LDS pixel of input x = S ox + tx - 1 is S ox + tx
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            acc[ty * 3 + tx][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[nf], acc[ty * 3 + tx][nf], 0, 0, 0);
        }
      }
    }
  }
  float* const dst = to_ws ? out + (size_t)blockIdx.x * 9 * C * C : out;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* p = dst + ((size_t)tp * C + wv * 16 + kq * 4 + r) * C + nf * 16 + li;
        *p = (!to_ws && accumulate) ? (*p + acc[tp][nf][r]) : acc[tp][nf][r];
      }
}

// dW (+)= sum over the workgroups' partial tensors, in workgroup order
__global__ void k_img_wgrad_reduce(const float* __restrict__ ws, int parts, int n, float* __restrict__ dW, int accumulate) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  int p = 0;
  for (; p + 8 <= parts; p += 8) {                            // eight loads in flight, added in order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(p + u) * n + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; p < parts; ++p) s += ws[(size_t)p * n + e];
  dW[e] = accumulate ? dW[e] + s : s;
}

static int ES_OPT_IMG_WGRAD = 1;
static int ES_OPT_ROWS_WGRAD_320 = 1;                        // ... and for 320 output columns as one tile (key 44)
static int ES_OPT_ROWS_WGRAD_MIN_ROWS = 500000;              // 1x1 streaming kernel: rows from which it is taken whatever the width (key 43)
static int ES_OPT_IMG_WGRAD_WGS32 = 400, ES_OPT_IMG_WGRAD_WGS64 = 160;      // workgroups a launch aims for (partial tensors: 36 / 147 KB each)
extern "C" int es_img_wgrad_set_option(int key, int value) {
  if (key == 40) { ES_OPT_IMG_WGRAD = value; return 0; }
  if (key == 41) { ES_OPT_IMG_WGRAD_WGS32 = value; return 0; }
  if (key == 42) { ES_OPT_IMG_WGRAD_WGS64 = value; return 0; }
  if (key == 43) { ES_OPT_ROWS_WGRAD_MIN_ROWS = value; return 0; }
  if (key == 44) { ES_OPT_ROWS_WGRAD_320 = value; return 0; }
  return -1;
}

// ------------------------------------------------------------------ 1x1 layers: dW[Cin][Cout] = X^T . dY over contiguous rows
// The image backbone's 1x1 convolutions (Bottleneck.conv1 / conv3: 64 -> 32, 128 -> 32, 32 -> 128, ... on 10^5 .. 10^6 pixel rows) ran on
// the same ring / gather kernel as the 3x3 ones (k_spconv_wgrad_bf16<1, 0>: 1.0 - 2.8 TB/s of operand bytes).  With the identity map there
// is nothing to gather: a workgroup streams a slice of rows in 64-row steps -- X tile [64][CI] by LDS-DMA, dY tile [64][CO] f32 through
// registers (prefetched one step ahead, rounded to bf16) -- into two LDS buffers and reads both operands transposed; a (CI x CO) tile of dW
// per workgroup (128 x 128 at most: 16 accumulator fragments per wave), partial tensors per slice added in slice order.
template <int CI, int CO>
__global__ __launch_bounds__(256) void k_rows_wgrad1(const unsigned short* __restrict__ Xh, int ldx, const float* __restrict__ dY, int ldy,
                                                     int n, int Cin, int Cout, int rows_per_slice, float* __restrict__ out, int to_ws,
                                                     int accumulate) {
  constexpr int FA = CI / 16, FB = CO / 16;
  constexpr bool CI_SPLIT = FA >= 4;                         // waves split the ci blocks (else the co blocks)
  constexpr int WA = CI_SPLIT ? FA / 4 : FA, WB = CI_SPLIT ? FB : FB / 4;
  static_assert(CI_SPLIT || FB >= 4, "a 32-channel side needs at least 64 on the other");
  constexpr int RA = CI * 2, RBY = CO * 2;                   // bytes per row of the X / dY tile
  constexpr int XT = 64 * RA, YT = 64 * RBY;
  constexpr int NPX = 64 * (RA / 16) / 256;                  // LDS-DMA pieces per thread and step (CI = 32: 1, 64: 2, 128: 4)
  constexpr int NLY = 64 * CO / 4 / 256;                     // float4 loads per thread and step
  static_assert(NPX >= 1 && NLY >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * XT + 2 * YT];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int c0 = blockIdx.y * CI, n0 = blockIdx.z * CO;
  const int r0 = blockIdx.x * rows_per_slice, r1 = min(n, r0 + rows_per_slice);
  f32x4 acc[WA][WB];
#pragma unroll
  for (int a = 0; a < WA; ++a)
#pragma unroll
    for (int b = 0; b < WB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto issue_x = [&](int buf, int row) {                     // rows row .. row + 63 (past r1: zero granules)
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const int e = (j * 4 + wv) * 64 + lane, rr = e / (RA / 16), g = e - rr * (RA / 16);
      const unsigned short* p = row + rr < r1 ? (Xh + (size_t)(row + rr) * ldx + c0 + g * 8) : g_iw_zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + buf * XT + (j * 4 + wv) * 1024), 16, 0, 0);
    }
  };
  float4 yreg[NLY];
  auto load_y = [&](int row) {
#pragma unroll
    for (int j = 0; j < NLY; ++j) {
      const int e = j * 256 + t, rr = e / (CO / 4), c4 = e - rr * (CO / 4);
      yreg[j] = row + rr < r1 ? *(const float4*)(dY + (size_t)(row + rr) * ldy + n0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_y = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NLY; ++j) {
      const int e = j * 256 + t, rr = e / (CO / 4), c4 = e - rr * (CO / 4);
      uint2 v;
      v.x = es_pack_bf16(yreg[j].x, yreg[j].y);
      v.y = es_pack_bf16(yreg[j].z, yreg[j].w);
      *(uint2*)(smem + 2 * XT + buf * YT + rr * RBY + c4 * 8) = v;
    }
  };
  auto frag = [&](const unsigned char* tile, int rb, int p0, int cb) -> bf16x8_t {      // rows p0 + kq * 8 .. of channel cb * 16 + li
    s16x4_t h[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int pr = p0 + kq * 8 + r * 4 + (li >> 2);
      h[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + pr * rb + cb * 32 + (li & 3) * 8));
    }
    s16x8_t v = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  };
  if (r0 < r1) {
    issue_x(0, r0);
    load_y(r0);
    int buf = 0;
    for (int row = r0; row < r1; row += 64, buf ^= 1) {
      store_y(buf);                                          // (this buffer was last read two steps ago: a barrier has passed since)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the X tile have landed ...
      __syncthreads();                                        // ... everybody's; the dY tile is complete; the other buffer has been consumed
      if (row + 64 < r1) { issue_x(buf ^ 1, row + 64); load_y(row + 64); }
      const unsigned char* xt = smem + buf * XT;
      const unsigned char* yt = smem + 2 * XT + buf * YT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t a[WA], b[WB];
#pragma unroll
        for (int i = 0; i < WA; ++i) a[i] = frag(xt, RA, ks * 32, CI_SPLIT ? wv * WA + i : i);
#pragma unroll
        for (int i = 0; i < WB; ++i) b[i] = frag(yt, RBY, ks * 32, CI_SPLIT ? i : wv * WB + i);
#pragma unroll
        for (int i = 0; i < WA; ++i)
#pragma unroll
          for (int k = 0; k < WB; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[k], acc[i][k], 0, 0, 0);
      }
    }
  }
  // partial tensor of this slice: the full [Cin][Cout] layout (tiles of other workgroups interleave)
  float* const dst = to_ws ? out + (size_t)blockIdx.x * Cin * Cout : out;
#pragma unroll
  for (int i = 0; i < WA; ++i)
#pragma unroll
    for (int k = 0; k < WB; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = c0 + (CI_SPLIT ? wv * WA + i : i) * 16 + kq * 4 + r, co = n0 + (CI_SPLIT ? k : wv * WB + k) * 16 + li;
        float* p = dst + (size_t)ci * Cout + co;
        *p = (!to_ws && accumulate) ? (*p + acc[i][k][r]) : acc[i][k][r];
      }
}

static bool rows_wgrad_plan(int n, int Cin, int Cout, int& ci, int& co, int& slices, int& rows) {
  if (!ES_OPT_IMG_WGRAD || n < 4096) return false;
  // A/B against the ring kernel (profiles/r6h_imgwgrad_ab.txt): that one already streams these launches at 2 - 3 TB/s; this kernel wins from
  // 256 input channels (38 -> 27 us) or half a million rows (144 -> 98 us, 160 -> 129 us) and loses a few us below
  // ... and 320 output columns (fcaf3d_head.py: a level's class / box / centerness outputs as one GEMM): ONE (128 x 320) tile per slice, both
  // operands read once (the 64-column tiles of the ring kernel read the input rows five times: 369 us on 352 k rows)
  const bool whole = ES_OPT_ROWS_WGRAD_320 && Cout == 320 && Cin % 128 == 0;
  if (!(Cin >= 256 || n >= ES_OPT_ROWS_WGRAD_MIN_ROWS || whole)) return false;
  ci = Cin % 128 == 0 ? 128 : Cin % 64 == 0 ? 64 : Cin == 32 ? 32 : 0;
  co = whole ? 320 : Cout % 128 == 0 ? 128 : Cout % 64 == 0 ? 64 : Cout == 32 ? 32 : 0;
  if (!ci || !co || (ci == 32 && co == 32) || Cin > 512 || Cout > 512) return false;
  const long long tiles = (long long)(Cin / ci) * (Cout / co) * (whole ? 2 : 1);      // (one workgroup per CU: half the slices)
  // slices: enough workgroups to stream at full bandwidth, few enough that the partial tensors stay a fraction of the operand bytes
  const long long in_bytes = (long long)n * (Cin * 2 + Cout * 4), dw_bytes = (long long)Cin * Cout * 4;
  long long s = in_bytes / (4 * dw_bytes);
  const long long cap = 1024 / tiles > 1 ? 1024 / tiles : 1;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  rows = es_cdiv(es_cdiv(n, (int)s), 64) * 64;
  slices = es_cdiv(n, rows);
  return true;
}

extern "C" size_t es_rows_wgrad1_workspace_floats(int n, int Cin, int Cout) {
  int ci, co, slices, rows;
  if (!rows_wgrad_plan(n, Cin, Cout, ci, co, slices, rows)) return 0;
  return (size_t)slices * Cin * Cout;
}

extern "C" int es_rows_wgrad1_bf16(const void* Xh, int ldx, const float* dY, int ldy, int n, int Cin, int Cout, float* dW, int accumulate,
                                   float* ws, size_t ws_floats, void* stream) {
  int ci, co, slices, rows;
  if (!rows_wgrad_plan(n, Cin, Cout, ci, co, slices, rows)) return -4;
  if ((ldx % 8) || (ldy % 4) || ((((uintptr_t)Xh) | ((uintptr_t)dY)) & 15)) return -4;
  if ((long long)n * (ldx > ldy ? ldx : ldy) >= (1ll << 31)) return -4;
  if (slices > 1 && (ws == nullptr || ws_floats < (size_t)slices * Cin * Cout)) return -5;
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* X = (const unsigned short*)Xh;
  float* out = slices > 1 ? ws : dW;
  const int to_ws = slices > 1;
  dim3 grid(slices, Cin / ci, Cout / co);
#define RW_LAUNCH(CI_, CO_) hipLaunchKernelGGL((k_rows_wgrad1<CI_, CO_>), grid, dim3(256), 0, st, X, ldx, dY, ldy, n, Cin, Cout, rows, out, to_ws, accumulate)
  if (ci == 128 && co == 320) RW_LAUNCH(128, 320);
  else if (ci == 128 && co == 128) RW_LAUNCH(128, 128);
  else if (ci == 128 && co == 64) RW_LAUNCH(128, 64);
  else if (ci == 128 && co == 32) RW_LAUNCH(128, 32);
  else if (ci == 64 && co == 128) RW_LAUNCH(64, 128);
  else if (ci == 64 && co == 64) RW_LAUNCH(64, 64);
  else if (ci == 64 && co == 32) RW_LAUNCH(64, 32);
  else if (ci == 32 && co == 128) RW_LAUNCH(32, 128);
  else RW_LAUNCH(32, 64);
#undef RW_LAUNCH
  ES_CHECK_LAUNCH();
  if (slices > 1) {
    const int nn = Cin * Cout;
    hipLaunchKernelGGL(k_img_wgrad_reduce, dim3(es_cdiv(nn, 256)), dim3(256), 0, st, ws, slices, nn, dW, accumulate);
    ES_CHECK_LAUNCH();
  }
  return 0;
}

static bool img_wgrad_plan(int n_img, int H, int W, int C, int stride, int& wp, int& rows, int& bands) {
  if (!ES_OPT_IMG_WGRAD || n_img <= 0 || H <= 0 || W <= 0) return false;
  if (stride != 1 && !(stride == 2 && H % 2 == 0 && W % 2 == 0)) return false;
  const int Ho = H / stride, Wo = W / stride;
  if (C == 32 && Wo <= 128) wp = Wo <= 64 ? 64 : 128;
  else if (C == 64 && Wo <= 64) wp = Wo <= 32 ? 32 : 64;
  else return false;
  if (stride == 2 && wp == 128) return false;               // (8 ring slots of 258 pixels: 132 KB)
  // workgroups aimed for: enough to fill the chip, few enough that the partial tensors (36 / 147 KB each) stay small next to the
  // operands; with many images (the grounder: 240) three bands per image at C = 32 (A/B: profiles/r6g_imgwgrad_ab.txt)
  int target = C == 32 ? ES_OPT_IMG_WGRAD_WGS32 : ES_OPT_IMG_WGRAD_WGS64;
  if (C == 32 && target < 3 * n_img) target = 3 * n_img;
  bands = target / n_img;
  if (bands < 1) bands = 1;
  if (bands > Ho) bands = Ho;
  rows = es_cdiv(Ho, bands);
  bands = es_cdiv(Ho, rows);
  return true;
}

extern "C" size_t es_img_wgrad9_workspace_floats(int n_img, int H, int W, int C, int stride) {
  int wp, rows, bands;
  if (!img_wgrad_plan(n_img, H, W, C, stride, wp, rows, bands)) return 0;
  return (size_t)n_img * bands * 9 * C * C;
}

extern "C" int es_img_wgrad9_bf16(const void* Xh, int ldx, const float* dY, int ldy, int n_img, int H, int W, int C, int stride,
                                  float* dW, int accumulate, float* ws, size_t ws_floats, void* stream) {
  int wp, rows, bands;
  if (!img_wgrad_plan(n_img, H, W, C, stride, wp, rows, bands)) return -4;
  if ((ldx % 8) || (ldy % 4) || ((((uintptr_t)Xh) | ((uintptr_t)dY)) & 15)) return -4;
  if ((long long)n_img * H * W * (ldx > ldy ? ldx : ldy) >= (1ll << 31)) return -4;
  const int parts = n_img * bands;
  if (parts > 1 && (ws == nullptr || ws_floats < (size_t)parts * 9 * C * C)) return -5;
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* X = (const unsigned short*)Xh;
  float* out = parts > 1 ? ws : dW;
  const int to_ws = parts > 1;
#define IW_LAUNCH(C_, WP_, S_)                                                                                              \
  hipLaunchKernelGGL((k_img_wgrad9<C_, WP_, S_>), dim3(parts), dim3(C_ * 4), 0, st, X, ldx, dY, ldy, H, W, rows, bands, out, \
                     to_ws, accumulate)
  if (stride == 1) {
    if (C == 32 && wp == 128) IW_LAUNCH(32, 128, 1);
    else if (C == 32) IW_LAUNCH(32, 64, 1);
    else if (wp == 64) IW_LAUNCH(64, 64, 1);
    else IW_LAUNCH(64, 32, 1);
  } else {
    if (C == 32) IW_LAUNCH(32, 64, 2);
    else if (wp == 64) IW_LAUNCH(64, 64, 2);
    else IW_LAUNCH(64, 32, 2);
  }
#undef IW_LAUNCH
  ES_CHECK_LAUNCH();
  if (parts > 1) {
    const int n = 9 * C * C;
    hipLaunchKernelGGL(k_img_wgrad_reduce, dim3(es_cdiv(n, 256)), dim3(256), 0, st, ws, parts, n, dW, accumulate);
    ES_CHECK_LAUNCH();
  }
  return 0;
}
