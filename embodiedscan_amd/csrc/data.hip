// Data-side kernels pulled on-device (rows A1-A3, A18):
//   * es_depth_to_points : depth pixel -> camera point (ConvertRGBDToPoints / points_img2cam,
//     embodiedscan/datasets/transforms/points.py:30-81, structures/bbox_3d/utils.py:335-368)
//     -> global frame (AggregateMultiViewPoints, datasets/transforms/multiview.py:139-169)
//     -> RandomFlip3D / GlobalRotScaleTrans (datasets/transforms/augmentation.py:87-139,322-348)
//     for the (view, pixel) pairs chosen by PointSample (points.py:155-213); un-projecting only the
//     sampled pixels is identical to un-projecting all 307 200 and indexing.
//   * es_preprocess_img  : uint8 BGR NCHW -> f32 RGB normalised, channels-last
//     (Det3DDataPreprocessor.preprocess_img, models/data_preprocessors/data_preprocessor.py:249-264)
#include "common.h"
#include "../../include/es_hip.h"

// mats per view: [0..15] inv(pad4(K))  [16..31] inv(global2cam) ; aug: [0..8] rot_mat_T, [9] scale, [10..12] trans,
// [13] hflip, [14] vflip
__global__ void k_depth_to_points(const float* __restrict__ depth, int H, int W, const int* __restrict__ sel_view,
                                  const int* __restrict__ sel_pix, int n, const float* __restrict__ mats,
                                  const float* __restrict__ aug, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int v = sel_view[i], p = sel_pix[i];
  float d = depth[((size_t)v * H) * W + p];
  float u = (float)(p % W), w_ = (float)(p / W);
  const float* Ki = mats + v * 32;
  const float* Ei = Ki + 16;
  float h0 = u * d, h1 = w_ * d, h2 = d, h3 = 1.f;
  // homo @ inv(K)^T  -> camera point (first three components)
  float c[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) c[r] = fmaf(h3, Ki[r * 4 + 3], fmaf(h2, Ki[r * 4 + 2], fmaf(h1, Ki[r * 4 + 1], h0 * Ki[r * 4])));
  float g[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) g[r] = fmaf(1.f, Ei[r * 4 + 3], fmaf(c[2], Ei[r * 4 + 2], fmaf(c[1], Ei[r * 4 + 1], c[0] * Ei[r * 4])));
  if (aug[13] != 0.f) g[0] = -g[0];
  if (aug[14] != 0.f) g[1] = -g[1];
  float q[3];
#pragma unroll
  for (int cidx = 0; cidx < 3; ++cidx) q[cidx] = fmaf(g[2], aug[6 + cidx], fmaf(g[1], aug[3 + cidx], g[0] * aug[cidx]));
#pragma unroll
  for (int cidx = 0; cidx < 3; ++cidx) out[(size_t)i * 3 + cidx] = q[cidx] * aug[9] + aug[10 + cidx];
}
extern "C" int es_depth_to_points(const float* depth, int H, int W, const int* sel_view, const int* sel_pix, int n,
                                  const float* mats, const float* aug, float* out, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_depth_to_points, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, depth, H, W,
                     sel_view, sel_pix, n, mats, aug, out);
  ES_CHECK_LAUNCH();
  return 0;
}

// in: (NI, 3, H, W) u8 ; out: (NI, Hp, Wp, 3) f32 normalised, channels-last (feeds the conv engine).  flip != 0: output
// channel c reads input channel 2-c (bgr_to_rgb / rgb_to_bgr, data_preprocessor.py:256-258).  Rows >= H / columns >= W of
// the padded output (pad_size_divisor, bottom/right padding of multiview_img_stack_batch, data_preprocessors/utils.py:9-63)
// hold pad_value, which the reference applies AFTER normalisation.
__global__ void k_preprocess_img(const unsigned char* __restrict__ in, int NI, int H, int W, int Hp, int Wp, int flip,
                                 float m0, float m1, float m2, float s0, float s1, float s2, float pad_value,
                                 float* __restrict__ out) {
  size_t HWp = (size_t)Hp * Wp, HW = (size_t)H * W;
  size_t tot = (size_t)NI * HWp;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    size_t im = e / HWp, px = e - im * HWp;
    int y = (int)(px / Wp), x = (int)(px - (size_t)y * Wp);
    float* o = out + e * 3;
    if (y >= H || x >= W) {
      o[0] = pad_value; o[1] = pad_value; o[2] = pad_value;
      continue;
    }
    const unsigned char* b = in + im * 3 * HW + (size_t)y * W + x;
    float c0 = (float)b[0], c1 = (float)b[HW], c2 = (float)b[2 * HW];
    if (flip) { float t = c0; c0 = c2; c2 = t; }
    o[0] = (c0 - m0) / s0;
    o[1] = (c1 - m1) / s1;
    o[2] = (c2 - m2) / s2;
  }
}
extern "C" int es_preprocess_img(const unsigned char* img, int n_img, int H, int W, int Hp, int Wp, int flip,
                                 const float* mean, const float* std, float pad_value, float* out, void* stream) {
  if (n_img <= 0) return 0;
  if (Hp < H || Wp < W) return -2;
  hipLaunchKernelGGL(k_preprocess_img, dim3(4096), dim3(256), 0, (hipStream_t)stream, img, n_img, H, W, Hp, Wp, flip,
                     mean[0], mean[1], mean[2], std[0], std[1], std[2], pad_value, out);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ N4: Resize(keep_ratio=False) of the decoded frames
// (V,H,W,3) interleaved u8 -> (V,3,h,w) planar u8, bilinear with OpenCV's 8-bit fixed-point rule (mmcv.imresize ->
// cv2.resize INTER_LINEAR): 11-bit coefficients from the host tables (xofs/ialpha, yofs/ibeta: first source index and the
// two weights per output column / row, built once per size pair by pipeline.resize_tables), horizontal pass in int32,
// vertical pass  ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2 >> 2.  Integer only: bit-exact against oracle/resize.py.
__global__ void k_resize_u8(const unsigned char* __restrict__ src, int V, int H, int W, const int* __restrict__ xofs,
                            const short* __restrict__ ialpha, const int* __restrict__ yofs,
                            const short* __restrict__ ibeta, int h, int w, unsigned char* __restrict__ dst) {
  size_t hw = (size_t)h * w, tot = (size_t)V * hw;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    int v = (int)(e / hw);
    int px = (int)(e - (size_t)v * hw);
    int y = px / w, x = px - y * w;
    int sx0 = xofs[x], sx1 = min(sx0 + 1, W - 1), sy0 = yofs[y], sy1 = min(sy0 + 1, H - 1);
    int a0 = ialpha[2 * x], a1 = ialpha[2 * x + 1], b0 = ibeta[2 * y], b1 = ibeta[2 * y + 1];
    const unsigned char* r0 = src + ((size_t)v * H + sy0) * W * 3;
    const unsigned char* r1 = src + ((size_t)v * H + sy1) * W * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int S0 = r0[sx0 * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
      int S1 = r1[sx0 * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
      int o = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
      dst[((size_t)v * 3 + c) * hw + px] = (unsigned char)min(max(o, 0), 255);
    }
  }
}
extern "C" int es_resize_u8(const unsigned char* src, int V, int H, int W, const int* xofs, const short* ialpha,
                            const int* yofs, const short* ibeta, int h, int w, unsigned char* dst, void* stream) {
  if (V <= 0 || h <= 0 || w <= 0) return 0;
  if (H <= 0 || W <= 0) return -2;
  size_t tot = (size_t)V * h * w;
  int g = (int)((tot + 255) / 256);
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_resize_u8, dim3(g), dim3(256), 0, (hipStream_t)stream, src, V, H, W, xofs, ialpha, yofs, ibeta, h,
                     w, dst);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ ResNet stem: 7x7 s2 p3 conv (3 -> Cout<=32) + frozen-BN
// affine + ReLU in one direct kernel.  The stem is frozen (frozen_stages=1, configs/detection/mv-det3d_...py:29) so only the
// forward pass exists.  A workgroup computes a 16x16 output tile from a 37x37x3 input patch staged in LDS together with
// the [49][3][Cout] weights; this replaces two generic-engine launches that padded C_in = 3 to a 16-wide K chunk.
#define ST_T 16
#define ST_P (ST_T * 2 + 5)
template <int CO>
__global__ __launch_bounds__(256) void k_stem_conv(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   int H, int W, int Ho, int Wo, float* __restrict__ y) {
  __shared__ float patch[ST_P * ST_P * 3];
  __shared__ float ws[49 * 3 * CO];
  const int t = threadIdx.x;
  const int im = blockIdx.z, ty0 = blockIdx.y * ST_T, tx0 = blockIdx.x * ST_T;
  for (int e = t; e < 49 * 3 * CO; e += 256) ws[e] = w[e];
  const int hi0 = ty0 * 2 - 3, wi0 = tx0 * 2 - 3;
  const float* xi = x + (size_t)im * H * W * 3;
  for (int e = t; e < ST_P * ST_P * 3; e += 256) {
    int c = e % 3, p = e / 3;
    int pw = p % ST_P, ph = p / ST_P;
    int hi = hi0 + ph, wi = wi0 + pw;
    patch[e] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? xi[((size_t)hi * W + wi) * 3 + c] : 0.f;
  }
  __syncthreads();
  const int oy = t / ST_T, ox = t % ST_T;
  const int ho = ty0 + oy, wo = tx0 + ox;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  for (int ky = 0; ky < 7; ++ky)
    for (int kx = 0; kx < 7; ++kx) {
      const float* pp = &patch[((oy * 2 + ky) * ST_P + ox * 2 + kx) * 3];
      const float* wp = &ws[(ky * 7 + kx) * 3 * CO];
      float v0 = pp[0], v1 = pp[1], v2 = pp[2];
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] = fmaf(v2, wp[2 * CO + c], fmaf(v1, wp[CO + c], fmaf(v0, wp[c], acc[c])));
    }
  if (ho < Ho && wo < Wo) {
    float* yo = y + (((size_t)im * Ho + ho) * Wo + wo) * CO;
#pragma unroll
    for (int c = 0; c < CO; ++c) yo[c] = fmaxf(acc[c] * scale[c] + shift[c], 0.f);
  }
}
extern "C" int es_stem_conv_fwd(const float* x, const float* w, const float* scale, const float* shift, int n_img, int H,
                                int W, int Cout, float* y, void* stream) {
  int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  dim3 grid(es_cdiv(Wo, ST_T), es_cdiv(Ho, ST_T), n_img);
  if (Cout == 16)
    hipLaunchKernelGGL(k_stem_conv<16>, grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, H, W, Ho, Wo, y);
  else if (Cout == 32)
    hipLaunchKernelGGL(k_stem_conv<32>, grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, H, W, Ho, Wo, y);
  else if (Cout == 64)
    hipLaunchKernelGGL(k_stem_conv<64>, grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, H, W, Ho, Wo, y);
  else
    return -6;
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ stem + max pool in one launch (round 6)
// ResNet stem as above followed by MaxPool2d(3, stride 2, padding 1) (mmdet ResNet.forward: conv1 - norm1 - relu - maxpool), forward only
// (frozen_stages >= 1), bf16 rows out.  The pair used to be two launches with 393 MB of f32 stem rows written and read back through a
// 9-wide map (446 + 291 us for 80 views of 480 x 640).  Here a workgroup owns 7 x 8 POOLED pixels: its 15 x 17 stem pixels (one per
// thread, 255 of 256; the one-pixel overlap between tiles is recomputed: x 1.14) from a 35 x 39 x 3 input patch in LDS, the stem values
// through LDS into the 3 x 3 maxima.  The arithmetic per stem value is the old kernel's, term for term (f32 FMAs in (ky, kx, c_in) order)
// -- the bf16 rows are bit-identical to the pair's -- but the weights are read with wave-uniform addresses (scalar loads, no LDS traffic)
// and two output channels share one v_pk_fma_f32.
#define SP_PY 7
#define SP_SY (2 * SP_PY + 1)
#define SP_IY (2 * SP_SY + 5)
typedef float es_f2 __attribute__((ext_vector_type(2)));
// PX = pooled pixels per tile row: 8 (15 x 17 stem pixels, one per thread).  PX = 16 (15 x 33 = 495 stem pixels, two per thread sharing the
// scalar-loaded weight pairs, overlap x 1.10) was measured: 207 us against 161 on 20 views (169 VGPRs, 61 KB LDS, SGPR spills) -- not instantiated.
template <int CO, int PX>
__global__ __launch_bounds__(256) void k_stem_pool(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift, int H, int W,
                                                   int Ho, int Wo, int Hp, int Wp, unsigned short* __restrict__ y) {
  constexpr int SX = 2 * PX + 1, IX = 2 * SX + 5, NS = SP_SY * SX, NPT = (NS + 255) / 256, CI_UNROLL = CO <= 16 ? 3 : 1;
  __shared__ float patch[SP_IY * IX * 3];
  __shared__ __attribute__((aligned(16))) float stemS[NS * CO];
  const int t = threadIdx.x;
  const int im = blockIdx.z, py0 = blockIdx.y * SP_PY, px0 = blockIdx.x * PX;
  const int hs0 = 2 * py0 - 1, ws0 = 2 * px0 - 1;          // first stem pixel of the tile (pool padding 1)
  const int hi0 = 2 * hs0 - 3, wi0 = 2 * ws0 - 3;          // first input pixel (stem padding 3)
  const float* xi = x + (size_t)im * H * W * 3;
  for (int e = t; e < SP_IY * IX * 3; e += 256) {
    const int ph = e / (IX * 3), r = e - ph * (IX * 3), pw = r / 3, c = r - pw * 3;
    const int hi = hi0 + ph, wi = wi0 + pw;
    patch[e] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? xi[((size_t)hi * W + wi) * 3 + c] : 0.f;
  }
  __syncthreads();
  {
    int po[NPT];                                            // patch offset of this thread's stem pixels (pixels past the tile: the first one's)
    bool in_tile[NPT], ok[NPT];
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
      const int sp = t + q * 256;
      in_tile[q] = sp < NS;
      const int spc = in_tile[q] ? sp : t;
      const int sy = spc / SX, sx = spc - sy * SX;
      const int hs = hs0 + sy, ws_ = ws0 + sx;
      ok[q] = in_tile[q] && hs >= 0 && hs < Ho && ws_ >= 0 && ws_ < Wo;
      po[q] = (sy * 2 * IX + sx * 2) * 3;
    }
    es_f2 acc[NPT][CO / 2];
#pragma unroll
    for (int q = 0; q < NPT; ++q)
#pragma unroll
      for (int j = 0; j < CO / 2; ++j) acc[q][j] = (es_f2){0.f, 0.f};
    for (int ky = 0; ky < 7; ++ky) {
      const float* wr = w + ky * 7 * 3 * CO;
#pragma unroll 1
      for (int kx = 0; kx < 7; ++kx)             // (rolled: a whole ky row of weights in flight wants 336 SGPRs and spilled 48 of them)
#pragma unroll CI_UNROLL
        for (int ci = 0; ci < 3; ++ci) {         // (32 / 64 output channels: one input channel's weights at a time fill the SGPR file)
          const float* wp = wr + (kx * 3 + ci) * CO;
          es_f2 vv[NPT];
#pragma unroll
          for (int q = 0; q < NPT; ++q) {
            const float v = patch[po[q] + ky * IX * 3 + kx * 3 + ci];
            vv[q] = (es_f2){v, v};
          }
#pragma unroll
          for (int j = 0; j < CO / 2; ++j) {
            const es_f2 ww = (es_f2){wp[2 * j], wp[2 * j + 1]};
#pragma unroll
            for (int q = 0; q < NPT; ++q) acc[q][j] = __builtin_elementwise_fma(vv[q], ww, acc[q][j]);
          }
        }
    }
#pragma unroll
    for (int q = 0; q < NPT; ++q)
      if (in_tile[q]) {
        float* so = &stemS[(t + q * 256) * CO];
#pragma unroll
        for (int j = 0; j < CO / 2; ++j) {                   // outside the stem grid: the pool's padding
          so[2 * j] = ok[q] ? fmaxf(acc[q][j].x * scale[2 * j] + shift[2 * j], 0.f) : -INFINITY;
          so[2 * j + 1] = ok[q] ? fmaxf(acc[q][j].y * scale[2 * j + 1] + shift[2 * j + 1], 0.f) : -INFINITY;
        }
      }
  }
  __syncthreads();
  for (int e = t; e < SP_PY * PX * CO; e += 256) {
    const int p = e / CO, c = e - p * CO;
    const int py = p / PX, px = p - py * PX;
    if (py0 + py >= Hp || px0 + px >= Wp) continue;
    float m = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, stemS[((2 * py + dy) * SX + 2 * px + dx) * CO + c]);
    uint32_t u = __float_as_uint(m == -INFINITY ? 0.f : m);
    u += 0x7fffu + ((u >> 16) & 1u);                       // RNE to bf16 (finite values)
    y[(((size_t)im * Hp + py0 + py) * Wp + px0 + px) * CO + c] = (unsigned short)(u >> 16);
  }
}
extern "C" int es_stem_pool_fwd(const float* x, const float* w, const float* scale, const float* shift, int n_img, int H, int W,
                                int Cout, void* y_bf16, void* stream) {
  if (n_img <= 0) return 0;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
  if (Hp <= 0 || Wp <= 0 || n_img > 65535) return -4;
  const int px = 8;
  dim3 grid(es_cdiv(Wp, px), es_cdiv(Hp, SP_PY), n_img);
#define SP_LAUNCH(CO_, PX_) hipLaunchKernelGGL((k_stem_pool<CO_, PX_>), grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, H, W, Ho, Wo, Hp, Wp, (unsigned short*)y_bf16)
  if (Cout == 16) SP_LAUNCH(16, 8);
  else if (Cout == 32) SP_LAUNCH(32, 8);
  else if (Cout == 64) SP_LAUNCH(64, 8);            // (the occupancy detector's full-width ResNet-50: 81 KB of LDS, one workgroup per CU)
#undef SP_LAUNCH
  else
    return -6;
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ N4: PointSample on the device (counter-based draws)
// datasets/transforms/points.py:155-213 draws `np.random.choice(range(n), k, replace=False)` per depth frame and once more over
// the aggregated cloud: a uniform k-subset in uniform random order.  On the host that is a Fisher-Yates over all ~3e5 valid pixels
// of every frame -- 52 % of the loader's time per scan (profiles/r3_loader_profile.txt).  Here every element gets a 30-bit key
// from a counter-based generator (splitmix64 of seed, stream and index: no sequential state, any element in any order), the k
// LARGEST keys are selected (es_topk_mask_ws on the keys' float encodings: positive finite floats order like their bit patterns;
// invalid pixels carry -1) and emitted in descending key order (es_sort_u64 on 2^30 - 1 - key).  The same law as the reference's
// draw, not its numpy stream; oracle/draws.py restates these kernels integer for integer (tests: exact equality + frequencies).
__device__ inline uint64_t es_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline uint32_t es_draw_key(uint64_t seed, uint32_t stream, uint64_t index) {
  const uint64_t s = seed ^ ((uint64_t)stream * 0x9E3779B97F4A7C15ull);
  return (uint32_t)(es_splitmix64(s + index * 0xD1B54A32D192ED03ull) >> 34);
}
// per (view, pixel): values = the key's float encoding (valid depth) or -1; keys = view << 54 | (2^30 - 1 - key) << 24 | pixel
__global__ void k_draw_keys(const float* __restrict__ depth, int V, int HW, uint64_t seed, float* __restrict__ values,
                            int64_t* __restrict__ keys) {
  const size_t tot = (size_t)V * HW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    const int v = (int)(e / HW), p = (int)(e - (size_t)v * HW);
    const uint32_t k = es_draw_key(seed, (uint32_t)v, (uint64_t)p);
    values[e] = depth[e] != 0.f ? __uint_as_float(k) : -1.f;
    keys[e] = ((int64_t)v << 54) | ((int64_t)(0x3fffffffu - k) << 24) | (int64_t)p;
  }
}
extern "C" int es_draw_keys(const float* depth, int V, int HW, size_t seed, float* values, int64_t* keys, void* stream) {
  if (V <= 0 || HW <= 0) return 0;
  if (V > 256 || HW > (1 << 24)) return -4;
  int g = es_cdiv((long long)V * HW, 256);
  hipLaunchKernelGGL(k_draw_keys, dim3(g > 16384 ? 16384 : g), dim3(256), 0, (hipStream_t)stream, depth, V, HW, (uint64_t)seed, values,
                     keys);
  ES_CHECK_LAUNCH();
  return 0;
}
// the aggregate draw: element j of `stream`: values = key encoding, keys = (2^30 - 1 - key) << 24 | j
__global__ void k_draw_keys_index(int n, uint64_t seed, uint32_t stream, float* __restrict__ values, int64_t* __restrict__ keys) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t k = es_draw_key(seed, stream, (uint64_t)j);
  values[j] = __uint_as_float(k);
  keys[j] = ((int64_t)(0x3fffffffu - k) << 24) | (int64_t)j;
}
extern "C" int es_draw_keys_index(int n, size_t seed, int stream_id, float* values, int64_t* keys, void* stream) {
  if (n <= 0) return 0;
  if (n > (1 << 24)) return -4;
  hipLaunchKernelGGL(k_draw_keys_index, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, (uint64_t)seed, (uint32_t)stream_id,
                     values, keys);
  ES_CHECK_LAUNCH();
  return 0;
}
// sorted keys -> (view, pixel) lists; with in_view / in_pix: the low 24 bits index those lists (aggregate draw)
__global__ void k_draw_unpack(const int64_t* __restrict__ keys, int n, const int* __restrict__ in_view, const int* __restrict__ in_pix,
                              int* __restrict__ out_view, int* __restrict__ out_pix) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t k = keys[i];
  const int low = (int)(k & 0xffffff);
  if (in_view) { out_view[i] = in_view[low]; out_pix[i] = in_pix[low]; }
  else { out_view[i] = (int)(k >> 54); out_pix[i] = low; }
}
extern "C" int es_draw_unpack(const int64_t* keys, int n, const int* in_view, const int* in_pix, int* out_view, int* out_pix,
                              void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_draw_unpack, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, keys, n, in_view, in_pix, out_view, out_pix);
  ES_CHECK_LAUNCH();
  return 0;
}
