// 3x3 convolutions of the image backbone by address arithmetic (round 6; SURVEY 8a row A7: mmdet.ResNet depth 50, base_channels 16,
// configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:24-34 -- Bottleneck.conv2 of layer1 .. layer3: 16 / 32 / 64
// channels on 120^2 / 60^2 / 30^2 feature maps, 80 images per mv-3ddet step; the fused launches of engine.conv_affine):
//   MODE 0  forward:  Y = act((X * W) * scale[c] + shift[c])   X bf16 rows, Y bf16 (or f32) rows, stride 1 or 2, pad 1
//   MODE 1  gated data gradient of a stride-1 layer:  dX = (A > 0) ? (G * W^T mirrored) * scale[c] : 0
//           G f32 rows (rounded to bf16 on the way into LDS, like every gradient shadow), A = the layer input's bf16 activation rows
// The map kernels (k_spconv_bf16_fast / k_spconv_bf16<64>, spconv.hip) gather every tap's rows through a 9-wide int32 map: 59 us for
// a 37 MB problem (32 channels), 153 us at 16 channels (no fast path below 32 input channels).  On an image grid the neighbour of pixel
// x under tap tx is pixel x + tx - 1 of an image row that is already in LDS (the scheme of imgwgrad.hip):
//   * a workgroup owns a band of output rows of ONE image; the input rows live in a 4- (stride 2: 8-) slot LDS ring of W + 2 pixels
//     whose pad pixels stay zero (no border test in the loop), filled one output row ahead (MODE 0: LDS-DMA; MODE 1: f32 loads ->
//     bf16 -> ds_write); one barrier per output row;
//   * the 9 x C x C weights stay in REGISTERS for the whole launch (18 B fragments per wave at 32 / 64 channels; 16 channels: two taps
//     share one K = 32 MFMA: 5 fragments); A fragments are plain 16-byte LDS reads of [pixel][channel] rows;
//   * C <= 32: the four waves split the 16-pixel blocks of a row (each all output channels); C = 64: the waves split the output-channel
//     blocks (each all pixel blocks);
//   * the epilogue of the map kernels, fused: folded-BN scale / shift + ReLU + bf16 rows (forward), BN scale + ReLU gate (data gradient).
// Arithmetic: bf16 operands, f32 accumulation, taps in (ty, tx) order then channel chunks -- a fixed order: bit-reproducible.
#include "common.h"
#include "../../include/es_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) unsigned short g_ic_zero[8];

template <int C, int WP, int S, int MODE>
__global__ __launch_bounds__(256) void k_img_conv3(const void* __restrict__ Xv, int ldx, const unsigned short* __restrict__ Wt,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const unsigned short* __restrict__ gate, int ldg, void* __restrict__ Yv, int ldy,
                                                   int y_half, int act, int H, int W, int rows_per_wg, int bands) {
  static_assert(MODE == 0 || S == 1, "the gated data gradient is built for stride 1");
  constexpr int RB = C * 2;                                 // bytes per pixel row of the ring
  constexpr int XP = S * WP + 2, X_BYTES = XP * RB;
  constexpr int NS = S == 1 ? 4 : 8;                        // ring slots
  constexpr int GX = RB / 16;                               // 16-byte granules per pixel
  constexpr int NCB = C / 16;                               // output-channel blocks
  constexpr bool CO_SPLIT = NCB >= 4;                       // waves split output-channel blocks (else: pixel blocks)
  constexpr int CBW = CO_SPLIT ? NCB / 4 : NCB;             // channel blocks per wave
  constexpr bool PAIR = C == 16;                            // two taps per K = 32 MFMA
  constexpr int KS = PAIR ? 1 : C / 32;                     // MFMA k steps per tap
  constexpr int NT = PAIR ? 5 : 9;                          // tap groups
  constexpr int NPB = WP / 16;                              // 16-pixel blocks per output row
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * X_BYTES];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int im = blockIdx.x / bands, band = blockIdx.x - im * bands;
  const int Ho = H / S, Wo = W / S;
  const int oy0 = band * rows_per_wg, oy1 = min(Ho, oy0 + rows_per_wg);
  if (oy0 >= oy1) return;
  for (int i = t; i < NS * X_BYTES / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0u, 0u, 0u, 0u);

  // weights -> registers.  Forward: Wt = [tap][Cout][Cin] (Cin contiguous), tap t of the loop reads tap t.  Data gradient: Wt = the natural
  // [tap][Cin][Cout] copy read as [tap][N = Cin][K = Cout], and tap t of the loop (offset (ty - 1, tx - 1) on the GRADIENT grid) multiplies the
  // weights of the mirrored tap 8 - t.
  bf16x8_t Bf[NT][KS][CBW];
#pragma unroll
  for (int g = 0; g < NT; ++g)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int cw = 0; cw < CBW; ++cw) {
        const int cb = CO_SPLIT ? wv * CBW + cw : cw;
        int tap = PAIR ? 2 * g + (kq >> 1) : g;
        const int kc = PAIR ? (kq & 1) * 8 : ks * 32 + kq * 8;
        const bool ok = tap < 9;
        if (MODE == 1) tap = 8 - tap;
        const uint4 v = ok ? *(const uint4*)(Wt + ((size_t)(ok ? tap : 0) * C + cb * 16 + li) * C + kc) : make_uint4(0u, 0u, 0u, 0u);
        Bf[g][ks][cw] = __builtin_bit_cast(bf16x8_t, v);
      }
  __syncthreads();                                            // the ring is zero

  const unsigned short* Xh = (const unsigned short*)Xv;
  const float* Xf = (const float*)Xv;
  constexpr int NPX = (S * WP * GX + 255) / 256;            // 16-byte pieces per thread and image row
  // MODE 0: image row iy -> ring slot by LDS-DMA (pixels 0 .. W - 1 at LDS pixels 1 .. W; granules past W copy a zero granule)
  auto issue_x = [&](int iy) {
    if (iy < 0 || iy >= H) return;                          // (uniform)
    const unsigned short* src = Xh + ((size_t)im * H + iy) * W * ldx;
    unsigned char* dst = smem + (iy & (NS - 1)) * X_BYTES + RB;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const int e = (j * 4 + wv) * 64 + lane;
      if ((j * 4 + wv) * 64 < S * WP * GX) {                // (wave-uniform: whole 1 KB pieces)
        const int px = e / GX, g = e - px * GX;
        const unsigned short* p = px < W ? (src + (size_t)px * ldx + g * 8) : g_ic_zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(dst + (j * 4 + wv) * 1024), 16, 0, 0);
      }
    }
  };
  // MODE 1: gradient row iy (f32) -> registers -> bf16 -> ring slot
  constexpr int NLG = MODE == 1 ? (WP * C / 4 + 255) / 256 : 1;
  float4 greg[NLG];
  auto load_g = [&](int iy) {
    const float* src = Xf + ((size_t)im * H + iy) * W * ldx;
#pragma unroll
    for (int j = 0; j < NLG; ++j) {
      const int e = j * 256 + t, px = e / (C / 4), c4 = e - px * (C / 4);
      greg[j] = (iy >= 0 && iy < H && px < W) ? *(const float4*)(src + (size_t)px * ldx + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_g = [&](int iy) {
    if (iy < 0 || iy >= H) return;
    unsigned char* dst = smem + (iy & (NS - 1)) * X_BYTES + RB;
#pragma unroll
    for (int j = 0; j < NLG; ++j) {
      const int e = j * 256 + t, px = e / (C / 4), c4 = e - px * (C / 4);
      if (px < WP) {
        uint2 v;
        v.x = es_pack_bf16(greg[j].x, greg[j].y);
        v.y = es_pack_bf16(greg[j].z, greg[j].w);
        *(uint2*)(dst + px * RB + c4 * 8) = v;
      }
    }
  };

  // prologue: the input rows of the first output row and the rows the second one adds
  if (MODE == 0) {
    for (int r = -1; r <= 1 + (S - 1); ++r) issue_x(S * oy0 + r);
  } else {
    for (int r = -1; r <= 0; ++r) { load_g(oy0 + r); store_g(oy0 + r); }
    load_g(oy0 + 1);
  }
  for (int oy = oy0; oy < oy1; ++oy) {
    if (MODE == 1) store_g(oy + 1);                          // (its slot held row oy - 3: nobody reads it any more)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of the rows of output row oy have landed ...
    __syncthreads();                                          // ... everybody's; output row oy - 1 has been consumed
    if (MODE == 0) {
      if (S == 1) issue_x(oy + 2);
      else { issue_x(2 * oy + 3); issue_x(2 * oy + 4); }
    } else {
      load_g(oy + 2);
    }
#pragma unroll 1
    for (int pb = CO_SPLIT ? 0 : wv; pb < NPB; pb += CO_SPLIT ? 1 : 4) {
      const int p0 = pb * 16;
      if (p0 >= Wo) break;
      f32x4 acc[CBW];
#pragma unroll
      for (int cw = 0; cw < CBW; ++cw) acc[cw] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < NT; ++g) {
        // this lane's tap: 16 channels share one K = 32 step between two taps (lanes kq < 2: tap 2 g, kq >= 2: tap 2 g + 1)
        const int tap = PAIR ? 2 * g + (kq >> 1) : g;
        const int ty = tap / 3, tx = tap - ty * 3;
        const int iy = S * oy + ty - 1;
        const bool rowok = tap < 9 && iy >= 0 && iy < H;    // (a row outside the image contributes nothing: its slot may hold another row)
        const unsigned char* xr = smem + ((iy < 0 ? 0 : iy) & (NS - 1)) * X_BYTES + ((p0 + li) * S + tx) * RB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int kc = PAIR ? (kq & 1) * 8 : ks * 32 + kq * 8;
          uint4 av = *(const uint4*)(xr + kc * 2);
          if (!rowok) av = make_uint4(0u, 0u, 0u, 0u);
          const bf16x8_t a = __builtin_bit_cast(bf16x8_t, av);
#pragma unroll
          for (int cw = 0; cw < CBW; ++cw) acc[cw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Bf[g][ks][cw], acc[cw], 0, 0, 0);
        }
      }
      // epilogue: lane (li, kq) holds pixels p0 + kq * 4 + r of channel cb * 16 + li
#pragma unroll
      for (int cw = 0; cw < CBW; ++cw) {
        const int col = (CO_SPLIT ? wv * CBW + cw : cw) * 16 + li;
        const float sc = scale ? scale[col] : 1.f, sh = (MODE == 0 && shift) ? shift[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = p0 + kq * 4 + r;
          const size_t row = ((size_t)im * Ho + oy) * Wo + px;
          float v = acc[cw][r] * sc + sh;
          if (MODE == 0) {
            if (act) v = fmaxf(v, 0.f);
            if (y_half) {                                    // bf16 rows: lanes (li, li ^ 1) share one 4-byte store
              const float vn = __shfl_xor(v, 1, 64);
              if (!(li & 1) && px < Wo) *(uint32_t*)((unsigned short*)Yv + row * ldy + col) = es_pack_bf16(v, vn);
            } else if (px < Wo) {
              ((float*)Yv)[row * ldy + col] = v;
            }
          } else if (px < Wo) {
            const float gt = __uint_as_float((uint32_t)gate[row * ldg + col] << 16);
            ((float*)Yv)[row * ldy + col] = gt > 0.f ? v : 0.f;
          }
        }
      }
    }
  }
}

static int ES_OPT_IMG_CONV = 1;
static int ES_OPT_IMG_CONV_WGS = 1024;
extern "C" int es_img_conv_set_option(int key, int value) {
  if (key == 50) { ES_OPT_IMG_CONV = value; return 0; }
  if (key == 51) { ES_OPT_IMG_CONV_WGS = value; return 0; }
  return -1;
}

static bool img_conv_plan(int n_img, int H, int W, int C, int stride, int mode, int& wp, int& rows, int& bands) {
  if (!ES_OPT_IMG_CONV || n_img <= 0 || H <= 0 || W <= 0) return false;
  if (mode == 1 && stride != 1) return false;
  if (stride != 1 && !(stride == 2 && H % 2 == 0 && W % 2 == 0)) return false;
  const int Ho = H / stride, Wo = W / stride;
  if (C == 16 && Wo <= 128 && stride == 1) wp = Wo <= 64 ? 64 : 128;
  else if (C == 32 && Wo <= 64) wp = 64;
  else if (C == 64 && Wo <= 32) wp = 32;
  else return false;
  if (C == 64 && stride == 2) return false;                 // (A/B, profiles/r6g_imgconv_ab.txt: 32 us against the map kernel's 26.5)
  // workgroups aimed for (A/B on 80 images): 1 024 at 16 / 32 channels, 512 for the 64-channel and the stride-2 launches
  const int target = (C == 64 || stride == 2) ? ES_OPT_IMG_CONV_WGS / 2 : ES_OPT_IMG_CONV_WGS;
  bands = target / n_img;
  if (bands < 1) bands = 1;
  if (bands > Ho) bands = Ho;
  rows = es_cdiv(Ho, bands);
  bands = es_cdiv(Ho, rows);
  return true;
}

extern "C" int es_img_conv3_supported(int n_img, int H, int W, int C, int stride, int mode) {
  int wp, rows, bands;
  return img_conv_plan(n_img, H, W, C, stride, mode, wp, rows, bands) ? 1 : 0;
}

extern "C" int es_img_conv3_bf16(const void* X, int ldx, const void* W_bf16, int n_img, int H, int W, int C, int stride, int mode,
                                 const float* scale, const float* shift, const void* gate, int ldg, int act, void* Y, int y_half,
                                 int ldy, void* stream) {
  int wp, rows, bands;
  if (!img_conv_plan(n_img, H, W, C, stride, mode, wp, rows, bands)) return -4;
  if ((ldx % (mode ? 4 : 8)) || ((((uintptr_t)X) | ((uintptr_t)W_bf16)) & 15) || (y_half ? ((ldy % 2) || (((uintptr_t)Y) & 3)) : 0)) return -4;
  if (mode == 1 && (gate == nullptr || y_half)) return -4;
  if ((long long)n_img * H * W * (ldx > ldy ? ldx : ldy) >= (1ll << 31)) return -4;
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* Wt = (const unsigned short*)W_bf16;
  const unsigned short* G = (const unsigned short*)gate;
#define IC_LAUNCH(C_, WP_, S_, M_)                                                                                           \
  hipLaunchKernelGGL((k_img_conv3<C_, WP_, S_, M_>), dim3(n_img * bands), dim3(256), 0, st, X, ldx, Wt, scale, shift, G, ldg, Y, ldy, \
                     y_half, act, H, W, rows, bands)
  if (mode == 0 && stride == 1) {
    if (C == 16 && wp == 128) IC_LAUNCH(16, 128, 1, 0);
    else if (C == 16) IC_LAUNCH(16, 64, 1, 0);
    else if (C == 32) IC_LAUNCH(32, 64, 1, 0);
    else IC_LAUNCH(64, 32, 1, 0);
  } else if (mode == 0) {
    if (C == 32) IC_LAUNCH(32, 64, 2, 0);
    else if (C == 64) IC_LAUNCH(64, 32, 2, 0);
    else return -4;
  } else {
    if (C == 16 && wp == 128) IC_LAUNCH(16, 128, 1, 1);
    else if (C == 16) IC_LAUNCH(16, 64, 1, 1);
    else if (C == 32) IC_LAUNCH(32, 64, 1, 1);
    else IC_LAUNCH(64, 32, 1, 1);
  }
#undef IC_LAUNCH
  ES_CHECK_LAUNCH();
  return 0;
}
