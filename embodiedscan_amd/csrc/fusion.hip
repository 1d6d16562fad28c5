// Point <-> image projection fusion (row A8), one fused kernel per level:
// voxel centre -> undo 3-D augmentation -> project into each of the V views -> nearest
// feature fetch (channels-last, coalesced along C) -> unmasked sum / valid-view count.
// Replaces batch_point_sample (+ apply_3d_transformation, batch_points_cam2img,
// F.grid_sample) at embodiedscan/models/layers/fusion_layers/point_fusion.py:20-107,208-311
// and embodiedscan/structures/bbox_3d/utils.py:289-332, called from
// embodiedscan/models/detectors/sparse_featfusion_single_stage.py:142-207.
#include "common.h"
#include "../../include/es_hip.h"

// per-sample meta block (floats), see es_hip.h: ES_FUSE_*
__device__ inline void undo_aug(const float* m, float& x, float& y, float& z) {
  int nops = (int)m[ES_FUSE_NOPS];
  for (int o = 0; o < nops; ++o) {
    int op = (int)m[ES_FUSE_OPS + o];
    if (op == 1) {            // T (reverse): p += (-trans)
      x = __fadd_rn(x, m[ES_FUSE_NTRANS + 0]);
      y = __fadd_rn(y, m[ES_FUSE_NTRANS + 1]);
      z = __fadd_rn(z, m[ES_FUSE_NTRANS + 2]);
    } else if (op == 2) {     // S (reverse): p *= 1/scale
      float s = m[ES_FUSE_ISCALE];
      x = __fmul_rn(x, s); y = __fmul_rn(y, s); z = __fmul_rn(z, s);
    } else if (op == 3) {     // R (reverse): p = p @ rot^-1
      const float* r = m + ES_FUSE_ROTINV;
      float nx = fmaf(z, r[6], fmaf(y, r[3], __fmul_rn(x, r[0])));
      float ny = fmaf(z, r[7], fmaf(y, r[4], __fmul_rn(x, r[1])));
      float nz = fmaf(z, r[8], fmaf(y, r[5], __fmul_rn(x, r[2])));
      x = nx; y = ny; z = nz;
    } else if (op == 4) {
      x = -x;
    } else if (op == 5) {
      y = -y;
    }
  }
}

// returns pixel linear index (or -1 when grid_sample pads with zero) and the valid flag
__device__ inline int project_view(const float* m, const float* P, float x, float y, float z, int Hf, int Wf,
                                   bool& valid) {
  float qx = fmaf(1.f, P[3], fmaf(z, P[2], fmaf(y, P[1], __fmul_rn(x, P[0]))));
  float qy = fmaf(1.f, P[7], fmaf(z, P[6], fmaf(y, P[5], __fmul_rn(x, P[4]))));
  float qz = fmaf(1.f, P[11], fmaf(z, P[10], fmaf(y, P[9], __fmul_rn(x, P[8]))));
  float zc = fmaxf(qz, 1e-3f);                                    // clamp(min=1e-3)
  float u = __fdiv_rn(qx, zc), v = __fdiv_rn(qy, zc);
  u = __fsub_rn(__fmul_rn(u, m[ES_FUSE_SFX]), m[ES_FUSE_CROPX]);
  v = __fsub_rn(__fmul_rn(v, m[ES_FUSE_SFY]), m[ES_FUSE_CROPY]);
  if (m[ES_FUSE_FLIP] != 0.f) u = __fsub_rn(m[ES_FUSE_ORIW], u);
  float w = m[ES_FUSE_PADW], h = m[ES_FUSE_PADH];
  valid = (u < w) && (u > 0.f) && (v < h) && (v > 0.f) && (qz > 0.f);
  float gx = __fsub_rn(__fmul_rn(__fdiv_rn(u, w), 2.f), 1.f);
  float gy = __fsub_rn(__fmul_rn(__fdiv_rn(v, h), 2.f), 1.f);
  // ATen CPU grid_sampler (vectorised): unnormalize = (g + 1) * ((size - 1) / 2); nearest = nearbyint
  float ix = nearbyintf(__fmul_rn(__fadd_rn(gx, 1.f), __fdiv_rn((float)(Wf - 1), 2.f)));
  float iy = nearbyintf(__fmul_rn(__fadd_rn(gy, 1.f), __fdiv_rn((float)(Hf - 1), 2.f)));
  if (!(ix > -1.f && ix < (float)Wf && iy > -1.f && iy < (float)Hf)) return -1;
  return (int)iy * Wf + (int)ix;
}

// One wave per point, 16 points per workgroup.  Round 4 (north_star: "per-view point projection ... staged through LDS with
// coalesced HBM reads"): the sample's meta block (reverse-augmentation ops + the V projection matrices, 32 + 16 V floats) is
// staged ONCE per workgroup in LDS (rows are batch-major, so the 16 points of a workgroup belong to one sample except at a
// sample boundary, where the waves read the block from global memory); lane v projects the point into view v -- ONE projection
// per lane instead of all V projections in all 64 lanes (rounds 1-3) -- the pixel indices go out as one coalesced row of V
// ints, and are then broadcast lane by lane for the channel-strided feature fetch (coalesced along C).  Same arithmetic per
// (point, view) as before: bit-identical outputs.
// PTS: the location comes from a float (n,3) array (the prior points of the occupancy detector, dense_fusion_occ.py:156-202)
// instead of integer voxel coordinates * voxel_size.
#define PS_PTS 16
#define PS_MAXMETA 1088                                    // 32 + 16 * 66 floats: up to 66 views staged
template <bool PTS, bool FH = false>
__global__ __launch_bounds__(256) void k_point_sample_fwd(const int* __restrict__ coords, const float* __restrict__ pts,
                                                          int n, float voxel_size,
                                                          const float* __restrict__ meta, int meta_stride, int V,
                                                          const float* __restrict__ feats, int Hf, int Wf, int C,
                                                          float* __restrict__ out, int ldo, int* __restrict__ pix,
                                                          int* __restrict__ cnt) {
  __shared__ float metaS[PS_MAXMETA];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i0 = blockIdx.x * PS_PTS, i1 = min(n, i0 + PS_PTS);
  const int b0 = coords[(size_t)i0 * 4], b1 = coords[(size_t)(i1 - 1) * 4];
  const int nmeta = ES_FUSE_PROJ + 16 * V;
  const bool staged = (b0 == b1) && (nmeta <= PS_MAXMETA);   // workgroup-uniform
  if (staged)
    for (int e = threadIdx.x; e < nmeta; e += 256) metaS[e] = meta[(size_t)b0 * meta_stride + e];
  __syncthreads();
  const int MAXC = 8;                                   // supports C <= 512
  for (int i = i0 + wv; i < i1; i += 4) {               // wave-uniform
    int4 c = ((const int4*)coords)[i];
    const float* m = staged ? metaS : meta + (size_t)c.x * meta_stride;
    float x, y, z;
    if (PTS) {
      x = pts[(size_t)i * 3]; y = pts[(size_t)i * 3 + 1]; z = pts[(size_t)i * 3 + 2];
    } else {
      x = __fmul_rn((float)c.y, voxel_size); y = __fmul_rn((float)c.z, voxel_size); z = __fmul_rn((float)c.w, voxel_size);
    }
    undo_aug(m, x, y, z);
    int nvalid = 0;
    float acc[MAXC];
#pragma unroll
    for (int q = 0; q < MAXC; ++q) acc[q] = 0.f;
    for (int v0 = 0; v0 < V; v0 += 64) {                // (V <= 64 in every shipped config: one trip)
      const int v = v0 + lane;
      bool valid = false;
      int p = -1;
      if (v < V) {
        p = project_view(m, m + ES_FUSE_PROJ + v * 16, x, y, z, Hf, Wf, valid);
        pix[(size_t)i * V + v] = p;
      }
      nvalid += (int)__popcll(__ballot(valid));
      const int nv = min(64, V - v0);
      for (int u = 0; u < nv; ++u) {                    // views in ascending order: the summation order of rounds 1-3
        const int pu = __shfl(p, u, 64);
        if (pu >= 0) {
          const size_t off = (((size_t)c.x * V + (v0 + u)) * Hf * Wf + pu) * C;
#pragma unroll
          for (int q = 0; q < MAXC; ++q) {
            int ch = lane + q * 64;                      // sum over ALL views, not masked
            if (ch < C) acc[q] += FH ? __uint_as_float((uint32_t)((const unsigned short*)feats)[off + ch] << 16) : feats[off + ch];
          }
        }
      }
    }
    if (lane == 0) cnt[i] = nvalid;
    float d = (float)max(nvalid, 1);
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
      int ch = lane + q * 64;
      if (ch < C) out[(size_t)i * ldo + ch] = nvalid > 0 ? __fdiv_rn(acc[q], d) : 0.f;
    }
  }
}
extern "C" int es_point_sample_fwd(const int* coords, int n, float voxel_size, const float* meta, int meta_stride,
                                   int V, const float* feats, int Hf, int Wf, int C, float* out, int ldo, int* pix,
                                   int* cnt, void* stream) {
  if (n <= 0) return 0;
  if (C > 512) return -4;
  hipLaunchKernelGGL(k_point_sample_fwd<false>, dim3(es_cdiv(n, PS_PTS)), dim3(256), 0, (hipStream_t)stream, coords,
                     (const float*)nullptr, n, voxel_size, meta, meta_stride, V, feats, Hf, Wf, C, out, ldo, pix, cnt);
  ES_CHECK_LAUNCH();
  return 0;
}
// the same with the feature maps stored in bf16 (activation storage of the image backbone, round 3)
extern "C" int es_point_sample_fwd_h(const int* coords, int n, float voxel_size, const float* meta, int meta_stride,
                                     int V, const void* feats_bf16, int Hf, int Wf, int C, float* out, int ldo, int* pix,
                                     int* cnt, void* stream) {
  if (n <= 0) return 0;
  if (C > 512) return -4;
  hipLaunchKernelGGL((k_point_sample_fwd<false, true>), dim3(es_cdiv(n, PS_PTS)), dim3(256), 0, (hipStream_t)stream, coords,
                     (const float*)nullptr, n, voxel_size, meta, meta_stride, V, (const float*)feats_bf16, Hf, Wf, C, out, ldo,
                     pix, cnt);
  ES_CHECK_LAUNCH();
  return 0;
}
extern "C" int es_point_sample_fwd_pts(const int* coords, const float* points, int n, const float* meta, int meta_stride,
                                       int V, const float* feats, int Hf, int Wf, int C, float* out, int ldo, int* pix,
                                       int* cnt, void* stream) {
  if (n <= 0) return 0;
  if (C > 512) return -4;
  hipLaunchKernelGGL(k_point_sample_fwd<true>, dim3(es_cdiv(n, PS_PTS)), dim3(256), 0, (hipStream_t)stream, coords, points, n,
                     0.f, meta, meta_stride, V, feats, Hf, Wf, C, out, ldo, pix, cnt);
  ES_CHECK_LAUNCH();
  return 0;
}

// Backward of the projection fusion, deterministic (round 3; rounds 1-2 scattered with f32 atomics): the feature-map
// gradient is GATHERED.  (1) every hit (voxel i, view v -> pixel p) is linked into the list of its feature-map pixel
// (integer atomicExch: the list CONTENT is deterministic, its order is not); (2) one wave per feature-map pixel walks its
// list and adds the rows dout[i] / cnt[i] in ASCENDING voxel order (64 smallest remaining hits at a time, extracted by
// wave-min), so the sum order is fixed whatever the link order was.  Every pixel is written (zeros when nothing projects
// to it): the caller needs no memset and no float atomics are left on the path.
__global__ void k_ps_link(const int* __restrict__ coords, int n, int V, const int* __restrict__ pix,
                          const int* __restrict__ cnt, int HW, int* __restrict__ head, int* __restrict__ next) {
  size_t tot = (size_t)n * V;
  for (size_t h = (size_t)blockIdx.x * blockDim.x + threadIdx.x; h < tot; h += (size_t)gridDim.x * blockDim.x) {
    int p = pix[h];
    if (p < 0) continue;
    int i = (int)(h / V), v = (int)(h - (size_t)i * V);
    if (cnt[i] <= 0) continue;       // no valid view: the forward output of this voxel is 0 whatever was sampled
    int b = coords[(size_t)i * 4];
    next[h] = atomicExch(&head[((size_t)b * V + v) * HW + p], (int)h);
  }
}
#define PS_MAXC 8
__global__ __launch_bounds__(256) void k_ps_gather(const int* __restrict__ head, const int* __restrict__ next, int n_pix,
                                                   int V, const float* __restrict__ dout, int ldo,
                                                   const int* __restrict__ cnt, int C, float* __restrict__ dfeats,
                                                   int accumulate) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= n_pix) return;
  float acc[PS_MAXC];
#pragma unroll
  for (int q = 0; q < PS_MAXC; ++q) acc[q] = 0.f;
  const int h0 = head[g];
  if (h0 >= 0) {
    int last = -1;
    while (true) {
      int mine = 0x7fffffff, got = 0;                     // got is wave-uniform
      for (int h = h0; h >= 0; h = next[h]) {             // wave-uniform walk
        if (h <= last) continue;
        if (got < 64) {
          if (lane == got) mine = h;
          ++got;
        } else {                                          // keep the 64 smallest: replace the current maximum
          int mx = es_wave_max_i(mine);
          if (h < mx && mine == mx) mine = h;
        }
      }
      if (got == 0) break;
      for (int r = 0; r < got; ++r) {
        const int m = es_wave_min_i(mine);
        const int i = m / V;
        const float inv = __fdiv_rn(1.f, (float)cnt[i]);
        const float* row = dout + (size_t)i * ldo;
#pragma unroll
        for (int q = 0; q < PS_MAXC; ++q) {
          int ch = lane + q * 64;
          if (ch < C) acc[q] += row[ch] * inv;
        }
        if (mine == m) mine = 0x7fffffff;
        last = m;
      }
      if (got < 64) break;                                // the walk has seen every remaining hit
    }
  }
  float* f = dfeats + (size_t)g * C;
#pragma unroll
  for (int q = 0; q < PS_MAXC; ++q) {
    int ch = lane + q * 64;
    if (ch < C) f[ch] = accumulate ? (f[ch] + acc[q]) : acc[q];
  }
}
extern "C" int es_point_sample_bwd(const int* coords, int n, int V, const float* dout, int ldo, const int* pix,
                                   const int* cnt, int Hf, int Wf, int C, float* dfeats, int n_img, int* head, int* next,
                                   int accumulate, void* stream) {
  if (C > 64 * PS_MAXC) return -4;
  const long long n_pix = (long long)n_img * Hf * Wf;
  if (n_pix <= 0 || n_pix >= (1ll << 31) || (long long)n * V >= (1ll << 31)) return n_pix <= 0 ? 0 : -6;
  hipStream_t st = (hipStream_t)stream;
  ES_TRY(hipMemsetAsync(head, 0xff, (size_t)n_pix * sizeof(int), st));
  if (n > 0) {
    int g = es_cdiv((long long)n * V, 256);
    hipLaunchKernelGGL(k_ps_link, dim3(g > 8192 ? 8192 : g), dim3(256), 0, st, coords, n, V, pix, cnt, Hf * Wf, head, next);
    ES_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_ps_gather, dim3(es_cdiv(n_pix, 4)), dim3(256), 0, st, head, next, (int)n_pix, V, dout, ldo, cnt, C,
                     dfeats, accumulate);
  ES_CHECK_LAUNCH();
  return 0;
}
