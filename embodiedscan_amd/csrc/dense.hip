// Dense-volume side of the occupancy path (BASELINE config 5):
//   * es_volume_map / es_volume_up_index : static neighbour maps of a dense (B, X, Y, Z) voxel grid, so that
//     nn.Conv3d(k=3 / k=1, stride 1 / 2) and nn.ConvTranspose3d(k=2, s=2) of IndoorImVoxelNeck
//     (embodiedscan/models/necks/imvoxel_neck.py:34-143) run as implicit GEMMs on the same MFMA convolution engine as
//     the sparse path (rows = voxels, channels-last);
//   * es_voxel_keys_range : the occupancy detector's voxelisation ((p - range_min) / voxel_size, truncation, clamp;
//     embodiedscan/models/detectors/dense_fusion_occ.py:227-245);
//   * es_dense_index : ME SparseTensor.dense() row addresses (dense_fusion_occ.py:252-255);
//   * es_upsample_nearest_add_* : the top-down pathway of mmdet.FPN (F.interpolate(size=..., mode='nearest') + add);
//   * es_row_argmax : ImVoxelOccHead.predict (imvoxel_occ_head.py:104-108).
#include "common.h"
#include "../../include/es_hip.h"

// nbr[(o*K + k)], o = ((b*Xo + xo)*Yo + yo)*Zo + zo, k = (kx*ks + ky)*ks + kz  (the order of a torch (O,I,kD,kH,kW)
// kernel with D = x, H = y, W = z)
__global__ void k_volume_map(int B, int X, int Y, int Z, int Xo, int Yo, int Zo, int ks, int stride, int pad,
                             int* __restrict__ nbr) {
  const int K = ks * ks * ks;
  long long tot = (long long)B * Xo * Yo * Zo * K;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
    int k = (int)(e % K);
    long long o = e / K;
    int zo = (int)(o % Zo), yo = (int)((o / Zo) % Yo), xo = (int)((o / ((long long)Zo * Yo)) % Xo);
    int b = (int)(o / ((long long)Zo * Yo * Xo));
    int kz = k % ks, ky = (k / ks) % ks, kx = k / (ks * ks);
    int x = xo * stride - pad + kx, y = yo * stride - pad + ky, z = zo * stride - pad + kz;
    bool in = x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z;
    nbr[e] = in ? (int)((((long long)b * X + x) * Y + y) * Z + z) : -1;
  }
}
extern "C" int es_volume_map(int n_batch, int X, int Y, int Z, int Xo, int Yo, int Zo, int ksize, int stride, int pad,
                             int* nbr, void* stream) {
  if (n_batch <= 0) return 0;
  hipLaunchKernelGGL(k_volume_map, dim3(2048), dim3(256), 0, (hipStream_t)stream, n_batch, X, Y, Z, Xo, Yo, Zo, ksize,
                     stride, pad, nbr);
  ES_CHECK_LAUNCH();
  return 0;
}

// ConvTranspose3d(k=2,s=2) runs as 8 row GEMMs whose output row is 8*i + tap (the generative layout of the sparse
// engine); idx[o] = that row for the dense output voxel o of the (B, 2X, 2Y, 2Z) grid.
__global__ void k_volume_up_index(int B, int X, int Y, int Z, int* __restrict__ idx) {
  long long tot = (long long)B * X * Y * Z * 8;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < tot; o += (long long)gridDim.x * blockDim.x) {
    int Z2 = 2 * Z, Y2 = 2 * Y, X2 = 2 * X;
    int zo = (int)(o % Z2), yo = (int)((o / Z2) % Y2), xo = (int)((o / ((long long)Z2 * Y2)) % X2);
    int b = (int)(o / ((long long)Z2 * Y2 * X2));
    long long i = (((long long)b * X + xo / 2) * Y + yo / 2) * Z + zo / 2;
    int tap = ((xo & 1) * 2 + (yo & 1)) * 2 + (zo & 1);
    idx[o] = (int)(i * 8 + tap);
  }
}
extern "C" int es_volume_up_index(int n_batch, int X, int Y, int Z, int* idx, void* stream) {
  if (n_batch <= 0) return 0;
  hipLaunchKernelGGL(k_volume_up_index, dim3(1024), dim3(256), 0, (hipStream_t)stream, n_batch, X, Y, Z, idx);
  ES_CHECK_LAUNCH();
  return 0;
}

// c = trunc((p - range_min) / voxel_size) per axis (f32 subtraction, TRUE f32 division, C truncation), clamped to
// [0, cmax]: dense_fusion_occ.py:227-245 (use_xyz_feat=True branch).  rng: 9 floats on the HOST
// {min x,y,z, voxel size x,y,z, clamp max x,y,z}.
__global__ void k_voxel_keys_range(const float* __restrict__ pts, int n, int ld, int batch, float mx, float my, float mz,
                                   float vx, float vy, float vz, int cx, int cy, int cz, int64_t* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x = (int)__fdiv_rn(__fsub_rn(pts[(size_t)i * ld + 0], mx), vx);
  int y = (int)__fdiv_rn(__fsub_rn(pts[(size_t)i * ld + 1], my), vy);
  int z = (int)__fdiv_rn(__fsub_rn(pts[(size_t)i * ld + 2], mz), vz);
  x = min(max(x, 0), cx); y = min(max(y, 0), cy); z = min(max(z, 0), cz);
  keys[i] = es_pack(batch, x, y, z);
}
extern "C" int es_voxel_keys_range(const float* points, int n, int ld, int batch, const float* rng_host, int64_t* keys,
                                   void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_voxel_keys_range, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, points, n, ld, batch,
                     rng_host[0], rng_host[1], rng_host[2], rng_host[3], rng_host[4], rng_host[5], (int)rng_host[6],
                     (int)rng_host[7], (int)rng_host[8], keys);
  ES_CHECK_LAUNCH();
  return 0;
}

// dense row of each sparse voxel: ((b*X + x/ts)*Y + y/ts)*Z + z/ts, or -1 outside the volume
__global__ void k_dense_index(const int* __restrict__ coords, int n, int ts, int X, int Y, int Z, int* __restrict__ idx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = ((const int4*)coords)[i];
  int x = c.y / ts, y = c.z / ts, z = c.w / ts;
  bool in = c.y >= 0 && c.z >= 0 && c.w >= 0 && x < X && y < Y && z < Z;
  idx[i] = in ? (int)((((long long)c.x * X + x) * Y + y) * Z + z) : -1;
}
extern "C" int es_dense_index(const int* coords, int n, int ts, int X, int Y, int Z, int* idx, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_dense_index, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, coords, n, ts, X, Y, Z, idx);
  ES_CHECK_LAUNCH();
  return 0;
}

// mmdet.FPN top-down step on channels-last maps: fine[(n,h,w),c] += coarse[(n, floor(h*Hc/Hf), floor(w*Wc/Wf)), c]
// (F.interpolate(size=(Hf,Wf), mode='nearest'): src = floor(dst * in/out) with the scale computed in f32)
__device__ inline int nearest_src(int d, int in, int out) {
  float s = (float)in / (float)out;
  int v = (int)floorf((float)d * s);
  return min(v, in - 1);
}
__global__ void k_upsample_add_fwd(float* __restrict__ fine, const float* __restrict__ coarse, int NI, int Hf, int Wf,
                                   int Hc, int Wc, int C4) {
  long long tot = (long long)NI * Hf * Wf * C4;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
    int c = (int)(e % C4);
    long long p = e / C4;
    int w = (int)(p % Wf), h = (int)((p / Wf) % Hf), im = (int)(p / ((long long)Wf * Hf));
    long long q = ((long long)im * Hc + nearest_src(h, Hc, Hf)) * Wc + nearest_src(w, Wc, Wf);
    float4 a = ((float4*)fine)[e], b = ((const float4*)coarse)[q * C4 + c];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    ((float4*)fine)[e] = a;
  }
}
extern "C" int es_upsample_nearest_add_fwd(float* fine, const float* coarse, int n_img, int Hf, int Wf, int Hc, int Wc,
                                           int C, void* stream) {
  if (n_img <= 0) return 0;
  if (C % 4) return -4;
  hipLaunchKernelGGL(k_upsample_add_fwd, dim3(4096), dim3(256), 0, (hipStream_t)stream, fine, coarse, n_img, Hf, Wf, Hc, Wc,
                     C / 4);
  ES_CHECK_LAUNCH();
  return 0;
}
// dcoarse[(n,hc,wc),c] (+)= sum of dfine over the fine pixels that read it (gather form: no atomics)
__global__ void k_upsample_add_bwd(const float* __restrict__ dfine, float* __restrict__ dcoarse, int NI, int Hf, int Wf,
                                   int Hc, int Wc, int C4, int accumulate) {
  long long tot = (long long)NI * Hc * Wc * C4;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
    int c = (int)(e % C4);
    long long p = e / C4;
    int wc = (int)(p % Wc), hc = (int)((p / Wc) % Hc), im = (int)(p / ((long long)Wc * Hc));
    // candidate fine rows / columns: a superset window, filtered by the forward rule
    int h0 = (int)((long long)hc * Hf / Hc) - 1, h1 = (int)((long long)(hc + 1) * Hf / Hc) + 1;
    int w0 = (int)((long long)wc * Wf / Wc) - 1, w1 = (int)((long long)(wc + 1) * Wf / Wc) + 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = max(h0, 0); h <= min(h1, Hf - 1); ++h) {
      if (nearest_src(h, Hc, Hf) != hc) continue;
      for (int w = max(w0, 0); w <= min(w1, Wf - 1); ++w) {
        if (nearest_src(w, Wc, Wf) != wc) continue;
        float4 g = ((const float4*)dfine)[(((long long)im * Hf + h) * Wf + w) * C4 + c];
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
      }
    }
    if (accumulate) {
      float4 o = ((float4*)dcoarse)[e];
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    ((float4*)dcoarse)[e] = acc;
  }
}
extern "C" int es_upsample_nearest_add_bwd(const float* dfine, float* dcoarse, int n_img, int Hf, int Wf, int Hc, int Wc,
                                           int C, int accumulate, void* stream) {
  if (n_img <= 0) return 0;
  if (C % 4) return -4;
  hipLaunchKernelGGL(k_upsample_add_bwd, dim3(2048), dim3(256), 0, (hipStream_t)stream, dfine, dcoarse, n_img, Hf, Wf, Hc,
                     Wc, C / 4, accumulate);
  ES_CHECK_LAUNCH();
  return 0;
}

// first index of the row maximum (torch.max(softmax(x), dim=1) ties: lowest index; softmax is monotone)
__global__ __launch_bounds__(256) void k_row_argmax(const float* __restrict__ x, int ldx, int n, int C, int* __restrict__ out) {
  int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    float v = x[(size_t)i * ldx + c];
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[i] = bi;
}
extern "C" int es_row_argmax(const float* x, int ldx, int n, int C, int* out, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_row_argmax, dim3(es_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, n, C, out);
  ES_CHECK_LAUNCH();
  return 0;
}
