// Spatial (Morton / Z-order) ordering of the voxel rows.  Row order is a free choice of the coordinate manager
// (MinkowskiEngine's is hash-map order); sorting the unique voxels along a Z-curve makes the 3^3 neighbourhoods of
// consecutive rows overlap in memory, which is what the gather side of the convolution engine lives on
// (measured before: L2 hit rate 45 %, waves parked on memory 72 % of their cycles).
// The sort itself is a hand-written LSD radix sort (round 3; rounds 1-2 called rocPRIM through the hipCUB header): 8-bit
// digits, 8 passes over the 62 key bits, stable.  One pass = (1) per-tile digit histograms in LDS -> a digit-major table,
// (2) one exclusive scan of that table (its entry [digit][tile] is then the global position of the tile's first key with
// that digit), (3) the scatter: a tile of 2048 (key, row) pairs is read coalesced in 8 rounds of 256, every round ranks its
// keys among equal digits with 8 wave ballots + per-wave LDS counters (round-major, thread-minor = input order: stable),
// and writes to the ping-pong buffer.  Digits on which all keys agree (the batch bits, constant coordinate bits) are found
// by one histogram kernel up front and their passes return immediately.
#include "common.h"
#include "../../include/es_hip.h"

__device__ __host__ inline uint64_t spread3(uint64_t v) {     // 18 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__global__ void k_morton(const int64_t* __restrict__ keys, int n, uint64_t* __restrict__ mk, int* __restrict__ idx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b, x, y, z;
  es_unpack(keys[i], b, x, y, z);
  uint64_t m = spread3((uint64_t)(z + ES_OFF)) | (spread3((uint64_t)(y + ES_OFF)) << 1) |
               (spread3((uint64_t)(x + ES_OFF)) << 2);
  mk[i] = ((uint64_t)b << 54) | m;
  idx[i] = i;
}
__global__ void k_apply_perm(const int64_t* __restrict__ keys, const int* __restrict__ src,
                             const int* __restrict__ perm, int n, int64_t* __restrict__ out_keys,
                             int* __restrict__ out_src) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int p = perm[i];
  out_keys[i] = keys[p];
  if (src) out_src[i] = src[p];
}

#define RS_TILE 2048          // keys per workgroup and pass (8 rounds of 256 threads)
#define RS_PASSES 8

// all 8 digit histograms in one read of the keys; hist[p][d]
__global__ __launch_bounds__(256) void k_rs_hist_all(const uint64_t* __restrict__ keys, int n, unsigned int* __restrict__ hist) {
  __shared__ unsigned int h[RS_PASSES * 256];
  for (int e = threadIdx.x; e < RS_PASSES * 256; e += 256) h[e] = 0;
  __syncthreads();
  // (digits on which a whole wave agrees -- batch bits, constant coordinate bits: most of the upper passes -- cost one LDS atomic
  // per wave instead of 64 serialised ones on the same bin)
  for (int b0 = blockIdx.x * 256; b0 < n; b0 += gridDim.x * 256) {
    const int i = b0 + threadIdx.x;
    const bool ok = i < n;
    const uint64_t k = ok ? keys[i] : 0ull;
    const unsigned long long act = __ballot(ok);
    const int lane = threadIdx.x & 63;
    const int first = __ffsll((long long)act) - 1;            // lowest valid lane of this wave (-1: none)
#pragma unroll
    for (int p = 0; p < RS_PASSES; ++p) {
      const int d = (int)((k >> (8 * p)) & 255);
      const int d0 = __shfl(d, first < 0 ? 0 : first, 64);
      const unsigned long long same = __ballot(ok && d == d0);
      if (same == act) {
        if (lane == first) atomicAdd(&h[p * 256 + d0], (unsigned int)__popcll(act));
      } else if (ok) {
        atomicAdd(&h[p * 256 + d], 1u);
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < RS_PASSES * 256; e += 256)
    if (h[e]) atomicAdd(&hist[e], h[e]);
}
// plan[p] = 1 when pass p permutes nothing (one digit holds every key); plan[8 + p] = which of the two buffers holds the data
// when pass p starts (0 / 1); plan[16] = the buffer that holds the sorted result
__global__ void k_rs_plan(const unsigned int* __restrict__ hist, int n, int* __restrict__ plan) {
  __shared__ int triv[RS_PASSES];
  if (threadIdx.x < RS_PASSES) triv[threadIdx.x] = 0;
  __syncthreads();
  for (int e = threadIdx.x; e < RS_PASSES * 256; e += blockDim.x)
    if (hist[e] == (unsigned int)n) triv[e >> 8] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    int cur = 0;
    for (int p = 0; p < RS_PASSES; ++p) {
      plan[p] = triv[p];
      plan[8 + p] = cur;
      if (!triv[p]) cur ^= 1;
    }
    plan[16] = cur;
  }
}
// table[d * n_tiles + tile] = number of keys of the tile with digit d in pass p
__global__ __launch_bounds__(256) void k_rs_count(const uint64_t* __restrict__ buf0, const uint64_t* __restrict__ buf1, int n,
                                                  int pass, const int* __restrict__ plan, int n_tiles,
                                                  unsigned int* __restrict__ table) {
  if (plan[pass]) return;
  __shared__ unsigned int h[256];
  const uint64_t* keys = plan[8 + pass] ? buf1 : buf0;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int t0 = blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_TILE / 256; ++r) {
    const int i = t0 + r * 256 + threadIdx.x;
    const bool ok = i < n;
    const int d = ok ? (int)((keys[i] >> (8 * pass)) & 255) : 0;
    unsigned long long m = __ballot(ok);                    // wave-aggregated: lanes with the same digit add once
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      unsigned long long bal = __ballot((d >> b) & 1);
      m &= ((d >> b) & 1) ? bal : ~bal;
    }
    if (ok && (m & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0ull) atomicAdd(&h[d], (unsigned int)__popcll(m));
  }
  __syncthreads();
  table[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}
// in-place exclusive scan of the digit-major table (256 * n_tiles entries) by ONE workgroup of 1024 threads
__global__ __launch_bounds__(1024) void k_rs_scan(unsigned int* __restrict__ table, int total, int pass, const int* __restrict__ plan) {
  if (plan[pass]) return;
  __shared__ unsigned int wsum[16];
  __shared__ unsigned int carry;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < total; base += 1024 * 4) {
    int i0 = base + t * 4;
    unsigned int v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = (i0 + q < total) ? table[i0 + q] : 0u; s += v[q]; }
    unsigned int inc = s;                                 // inclusive scan of s over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned int u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned int woff = 0;
    for (int w = 0; w < wv; ++w) woff += wsum[w];
    unsigned int excl = carry + woff + inc - s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (i0 + q < total) table[i0 + q] = excl;
      excl += v[q];
    }
    __syncthreads();
    if (t == 1023) carry += woff + inc;                   // total of this slab (t = 1023 holds the last inclusive value)
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_rs_scatter(uint64_t* __restrict__ kb0, uint64_t* __restrict__ kb1,
                                                    int* __restrict__ vb0, int* __restrict__ vb1, int n, int pass,
                                                    const int* __restrict__ plan, int n_tiles,
                                                    const unsigned int* __restrict__ table) {
  if (plan[pass]) return;
  __shared__ unsigned int base[256];                      // global position of the next key of each digit for this tile
  __shared__ unsigned int wcnt[4][256];
  const int src = plan[8 + pass];
  const uint64_t* kin = src ? kb1 : kb0;
  uint64_t* kout = src ? kb0 : kb1;
  const int* vin = src ? vb1 : vb0;
  int* vout = src ? vb0 : vb1;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  base[t] = table[(size_t)t * n_tiles + blockIdx.x];
  const int t0 = blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_TILE / 256; ++r) {
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][t] = 0;
    __syncthreads();
    const int i = t0 + r * 256 + t;
    const bool ok = i < n;
    uint64_t k = ok ? kin[i] : 0ull;
    int v = ok ? vin[i] : 0;
    const int d = (int)((k >> (8 * pass)) & 255);
    // lanes of this wave with the same digit (8 ballots), invalid lanes excluded
    unsigned long long m = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      unsigned long long bal = __ballot((d >> b) & 1);
      m &= ((d >> b) & 1) ? bal : ~bal;
    }
    const int rank_w = __popcll(m & ((1ull << lane) - 1ull));
    if (ok && rank_w == 0) wcnt[wv][d] = (unsigned int)__popcll(m);      // the lowest lane of each digit group
    __syncthreads();
    if (ok) {
      unsigned int off = base[d] + (unsigned int)rank_w;
      for (int w = 0; w < wv; ++w) off += wcnt[w][d];
      kout[off] = k;
      vout[off] = v;
    }
    __syncthreads();
    base[t] += wcnt[0][t] + wcnt[1][t] + wcnt[2][t] + wcnt[3][t];
    __syncthreads();
  }
}

// scratch layout: [hist 8*256 u32][plan 32 int][table 256 * n_tiles u32][keys0 n u64][keys1 n u64][vals0 n int][vals1 n int]
static size_t rs_align(size_t x) { return (x + 255) / 256 * 256; }
extern "C" size_t es_sort_scratch_bytes(int n) {
  size_t nn = (size_t)(n > 0 ? n : 1), tiles = (nn + RS_TILE - 1) / RS_TILE;
  return rs_align(RS_PASSES * 256 * 4) + rs_align(32 * 4) + rs_align(256 * tiles * 4) + 2 * rs_align(nn * 8) + 2 * rs_align(nn * 4) + 1024;
}
__global__ void k_apply_sorted(const int64_t* __restrict__ keys, const int* __restrict__ src, const int* __restrict__ vb0,
                               const int* __restrict__ vb1, const int* __restrict__ plan, int n,
                               int64_t* __restrict__ out_keys, int* __restrict__ out_src) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int p = (plan[16] ? vb1 : vb0)[i];
  out_keys[i] = keys[p];
  if (src) out_src[i] = src[p];
}
__global__ void k_plain_keys(const int64_t* __restrict__ keys, int n, uint64_t* __restrict__ mk, int* __restrict__ idx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  mk[i] = (uint64_t)keys[i];
  idx[i] = i;
}
// out_keys / out_src = keys / src permuted into Z-curve order (batch major) -- or, morton = false (es_sort_u64), into ascending
// order of the keys themselves (non-negative, < 2^62; stable).  scratch: es_sort_scratch_bytes(n).
static int radix_sort(const int64_t* keys, const int* src, int n, void* scratch, size_t scratch_bytes, int64_t* out_keys, int* out_src,
                      void* stream, bool morton) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  if (es_sort_scratch_bytes(n) > scratch_bytes) return -5;
  const int n_tiles = es_cdiv(n, RS_TILE);
  char* p = (char*)scratch;
  unsigned int* hist = (unsigned int*)p;              p += rs_align(RS_PASSES * 256 * 4);
  int* plan = (int*)p;                                p += rs_align(32 * 4);
  unsigned int* table = (unsigned int*)p;             p += rs_align((size_t)256 * n_tiles * 4);
  uint64_t* kb0 = (uint64_t*)p;                       p += rs_align((size_t)n * 8);
  uint64_t* kb1 = (uint64_t*)p;                       p += rs_align((size_t)n * 8);
  int* vb0 = (int*)p;                                 p += rs_align((size_t)n * 4);
  int* vb1 = (int*)p;
  int g = es_cdiv(n, 256);
  ES_TRY(hipMemsetAsync(hist, 0, RS_PASSES * 256 * 4, st));
  if (morton) hipLaunchKernelGGL(k_morton, dim3(g), dim3(256), 0, st, keys, n, kb0, vb0);
  else hipLaunchKernelGGL(k_plain_keys, dim3(g), dim3(256), 0, st, keys, n, kb0, vb0);
  hipLaunchKernelGGL(k_rs_hist_all, dim3(g > 1024 ? 1024 : g), dim3(256), 0, st, kb0, n, hist);
  hipLaunchKernelGGL(k_rs_plan, dim3(1), dim3(256), 0, st, hist, n, plan);
  for (int pass = 0; pass < RS_PASSES; ++pass) {
    hipLaunchKernelGGL(k_rs_count, dim3(n_tiles), dim3(256), 0, st, kb0, kb1, n, pass, plan, n_tiles, table);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, st, table, 256 * n_tiles, pass, plan);
    hipLaunchKernelGGL(k_rs_scatter, dim3(n_tiles), dim3(256), 0, st, kb0, kb1, vb0, vb1, n, pass, plan, n_tiles, table);
  }
  hipLaunchKernelGGL(k_apply_sorted, dim3(g), dim3(256), 0, st, keys, src, vb0, vb1, plan, n, out_keys, out_src);
  ES_CHECK_LAUNCH();
  return 0;
}
extern "C" int es_morton_sort(const int64_t* keys, const int* src, int n, void* scratch, size_t scratch_bytes,
                              int64_t* out_keys, int* out_src, void* stream) {
  return radix_sort(keys, src, n, scratch, scratch_bytes, out_keys, out_src, stream, true);
}
// plain ascending stable sort of non-negative 62-bit keys with an int payload (N4: the random-key order of the device-side
// PointSample draws, data.hip)
extern "C" int es_sort_u64(const int64_t* keys, const int* src, int n, void* scratch, size_t scratch_bytes, int64_t* out_keys,
                           int* out_src, void* stream) {
  return radix_sort(keys, src, n, scratch, scratch_bytes, out_keys, out_src, stream, false);
}
