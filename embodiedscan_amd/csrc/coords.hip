// Coordinate manager kernels: voxelisation, hash-unique (first occurrence), strided /
// generative / union coordinate sets, kernel maps.  HBM/L2-bound integer work.
// Replaces MinkowskiEngine's coordinate manager as used by the reference at
//   embodiedscan/models/detectors/sparse_featfusion_single_stage.py:109-118
//   embodiedscan/models/backbones/mink_resnet.py:58-74
//   embodiedscan/models/dense_heads/fcaf3d_head.py:937-947,1006-1010,1091-1114
#include "common.h"
#include "../../include/es_hip.h"

// ---------------------------------------------------------------- voxel keys (A4)
__global__ void k_voxel_keys(const float* __restrict__ pts, int n, int ld, int batch, float vs,
                             int64_t* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // TRUE f32 division then C truncation toward zero; never p * (1/vs).
  int x = (int)__fdiv_rn(pts[(size_t)i * ld + 0], vs);
  int y = (int)__fdiv_rn(pts[(size_t)i * ld + 1], vs);
  int z = (int)__fdiv_rn(pts[(size_t)i * ld + 2], vs);
  keys[i] = es_pack(batch, x, y, z);
}

extern "C" int es_voxel_keys(const float* points, int n, int ld, int batch, float voxel_size, int64_t* keys,
                             void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_voxel_keys, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, points, n, ld, batch,
                     voxel_size, keys);
  ES_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- exclusive scan (int32)
#define SCAN_T 256
#define SCAN_E 8
#define SCAN_B (SCAN_T * SCAN_E)
__device__ inline int block_excl_scan(int v, int* lds, int& total) {
  // inclusive wave scan
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_T / 64; ++i) {
    int s = lds[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

__global__ void k_scan_local(const int* __restrict__ in, int n, int* __restrict__ out, int* __restrict__ bsum) {
  __shared__ int lds[8];
  int base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_E;
  int v[SCAN_E], s = 0;
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) {
    v[e] = (base + e < n) ? in[base + e] : 0;
    s += v[e];
  }
  int tot, ex = block_excl_scan(s, lds, tot);
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) {
    if (base + e < n) out[base + e] = ex;
    ex += v[e];
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ void k_scan_bsums(int* __restrict__ bsum, int nb, int* __restrict__ total) {
  __shared__ int lds[8];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += SCAN_T) {
    int i = b0 + threadIdx.x;
    int v = (i < nb) ? bsum[i] : 0;
    int tot, ex = block_excl_scan(v, lds, tot);
    int carry = carry_s;
    if (i < nb) bsum[i] = ex + carry;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}
__global__ void k_scan_add(int* __restrict__ out, int n, const int* __restrict__ bsum) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += bsum[i / SCAN_B];
}
// out[n] exclusive scan, *total = sum.  bsum: >= cdiv(n, 2048) ints.
static int scan_i32(const int* in, int n, int* out, int* bsum, int* total, hipStream_t st) {
  int nb = es_cdiv(n, SCAN_B);
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(SCAN_T), 0, st, in, n, out, bsum);
  hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(SCAN_T), 0, st, bsum, nb, total);
  hipLaunchKernelGGL(k_scan_add, dim3(es_cdiv(n, 256)), dim3(256), 0, st, out, n, bsum);
  ES_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- hash table
__device__ inline uint32_t table_insert_slot(int64_t* tkeys, uint32_t mask, int64_t key) {
  uint32_t s = es_hash(key, mask);
  while (true) {
    unsigned long long prev =
        atomicCAS((unsigned long long*)&tkeys[s], (unsigned long long)ES_EMPTY_KEY, (unsigned long long)key);
    if (prev == (unsigned long long)ES_EMPTY_KEY || prev == (unsigned long long)key) return s;
    s = (s + 1) & mask;
  }
}
__device__ inline int table_find_slot(const int64_t* tkeys, uint32_t mask, int64_t key) {
  uint32_t s = es_hash(key, mask);
  for (uint32_t it = 0; it <= mask; ++it) {
    int64_t k = tkeys[s];
    if (k == key) return (int)s;
    if (k == ES_EMPTY_KEY) return -1;
    s = (s + 1) & mask;
  }
  return -1;
}

__global__ void k_insert_min(const int64_t* __restrict__ keys, int n, int64_t* tkeys, int* tvals, uint32_t mask) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = table_insert_slot(tkeys, mask, keys[i]);
  atomicMin(&tvals[s], i);                 // first occurrence wins, deterministic
}
__global__ void k_flag_winner(const int64_t* __restrict__ keys, int n, const int64_t* tkeys, const int* tvals,
                              uint32_t mask, int* __restrict__ flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = table_find_slot(tkeys, mask, keys[i]);
  flag[i] = (tvals[s] == i) ? 1 : 0;
}
__global__ void k_compact_unique(const int64_t* __restrict__ keys, int n, const int* __restrict__ flag,
                                 const int* __restrict__ pos, const int64_t* tkeys, int* tvals, uint32_t mask,
                                 int64_t* __restrict__ out_keys, int* __restrict__ out_src) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  int p = pos[i];
  int64_t k = keys[i];
  out_keys[p] = k;
  if (out_src) out_src[p] = i;
  tvals[table_find_slot(tkeys, mask, k)] = p;     // table now maps key -> unique row
}

// scratch: ints [flag n][pos n][bsum cdiv(n,2048)+1][total 1]
extern "C" int es_unique_first(const int64_t* keys, int n, int64_t* tkeys, int* tvals, int cap, int* scratch,
                               int64_t* out_keys, int* out_src, int* count_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  *count_host = 0;
  ES_TRY(hipMemsetAsync(tkeys, 0xFF, (size_t)cap * 8, st));
  ES_TRY(hipMemsetAsync(tvals, 0x7F, (size_t)cap * 4, st));
  if (n <= 0) return 0;
  uint32_t mask = (uint32_t)cap - 1;
  int* flag = scratch;
  int* pos = scratch + n;
  int* bsum = pos + n;
  int* total = bsum + es_cdiv(n, SCAN_B) + 1;
  int g = es_cdiv(n, 256);
  hipLaunchKernelGGL(k_insert_min, dim3(g), dim3(256), 0, st, keys, n, tkeys, tvals, mask);
  hipLaunchKernelGGL(k_flag_winner, dim3(g), dim3(256), 0, st, keys, n, tkeys, tvals, mask, flag);
  int rc = scan_i32(flag, n, pos, bsum, total, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_compact_unique, dim3(g), dim3(256), 0, st, keys, n, flag, pos, tkeys, tvals, mask, out_keys,
                     out_src);
  ES_CHECK_LAUNCH();
  ES_TRY(hipMemcpyAsync(count_host, total, 4, hipMemcpyDeviceToHost, st));
  ES_TRY(hipStreamSynchronize(st));
  return 0;
}

__global__ void k_insert_rows(const int64_t* __restrict__ keys, int n, int64_t* tkeys, int* tvals, uint32_t mask) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = table_insert_slot(tkeys, mask, keys[i]);
  tvals[s] = i;
}
// build key -> row table for an already-unique key list
extern "C" int es_build_table(const int64_t* keys, int n, int64_t* tkeys, int* tvals, int cap, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  ES_TRY(hipMemsetAsync(tkeys, 0xFF, (size_t)cap * 8, st));
  ES_TRY(hipMemsetAsync(tvals, 0xFF, (size_t)cap * 4, st));
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_insert_rows, dim3(es_cdiv(n, 256)), dim3(256), 0, st, keys, n, tkeys, tvals,
                     (uint32_t)cap - 1);
  ES_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- key transforms
__device__ inline int floor_to(int v, int ts) {
  int q = v / ts;
  if ((v % ts != 0) && ((v < 0) != (ts < 0))) --q;    // true floor for negatives (ME semantics)
  return q * ts;
}
__global__ void k_stride_keys(const int64_t* __restrict__ in, int n, int ts, int64_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b, x, y, z;
  es_unpack(in[i], b, x, y, z);
  out[i] = es_pack(b, floor_to(x, ts), floor_to(y, ts), floor_to(z, ts));
}
extern "C" int es_stride_keys(const int64_t* in_keys, int n, int out_ts, int64_t* out_keys, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_stride_keys, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in_keys, n, out_ts,
                     out_keys);
  ES_CHECK_LAUNCH();
  return 0;
}
__global__ void k_keys_to_coords(const int64_t* __restrict__ keys, int n, int* __restrict__ c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b, x, y, z;
  es_unpack(keys[i], b, x, y, z);
  ((int4*)c)[i] = make_int4(b, x, y, z);
}
extern "C" int es_keys_to_coords(const int64_t* keys, int n, int* coords, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_keys_to_coords, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, keys, n, coords);
  ES_CHECK_LAUNCH();
  return 0;
}
__global__ void k_batch_offsets(const int64_t* __restrict__ keys, int n, int nb, int* __restrict__ off) {
  int b = threadIdx.x;
  if (b > nb) return;
  int lo = 0, hi = n;                       // first row with batch >= b (rows are batch-major)
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if ((int)(keys[mid] >> (3 * ES_FIELD)) < b) lo = mid + 1; else hi = mid;
  }
  off[b] = lo;
}
extern "C" int es_batch_offsets(const int64_t* keys, int n, int n_batch, int* offsets_dev, void* stream) {
  hipLaunchKernelGGL(k_batch_offsets, dim3(1), dim3(n_batch + 1 > 64 ? 256 : 64), 0, (hipStream_t)stream, keys, n,
                     n_batch, offsets_dev);
  ES_CHECK_LAUNCH();
  return 0;
}
// ---------------------------------------------------------------- all strided sets of a backbone in ONE host round trip
// MinkResNet needs the coordinate sets at tensor strides 2, 4, ... of the root set; each is the hash-unique (first occurrence)
// of the previous one floored to the next stride, and each used to cost a row-count read-back plus one for its per-sample
// offsets -- on the critical path of every step (profiles/r3_stream_timeline.txt: the main stream is ~25 % busy for the first
// 4.5 ms).  The chain is not needed: flooring commutes (floor(floor(x, 2), 4) == floor(x, 4)) and first-occurrence order is
// preserved -- the first row of level l that maps to a level-(l+1) key K is the image of the first ROOT row that maps to K
// (any earlier root row mapping to K would have an earlier image) -- so every level is the hash-unique of the ROOT keys floored
// to its stride, in root order: same rows, same row order, same key -> row tables as the chain.  All levels are queued back to
// back, their counts and per-sample offsets land in one small device array, and ONE copy + ONE synchronisation brings them to
// the host.  res layout per level: [count, offsets[0 .. n_batch]].
__global__ void k_batch_offsets_dev(const int64_t* __restrict__ keys, const int* __restrict__ n_dev, int nb, int* __restrict__ off) {
  int b = threadIdx.x;
  if (b > nb) return;
  int lo = 0, hi = *n_dev;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if ((int)(keys[mid] >> (3 * ES_FIELD)) < b) lo = mid + 1; else hi = mid;
  }
  off[b] = lo;
}
extern "C" int es_strided_chain(const int64_t* root_keys, int n, int n_batch, int n_levels, const int* ts_host,
                                int64_t* tmp_keys, int* scratch, void** tkeys, void** tvals, const int* caps_host,
                                void** out_keys, int* res_dev, int* res_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int per = n_batch + 2;
  for (int i = 0; i < n_levels * per; ++i) res_host[i] = 0;
  if (n <= 0 || n_levels <= 0) return 0;
  int* flag = scratch;
  int* pos = scratch + n;
  int* bsum = pos + n;
  const int g = es_cdiv(n, 256);
  for (int l = 0; l < n_levels; ++l) {
    int64_t* tk = (int64_t*)tkeys[l];
    int* tv = (int*)tvals[l];
    int64_t* ok = (int64_t*)out_keys[l];
    const int cap = caps_host[l];
    if (cap <= 0 || (cap & (cap - 1))) return -4;
    const uint32_t mask = (uint32_t)cap - 1;
    int* total = res_dev + l * per;
    ES_TRY(hipMemsetAsync(tk, 0xFF, (size_t)cap * 8, st));
    ES_TRY(hipMemsetAsync(tv, 0x7F, (size_t)cap * 4, st));
    hipLaunchKernelGGL(k_stride_keys, dim3(g), dim3(256), 0, st, root_keys, n, ts_host[l], tmp_keys);
    hipLaunchKernelGGL(k_insert_min, dim3(g), dim3(256), 0, st, tmp_keys, n, tk, tv, mask);
    hipLaunchKernelGGL(k_flag_winner, dim3(g), dim3(256), 0, st, tmp_keys, n, tk, tv, mask, flag);
    int rc = scan_i32(flag, n, pos, bsum, total, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compact_unique, dim3(g), dim3(256), 0, st, tmp_keys, n, flag, pos, tk, tv, mask, ok, (int*)nullptr);
    hipLaunchKernelGGL(k_batch_offsets_dev, dim3(1), dim3(n_batch + 1 > 64 ? 256 : 64), 0, st, ok, total, n_batch, total + 1);
    ES_CHECK_LAUNCH();
  }
  ES_TRY(hipMemcpyAsync(res_host, res_dev, (size_t)n_levels * per * 4, hipMemcpyDeviceToHost, st));
  ES_TRY(hipStreamSynchronize(st));
  return 0;
}

__global__ void k_gen_children(const int64_t* __restrict__ in, int n, int half, int64_t* __restrict__ out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 8) return;
  int i = t >> 3, k = t & 7;
  int b, x, y, z;
  es_unpack(in[i], b, x, y, z);
  out[t] = es_pack(b, x + (k & 1) * half, y + ((k >> 1) & 1) * half, z + ((k >> 2) & 1) * half);
}
extern "C" int es_gen_children_keys(const int64_t* in_keys, int n, int half_ts, int64_t* out_keys, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_gen_children, dim3(es_cdiv((long long)n * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     in_keys, n, half_ts, out_keys);
  ES_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- kernel maps (A6)
__global__ void k_kernel_map(const int64_t* __restrict__ out_keys, int n_out, const int64_t* __restrict__ tkeys,
                             const int* __restrict__ tvals, uint32_t mask, int ksize, int in_ts,
                             int* __restrict__ nbr) {
  int K = ksize * ksize * ksize;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_out * K) return;
  int j = (int)(t / K), k = (int)(t % K);
  int c = (ksize & 1) ? ksize / 2 : 0;
  int ox = (k % ksize - c) * in_ts, oy = ((k / ksize) % ksize - c) * in_ts, oz = (k / (ksize * ksize) - c) * in_ts;
  int b, x, y, z;
  es_unpack(out_keys[j], b, x, y, z);
  nbr[t] = es_table_find(tkeys, tvals, mask, es_pack(b, x + ox, y + oy, z + oz));
}
extern "C" int es_kernel_map(const int64_t* out_keys, int n_out, const int64_t* tkeys, const int* tvals, int cap,
                             int ksize, int in_ts, int* nbr, void* stream) {
  if (n_out <= 0) return 0;
  long long tot = (long long)n_out * ksize * ksize * ksize;
  hipLaunchKernelGGL(k_kernel_map, dim3(es_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, out_keys, n_out,
                     tkeys, tvals, (uint32_t)cap - 1, ksize, in_ts, nbr);
  ES_CHECK_LAUNCH();
  return 0;
}
__global__ void k_inverse_map(const int* __restrict__ nbr, long long tot, int K, int* __restrict__ inv) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= tot) return;
  int i = nbr[t];
  if (i >= 0) inv[(long long)i * K + (t % K)] = (int)(t / K);
}
extern "C" int es_inverse_map(const int* nbr, int n_out, int K, int n_in, int* inv, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_in > 0) ES_TRY(hipMemsetAsync(inv, 0xFF, (size_t)n_in * K * 4, st));
  if (n_out <= 0) return 0;
  long long tot = (long long)n_out * K;
  hipLaunchKernelGGL(k_inverse_map, dim3(es_cdiv(tot, 256)), dim3(256), 0, st, nbr, tot, K, inv);
  ES_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- union (sparse a + b)
__global__ void k_union_flag(const int64_t* __restrict__ kb, int nb, const int64_t* tka, const int* tva,
                             uint32_t mask, int* __restrict__ hit, int* __restrict__ isnew) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nb) return;
  int h = es_table_find(tka, tva, mask, kb[j]);
  hit[j] = h;
  isnew[j] = h < 0;
}
__global__ void k_union_place(const int64_t* __restrict__ ka, int na, const int64_t* __restrict__ kb, int nb,
                              const int* __restrict__ a_off, const int* __restrict__ b_off, int n_batch,
                              const int* __restrict__ hit, const int* __restrict__ scan, const int* __restrict__ total,
                              int* __restrict__ pos_a, int* __restrict__ pos_b, int64_t* __restrict__ out_keys) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < na) {
    int64_t k = ka[t];
    int b = (int)(k >> (3 * ES_FIELD));
    int bo = b_off[b];
    int new_before = (bo < nb) ? scan[bo] : *total;     // new b-rows in batches < b
    int p = t + new_before;
    pos_a[t] = p;
    out_keys[p] = k;
  } else if (t < na + nb) {
    int j = t - na;
    int64_t k = kb[j];
    int b = (int)(k >> (3 * ES_FIELD));
    if (hit[j] < 0) {
      int p = a_off[b + 1] + scan[j];
      pos_b[j] = p;
      out_keys[p] = k;
    }
  }
}
__global__ void k_union_fix(int nb, const int* __restrict__ hit, const int* __restrict__ pos_a,
                            int* __restrict__ pos_b) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nb && hit[j] >= 0) pos_b[j] = pos_a[hit[j]];
}
// scratch ints: [hit nb][isnew nb][scan nb][bsum cdiv(nb,2048)+1][total 1]
extern "C" int es_union_plan(const int64_t* keys_a, int na, const int64_t* tkeys_a, const int* tvals_a, int cap_a,
                             const int64_t* keys_b, int nb, const int* a_off_dev, const int* b_off_dev, int n_batch,
                             int* scratch, int* pos_a, int* pos_b, int64_t* out_keys, int* count_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int* hit = scratch;
  int* isnew = hit + nb;
  int* scan = isnew + nb;
  int* bsum = scan + nb;
  int* total = bsum + es_cdiv(nb > 0 ? nb : 1, SCAN_B) + 1;
  ES_TRY(hipMemsetAsync(total, 0, 4, st));
  if (nb > 0) {
    hipLaunchKernelGGL(k_union_flag, dim3(es_cdiv(nb, 256)), dim3(256), 0, st, keys_b, nb, tkeys_a, tvals_a,
                       (uint32_t)cap_a - 1, hit, isnew);
    int rc = scan_i32(isnew, nb, scan, bsum, total, st);
    if (rc) return rc;
  }
  if (na + nb > 0) {
    hipLaunchKernelGGL(k_union_place, dim3(es_cdiv(na + nb, 256)), dim3(256), 0, st, keys_a, na, keys_b, nb,
                       a_off_dev, b_off_dev, n_batch, hit, scan, total, pos_a, pos_b, out_keys);
    if (nb > 0) hipLaunchKernelGGL(k_union_fix, dim3(es_cdiv(nb, 256)), dim3(256), 0, st, nb, hit, pos_a, pos_b);
  }
  ES_CHECK_LAUNCH();
  // (straight into the caller's buffer -- pinned host memory on the product path: a copy into a pageable stack variable is staged by
  // the runtime and was seen to wait for work queued on OTHER streams)
  ES_TRY(hipMemcpyAsync(count_host, total, 4, hipMemcpyDeviceToHost, st));
  ES_TRY(hipStreamSynchronize(st));
  *count_host += na;
  return 0;
}

// ---------------------------------------------------------------- interpolation map (features_at_coordinates)
__global__ void k_interp_map(const int64_t* __restrict__ q, int n, const int64_t* tkeys, const int* tvals,
                             uint32_t mask, int ts, int* __restrict__ idx, float* __restrict__ w) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 8) return;
  int i = t >> 3, k = t & 7;
  int b, x, y, z;
  es_unpack(q[i], b, x, y, z);
  int lx = floor_to(x, ts), ly = floor_to(y, ts), lz = floor_to(z, ts);
  float fx = (float)(x - lx) / (float)ts, fy = (float)(y - ly) / (float)ts, fz = (float)(z - lz) / (float)ts;
  int sx = k & 1, sy = (k >> 1) & 1, sz = (k >> 2) & 1;
  float wk = 1.0f;
  wk = wk * (sx ? fx : (1.0f - fx));
  wk = wk * (sy ? fy : (1.0f - fy));
  wk = wk * (sz ? fz : (1.0f - fz));
  idx[t] = es_table_find(tkeys, tvals, mask, es_pack(b, lx + sx * ts, ly + sy * ts, lz + sz * ts));
  w[t] = wk;
}
extern "C" int es_interp_map(const int64_t* query_keys, int n, const int64_t* tkeys, const int* tvals, int cap,
                             int table_ts, int* idx, float* w, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_interp_map, dim3(es_cdiv((long long)n * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     query_keys, n, tkeys, tvals, (uint32_t)cap - 1, table_ts, idx, w);
  ES_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- mask compaction
__global__ void k_compact_keys(const int64_t* __restrict__ keys, int n, const int* __restrict__ mask,
                               const int* __restrict__ pos, int64_t* __restrict__ out_keys, int* __restrict__ src) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !mask[i]) return;
  out_keys[pos[i]] = keys[i];
  src[pos[i]] = i;
}
// scratch ints: [pos n][bsum cdiv(n,2048)+1][total 1];  mask is int32 0/1
extern "C" int es_compact_mask(const int64_t* keys, int n, const int* mask, int* scratch, int64_t* out_keys,
                               int* out_src, int* count_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (count_host) *count_host = 0;
  if (n <= 0) return 0;
  int* pos = scratch;
  int* bsum = pos + n;
  int* total = bsum + es_cdiv(n, SCAN_B) + 1;
  int rc = scan_i32(mask, n, pos, bsum, total, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_compact_keys, dim3(es_cdiv(n, 256)), dim3(256), 0, st, keys, n, mask, pos, out_keys, out_src);
  ES_CHECK_LAUNCH();
  if (count_host) {                   // NULL: the caller knows the count (top-k pruning keeps min(n_b, k) rows per sample)
    ES_TRY(hipMemcpyAsync(count_host, total, 4, hipMemcpyDeviceToHost, st));
    ES_TRY(hipStreamSynchronize(st));
  }
  return 0;
}

__global__ void k_coords_to_points(const int* __restrict__ c, int n, float vs, float* __restrict__ p) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 v = ((const int4*)c)[i];
  p[(size_t)i * 3 + 0] = __fmul_rn((float)v.y, vs);
  p[(size_t)i * 3 + 1] = __fmul_rn((float)v.z, vs);
  p[(size_t)i * 3 + 2] = __fmul_rn((float)v.w, vs);
}
extern "C" int es_coords_to_points(const int* coords, int n, float voxel_size, float* points, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_coords_to_points, dim3(es_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, coords, n,
                     voxel_size, points);
  ES_CHECK_LAUNCH();
  return 0;
}
