// Shared device/host helpers for the es_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ES_OFF (1 << 17)          // coordinate bias (matches oracle/coords.py)
#define ES_FIELD 18
#define ES_FMASK ((1ll << ES_FIELD) - 1)
#define ES_EMPTY_KEY (-1ll)

#define ES_CHECK_LAUNCH()                          \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)
#define ES_TRY(x)                                  \
  do {                                             \
    hipError_t e__ = (x);                          \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int es_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__host__ __device__ inline int64_t es_pack(int b, int x, int y, int z) {
  return ((int64_t)b << (3 * ES_FIELD)) | ((int64_t)(x + ES_OFF) << (2 * ES_FIELD)) |
         ((int64_t)(y + ES_OFF) << ES_FIELD) | (int64_t)(z + ES_OFF);
}
__host__ __device__ inline void es_unpack(int64_t k, int& b, int& x, int& y, int& z) {
  b = (int)(k >> (3 * ES_FIELD));
  x = (int)((k >> (2 * ES_FIELD)) & ES_FMASK) - ES_OFF;
  y = (int)((k >> ES_FIELD) & ES_FMASK) - ES_OFF;
  z = (int)(k & ES_FMASK) - ES_OFF;
}

// 64-bit mix (splitmix64 finaliser) -> slot
__device__ inline uint32_t es_hash(int64_t k, uint32_t mask) {
  uint64_t x = (uint64_t)k;
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return (uint32_t)x & mask;
}

// open-addressing lookup: returns value or -1
__device__ inline int es_table_find(const int64_t* __restrict__ tkeys, const int* __restrict__ tvals,
                                    uint32_t mask, int64_t key) {
  uint32_t s = es_hash(key, mask);
  for (uint32_t it = 0; it <= mask; ++it) {
    int64_t k = tkeys[s];
    if (k == key) return tvals[s];
    if (k == ES_EMPTY_KEY) return -1;
    s = (s + 1) & mask;
  }
  return -1;
}

__device__ inline float es_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double es_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ inline int es_wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline int es_wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// two floats -> packed bf16x2, round to nearest even (v_cvt_pk_bf16_f32): the one conversion every bf16 row / shadow uses
__device__ inline uint32_t es_pack_bf16(float a, float b) {
  typedef float es_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 es_bf16x2 __attribute__((ext_vector_type(2)));
  es_f32x2 x = {a, b};
  es_bf16x2 y = __builtin_convertvector(x, es_bf16x2);
  return *(uint32_t*)&y;
}

// Deterministic cross-workgroup reduction without a second launch and without float atomics: every workgroup stores its
// partial result in a workspace, then calls es_last_block(); exactly one workgroup -- the last to arrive -- gets `true` and
// adds the partials up in WORKGROUP ORDER (a fixed order: bit-reproducible run to run).  `ticket` is one unsigned int in
// global memory that must be 0 before the launch and is 0 again after it (launches sharing a ticket must be stream-ordered).
// Release / acquire: every thread fences its own stores before the barrier; the winner fences again before it reads.
__device__ inline bool es_last_block(unsigned int* ticket, unsigned int nblocks) {
  __shared__ unsigned int es_s_last;
  __threadfence();
  __syncthreads();
  if ((threadIdx.x | threadIdx.y | threadIdx.z) == 0) {
    unsigned int t = atomicAdd(ticket, 1u);
    es_s_last = (t == nblocks - 1u) ? 1u : 0u;
    if (es_s_last) *ticket = 0u;
  }
  __syncthreads();
  const bool last = es_s_last != 0u;
  if (last) __threadfence();
  return last;
}
// The same election with a RELEASE-only fence in every workgroup and NO acquire in the winner.  A seq_cst agent-scope fence is
// buffer_wbl2 + buffer_inv on gfx950: executed by every workgroup of a streaming kernel the invalidate throws away the XCD's whole
// L2 again and again (measured: the tap-split convolutions ran 2x longer, profiles/r4k_critical.txt).  The winner may read the
// partials with plain loads when -- as in split_tail -- (a) nobody reads them before the election, so its CU / XCD caches hold
// no stale copy (caches are invalidated at kernel start), (b) no two workgroups write into one 128-byte line, and (c) the reads
// are control-dependent on the ticket value.
__device__ inline bool es_last_block_release_only(unsigned int* ticket, unsigned int nblocks) {
  __shared__ unsigned int es_s_last2;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if ((threadIdx.x | threadIdx.y | threadIdx.z) == 0) {
    unsigned int t = atomicAdd(ticket, 1u);
    es_s_last2 = (t == nblocks - 1u) ? 1u : 0u;
    if (es_s_last2) *ticket = 0u;
  }
  __syncthreads();
  return es_s_last2 != 0u;
}
#define ES_TICKET_FLOATS 4          // floats reserved at the head of a workspace for the ticket (keeps partials 16-byte aligned)
