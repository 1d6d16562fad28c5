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
// The light election (round 4, after profiles/r4q: the grounding step had become 9 ms SLOWER with es_last_block in its small
// reductions): an agent-scope fence is buffer_wbl2 (+ buffer_inv) on gfx950 -- it writes back / drops the WHOLE L2 of the XCD, i.e.
// also the tiles of the large kernels running beside this one on other streams.  When the partial results travel through
// device-scope ATOMIC stores / loads instead of plain ones (es_coh_store / es_coh_load: performed at the memory-side coherence
// point, never held dirty or stale in an XCD's L2) no cache maintenance is needed at all: wait until this thread's stores have
// been performed (s_waitcnt), barrier, count the ticket, and the winner reads the partials with coherent loads.
__device__ inline void es_coh_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float es_coh_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void es_coh_store_u(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned int es_coh_load_u(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// sum of `count` coherent loads p[0], p[stride], ... IN INDEX ORDER, with 16 loads in flight: a coherent load travels to the memory
// side (~1-2 us), so a loop that waits for each one before issuing the next serialises 100 partials into 0.2 ms (round 4: that
// alone made the grounding step 8 ms slower); the order of the additions -- and so the result -- is unchanged.
__device__ inline float es_coh_sum(const float* p, int count, size_t stride) {
  float t = 0.f;
  int b = 0;
  for (; b + 32 <= count; b += 32) {          // (round 6: 32 in flight where the list is long enough -- same order of additions)
    float v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = es_coh_load(p + (size_t)(b + u) * stride);
#pragma unroll
    for (int u = 0; u < 32; ++u) t += v[u];
  }
  for (; b + 16 <= count; b += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = es_coh_load(p + (size_t)(b + u) * stride);
#pragma unroll
    for (int u = 0; u < 16; ++u) t += v[u];
  }
  for (; b < count; ++b) t += es_coh_load(p + (size_t)b * stride);
  return t;
}
// two lists at once (LayerNorm backward: dw and db partials of the same workgroups): 2 x 16 loads in flight, each sum in index order
__device__ inline void es_coh_sum2(const float* pa, const float* pb, int count, size_t stride, float& ta, float& tb) {
  ta = tb = 0.f;
  int b = 0;
  for (; b + 16 <= count; b += 16) {
    float va[16], vb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { va[u] = es_coh_load(pa + (size_t)(b + u) * stride); vb[u] = es_coh_load(pb + (size_t)(b + u) * stride); }
#pragma unroll
    for (int u = 0; u < 16; ++u) { ta += va[u]; tb += vb[u]; }
  }
  for (; b < count; ++b) { ta += es_coh_load(pa + (size_t)b * stride); tb += es_coh_load(pb + (size_t)b * stride); }
}
__device__ inline bool es_last_block_light(unsigned int* ticket, unsigned int nblocks) {
  __shared__ unsigned int es_s_last3;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);                       // vmcnt(0) lgkmcnt(0): every coherent store / atomic of this thread has been performed
  __syncthreads();
  if ((threadIdx.x | threadIdx.y | threadIdx.z) == 0) {
    unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    es_s_last3 = (t == nblocks - 1u) ? 1u : 0u;
    if (es_s_last3) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return es_s_last3 != 0u;
}
#ifdef ES_EMU
#define ES_WAIT_VM0() ((void)0)                        // (tests/emu: x86 cannot assemble it; the emulated stores are performed at once)
#else
#define ES_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
// Selectable form (es_set_option key 18): 0 (default) = the light election above -- the exact sc1-store / vmcnt(0) / barrier /
// relaxed-ticket / sc1-load recipe cdna_hip_programming.md section 5 lists as "equally valid" for a split-K seam; 1 = every
// workgroup additionally issues an agent-scope RELEASE fence before the ticket and the winner an agent-scope ACQUIRE fence after
// it (buffer_wbl2 / buffer_inv: what the LLVM AMDGPU memory model guarantees for ANY store / load flavour) -- slower (it writes
// back / drops the XCD's L2 under the big kernels on the other streams), kept for the stress test and as the fall-back.
__device__ inline bool es_last_block_sel(unsigned int* ticket, unsigned int nblocks, int conservative) {
  if (!conservative) return es_last_block_light(ticket, nblocks);
  __shared__ unsigned int es_s_last4;
  ES_WAIT_VM0();
  __syncthreads();
  if ((threadIdx.x | threadIdx.y | threadIdx.z) == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    ES_WAIT_VM0();                                     // (the wait behind buffer_wbl2 restated where the compiler cannot drop it)
    unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    es_s_last4 = (t == nblocks - 1u) ? 1u : 0u;
    if (es_s_last4) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  return es_s_last4 != 0u;
}
#define ES_TICKET_FLOATS 4          // floats reserved at the head of a workspace for the ticket (keeps partials 16-byte aligned)
