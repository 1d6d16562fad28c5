// TEST INFRASTRUCTURE ONLY (tests/emu): a host-side stand-in for <hip/hip_runtime.h> that lets the kernel SOURCES of
// embodiedscan_amd/csrc be compiled for x86 (amdclang++ in host mode: it understands ext_vector_type, __bf16,
// __builtin_convertvector like the device compiler does) and executed on the CPU with the CDNA execution model emulated:
//   * a workgroup = blockDim fibers (own stacks, a 15-instruction context switch) in ONE OS thread, workgroups run one after the
//     other -> `__shared__` = static;
//   * __syncthreads / s_barrier = a fiber barrier over the workgroup's live threads;
//   * wave operations (__shfl*, __ballot, MFMA) = a rendezvous of the wave's 64 lanes with the operands exchanged through a
//     per-wave buffer; v_mfma_f32_16x16x32_bf16 / 16x16x4f32 are evaluated from the lanes' fragments with the operand layout
//     the shipped kernels rely on (A: lane l = row l % 16, k-slice l / 16; B: column l % 16, k-slice l / 16; D: column
//     l % 16, rows 4 (l / 16) .. + 3) -- the layout the GPU parity tests pin;
//   * global_load_lds = 16 bytes per lane to (LDS base + 16 * lane), performed at issue (the emulator is MORE permissive than
//     the hardware about waitcnt / barrier placement: it checks indexing, tiling and arithmetic, not the memory model).
// Nothing in the product loads the library built from this (embodiedscan_amd/hip.py binds libes_hip.so only); it exists so that
// kernel LOGIC can be unit-tested in the CPU suite and developed between GPU sessions.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>
#include <functional>

#define ES_EMU 1
#define __host__
#define __device__
#define __global__
#define __shared__ static
#define __launch_bounds__(...)
#define __forceinline__ inline

struct uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDeviceToHost 2
#define hipMemcpyHostToDevice 1

namespace emu {
struct Fiber;
extern Fiber* g_cur;
extern dim3 g_block_idx, g_block_dim, g_grid_dim;
const uint3& tid();
int lane();
void launch(const char* name, dim3 grid, dim3 block, size_t dyn_shared_bytes, const std::function<void()>& body);
void* dyn_shared();
// LDS-DMA pieces of the calling lane.  Mode 0 (eager): the copy happens at issue.  Mode 1 (lazy): it is queued and happens as
// LATE as the program allows -- when an s_waitcnt vmcnt(n) retires it (oldest first, until n remain) or the thread ends: a
// kernel that reads a tile before waiting for it sees stale LDS and fails its parity test
void dma_issue(void* lds_dst, const void* src, int bytes);
void count_mfma();       // (work counters of the launches since es_emu_take_counters: MFMA / LDS-DMA wave instructions, barriers)
void waitcnt_vm(int n);      // the launch's dynamic LDS (`extern __shared__ T name[]` is rewritten to `T* name = (T*)emu::dyn_shared()`)
void block_barrier();
// wave rendezvous at call site `site`: the lane deposits `bytes` (<= 64) bytes and waits for the lanes that execute the same
// operation with it; returns their operands (64 slots of 64 bytes; zeros for lanes that do not take part) and the mask of the
// participating lanes, valid until the lane's next wave operation.  Divergence: see emu_runtime.cpp.
struct WaveView { const unsigned char* slots; unsigned long long mask; };
WaveView wave_exchange(const void* site, const void* mine, int bytes);
}  // namespace emu

#define threadIdx (emu::tid())
#define blockIdx (emu::g_block_idx)
#define blockDim (emu::g_block_dim)
#define gridDim (emu::g_grid_dim)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch(#kernel, dim3(grid), dim3(block), (size_t)(shmem), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { emu::block_barrier(); }
static inline void __threadfence() {}
#define __builtin_amdgcn_s_barrier() emu::block_barrier()
#define __builtin_amdgcn_s_waitcnt(x) emu::waitcnt_vm(0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define ES_EMU_WAITCNT_VM(n) emu::waitcnt_vm((n))     // build.py: asm volatile("s_waitcnt vmcnt(n)") -> this
#define ES_EMU_WAITCNT_NONE() ((void)0)               // ... lgkmcnt-only waits: LDS reads are synchronous here

#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
#define __hip_atomic_load(p, order, scope) (*(p))
template <class T>
static inline T emu_fetch_add(T* p, T v) { T o = *p; *p = o + v; return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) emu_fetch_add((p), (v))
template <class T>
static inline T atomicAdd(T* p, T v) { return emu_fetch_add(p, v); }
static inline unsigned atomicAdd(unsigned* p, int v) { return emu_fetch_add(p, (unsigned)v); }
template <class T>
static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T>
static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T>
static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
template <class T>
static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T>
static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }

// IEEE single operations that must not be contracted into FMAs (integer decisions depend on them): plain operators here, the
// emulated build is compiled with -ffp-contract=off
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
#define __expf(x) expf(x)          // (fast-math intrinsics of the device: the tolerances of the callers' tests cover the difference)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
template <class T>
static inline T unsafeAtomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
using std::max;
using std::min;
static inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }
static inline long long min(long long a, int b) { return a < b ? a : b; }
static inline long long max(long long a, int b) { return a > b ? a : b; }

// Every wave operation below is a NON-inlined function: its return address identifies the call site in the kernel, which is
// what lanes rendezvous on (emu_runtime.cpp).
#define ES_EMU_WAVEOP __attribute__((noinline))
template <class T>
ES_EMU_WAVEOP T __shfl(T v, int src, int width = 64) {
  static_assert(sizeof(T) <= 64, "exchange slot");
  const emu::WaveView x = emu::wave_exchange(__builtin_return_address(0), &v, (int)sizeof(T));
  const int l = emu::lane(), base = l & ~(width - 1);
  T r;
  memcpy(&r, x.slots + 64 * (base + (src & (width - 1))), sizeof(T));
  return r;
}
template <class T>
ES_EMU_WAVEOP T __shfl_xor(T v, int mask, int width = 64) {
  const emu::WaveView x = emu::wave_exchange(__builtin_return_address(0), &v, (int)sizeof(T));
  const int l = emu::lane(), src = l ^ mask;
  T r = v;
  if ((src & ~(width - 1)) == (l & ~(width - 1))) memcpy(&r, x.slots + 64 * src, sizeof(T));
  return r;
}
template <class T>
ES_EMU_WAVEOP T __shfl_up(T v, unsigned delta, int width = 64) {
  const emu::WaveView x = emu::wave_exchange(__builtin_return_address(0), &v, (int)sizeof(T));
  const int l = emu::lane(), src = l - (int)delta;
  T r = v;
  if (src >= (l & ~(width - 1))) memcpy(&r, x.slots + 64 * src, sizeof(T));
  return r;
}
template <class T>
ES_EMU_WAVEOP T __shfl_down(T v, unsigned delta, int width = 64) {
  const emu::WaveView x = emu::wave_exchange(__builtin_return_address(0), &v, (int)sizeof(T));
  const int l = emu::lane(), src = l + (int)delta;
  T r = v;
  if (src < (l & ~(width - 1)) + width) memcpy(&r, x.slots + 64 * src, sizeof(T));
  return r;
}
ES_EMU_WAVEOP static unsigned long long __ballot(int pred) {
  const unsigned char p = pred ? 1 : 0;
  const emu::WaveView x = emu::wave_exchange(__builtin_return_address(0), &p, 1);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (((x.mask >> l) & 1ull) && x.slots[64 * l]) m |= 1ull << l;
  return m;
}

// ---- matrix cores.  Fragment types as the kernels declare them (ext_vector_type): 8 x __bf16 / 4 x float.
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline float emu_bf16_to_f32(__bf16 h) {
  unsigned short s;
  memcpy(&s, &h, 2);
  return __uint_as_float((unsigned)s << 16);
}
// D = A (16 x 32) * B (32 x 16) + C; products of bf16 values are exact in f32, the sum is taken in double and rounded once
// (the hardware's internal order is not specified; the parity tolerances of the tests are what both must meet)
ES_EMU_WAVEOP static emu_f32x4 emu_mfma_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c) {
  struct Slot { float a[8]; float b[8]; } mine;              // the lane's fragments, widened once (bf16 -> f32 is exact)
  emu::count_mfma();
  for (int t = 0; t < 8; ++t) {
    mine.a[t] = emu_bf16_to_f32(a[t]);
    mine.b[t] = emu_bf16_to_f32(b[t]);
  }
  const unsigned char* x = emu::wave_exchange(__builtin_return_address(0), &mine, (int)sizeof(Slot)).slots;
  const int l = emu::lane(), j = l & 15, i0 = 4 * (l >> 4);
  emu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + r;
    double s = 0;
    for (int kc = 0; kc < 4; ++kc) {                         // k = 8 kc + t lives in lane (row or column) + 16 kc, element t
      const Slot* sa = (const Slot*)(x + 64 * (i + 16 * kc));
      const Slot* sb = (const Slot*)(x + 64 * (j + 16 * kc));
      for (int t = 0; t < 8; ++t) s += (double)(sa->a[t] * sb->b[t]);     // (a product of two bf16 values is exact in f32)
    }
    d[r] = (float)((double)c[r] + s);
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_16x16x32_bf16((a), (b), (c))
// exact-f32 tile: A lane l = A[l % 16][l / 16], B lane l = B[l / 16][l % 16]
ES_EMU_WAVEOP static emu_f32x4 emu_mfma_16x16x4_f32(float a, float b, emu_f32x4 c) {
  struct Slot { float a, b; } mine = {a, b};
  emu::count_mfma();
  const unsigned char* x = emu::wave_exchange(__builtin_return_address(0), &mine, (int)sizeof(Slot)).slots;
  const int l = emu::lane(), j = l & 15, i0 = 4 * (l >> 4);
  emu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    double s = 0;
    for (int k = 0; k < 4; ++k)
      s += (double)((const Slot*)(x + 64 * (i0 + r + 16 * k)))->a * (double)((const Slot*)(x + 64 * (j + 16 * k)))->b;
    d[r] = (float)((double)c[r] + s);
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_16x16x4_f32((a), (b), (c))

// LDS-DMA: the LDS operand is the wave-uniform base, lane l lands at base + size * l
template <class G, class L>
static inline void emu_global_load_lds(G g, L lds, int size, int offset, int) {
  emu::dma_issue((unsigned char*)(uintptr_t)lds + offset + (size_t)size * emu::lane(), (const void*)(uintptr_t)g, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_global_load_lds((g), (l), (size), (off), (aux))

// ds_read_b64_tr_b16 (lane mapping CONFIRMED on MI355X, profiles/r4a_tr_read.txt): within a 16-lane group, lane i supplies the
// address of the 8 bytes at block row (i >> 2), block columns 4 (i & 3) ..; it receives column i of rows 0 .. 3
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));
template <class P>
ES_EMU_WAVEOP emu_s16x4 emu_ds_read_tr16_b64(P p) {
  const uintptr_t mine = (uintptr_t)p;
  const unsigned char* x = emu::wave_exchange(__builtin_return_address(0), &mine, (int)sizeof(mine)).slots;
  const int l = emu::lane(), g = l & ~15, i = l & 15;
  emu_s16x4 r;
  for (int row = 0; row < 4; ++row) {
    uintptr_t a;
    memcpy(&a, x + 64 * (g + 4 * row + (i >> 2)), sizeof(a));
    r[row] = ((const short*)a)[i & 3];
  }
  return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16_b64((p))
