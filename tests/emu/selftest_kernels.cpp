// TEST INFRASTRUCTURE ONLY: two deliberately breakable kernels that show the emulator DETECTS what it claims to detect
// (tests/test_emu_kernels.py::test_the_emulator_catches_a_missing_wait_and_a_missing_barrier).
#include <hip/hip_runtime.h>

// every wave stages 1 KiB through LDS with one LDS-DMA piece per lane, then copies it out.  wait = 0 omits the s_waitcnt
// between the DMA and the barrier: legal-looking code that reads the tile before it has landed.
__global__ void k_selftest_dma(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int wait) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[4 * 1024];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  for (int i = t; i < 4 * 1024; i += blockDim.x) tile[i] = 0xEE;
  __syncthreads();
  __builtin_amdgcn_global_load_lds(src + (size_t)blockIdx.x * 4096 + wv * 1024 + lane * 16, tile + wv * 1024, 16, 0, 0);
  if (wait) ES_EMU_WAITCNT_VM(0);
  __syncthreads();
  for (int i = t; i < 4 * 1024; i += blockDim.x) dst[(size_t)blockIdx.x * 4096 + i] = tile[i];
}
extern "C" int es_emu_selftest_dma(const unsigned char* src, unsigned char* dst, int blocks, int wait) {
  hipLaunchKernelGGL(k_selftest_dma, dim3(blocks), dim3(256), 0, 0, src, dst, wait);
  return 0;
}

// neighbour exchange through LDS; barrier = 0 omits the __syncthreads between the write and the read
__global__ void k_selftest_barrier(const int* __restrict__ src, int* __restrict__ dst, int barrier) {
  __shared__ int buf[256];
  const int t = threadIdx.x;
  buf[t] = -1;
  __syncthreads();
  buf[t] = src[blockIdx.x * 256 + t];
  if (barrier) __syncthreads();
  dst[blockIdx.x * 256 + t] = buf[(t + 1) & 255] + buf[(t + 255) & 255];
}
extern "C" int es_emu_selftest_barrier(const int* src, int* dst, int blocks, int barrier) {
  hipLaunchKernelGGL(k_selftest_barrier, dim3(blocks), dim3(256), 0, 0, src, dst, barrier);
  return 0;
}
