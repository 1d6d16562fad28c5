// dev probe (next round): what does ds_read_b64_tr_b16 return?  LDS is filled with element indices (a 64 x 64 bf16-sized grid:
// value = row * 64 + col); every lane passes the address of row (lane & 15) .. as the weight-gradient kernel would, and the
// program prints, per lane, the four 16-bit values it received -- i.e. which (row, col) each output element came from.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read.hip -o /tmp/tr_read && /tmp/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int row_stride_elems, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
  // mode 0: lane i of a 16-lane group points at row (i >> 2), columns 4 * (i & 3) .. of a [k][16] block (the layout the guide's
  //         formula implies); group kq takes k-rows 4 kq ..;  mode 1: lane li points at row li, columns 4 kq ..
  const unsigned short* p = mode == 0 ? &lds[(4 * kq + (li >> 2)) * row_stride_elems + 4 * (li & 3)]
                                      : &lds[li * row_stride_elems + 4 * kq];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d;
  hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 64, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (LDS value = row * 64 + col)\n", mode);
    for (int lane = 0; lane < 64; ++lane) {
      printf("lane %2d:", lane);
      for (int j = 0; j < 4; ++j) printf("  (r%2d,c%2d)", h[lane * 4 + j] / 64, h[lane * 4 + j] % 64);
      printf("\n");
    }
  }
  return 0;
}
