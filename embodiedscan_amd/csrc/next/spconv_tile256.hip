// NOT PART OF libes_hip.so (not in the Makefile's SRCS): a kernel for the NEXT round, written at the end of round 4 when the GPU
// budget was spent.  Its logic is verified under the CDNA emulator of tests/emu (tests/test_emu_next.py: bit-identical to the
// shipped k_spconv_bf16_dma on the same inputs, under both thread schedules and with late LDS-DMA delivery); it compiles for
// gfx950 within the register / LDS budget (same test).  IT HAS NEVER RUN ON A GPU: first action next round = build it into the
// library, run its parity test on the MI355X, then measure.
//
// What: the LDS-DMA gather-convolution kernel of spconv.hip (k_spconv_bf16_dma) with a 256-row workgroup tile.
// Why (DESIGN.md "Next round" 1): per wave and 32-channel chunk the 128 x 128 tile issues 4 LDS-DMA pieces (2 of A, 2 of B) for
// 16 MFMAs; the cycle table of MI355X_MICROARCH.md prices a piece at 60-185 issue cycles beside ~320 cycles of MFMA -- the
// staging instructions, not HBM bandwidth or latency, hold the kernel at 25-29 % MFMA busy (the three-buffer ring changed
// nothing).  Pieces per MFMA of an M x N tile = 16 (M + N) / (M N): 0.25 at 128 x 128, 0.1875 at 256 x 128 (this kernel, -25 %),
// 0.125 at 256 x 256.  The weight tile (B) is staged once for twice as many rows; the waves stay at 64 x (BNT / 2) each
// (8 fragment reads per 16 MFMAs, as before), there are just 8 of them (512 threads), tiled 4 x 2 over the 256 x BNT output.
// LDS: A 2 x 256 x RB + B 2 x BNT x RB + the 256 x 27 map tile = 123.6 KB at 64-channel chunks (one workgroup = 8 waves per
// CU), 75.6 KB at 32-channel chunks (two workgroups per CU).
// Per output element the (tap, chunk, k) accumulation order is that of k_spconv_bf16_dma: results are bit-identical.
// Tap split (gridDim.z > 1) is not supported: 256-row tiles are for launches with thousands of row tiles.
#include "common.h"       // (compile with -I embodiedscan_amd/csrc)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define T256_MAXK 27
#define T256_IO_Y16 1
#define T256_IO_R16 2
__device__ __attribute__((aligned(16))) unsigned short g_zero_granule_t256[8];

__device__ inline uint32_t t256_pack_bf16(float a, float b) {
  f32x2_t x = {a, b};
  bf16x2_t y = __builtin_convertvector(x, bf16x2_t);
  return *(uint32_t*)&y;
}

// BMT rows x BNT output channels per workgroup of BMT / 32 waves; KB * 32 channels per chunk.
template <int BMT, int BNT, int KB>
__global__ __launch_bounds__(BMT * 2, (BMT == 128 && KB == 1) ? 3 : 2) void k_spconv_bf16_dma_t(
    const unsigned short* __restrict__ Xh, int ldx, const unsigned short* __restrict__ W, const int* __restrict__ nbr,
    int n_out, int n_in, int K, int Cin, int Cout, const float* __restrict__ bias, float* __restrict__ Y, int ldy,
    int accumulate, const float* __restrict__ ep_scale, const float* __restrict__ ep_shift,
    const float* __restrict__ ep_res, int ep_ldr, int ep_act, int io) {
  constexpr int NW = BMT / 32;                   // waves (each 64 rows x BNT / 2 columns; 2 wave columns)
  constexpr int NT = NW * 64;                    // threads = 2 * BMT
  constexpr int G = 4 * KB;                      // 16-byte granules per tile row
  constexpr int RB = 64 * KB;                    // bytes per tile row
  constexpr int BKT = 32 * KB;                   // channels per chunk
  constexpr int A_BYTES = BMT * RB, B_BYTES = BNT * RB;
  constexpr int NA = BMT * G / NT, NBI = BNT * G / NT;     // DMA instructions per thread and chunk (A, B)
  static_assert(BMT * G % NT == 0 && BNT * G % NT == 0 && NBI >= 1, "every thread issues whole pieces");
  constexpr int NFW = BNT / 32;                  // 16-wide column fragments per wave
  constexpr int OFF_B = 2 * A_BYTES, OFF_MAP = OFF_B + 2 * B_BYTES, OFF_TAPS = OFF_MAP + BMT * T256_MAXK * 4,
                OFF_FLAG = OFF_TAPS + 32 * 4, OFF_NT = OFF_FLAG + 32 * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_NT + 16];
  int* const nbrS = (int*)(smem + OFF_MAP);
  int* const taps = (int*)(smem + OFF_TAPS);
  int* const tapFlag = (int*)(smem + OFF_FLAG);
  int* const nTapsP = (int*)(smem + OFF_NT);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int row0 = blockIdx.x * BMT, n0 = blockIdx.y * BNT;

  if (t < 32) tapFlag[t] = 0;
  __syncthreads();
  {                                   // kernel-map tile -> LDS: two threads per row, half of the taps each
    int r = t >> 1, kh = (K + 1) >> 1, k0 = (t & 1) * kh, k1 = min(K, k0 + kh);
    int j = row0 + r;
    const int* src = nbr ? nbr + (size_t)j * K : nullptr;
    for (int k = k0; k < k1; ++k) {
      int v = -1;
      if (j < n_out) v = src ? src[k] : (j < n_in ? j : -1);
      nbrS[r * K + k] = v;
      if (v >= 0) tapFlag[k] = 1;
    }
  }
  __syncthreads();
  if (t < 64) {                       // compact the taps any row of the tile uses
    int f = (t < K) ? tapFlag[t] : 0;
    unsigned long long m = __ballot(f);
    if (f) taps[__popcll(m & ((1ull << t) - 1ull))] = t;
    if (t == 0) *nTapsP = __popcll(m);
  }
  __syncthreads();
  const int nT = *nTapsP;

  f32x4 acc[4][NFW];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NFW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // DMA piece e = (j * NW + wv) * 64 + lane of a tile lands at byte e * 16: tile row e / G, slot e % G, and holds the row's
  // granule slot ^ key(row) (the swizzle of k_spconv_bf16_dma: conflict-free fragment reads, tools/lds_conflicts.py)
  auto key = [](int row) { return KB == 1 ? (((row >> 2) * 3) & 3) : ((row >> 1) & 7); };
  int a_row[NA], a_g8[NA], b_off[NBI];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int e = (j * NW + wv) * 64 + lane, row = e / G, slot = e % G;
    a_row[j] = row;
    a_g8[j] = (slot ^ key(row)) * 8;
  }
#pragma unroll
  for (int j = 0; j < NBI; ++j) {
    int e = (j * NW + wv) * 64 + lane, row = e / G, slot = e % G;
    b_off[j] = (n0 + row) * Cin + (slot ^ key(row)) * 8;
  }
  const int w_tap = Cout * Cin;
  int it_ti = 0, it_c0 = 0, tap_off = 0;
  int a_off[NA];                                  // element offset of this lane's granule in X (< 0: absent neighbour)
  auto set_tap = [&]() {
    int k = taps[it_ti < nT ? it_ti : (nT - 1)];
    tap_off = k * w_tap;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int idx = nbrS[a_row[j] * K + k];
      a_off[j] = idx >= 0 ? idx * ldx + a_g8[j] : -1;
    }
  };
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned short* p = a_off[j] >= 0 ? (Xh + a_off[j] + it_c0) : g_zero_granule_t256;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + buf * A_BYTES + (j * NW + wv) * 1024),
                                       16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NBI; ++j) {
      const unsigned short* p = W + tap_off + b_off[j] + it_c0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + OFF_B + buf * B_BYTES + (j * NW + wv) * 1024),
                                       16, 0, 0);
    }
    it_c0 += BKT;                                 // next chunk of the stream
    if (it_c0 >= Cin) {
      it_c0 = 0;
      ++it_ti;
      set_tap();
    }
  };
  const int li = lane & 15, kq = lane >> 4;
  const int f_key = key(li);                      // tile rows of a fragment are 16 * x + li: the key depends on li only
  const unsigned char* a_frag = smem + (wr * 64 + li) * RB;
  const unsigned char* b_frag = smem + OFF_B + (wc * (BNT / 2) + li) * RB;
  auto compute = [&](int buf) {
#pragma unroll
    for (int h = 0; h < KB; ++h) {
      const int so = ((h * 4 + kq) ^ f_key) * 16;
      bf16x8_t a[4], b[NFW];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) a[mf] = *(const bf16x8_t*)(a_frag + buf * A_BYTES + mf * 16 * RB + so);
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) b[nf] = *(const bf16x8_t*)(b_frag + buf * B_BYTES + nf * 16 * RB + so);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
    }
  };

  if (nT > 0) {
    const int nch = nT * (Cin / BKT);
    set_tap();
    issue(0);
    for (int c = 0; c < nch; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of chunk c have landed ...
      __syncthreads();                                      // ... everybody's have, and chunk c - 1 has been consumed
      if (c + 1 < nch) issue((c + 1) & 1);
      compute(c & 1);
    }
  }
  // epilogue of k_spconv_bf16_dma (bias, folded norm, residual / gate, ReLU, f32 or bf16 rows), for the NW / 2 x 2 wave tiling
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      int col = n0 + wc * (BNT / 2) + nf * 16 + li;
      float bv = bias ? bias[col] : 0.f;
      float sc = ep_scale ? ep_scale[col] : 1.f, sh = ep_shift ? ep_shift[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + wr * 64 + mf * 16 + kq * 4 + r;
        if (row < n_out) {
          float* p = Y + (size_t)row * ldy + col;
          float v = acc[mf][nf][r] + bv;
          if (ep_scale) v = v * sc + sh;
          float rv = 0.f;
          if (ep_res) {
            if (io & T256_IO_R16) rv = __uint_as_float((uint32_t)((const unsigned short*)ep_res)[(size_t)row * ep_ldr + col] << 16);
            else rv = ep_res[(size_t)row * ep_ldr + col];
          }
          if (ep_act == 3) {
            if (!(rv > 0.f)) v = 0.f;
          } else {
            if (ep_res) v += rv;
            if (ep_act) v = fmaxf(v, 0.f);
          }
          if (io & T256_IO_Y16) {
            float vn = __shfl_xor(v, 1, 64);
            if (!(li & 1)) *(uint32_t*)((unsigned short*)Y + (size_t)row * ldy + col) = t256_pack_bf16(v, vn);
          } else {
            *p = accumulate ? (*p + v) : v;
          }
        }
      }
    }
}

// Same operand contract as the fast path of es_spconv_fwd_bf16_io (bf16 rows Xh with ldx in elements, W = the transposed bf16
// copy [K][Cout][Cin], nbr (n_out, K) or NULL for the identity map).  rows = 256 or 128 (128: the shipped tile through this
// template, for A/B runs); cols = 0 (128 when C_out % 128 == 0, else 64) or 256; chunk = 1 or 2 (32 / 64 channels).  Requires Cin % (32 * chunk) == 0, Cout % 64 == 0, ldx % 8 == 0,
// 16-byte aligned rows; returns -4 for a shape it does not take.
extern "C" int es_next_spconv_fwd_bf16_tile(const void* Xh_, int ldx, const void* W_bf16, const int* nbr, int n_out, int n_in, int K,
                                            int Cin, int Cout, const float* bias, float* Y, int ldy, int accumulate,
                                            const float* ep_scale, const float* ep_shift, const float* ep_res, int ep_ldr, int ep_act,
                                            int io, int rows, int cols, int chunk, void* stream) {
  const unsigned short* Xh = (const unsigned short*)Xh_;
  const unsigned short* Wh = (const unsigned short*)W_bf16;
  hipStream_t st = (hipStream_t)stream;
  if (n_out <= 0) return 0;
  if (K > T256_MAXK || (chunk != 1 && chunk != 2) || Cin % (32 * chunk) != 0 || Cout % 64 != 0 || (ldx & 7) != 0 ||
      ((uintptr_t)Xh & 15) != 0 || (rows != 128 && rows != 256))
    return -4;
  const bool wide = Cout % 128 == 0;
  if (!wide && rows == 256 && chunk == 1) return -4;             // 64 columns x 4 granules < 512 threads: no whole pieces
#define T256_LAUNCH(BMT_, BNT_, KB_)                                                                                          \
  hipLaunchKernelGGL((k_spconv_bf16_dma_t<BMT_, BNT_, KB_>), dim3(es_cdiv(n_out, BMT_), Cout / BNT_), dim3(BMT_ * 2), 0, st, Xh, ldx, \
                     Wh, nbr, n_out, n_in, K, Cin, Cout, bias, Y, ldy, accumulate, ep_scale, ep_shift, ep_res, ep_ldr, ep_act, io)
  if (rows == 256 && cols == 256) {                              // the dense occupancy neck (768 / 1536 / 3072 channels): 64 x 128 wave
    if (Cout % 256 != 0) return -4;                               // tiles, 12 fragment reads per 32 MFMAs, 0.125 pieces per MFMA;
    if (chunk == 2) T256_LAUNCH(256, 256, 2); else T256_LAUNCH(256, 256, 1);   // 158.7 KB of LDS at 64-channel chunks
  } else if (rows == 256) {
    if (wide) { if (chunk == 2) T256_LAUNCH(256, 128, 2); else T256_LAUNCH(256, 128, 1); }
    else T256_LAUNCH(256, 64, 2);
  } else {
    if (wide) { if (chunk == 2) T256_LAUNCH(128, 128, 2); else T256_LAUNCH(128, 128, 1); }
    else { if (chunk == 2) T256_LAUNCH(128, 64, 2); else T256_LAUNCH(128, 64, 1); }
  }
#undef T256_LAUNCH
  ES_CHECK_LAUNCH();
  return 0;
}
