// Dense-volume convolution engine of the occupancy neck (BASELINE config 5; round 5).
//
// nn.Conv3d(k = 3, stride 1 / 2, pad 1) of IndoorImVoxelNeck (embodiedscan/models/necks/imvoxel_neck.py:78-143: ResModule.conv1 /
// conv2, the 3x3x3 convolutions of _make_block / _make_up_block) on a channels-last (B*X*Y*Z, C) bf16 row matrix, as an
// implicit GEMM by ADDRESS ARITHMETIC: the neighbour of an output voxel under a tap is row + constant, guarded by three bounds
// checks -- no neighbour map is built, read or staged (the sparse engine of spconv.hip spends 27 KB of LDS and one index load
// per row and tap on it), and the tile is sized for the one MFMA-bound block of the suite instead of for gathers:
//
//   * forward / data gradient (k_dconv): C[M x N] = sum_t A_t[M x Kd] . W_t[N x Kd]^T.  Workgroup tile 256 x 256 or 320 x 256
//     (M = 25 600 rows = 80 tiles of 320 -> 240 workgroups = ONE round on 256 CUs; 256-row tiles would need two rounds at 59 %),
//     8 waves as 4 x 2 (wave tile 64|80 x 128: 32|40 accumulator fragments), 64 reduction channels per step, both operands
//     staged by global_load_lds_dwordx4 into two LDS buffers (128 | 144 KB, one workgroup per CU) with the swizzle on the
//     SOURCE address (conflict-free ds_read_b128 fragments: the layout of k_spconv_bf16_dma<*, 2>), one barrier per step.
//     Per wave and step 24|26 KB of fragment reads for 64|80 MFMAs (1 024|1 280 matrix-pipe cycles): LDS traffic is 40 % of the
//     matrix time, the 128 x 128 tile of spconv.hip sat at 100 %.
//     Loop order: tap OUTER, channel chunk INNER (the other order -- the 27 taps of one chunk re-reading the same rows from L2 --
//     measured 5 - 20 % slower: the per-step address recomputation costs more than the locality buys; option 21); workgroups are
//     numbered so that one XCD holds consecutive row tiles of ONE column tile (shared weight tiles, overlapping halos).
//     Under-filled launches (the 3 200- and 400-voxel levels) split the linear (chunk, tap) sequence over `nsplit` workgroups
//     per tile; partial tiles go to a workspace and are added in slice order by k_dconv_reduce (bit-reproducible).
//     The data gradient of a stride-1 convolution is the same kernel on the output gradient with the natural-layout weights and
//     the tap list mirrored (host side: dc_geometry).
//   * weight gradient (k_dconv_wgrad): dW_t[Cin x Cout] = sum_j X[src(j, t)]^T . dY[j], one 256 x 256 tile of one tap per
//     workgroup (27 x 3 x 3 = 243 workgroups at 768 channels), 64 voxels per step.  Both operands stay in their natural
//     [voxel][channel] layout in LDS (global_load_lds again) and are read TRANSPOSED by ds_read_b64_tr_b16 (lane mapping
//     probed in round 4, profiles/r4a_tr_read.txt), so no operand is ever transposed in registers or in HBM.  The 27 taps of a
//     (Cin tile, Cout tile) pair run on one XCD: they stream the same dY rows and X rows shifted by at most one x-slab.
//
// Arithmetic: bf16 operands, f32 accumulation on v_mfma_f32_16x16x32_bf16; the order of the additions differs from the sparse
// engine's (chunk-outer instead of tap-outer), the operands and the precision do not.
#include "common.h"
#include "../../include/es_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

#define DC_MAXT 27
struct DcTap { short w, dx, dy, dz; };    // weight tap index; source voxel = row voxel * sm + (dx, dy, dz)
struct DcGeom {
  int rX, rY, rZ;                         // row grid per sample (the GEMM's M axis is (b, x, y, z) over it)
  unsigned mS, mYZ, mZ;                   // floor(2^32 / d) for d = rX*rY*rZ, rY*rZ, rZ (dc_div)
  int sX, sY, sZ, sm;                     // source grid per sample, row -> source coordinate multiplier (the stride)
  int nT;
  // cls = 1 (the data gradient of a stride-2 convolution, the forward of a k = 2 / s = 2 transposed convolution): the rows are
  // (parity class p = (px, py, pz), b, x, y, z) -- every row-grid voxel once per class -- row (p, b, x, y, z) is written to the
  // voxel (2x + px, 2y + py, 2z + pz) of the output grid (2 rX, 2 rY, 2 rZ), and class p only has the taps
  // taps[clsBeg[p] .. clsBeg[p + 1]) (a stride-2 3x3x3 kernel reaches an output voxel through 1, 2, 4 or 8 of its 27 taps,
  // depending on the parity of its coordinates: no zero tap is ever multiplied).  A row tile never straddles two classes.
  int cls;
  short clsBeg[9];
  DcTap taps[DC_MAXT];
};

__device__ __attribute__((aligned(16))) unsigned short g_dc_zero[8];      // what an absent neighbour's LDS-DMA piece reads

// n / d for any n < 2^32 with mul = floor(2^32 / d) (d = 1: 0xffffffff): the estimate is at most one low
__device__ __forceinline__ void dc_div(unsigned n, unsigned d, unsigned mul, unsigned& q, unsigned& r) {
  q = __umulhi(n, mul);
  r = n - q * d;
  if (r >= d) { r -= d; ++q; }
}
// packed row coordinates (x * sm | y * sm << 11 | z * sm << 22) and first source row of the sample; < 0: row past the end
__device__ __forceinline__ void dc_row(const DcGeom& g, int m, int M, int& pk, int& base) {
  const bool ok = m < M;                         // (branch-free: rows past the end decompose row 0 and are masked)
  unsigned b, rem, x, y, z;
  dc_div(ok ? (unsigned)m : 0u, (unsigned)(g.rX * g.rY * g.rZ), g.mS, b, rem);
  dc_div(rem, (unsigned)(g.rY * g.rZ), g.mYZ, x, rem);
  dc_div(rem, (unsigned)g.rZ, g.mZ, y, z);
  pk = ok ? (int)((x * g.sm) | ((y * g.sm) << 11) | ((z * g.sm) << 22)) : -1;
  base = (int)b * g.sX * g.sY * g.sZ;
}
// source row of packed coordinates under a tap, or -1
__device__ __forceinline__ int dc_src(const DcGeom& g, int pk, int base, int dx, int dy, int dz) {
  const int x = (pk & 0x7ff) + dx, y = ((pk >> 11) & 0x7ff) + dy, z = (pk >> 22) + dz;
  const bool ok = pk >= 0 && (unsigned)x < (unsigned)g.sX && (unsigned)y < (unsigned)g.sY && (unsigned)z < (unsigned)g.sZ;
  return ok ? base + (x * g.sY + y) * g.sZ + z : -1;
}
// logical workgroup of a 1-D grid of 8 * per blocks: XCD x (= blockIdx.x % 8: MI355X_MICROARCH.md, observed, used for speed
// only) runs the logical ids [x * per, (x + 1) * per) in dispatch order
__device__ __forceinline__ int dc_logical(int per) { return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3); }

// ------------------------------------------------------------------ forward / data gradient
// Xh (rows x ldx) bf16 source rows; W [tap][N][Kd] bf16 (Kd contiguous); Y (M x ldy) f32, or partial tiles ws[slice][M][N].
template <int BMT, bool TAP_INNER, int BNT = 256>
__global__ __launch_bounds__(512, 2) void k_dconv(const unsigned short* __restrict__ Xh, int ldx,
                                                  const unsigned short* __restrict__ W, int Kd, int N, int M, DcGeom g,
                                                  float* __restrict__ Y, int ldy, int accumulate, float* __restrict__ ws,
                                                  int nsplit, int rowTiles, int colTiles, int per) {
  constexpr int RB = 128;                        // bytes per tile row: 64 channels
  constexpr int A_BYTES = BMT * RB, B_BYTES = BNT * RB;
  constexpr int NA = BMT / 64, NB = BNT / 64;    // LDS-DMA pieces per thread and step (A, B): 8 granules per row / 512 threads
  constexpr int MF = BMT / 64, NF = BNT / 32;    // 16 x 16 fragments per wave: (BMT / 4) rows x BNT / 2 columns (BNT = 128: the neck's out blocks)
  constexpr int OFF_B = 2 * A_BYTES, OFF_TAP = OFF_B + 2 * B_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_TAP + 32 * 8];
  DcTap* const tapS = (DcTap*)(smem + OFF_TAP);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int L = dc_logical(per);
  if (L >= rowTiles * colTiles * nsplit) return;
  const int rt = L % rowTiles, ct = (L / rowTiles) % colTiles, bz = L / (rowTiles * colTiles);
  const int tpc = g.cls ? rowTiles >> 3 : rowTiles;             // row tiles per parity class
  const int pc = rt / tpc;                                      // this tile's class (0 without classes)
  const int row0 = (rt - pc * tpc) * BMT, n0 = ct * BNT;
  const int tb = g.cls ? g.clsBeg[pc] : 0, nTl = g.cls ? g.clsBeg[pc + 1] - tb : g.nT;     // this tile's taps
  if (t < nTl) tapS[t] = g.taps[tb + t];
  // this thread's A pieces: piece e = (j * 8 + wv) * 64 + lane lands at LDS byte e * 16 = tile row e >> 3, slot e & 7, and holds
  // the row's granule slot ^ key(row), key(row) = (row >> 1) & 7 (the same for all j: rows differ by multiples of 64)
  int a_pk[NA], a_base[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) dc_row(g, row0 + (j * 8 + wv) * 8 + (lane >> 3), M, a_pk[j], a_base[j]);
  const int kslot = ((lane & 7) ^ (((wv & 1) * 4 + (lane >> 4)) & 7)) * 8;      // element offset of the source granule
  const int b_off0 = (n0 + wv * 8 + (lane >> 3)) * Kd + kslot;                  // piece j: + j * 64 * Kd
  const int nC = Kd >> 6, nIt = nC * nTl;
  const int it0 = (int)(((long long)nIt * bz) / nsplit), it1 = (int)(((long long)nIt * (bz + 1)) / nsplit);
  int it_t, it_c0;                               // tap index and channel offset of the NEXT step to stage
  if (TAP_INNER) { it_t = it0 % nTl; it_c0 = (it0 / nTl) * 64; }
  else { it_c0 = (it0 % nC) * 64; it_t = it0 / nC; }
  __syncthreads();                               // the tap table

  f32x4 acc[MF][NF];
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int a_off[NA], w_off = 0;
  auto set_tap = [&]() {
    const DcTap tp = tapS[it_t < nTl ? it_t : 0];
    w_off = (int)tp.w * N * Kd;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int s = dc_src(g, a_pk[j], a_base[j], tp.dx, tp.dy, tp.dz);
      a_off[j] = s >= 0 ? s * ldx + kslot : -1;
    }
  };
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned short* p = a_off[j] >= 0 ? (Xh + a_off[j] + it_c0) : g_dc_zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + buf * A_BYTES + (j * 8 + wv) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const unsigned short* p = W + w_off + b_off0 + j * 64 * Kd + it_c0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + OFF_B + buf * B_BYTES + (j * 8 + wv) * 1024), 16, 0, 0);
    }
    if (TAP_INNER) {                             // next step of the sequence
      if (++it_t == nTl) { it_t = 0; it_c0 += 64; }
      set_tap();
    } else {
      it_c0 += 64;
      if (it_c0 >= Kd) { it_c0 = 0; ++it_t; set_tap(); }
    }
  };
  const int li = lane & 15, kq = lane >> 4;
  const int f_key = (li >> 1) & 7;               // tile rows of a fragment are 16 * x + li: the key depends on li only
  const unsigned char* a_frag = smem + (wr * (BMT / 4) + li) * RB;
  const unsigned char* b_frag = smem + OFF_B + (wc * (BNT / 2) + li) * RB;
  auto compute = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int so = ((h * 4 + kq) ^ f_key) * 16;
      bf16x8_t a[MF], b[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) b[nf] = *(const bf16x8_t*)(b_frag + buf * B_BYTES + nf * 16 * RB + so);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) a[mf] = *(const bf16x8_t*)(a_frag + buf * A_BYTES + mf * 16 * RB + so);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
    }
  };

  const int n = it1 - it0;
  if (n > 0) {
    set_tap();
    issue(0);
    for (int c = 0; c < n; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of step c have landed ...
      __syncthreads();                                      // ... everybody's have, and step c - 1 has been consumed
      if (c + 1 < n) issue((c + 1) & 1);
      compute(c & 1);
    }
  }
  // partial tiles of a split launch: ws[slice][row][N]; with parity classes the rows are class-major (class p, row): 8 M per slice
  float* const out = nsplit > 1 ? ws + ((size_t)bz * (g.cls ? 8 : 1) + (g.cls ? pc : 0)) * M * N : Y;
  const int ldo = nsplit > 1 ? N : ldy;
  const bool add = nsplit > 1 ? false : (accumulate != 0);
  if (g.cls && nsplit == 1) {                    // rows of a parity class: scatter to the voxels (2x + px, 2y + py, 2z + pz)
    const int px = pc >> 2, py = (pc >> 1) & 1, pz = pc & 1;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wr * (BMT / 4) + mf * 16 + kq * 4 + r;
        if (row < M) {
          unsigned b, rem, x, y, z;
          dc_div((unsigned)row, (unsigned)(g.rX * g.rY * g.rZ), g.mS, b, rem);
          dc_div(rem, (unsigned)(g.rY * g.rZ), g.mYZ, x, rem);
          dc_div(rem, (unsigned)g.rZ, g.mZ, y, z);
          const size_t orow = (((size_t)b * (2 * g.rX) + 2 * x + px) * (2 * g.rY) + 2 * y + py) * (2 * g.rZ) + 2 * z + pz;
          float* p = Y + orow * ldy + n0 + wc * (BNT / 2) + li;
          if (add) {
            float y0[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) y0[nf] = p[nf * 16];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) p[nf * 16] = y0[nf] + acc[mf][nf][r];
          } else {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) p[nf * 16] = acc[mf][nf][r];
          }
        }
      }
    return;
  }
  if (add) {                                     // (hoisted: one uniform branch, not one per element)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wr * (BMT / 4) + mf * 16 + kq * 4 + r;
        if (row < M) {
          float* p = out + (size_t)row * ldo + n0 + wc * (BNT / 2) + li;
          float y0[NF];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) y0[nf] = p[nf * 16];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) p[nf * 16] = y0[nf] + acc[mf][nf][r];
        }
      }
  } else {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wr * (BMT / 4) + mf * 16 + kq * 4 + r;
        if (row < M) {
          float* p = out + (size_t)row * ldo + n0 + wc * (BNT / 2) + li;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) p[nf * 16] = acc[mf][nf][r];
        }
      }
  }
}

// Y (+)= sum of the slices' partial tiles in slice order
__global__ void k_dconv_reduce(const float4* __restrict__ ws, int nsplit, size_t tot4, int N4, float* __restrict__ Y, int ldy,
                               int accumulate) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot4; e += (size_t)gridDim.x * blockDim.x) {
    float4 s = ws[e];
    for (int z = 1; z < nsplit; ++z) {
      const float4 v = ws[(size_t)z * tot4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const size_t row = e / N4;
    float4* p = (float4*)(Y + row * ldy + (e - row * N4) * 4);
    if (accumulate) { const float4 y0 = *p; s.x = y0.x + s.x; s.y = y0.y + s.y; s.z = y0.z + s.z; s.w = y0.w + s.w; }
    *p = s;
  }
}

// the same for a split parity-class launch: partial row (class p, m) of every slice -> voxel 2 r(m) + p of the output grid
__global__ void k_dconv_reduce_cls(const float4* __restrict__ ws, int nsplit, int M, int N4, DcGeom g, float* __restrict__ Y, int ldy,
                                   int accumulate) {
  const size_t tot4 = (size_t)8 * M * N4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot4; e += (size_t)gridDim.x * blockDim.x) {
    float4 s = ws[e];
    for (int z = 1; z < nsplit; ++z) {
      const float4 v = ws[(size_t)z * tot4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const size_t rc = e / N4;
    const int pc = (int)(rc / M), m = (int)(rc - (size_t)pc * M);
    unsigned b, rem, x, y, z;
    dc_div((unsigned)m, (unsigned)(g.rX * g.rY * g.rZ), g.mS, b, rem);
    dc_div(rem, (unsigned)(g.rY * g.rZ), g.mYZ, x, rem);
    dc_div(rem, (unsigned)g.rZ, g.mZ, y, z);
    const size_t orow = (((size_t)b * (2 * g.rX) + 2 * x + (pc >> 2)) * (2 * g.rY) + 2 * y + ((pc >> 1) & 1)) * (2 * g.rZ) + 2 * z + (pc & 1);
    float4* p = (float4*)(Y + orow * ldy + (e - rc * N4) * 4);
    if (accumulate) { const float4 y0 = *p; s.x = y0.x + s.x; s.y = y0.y + s.y; s.z = y0.z + s.z; s.w = y0.w + s.w; }
    *p = s;
  }
}

// ------------------------------------------------------------------ weight gradient
// dW[w][ci][co] (+)= sum over rows j of X[src(j, tap)][ci] * dY[j][co]; one (tap, 256 ci, 256 co) tile per workgroup.
__global__ __launch_bounds__(512, 2) void k_dconv_wgrad(const unsigned short* __restrict__ Xh, int ldx,
                                                        const unsigned short* __restrict__ dYh, int ldy, int Cin, int Cout,
                                                        int M, DcGeom g, float* __restrict__ dW, int accumulate, int nCo,
                                                        int nWG, int per, int gatherB, int msplit, float* __restrict__ ws) {
  constexpr int TB = 64 * 512;                   // bytes of one operand tile: 64 voxels x 256 channels
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TB];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kq = lane >> 4;
  const int wr = wv >> 1, wc = wv & 1;
  const int L0 = dc_logical(per);
  if (L0 >= nWG * msplit) return;
  // msplit > 1 (a launch of few tiles over many rows: the 2-D 3x3 layers, 9 tiles over 192 000 pixels): slice z of the 64-row steps,
  // partial dW tiles through the workspace ws[z][K][Cin][Cout], added in slice order by k_dconv_reduce
  const int L = L0 % nWG, zs = L0 / nWG;
  const int ti = L % g.nT, pair = L / g.nT;
  const int c0 = (pair / nCo) * 256, n0 = (pair % nCo) * 256;
  const DcTap tp = g.taps[ti];
  // piece e = (j * 8 + wv) * 64 + lane of a tile: voxel p = e >> 5 = j * 16 + wv * 2 + (lane >> 5), slot e & 31 (32 granules per
  // 512-byte row), holding granule slot ^ (key(p) << 1), key(p) = (p & 3) | ((p >> 3) & 1) << 2 (independent of j): the 8 voxel
  // rows a 32-lane transposed-read group touches land on 8 distinct 32-byte bank positions
  const int p0 = wv * 2 + (lane >> 5);
  const int gk = ((lane & 31) ^ (((p0 & 3) | (((p0 >> 3) & 1) << 2)) << 1)) * 8;        // element offset of the source granule
  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int buf, int j0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = j0 + j * 16 + p0;
      int pk, base;
      dc_row(g, m, M, pk, base);
      const int s = dc_src(g, pk, base, tp.dx, tp.dy, tp.dz);
      // gatherB = 0 (convolution): X rows are gathered under the tap, dY rows are the rows themselves; 1 (transposed convolution:
      // dW[p] = X^T dY[2 . + p]): the other way round
      const int sx = gatherB ? (m < M ? m : -1) : s, sy = gatherB ? s : (m < M ? m : -1);
      const unsigned short* px = sx >= 0 ? (Xh + sx * ldx + c0 + gk) : g_dc_zero;
      const unsigned short* py = (sx >= 0 && sy >= 0) ? (dYh + sy * ldy + n0 + gk) : g_dc_zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)px,
                                       (__attribute__((address_space(3))) void*)(smem + (buf * 2 + 0) * TB + (j * 8 + wv) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)py,
                                       (__attribute__((address_space(3))) void*)(smem + (buf * 2 + 1) * TB + (j * 8 + wv) * 1024), 16, 0, 0);
    }
  };
  // transposed fragment: channel block cb (16 channels) of a tile, voxels h * 32 + kq * 8 .. + 7, for channel li
  auto frag = [&](const unsigned char* tile, int cb, int h) -> bf16x8_t {
    s16x4_t v[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int pr = h * 32 + kq * 8 + r * 4 + (li >> 2);
      const int key = ((pr & 3) | (((pr >> 3) & 1) << 2)) << 1;
      const int gr = cb * 2 + ((li & 3) >> 1);
      const unsigned char* a = tile + pr * 512 + ((gr ^ key) * 16) + (li & 1) * 8;
      v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a);
    }
    s16x8_t u = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, u);
  };
  const int nst = (M + 63) >> 6;
  const int s0 = (int)(((long long)nst * zs) / msplit), n = (int)(((long long)nst * (zs + 1)) / msplit) - s0;
  if (n > 0) issue(0, s0 * 64);
  for (int c = 0; c < n; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 < n) issue((c + 1) & 1, (s0 + c + 1) * 64);
    const unsigned char* xt = smem + ((c & 1) * 2 + 0) * TB;
    const unsigned char* yt = smem + ((c & 1) * 2 + 1) * TB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8_t a[4], b[8];
#pragma unroll
      for (int nf = 0; nf < 8; ++nf) b[nf] = frag(yt, wc * 8 + nf, h);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) a[mf] = frag(xt, wr * 4 + mf, h);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < 8; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
    }
  }
  float* const dst = (msplit > 1 ? ws + (size_t)zs * g.nT * Cin * Cout : dW) + (size_t)tp.w * Cin * Cout;
  if (accumulate && msplit == 1) {
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = c0 + (wr * 4 + mf) * 16 + kq * 4 + r;
        float* p = dst + (size_t)ci * Cout + n0 + wc * 128 + li;
        float y0[8];
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) y0[nf] = p[nf * 16];
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) p[nf * 16] = y0[nf] + acc[mf][nf][r];
      }
  } else {
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = c0 + (wr * 4 + mf) * 16 + kq * 4 + r;
        float* p = dst + (size_t)ci * Cout + n0 + wc * 128 + li;
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) p[nf * 16] = acc[mf][nf][r];
      }
  }
}


// ------------------------------------------------------------------ host side
static unsigned dc_magic(unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)((1ull << 32) / d); }

// geom_host: {B, X, Y, Z, ksize, stride, pad} of the operator's INPUT grid (nn.Conv3d: its input; nn.ConvTranspose3d: ITS input, the
// coarse grid).  Modes:
//   0  convolution forward            rows = output voxels, sources = input voxels * stride - pad + k
//   1  convolution data gradient      stride 1: rows = input voxels, sources = output voxels + pad - k;
//                                     stride 2 (k = 3, pad = 1 or the 1x1x1 down-sample k = 1, pad = 0; even input sizes): rows = (parity
//                                     class, output-grid voxel), written to the input voxel 2 r + p, class p only through the taps with
//                                     p + pad - k even (source r + (p + pad - k) / 2); k = 1: class 0 alone has a tap, the others write zeros
//   2  convolution weight gradient    rows = output voxels; X gathered as in mode 0, dY direct
//   3  transposed convolution (k = 2, s = 2, pad = 0) forward: rows = (class p, input voxel r), written to 2 r + p, one tap (p) per class
//   4  ... its data gradient          rows = input voxels, sources = output voxels 2 r + p, 8 taps
//   5  ... its weight gradient        rows = input voxels; X direct, dY gathered at 2 r + p
static int dc_geometry(const int* gh, int mode, DcGeom& g, int& M, int& n_src) {
  // Z = 0: a FLAT grid (nn.Conv2d on (B, X, Y) images, rows (b * X + x) * Y + y): one z slab, ks x ks taps, weights [ks*ks][..][..]
  const bool flat = gh[3] == 0;
  const int B = gh[0], X = gh[1], Y = gh[2], Z = flat ? 1 : gh[3], ks = gh[4], st = gh[5], pad = gh[6];
  const int ksz = flat ? 1 : ks, padz = flat ? 0 : pad;           // kernel extent / padding along z
  if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || ks <= 0 || ks * ks * ksz > DC_MAXT || st <= 0 || pad < 0 || mode < 0 || mode > 5) return -2;
  const bool tr = mode >= 3;
  if (tr && (flat || ks != 2 || st != 2 || pad != 0)) return -4;
  const int Xo = tr ? 2 * X : (X + 2 * pad - ks) / st + 1, Yo = tr ? 2 * Y : (Y + 2 * pad - ks) / st + 1,
            Zo = tr ? 2 * Z : (flat ? 1 : (Z + 2 * pad - ks) / st + 1);
  if (Xo <= 0 || Yo <= 0 || Zo <= 0) return -2;
  const bool strided_dgrad = mode == 1 && st != 1;
  if (flat && strided_dgrad) return -4;                           // (the parity classes are written for three axes)
  if (mode == 1 && st == 1 && ks != 2 * pad + 1) return -4;
  if (strided_dgrad && (st != 2 || !((ks == 3 && pad == 1) || (ks == 1 && pad == 0)) || X != 2 * Xo || Y != 2 * Yo || Z != 2 * Zo)) return -4;
  const int bigX = X > Xo ? X : Xo, bigY = Y > Yo ? Y : Yo, bigZ = Z > Zo ? Z : Zo;
  if ((long long)bigX * st >= 2048 || (long long)bigY * st >= 2048 || (long long)bigZ * st >= 512)
    return -4;                                                           // (11 / 11 / 9 bits of the packed row coordinates)
  g.cls = (strided_dgrad || mode == 3) ? 1 : 0;
  // row grid / source grid
  const bool rows_out = mode == 0 || mode == 2 || strided_dgrad;          // rows over the convolution's OUTPUT grid
  const bool rows_in = !rows_out;                                         // ... over the operator's input grid
  g.rX = rows_out ? Xo : X; g.rY = rows_out ? Yo : Y; g.rZ = rows_out ? Zo : Z;
  if (mode == 0 || mode == 2) { g.sX = X; g.sY = Y; g.sZ = Z; g.sm = st; }
  else if (mode == 1) { g.sX = Xo; g.sY = Yo; g.sZ = Zo; g.sm = 1; }
  else if (mode == 3) { g.sX = X; g.sY = Y; g.sZ = Z; g.sm = 1; }
  else { g.sX = Xo; g.sY = Yo; g.sZ = Zo; g.sm = 2; }
  (void)rows_in;
  g.mS = dc_magic((unsigned)(g.rX * g.rY * g.rZ)); g.mYZ = dc_magic((unsigned)(g.rY * g.rZ)); g.mZ = dc_magic((unsigned)g.rZ);
  g.nT = 0;
  for (int p = 0; p < 9; ++p) g.clsBeg[p] = 0;
  if (g.cls) {
    for (int p = 0; p < 8; ++p) {
      g.clsBeg[p] = (short)g.nT;
      const int q[3] = {p >> 2, (p >> 1) & 1, p & 1};
      if (mode == 3) {                                                     // transposed convolution: output 2 r + p comes from input r through tap p
        g.taps[g.nT++] = DcTap{(short)p, 0, 0, 0};
        continue;
      }
      for (int t = 0; t < ks * ks * ks; ++t) {
        const int k[3] = {t / (ks * ks), (t / ks) % ks, t % ks};
        int d[3];
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
          const int e = q[a] + pad - k[a];
          if (e & 1) ok = false;
          d[a] = e / 2;                                                    // (exact when even)
        }
        if (ok) g.taps[g.nT++] = DcTap{(short)t, (short)d[0], (short)d[1], (short)d[2]};
      }
    }
    g.clsBeg[8] = (short)g.nT;
  } else {
    g.nT = ks * ks * ksz;
    for (int t = 0; t < g.nT; ++t) {
      const int kx = t / (ks * ksz), ky = (t / ksz) % ks, kz = t % ksz;
      // forward: source = out * stride - pad + k.  data gradient (stride 1): dX[i] += dY[i + pad - k] . W[k]^T.
      // transposed convolution, data / weight gradient: source = 2 r + k
      const bool back = mode == 1;
      g.taps[t].w = (short)t;
      g.taps[t].dx = (short)(back ? pad - kx : kx - pad);
      g.taps[t].dy = (short)(back ? pad - ky : ky - pad);
      g.taps[t].dz = (short)(back ? padz - kz : kz - padz);
    }
  }
  const long long m = (long long)B * g.rX * g.rY * g.rZ, ns = (long long)B * g.sX * g.sY * g.sZ;
  if (m * (g.cls ? 8 : 1) >= (1ll << 31) || ns >= (1ll << 31)) return -4;
  M = (int)m; n_src = (int)ns;
  return 0;
}

static int ES_OPT_DC_ROWS = 0;          // es_set_option 20: row tile of k_dconv (0 = pick per launch, 256, 320)
static int ES_OPT_DC_ORDER = 0;         // 21: 0 = tap outer / channel chunk inner (default: +5 .. 20 % on every neck shape, profiles/r5a_dconv_ab.txt), 1 = chunk outer / tap inner
static int ES_OPT_DC_SPLIT = 0;         // 22: 0 = pick the slice count per launch, n = force
static int ES_OPT_DC_WSPLIT = 0;        // 23: row slices of the weight gradient with a workspace (0 = pick, n = force)
extern "C" int es_dconv_set_option(int key, int value) {
  if (key == 20) { ES_OPT_DC_ROWS = value; return 0; }
  if (key == 21) { ES_OPT_DC_ORDER = value; return 0; }
  if (key == 22) { ES_OPT_DC_SPLIT = value; return 0; }
  if (key == 23) { ES_OPT_DC_WSPLIT = value; return 0; }
  return -2;
}

struct DcPlan { int bm, rowTiles, colTiles, nsplit; };
static DcPlan dc_plan(int M, int N, int nIt, int cls = 0, int nT = 27) {
  const int bn = (N % 256 == 0) ? 256 : 128;     // column tile
  if (cls) {                                     // parity classes: 8 x tiles-per-class row tiles, no slices (the classes ARE the split)
    int bm = (ES_OPT_DC_ROWS == 256 || ES_OPT_DC_ROWS == 320) ? ES_OPT_DC_ROWS
             : (es_cdiv(M, 320) * 320 < es_cdiv(M, 256) * 256 || (es_cdiv(M, 320) * 320 == es_cdiv(M, 256) * 256 && M >= 320) ? 320 : 256);
    // slices of every class's (tap, chunk) sequence when the launch would leave most CUs idle (the 400-voxel level: 96 workgroups,
    // the 8-tap class walking 384 steps); partial rows class-major through the workspace, k_dconv_reduce_cls scatters them
    const int wgs = 8 * es_cdiv(M, bm) * (N / bn);
    int s = ES_OPT_DC_SPLIT ? ES_OPT_DC_SPLIT : (wgs < 200 ? (es_cdiv(384, wgs) < 8 ? es_cdiv(384, wgs) : 8) : 1);
    if (s > nIt / nT) s = nIt / nT > 0 ? nIt / nT : 1;              // (a one-tap class keeps >= one channel chunk per slice)
    return DcPlan{bm, 8 * es_cdiv(M, bm), N / bn, s};
  }
  DcPlan best{256, es_cdiv(M, 256), N / bn, 1};
  double best_cost = 1e30;
  for (int bm = 256; bm <= 320; bm += 64) {
    if (ES_OPT_DC_ROWS && bm != ES_OPT_DC_ROWS) continue;
    const int rtl = es_cdiv(M, bm), tiles = rtl * (N / bn);
    for (int s = 1; s <= 16 && s <= nIt; ++s) {
      if (ES_OPT_DC_SPLIT && s != ES_OPT_DC_SPLIT) continue;
      // time in units of one 64-row x 256-column x 64-channel step on one CU: rounds x steps per workgroup x rows, plus the
      // partial-tile round trip (s slices written and read: ~ (s + 1) x 1 KB per row of a tile against ~ 0.45 us per step)
      const double rounds = (double)es_cdiv((long long)tiles * s, 256);
      double cost = rounds * es_cdiv(nIt, s) * (bm / 64.0) * (bn == 256 ? 1.0 : 0.6);
      if (s > 1) cost += (double)tiles * (s + 1) * bm / 256.0 * 0.35 * (bn / 256.0);
      if (cost < best_cost) { best_cost = cost; best = DcPlan{bm, rtl, N / bn, s}; }
    }
  }
  return best;
}

// which (Kd, N) a mode runs with: the reduction runs over the channels of the SOURCE rows
static void dc_roles(int mode, int Cin, int Cout, int& Kd, int& N) {
  const bool back = mode == 1 || mode == 4;      // data gradients: source = the output gradient (Cout channels), result has Cin
  Kd = back ? Cout : Cin; N = back ? Cin : Cout;
}

extern "C" int es_dconv_supported(const int* geom_host, int mode, int Cin, int Cout) {
  DcGeom g; int M, ns;
  if (dc_geometry(geom_host, mode, g, M, ns) != 0) return 0;
  if (mode == 2 || mode == 5) return (Cin % 256 == 0 && Cout % 256 == 0) ? 1 : 0;
  int Kd, N;
  dc_roles(mode, Cin, Cout, Kd, N);
  return (Kd % 64 == 0 && N % 128 == 0) ? 1 : 0;
}

extern "C" size_t es_dconv_workspace_floats(const int* geom_host, int mode, int Cin, int Cout) {
  DcGeom g; int M, ns;
  if (mode == 2 || mode == 5 || dc_geometry(geom_host, mode, g, M, ns) != 0) return 0;
  int Kd, N;
  dc_roles(mode, Cin, Cout, Kd, N);
  if (Kd % 64 != 0 || N % 128 != 0) return 0;
  const DcPlan p = dc_plan(M, N, (Kd / 64) * g.nT, g.cls, g.nT);
  return p.nsplit > 1 ? (size_t)p.nsplit * (g.cls ? 8 : 1) * M * N : 0;
}

// modes 0 / 3 (forward): Xh = the operator's bf16 input rows, W_bf16 = the [K][Cout][Cin] copy, Y = its output rows (f32).
// modes 1 / 4 (data gradient): Xh = bf16 rows of the output gradient, W_bf16 = the natural [K][Cin][Cout] copy, Y = the input gradient.
extern "C" int es_dconv_fwd_bf16(const void* Xh, int ldx, const void* W_bf16, const int* geom_host, int mode, int Cin, int Cout,
                                 float* Y, int ldy, int accumulate, float* ws, size_t ws_floats, void* stream) {
  DcGeom g; int M, ns;
  if (mode != 0 && mode != 1 && mode != 3 && mode != 4) return -2;
  int rc = dc_geometry(geom_host, mode, g, M, ns);
  if (rc != 0) return rc;
  int Kd, N;
  dc_roles(mode, Cin, Cout, Kd, N);
  const int nTw = mode >= 3 ? 8 : g.nT;           // taps of the weight tensor
  if (Kd % 64 != 0 || N % 128 != 0 || (ldx & 7) != 0 || (ldy & 3) != 0 || ((uintptr_t)Xh & 15) != 0 || ((uintptr_t)W_bf16 & 15) != 0 ||
      ((uintptr_t)Y & 15) != 0 || (long long)ns * ldx >= (1ll << 31) || (long long)(nTw > 27 ? nTw : 27) * N * Kd >= (1ll << 31))
    return -4;
  DcPlan p = dc_plan(M, N, (Kd / 64) * g.nT, g.cls, g.nT);
  if (p.nsplit > 1 && (ws == nullptr || ws_floats < (size_t)p.nsplit * (g.cls ? 8 : 1) * M * N || ((uintptr_t)ws & 15) != 0)) return -5;
  hipStream_t st = (hipStream_t)stream;
  const int nwg = p.rowTiles * p.colTiles * p.nsplit, per = es_cdiv(nwg, 8);
  const unsigned short* X = (const unsigned short*)Xh;
  const unsigned short* Wh = (const unsigned short*)W_bf16;
#define DC_LAUNCH(BM_, TI_)                                                                                                   \
  hipLaunchKernelGGL((k_dconv<BM_, TI_>), dim3(per * 8), dim3(512), 0, st, X, ldx, Wh, Kd, N, M, g, Y, ldy, accumulate, ws, p.nsplit, \
                     p.rowTiles, p.colTiles, per)
  if (N % 256 != 0) {                            // 128-column tiles (tap-outer order only)
    if (p.bm == 320) hipLaunchKernelGGL((k_dconv<320, false, 128>), dim3(per * 8), dim3(512), 0, st, X, ldx, Wh, Kd, N, M, g, Y, ldy, accumulate, ws,
                                        p.nsplit, p.rowTiles, p.colTiles, per);
    else hipLaunchKernelGGL((k_dconv<256, false, 128>), dim3(per * 8), dim3(512), 0, st, X, ldx, Wh, Kd, N, M, g, Y, ldy, accumulate, ws, p.nsplit,
                            p.rowTiles, p.colTiles, per);
  } else if (p.bm == 320) { if (ES_OPT_DC_ORDER) DC_LAUNCH(320, true); else DC_LAUNCH(320, false); }
  else                    { if (ES_OPT_DC_ORDER) DC_LAUNCH(256, true); else DC_LAUNCH(256, false); }
#undef DC_LAUNCH
  ES_CHECK_LAUNCH();
  if (p.nsplit > 1 && g.cls) {
    const int gr = es_cdiv((long long)8 * M * (N / 4), 256);
    hipLaunchKernelGGL(k_dconv_reduce_cls, dim3(gr > 8192 ? 8192 : gr), dim3(256), 0, st, (const float4*)ws, p.nsplit, M, N / 4, g, Y, ldy,
                       accumulate);
    ES_CHECK_LAUNCH();
  } else if (p.nsplit > 1) {
    const size_t tot4 = (size_t)M * (N / 4);
    const int gr = es_cdiv((long long)tot4, 256);
    hipLaunchKernelGGL(k_dconv_reduce, dim3(gr > 8192 ? 8192 : gr), dim3(256), 0, st, (const float4*)ws, p.nsplit, tot4, N / 4, Y, ldy,
                       accumulate);
    ES_CHECK_LAUNCH();
  }
  return 0;
}

// dW[K][Cin][Cout] (+)= X^T dY over the dense grid; Xh = bf16 rows of the operator's input, dYh = bf16 rows of its output gradient;
// transposed 0: nn.Conv3d (mode 2), 1: nn.ConvTranspose3d(k = 2, s = 2) (mode 5)
// row slices of a weight-gradient launch: only for launches of few tiles (< 64 workgroups; every shape of the neck keeps one pass)
static int dc_wgrad_split(int nwg, int M) {
  if (ES_OPT_DC_WSPLIT > 0) return ES_OPT_DC_WSPLIT;
  if (nwg >= 64) return 1;
  const int nst = (M + 63) >> 6;
  int s = es_cdiv(240, nwg);
  if (s > nst / 16) s = nst / 16;                // (>= 16 steps of 64 rows per slice)
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}
extern "C" size_t es_dconv_wgrad_workspace_floats(const int* geom_host, int transposed, int Cin, int Cout) {
  DcGeom g; int M, ns;
  if (dc_geometry(geom_host, transposed ? 5 : 2, g, M, ns) != 0 || Cin % 256 != 0 || Cout % 256 != 0) return 0;
  const int nwg = g.nT * (Cin / 256) * (Cout / 256), ms = dc_wgrad_split(nwg, M);
  return ms > 1 ? (size_t)ms * g.nT * Cin * Cout : 0;
}
static int dconv_wgrad_impl(const void* Xh, int ldx, const void* dYh, int ldy, const int* geom_host, int transposed, int Cin, int Cout,
                            float* dW, int accumulate, float* ws, size_t ws_floats, int allow_split, void* stream) {
  DcGeom g; int M, ns;
  int rc = dc_geometry(geom_host, transposed ? 5 : 2, g, M, ns);
  if (rc != 0) return rc;
  const long long nx = transposed ? M : ns, ny = transposed ? ns : M;     // rows of X / of dY
  if (Cin % 256 != 0 || Cout % 256 != 0 || (ldx & 7) != 0 || (ldy & 7) != 0 || ((uintptr_t)Xh & 15) != 0 || ((uintptr_t)dYh & 15) != 0 ||
      nx * ldx >= (1ll << 31) || ny * ldy >= (1ll << 31))
    return -4;
  const int nCo = Cout / 256, nwg = g.nT * (Cin / 256) * nCo;
  const int ms = allow_split ? dc_wgrad_split(nwg, M) : 1;
  const size_t tot = (size_t)g.nT * Cin * Cout;
  if (ms > 1 && (ws == nullptr || ws_floats < (size_t)ms * tot || ((uintptr_t)ws & 15) != 0)) return -5;
  const int per = es_cdiv(nwg * ms, 8);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_dconv_wgrad, dim3(per * 8), dim3(512), 0, st, (const unsigned short*)Xh, ldx,
                     (const unsigned short*)dYh, ldy, Cin, Cout, M, g, dW, accumulate, nCo, nwg, per, transposed ? 1 : 0, ms, ws);
  ES_CHECK_LAUNCH();
  if (ms > 1) {
    const size_t tot4 = tot / 4;
    const int gr = es_cdiv((long long)tot4, 256);
    hipLaunchKernelGGL(k_dconv_reduce, dim3(gr > 8192 ? 8192 : gr), dim3(256), 0, st, (const float4*)ws, ms, tot4, Cout / 4, dW, Cout,
                       accumulate);
    ES_CHECK_LAUNCH();
  }
  return 0;
}
// dW[K][Cin][Cout] (+)= X^T dY over the dense grid; Xh = bf16 rows of the operator's input, dYh = bf16 rows of its output gradient;
// transposed 0: nn.Conv3d (mode 2), 1: nn.ConvTranspose3d(k = 2, s = 2) (mode 5)
extern "C" int es_dconv_wgrad_bf16(const void* Xh, int ldx, const void* dYh, int ldy, const int* geom_host, int transposed, int Cin,
                                   int Cout, float* dW, int accumulate, void* stream) {
  return dconv_wgrad_impl(Xh, ldx, dYh, ldy, geom_host, transposed, Cin, Cout, dW, accumulate, nullptr, 0, 0, stream);
}
// ... with a workspace (es_dconv_wgrad_workspace_floats; 0 = not needed): launches of few tiles over many rows slice the rows
extern "C" int es_dconv_wgrad_ws_bf16(const void* Xh, int ldx, const void* dYh, int ldy, const int* geom_host, int transposed, int Cin,
                                      int Cout, float* dW, int accumulate, float* ws, size_t ws_floats, void* stream) {
  return dconv_wgrad_impl(Xh, ldx, dYh, ldy, geom_host, transposed, Cin, Cout, dW, accumulate, ws, ws_floats, 1, stream);
}
