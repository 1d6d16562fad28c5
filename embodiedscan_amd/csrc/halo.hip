// Halo-tile sparse convolution (round 6): K = 27 forward / data gradient of the big sparse levels of MinkResNet and the FCAF3D head
// (embodiedscan/models/backbones/mink_resnet.py:88-140 BasicBlock convolutions, dense_heads/fcaf3d_head.py:907-1020 up / out
// blocks: MinkowskiConvolution(kernel_size=3) on 10^4 .. 4*10^5 voxels, 128 / 256 channels).
//
// k_spconv_bf16_fast / _dma (spconv.hip) stage, for EVERY tap, the 128 gathered source rows of a 128-row output tile: 27 x 128 row
// pieces per tile and channel chunk through L2, each followed by 16 MFMAs per wave.  But the rows of a coordinate set are in Z-curve
// order (sort.hip; derived sets inherit it), so the 27-neighbourhoods of the 256 consecutive rows of a tile overlap almost
// completely: measured on the synthetic scans 464 distinct source rows per 256-row tile on the head's finest level (24 of 27
// neighbours present), 320 on the backbone's surface levels (9 of 27) -- against 6 912 staged row pieces.  Here
//   * es_halo_plan (once per kernel map, cached with it): per 256-row tile the SORTED list of distinct source rows (`hrows`, the
//     tile's halo) and, per (row, tap), the 16-bit position of the neighbour in that list (`loc`, 0xFFFF = absent);
//   * k_spconv_halo: the halo's rows of one 64-channel chunk are staged in LDS ONCE (LDS-DMA, swizzled on the source address),
//     then the 27 taps run out of LDS: the A fragment of an MFMA is read at the halo position `loc` names (an absent neighbour
//     reads a zero row), only the weight tile of the (tap, chunk) streams in (16 KB per step, double-buffered LDS-DMA, one
//     barrier per step).  Tile 256 rows x 128 columns, 8 waves as 4 x 2 (64 x 64 each: 16 accumulator fragments), one
//     workgroup per CU (126 KB of LDS).  L2 -> LDS traffic per tile and chunk: 59 KB of halo + 27 x 16 KB of weights, where the
//     gather kernels move 27 x (32 + 16) KB for the same 256 rows.
//   * a tile whose halo exceeds the 704 resident rows (never seen on Z-ordered sets; possible on adversarial row orders) is
//     processed in PAGES of 704 halo rows: every page runs all taps, positions outside the page read the zero row -- slower,
//     never wrong, no second kernel.
// Arithmetic: bf16 operands, f32 accumulation on v_mfma_f32_16x16x32_bf16, additions in (chunk, page, tap) order -- a fixed
// order: run-to-run bit-identical; differs from the gather kernels' (tap, chunk) order in f32 rounding only.
#include "common.h"
#include "../../include/es_hip.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#ifdef ES_EMU
#define ES_SCHED_FENCE() ((void)0)
#define ES_UNIFORM(x) (x)                                        // (tests/emu: a ballot is the same value in every lane already)
#define ES_WAIT_LGKM0() ((void)0)                                // (tests/emu: LDS reads complete at issue; must NOT retire LDS-DMA pieces)
#else
#define ES_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)      // lgkmcnt(0) only: vmcnt / expcnt fields at their maxima
#define ES_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)      // nothing is scheduled across this point
#define ES_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)         // a wave-uniform value into a scalar register
#endif

#define HL_BM 256                 // output rows per tile
#define HL_K 27
#define HL_UMAX 704               // halo rows resident per page (in situ: mean 470 - 520, max 711 on the pruned finest level)
#define HL_HT 8192                // plan: LDS hash slots per tile (>= 256 * 27 = 6 912 distinct rows at worst)

__device__ __attribute__((aligned(16))) unsigned short g_hl_zero[8];

// ------------------------------------------------------------------ plan
// one workgroup per tile of HL_BM rows of the map nbr[n_out][K]: distinct source rows -> sorted -> hrows[tile][0 .. hcnt[tile]),
// loc[row][k] = position of nbr[row][k] in that list (0xFFFF: absent).  loc / hrows are padded to whole tiles by the caller.
// The tile's 6 912 map entries are staged in LDS with coalesced 16-byte loads and the positions leave as coalesced words (the
// first version read and wrote them per row, 108 / 54 bytes apart between neighbouring lanes: 131 us on 1 480 tiles).
__global__ __launch_bounds__(512) void k_halo_plan(const int* __restrict__ nbr, int n_out, int K,
                                                   unsigned short* __restrict__ loc, int* __restrict__ hrows,
                                                   int* __restrict__ hcnt) {
  constexpr int NE = HL_BM * HL_K;                  // map entries of a tile
  __shared__ __attribute__((aligned(16))) int nb[NE];     // the tile's entries; after the inserts: the list of distinct rows
  __shared__ unsigned short slotOf[NE];             // hash slot of every entry (0xFFFF: absent neighbour)
  __shared__ int hk[HL_HT];
  __shared__ unsigned short hv[HL_HT];
  __shared__ int cnt;
  const int t = threadIdx.x, tile = blockIdx.x;
  const long long e0 = (long long)tile * NE, eN = (long long)n_out * K;
  for (int i = t; i < HL_HT; i += 512) hk[i] = -1;
  if (t == 0) cnt = 0;
  for (int i = t; i < NE / 4; i += 512) {           // (e0 is a multiple of 4: 16-byte aligned)
    int4 v = make_int4(-1, -1, -1, -1);
    const long long e = e0 + (long long)i * 4;
    if (e + 3 < eN) v = *(const int4*)(nbr + e);
    else {
      if (e < eN) v.x = nbr[e];
      if (e + 1 < eN) v.y = nbr[e + 1];
      if (e + 2 < eN) v.z = nbr[e + 2];
    }
    ((int4*)nb)[i] = v;
  }
  __syncthreads();
  for (int e = t; e < NE; e += 512) {
    const int v = nb[e];
    unsigned short so = 0xFFFFu;
    if (v >= 0) {
      unsigned s = ((unsigned)v * 2654435761u) >> (32 - 13);
      for (;;) {
        const int old = atomicCAS(&hk[s], -1, v);
        if (old == -1 || old == v) break;
        s = (s + 1) & (HL_HT - 1);
      }
      so = (unsigned short)s;
    }
    slotOf[e] = so;
  }
  __syncthreads();
  int* const list = nb;                             // (the entries are dead: every one has its slot)
  for (int sl = t; sl < HL_HT; sl += 512) {         // occupied slots -> list (order irrelevant: positions come from the ranks);
    const int v = hk[sl];                           // one counter atomic per wave and round, not one per distinct row
    const unsigned long long m = __ballot(v >= 0);
    int base = 0;
    if ((t & 63) == 0 && m) base = atomicAdd(&cnt, __popcll(m));
    base = __shfl(base, 0, 64);
    if (v >= 0) {
      const int pos = base + __popcll(m & ((1ull << (t & 63)) - 1ull));
      list[pos] = v;
      hv[sl] = (unsigned short)pos;                 // slot -> list index
    }
  }
  __syncthreads();
  const int nU = cnt;
  unsigned short* const rk = (unsigned short*)hk;   // list index -> rank (the keys are dead: every slot knows its list index)
  int* const hr = hrows + (size_t)tile * NE;
  for (int i = t; i < nU; i += 512) {               // rank by counting (nU ~ 300 .. 700: broadcast reads)
    const int v = list[i];
    int r = 0;
    for (int j = 0; j < nU; ++j) r += list[j] < v ? 1 : 0;
    rk[i] = (unsigned short)r;
    hr[r] = v;
  }
  if (t == 0) hcnt[tile] = nU;
  __syncthreads();
  unsigned int* const dst = (unsigned int*)(loc + (size_t)tile * NE);
  for (int i = t; i < NE / 2; i += 512) {
    const unsigned int s0 = slotOf[2 * i], s1 = slotOf[2 * i + 1];
    const unsigned int p0 = s0 == 0xFFFFu ? 0xFFFFu : rk[hv[s0]], p1 = s1 == 0xFFFFu ? 0xFFFFu : rk[hv[s1]];
    dst[i] = p0 | (p1 << 16);
  }
}

extern "C" size_t es_halo_plan_rows(int n_out) { return (size_t)es_cdiv(n_out, HL_BM) * HL_BM; }

extern "C" int es_halo_plan(const int* nbr, int n_out, int K, void* loc, int* hrows, int* hcnt, void* stream) {
  if (n_out <= 0) return 0;
  if (K != HL_K || nbr == nullptr) return -4;
  hipLaunchKernelGGL(k_halo_plan, dim3(es_cdiv(n_out, HL_BM)), dim3(512), 0, (hipStream_t)stream, nbr, n_out, K,
                     (unsigned short*)loc, hrows, hcnt);
  ES_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ convolution
// Xh (n_in x ldx) bf16 source rows; W [tap][N][Kd] bf16 (Kd = reduction channels contiguous); Y (n_out x ldy) f32.
//
// Software pipeline at HALF-step granularity.  A step s = (chunk, page, tap) is two blocks of 16 MFMAs (k halves h = 0, 1) with a
// fixed register set each.  While the h = 0 block runs, the h = 1 fragments of the same step are being read from LDS; while the
// h = 1 block runs, the h = 0 fragments of step s + 1 are (A: the static halo, B: weight buffer (s + 1) % 3), and the weight tile
// of step s + 2 is in flight by LDS-DMA into buffer (s + 2) % 3.  One barrier per step, between the two blocks, whose only job is
// to publish a landed weight tile; no wave starts a block of matrix instructions behind its own LDS reads.  The steady-state
// step is ONE basic block (with control flow between the reads and the MFMAs the compiler's wait-count pass drains lgkmcnt at the
// join); group boundaries (next chunk / page: the new halo is staged right behind the barrier of the group's last step, whose
// fragments are in registers by then) and the last two steps go through a generic variant with the same arithmetic.
struct HlSeq { int ti, pg, c, tap; };     // step: tap ordinal within the tile's active taps, halo page, channel chunk, tap id

template <int BNT>
__global__ __launch_bounds__(512, 2) void k_spconv_halo(const unsigned short* __restrict__ Xh, int ldx,
                                                        const unsigned short* __restrict__ W, int Kd, int N,
                                                        const unsigned short* __restrict__ loc, const int* __restrict__ hrows,
                                                        const int* __restrict__ hcnt, int n_out, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int ldy, int accumulate, int colTiles, int total,
                                                        int per, int mirror) {
  constexpr int RB = 128;                                   // bytes per LDS row: 64 channels
  constexpr int ZROW = HL_UMAX;                             // the zero row
  constexpr int H_BYTES = (HL_UMAX + 1) * RB;
  constexpr int B_BYTES = BNT * RB;
  constexpr int OFF_B = H_BYTES, OFF_LOC = OFF_B + 3 * B_BYTES, OFF_TAP = OFF_LOC + HL_BM * HL_K * 2;
  constexpr int NH = HL_UMAX * 8 / 512;                     // halo pieces per thread and page (10)
  constexpr int NB = BNT / 64;                              // weight pieces per thread and step
  constexpr int NF = BNT / 32;                              // column fragments per wave (BNT / 2 columns)
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_TAP + 64 * 4];
  unsigned short* const locS = (unsigned short*)(smem + OFF_LOC);
  int* const tapS = (int*)(smem + OFF_TAP);                 // [0..26] flags of the taps with at least one neighbour in this tile
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  // PERSISTENT workgroups (one per CU: 142 KB of LDS): workgroup b walks the tiles (b & 7) * per + (b >> 3) + it * (gridDim.x / 8) --
  // XCD x = b % 8 (observed placement, used for speed only) runs a contiguous range of `per` tiles, so neighbouring tiles share halo
  // rows and every tile the weights in that XCD's L2 -- and the 128 KB of f32 output stores of a tile drain under the next tile's work
  // instead of holding the CU until the workgroup retires.
  for (int it = 0;; ++it) {
  const int lx = (int)(blockIdx.x >> 3) + it * (int)(gridDim.x >> 3);
  const int L = (int)(blockIdx.x & 7) * per + lx;
  if (lx >= per || L >= total) break;
  if (it > 0) __syncthreads();                               // every wave is through the previous tile: its LDS may be overwritten
  const int rt = L / colTiles, ct = L - rt * colTiles;
  const int row0 = rt * HL_BM, n0 = ct * BNT;
  const int nU = hcnt[rt];
  const int nP = nU > HL_UMAX ? (nU + HL_UMAX - 1) / HL_UMAX : 1;
  const int* const hr = hrows + (size_t)rt * HL_BM * HL_K;

  if (t < 32) tapS[t] = 0;
  if (t < 32) ((unsigned int*)(smem + ZROW * RB))[t] = 0u;  // the zero row
  __syncthreads();
  {                                                          // the tile's positions -> LDS (13 824 contiguous bytes), tap flags
    const uint4* g = (const uint4*)(loc + (size_t)row0 * HL_K);
    for (int i = t; i < HL_BM * HL_K / 8; i += 512) {
      const uint4 v = g[i];
      ((uint4*)locS)[i] = v;
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned int p = (w4[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu;
        if (p != 0xFFFFu) tapS[mirror ? HL_K - 1 - (i * 8 + q) % HL_K : (i * 8 + q) % HL_K] = 1;
      }
    }
  }
  __syncthreads();
  // the tile's active taps as a 27-bit mask in a scalar register (the step sequence walks its set bits: no LDS read, no wait)
  unsigned int tm = 0;
  {
    const int f = (lane < HL_K) ? tapS[lane] : 0;
    tm = ES_UNIFORM((unsigned int)__ballot(f));
  }
  const int nT = __popc(tm);

  f32x4 acc[4][NF];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // LDS-DMA pieces: piece (j, wave) covers LDS bytes [(j * 8 + wv) * 1024, + 1024) = 8 rows of 128 B; lane -> row (lane >> 3),
  // 16-byte slot lane & 7, which receives the row's granule slot ^ key(row), key(row) = (row >> 1) & 7 (independent of j)
  const int kslot = ((lane & 7) ^ (((wv & 1) * 4 + (lane >> 4)) & 7)) * 8;       // element offset of the source granule
  const int b_off0 = (n0 + wv * 8 + (lane >> 3)) * Kd + kslot;                   // piece j: + j * 64 * Kd
  const int nC = Kd >> 6;
  const int nSteps = nC * nP * nT;
  auto issue_halo = [&](int c0, int page) {
    int h_row[NH];                                          // source rows of this thread's pieces (-1: past the halo)
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int u = page * HL_UMAX + (j * 8 + wv) * 8 + (lane >> 3);
      const int r = hr[u < HL_BM * HL_K ? u : HL_BM * HL_K - 1];     // (unconditional loads: all ten in flight before the first DMA)
      h_row[j] = u < nU ? r : -1;
    }
#pragma unroll
    for (int j = 0; j < NH; ++j) {                          // (no branch: a piece past the halo copies zero granules -- with a branch per piece
      const unsigned short* p = h_row[j] >= 0 ? (Xh + (size_t)h_row[j] * ldx + c0 + kslot) : g_hl_zero;     // the compiler waits vmcnt(0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,                     // before each one)
                                       (__attribute__((address_space(3))) void*)(smem + (j * 8 + wv) * 1024), 16, 0, 0);
    }
  };
  auto issue_b = [&](int buf, int tap, int c0) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const unsigned short* p = W + (size_t)tap * N * Kd + b_off0 + j * 64 * Kd + c0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + OFF_B + buf * B_BYTES + (j * 8 + wv) * 1024), 16, 0, 0);
    }
  };
  const int li = lane & 15, kq = lane >> 4;
  const int f_key = (li >> 1) & 7;                           // B tile rows of a fragment are 16 * x + li
  const unsigned short* const lp = locS + (wr * 64 + li) * HL_K;
  const unsigned char* const b_frag = smem + OFF_B + (wc * (BNT / 2) + li) * RB;
  auto load_idx = [&](int (&raw)[4], const HlSeq& q) {      // raw 16-bit positions; page offset and range check at the use (read_a)
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) raw[mf] = (int)lp[mf * 16 * HL_K + (mirror ? HL_K - 1 - q.tap : q.tap)];
  };
  auto proc = [&](int (&idx)[4], const int (&raw)[4], int pg) {       // raw positions -> LDS rows of the resident page (or the zero row)
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
      const int p = raw[mf] - pg * HL_UMAX;
      idx[mf] = ((unsigned)p < (unsigned)HL_UMAX) ? p : ZROW;
    }
  };
  auto read_a = [&](bf16x8_t (&a)[4], const int (&idx)[4], int h) {
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
      a[mf] = *(const bf16x8_t*)(smem + idx[mf] * RB + (((h * 4 + kq) ^ ((idx[mf] >> 1) & 7)) * 16));
  };
  auto read_b = [&](bf16x8_t (&b)[NF], int buf, int h) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
      b[nf] = *(const bf16x8_t*)(b_frag + buf * B_BYTES + nf * 16 * RB + (((h * 4 + kq) ^ f_key) * 16));
  };
  auto mma = [&](const bf16x8_t (&a)[4], const bf16x8_t (&b)[NF]) {
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
  };
  const int tap_first = tm ? __builtin_ctz(tm) : 0;
  auto advance = [&](HlSeq q) {
    if (++q.ti == nT) {
      q.ti = 0; q.tap = tap_first;
      if (++q.pg == nP) { q.pg = 0; ++q.c; }
    } else {
      q.tap = __builtin_ctz(tm & ~((2u << q.tap) - 1u));     // next set bit above the current tap
    }
    return q;
  };

  if (nSteps > 0) {
    bf16x8_t a0[4], b0[NF], a1[4], b1[NF];                   // fragments of the h = 0 / h = 1 block
    int idxC[4], rawN[4];                                    // LDS rows of the current step; raw positions of the next one
    int s = 0;                                               // global step counter: weight tile of step s lives in buffer s % 3
    HlSeq q1 = {0, 0, 0, tap_first}, q2 = advance(q1);       // steps s + 1 and s + 2 while step s runs (q1 = the step itself before the loop)
    issue_b(0, q1.tap, 0);
    if (nSteps > 1) issue_b(1, q2.tap, q2.c * 64);
    for (int c = 0; c < nC; ++c)
      for (int pg = 0; pg < nP; ++pg) {
        // ---- group prologue (one bubble per group): stage the halo of (chunk c, page pg), then read the first step's h = 0 block
        // (every wave's reads of the previous halo were retired by the barrier inside the previous group's last step: its h = 1
        // fragments are read before that barrier and nothing is prefetched behind it)
        issue_halo(c * 64, pg);
        load_idx(rawN, q1);                                  // (q1 = the group's first step here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                      // halo (and the first weight tiles) have landed
        proc(idxC, rawN, pg);
        read_a(a0, idxC, 0);
        read_b(b0, s % 3, 0);
        q1 = q2; q2 = advance(q2);                           // q1 = step s + 1, q2 = step s + 2
        if (nT > 1) load_idx(rawN, q1);
        // one step; INNER: the next step belongs to this group and its h = 0 block is prefetched under this step's h = 1 block.
        // The INNER variant has no branch between its LDS reads and its matrix instructions (a join there makes the compiler's
        // wait-count pass drain lgkmcnt and the reads no longer run under the MFMAs).
        auto step = [&](auto inner_c) {
          constexpr bool INNER = decltype(inner_c)::value;
          ES_WAIT_LGKM0();                                   // (s, h = 0) and rawN, read under the previous block, are retired
          read_a(a1, idxC, 1);
          read_b(b1, s % 3, 1);
          ES_SCHED_FENCE();                                  // (the machine scheduler must neither sink the reads below the block nor
          mma(a0, b0);                                       //  hoist the block's successor: the two phases are the pipeline)
          ES_SCHED_FENCE();
          if (INNER || s + 1 < nSteps) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // weights of step s + 1 have landed (this wave's pieces) ...
            __syncthreads();                                  // ... everybody's; (s, h = 1) is retired
            if (s + 2 < nSteps) {
              const HlSeq qn = INNER ? q2 : advance(q1);     // step s + 2 (at a group's last step q1 is the next group's first)
              issue_b((s + 2) % 3, qn.tap, qn.c * 64);
            }
          } else {
            ES_WAIT_LGKM0();
          }
          if (INNER) {
            proc(idxC, rawN, pg);
            read_a(a0, idxC, 0);
            read_b(b0, (s + 1) % 3, 0);
            q1 = q2; q2 = advance(q2);
            load_idx(rawN, q1);                              // (behind the group's last inner step: positions nobody uses -- in bounds)
          }
          ES_SCHED_FENCE();
          mma(a1, b1);
          ES_SCHED_FENCE();
          ++s;
        };
        for (int ti = 0; ti + 1 < nT; ++ti) step(std::true_type());
        step(std::false_type());
      }
  }
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + wr * 64 + mf * 16 + kq * 4 + r;
      if (row < n_out) {
        float* p = Y + (size_t)row * ldy + n0 + wc * (BNT / 2) + li;
        float v[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) v[nf] = acc[mf][nf][r] + (bias ? bias[n0 + wc * (BNT / 2) + nf * 16 + li] : 0.f);
        if (accumulate) {
          float y0[NF];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) y0[nf] = p[nf * 16];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) p[nf * 16] = y0[nf] + v[nf];
        } else {
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) p[nf * 16] = v[nf];
        }
      }
    }
  }                                                          // (next tile of this workgroup)
}

static int ES_OPT_HALO_WGS = 0;           // workgroups of a launch (0: 256 = one per CU; a multiple of 8)
static int ES_OPT_HALO_MIN_WGS = 192;     // the halo kernel takes launches with at least this many workgroups (smaller ones: tap-split gather kernels)

extern "C" int es_halo_set_option(int key, int value) {
  if (key == 30) { ES_OPT_HALO_MIN_WGS = value; return 0; }
  if (key == 31) { ES_OPT_HALO_WGS = value; return 0; }
  return -1;
}

extern "C" int es_spconv_halo_supported(int n_out, int n_in, int ldx, int K, int Cin, int Cout) {
  if (K != HL_K || n_out <= 0) return 0;
  if ((Cin % 64) || (Cout % 128) || (ldx % 8)) return 0;
  if ((long long)n_in * ldx >= (1ll << 31) || (long long)K * Cout * Cin >= (1ll << 31)) return 0;
  return es_cdiv(n_out, HL_BM) * (Cout / 128) >= ES_OPT_HALO_MIN_WGS ? 1 : 0;
}

extern "C" int es_spconv_halo_bf16(const void* Xh, int ldx, const void* W_bf16, const void* loc, const int* hrows, const int* hcnt,
                                   int n_out, int n_in, int K, int Cin, int Cout, const float* bias, float* Y, int ldy,
                                   int accumulate, int mirror, void* stream) {
  if (n_out <= 0) return 0;
  if (K != HL_K || (Cin % 64) || (Cout % 128) || (ldx % 8) || ((((uintptr_t)Xh) | ((uintptr_t)W_bf16) | ((uintptr_t)loc)) & 15)) return -4;
  if ((long long)n_in * ldx >= (1ll << 31) || (long long)K * Cout * Cin >= (1ll << 31)) return -4;
  const int rowTiles = es_cdiv(n_out, HL_BM), colTiles = Cout / 128, total = rowTiles * colTiles, per = es_cdiv(total, 8);
  const int wgs = ES_OPT_HALO_WGS > 0 ? ES_OPT_HALO_WGS : 256;            // persistent: one workgroup per CU (MI355X: 256 CUs)
  hipLaunchKernelGGL((k_spconv_halo<128>), dim3(8 * per < wgs ? 8 * per : (wgs & ~7)), dim3(512), 0, (hipStream_t)stream, (const unsigned short*)Xh, ldx,
                     (const unsigned short*)W_bf16, Cin, Cout, (const unsigned short*)loc, hrows, hcnt, n_out, bias, Y, ldy, accumulate,
                     colTiles, total, per, mirror);
  ES_CHECK_LAUNCH();
  return 0;
}
