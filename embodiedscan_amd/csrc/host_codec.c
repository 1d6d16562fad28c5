// Host-side codec helper of the real-data path (reference: embodiedscan/datasets/transforms/loading.py:53-81,
// LoadDepthFromFile = mmcv.imfrombytes(flag='unchanged').astype(float32) / depth_shift).  Plain C for the loader's worker
// processes (built by gcc into libes_host.so; no HIP in here: the workers are forked CPU processes and must not touch the GPU
// runtime).  The depth maps of the dataset are 16-bit greyscale PNGs; the generic decoder the workers used (PIL) spends as long
// on its row pipeline (unfilter, byte swap, copy-out, then numpy's u16 -> f32 -> divide passes) as zlib spends inflating.  Here
// the inflated scanlines (python's zlib module inflates, in C) are unfiltered, byte-swapped, converted to float32 metres and
// written to the destination -- a frame of the worker's shared slot -- in ONE pass over the data.
// Declared in include/es_host.h; bound by embodiedscan_amd/datasets/loading.py (ctypes); PIL stays the decoder for every
// other kind of file and when this library is absent (same values either way: tests/test_dataset.py).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

// PNG specification, 9.4: a = left, b = above, c = upper left; p = a + b - c, the neighbour nearest to p wins (ties: a, b, c).
// Written without branches (p - a = b - c, p - b = a - c, p - c = their sum; selects compile to conditional moves): on
// real depth maps the outcome is close to random, a branchy version ran at 8 ns per byte (mispredictions).
static inline int paeth(int a, int b, int c) {
  int pa = b - c, pb = a - c;
  int pc = pa + pb;
  pa = abs(pa);
  pb = abs(pb);
  pc = abs(pc);
  const int bc = pb <= pc ? b : c;
  return ((pa <= pb) & (pa <= pc)) ? a : bc;
}

// raw: H scanlines of (1 filter-type byte + 2 W bytes of big-endian 16-bit samples), as inflated from the IDAT stream of a
// non-interlaced PNG with colour type 0, bit depth 16.  out[r * W + c] = (float)sample / shift (IEEE float32 division: the value
// numpy computes for `a.astype(float32) / float32(shift)`).  Returns 0, -1 for an unknown filter type, -2 for bad arguments /
// allocation failure (the caller then falls back to its generic decoder).  The input is not modified.
int es_png_gray16_to_f32(const uint8_t* raw, int H, int W, float shift, float* out) {
  if (!raw || !out || H <= 0 || W <= 0) return -2;
  const int n = 2 * W;
  uint8_t* rows = (uint8_t*)calloc((size_t)2 * n, 1);             // previous (initially zero) and current unfiltered scanline
  if (!rows) return -2;
  uint8_t* prev = rows;
  uint8_t* cur = rows + n;
  int rc = 0;
  for (int r = 0; r < H && rc == 0; ++r) {
    const uint8_t* in = raw + (size_t)r * (n + 1);
    const int ft = in[0];
    ++in;
    switch (ft) {
      case 0:
        memcpy(cur, in, (size_t)n);
        break;
      case 1:                                                      // Sub: left neighbour = 2 bytes back (bytes per pixel)
        cur[0] = in[0];
        cur[1] = in[1];
        {
          int a0 = cur[0], a1 = cur[1];
          for (int i = 2; i < n; i += 2) {
            a0 = (uint8_t)(in[i] + a0);
            a1 = (uint8_t)(in[i + 1] + a1);
            cur[i] = (uint8_t)a0;
            cur[i + 1] = (uint8_t)a1;
          }
        }
        break;
      case 2:                                                      // Up
        for (int i = 0; i < n; ++i) cur[i] = (uint8_t)(in[i] + prev[i]);
        break;
      case 3:                                                      // Average (floor of the 9-bit sum / 2)
        cur[0] = (uint8_t)(in[0] + (prev[0] >> 1));
        cur[1] = (uint8_t)(in[1] + (prev[1] >> 1));
        {
          int a0 = cur[0], a1 = cur[1];
          for (int i = 2; i < n; i += 2) {
            a0 = (uint8_t)(in[i] + ((a0 + prev[i]) >> 1));
            a1 = (uint8_t)(in[i + 1] + ((a1 + prev[i + 1]) >> 1));
            cur[i] = (uint8_t)a0;
            cur[i + 1] = (uint8_t)a1;
          }
        }
        break;
      case 4:                                                      // Paeth.  (Measured: running row r + 1 one pixel behind row r in the same
                                                                   //  loop -- four chains instead of two -- is no faster: the loop is bound
                                                                   //  by its ~25 instructions per byte, not by the chain's latency.)
        cur[0] = (uint8_t)(in[0] + prev[0]);                       // (left = upper left = 0: the predictor is `above`)
        cur[1] = (uint8_t)(in[1] + prev[1]);
        {                                                          // the two bytes of a sample are independent chains;
          int a0 = cur[0], a1 = cur[1], c0 = prev[0], c1 = prev[1];  // left / upper-left stay in registers
          for (int i = 2; i < n; i += 2) {
            const int b0 = prev[i], b1 = prev[i + 1];
            a0 = (uint8_t)(in[i] + paeth(a0, b0, c0));
            a1 = (uint8_t)(in[i + 1] + paeth(a1, b1, c1));
            cur[i] = (uint8_t)a0;
            cur[i + 1] = (uint8_t)a1;
            c0 = b0;
            c1 = b1;
          }
        }
        break;
      default:
        rc = -1;
        continue;
    }
    float* o = out + (size_t)r * W;
    for (int c = 0; c < W; ++c) o[c] = (float)(((unsigned)cur[2 * c] << 8) | cur[2 * c + 1]) / shift;
    uint8_t* t = prev;
    prev = cur;
    cur = t;
  }
  free(rows);
  return rc;
}
