/* C ABI of libes_host.so: host-side (CPU) helpers of the real-data path.  Plain C, no GPU runtime -- it is loaded
 * by the loader's forked worker processes.  The GPU entry points are in es_hip.h. */
#ifndef ES_HOST_H
#define ES_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One 16-bit greyscale depth PNG -> float32 metres, the value LoadDepthFromFile produces
 * (embodiedscan/datasets/transforms/loading.py:68-73: imfrombytes(flag='unchanged').astype(float32) / depth_shift).
 * raw: the INFLATED IDAT stream of a non-interlaced PNG of colour type 0, bit depth 16: H scanlines of 1 filter byte + 2 W
 * bytes (the caller parses the chunks, checks their CRCs and inflates with zlib).  out: (H, W) float32, e.g. one frame of a
 * worker's shared slot.  0 = ok, -1 = unknown filter type, -2 = bad arguments / no memory (fall back to the generic decoder). */
int es_png_gray16_to_f32(const uint8_t* raw, int H, int W, float shift, float* out);

#ifdef __cplusplus
}
#endif
#endif
