/*
 * jpeg_oracle.h -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the reference baseline-JPEG decode path (entropy decode ->
 * dequantise + IDCT -> planes -> upsample + colour).  It is the checker the parity
 * tests compare the HIP path against, and it is itself pinned, bit for bit, against
 * the compiled reference (oracle/_ref/libjpegref.so) on every fixture under
 * tests/golden and on exhaustive leaf sweeps (tests/test_oracle_vs_ref.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef JPEG_ORACLE_H
#define JPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field layout as RefOut in ref_harness.c so one ctypes struct serves both. */
typedef struct {
    int32_t X, Y, ncs;
    int32_t hsf[4], vsf[4];
    int32_t plane_w[4], plane_h[4];
    uint8_t *plane[4];
    uint8_t *rgb;
    int32_t pitch;
    int64_t nblocks;
    int16_t *coefs;
} OrcOut;

int orc_decode(const uint8_t *buf, size_t len, int want_tap, int want_rgb, OrcOut *out);
void orc_free(OrcOut *out);
double orc_time_decode(const uint8_t *buf, size_t len, int iters);

void orc_dequant_idct(const int16_t *zz, const uint16_t *qt_natural, int16_t *out64);
void orc_idct8(float *d);
int orc_ycbcr_row(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int n, uint8_t *rgb);

#ifdef __cplusplus
}
#endif
#endif
