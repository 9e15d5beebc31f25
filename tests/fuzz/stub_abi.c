/* Stand-in for the HIP side of the C ABI (csrc/mj_abi.hip) so that the host walker in
 * decode_host.c links into the sanitizer build of the damage fuzzer: there is no device
 * here, every entry point that would need one fails, and decode_jpeg_buffer therefore
 * runs its marker walk and segment parsers up to the scan and returns -1. */
#include "mi355jpeg.h"

const char *mj_last_error(void) { return "stub: no HIP device in the sanitizer build"; }
int mj_create(int device, mj_ctx **out) { (void) device; (void) out; return -1; }
void mj_destroy(mj_ctx *ctx) { (void) ctx; }
int mj_batch_create_scan(mj_ctx *ctx, const uint8_t *entropy, size_t entropy_len, const FrameHeader *fh, const ScanHeader *sh,
                         const HuffTables *hts, const QuantTables *qts, RestartInterval ri, uint32_t flags, mj_batch **out) {
    (void) ctx; (void) entropy; (void) entropy_len; (void) fh; (void) sh; (void) hts; (void) qts; (void) ri; (void) flags; (void) out;
    return -1;
}
int mj_batch_create(mj_ctx *ctx, const uint8_t *const *bufs, const size_t *lens, int n, uint32_t flags, mj_batch **out) {
    (void) ctx; (void) bufs; (void) lens; (void) n; (void) flags; (void) out;
    return -1;
}
void mj_batch_destroy(mj_batch *b) { (void) b; }
int mj_batch_decode(mj_batch *b) { (void) b; return -1; }
int mj_batch_image_info(const mj_batch *b, int i, mj_image_info *info) { (void) b; (void) i; (void) info; return -1; }
int mj_batch_copy_plane(mj_batch *b, int i, int comp, uint8_t *dst) { (void) b; (void) i; (void) comp; (void) dst; return -1; }
int mj_batch_copy_rgb(mj_batch *b, int i, uint8_t *dst, size_t dst_pitch) { (void) b; (void) i; (void) dst; (void) dst_pitch; return -1; }
int mj_batch_scan_end(mj_batch *b, int i, size_t *offset) { (void) b; (void) i; (void) offset; return -1; }
int mj_planes_to_rgb(mj_ctx *ctx, int ncs, const uint8_t *const *planes, const int32_t *strides, const int32_t *hsf, const int32_t *vsf,
                     int width, int height, uint8_t *pixels, size_t pitch) {
    (void) ctx; (void) ncs; (void) planes; (void) strides; (void) hsf; (void) vsf; (void) width; (void) height; (void) pixels; (void) pitch;
    return -1;
}
