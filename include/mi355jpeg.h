/*
 * mi355jpeg.h -- C ABI of libmi355jpeg.so, the MI355X (gfx950) baseline-JPEG
 * decode path.  Plain pointers and sizes only; callable from C, ctypes, cgo, JNI.
 *
 * The reference decoder has no library or FFI boundary of its own: it is one
 * executable whose main() calls decode_jpeg_buffer() and ycbcr_rgb()
 * (jpeg_decoder.c:55, :95-98).  Those names are exported too (see
 * mi355jpeg_compat.h).  The functions below are the part the single-image API
 * cannot express: a device context and a batch of images that is decoded by one
 * sequence of kernel launches with every input and output resident in HBM.
 *
 * Hot path replaced, per stage (reference file:line -> kernel):
 *   bitstream.c:22-176 (FF00 unstuffing, RSTn discovery)      -> mj_k_scan_*
 *   decode.c:535-723   (decode_scan / decode_data_unit)       -> mj_k_entropy_*
 *   quant_table.c:131-152 + dct.c:158-204,296-341 + decode.c:436-533
 *   + colour_conversion.c:5-83 (dequant, IDCT, planes, RGB)   -> mj_k_idct_colour
 *
 * All functions return 0 on success and a negative value on failure;
 * mj_last_error() returns a thread-local message for the last failure.
 */
#ifndef MI355JPEG_H
#define MI355JPEG_H

#include <stddef.h>
#include <stdint.h>

#include "mi355jpeg_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mj_ctx mj_ctx;
typedef struct mj_batch mj_batch;

/* flags for mj_batch_create */
#define MJ_WANT_PLANES 0x1u   /* also store the padded u8 component planes (Image.buf) */
#define MJ_TIMING      0x2u   /* bracket every kernel with HIP events on the batch stream */
#define MJ_ENTROPY_SERIAL 0x4u /* diagnostic: one lane per restart interval, no subsequence sync */
#define MJ_OUT_RGBA    0x8u   /* pixels leave as R, G, B, 255 (four bytes) instead of RGB24; rgb_pitch = 4 * X */
#define MJ_FLAGS_ALL   0xFu   /* any other bit is refused */

typedef struct {
    int32_t X, Y;            /* frame size in pixels */
    int32_t ncs;             /* 1 (grey) or 3 (YCbCr) */
    int32_t hsf[4], vsf[4];  /* sampling factors per component */
    int32_t plane_w[4], plane_h[4]; /* padded plane sizes (decode.c:108-115) */
    int32_t mcus_x, mcus_y;  /* MCU grid */
    int32_t blocks_per_mcu;
    int32_t restart_interval; /* MCUs per interval, 0 = none */
    int32_t n_intervals;
    int64_t n_blocks;        /* 8x8 blocks in the scan */
    int64_t entropy_bytes;   /* bytes handed to the GPU (from SOS data to end of file) */
    int64_t rgb_pitch;       /* bytes per RGB row in the device buffer */
    int64_t coef_offset;     /* first block of this image in the batch coefficient buffer */
    int64_t rgb_offset;      /* first byte of this image in the batch RGB buffer */
} mj_image_info;

/* kernel slots reported by mj_batch_kernel_ms (MJ_TIMING) */
enum {
    MJ_T_SCAN = 0,     /* unstuff + restart-marker discovery + subsequence layout */
    MJ_T_ENTROPY = 1,  /* Huffman decode: cold pass + synchronisation rounds (mj_k_entropy_sync), or the speculative form (mj_k_spec_*) */
    MJ_T_ENTROPY_FIX = 2, /* cross-sequence fix-up (no-RST scans) + the writing pass (mj_k_entropy_fix, _write) */
    MJ_T_IDCT_COLOUR = 3, /* dequant + IDCT + upsample + YCbCr->RGB */
    MJ_T_MEMSET = 4,   /* zero-fill of the coefficients in the serial diagnostic mode; else nothing (the counters are cleared by the first scan kernel) */
    MJ_T_COUNT = 5
};

const char *mj_last_error(void);

/* Device context: owns a HIP stream on `device`.  Fails (no CPU fallback) when
 * no HIP device is usable. */
int mj_create(int device, mj_ctx **out);
void mj_destroy(mj_ctx *ctx);
/* Use the caller's hipStream_t (e.g. the framework's current stream) instead of
 * the context's own.  Pass NULL to go back to the context's stream. */
int mj_set_stream(mj_ctx *ctx, void *hip_stream);
/* The PCI address of HIP device `device` as sysfs spells it ("0000:0d:00.0", lower case,
 * NUL-terminated; cap >= 13): the key under /sys/bus/pci/devices/ to the NUMA node and the
 * CPUs next to the card.  Needs no context.  No counterpart in the reference, which is one
 * thread on one CPU (jpeg_decoder.c:29-112); a process per GPU uses it to put itself and
 * its page-locked buffers on the GPU's side of the host before it allocates them. */
int mj_device_pci_bus_id(int device, char *buf, size_t cap);
/* How the batches built from now on (process-wide) bring the lanes of the Huffman decode into
 * step -- the part of decode_scan (decode.c:535-663) that a serial decoder does not have.
 * mode 0: the library chooses by the size of the batch (the default); 1: always the
 * synchronisation rounds; 2: always the speculative form (every lane decoded once per block
 * slot of the MCU and a little further, the true decode looked up: several times the work, no
 * serial chain -- what a batch that leaves the chip idle gets by itself).  sub_bits, lane_subs
 * (1, 2, 4 or 8) and extra (subsequences a lane decodes past its own, 1 or more) are the
 * parameters of the speculative form; 0 leaves each to the library.  The results are the same
 * bit for bit whatever is set here; returns 0, or -1 for values outside those ranges.  No
 * counterpart in the reference; for callers that know their workload, and for the tests. */
int mj_set_sync_policy(int mode, int sub_bits, int lane_subs, int extra);

/* Parse n baseline JPEG files with the host parsers (decode_frame_header,
 * decode_huff_tables, decode_quant_table, decode_restart_interval,
 * decode_scan_header -- the reference's decode.c:160-415 marker walk), flatten
 * their tables, allocate device buffers and copy the entropy-coded bytes to HBM.
 * bufs[i] is borrowed for the duration of the call only.  Device memory is not cleared:
 * where a damaged stream leaves blocks undecoded, the outputs hold whatever was there.
 * Accepted: 8-bit SOF0 / SOF1 Huffman files, grey or three components, coded in one
 * interleaved scan (what the reference decodes, decode.c:535-663) or in several scans that
 * each hold one component, or one component alone and the other two together (T.81 A.2.2,
 * A.2.3; the reference has no defined output for files of several scans).  For a file of the
 * second kind mj_image_info reports n_blocks / entropy_bytes / n_intervals summed over its
 * scans, blocks_per_mcu 1 when every scan holds one component and 0 otherwise, and the plane
 * of a component padded to whole blocks when it has a scan of its own, to whole MCUs when it
 * shares one; mj_batch_copy_coefs returns the blocks scan after scan in the order of the
 * file, each scan in its own coding order; the planes are always kept. */
int mj_batch_create(mj_ctx *ctx, const uint8_t *const *bufs, const size_t *lens, int n, uint32_t flags,
                    mj_batch **out);
/* A batch of one image from already-parsed structs: what the reference hands to
 * decode_scan() at its SOS case (decode.c:333; prototype decode.c:37-45).
 * `entropy` points at the first entropy-coded byte; entropy_len may run to the end
 * of the file -- the scan kernels find the marker that ends the segment. */
int mj_batch_create_scan(mj_ctx *ctx, const uint8_t *entropy, size_t entropy_len, const FrameHeader *fh,
                         const ScanHeader *sh, const HuffTables *hts, const QuantTables *qts, RestartInterval ri,
                         uint32_t flags, mj_batch **out);
void mj_batch_destroy(mj_batch *b);

/* Page-locked host memory for JPEG files: entropy-coded bytes in such memory (or in
 * memory the caller registered with the HIP runtime) are copied to HBM straight from
 * where they lie; bytes in pageable memory are gathered in a staging buffer first. */
void *mj_host_alloc(size_t bytes);
void mj_host_free(void *p);

/* A stream of batches, for callers that decode batch after batch (the role of the
 * reference's mmap -> decode_jpeg_buffer hand-off, jpeg_decoder.c:44-55, at batch
 * granularity).  mj_pipe_submit parses the files, enqueues the copy of their
 * entropy-coded bytes on the context's copy stream and the decode behind it on the
 * context's stream (consecutive batches alternate between that stream and a second one
 * of the pipeline's, so that the tail of one decode overlaps the next, unless the caller
 * has set a stream of its own with mj_set_stream), and returns without waiting: the host
 * work and the copy of batch k + 1 run while batch k decodes.  `depth` batches are in flight at most; each has
 * its own device memory, allocated on first use and reused from then on.  *out is
 * owned by the pipeline and is recycled by the depth-th submit after it (that submit
 * waits for its decode first): read results (mj_batch_sync, mj_batch_device_rgb,
 * copies, status) before then; mj_batch_destroy on it does nothing.  bufs[i] must stay valid and
 * unchanged until mj_batch_sync(*out) returns.  Flags as for mj_batch_create. */
typedef struct mj_pipe mj_pipe;
int mj_pipe_create(mj_ctx *ctx, int depth, uint32_t flags, mj_pipe **out);
int mj_pipe_submit(mj_pipe *p, const uint8_t *const *bufs, const size_t *lens, int n, mj_batch **out);
void mj_pipe_destroy(mj_pipe *p);
/* Host time spent inside mj_pipe_submit since the pipeline was created (or since the last
 * call with reset != 0), in milliseconds: waiting for the slot to come free (the GPU is
 * the slower side) and working (header parse, table flattening, layout, enqueueing);
 * *submits counts all submits so far. */
int mj_pipe_host_ms(mj_pipe *p, double *wait_ms, double *work_ms, int64_t *submits, int reset);

/* No counterpart in the reference.  For callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid it):
 * mj_batch_create + mj_batch_decode in one call that allocates nothing on the device.
 * The caller passes the files and the two output buffers (device memory, 16-byte
 * aligned: coefficients as mj_batch_device_coefs describes, RGB as mj_batch_device_rgb);
 * everything in between -- the entropy-coded bytes, the unstuffed streams, the
 * subsequence records, the descriptor tables -- lives in a static device array of
 * MJ_STATIC_SCRATCH_BYTES per GPU, which the library holds from load to process exit.
 * A batch that needs more (about three times its compressed size plus tables: e.g.
 * eight 4K images, or one 8K image) is refused with an error; so is any flag but
 * MJ_OUT_RGBA.  Uploads and kernels are enqueued in order on the context's stream;
 * the call returns once the uploads are through (bufs[i] may then be released), the
 * kernels run on.  *view (optional) is the context's own batch object for this call:
 * valid until the next mj_decode_static or mj_destroy on this context, accepted by
 * the mj_batch_* accessors (image info, status, sync, copies), ignored by
 * mj_batch_destroy.  Every call on one GPU uses the same work memory: a call waits
 * for the previous call of its own context, and calls from different contexts on one
 * GPU must not overlap -- they would race. */
#define MJ_STATIC_SCRATCH_BYTES (64u << 20)
int mj_decode_static(mj_ctx *ctx, const uint8_t *const *bufs, const size_t *lens, int n, uint32_t flags, void *dev_coefs,
                     size_t coefs_bytes, void *dev_rgb, size_t rgb_bytes, mj_batch **view);

/* The upsample + colour step on its own, from planes in host memory to RGB24 in host memory:
 * what ycbcr_rgb (colour_conversion.c:30-83; ncs = 3) and y_rgb (:5-28; ncs = 1) do with an
 * Image.  planes[c] / strides[c] are Image.buf[c] and its row stride (the reference derives it
 * from the frame header: the component's width padded to whole MCUs), hsf / vsf the components'
 * sampling factors; output pixel (i, j) takes component c at row (uint32)(i * vsf/vmax), column
 * (uint32)(j * hsf/hmax), nearest neighbour as in the reference.  Of every plane, the rows and
 * bytes that those loops read are uploaded, no more.  Synchronous; device memory is kept in
 * the context from call to call.  ncs = 1 reads the plane at the stride given (the reference's
 * y_rgb strides by `width`: pass width as the stride to get that). */
int mj_planes_to_rgb(mj_ctx *ctx, int ncs, const uint8_t *const *planes, const int32_t *strides, const int32_t *hsf,
                     const int32_t *vsf, int width, int height, uint8_t *pixels, size_t pitch);

/* decode_scan (decode.c:535-663) called on its own, outside decode_jpeg_buffer, gets a
 * Bitstream with no length (bitstream.h:4-8) and, like the reference, trusts the stream to end
 * in a marker.  This bounds the search for it: decode_scan fails when no marker other than
 * RSTn follows within `bytes` of the cursor.  Default (and bytes = 0): 2^28 - 1, the largest
 * entropy-coded segment the batch entry points accept.  Process-wide. */
void mj_set_scan_limit(size_t bytes);

/* Enqueue the whole hot path for the batch on the stream (asynchronous).  Inputs
 * are already resident in HBM; outputs (coefficients, RGB, optional planes) stay
 * in HBM.  May be called repeatedly; every call decodes from scratch. */
int mj_batch_decode(mj_batch *b);
int mj_batch_sync(mj_batch *b);

int mj_batch_size(const mj_batch *b);
int mj_batch_image_info(const mj_batch *b, int i, mj_image_info *info);
/* Per-image status word written by the kernels: 0 = ok, bit 0 = entropy data
 * ended early or overran, bit 1 = fewer restart markers than the header implies. */
int mj_batch_status(mj_batch *b, int i, uint32_t *status);

/* Offset, from the first entropy-coded byte of image i, of the marker that ends its
 * segment (EOI or the next table/scan segment); the segment length if none was
 * found.  This is where the reference's cursor stands when decode_scan returns. */
int mj_batch_scan_end(mj_batch *b, int i, size_t *offset);

/* Device pointers (valid until mj_batch_destroy). */
void *mj_batch_device_rgb(const mj_batch *b);    /* uint8, image i at rgb_offset, rows of rgb_pitch; 3 bytes per pixel, 4 with MJ_OUT_RGBA */
void *mj_batch_device_coefs(const mj_batch *b);  /* int16[64] per block in natural (row-major) order, blocks in scan order */

/* Synchronous device-to-host copies for checking results. */
int mj_batch_copy_rgb(mj_batch *b, int i, uint8_t *dst, size_t dst_pitch);
int mj_batch_copy_coefs(mj_batch *b, int i, int16_t *dst);          /* n_blocks*64, each block in zigzag order (what dequant_data_unit receives) */
int mj_batch_copy_plane(mj_batch *b, int i, int comp, uint8_t *dst); /* plane_w*plane_h, needs MJ_WANT_PLANES */

/* Accumulated kernel time since the last mj_batch_reset_timing, from HIP events
 * recorded on the stream the kernels run on.  Also returns the launch count. */
int mj_batch_kernel_ms(mj_batch *b, float ms[MJ_T_COUNT], int launches[MJ_T_COUNT]);
int mj_batch_reset_timing(mj_batch *b);

/* Entropy-stage statistics of the last decode (device counters, read back):
 * [0] subsequences, [1] synchronisation rounds summed over workgroups, [2] max rounds
 * in a workgroup, [3] sequences, [4] multi-sequence intervals, [5] fix-up iterations,
 * [6] subsequence decodes spent in the rounds, [7] workgroups that ran.  A batch that took the
 * speculative form (mj_set_sync_policy) has no rounds: [7] and [2] are 0, [1] counts the hops of
 * the true chain that were guesses and [6] the subsequences decoded again from them. */
int mj_batch_entropy_stats(mj_batch *b, int64_t stats[8]);

#ifdef __cplusplus
}
#endif
#endif /* MI355JPEG_H */
