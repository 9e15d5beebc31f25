/*
 * ref_harness.c -- TEST INFRASTRUCTURE, not product code.
 *
 * A headless driver that is linked against the *unmodified* reference sources
 * where they lie under /root/reference/src (see oracle/Makefile).  It repeats the
 * call sequence of the reference viewer (jpeg_decoder.c:44-104) without SDL and
 * exposes the results through a small ctypes-friendly C interface.
 *
 * The Huffman-decoded coefficient blocks are captured with a link-time wrap of
 * dequant_data_unit (-Wl,--wrap=dequant_data_unit): the reference calls it once
 * per 8x8 block right after entropy decoding (decode.c:616-637), so the block it
 * receives is exactly "zigzag-ordered coefficients with DC prediction applied".
 * Nothing in the reference tree is touched or copied.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bitstream.h"
#include "colour_conversion.h"
#include "dct.h"
#include "decode.h"
#include "frame_header.h"
#include "quant_table.h"

/* ---- coefficient tap ---------------------------------------------------- */
static int16_t *g_tap = NULL;
static size_t g_tap_blocks = 0, g_tap_cap = 0;
static int g_tap_on = 0;
static void (*g_trace_hook)(const int16_t *du) = NULL;   /* set while ref_decode_traced runs */

int32_t __real_dequant_data_unit(QuantTable *qt, int16_t *du);

int32_t __wrap_dequant_data_unit(QuantTable *qt, int16_t *du);

int32_t __wrap_dequant_data_unit(QuantTable *qt, int16_t *du) {
    if (g_trace_hook) g_trace_hook(du);
    if (g_tap_on) {
        if (g_tap_blocks == g_tap_cap) {
            g_tap_cap = g_tap_cap ? g_tap_cap * 2 : 4096;
            g_tap = (int16_t *) realloc(g_tap, g_tap_cap * 64 * sizeof(int16_t));
        }
        memcpy(g_tap + g_tap_blocks * 64, du, 64 * sizeof(int16_t));
        g_tap_blocks++;
    }
    return __real_dequant_data_unit(qt, du);
}

/* ---- trace into memory the caller shares with this process -----------------
 * The reference walks off its buffers on some streams (bitstream.c:84-176) and the
 * process dies with it.  ref_decode_traced() therefore records into a buffer the
 * caller maps MAP_SHARED before it forks: every block the coefficient tap sees, and
 * for every MCU the restart check's view of the stream (check_marker, decode.c:579;
 * next_byte_restart_marker, decode.c:583; both wrapped at link time like the tap).
 * The parent reads what was recorded up to the moment the child died. */
typedef struct {
    int64_t pos;          /* cursor at the call: byte offset into the file, */
    int32_t off;          /* and bits of that byte already consumed */
    int32_t ret;          /* what check_marker returned: 0 none, 1 restart, 2 end */
    int64_t blocks;       /* blocks tapped before this MCU */
    int64_t pos_after;    /* cursor after next_byte_restart_marker, -1 if it was not called */
} RefCheck;

typedef struct {
    int64_t cap_blocks, cap_checks;    /* set by the caller */
    volatile int64_t n_blocks, n_checks;
    volatile int64_t finished, rc;     /* decode_jpeg_buffer came back, and with what */
    /* int16_t coefs[cap_blocks * 64]; RefCheck checks[cap_checks]; */
} RefTrace;

static RefTrace *g_trace = NULL;
static const uint8_t *g_trace_base = NULL;

static int16_t *trace_coefs(RefTrace *t) {
    return (int16_t *) (t + 1);
}

static RefCheck *trace_checks(RefTrace *t) {
    return (RefCheck *) (trace_coefs(t) + t->cap_blocks * 64);
}

static void trace_block(const int16_t *du) {
    RefTrace *t = g_trace;
    if (t && t->n_blocks < t->cap_blocks) {
        memcpy(trace_coefs(t) + t->n_blocks * 64, du, 64 * sizeof(int16_t));
        t->n_blocks++;
    }
}

uint8_t __real_check_marker(Bitstream *bs);
int __real_next_byte_restart_marker(Bitstream *bs);

uint8_t __wrap_check_marker(Bitstream *bs) {
    uint8_t ret = __real_check_marker(bs);
    RefTrace *t = g_trace;
    if (t && t->n_checks < t->cap_checks) {
        RefCheck *c = trace_checks(t) + t->n_checks;
        c->pos = (int64_t) (bs->encoded_data - g_trace_base);
        c->off = bs->offset;
        c->ret = ret;
        c->blocks = t->n_blocks;
        c->pos_after = -1;
        t->n_checks++;
    }
    return ret;
}

int __wrap_next_byte_restart_marker(Bitstream *bs) {
    int r = __real_next_byte_restart_marker(bs);
    RefTrace *t = g_trace;
    if (t && t->n_checks > 0 && t->n_checks <= t->cap_checks) {
        trace_checks(t)[t->n_checks - 1].pos_after = (int64_t) (bs->encoded_data - g_trace_base);
    }
    return r;
}

/* bytes the caller must map for the given capacities */
size_t ref_trace_bytes(int64_t cap_blocks, int64_t cap_checks) {
    return sizeof(RefTrace) + (size_t) cap_blocks * 64 * sizeof(int16_t) + (size_t) cap_checks * sizeof(RefCheck);
}

/* decode_jpeg_buffer with the trace on; may not return.  `shared` starts with a RefTrace
 * whose cap_* fields are set and whose other fields are zero. */
int ref_decode_traced(const uint8_t *buf, size_t len, void *shared) {
    uint8_t *copy = (uint8_t *) calloc(len + 64, 1);
    memcpy(copy, buf, len);
    g_trace = (RefTrace *) shared;
    g_trace_base = copy;
    g_trace_hook = trace_block;
    Image *img = NULL;
    FrameHeader fh;
    memset(&fh, 0, sizeof fh);
    fh.process = -1;
    int rc = decode_jpeg_buffer(copy, len, &fh, &img);
    g_trace->rc = rc;
    g_trace->finished = 1;
    g_trace = NULL;
    g_trace_hook = NULL;
    if (rc == 0 && img) {
        free_img(img);
        free_frame_header(&fh);
    }
    free(copy);
    return rc;
}

/* ---- result object ------------------------------------------------------ */
typedef struct {
    int32_t X, Y, ncs;
    int32_t hsf[4], vsf[4];
    int32_t plane_w[4], plane_h[4];
    uint8_t *plane[4];   /* copies of Image.buf[c], plane_w*plane_h bytes */
    uint8_t *rgb;        /* Y rows of `pitch` bytes, RGB24 */
    int32_t pitch;
    int64_t nblocks;     /* blocks seen by the tap */
    int16_t *coefs;      /* nblocks*64, zigzag order, scan order */
} RefOut;

static uint16_t pad_to(uint16_t v, int unit) {
    return (uint16_t) (v + ((v % unit) ? (unit - (v % unit)) : 0));
}

/* Same sequence as jpeg_decoder.c:52-104.  `want_tap`: record coefficients.
 * `want_rgb`: run the colour step.  Returns 0 / -1. */
int ref_decode(const uint8_t *buf, size_t len, int want_tap, int want_rgb, RefOut *out) {
    memset(out, 0, sizeof *out);
    /* check_marker reads up to 3 bytes past the cursor (bitstream.c:86-103) */
    uint8_t *copy = (uint8_t *) calloc(len + 64, 1);
    memcpy(copy, buf, len);

    g_tap_blocks = 0;
    g_tap_on = want_tap;
    Image *img = NULL;
    FrameHeader fh;
    memset(&fh, 0, sizeof fh);
    fh.process = -1;
    int rc = decode_jpeg_buffer(copy, len, &fh, &img);
    g_tap_on = 0;
    if (rc != 0 || img == NULL) {
        free(copy);
        return -1;
    }
    out->X = fh.X;
    out->Y = fh.Y;
    out->ncs = fh.ncs;
    for (int i = 0; i < fh.ncs && i < 4; i++) {
        Component *c = fh.cs + i;
        out->hsf[i] = c->hsf;
        out->vsf[i] = c->vsf;
        out->plane_w[i] = pad_to(c->x, 8 * c->hsf);
        out->plane_h[i] = pad_to(c->y, 8 * c->vsf);
        size_t n = (size_t) out->plane_w[i] * out->plane_h[i];
        out->plane[i] = (uint8_t *) malloc(n);
        memcpy(out->plane[i], img->buf[i], n);
    }
    if (want_tap) {
        out->nblocks = (int64_t) g_tap_blocks;
        out->coefs = (int16_t *) malloc(g_tap_blocks * 64 * sizeof(int16_t) + 16);
        memcpy(out->coefs, g_tap, g_tap_blocks * 64 * sizeof(int16_t));
    }
    if (want_rgb) {
        uint16_t pitch = (uint16_t) (3 * fh.X);
        out->pitch = pitch;
        /* the viewer zeroes the surface first (jpeg_decoder.c:79-81) */
        out->rgb = (uint8_t *) calloc((size_t) pitch * fh.Y + 16, 1);
        if (fh.ncs == 1) {
            y_rgb(&fh, img, out->rgb, fh.X, fh.Y, pitch);
        } else if (fh.ncs == 3) {
            ycbcr_rgb(&fh, img, out->rgb, fh.X, fh.Y, pitch);
        } else {
            rc = -1;
        }
    }
    free_img(img);
    free_frame_header(&fh);
    free(copy);
    return rc;
}

void ref_free(RefOut *out) {
    for (int i = 0; i < 4; i++) {
        free(out->plane[i]);
    }
    free(out->rgb);
    free(out->coefs);
    memset(out, 0, sizeof *out);
}

/* Timed loop for the CPU baseline: decode + colour only, buffer already in
 * memory.  Returns seconds for `iters` repetitions (wall, CLOCK_MONOTONIC). */
double ref_time_decode(const uint8_t *buf, size_t len, int iters) {
    uint8_t *copy = (uint8_t *) calloc(len + 64, 1);
    memcpy(copy, buf, len);
    uint8_t *rgb = NULL;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int it = 0; it < iters; it++) {
        Image *img = NULL;
        FrameHeader fh;
        memset(&fh, 0, sizeof fh);
        fh.process = -1;
        if (decode_jpeg_buffer(copy, len, &fh, &img) != 0 || !img) {
            free(copy);
            free(rgb);
            return -1.0;
        }
        uint16_t pitch = (uint16_t) (3 * fh.X);
        if (!rgb) {
            rgb = (uint8_t *) calloc((size_t) pitch * fh.Y + 16, 1);
        }
        if (fh.ncs == 1) {
            y_rgb(&fh, img, rgb, fh.X, fh.Y, pitch);
        } else if (fh.ncs == 3) {
            ycbcr_rgb(&fh, img, rgb, fh.X, fh.Y, pitch);
        }
        free_img(img);
        free_frame_header(&fh);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(copy);
    free(rgb);
    return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

/* ---- leaf probes for unit-level pinning of the restatement -------------- */

/* dequant (quant_table.c:131-152) then IDCT (dct.c:158-204) on one block.
 * `zz` is a zigzag-order block, `qt_natural` a natural-order table. */
void ref_dequant_idct(const int16_t *zz, const uint16_t *qt_natural, int16_t *out64) {
    QuantTable qt;
    qt.precision = 0;
    memcpy(qt.table, qt_natural, sizeof qt.table);
    int16_t du[64];
    memcpy(du, zz, sizeof du);
    __real_dequant_data_unit(&qt, du);
    fast_2didct(du, 0);
    memcpy(out64, du, sizeof du);
}

void ref_idct8(float *d) {
    fast_idct_new(d);
}

/* Colour-convert a single (Y,Cb,Cr) triple through ycbcr_rgb with 4:4:4
 * geometry, n pixels laid out as one row (n <= 65535/3). */
int ref_ycbcr_row(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int n, uint8_t *rgb) {
    FrameHeader fh;
    memset(&fh, 0, sizeof fh);
    Component cs[3];
    for (int i = 0; i < 3; i++) {
        cs[i].id = (uint8_t) (i + 1);
        cs[i].hsf = cs[i].vsf = 1;
        cs[i].x = (uint16_t) n;
        cs[i].y = 1;
        cs[i].qtid = 0;
    }
    fh.ncs = 3;
    fh.cs = cs;
    fh.X = (uint16_t) n;
    fh.Y = 1;
    int w = pad_to((uint16_t) n, 8);
    uint8_t *planes[3];
    int widths[3] = {w, w, w};
    const uint8_t *src[3] = {y, cb, cr};
    for (int i = 0; i < 3; i++) {
        planes[i] = (uint8_t *) calloc((size_t) w * 8, 1);
        memcpy(planes[i], src[i], (size_t) n);
    }
    Image img;
    img.buf = planes;
    img.width = widths;
    img.mcu = 0;
    img.n_du = 3;
    int rc = ycbcr_rgb(&fh, &img, rgb, (uint16_t) n, 1, (uint16_t) (3 * n));
    for (int i = 0; i < 3; i++) {
        free(planes[i]);
    }
    return rc;
}
