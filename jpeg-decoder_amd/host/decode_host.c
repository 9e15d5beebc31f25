/*
 * decode_host.c -- the reference's top-level entry points over the GPU path.
 *
 *   decode_jpeg_buffer   decode.c:138-424   marker walk on the host; at SOS it calls
 *   decode_scan          decode.c:535-663   which here uploads the entropy-coded bytes
 *                                           and runs the HIP kernels;
 *   allocate_img/free_img decode.c:95-136
 *   ycbcr_rgb / y_rgb    colour_conversion.c:5-83   the RGB the fused IDCT+colour kernel
 *                                           produced while the planes are as the decode
 *                                           left them; else planes -> RGB on the GPU.
 *
 * There is no CPU decode behind these names.  If no HIP device is usable they
 * return -1 and say so on stderr.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355jpeg.h"
#include "mj_host.h"

/* ---- process-wide default context and the Image -> batch side table ------ */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static mj_ctx *g_ctx = NULL;

/* Every live Image made by allocate_img: the bytes behind each of its planes (Image
 * itself records only the strides, decode.h:12-17) and, once a scan has been decoded
 * into it, the batch that holds its GPU results.  Grows as needed; guarded by g_lock. */
typedef struct {
    Image *img;
    mj_batch *batch;
    size_t plane_bytes[4];
    uint64_t plane_hash[4];    /* of the planes as the decode left them in Image.buf (valid while batch != NULL) */
} SideEntry;
static SideEntry *g_side = NULL;
static size_t g_side_n = 0, g_side_cap = 0;

/* How far decode_scan, called on its own, looks for the marker that ends its scan.  The default
 * is the largest entropy-coded segment the batch entry points accept. */
#define MJ_SCAN_LIMIT_DEFAULT (((size_t) 1 << 28) - 1)
static size_t g_scan_limit = MJ_SCAN_LIMIT_DEFAULT;

void mj_set_scan_limit(size_t bytes) {
    pthread_mutex_lock(&g_lock);
    g_scan_limit = bytes ? bytes : MJ_SCAN_LIMIT_DEFAULT;
    pthread_mutex_unlock(&g_lock);
}

/* end of the caller's buffer while decode_jpeg_buffer is on the stack */
static _Thread_local const uint8_t *t_buf_end = NULL;

static mj_ctx *default_ctx(void) {
    if (!g_ctx) {
        int dev = 0;
        const char *e = getenv("MJ_DEVICE");
        if (e) dev = atoi(e);
        if (mj_create(dev, &g_ctx) != 0) {
            fprintf(stderr, "mi355jpeg: %s\n", mj_last_error());
            g_ctx = NULL;
        }
    }
    return g_ctx;
}

static SideEntry *side_find(const Image *img) {
    for (size_t i = 0; i < g_side_n; i++) {
        if (g_side[i].img == img) return &g_side[i];
    }
    return NULL;
}

static SideEntry *side_add(Image *img) {
    if (g_side_n == g_side_cap) {
        const size_t cap = g_side_cap ? 2 * g_side_cap : 16;
        SideEntry *t = (SideEntry *) realloc(g_side, cap * sizeof *t);
        if (!t) return NULL;
        g_side = t;
        g_side_cap = cap;
    }
    SideEntry *e = &g_side[g_side_n++];
    memset(e, 0, sizeof *e);
    e->img = img;
    return e;
}

static void side_drop(Image *img) {
    SideEntry *e = side_find(img);
    if (!e) return;
    if (e->batch) mj_batch_destroy(e->batch);
    *e = g_side[--g_side_n];
}

/* A quick 64-bit digest of a plane: tells whether the caller has touched the samples between
 * the decode and the colour step (it then gets its samples converted, not the cached result). */
static uint64_t plane_digest(const uint8_t *p, size_t n) {
    uint64_t a = 0x9E3779B97F4A7C15ull ^ n, b = 0xC2B2AE3D27D4EB4Full;
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        uint64_t w0, w1;
        memcpy(&w0, p + i, 8);
        memcpy(&w1, p + i + 8, 8);
        a = (a ^ w0) * 0x9FB21C651E98DF25ull;
        a ^= a >> 32;
        b = (b ^ w1) * 0xD6E8FEB86659FD93ull;
        b ^= b >> 32;
    }
    for (; i < n; i++) a = (a ^ p[i]) * 0x100000001B3ull;
    return a ^ (b * 0x9E3779B97F4A7C15ull);
}

/* ---- Image (decode.c:95-136) -------------------------------------------- */
static size_t pad_to_mcu(size_t v, size_t unit) {
    return (v % unit) ? v + unit - (v % unit) : v;
}

Image *allocate_img(FrameHeader *fh) {
    if (!fh || !fh->cs || fh->ncs == 0 || fh->ncs > 4) return NULL;
    Image *img = (Image *) calloc(1, sizeof *img);
    if (!img) return NULL;
    size_t bytes[4] = {0, 0, 0, 0};
    img->buf = (uint8_t **) calloc(fh->ncs, sizeof *img->buf);
    img->width = (int *) calloc(fh->ncs, sizeof *img->width);
    if (!img->buf || !img->width) goto fail;
    for (int i = 0; i < fh->ncs; i++) {
        const Component *c = &fh->cs[i];
        if (c->hsf == 0 || c->vsf == 0) goto fail;
        /* each plane is padded to whole MCUs */
        const size_t w = pad_to_mcu(c->x, 8u * c->hsf), h = pad_to_mcu(c->y, 8u * c->vsf);
        bytes[i] = (w && h) ? w * h : 1;
        img->buf[i] = (uint8_t *) calloc(bytes[i], 1);
        if (!img->buf[i]) goto fail;
        img->width[i] = (int) w;
    }
    img->n_du = fh->ncs;
    img->mcu = 0;
    pthread_mutex_lock(&g_lock);
    SideEntry *e = side_add(img);
    if (e) memcpy(e->plane_bytes, bytes, sizeof bytes);
    pthread_mutex_unlock(&g_lock);
    if (!e) goto fail;
    return img;
fail:
    if (img->buf) {
        for (int i = 0; i < fh->ncs; i++) free(img->buf[i]);
    }
    free(img->buf);
    free(img->width);
    free(img);
    return NULL;
}

void free_img(Image *img) {
    if (!img) return;
    pthread_mutex_lock(&g_lock);
    side_drop(img);
    pthread_mutex_unlock(&g_lock);
    for (int i = 0; i < img->n_du; i++) free(img->buf[i]);
    free(img->buf);
    free(img->width);
    free(img);
}

/* ---- the SOS seam (decode.c:535-663) ------------------------------------ */
int decode_scan(Bitstream *bs, Image *img, FrameHeader *fh, ScanHeader *sh, HuffTables *hts, QuantTables *qts,
                RestartInterval ri) {
    if (!bs || !img || !fh || !sh || !hts || !qts) return -1;
    const uint8_t *ent = bs->encoded_data;
    size_t ent_len;
    if (t_buf_end) {
        if (t_buf_end <= ent) return -1;   /* the file ends with its scan header */
        ent_len = (size_t) (t_buf_end - ent);
    } else {
        /* called on its own: the Bitstream carries no length (bitstream.h:4-8), so, like the
         * reference, trust the stream to end in a marker -- but look for it no further than
         * the limit (mj_set_scan_limit) */
        pthread_mutex_lock(&g_lock);
        const size_t limit = g_scan_limit;
        pthread_mutex_unlock(&g_lock);
        const uint8_t *p = ent, *stop = ent + limit;
        for (;;) {
            p = (p < stop) ? (const uint8_t *) memchr(p, 0xFF, (size_t) (stop - p)) : NULL;
            if (!p) {
                fprintf(stderr, "mi355jpeg: decode_scan: no marker ends the scan within %zu bytes (mj_set_scan_limit)\n", limit);
                return -1;
            }
            if (p[1] != 0x00 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7)) break;
            p += 1;
        }
        ent_len = (size_t) (p - ent) + 2;
    }
    int rc = -1;
    pthread_mutex_lock(&g_lock);
    mj_ctx *ctx = default_ctx();
    mj_batch *b = NULL;
    if (!ctx) goto out;
    if (mj_batch_create_scan(ctx, ent, ent_len, fh, sh, hts, qts, ri, MJ_WANT_PLANES, &b) != 0 ||
        mj_batch_decode(b) != 0) {
        fprintf(stderr, "mi355jpeg: %s\n", mj_last_error());
        goto out;
    }
    mj_image_info inf;
    if (mj_batch_image_info(b, 0, &inf) != 0) goto out;
    SideEntry *se = side_find(img);
    if (!se) {
        fprintf(stderr, "mi355jpeg: decode_scan needs an Image made by allocate_img (its plane sizes are unknown otherwise)\n");
        goto out;
    }
    /* fill the caller's planes (write_data_unit, decode.c:517-531): same stride and no
     * more bytes than the plane was allocated with */
    for (int c = 0; c < inf.ncs && c < img->n_du; c++) {
        if (img->width[c] != inf.plane_w[c] || (size_t) inf.plane_w[c] * (size_t) inf.plane_h[c] > se->plane_bytes[c]) {
            fprintf(stderr, "mi355jpeg: Image plane %d (stride %d, %zu bytes) does not fit the frame's %d x %d plane\n", c,
                    img->width[c], se->plane_bytes[c], inf.plane_w[c], inf.plane_h[c]);
            goto out;
        }
        if (mj_batch_copy_plane(b, 0, c, img->buf[c]) != 0) {
            fprintf(stderr, "mi355jpeg: %s\n", mj_last_error());
            goto out;
        }
        se->plane_hash[c] = plane_digest(img->buf[c], se->plane_bytes[c]);
    }
    size_t end_off = 0;
    if (mj_batch_scan_end(b, 0, &end_off) != 0) goto out;
    bs->encoded_data = (uint8_t *) ent + end_off;
    bs->offset = 0;
    int64_t mcus = (int64_t) inf.mcus_x * inf.mcus_y;
    img->mcu = (uint16_t) (mcus > 65535 ? 65535 : mcus);
    if (se->batch) mj_batch_destroy(se->batch);   /* a second scan into the same Image */
    se->batch = b;
    b = NULL; /* now owned by the side table */
    rc = 0;
out:
    if (b) mj_batch_destroy(b);
    pthread_mutex_unlock(&g_lock);
    return rc;
}

/* A frame coded in several scans (one per component, or one component alone and two together): the reference would re-enter decode_scan for every
 * SOS (decode.c:187-201, :333) and write past its planes from the second on (its MCU counter
 * runs on, decode.c:125, :473, :504).  Here the whole file goes to the batch entry point once,
 * which knows such files; the planes of the three scans land in the Image (rows at the Image's
 * own stride, which is padded to whole MCUs where the scans' planes are padded to whole blocks). */
static int decode_all_scans(const uint8_t *buf, size_t len, Image *img) {
    int rc = -1;
    pthread_mutex_lock(&g_lock);
    mj_ctx *ctx = default_ctx();
    mj_batch *b = NULL;
    uint8_t *tmp = NULL;
    if (!ctx) goto out;
    if (mj_batch_create(ctx, &buf, &len, 1, MJ_WANT_PLANES, &b) != 0 || mj_batch_decode(b) != 0) {
        fprintf(stderr, "mi355jpeg: %s\n", mj_last_error());
        goto out;
    }
    mj_image_info inf;
    if (mj_batch_image_info(b, 0, &inf) != 0) goto out;
    SideEntry *se = side_find(img);
    if (!se) goto out;
    for (int c = 0; c < inf.ncs && c < img->n_du; c++) {
        const size_t pw = (size_t) inf.plane_w[c], ph = (size_t) inf.plane_h[c], stride = (size_t) img->width[c];
        if (pw > stride || stride * ph > se->plane_bytes[c]) {
            fprintf(stderr, "mi355jpeg: Image plane %d (stride %zu, %zu bytes) does not fit the scan's %zu x %zu plane\n", c, stride,
                    se->plane_bytes[c], pw, ph);
            goto out;
        }
        uint8_t *t = (uint8_t *) realloc(tmp, (pw * ph) != 0 ? pw * ph : 1);
        if (!t) goto out;
        tmp = t;
        if (mj_batch_copy_plane(b, 0, c, tmp) != 0) {
            fprintf(stderr, "mi355jpeg: %s\n", mj_last_error());
            goto out;
        }
        for (size_t y = 0; y < ph; y++) memcpy(img->buf[c] + y * stride, tmp + y * pw, pw);
        se->plane_hash[c] = plane_digest(img->buf[c], se->plane_bytes[c]);
    }
    {
        int64_t mcus = (int64_t) inf.mcus_x * inf.mcus_y;
        img->mcu = (uint16_t) (mcus > 65535 ? 65535 : mcus);
    }
    if (se->batch) mj_batch_destroy(se->batch);
    se->batch = b;
    b = NULL;
    rc = 0;
out:
    free(tmp);
    if (b) mj_batch_destroy(b);
    pthread_mutex_unlock(&g_lock);
    return rc;
}

/* ---- marker walk (decode.c:138-424) ------------------------------------- */
int decode_jpeg_buffer(uint8_t *buf, size_t len, FrameHeader *fh, Image **img) {
    if (!fh) {
        printf("Frame Header provided is NULL\n"); /* decode.c:141-144 */
        return -1;
    }
    if (!buf || !img || len < 4) return -1;
    ScanHeader sh;
    HuffTables hts;
    QuantTables qts;
    RestartInterval ri = 0;
    int have_hts = 0, have_qts = 0, have_fh = 0, rc = 0;
    Image *im = NULL;
    memset(&qts, 0, sizeof qts);
    memset(&hts, 0, sizeof hts);
    fh->cs = NULL;

    Bitstream bs = {buf, 0, 0};
    const uint8_t *end = buf + len;
    t_buf_end = end;
    while (bs.encoded_data + 1 < end) {
        if (next_byte(&bs) != 0xFF) continue;
        uint8_t m = get_byte(&bs);
        if (m == 0xFF) continue;
        next_byte(&bs);
        if (m == 0x00 || m == 0x01 || m == 0xD8 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) break;
        if (bs.encoded_data + 2 > end) { rc = -1; break; }
        uint8_t *seg = bs.encoded_data;
        size_t seglen = ((size_t) seg[0] << 8) | seg[1];
        if (seglen < 2 || seg + seglen > end) { rc = -1; break; }
        int advance = 1;
        switch (m) {
            case 0xC0:
            case 0xC1:
                if (have_fh) free_frame_header(fh);
                have_fh = 0;
                if (decode_frame_header(m == 0xC0 ? BDCT : ESDCTHC, &bs, fh) != 0) { rc = -1; break; }
                have_fh = 1;
                print_frame_header(fh);
                /* the planes always match the frame header in force (the reference keeps the
                 * Image of the first SOF, decode.c:179-185, whatever a later one says) */
                if (im) free_img(im);
                im = allocate_img(fh);
                if (!im) rc = -1;
                break;
            case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB:
            case 0xCD: case 0xCE: case 0xCF:
                fprintf(stderr, "mi355jpeg: only baseline sequential Huffman JPEG is decoded (SOF marker FF%02X)\n", m);
                rc = -1;
                break;
            case 0xC4:
                if (!have_hts) {
                    new_huff_tables(have_fh ? fh->process : (Encoding) -1, &hts);
                    have_hts = 1;
                }
                if (decode_huff_tables(&bs, &hts) != 0) rc = -1;
                break;
            case 0xDB:
                if (decode_quant_table(&bs, &qts) != 0) { rc = -1; break; }
                have_qts = 1;
                break;
            case 0xDC:
                if (have_fh) decode_number_of_lines(&bs, fh);
                break;
            case 0xDD:
                if (decode_restart_interval(&bs, &ri) != 0) rc = -1;
                break;
            case 0xDA:
                if (decode_scan_header(&bs, &sh) != 0) { rc = -1; break; }
                print_scan_header(&sh);
                if (!have_qts || !have_hts || !have_fh || !im) { rc = -1; break; }
                if (sh.nics < fh->ncs && fh->ncs == 3) {
                    /* the frame takes several scans: all of them at once, and the walk is over */
                    if (decode_all_scans(buf, len, im) != 0) rc = -1;
                    bs.encoded_data = (uint8_t *) end;
                    advance = 0;
                    break;
                }
                bs.encoded_data = seg + seglen;
                bs.offset = 0;
                if (decode_scan(&bs, im, fh, &sh, &hts, &qts, ri) != 0) rc = -1;
                advance = 0; /* decode_scan left the cursor on the marker that ended the scan */
                break;
            default:
                break;
        }
        if (rc != 0) break;
        if (advance) {
            bs.encoded_data = seg + seglen;
            bs.offset = 0;
        }
    }
    t_buf_end = NULL;
    if (have_hts) free_huff_tables(&hts);
    if (rc != 0 || !im) {
        if (im) free_img(im);
        if (have_fh) free_frame_header(fh);
        return -1;
    }
    *img = im;
    return 0;
}

/* ---- colour (colour_conversion.c:5-83, :164-180) ------------------------ */
int get_max_sf(FrameHeader *fh, uint8_t *max_hsf, uint8_t *max_vsf) {
    if (!fh || fh->ncs == 0) return -1;
    uint8_t h = 0, v = 0;
    for (int i = 0; i < fh->ncs; i++) {
        if (fh->cs[i].hsf > h) h = fh->cs[i].hsf;
        if (fh->cs[i].vsf > v) v = fh->cs[i].vsf;
    }
    *max_hsf = h;
    *max_vsf = v;
    return 0;
}

/* The row stride the reference's colour loops give a component's plane: its width padded to
 * whole MCUs (colour_conversion.c:46-47). */
static size_t frame_stride(const Component *c) {
    return pad_to_mcu(c->x, 8u * c->hsf);
}

/* Does the result the decode kept for this Image still stand for what the caller asks: the same
 * frame geometry in `fh`, the whole frame, and planes that hold what the decode put there? */
static int cached_rgb_is_current(const SideEntry *se, const FrameHeader *fh, const Image *img, uint16_t width, uint16_t height, int ncs) {
    mj_image_info inf;
    if (!se || !se->batch || mj_batch_image_info(se->batch, 0, &inf) != 0) return 0;
    if (inf.ncs != ncs || fh->ncs != ncs || width != inf.X || height != inf.Y) return 0;
    for (int c = 0; c < ncs; c++) {
        if (fh->cs[c].hsf != inf.hsf[c] || fh->cs[c].vsf != inf.vsf[c]) return 0;
        if ((size_t) img->width[c] != frame_stride(&fh->cs[c])) return 0;
    }
    for (int c = 0; c < ncs; c++) {
        if (plane_digest(img->buf[c], se->plane_bytes[c]) != se->plane_hash[c]) return 0;
    }
    return 1;
}

/* ycbcr_rgb / y_rgb (colour_conversion.c:30-83, :5-28): the samples in Image.buf, at the strides
 * and sampling factors `fh` gives, to RGB24 at `pitch`.  An Image that this library's decode
 * filled and nobody has touched since gets the RGB the fused kernel made at decode time; any
 * other (planes the caller filled or edited, an Image not made by allocate_img, another frame
 * header, a window of the frame) goes through the planes -> RGB kernel. */
static int colour_out(FrameHeader *fh, Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch, int ncs) {
    if (!fh || !fh->cs || !img || !img->buf || !img->width || !pixels) return -1;
    if (fh->ncs < ncs || width == 0 || height == 0) return -1;
    int rc = -1;
    pthread_mutex_lock(&g_lock);
    SideEntry *se = side_find(img);
    if (cached_rgb_is_current(se, fh, img, width, height, ncs)) {
        rc = mj_batch_copy_rgb(se->batch, 0, pixels, pitch);
    } else {
        mj_ctx *ctx = default_ctx();
        const uint8_t *planes[3] = {NULL, NULL, NULL};
        int32_t strides[3] = {0, 0, 0}, hsf[3] = {1, 1, 1}, vsf[3] = {1, 1, 1};
        for (int c = 0; c < ncs; c++) {
            planes[c] = img->buf[c];
            strides[c] = (int32_t) frame_stride(&fh->cs[c]);
            hsf[c] = fh->cs[c].hsf;
            vsf[c] = fh->cs[c].vsf;
        }
        if (ctx) rc = mj_planes_to_rgb(ctx, ncs, planes, strides, hsf, vsf, width, height, pixels, pitch);
    }
    if (rc != 0) fprintf(stderr, "mi355jpeg: %s\n", mj_last_error());
    pthread_mutex_unlock(&g_lock);
    return rc;
}

int ycbcr_rgb(FrameHeader *fh, Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch) {
    return colour_out(fh, img, pixels, width, height, pitch, 3);
}

/* One component: R = G = B.  The plane is read at its own stride (the reference strides it by
 * `width`, colour_conversion.c:20, which shears every frame whose width is not a multiple of 8). */
int y_rgb(FrameHeader *fh, Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch) {
    return colour_out(fh, img, pixels, width, height, pitch, 1);
}

int yccb_rgb(FrameHeader *fh, Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch) {
    (void) fh;
    (void) img;
    (void) pixels;
    (void) width;
    (void) height;
    (void) pitch;
    fprintf(stderr, "mi355jpeg: four-component (YCCK) frames are not decoded by the GPU path\n");
    return -1;
}
