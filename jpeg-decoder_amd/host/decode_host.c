/*
 * decode_host.c -- the reference's top-level entry points over the GPU path.
 *
 *   decode_jpeg_buffer   decode.c:138-424   marker walk on the host; at SOS it calls
 *   decode_scan          decode.c:535-663   which here uploads the entropy-coded bytes
 *                                           and runs the HIP kernels;
 *   allocate_img/free_img decode.c:95-136
 *   ycbcr_rgb / y_rgb    colour_conversion.c:5-83   copy out the RGB the fused
 *                                           IDCT+colour kernel already produced.
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

#define MJ_SIDE_SLOTS 64
static struct {
    Image *img;
    mj_batch *batch;
} g_side[MJ_SIDE_SLOTS];

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

static int side_put(Image *img, mj_batch *b) {
    for (int i = 0; i < MJ_SIDE_SLOTS; i++) {
        if (g_side[i].img == img) {     /* a second scan into the same Image */
            mj_batch_destroy(g_side[i].batch);
            g_side[i].batch = b;
            return 0;
        }
    }
    for (int i = 0; i < MJ_SIDE_SLOTS; i++) {
        if (!g_side[i].img) {
            g_side[i].img = img;
            g_side[i].batch = b;
            return 0;
        }
    }
    return -1;
}

static mj_batch *side_get(Image *img) {
    for (int i = 0; i < MJ_SIDE_SLOTS; i++) {
        if (g_side[i].img == img) return g_side[i].batch;
    }
    return NULL;
}

static void side_drop(Image *img) {
    for (int i = 0; i < MJ_SIDE_SLOTS; i++) {
        if (g_side[i].img == img) {
            mj_batch_destroy(g_side[i].batch);
            g_side[i].img = NULL;
            g_side[i].batch = NULL;
        }
    }
}

/* ---- Image (decode.c:95-136) -------------------------------------------- */
static uint16_t pad_to_mcu(uint16_t v, int unit) {
    return (uint16_t) ((v % unit) ? v + unit - (v % unit) : v);
}

Image *allocate_img(FrameHeader *fh) {
    if (!fh || !fh->cs || fh->ncs == 0) return NULL;
    Image *img = (Image *) calloc(1, sizeof *img);
    if (!img) return NULL;
    img->buf = (uint8_t **) calloc(fh->ncs, sizeof *img->buf);
    img->width = (int *) calloc(fh->ncs, sizeof *img->width);
    if (!img->buf || !img->width) goto fail;
    for (int i = 0; i < fh->ncs; i++) {
        const Component *c = &fh->cs[i];
        if (c->hsf == 0 || c->vsf == 0) goto fail;
        /* each plane is padded to whole MCUs */
        size_t w = pad_to_mcu(c->x, 8 * c->hsf), h = pad_to_mcu(c->y, 8 * c->vsf);
        img->buf[i] = (uint8_t *) calloc((w && h) ? w * h : 1, 1);
        if (!img->buf[i]) goto fail;
        img->width[i] = (int) w;
        img->n_du = (uint8_t) (i + 1);
    }
    img->n_du = fh->ncs;
    img->mcu = 0;
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
    if (t_buf_end && t_buf_end > ent) {
        ent_len = (size_t) (t_buf_end - ent);
    } else {
        /* called on its own: like the reference, trust the stream to end in a marker */
        const uint8_t *p = ent;
        for (;;) {
            p = (const uint8_t *) memchr(p, 0xFF, (size_t) 1 << 30);
            if (!p) return -1;
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
    /* fill the caller's planes (write_data_unit, decode.c:517-531) */
    for (int c = 0; c < inf.ncs && c < img->n_du; c++) {
        if (img->width[c] != inf.plane_w[c]) {
            fprintf(stderr, "mi355jpeg: Image plane %d has stride %d, expected %d\n", c, img->width[c], inf.plane_w[c]);
            goto out;
        }
        if (mj_batch_copy_plane(b, 0, c, img->buf[c]) != 0) {
            fprintf(stderr, "mi355jpeg: %s\n", mj_last_error());
            goto out;
        }
    }
    size_t end_off = 0;
    if (mj_batch_scan_end(b, 0, &end_off) != 0) goto out;
    bs->encoded_data = (uint8_t *) ent + end_off;
    bs->offset = 0;
    int64_t mcus = (int64_t) inf.mcus_x * inf.mcus_y;
    img->mcu = (uint16_t) (mcus > 65535 ? 65535 : mcus);
    if (side_put(img, b) != 0) {
        fprintf(stderr, "mi355jpeg: too many live Images (free_img releases them)\n");
        goto out;
    }
    b = NULL; /* now owned by the side table */
    rc = 0;
out:
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
                if (decode_frame_header(m == 0xC0 ? BDCT : ESDCTHC, &bs, fh) != 0) { rc = -1; break; }
                have_fh = 1;
                print_frame_header(fh);
                if (!im) {
                    im = allocate_img(fh);
                    if (!im) rc = -1;
                }
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
                decode_quant_table(&bs, &qts);
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

static int copy_rgb_out(Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch, int want_ncs) {
    if (!img || !pixels) return -1;
    int rc = -1;
    pthread_mutex_lock(&g_lock);
    mj_batch *b = side_get(img);
    mj_image_info inf;
    if (!b) {
        fprintf(stderr, "mi355jpeg: Image was not produced by this library's decode_scan (no GPU result to convert)\n");
    } else if (mj_batch_image_info(b, 0, &inf) == 0 && inf.ncs == want_ncs) {
        if (width == inf.X && height == inf.Y) {
            rc = mj_batch_copy_rgb(b, 0, pixels, pitch);
        } else if (width <= inf.X && height <= inf.Y && (size_t) pitch >= (size_t) 3 * width) {
            /* a caller-chosen window of the frame: copy through a full-size staging image */
            uint8_t *tmp = (uint8_t *) malloc((size_t) 3 * inf.X * inf.Y);
            if (tmp && mj_batch_copy_rgb(b, 0, tmp, (size_t) 3 * inf.X) == 0) {
                for (int y = 0; y < height; y++) memcpy(pixels + (size_t) y * pitch, tmp + (size_t) y * 3 * inf.X, (size_t) 3 * width);
                rc = 0;
            }
            free(tmp);
        }
        if (rc != 0) fprintf(stderr, "mi355jpeg: %s\n", mj_last_error());
    }
    pthread_mutex_unlock(&g_lock);
    return rc;
}

int ycbcr_rgb(FrameHeader *fh, Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch) {
    (void) fh;
    return copy_rgb_out(img, pixels, width, height, pitch, 3);
}

int y_rgb(FrameHeader *fh, Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch) {
    (void) fh;
    return copy_rgb_out(img, pixels, width, height, pitch, 1);
}
