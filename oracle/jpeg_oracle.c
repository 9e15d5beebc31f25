/*
 * jpeg_oracle.c -- TEST INFRASTRUCTURE, not product code.  See jpeg_oracle.h.
 *
 * Every routine cites the reference lines (relative to /root/reference/src) whose
 * behaviour it restates.  Written from the behaviour, not from the text: the
 * structure (parse once into a descriptor, decode the scan by MCU count, then run
 * the pixel stages over whole planes) is this file's own.
 *
 * Deliberate differences from the reference, all on inputs where the reference
 * itself misbehaves (none of them is reachable from the parity fixtures):
 *   - the scan ends after the frame's MCU count, not at the next marker
 *     (decode.c:578-589 never counts MCUs and can drop a final MCU that starts in
 *     the last entropy byte);
 *   - the MCU counter is 32-bit (decode.h:15 keeps it in a uint16_t and misplaces
 *     blocks past 65,535 MCUs);
 *   - grey output reads the padded plane with its real stride
 *     (colour_conversion.c:20 strides by the image width).
 *
 * Files whose three-component frame is coded in several scans (one component each,
 * or one alone and two together: T.81 A.2.2, A.2.3, B.2.3) are decoded by
 * orc_run_scans below.  PARITY UNPINNED for that path: the reference re-enters its
 * scan loop without resetting its MCU counter (decode.c:125, :473, :504) and writes
 * past its planes, so it has no output to be equal to; the tests pin this path
 * instead against the interleaved coding of the same coefficients, which the
 * reference does decode.
 *
 * Build with -ffp-contract=off: the float/double rounding points below are the
 * reference's (gcc -std=c17 -O2 on x86-64, FLT_EVAL_METHOD 0, no FMA contraction).
 */
#define _POSIX_C_SOURCE 199309L
#include "jpeg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* zigzag position -> natural (row-major) index; quant_table.c:25-34 */
static const uint8_t ZZ[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* ------------------------------------------------------------------------ */
/* parsed file                                                              */
/* ------------------------------------------------------------------------ */
typedef struct {
    /* canonical code ranges per length, huff_table.c:187-216 */
    int32_t min_code[16], max_code[16];
    uint8_t count[16];
    uint8_t sym[16][256];
    int defined;
} OHuff;

typedef struct {
    int id, hsf, vsf, qtid;
    int x, y;          /* component sample dims, frame_header.c:49-56 */
    int pw, ph;        /* plane dims padded to whole MCUs, decode.c:108-115 */
    int dc, ac;        /* table selectors from the scan header */
} OComp;

typedef struct {
    int X, Y, ncs, precision;
    int hmax, vmax;
    OComp c[4];
    uint16_t qt[4][64]; /* natural order, quant_table.c:108-114 */
    int qt_defined[4];
    OHuff ht[2][4];     /* [class][id] */
    int ri;             /* restart_interval.c:21-22 */
    int nscan;          /* components in the scan, in scan order */
    int scan_comp[4];   /* index into c[] */
    const uint8_t *ent; /* first entropy-coded byte */
    size_t ent_len;     /* bytes available from ent to the end of the buffer */
} OJpeg;

static int be16(const uint8_t *p) {
    return (p[0] << 8) | p[1];
}

/* DHT, huff_table.c:165-217.  Empty lengths get min=max=-1 so the compare
 * "max >= code" (decode.c:675) can never match them. */
static int parse_dht(OJpeg *j, const uint8_t *p, int len) {
    const uint8_t *end = p + len;
    while (p < end) {
        int cls = (p[0] >> 4) & 0xF, id = p[0] & 0xF;
        p++;
        if (cls > 1 || id > 3 || p + 16 > end) {
            return -1;
        }
        OHuff *h = &j->ht[cls][id];
        memcpy(h->count, p, 16);
        p += 16;
        int code = 0;
        for (int l = 0; l < 16; l++) {
            int n = h->count[l];
            if (n == 0) {
                code <<= 1;
                h->min_code[l] = h->max_code[l] = -1;
                continue;
            }
            if (p + n > end) {
                return -1;
            }
            memcpy(h->sym[l], p, (size_t) n);
            p += n;
            h->min_code[l] = code;
            code += n;
            h->max_code[l] = code - 1;
            code <<= 1;
        }
        h->defined = 1;
    }
    return 0;
}

/* DQT, quant_table.c:91-129: entries arrive in zigzag order, stored natural. */
static int parse_dqt(OJpeg *j, const uint8_t *p, int len) {
    const uint8_t *end = p + len;
    while (p < end) {
        int prec = (p[0] >> 4) & 0xF, id = p[0] & 0xF;
        p++;
        if (id > 3 || p + 64 * (prec ? 2 : 1) > end) {
            return -1;
        }
        for (int k = 0; k < 64; k++) {
            if (prec) {
                j->qt[id][ZZ[k]] = (uint16_t) be16(p);
                p += 2;
            } else {
                j->qt[id][ZZ[k]] = *p++;
            }
        }
        j->qt_defined[id] = 1;
    }
    return 0;
}

static int pad_up(int v, int unit) {
    return (v % unit) ? v + unit - (v % unit) : v;
}

/* SOF0/SOF1, frame_header.c:9-58 */
static int parse_sof(OJpeg *j, const uint8_t *p, int len) {
    if (len < 6) {
        return -1;
    }
    j->precision = p[0];
    j->Y = be16(p + 1);
    j->X = be16(p + 3);
    j->ncs = p[5];
    if (j->ncs == 0 || j->ncs > 4 || len != 6 + 3 * j->ncs) {
        return -1;
    }
    j->hmax = j->vmax = 0;
    for (int i = 0; i < j->ncs; i++) {
        OComp *c = &j->c[i];
        c->id = p[6 + 3 * i];
        c->hsf = p[7 + 3 * i] >> 4;
        c->vsf = p[7 + 3 * i] & 0xF;
        c->qtid = p[8 + 3 * i];
        if (c->hsf > j->hmax) j->hmax = c->hsf;
        if (c->vsf > j->vmax) j->vmax = c->vsf;
    }
    if (j->hmax == 0 || j->vmax == 0) {
        return -1;
    }
    for (int i = 0; i < j->ncs; i++) {
        OComp *c = &j->c[i];
        /* float ratio then ceil, exactly as frame_header.c:52-54 */
        c->x = (uint16_t) ceil(j->X * ((float) c->hsf / j->hmax));
        c->y = (uint16_t) ceil(j->Y * ((float) c->vsf / j->vmax));
        c->pw = pad_up(c->x, 8 * c->hsf);
        c->ph = pad_up(c->y, 8 * c->vsf);
    }
    return 0;
}

/* SOS header, scan_header.c:10-35 */
static int parse_sos(OJpeg *j, const uint8_t *p, int len) {
    if (len < 4) {
        return -1;
    }
    j->nscan = p[0];
    if (j->nscan == 0 || j->nscan > 4 || len < 1 + 2 * j->nscan + 3) {
        return -1;
    }
    for (int i = 0; i < j->nscan; i++) {
        int sc = p[1 + 2 * i];
        int k;
        for (k = 0; k < j->ncs; k++) {
            if (j->c[k].id == sc) break;   /* decode.c:595-604 */
        }
        if (k == j->ncs) {
            return -1;
        }
        j->scan_comp[i] = k;
        j->c[k].dc = p[2 + 2 * i] >> 4;
        j->c[k].ac = p[2 + 2 * i] & 0xF;
    }
    return 0;
}

/* Marker walk, decode.c:138-424.  Stops at the first SOS of a baseline frame and
 * records where its entropy-coded bytes begin. */
static int orc_parse(const uint8_t *buf, size_t len, OJpeg *j) {
    memset(j, 0, sizeof *j);
    size_t pos = 0;
    int have_sof = 0;
    while (pos + 1 < len) {
        if (buf[pos] != 0xFF) {
            pos++;
            continue;
        }
        int m = buf[pos + 1];
        pos += 2;
        if (m == 0xD8 || m == 0x01 || m == 0xFF || (m >= 0xD0 && m <= 0xD7)) {
            if (m == 0xFF) pos--;   /* fill byte */
            continue;
        }
        if (m == 0xD9) {
            break;
        }
        if (pos + 2 > len) {
            return -1;
        }
        int seglen = be16(buf + pos);
        if (seglen < 2 || pos + (size_t) seglen > len) {
            return -1;
        }
        const uint8_t *body = buf + pos + 2;
        int blen = seglen - 2;
        switch (m) {
            case 0xC0:
            case 0xC1:
                if (parse_sof(j, body, blen)) return -1;
                have_sof = 1;
                break;
            case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7:
            case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return -1;      /* not baseline: outside this path */
            case 0xC4:
                if (parse_dht(j, body, blen)) return -1;
                break;
            case 0xDB:
                if (parse_dqt(j, body, blen)) return -1;
                break;
            case 0xDD:
                if (blen != 2) return -1;
                j->ri = be16(body);
                break;
            case 0xDA:
                if (!have_sof || parse_sos(j, body, blen)) return -1;
                j->ent = buf + pos + seglen;
                j->ent_len = len - (pos + (size_t) seglen);
                return 0;
            default:
                break;          /* APPn, COM, DNL...: skipped by length */
        }
        pos += (size_t) seglen;
    }
    return -1;
}

/* ------------------------------------------------------------------------ */
/* entropy decoding                                                         */
/* ------------------------------------------------------------------------ */
/* Bit cursor with the reference's semantics on well-formed data
 * (bitstream.c:22-78): MSB first; when a byte is used up the cursor moves on
 * eagerly, and a data FF's stuffed 00 is stepped over. */
typedef struct {
    const uint8_t *p, *end;
    int off;
} OBits;

static void obits_advance(OBits *b) {
    if (b->p + 1 < b->end && b->p[0] == 0xFF && b->p[1] == 0x00) {
        b->p += 2;
    } else {
        b->p += 1;
    }
    b->off = 0;
}

static int obit(OBits *b) {
    int v = (b->p < b->end) ? ((*b->p >> (7 - b->off)) & 1) : 1;
    if (++b->off >= 8) {
        obits_advance(b);
    }
    return v;
}

static int obits_n(OBits *b, int n) {
    int v = 0;
    for (int i = 0; i < n; i++) {
        v = (v << 1) | obit(b);
    }
    return v;
}

/* canonical walk, decode.c:672-681 / 693-702 */
static int ohuff_symbol(OBits *b, const OHuff *h) {
    int code = obit(b);
    for (int l = 0; l < 16; l++) {
        if (h->max_code[l] >= code) {
            return h->sym[l][(code - h->min_code[l]) & 0xFF];
        }
        code = (code << 1) | obit(b);
    }
    return 0;
}

/* EXTEND, decode.c:684-686 / 716-718; size 0 yields 0 */
static int oextend(int v, int size) {
    if (size == 0) {
        return 0;
    }
    return (v < (1 << (size - 1))) ? v - (1 << size) + 1 : v;
}

/* One 8x8 block, decode.c:665-723.  du is pre-zeroed by the caller
 * (decode.c:612); coefficients land at their zigzag position. */
static void oblock(OBits *b, const OHuff *dc, const OHuff *ac, int16_t *pred, int16_t *du) {
    int s = ohuff_symbol(b, dc);
    int diff = oextend(obits_n(b, s), s);
    *pred = (int16_t) (diff + *pred);       /* int16 wrap, decode.c:688-690 */
    du[0] = *pred;
    for (int k = 1; k < 64; k++) {
        int rs = ohuff_symbol(b, ac);
        int size = rs & 15, run = rs >> 4;
        k += run;
        if (rs == 0x00) {
            break;
        }
        if (rs == 0xF0) {
            continue;
        }
        int v = oextend(obits_n(b, size), size);
        if (k < 64) {
            du[k] = (int16_t) v;
        }
    }
}

/* At an interval boundary the encoder pads to a byte boundary and emits FF Dn
 * (decode.c:580-584, bitstream.c:142-176).  Move to the byte after it. */
static void orestart(OBits *b) {
    if (b->off != 0) {
        obits_advance(b);
    }
    while (b->p + 1 < b->end) {
        if (b->p[0] == 0xFF && b->p[1] >= 0xD0 && b->p[1] <= 0xD7) {
            b->p += 2;
            break;
        }
        if (b->p[0] == 0xFF && b->p[1] != 0x00) {
            break;              /* some other marker: leave it */
        }
        b->p++;
    }
    b->off = 0;
}

static int64_t orc_nblocks(const OJpeg *j, int *mcus_x, int *mcus_y) {
    const OComp *c0 = &j->c[j->scan_comp[0]];
    int bpm = 0;
    for (int i = 0; i < j->nscan; i++) {
        const OComp *c = &j->c[j->scan_comp[i]];
        bpm += c->hsf * c->vsf;
    }
    *mcus_x = c0->pw / (8 * c0->hsf);
    *mcus_y = c0->ph / (8 * c0->vsf);
    return (int64_t) bpm * *mcus_x * *mcus_y;
}

/* The MCU loop, decode.c:535-663: components in scan order, then v, then h. */
static int orc_entropy(const OJpeg *j, int16_t *coefs) {
    int mx, my;
    int64_t nb = orc_nblocks(j, &mx, &my);
    memset(coefs, 0, (size_t) nb * 64 * sizeof(int16_t));
    OBits b = {j->ent, j->ent + j->ent_len, 0};
    int16_t pred[4] = {0, 0, 0, 0};
    int64_t blk = 0;
    int64_t nmcu = (int64_t) mx * my;
    for (int64_t m = 0; m < nmcu; m++) {
        if (j->ri && m && (m % j->ri) == 0) {
            orestart(&b);
            memset(pred, 0, sizeof pred);   /* decode.c:1289-1293 */
        }
        for (int i = 0; i < j->nscan; i++) {
            const OComp *c = &j->c[j->scan_comp[i]];
            const OHuff *dc = &j->ht[0][c->dc & 3], *ac = &j->ht[1][c->ac & 3];
            for (int n = 0; n < c->hsf * c->vsf; n++) {
                oblock(&b, dc, ac, &pred[i], coefs + blk * 64);
                blk++;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* dequant + IDCT                                                           */
/* ------------------------------------------------------------------------ */
/* 8-point inverse butterfly, dct.c:296-341.  Float storage, double constants:
 * (float)(C * (double)x) marks a product rounded to double and then to float. */
void orc_idct8(float *d) {
    float t0 = (float) (1.414213562 * (double) d[0]);
    float t1 = d[4], t2 = d[2], t3 = d[6];
    float t4 = (float) (0.5 * (double) (d[1] - d[7]));
    float t5 = (float) (0.707106781 * (double) d[3]);
    float t6 = (float) (0.707106781 * (double) d[5]);
    float t7 = (float) (0.5 * (double) (d[1] + d[7]));

    float u0 = (float) (0.5 * (double) (t0 + t1));
    float u1 = (float) (0.5 * (double) (t0 - t1));
    float u2 = (float) (0.707106781 * (0.38268343236 * (double) t2 + -0.92387953251 * (double) t3));
    float u3 = (float) (0.707106781 * (0.92387953251 * (double) t2 + 0.38268343236 * (double) t3));
    float u4 = (float) (0.5 * (double) (t4 + t6));
    float u5 = (float) (0.5 * (double) (-t5 + t7));
    float u6 = (float) (0.5 * (double) (t4 - t6));
    float u7 = (float) (0.5 * (double) (t5 + t7));

    float w0 = (float) (0.5 * (double) (u0 + u3));
    float w1 = (float) (0.5 * (double) (u1 + u2));
    float w2 = (float) (0.5 * (double) (u1 - u2));
    float w3 = (float) (0.5 * (double) (u0 - u3));
    float w4 = (float) (0.8314696123 * (double) u4 + -0.55557023302 * (double) u7);
    float w5 = (float) (0.9807852804 * (double) u5 + -0.19509032201 * (double) u6);
    float w6 = (float) (0.19509032201 * (double) u5 + 0.9807852804 * (double) u6);
    float w7 = (float) (0.55557023302 * (double) u4 + 0.8314696123 * (double) u7);

    const double K = 1.414213562 * 2;
    d[0] = (float) (K * (double) (w0 + w7));
    d[1] = (float) (K * (double) (w1 + w6));
    d[2] = (float) (K * (double) (w2 + w5));
    d[3] = (float) (K * (double) (w3 + w4));
    d[4] = (float) (K * (double) (w3 - w4));
    d[5] = (float) (K * (double) (w2 - w5));
    d[6] = (float) (K * (double) (w1 - w6));
    d[7] = (float) (K * (double) (w0 - w7));
}

/* dequant (quant_table.c:131-138, int16 wrap) + 2-D IDCT + level shift + clamp +
 * truncate (dct.c:158-204).  out is row-major spatial: out[8*y + x]. */
void orc_dequant_idct(const int16_t *zz, const uint16_t *qt, int16_t *out) {
    float c[64];
    for (int i = 0; i < 64; i++) {
        int n = ZZ[i];
        c[n] = (float) (int16_t) ((int) zz[i] * (int) qt[n]);
    }
    for (int i = 0; i < 8; i++) {          /* dct.c:164-167; c[0] is scaled twice */
        c[i] = (float) ((double) c[i] * 0.707106781);
        c[8 * i] = (float) ((double) c[8 * i] * 0.707106781);
    }
    for (int r = 0; r < 8; r++) {
        orc_idct8(c + 8 * r);
    }
    float t[64];
    for (int r = 0; r < 8; r++) {
        for (int k = 0; k < 8; k++) {
            t[8 * k + r] = c[8 * r + k];
        }
    }
    for (int r = 0; r < 8; r++) {
        orc_idct8(t + 8 * r);
    }
    /* t[8*x + y]; the reference stores du[x + 8*y] (dct.c:189-193) */
    for (int x = 0; x < 8; x++) {
        for (int y = 0; y < 8; y++) {
            double v = 0.25 * (double) t[8 * x + y] + 128.0;
            double cl = (v > 255.0) ? 255.0 : ((v < 0.0) ? 0.0 : v);
            out[x + 8 * y] = (int16_t) cl;
        }
    }
}

/* Planes, decode.c:436-533: block (v,h) of MCU m of a component with sampling
 * (hsf,vsf) sits at plane pixel (8*(mx*hsf+h), 8*(my*vsf+v)) -- plain MCU raster.
 * A single-component scan is treated as 1x1 blocks (decode.c:454-456). */
static void orc_planes(const OJpeg *j, const int16_t *coefs, uint8_t **plane) {
    int mx, my;
    orc_nblocks(j, &mx, &my);
    int64_t blk = 0;
    int16_t px[64];
    for (int m_y = 0; m_y < my; m_y++) {
        for (int m_x = 0; m_x < mx; m_x++) {
            for (int i = 0; i < j->nscan; i++) {
                int ci = j->scan_comp[i];
                const OComp *c = &j->c[ci];
                for (int v = 0; v < c->vsf; v++) {
                    for (int h = 0; h < c->hsf; h++) {
                        orc_dequant_idct(coefs + blk * 64, j->qt[c->qtid & 3], px);
                        blk++;
                        int x0 = 8 * (m_x * c->hsf + h), y0 = 8 * (m_y * c->vsf + v);
                        for (int r = 0; r < 8; r++) {
                            for (int q = 0; q < 8; q++) {
                                plane[ci][(size_t) (y0 + r) * c->pw + x0 + q] = (uint8_t) px[8 * r + q];
                            }
                        }
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* upsample + colour                                                        */
/* ------------------------------------------------------------------------ */
/* colour_conversion.c:76-79: "(v < 0) ? 0 : (v > 256 ? 256 : v)" stored to a
 * uint8_t.  gcc 11.4 -O2 folds the out-of-range constant arm to 255 and converts
 * the in-range arm through a 32-bit truncation, so exactly 256.0f wraps to 0.  One
 * triple of the 2^24 reaches it: (Y, Cb, Cr) = (219, 228, 28), whose green sum rounds
 * to 256.0f and is stored as 0 (the sweep over all triples against the compiled
 * reference, tests/test_oracle_vs_ref.py, covers it). */
static uint8_t ostore(float v) {
    if (v < 0.0f) return 0;
    if (v > 256.0f) return 255;
    return (uint8_t) (int32_t) v;
}

/* colour_conversion.c:71-74: double arithmetic left to right, rounded to float */
static void oycc(int Y, int Cb, int Cr, uint8_t *o) {
    float R = (float) ((double) Y + 1.402 * ((double) (float) Cr - 128.0));
    float G = (float) (((double) Y - 0.34414 * ((double) (float) Cb - 128.0)) - 0.71414 * ((double) (float) Cr - 128.0));
    float B = (float) ((double) Y + 1.772 * ((double) (float) Cb - 128.0));
    o[0] = ostore(R);
    o[1] = ostore(G);
    o[2] = ostore(B);
}

int orc_ycbcr_row(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int n, uint8_t *rgb) {
    for (int i = 0; i < n; i++) {
        oycc(y[i], cb[i], cr[i], rgb + 3 * i);
    }
    return 0;
}

/* colour_conversion.c:30-83 (three components) and :5-28 (grey).  Nearest
 * neighbour: source index = (uint32)(ratio * i) with ratio = (float)sf / max_sf. */
static int orc_colour(const OJpeg *j, uint8_t **plane, uint8_t *rgb, int pitch) {
    if (j->ncs == 1) {
        for (int i = 0; i < j->Y; i++) {
            for (int k = 0; k < j->X; k++) {
                uint8_t v = plane[0][(size_t) i * j->c[0].pw + k];
                uint8_t *o = rgb + (size_t) i * pitch + 3 * k;
                o[0] = o[1] = o[2] = v;
            }
        }
        return 0;
    }
    if (j->ncs != 3) {
        return -1;
    }
    float hr[3], vr[3];
    for (int c = 0; c < 3; c++) {
        hr[c] = (float) j->c[c].hsf / j->hmax;
        vr[c] = (float) j->c[c].vsf / j->vmax;
    }
    for (int i = 0; i < j->Y; i++) {
        const uint8_t *row[3];
        for (int c = 0; c < 3; c++) {
            row[c] = plane[c] + (size_t) (uint32_t) (i * vr[c]) * j->c[c].pw;
        }
        for (int k = 0; k < j->X; k++) {
            oycc(row[0][(uint32_t) (hr[0] * k)], row[1][(uint32_t) (hr[1] * k)],
                 row[2][(uint32_t) (hr[2] * k)], rgb + (size_t) i * pitch + 3 * k);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* frames coded in several scans                                            */
/* ------------------------------------------------------------------------ */
/* where the entropy-coded segment that starts at p ends: the next marker that is neither a
 * stuffed FF00, a fill byte nor RSTn (T.81 B.1.1.5) */
static const uint8_t *oscan_end(const uint8_t *p, const uint8_t *end) {
    while (p + 1 < end) {
        if (p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7)) {
            return p;
        }
        p++;
    }
    return end;
}

/* T.81 A.2.2: a scan of one component codes its ceil(x/8) x ceil(y/8) blocks in raster order and
 * the restart interval counts blocks; A.2.3: a scan of two of the three components codes the
 * frame's MCUs with those components' blocks only.  Tables are those in force at the scan's SOS.
 * Coefficients come out scan after scan in coding order, the plane of a component coded alone is
 * padded to whole blocks, that of one coded with another to whole MCUs, and the colour stage is
 * the reference's (colour_conversion.c:30-83) on those planes. */
static int orc_run_scans(const uint8_t *buf, size_t len, int want_tap, int want_rgb, OrcOut *out) {
    OJpeg *j = (OJpeg *) calloc(1, sizeof(OJpeg));
    int16_t *coefs = NULL;
    int64_t nb_total = 0, nb_cap = 0;
    int seen[4] = {0, 0, 0, 0}, have_sof = 0, rc = -1;
    size_t pos = 0;
    if (!j) return -1;
    while (pos + 1 < len) {
        if (buf[pos] != 0xFF) {
            pos++;
            continue;
        }
        int m = buf[pos + 1];
        pos += 2;
        if (m == 0xD8 || m == 0x01 || m == 0xFF || (m >= 0xD0 && m <= 0xD7)) {
            if (m == 0xFF) pos--;
            continue;
        }
        if (m == 0xD9) break;
        if (pos + 2 > len) goto done;
        int seglen = be16(buf + pos);
        if (seglen < 2 || pos + (size_t) seglen > len) goto done;
        const uint8_t *body = buf + pos + 2;
        int blen = seglen - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (have_sof || parse_sof(j, body, blen) || j->precision != 8 || j->ncs != 3) goto done;
            have_sof = 1;
            for (int i = 0; i < 3; i++) {
                OComp *c = &j->c[i];
                c->pw = pad_up(c->x, 8);
                c->ph = pad_up(c->y, 8);
                out->plane_w[i] = c->pw;
                out->plane_h[i] = c->ph;
                out->hsf[i] = c->hsf;
                out->vsf[i] = c->vsf;
                out->plane[i] = (uint8_t *) calloc((size_t) c->pw * c->ph + 16, 1);
            }
        } else if (m == 0xC4) {
            if (parse_dht(j, body, blen)) goto done;
        } else if (m == 0xDB) {
            if (parse_dqt(j, body, blen)) goto done;
        } else if (m == 0xDD) {
            if (blen != 2) goto done;
            j->ri = be16(body);
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            goto done;
        } else if (m == 0xDA) {
            if (!have_sof || parse_sos(j, body, blen) || j->nscan < 1 || j->nscan > 2) goto done;
            for (int i = 0; i < j->nscan; i++) {
                if (seen[j->scan_comp[i]]) goto done;
                seen[j->scan_comp[i]] = 1;
            }
            const uint8_t *ent = buf + pos + seglen, *end = oscan_end(ent, buf + len);
            int mcus_x, mcus_y, bpm = 0;
            if (j->nscan == 1) {
                /* A.2.2: the component's own blocks, one per "MCU" */
                const OComp *c = &j->c[j->scan_comp[0]];
                mcus_x = c->pw / 8;
                mcus_y = c->ph / 8;
                bpm = 1;
            } else {
                /* A.2.3: the frame's MCU grid with the blocks of the scan's components only; their
                 * planes are padded to whole MCUs */
                int hmax = 1, vmax = 1;
                for (int i = 0; i < 3; i++) {
                    hmax = j->c[i].hsf > hmax ? j->c[i].hsf : hmax;
                    vmax = j->c[i].vsf > vmax ? j->c[i].vsf : vmax;
                }
                mcus_x = (j->X + 8 * hmax - 1) / (8 * hmax);
                mcus_y = (j->Y + 8 * vmax - 1) / (8 * vmax);
                for (int i = 0; i < j->nscan; i++) {
                    const int ci = j->scan_comp[i];
                    OComp *c = &j->c[ci];
                    c->pw = mcus_x * 8 * c->hsf;
                    c->ph = mcus_y * 8 * c->vsf;
                    out->plane_w[ci] = c->pw;
                    out->plane_h[ci] = c->ph;
                    free(out->plane[ci]);
                    out->plane[ci] = (uint8_t *) calloc((size_t) c->pw * c->ph + 16, 1);
                    bpm += c->hsf * c->vsf;
                }
            }
            const int64_t nmcu = (int64_t) mcus_x * mcus_y, nb = nmcu * bpm;
            if (nb_total + nb > nb_cap) {
                nb_cap = (nb_total + nb) * 2;
                coefs = (int16_t *) realloc(coefs, (size_t) nb_cap * 64 * sizeof(int16_t) + 16);
            }
            int16_t *cs = coefs + nb_total * 64;
            memset(cs, 0, (size_t) nb * 64 * sizeof(int16_t));
            OBits b = {ent, buf + len, 0};
            int16_t pred[4] = {0, 0, 0, 0}, px[64];
            int64_t k = 0;
            for (int64_t m = 0; m < nmcu; m++) {
                if (j->ri && m && (m % j->ri) == 0) {
                    orestart(&b);
                    memset(pred, 0, sizeof pred);
                }
                for (int i = 0; i < j->nscan; i++) {
                    const int ci = j->scan_comp[i];
                    const OComp *c = &j->c[ci];
                    const OHuff *dc = &j->ht[0][c->dc & 3], *ac = &j->ht[1][c->ac & 3];
                    const int nh = j->nscan == 1 ? 1 : c->hsf, nv = j->nscan == 1 ? 1 : c->vsf;
                    for (int v = 0; v < nv; v++) {
                        for (int h = 0; h < nh; h++, k++) {
                            oblock(&b, dc, ac, &pred[i], cs + k * 64);
                            orc_dequant_idct(cs + k * 64, j->qt[c->qtid & 3], px);
                            const int x0 = 8 * ((int) (m % mcus_x) * nh + h), y0 = 8 * ((int) (m / mcus_x) * nv + v);
                            for (int r = 0; r < 8; r++) {
                                for (int q = 0; q < 8; q++) {
                                    out->plane[ci][(size_t) (y0 + r) * c->pw + x0 + q] = (uint8_t) px[8 * r + q];
                                }
                            }
                        }
                    }
                }
            }
            nb_total += nb;
            pos = (size_t) (end - buf);
            continue;
        }
        pos += (size_t) seglen;
    }
    if (!have_sof || !seen[0] || !seen[1] || !seen[2]) goto done;
    out->X = j->X;
    out->Y = j->Y;
    out->ncs = 3;
    rc = 0;
    if (want_rgb) {
        out->pitch = 3 * j->X;
        out->rgb = (uint8_t *) calloc((size_t) out->pitch * j->Y + 16, 1);
        rc = orc_colour(j, out->plane, out->rgb, out->pitch);
    }
    if (want_tap) {
        out->nblocks = nb_total;
        out->coefs = coefs;
        coefs = NULL;
    }
done:
    free(coefs);
    free(j);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* public entry points                                                      */
/* ------------------------------------------------------------------------ */
static int orc_run(const uint8_t *buf, size_t len, int want_tap, int want_rgb, OrcOut *out, uint8_t *rgb_reuse) {
    OJpeg *j = (OJpeg *) malloc(sizeof(OJpeg));
    if (!j || orc_parse(buf, len, j) != 0) {
        free(j);
        return -1;
    }
    if (j->nscan < j->ncs && j->ncs == 3 && j->precision == 8 && !rgb_reuse) {
        free(j);
        return orc_run_scans(buf, len, want_tap, want_rgb, out);   /* the frame takes several scans */
    }
    if (j->nscan != j->ncs || j->precision != 8) {
        free(j);
        return -1;   /* interleaved single-scan, 8-bit only */
    }
    int mx, my;
    int64_t nb = orc_nblocks(j, &mx, &my);
    int16_t *coefs = (int16_t *) malloc((size_t) nb * 64 * sizeof(int16_t) + 16);
    orc_entropy(j, coefs);

    out->X = j->X;
    out->Y = j->Y;
    out->ncs = j->ncs;
    for (int i = 0; i < j->ncs; i++) {
        out->hsf[i] = j->c[i].hsf;
        out->vsf[i] = j->c[i].vsf;
        out->plane_w[i] = j->c[i].pw;
        out->plane_h[i] = j->c[i].ph;
        out->plane[i] = (uint8_t *) calloc((size_t) j->c[i].pw * j->c[i].ph, 1);
    }
    orc_planes(j, coefs, out->plane);
    int rc = 0;
    if (want_rgb) {
        out->pitch = 3 * j->X;
        out->rgb = rgb_reuse ? rgb_reuse : (uint8_t *) calloc((size_t) out->pitch * j->Y + 16, 1);
        rc = orc_colour(j, out->plane, out->rgb, out->pitch);
    }
    if (want_tap) {
        out->nblocks = nb;
        out->coefs = coefs;
    } else {
        free(coefs);
    }
    free(j);
    return rc;
}

int orc_decode(const uint8_t *buf, size_t len, int want_tap, int want_rgb, OrcOut *out) {
    memset(out, 0, sizeof *out);
    int rc = orc_run(buf, len, want_tap, want_rgb, out, NULL);
    if (rc != 0) {
        orc_free(out);
    }
    return rc;
}

void orc_free(OrcOut *out) {
    for (int i = 0; i < 4; i++) {
        free(out->plane[i]);
    }
    free(out->rgb);
    free(out->coefs);
    memset(out, 0, sizeof *out);
}

double orc_time_decode(const uint8_t *buf, size_t len, int iters) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int it = 0; it < iters; it++) {
        OrcOut o;
        memset(&o, 0, sizeof o);
        if (orc_run(buf, len, 0, 1, &o, NULL) != 0) {
            return -1.0;
        }
        orc_free(&o);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}
