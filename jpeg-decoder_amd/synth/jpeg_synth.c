/*
 * jpeg_synth.c -- deterministic baseline-JPEG (SOF0) generator for benchmark and
 * test inputs.  The reference ships no encoder (encode.c:12-74 is a stub that
 * prints one DCT) and no sample images, and the benchmark shapes (1080p 4:4:4 and
 * 4K 4:2:0 with one restart interval per MCU row, 8K without restart markers) are
 * too large to commit, so inputs are produced from a seed wherever they are needed.
 *
 * Integer-only pipeline (fixed-point colour transform, exact integer DCT, integer
 * quantiser) so the same pixels give the same bytes on every machine.  Huffman
 * tables are the Annex K typical tables of ITU-T T.81 or, on request, tables fitted to
 * the image (two passes: symbol histogram, then the code-length procedure of T.81
 * Annex K.2 with its 16-bit limit); quantiser tables are the Annex K tables under the
 * usual 1..100 quality scaling.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const uint8_t ZZ[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* T.81 Table K.1 / K.2, natural order */
static const uint8_t QL[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t QC[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* T.81 Tables K.3-K.6: BITS then HUFFVAL */
static const uint8_t DCL_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DCC_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t ACL_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t ACL_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t ACC_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t ACC_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct {
    uint16_t code[256];
    uint8_t len[256];
} EncTab;

static void build_enc(const uint8_t *bits, const uint8_t *vals, EncTab *t) {
    memset(t, 0, sizeof *t);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++) {
            t->code[vals[k]] = (uint16_t) code++;
            t->len[vals[k]] = (uint8_t) l;
            k++;
        }
        code <<= 1;
    }
}

typedef struct {
    uint8_t *out;
    size_t pos, cap;
    uint64_t acc;
    int nbits;
    int overflow;
} BW;

static void bw_byte(BW *w, int b) {
    if (w->pos < w->cap) {
        w->out[w->pos] = (uint8_t) b;
    } else {
        w->overflow = 1;
    }
    w->pos++;
}
static void bw_bits(BW *w, unsigned v, int n) {
    w->acc = (w->acc << n) | (v & ((1u << n) - 1));
    w->nbits += n;
    while (w->nbits >= 8) {
        int b = (int) ((w->acc >> (w->nbits - 8)) & 0xFF);
        bw_byte(w, b);
        if (b == 0xFF) {
            bw_byte(w, 0);
        }
        w->nbits -= 8;
    }
}
static void bw_flush(BW *w) {
    if (w->nbits) {
        bw_bits(w, 0x7F, 8 - w->nbits);   /* pad with ones */
    }
}
static void bw_u16(BW *w, int v) {
    bw_byte(w, v >> 8);
    bw_byte(w, v & 0xFF);
}

static int nbits_of(int v) {
    int a = v < 0 ? -v : v, n = 0;
    while (a) {
        n++;
        a >>= 1;
    }
    return n;
}

/* One block's symbols.  With freq_dc / freq_ac set they are only counted (the first pass
 * of an encode with fitted tables); otherwise they are written with the tables dc / ac. */
static void enc_block(BW *w, const int16_t *q, int *pred, const EncTab *dc, const EncTab *ac, long *freq_dc, long *freq_ac) {
    int diff = q[0] - *pred;
    *pred = q[0];
    int s = nbits_of(diff);
    if (freq_dc) {
        freq_dc[s]++;
    } else {
        bw_bits(w, dc->code[s], dc->len[s]);
        if (s) {
            bw_bits(w, (unsigned) (diff < 0 ? diff - 1 : diff), s);
        }
    }
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int v = q[ZZ[k]];
        if (v == 0) {
            run++;
            continue;
        }
        while (run > 15) {
            if (freq_ac) freq_ac[0xF0]++;
            else bw_bits(w, ac->code[0xF0], ac->len[0xF0]);
            run -= 16;
        }
        int sz = nbits_of(v);
        int rs = (run << 4) | sz;
        if (freq_ac) {
            freq_ac[rs]++;
        } else {
            bw_bits(w, ac->code[rs], ac->len[rs]);
            bw_bits(w, (unsigned) (v < 0 ? v - 1 : v), sz);
        }
        run = 0;
    }
    if (run) {
        if (freq_ac) freq_ac[0]++;
        else bw_bits(w, ac->code[0], ac->len[0]);
    }
}

/* T.81 Annex K.2: code lengths from symbol frequencies (Figure K.1), no code longer than
 * 16 bits (Figure K.3), symbols sorted by length then value (Figure K.4).  A reserved
 * 257th symbol of frequency 1 keeps the all-ones code word unused.  bits[l-1] = codes of
 * length l; vals = the symbols in code order; returns how many. */
static int fit_table(const long *freq_in, uint8_t *bits, uint8_t *vals) {
    const int max_len = 16;
    long freq[257];
    int codesize[257], others[257];
    int nbits[33];
    for (int i = 0; i < 256; i++) freq[i] = freq_in[i];
    freq[256] = 1;
    for (int i = 0; i < 257; i++) {
        codesize[i] = 0;
        others[i] = -1;
    }
    for (;;) {
        int c1 = -1, c2 = -1;
        long v = 0;
        for (int i = 0; i <= 256; i++) {            /* least frequency, largest symbol among equals */
            if (freq[i] && (c1 < 0 || freq[i] <= v)) {
                v = freq[i];
                c1 = i;
            }
        }
        v = 0;
        for (int i = 0; i <= 256; i++) {
            if (freq[i] && i != c1 && (c2 < 0 || freq[i] <= v)) {
                v = freq[i];
                c2 = i;
            }
        }
        if (c2 < 0) break;
        freq[c1] += freq[c2];
        freq[c2] = 0;
        codesize[c1]++;
        while (others[c1] >= 0) {
            c1 = others[c1];
            codesize[c1]++;
        }
        others[c1] = c2;
        codesize[c2]++;
        while (others[c2] >= 0) {
            c2 = others[c2];
            codesize[c2]++;
        }
    }
    memset(nbits, 0, sizeof nbits);
    for (int i = 0; i <= 256; i++) {
        if (codesize[i]) nbits[codesize[i] > 32 ? 32 : codesize[i]]++;
    }
    for (int i = 32; i > max_len; i--) {
        while (nbits[i] > 0) {
            int j = i - 2;
            while (nbits[j] == 0) j--;
            nbits[i] -= 2;
            nbits[i - 1]++;
            nbits[j + 1] += 2;
            nbits[j]--;
        }
    }
    int top = max_len;
    while (nbits[top] == 0) top--;
    nbits[top]--;                                   /* the reserved symbol's code */
    for (int l = 1; l <= 16; l++) bits[l - 1] = (uint8_t) nbits[l];
    int n = 0;
    for (int l = 1; l <= 32; l++) {
        for (int i = 0; i < 256; i++) {
            if (codesize[i] == l) vals[n++] = (uint8_t) i;
        }
    }
    return n;
}

/* Exact integer 2-D DCT-II of (sample-128), then divide by the quantiser step
 * with round-half-away-from-zero.  CT[u][x] = round(2^13 * c(u)/2 * cos((2x+1)u*pi/16)). */
static int32_t CT[8][8];
static int ct_ready = 0;
static void ct_init(void) {
    if (ct_ready) return;
    for (int u = 0; u < 8; u++) {
        for (int x = 0; x < 8; x++) {
            double c = (u == 0) ? sqrt(0.5) : 1.0;
            CT[u][x] = (int32_t) lrint(8192.0 * 0.5 * c * cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
        }
    }
    ct_ready = 1;
}

static void fdct_quant(const uint8_t *src, int stride, const uint16_t *qt, int16_t *out) {
    int32_t tmp[64];
    for (int y = 0; y < 8; y++) {
        for (int u = 0; u < 8; u++) {
            int32_t s = 0;
            for (int x = 0; x < 8; x++) {
                s += ((int) src[y * stride + x] - 128) * CT[u][x];
            }
            tmp[y * 8 + u] = s;
        }
    }
    for (int v = 0; v < 8; v++) {
        for (int u = 0; u < 8; u++) {
            int64_t s = 0;
            for (int y = 0; y < 8; y++) {
                s += (int64_t) tmp[y * 8 + u] * CT[v][y];
            }
            int64_t den = (int64_t) qt[v * 8 + u] << 26;
            int64_t q = (s >= 0) ? (s + den / 2) / den : -((-s + den / 2) / den);
            out[v * 8 + u] = (int16_t) q;
        }
    }
}

static void scale_qt(const uint8_t *base, int quality, uint16_t *qt) {
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; i++) {
        int v = (base[i] * scale + 50) / 100;
        qt[i] = (uint16_t) (v < 1 ? 1 : (v > 255 ? 255 : v));
    }
}

static void put_dht(BW *w, int tc_th, const uint8_t *bits, const uint8_t *vals) {
    int n = 0;
    for (int i = 0; i < 16; i++) n += bits[i];
    bw_byte(w, 0xFF);
    bw_byte(w, 0xC4);
    bw_u16(w, 2 + 1 + 16 + n);
    bw_byte(w, tc_th);
    for (int i = 0; i < 16; i++) bw_byte(w, bits[i]);
    for (int i = 0; i < n; i++) bw_byte(w, vals[i]);
}

/*
 * pix: H rows of W pixels, ncomp (1 = grey, 3 = RGB) interleaved bytes.
 * hs, vs: luma sampling factors (chroma is 1x1): 1,1 = 4:4:4; 2,1 = 4:2:2;
 *         2,2 = 4:2:0; 1,2 = 4:4:0; 4,1 = 4:1:1.  Ignored for grey.
 * ri: restart interval in MCUs (0 = no DRI segment, no RST markers).
 * flags: bit 0 = fit the Huffman tables to the image instead of using the Annex K ones;
 *        bit 1 = the three bytes of a pixel are Y, Cb, Cr already (no colour conversion);
 *        bit 2 = one scan per component (non-interleaved, T.81 A.2.2: a scan of one component
 *                codes its ceil(x/8) x ceil(y/8) blocks in raster order, the restart interval
 *                counts blocks) instead of one interleaved scan.
 * Returns the number of bytes the stream needs; it is complete only if that is
 * <= cap.  Negative on bad arguments.
 */
/* One scan over the frame components grp[0 .. n) of a three-component frame whose quantised
 * blocks Q lie in interleaved MCU order (bpm blocks per MCU: hs * vs of component 0, then one
 * each of components 1 and 2).  n = 1: the component's own blocks in raster order, those that
 * only pad the MCU grid left out (T.81 A.2.2); n > 1: the frame's MCU grid with the blocks of
 * the scan's components only (A.2.3).  ri counts the scan's own MCUs (blocks when n = 1).
 * With w = NULL nothing is written and the symbols are counted in freq (DC luma, AC luma, DC
 * chroma, AC chroma) instead. */
static void code_scan(BW *w, const int *grp, int n, const int16_t *Q, int W, int H, int hs, int vs, int mx, int my, int ri,
                      const EncTab *dcl, const EncTab *acl, const EncTab *dcc, const EncTab *acc, long (*freq)[256]) {
    const int bpm = hs * vs + 2;
    int pred[3] = {0, 0, 0}, rst = 0;
    int64_t done = 0;
    if (n == 1) {
        const int c = grp[0];
        const int hc = c ? 1 : hs, vc = c ? 1 : vs, base = c ? hs * vs + (c - 1) : 0;
        const int xc = (W * hc + hs - 1) / hs, yc = (H * vc + vs - 1) / vs;     /* samples of the component */
        const int bw_c = (xc + 7) / 8, bh_c = (yc + 7) / 8;
        for (int by = 0; by < bh_c; by++) {
            for (int bx = 0; bx < bw_c; bx++, done++) {
                if (ri && done && (done % ri) == 0) {
                    if (w) {
                        bw_flush(w);
                        bw_byte(w, 0xFF);
                        bw_byte(w, 0xD0 + (rst++ & 7));
                    }
                    pred[c] = 0;
                }
                const int64_t mcu = (int64_t) (by / vc) * mx + bx / hc;
                const int16_t *q = Q + (mcu * bpm + base + (by % vc) * hc + bx % hc) * 64;
                enc_block(w, q, &pred[c], c ? dcc : dcl, c ? acc : acl, freq ? freq[c ? 2 : 0] : NULL, freq ? freq[c ? 3 : 1] : NULL);
            }
        }
    } else {
        for (int64_t m = 0; m < (int64_t) mx * my; m++, done++) {
            if (ri && done && (done % ri) == 0) {
                if (w) {
                    bw_flush(w);
                    bw_byte(w, 0xFF);
                    bw_byte(w, 0xD0 + (rst++ & 7));
                }
                pred[0] = pred[1] = pred[2] = 0;
            }
            for (int g = 0; g < n; g++) {
                const int c = grp[g];
                const int nb = c ? 1 : hs * vs, base = c ? hs * vs + (c - 1) : 0;
                for (int i = 0; i < nb; i++) {
                    enc_block(w, Q + (m * bpm + base + i) * 64, &pred[c], c ? dcc : dcl, c ? acc : acl, freq ? freq[c ? 2 : 0] : NULL,
                              freq ? freq[c ? 3 : 1] : NULL);
                }
            }
        }
    }
    if (w) bw_flush(w);
}

/* how the three components are dealt out over the scans: flags bit 2 one scan each, bit 3
 * (Y)(Cb Cr), bit 4 (Y Cb)(Cr); groups end with -1 */
static int scan_groups(int flags, int grp[3][4]) {
    static const int per_comp[3][4] = {{0, -1}, {1, -1}, {2, -1}};
    static const int y_cbcr[3][4] = {{0, -1}, {1, 2, -1}, {-1}};
    static const int ycb_cr[3][4] = {{0, 1, -1}, {2, -1}, {-1}};
    const int (*src)[4] = (flags & 8) ? y_cbcr : (flags & 16) ? ycb_cr : per_comp;
    memcpy(grp, src, sizeof per_comp);
    return (flags & (8 | 16)) ? 2 : 3;
}

int64_t mjs_encode2(const uint8_t *pix, int W, int H, int ncomp, int hs, int vs, int quality, int ri, int flags,
                    uint8_t *out, size_t cap);

int64_t mjs_encode(const uint8_t *pix, int W, int H, int ncomp, int hs, int vs, int quality, int ri,
                   uint8_t *out, size_t cap) {
    return mjs_encode2(pix, W, H, ncomp, hs, vs, quality, ri, 0, out, cap);
}

int64_t mjs_encode2(const uint8_t *pix, int W, int H, int ncomp, int hs, int vs, int quality, int ri, int flags,
                    uint8_t *out, size_t cap) {
    if (!pix || W <= 0 || H <= 0 || W > 65535 || H > 65535 || (ncomp != 1 && ncomp != 3)) return -1;
    if (ncomp == 1) hs = vs = 1;
    if (hs < 1 || hs > 4 || vs < 1 || vs > 2 || ri < 0 || ri > 65535) return -1;
    ct_init();

    int mcu_w = 8 * hs, mcu_h = 8 * vs;
    int mx = (W + mcu_w - 1) / mcu_w, my = (H + mcu_h - 1) / mcu_h;
    int PW = mx * mcu_w, PH = my * mcu_h;          /* padded luma size */
    int CW = PW / hs, CH = PH / vs;                /* chroma plane size */

    uint8_t *Yp = (uint8_t *) malloc((size_t) PW * PH);
    uint8_t *Cb = NULL, *Cr = NULL;
    if (ncomp == 3) {
        Cb = (uint8_t *) malloc((size_t) CW * CH);
        Cr = (uint8_t *) malloc((size_t) CW * CH);
    }
    /* full-resolution planes with edge replication; chroma accumulated then averaged */
    uint8_t *fcb = NULL, *fcr = NULL;
    if (ncomp == 3) {
        fcb = (uint8_t *) malloc((size_t) PW * PH);
        fcr = (uint8_t *) malloc((size_t) PW * PH);
    }
    for (int y = 0; y < PH; y++) {
        int sy = y < H ? y : H - 1;
        for (int x = 0; x < PW; x++) {
            int sx = x < W ? x : W - 1;
            const uint8_t *p = pix + ((size_t) sy * W + sx) * ncomp;
            if (ncomp == 1) {
                Yp[(size_t) y * PW + x] = p[0];
            } else {
                int r = p[0], g = p[1], b = p[2];
                if (flags & 2) {
                    Yp[(size_t) y * PW + x] = p[0];
                    fcb[(size_t) y * PW + x] = p[1];
                    fcr[(size_t) y * PW + x] = p[2];
                    continue;
                }
                /* JFIF full-range BT.601, 16-bit fixed point */
                Yp[(size_t) y * PW + x] = (uint8_t) ((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
                fcb[(size_t) y * PW + x] = (uint8_t) ((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
                fcr[(size_t) y * PW + x] = (uint8_t) ((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
            }
        }
    }
    if (ncomp == 3) {
        int n = hs * vs;
        for (int y = 0; y < CH; y++) {
            for (int x = 0; x < CW; x++) {
                int sb = 0, sr = 0;
                for (int dy = 0; dy < vs; dy++) {
                    for (int dx = 0; dx < hs; dx++) {
                        size_t i = (size_t) (y * vs + dy) * PW + (x * hs + dx);
                        sb += fcb[i];
                        sr += fcr[i];
                    }
                }
                Cb[(size_t) y * CW + x] = (uint8_t) ((sb + n / 2) / n);
                Cr[(size_t) y * CW + x] = (uint8_t) ((sr + n / 2) / n);
            }
        }
        free(fcb);
        free(fcr);
    }

    uint16_t qt[2][64];
    scale_qt(QL, quality, qt[0]);
    scale_qt(QC, quality, qt[1]);
    /* quantised blocks of the whole scan, in scan order */
    const int bpm = hs * vs + (ncomp == 3 ? 2 : 0);
    const int64_t nm = (int64_t) mx * my;
    int16_t *Q = (int16_t *) malloc((size_t) nm * bpm * 64 * sizeof(int16_t));
    if (!Q || !Yp || (ncomp == 3 && (!Cb || !Cr))) {
        free(Q);
        free(Yp);
        free(Cb);
        free(Cr);
        return -1;
    }
    {
        int16_t *q = Q;
        for (int my_i = 0; my_i < my; my_i++) {
            for (int mx_i = 0; mx_i < mx; mx_i++) {
                for (int v = 0; v < vs; v++) {
                    for (int h = 0; h < hs; h++, q += 64) {
                        fdct_quant(Yp + (size_t) (my_i * mcu_h + 8 * v) * PW + mx_i * mcu_w + 8 * h, PW, qt[0], q);
                    }
                }
                if (ncomp == 3) {
                    fdct_quant(Cb + (size_t) (my_i * 8) * CW + mx_i * 8, CW, qt[1], q);
                    q += 64;
                    fdct_quant(Cr + (size_t) (my_i * 8) * CW + mx_i * 8, CW, qt[1], q);
                    q += 64;
                }
            }
        }
    }

    uint8_t bits[4][16], vals[4][256];     /* DC luma, AC luma, DC chroma, AC chroma */
    memcpy(bits[0], DCL_BITS, 16);
    memcpy(vals[0], DC_VALS, sizeof DC_VALS);
    memcpy(bits[1], ACL_BITS, 16);
    memcpy(vals[1], ACL_VALS, sizeof ACL_VALS);
    memcpy(bits[2], DCC_BITS, 16);
    memcpy(vals[2], DC_VALS, sizeof DC_VALS);
    memcpy(bits[3], ACC_BITS, 16);
    memcpy(vals[3], ACC_VALS, sizeof ACC_VALS);
    if (flags & 1) {
        long freq[4][256];
        memset(freq, 0, sizeof freq);
        int pred[3] = {0, 0, 0};
        const int16_t *q = Q;
        for (int64_t m = 0; m < nm; m++) {
            if (ri && m && (m % ri) == 0) pred[0] = pred[1] = pred[2] = 0;
            for (int i = 0; i < hs * vs; i++, q += 64) enc_block(NULL, q, &pred[0], NULL, NULL, freq[0], freq[1]);
            if (ncomp == 3) {
                enc_block(NULL, q, &pred[1], NULL, NULL, freq[2], freq[3]);
                q += 64;
                enc_block(NULL, q, &pred[2], NULL, NULL, freq[2], freq[3]);
                q += 64;
            }
        }
        if ((flags & 4) && ncomp == 3) {
            /* one scan per component: the DC differences follow the raster order of each
             * component's own blocks, and the blocks that only pad the MCU grid are not coded */
            memset(freq, 0, sizeof freq);
            for (int c = 0; c < 3; c++) {
                const int hc = c ? 1 : hs, vc = c ? 1 : vs, base = c ? hs * vs + (c - 1) : 0;
                const int xc = (W * hc + hs - 1) / hs, yc = (H * vc + vs - 1) / vs;
                const int bw_c = (xc + 7) / 8, bh_c = (yc + 7) / 8;
                int p1 = 0;
                int64_t done = 0;
                for (int by = 0; by < bh_c; by++) {
                    for (int bx = 0; bx < bw_c; bx++, done++) {
                        if (ri && done && (done % ri) == 0) p1 = 0;
                        const int64_t mcu = (int64_t) (by / vc) * mx + bx / hc;
                        enc_block(NULL, Q + (mcu * bpm + base + (by % vc) * hc + bx % hc) * 64, &p1, NULL, NULL, freq[c ? 2 : 0],
                                  freq[c ? 3 : 1]);
                    }
                }
            }
        }
        if ((flags & (8 | 16)) && ncomp == 3) {
            /* scans that interleave some of the components: the symbols as those scans code them */
            int grp[3][4];
            const int ns = scan_groups(flags, grp);
            memset(freq, 0, sizeof freq);
            for (int sc = 0; sc < ns; sc++) {
                int n = 0;
                while (grp[sc][n] >= 0) n++;
                code_scan(NULL, grp[sc], n, Q, W, H, hs, vs, mx, my, ri, NULL, NULL, NULL, NULL, freq);
            }
        }
        for (int t = 0; t < (ncomp == 3 ? 4 : 2); t++) fit_table(freq[t], bits[t], vals[t]);
    }
    EncTab dcl, dcc, acl, acc;
    build_enc(bits[0], vals[0], &dcl);
    build_enc(bits[1], vals[1], &acl);
    build_enc(bits[2], vals[2], &dcc);
    build_enc(bits[3], vals[3], &acc);

    BW w = {out, 0, out ? cap : 0, 0, 0, 0};
    bw_byte(&w, 0xFF); bw_byte(&w, 0xD8);
    /* APP0 JFIF 1.01, no thumbnail */
    static const uint8_t jfif[] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    for (size_t i = 0; i < sizeof jfif; i++) bw_byte(&w, jfif[i]);
    for (int t = 0; t < (ncomp == 3 ? 2 : 1); t++) {
        bw_byte(&w, 0xFF); bw_byte(&w, 0xDB); bw_u16(&w, 67); bw_byte(&w, t);
        for (int k = 0; k < 64; k++) bw_byte(&w, qt[t][ZZ[k]]);
    }
    bw_byte(&w, 0xFF); bw_byte(&w, 0xC0); bw_u16(&w, 8 + 3 * ncomp); bw_byte(&w, 8);
    bw_u16(&w, H); bw_u16(&w, W); bw_byte(&w, ncomp);
    bw_byte(&w, 1); bw_byte(&w, (hs << 4) | vs); bw_byte(&w, 0);
    if (ncomp == 3) {
        bw_byte(&w, 2); bw_byte(&w, 0x11); bw_byte(&w, 1);
        bw_byte(&w, 3); bw_byte(&w, 0x11); bw_byte(&w, 1);
    }
    put_dht(&w, 0x00, bits[0], vals[0]);
    put_dht(&w, 0x10, bits[1], vals[1]);
    if (ncomp == 3) {
        put_dht(&w, 0x01, bits[2], vals[2]);
        put_dht(&w, 0x11, bits[3], vals[3]);
    }
    if (ri) {
        bw_byte(&w, 0xFF); bw_byte(&w, 0xDD); bw_u16(&w, 4); bw_u16(&w, ri);
    }
    if ((flags & (8 | 16)) && ncomp == 3) {
        /* two scans: one component alone, the other two interleaved (or the other way round) */
        int grp[3][4];
        const int ns = scan_groups(flags, grp);
        for (int sc = 0; sc < ns; sc++) {
            int n = 0;
            while (grp[sc][n] >= 0) n++;
            bw_byte(&w, 0xFF); bw_byte(&w, 0xDA); bw_u16(&w, 6 + 2 * n); bw_byte(&w, n);
            for (int g = 0; g < n; g++) {
                bw_byte(&w, grp[sc][g] + 1); bw_byte(&w, grp[sc][g] ? 0x11 : 0x00);
            }
            bw_byte(&w, 0); bw_byte(&w, 63); bw_byte(&w, 0);
            code_scan(&w, grp[sc], n, Q, W, H, hs, vs, mx, my, ri, &dcl, &acl, &dcc, &acc, NULL);
        }
        bw_byte(&w, 0xFF); bw_byte(&w, 0xD9);
        free(Q);
        free(Yp);
        free(Cb);
        free(Cr);
        return (int64_t) w.pos;
    }
    if ((flags & 4) && ncomp == 3) {
        /* three scans of one component each */
        for (int c = 0; c < 3; c++) {
            const int hc = c ? 1 : hs, vc = c ? 1 : vs, base = c ? hs * vs + (c - 1) : 0;
            const int xc = (W * hc + hs - 1) / hs, yc = (H * vc + vs - 1) / vs;     /* samples of the component */
            const int bw_c = (xc + 7) / 8, bh_c = (yc + 7) / 8;
            bw_byte(&w, 0xFF); bw_byte(&w, 0xDA); bw_u16(&w, 8); bw_byte(&w, 1);
            bw_byte(&w, c + 1); bw_byte(&w, c ? 0x11 : 0x00);
            bw_byte(&w, 0); bw_byte(&w, 63); bw_byte(&w, 0);
            int pred = 0, rst = 0;
            int64_t done = 0;
            for (int by = 0; by < bh_c; by++) {
                for (int bx = 0; bx < bw_c; bx++, done++) {
                    if (ri && done && (done % ri) == 0) {
                        bw_flush(&w);
                        bw_byte(&w, 0xFF);
                        bw_byte(&w, 0xD0 + (rst++ & 7));
                        pred = 0;
                    }
                    const int64_t mcu = (int64_t) (by / vc) * mx + bx / hc;
                    const int16_t *q = Q + (mcu * bpm + base + (by % vc) * hc + bx % hc) * 64;
                    enc_block(&w, q, &pred, c ? &dcc : &dcl, c ? &acc : &acl, NULL, NULL);
                }
            }
            bw_flush(&w);
        }
        bw_byte(&w, 0xFF); bw_byte(&w, 0xD9);
        free(Q);
        free(Yp);
        free(Cb);
        free(Cr);
        return (int64_t) w.pos;
    }
    bw_byte(&w, 0xFF); bw_byte(&w, 0xDA); bw_u16(&w, 6 + 2 * ncomp); bw_byte(&w, ncomp);
    bw_byte(&w, 1); bw_byte(&w, 0x00);
    if (ncomp == 3) {
        bw_byte(&w, 2); bw_byte(&w, 0x11);
        bw_byte(&w, 3); bw_byte(&w, 0x11);
    }
    bw_byte(&w, 0); bw_byte(&w, 63); bw_byte(&w, 0);

    int pred[3] = {0, 0, 0};
    const int16_t *q = Q;
    int rst = 0;
    for (int64_t m = 0; m < nm; m++) {
        if (ri && m && (m % ri) == 0) {
            bw_flush(&w);
            bw_byte(&w, 0xFF);
            bw_byte(&w, 0xD0 + (rst++ & 7));
            pred[0] = pred[1] = pred[2] = 0;
        }
        for (int i = 0; i < hs * vs; i++, q += 64) enc_block(&w, q, &pred[0], &dcl, &acl, NULL, NULL);
        if (ncomp == 3) {
            enc_block(&w, q, &pred[1], &dcc, &acc, NULL, NULL);
            q += 64;
            enc_block(&w, q, &pred[2], &dcc, &acc, NULL, NULL);
            q += 64;
        }
    }
    bw_flush(&w);
    bw_byte(&w, 0xFF); bw_byte(&w, 0xD9);
    free(Q);
    free(Yp);
    free(Cb);
    free(Cr);
    return (int64_t) w.pos;
}
