/*
 * segments.c -- host parsers for the JPEG table and header segments, behind the
 * reference's names and structs (frame_header.h, scan_header.h, huff_table.h,
 * quant_table.h, restart_interval.h).  They run on the host for every image: a
 * few hundred bytes per file.  Their results are flattened into device
 * descriptors by flatten.c; nothing here touches pixel or entropy data.
 */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355jpeg_compat.h"

static uint16_t read_u16(Bitstream *bs) {
    uint16_t hi = next_byte(bs);
    return (uint16_t) ((hi << 8) | next_byte(bs));
}

/* ---- SOFn (frame_header.c:9-58) ---------------------------------------- */
int32_t decode_frame_header(Encoding encoding_process, Bitstream *bs, FrameHeader *fh) {
    uint16_t len = read_u16(bs);
    if (len < 8) {
        return -1;
    }
    fh->process = encoding_process;
    fh->precision = next_byte(bs);
    fh->Y = read_u16(bs);
    fh->X = read_u16(bs);
    fh->ncs = next_byte(bs);
    if (fh->ncs == 0 || fh->ncs * 3 != len - 8) {
        return -1;
    }
    fh->cs = (Component *) calloc(fh->ncs, sizeof(Component));
    if (!fh->cs) {
        return -1;
    }
    uint8_t hmax = 0, vmax = 0;
    for (int i = 0; i < fh->ncs; i++) {
        Component *c = &fh->cs[i];
        c->id = next_byte(bs);
        uint8_t hv = next_byte(bs);
        c->hsf = hv >> 4;
        c->vsf = hv & 0xF;
        c->qtid = next_byte(bs);
        if (c->hsf > hmax) hmax = c->hsf;
        if (c->vsf > vmax) vmax = c->vsf;
    }
    /* component size: the ratio is formed in float and the product rounded up,
     * frame_header.c:52-54 */
    for (int i = 0; i < fh->ncs; i++) {
        Component *c = &fh->cs[i];
        c->x = (uint16_t) ceil(fh->X * ((float) c->hsf / hmax));
        c->y = (uint16_t) ceil(fh->Y * ((float) c->vsf / vmax));
    }
    return 0;
}

/* ---- DNL (frame_header.c:60-75) ---------------------------------------- */
int32_t decode_number_of_lines(Bitstream *bs, FrameHeader *fh) {
    if (!fh) {
        return -1;
    }
    if (read_u16(bs) != 4) {
        return -1;
    }
    fh->Y = read_u16(bs);
    return 0;
}

int32_t free_frame_header(FrameHeader *fh) {
    if (fh && fh->cs) {
        free(fh->cs);
        fh->cs = NULL;
    }
    return 0;
}

char *encoding_str(Encoding process) {
    static char *const names[16] = {
        "Baseline DCT",
        "Extended Sequential DCT Huffman Coding",
        "Progressive DCT Huffman Coding",
        "Lossless (Sequential) Huffman Coding",
        NULL,
        "Differential Sequential DCT Huffman Coding",
        "Differential Progressive DCT Huffman Coding",
        "Differential Lossless (Sequential) Huffman Coding",
        NULL,
        "Extended sequential DCT Arithmetic Coding",
        "Progressive DCT Arithmetic Coding",
        "Lossless (sequential) Arithmetic Coding",
        NULL,
        "Differential Sequential DCT Arithmetic Coding",
        "Differential Progressive DCT Arithmetic Coding",
        "Differential Lossless (Sequential) Arithmetic Coding",
    };
    if ((int) process >= 0 && (int) process < 16 && names[process]) {
        return names[process];
    }
    return "Unknown encoding scheme";
}

/* The reference's print_* helpers only speak when its DEBUG macro is set
 * (debug.h:2); here the switch is the MJ_DEBUG environment variable. */
static int dbg(void) {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MJ_DEBUG");
        v = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return v;
}

void print_component(Component *comp, int len) {
    if (!comp || !dbg()) return;
    for (int i = 0; i < len; i++) {
        fprintf(stderr, "  component id=%d h=%d v=%d x=%d y=%d tq=%d\n", comp[i].id, comp[i].hsf, comp[i].vsf,
                comp[i].x, comp[i].y, comp[i].qtid);
    }
}

void print_frame_header(FrameHeader *fh) {
    if (!fh || !dbg()) return;
    fprintf(stderr, "frame: %s P=%d X=%d Y=%d Nf=%d\n", encoding_str(fh->process), fh->precision, fh->X, fh->Y,
            fh->ncs);
    print_component(fh->cs, fh->ncs);
}

/* ---- SOS header (scan_header.c:10-35) ---------------------------------- */
int32_t decode_scan_header(Bitstream *bs, ScanHeader *sh) {
    uint16_t len = read_u16(bs);
    if (len < 6) {
        return -1;
    }
    sh->nics = next_byte(bs);
    /* the fields read below are 2 * nics + 3 bytes: the segment must hold them */
    if (sh->nics == 0 || sh->nics > 4 || len < 6 + 2 * sh->nics) {
        return -1;
    }
    for (int i = 0; i < sh->nics; i++) {
        sh->ics[i].sc = next_byte(bs);
        uint8_t t = next_byte(bs);
        sh->ics[i].dc = t >> 4;
        sh->ics[i].ac = t & 0xF;
    }
    sh->ss = next_byte(bs);
    sh->se = next_byte(bs);
    uint8_t a = next_byte(bs);
    sh->ah = a >> 4;
    sh->al = a & 0xF;
    return 0;
}

void print_image_component(ImageComponent *ics, int len) {
    if (!ics || !dbg()) return;
    for (int i = 0; i < len; i++) {
        fprintf(stderr, "  scan component %d: Cs=%d Td=%d Ta=%d\n", i, ics[i].sc, ics[i].dc, ics[i].ac);
    }
}

void print_scan_header(ScanHeader *sh) {
    if (!sh || !dbg()) return;
    fprintf(stderr, "scan: Ns=%d Ss=%d Se=%d Ah=%d Al=%d\n", sh->nics, sh->ss, sh->se, sh->ah, sh->al);
    print_image_component(sh->ics, sh->nics);
}

/* ---- DHT (huff_table.c:9-43, :165-217, :282-301) ----------------------- */
int new_huff_tables(Encoding process, HuffTables *hts) {
    if (!hts) {
        return -1;
    }
    memset(hts, 0, sizeof *hts);
    /* baseline allows two tables per class, everything else four (huff_table.c:32-41) */
    hts->nDCAC = (process == BDCT) ? 2 : 4;
    return 0;
}

int32_t decode_huff_tables(Bitstream *bs, HuffTables *hts) {
    uint16_t len = read_u16(bs);
    if (len < 2 + 17) {
        return -1;
    }
    uint8_t *end = bs->encoded_data + len - 2;
    while (bs->encoded_data < end) {
        /* every read below stays inside the segment: the class/id byte and the sixteen
         * counts first, then exactly as many symbol bytes as the counts add up to */
        if (end - bs->encoded_data < 1 + MAX_CODE_LEN) {
            return -1;
        }
        uint8_t tc_th = next_byte(bs);
        uint8_t cls = tc_th >> 4, id = tc_th & 0xF;
        if (cls >= TC_NUM || id >= MAX_TABLES || id > hts->nDCAC) {
            return -1;
        }
        HuffTable *ht = &hts->DCAC[cls][id];
        uint32_t total = 0;
        for (int l = 0; l < MAX_CODE_LEN; l++) {
            total += bs->encoded_data[l];
        }
        if (total > 256 || (ptrdiff_t) total > end - (bs->encoded_data + MAX_CODE_LEN)) {
            return -1;
        }
        for (int l = 0; l < MAX_CODE_LEN; l++) {
            ht->len[l] = next_byte(bs);
        }
        /* canonical codes: lengths with no code get min = max = -1 so that a
         * "max >= code" test never matches them (huff_table.c:197-202) */
        uint16_t code = 0;
        for (int l = 0; l < MAX_CODE_LEN; l++) {
            free(ht->symbols[l]);
            ht->symbols[l] = NULL;
            uint8_t n = ht->len[l];
            if (n == 0) {
                ht->min_codes[l] = ht->max_codes[l] = -1;
            } else {
                ht->symbols[l] = (uint8_t *) malloc(n);
                if (!ht->symbols[l]) {
                    return -1;
                }
                for (int k = 0; k < n; k++) {
                    ht->symbols[l][k] = next_byte(bs);
                }
                ht->min_codes[l] = (int16_t) code;
                code = (uint16_t) (code + n);
                ht->max_codes[l] = (int16_t) (code - 1);
            }
            code = (uint16_t) (code << 1);
        }
    }
    return 0;
}

int32_t free_huff_tables(HuffTables *hts) {
    if (!hts) {
        return -1;
    }
    for (int c = 0; c < TC_NUM; c++) {
        for (int t = 0; t < MAX_TABLES; t++) {
            for (int l = 0; l < MAX_CODE_LEN; l++) {
                free(hts->DCAC[c][t].symbols[l]);
                hts->DCAC[c][t].symbols[l] = NULL;
            }
        }
    }
    return 0;
}

void print_huff_table(HuffTable *ht) {
    if (!ht || !dbg()) return;
    for (int l = 0; l < MAX_CODE_LEN; l++) {
        fprintf(stderr, "  len %2d: n=%3d min=%04X max=%04X\n", l + 1, ht->len[l], (uint16_t) ht->min_codes[l],
                (uint16_t) ht->max_codes[l]);
    }
}

void print_huff_tables(HuffTables *hts) {
    if (!hts || !dbg()) return;
    for (int c = 0; c < TC_NUM; c++) {
        for (int t = 0; t < hts->nDCAC && t < MAX_TABLES; t++) {
            fprintf(stderr, "%s table %d\n", c == TC_DC ? "DC" : "AC", t);
            print_huff_table(&hts->DCAC[c][t]);
        }
    }
}

/* ---- DQT (quant_table.c:25-34, :91-152) -------------------------------- */
const uint8_t zigzag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

int32_t decode_quant_table(Bitstream *bs, QuantTables *qts) {
    uint16_t len = read_u16(bs);
    if (len < 2 + 65) {
        return -1;
    }
    int remaining = len - 2;
    while (remaining >= 65) {
        uint8_t pq_tq = get_byte(bs);
        uint8_t prec = pq_tq >> 4, id = pq_tq & 0xF;
        /* Pq is 0 (8-bit entries) or 1 (16-bit); a table never runs past its segment */
        if (id > 3 || prec > 1 || remaining < (prec ? 129 : 65)) {
            return -1;
        }
        next_byte(bs);
        QuantTable *qt = &qts->tables[id];
        qt->precision = prec;
        /* the file lists entries in zigzag order; they are kept in natural order */
        for (int k = 0; k < 64; k++) {
            uint16_t v = next_byte(bs);
            if (prec == 1) {
                v = (uint16_t) ((v << 8) + next_byte(bs));
            }
            qt->table[zigzag[k]] = v;
        }
        remaining -= (prec == 1) ? 129 : 65;
    }
    return 0;
}

int32_t dequant_data_unit(QuantTable *qt, int16_t *du) {
    int16_t in[64];
    memcpy(in, du, sizeof in);
    for (int k = 0; k < 64; k++) {
        uint8_t n = zigzag[k];
        du[n] = (int16_t) (in[k] * qt->table[n]); /* int16 wrap, quant_table.c:137 */
    }
    return 0;
}

/* ---- DRI (restart_interval.c:6-25) ------------------------------------- */
int32_t decode_restart_interval(Bitstream *bs, RestartInterval *ri) {
    if (!ri) {
        return -1;
    }
    if (read_u16(bs) != 4) {
        return -1;
    }
    *ri = read_u16(bs);
    return 0;
}
