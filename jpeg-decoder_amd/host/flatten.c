/*
 * flatten.c -- from the reference's parsed structs to device descriptors.
 *
 * mj_parse_headers() is the marker walk of decode.c:160-415 restricted to what a
 * baseline scan needs; it drives the same segment parsers the reference calls
 * (decode_frame_header, decode_huff_tables, decode_quant_table,
 * decode_restart_interval, decode_scan_header).  Where the reference then calls
 * decode_scan() (decode.c:333), this library records where the entropy-coded
 * bytes start and hands them to the GPU.
 */
#include <stdlib.h>
#include <string.h>

#include "mj_host.h"

static int fail(const char **why, const char *msg) {
    if (why) {
        *why = msg;
    }
    return -1;
}

void mj_parsed_free(MjParsed *p);

/* a rejected file leaves nothing allocated behind */
static int fail_free(MjParsed *out, const char **why, const char *msg) {
    mj_parsed_free(out);
    return fail(why, msg);
}

/* The marker walk from offset `from` up to and including the next scan header.  0: a scan
 * header was read (out->sh, out->ent_pos); 1: the file ended first; -1: rejected.  `resumed`:
 * the walk continues after a scan of the same frame, which a frame header may not follow. */
static int walk_to_scan(const uint8_t *buf, size_t len, size_t from, MjParsed *out, const char **why, int resumed) {
    Bitstream bs = {(uint8_t *) buf + from, 0, 0};
    const uint8_t *end = buf + len;
    while (bs.encoded_data + 1 < end) {
        if (next_byte(&bs) != 0xFF) {
            continue;
        }
        uint8_t m = get_byte(&bs);
        if (m == 0xFF) {
            continue; /* fill byte before a marker */
        }
        next_byte(&bs);
        if (m == 0x00 || m == 0x01 || m == 0xD8 || (m >= 0xD0 && m <= 0xD7)) {
            continue; /* stand-alone markers */
        }
        if (m == 0xD9) {
            break;
        }
        if (bs.encoded_data + 2 > end) {
            return fail_free(out, why, "truncated segment header");
        }
        size_t seglen = ((size_t) bs.encoded_data[0] << 8) | bs.encoded_data[1];
        uint8_t *seg = bs.encoded_data;
        if (seglen < 2 || seg + seglen > end) {
            return fail_free(out, why, "segment length runs past the end of the file");
        }
        switch (m) {
            case 0xC0: /* SOF0 baseline */
            case 0xC1: /* SOF1 extended sequential, handled like baseline (decode.c:187-201) */
                if (resumed) {
                    return fail_free(out, why, "a second frame header");
                }
                if (out->have_fh) {
                    free_frame_header(&out->fh);
                }
                if (decode_frame_header(m == 0xC0 ? BDCT : ESDCTHC, &bs, &out->fh) != 0) {
                    return fail_free(out, why, "bad frame header");
                }
                out->have_fh = 1;
                break;
            case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB:
            case 0xCD: case 0xCE: case 0xCF:
                return fail_free(out, why, "not a baseline sequential Huffman JPEG (progressive/lossless/arithmetic)");
            case 0xC4:
                if (!out->have_hts) {
                    new_huff_tables(out->fh.process, &out->hts); /* decode.c:273-279 */
                    out->have_hts = 1;
                }
                if (decode_huff_tables(&bs, &out->hts) != 0) {
                    return fail_free(out, why, "bad Huffman table segment");
                }
                break;
            case 0xDB:
                if (decode_quant_table(&bs, &out->qts) != 0) {
                    return fail_free(out, why, "bad quantisation table segment");
                }
                out->have_qts = 1;
                break;
            case 0xDD:
                if (decode_restart_interval(&bs, &out->ri) != 0) {
                    return fail_free(out, why, "bad restart interval segment");
                }
                break;
            case 0xDC:
                if (out->have_fh) {
                    decode_number_of_lines(&bs, &out->fh);
                }
                break;
            case 0xDA:
                if (!out->have_fh) {
                    return fail_free(out, why, "scan before frame header");
                }
                if (!out->have_qts) {
                    return fail_free(out, why, "scan without a quantisation table"); /* decode.c:321-326 */
                }
                if (!out->have_hts) {
                    return fail_free(out, why, "scan without a Huffman table");
                }
                if (decode_scan_header(&bs, &out->sh) != 0) {
                    return fail_free(out, why, "bad scan header");
                }
                out->ent_pos = (size_t) (seg + seglen - buf);
                return 0;
            default:
                break; /* APPn (decode.c:385), COM and the rest: skipped by length */
        }
        bs.encoded_data = seg + seglen;
        bs.offset = 0;
    }
    return 1;
}

int mj_parse_headers(const uint8_t *buf, size_t len, MjParsed *out, const char **why) {
    memset(out, 0, sizeof *out);
    out->fh.process = (Encoding) -1; /* jpeg_decoder.c:54 */
    const int r = walk_to_scan(buf, len, 0, out, why, 0);
    return r == 1 ? fail_free(out, why, "no scan found") : r;
}

int mj_parse_next_scan(const uint8_t *buf, size_t len, size_t from, MjParsed *p, const char **why) {
    return walk_to_scan(buf, len, from, p, why, 1);
}

size_t mj_scan_end(const uint8_t *buf, size_t len, size_t ent_pos) {
    size_t i = ent_pos;
    while (i + 1 < len) {
        const uint8_t *f = (const uint8_t *) memchr(buf + i, 0xFF, len - 1 - i);
        if (!f) {
            break;
        }
        i = (size_t) (f - buf);
        const uint8_t m = buf[i + 1];
        if (m != 0x00 && m != 0xFF && !(m >= 0xD0 && m <= 0xD7)) {
            return i;
        }
        i++;
    }
    return len;
}

void mj_parsed_free(MjParsed *p) {
    if (p->have_fh) {
        free_frame_header(&p->fh);
    }
    if (p->have_hts) {
        free_huff_tables(&p->hts);
    }
    memset(p, 0, sizeof *p);
}

/* the lookup entry of the code that starts the 16 left-aligned bits w, 0 if it is not
 * resolved by the two lookup levels */
static uint32_t entry_of(const MjHuffDev *t, uint32_t w) {
    uint32_t e = t->lut[(w >> (16 - MJ_LUT_BITS)) & (MJ_LUT_SIZE - 1)];
    if (MJ_E_LEN(e) != 0) {
        return e;
    }
    if (e & MJ_LUT_LONG) {
        return t->sub[e & (MJ_SUB_MAX - 1)][w & (MJ_SUB_SIZE - 1)];
    }
    return 0;
}

static void flatten_huff_build(const HuffTable *ht, int is_ac, MjHuffDev *out, MjHuffWalk *walk, MjHuffRun *pair, MjHuffPair *next);

/* The images of a batch mostly carry the same few tables (an encoder's defaults): the
 * last tables built are kept, keyed by what the DHT segment said, and copied instead of
 * built again. */
#define MJ_FLAT_CACHE 8
typedef struct {
    int used, is_ac, full;              /* full: the AC side tables were built too */
    uint8_t len[16];
    uint8_t syms[256];
    MjHuffDev dev;
    MjHuffWalk walk;
    MjHuffRun run;
    MjHuffPair next;
} FlatCacheEntry;
static _Thread_local FlatCacheEntry *flat_cache;   /* MJ_FLAT_CACHE entries, allocated on first use */
static _Thread_local unsigned flat_cache_next;

/* a thread that will flatten no more tables gives its cache back */
void mj_flatten_thread_done(void) {
    free(flat_cache);
    flat_cache = NULL;
    flat_cache_next = 0;
}

void mj_flatten_huff(const HuffTable *ht, int is_ac, MjHuffDev *out, MjHuffWalk *walk, MjHuffRun *pair, MjHuffPair *next) {
    uint8_t len[16], syms[256];
    int nsym = 0, keyed = 1;
    memset(syms, 0, sizeof syms);
    for (int l = 0; l < 16; l++) {
        len[l] = ht->len[l];
        for (int i = 0; i < ht->len[l]; i++) {
            if (!ht->symbols[l] || nsym >= 256) {
                keyed = 0;
                break;
            }
            syms[nsym++] = ht->symbols[l][i];
        }
    }
    if (keyed && !flat_cache) flat_cache = (FlatCacheEntry *) calloc(MJ_FLAT_CACHE, sizeof *flat_cache);
    const int want_full = (pair != NULL && next != NULL);
    if (keyed && flat_cache) {
        for (int e = 0; e < MJ_FLAT_CACHE; e++) {
            const FlatCacheEntry *c = &flat_cache[e];
            if (c->used && c->is_ac == is_ac && (c->full || !want_full) && memcmp(c->len, len, 16) == 0 &&
                memcmp(c->syms, syms, 256) == 0) {
                *out = c->dev;
                *walk = c->walk;
                if (pair) *pair = c->run;
                if (next) *next = c->next;
                return;
            }
        }
    }
    flatten_huff_build(ht, is_ac, out, walk, pair, next);
    if (keyed && flat_cache && (pair == NULL) == (next == NULL)) {
        FlatCacheEntry *c = &flat_cache[flat_cache_next++ % MJ_FLAT_CACHE];
        c->used = 1;
        c->is_ac = is_ac;
        c->full = want_full;
        memcpy(c->len, len, 16);
        memcpy(c->syms, syms, 256);
        c->dev = *out;
        c->walk = *walk;
        if (want_full) {
            c->run = *pair;
            c->next = *next;
        }
    }
}

static void flatten_huff_build(const HuffTable *ht, int is_ac, MjHuffDev *out, MjHuffWalk *walk, MjHuffRun *pair, MjHuffPair *next) {
    memset(out, 0, sizeof *out);
    memset(walk, 0, sizeof *walk);
    if (pair) memset(pair, 0, sizeof *pair);
    if (next) memset(next, 0, sizeof *next);
    walk->maxcode[0] = -1;
    int nvals = 0;
    uint32_t nsub = 0;
    for (int l = 1; l <= 16; l++) {
        int n = ht->len[l - 1];
        if (n == 0 || ht->symbols[l - 1] == NULL) {
            walk->maxcode[l] = -1;
            walk->valoff[l] = 0;
            continue;
        }
        int mincode = (uint16_t) ht->min_codes[l - 1];
        walk->maxcode[l] = mincode + n - 1;
        walk->valoff[l] = nvals - mincode;
        for (int i = 0; i < n && nvals < 256; i++, nvals++) {
            uint8_t sym = ht->symbols[l - 1][i];
            uint32_t code = (uint32_t) (mincode + i);
            uint16_t entry = (uint16_t) mj_lut_entry((uint32_t) l, sym, is_ac);
            walk->vals[nvals] = sym;
            if (l <= MJ_LUT_BITS) {
                /* every first-level index whose leading l bits are this code */
                uint32_t first = code << (MJ_LUT_BITS - l);
                uint32_t count = 1u << (MJ_LUT_BITS - l);
                for (uint32_t j = 0; j < count && first + j < MJ_LUT_SIZE; j++) {
                    out->lut[first + j] = entry;
                }
            } else {
                /* long code: its first MJ_LUT_BITS bits pick a second-level table */
                uint32_t prefix = (code >> (l - MJ_LUT_BITS)) & (MJ_LUT_SIZE - 1);
                uint16_t e = out->lut[prefix];
                uint32_t si;
                if (e != 0 && MJ_E_LEN(e) == 0 && (e & MJ_LUT_LONG)) {
                    si = e & (MJ_SUB_MAX - 1);
                } else if (e == 0 && nsub < MJ_SUB_MAX) {
                    si = nsub++;
                    out->lut[prefix] = (uint16_t) (MJ_LUT_LONG | si);
                } else {
                    /* no room (or a malformed table): the whole prefix goes to the canonical walk */
                    if (e == 0) out->lut[prefix] = (uint16_t) MJ_LUT_WALK;
                    continue;
                }
                uint32_t rem = (uint32_t) l - MJ_LUT_BITS;   /* 1..MJ_SUB_BITS bits of the code index the table */
                uint32_t first = (code & ((1u << rem) - 1u)) << (MJ_SUB_BITS - rem);
                uint32_t count = 1u << (MJ_SUB_BITS - rem);
                for (uint32_t j = 0; j < count && first + j < MJ_SUB_SIZE; j++) {
                    out->sub[si][first + j] = entry;
                }
            }
        }
    }
    /* prefixes no code has: sixteen bits are taken as an end of block (AC) or a zero
     * difference (DC), resolved here so the kernels never see an empty entry */
    const uint16_t none = (uint16_t) mj_lut_entry(16, 0, is_ac);
    for (uint32_t j = 0; j < MJ_LUT_SIZE; j++) {
        if (out->lut[j] == 0) out->lut[j] = none;
    }
    for (uint32_t t = 0; t < MJ_SUB_MAX; t++) {
        for (uint32_t j = 0; j < MJ_SUB_SIZE; j++) {
            if (out->sub[t][j] == 0) out->sub[t][j] = none;
        }
    }
    if (!is_ac || !pair || !next) {
        return;
    }
    /* runs of whole symbols inside MJ_RUN_BITS bits (see MjHuffRun) */
    for (uint32_t idx = 0; idx < MJ_RUN_SIZE; idx++) {
        uint32_t pos = 0, n = 0, used = 0, before = 0, last = 0, steps = 0;
        while (pos < MJ_RUN_BITS) {
            const uint32_t e = entry_of(out, ((idx << (16 - MJ_RUN_BITS)) << pos) & 0xFFFFu);
            const uint32_t len = MJ_E_LEN(e);
            if (len == 0 || len > MJ_RUN_BITS - pos) {
                break;   /* a longer code, or one whose bits are not all known */
            }
            if (n >= 1 && steps > 15) {
                break;
            }
            before = steps;
            last = e & 127u;
            steps += last;
            used = pos + len + ((e >> 7) & 15u);
            pos = used;
            n++;
            if (last == 64) {
                break;   /* end of block: what follows belongs to the next block */
            }
        }
        if (n >= 2 && used <= 31) {
            pair->run[idx] = (uint16_t) ((used << 11) | (before << 7) | last);
        }
    }
    for (uint32_t idx = 0; idx < MJ_LUT_SIZE; idx++) {
        /* the symbol after the first one, for the pass that needs every value */
        const uint32_t e1 = out->lut[idx];
        const uint32_t used1 = MJ_E_LEN(e1) + ((e1 >> 7) & 15u);
        if (MJ_E_LEN(e1) != 0 && (e1 & 127u) != 64u && used1 < MJ_LUT_BITS) {
            const uint32_t e2 = out->lut[(idx << used1) & (MJ_LUT_SIZE - 1)];
            if (MJ_E_LEN(e2) != 0 && MJ_E_LEN(e2) <= MJ_LUT_BITS - used1) {
                next->pair[idx] = (uint16_t) e2;
            }
        }
    }
}

static uint32_t pad_up(uint32_t v, uint32_t unit) {
    return (v % unit) ? v + unit - (v % unit) : v;
}

static uint32_t env_u32(const char *name) {
    const char *e = getenv(name);
    return (e && *e) ? (uint32_t) strtoul(e, NULL, 0) : 0u;
}

/* Tuning overrides of the split (experiments only), read from the environment once per
 * process, not per image. */
static struct {
    int loaded;
    uint32_t min_lane_bits, lane_subs, sub_bits;
} g_tune;

static void tune_load(void) {
    if (__atomic_load_n(&g_tune.loaded, __ATOMIC_ACQUIRE)) return;
    g_tune.min_lane_bits = env_u32("MJ_TUNE_MIN_LANE_BITS");
    g_tune.lane_subs = env_u32("MJ_TUNE_LANE_SUBS");
    g_tune.sub_bits = env_u32("MJ_TUNE_SUB_BITS");
    __atomic_store_n(&g_tune.loaded, 1, __ATOMIC_RELEASE);
}

#ifndef MJ_SPLIT_FACTOR
#define MJ_SPLIT_FACTOR 2
#endif
#define MJ_SMALL_FILE_SEQS 4
void mj_choose_split(uint64_t raw_len, uint32_t n_ivl, uint32_t n_mcu, uint64_t batch_bits, uint32_t chip_lanes, uint32_t *sub_bits,
                     uint32_t *lane_subs) {
    tune_load();
    uint64_t bits = raw_len * 8;
    uint64_t avg = bits / (n_ivl ? n_ivl : 1);
    uint64_t lane_bits;
    uint32_t g;
    /* A batch whose lanes would not give every SIMD a wavefront or two is bound by the time
     * one lane takes, not by the chip: shorter lanes then, down to 256 bits (more lanes to
     * synchronise, all at once).  Measured on 512 x 512, 640 x 480 and 7680 x 4320 files alone
     * and in batches: about one and a half wavefronts per SIMD is where the two costs meet. */
    uint64_t fill = 2048;                /* the lane length that fills the chip with this batch, 256 .. 2048 */
    if (batch_bits && chip_lanes) {
        const uint64_t want = batch_bits / (chip_lanes + chip_lanes / 2);
        fill = 256;
        while (fill < want && fill < 2048) fill *= 2;
        /* ... but not much shorter than an MCU: a decoder falls into step MCU by MCU, and
         * subsequences of a fraction of one only multiply the rounds (half an MCU, rounded
         * down to a power of two, is where 512 x 512 4:2:0 files at 2 bit per pixel and
         * 640 x 480 4:2:2 files at 5 both do best) */
        const uint64_t mcu_bits = n_mcu ? bits / n_mcu : 0;
        while (fill < 2048 && fill * 4 <= mcu_bits) fill *= 2;
    }
    /* one wavefront per interval would mean lanes of this length */
    const uint64_t one_wave = (avg + avg / 16 + MJ_SEQ_LANES - 1) / MJ_SEQ_LANES;
    /* How many sequences an interval is cut into.  Long intervals (for the batch at hand):
     * as many as lanes of the batch's length need. */
    uint64_t k = one_wave > MJ_SPLIT_FACTOR * fill ? (one_wave + fill - 1) / fill : 1;
    /* A file that is one interval, in a batch large enough for long lanes, is cut into up to
     * MJ_SMALL_FILE_SEQS sequences as long as their lanes keep 512 bits: the workgroups of
     * the entropy kernels take their sequences from one image, so a file that is one
     * sequence leaves all but one wavefront of each of its workgroups idle, and its lanes
     * are as long as lanes get (a kernel takes at least as long as its longest lane).
     * Measured on batches of 400 - 1500 files of 24 - 48 KB: 2.69 -> 1.49 ms for 800 files of
     * 24 KB.  Files with restart markers keep the factor: cutting their many short intervals
     * costs more in the cross-sequence pass than the shorter lanes save. */
    if (n_ivl <= 1 && fill >= 1024) {
        uint64_t want_k = one_wave / 512;
        if (want_k > MJ_SMALL_FILE_SEQS) want_k = MJ_SMALL_FILE_SEQS;
        const uint64_t need_k = (one_wave + fill - 1) / fill;
        k = want_k > need_k ? want_k : need_k;
        if (k < 1) k = 1;
    }
    if (k > 1) {
        /* a few sequences of long lanes are made equally full */
        lane_bits = (k <= 8 && fill >= 1024) ? (one_wave + k - 1) / k : fill;
        g = lane_bits > 1024 ? 2 : 1;
    } else {
        /* one wavefront per interval with a little headroom for uneven intervals */
        lane_bits = (avg + avg / 16 + MJ_SEQ_LANES - 1) / MJ_SEQ_LANES;
        /* short intervals: fewer, longer lanes -- every synchronisation round has a fixed
         * cost, and 256 bits per round is where that stops to dominate */
        uint32_t minb = g_tune.min_lane_bits;
        if (!minb) minb = 256;
        if (lane_bits < minb) lane_bits = minb;
        if (lane_bits > 4096) lane_bits = 4096;
        /* records every ~512 bits: a decoder that continues past its own lane stops at the
         * first record it agrees with, so finer records mean less re-decoding */
        g = lane_bits >= 1024 ? 4 : lane_bits >= 512 ? 2 : 1;
    }
    /* tuning overrides (experiments only) */
    uint32_t t = g_tune.lane_subs;
    if (t >= 1 && t <= MJ_MAX_LANE_SUBS) g = t;
    uint64_t s = (lane_bits + g - 1) / g;
    t = g_tune.sub_bits;
    if (t) s = t;
    s = (s + 31) / 32 * 32;
    if (s > 32768) s = 32768;
    *sub_bits = (uint32_t) s;
    *lane_subs = g;
}

/* Everything the Huffman set of a scan is built from, as bytes: the table selectors of
 * the scan components and, for every table one of them names, the code counts and the
 * symbols of the DHT segment.  Equal keys give equal sets. */
size_t mj_huff_key(const MjParsed *p, uint8_t *key) {
    const ScanHeader *sh = &p->sh;
    size_t n = 0;
    int used[TC_NUM][MAX_TABLES] = {{0}};
    key[n++] = sh->nics;
    for (int s = 0; s < sh->nics && s < 4; s++) {
        key[n++] = sh->ics[s].dc;
        key[n++] = sh->ics[s].ac;
        if (sh->ics[s].dc < MAX_TABLES) used[TC_DC][sh->ics[s].dc] = 1;
        if (sh->ics[s].ac < MAX_TABLES) used[TC_AC][sh->ics[s].ac] = 1;
    }
    for (int c = 0; c < TC_NUM; c++) {
        for (int t = 0; t < MAX_TABLES; t++) {
            if (!used[c][t]) continue;
            const HuffTable *ht = &p->hts.DCAC[c][t];
            size_t total = 0;
            memcpy(key + n, ht->len, 16);
            n += 16;
            for (int l = 0; l < 16; l++) {
                const size_t k = ht->symbols[l] ? ht->len[l] : 0;
                if (total + k > 256) break;
                if (k) memcpy(key + n, ht->symbols[l], k);
                n += k;
                total += k;
            }
        }
    }
    return n;
}

int mj_flatten_image(const MjParsed *p, size_t file_len, MjImageDev *img, MjHuffSet *set, const char **why) {
    memset(img, 0, sizeof *img);
    /* set == NULL: the caller already holds the set of an image with the same key */
    if (set) memset(set, 0, sizeof *set);
    uint32_t nac = 0, ntab = 0;
    const FrameHeader *fh = &p->fh;
    const ScanHeader *sh = &p->sh;
    if (fh->precision != 8) {
        return fail(why, "only 8-bit samples are supported");
    }
    if (fh->ncs < 1 || fh->ncs > 3) {     /* two: a scan of two of a frame's three components, taken as an image of its own */
        return fail(why, "only grey and three-component images are supported");
    }
    if (sh->nics != fh->ncs) {
        return fail(why, "only one interleaved scan per image is supported");
    }
    if (fh->X == 0 || fh->Y == 0) {
        return fail(why, "empty frame");
    }
    img->X = fh->X;
    img->Y = fh->Y;
    img->ncs = fh->ncs;
    uint8_t hmax = 0, vmax = 0;
    for (int i = 0; i < fh->ncs; i++) {
        const Component *c = &fh->cs[i];
        if (c->hsf < 1 || c->hsf > 4 || c->vsf < 1 || c->vsf > 4 || c->qtid > 3) {
            return fail(why, "bad sampling factor or quantiser selector");
        }
        if (c->hsf > hmax) hmax = c->hsf;
        if (c->vsf > vmax) vmax = c->vsf;
    }
    img->hmax = hmax;
    img->vmax = vmax;
    for (int i = 0; i < fh->ncs; i++) {
        const Component *c = &fh->cs[i];
        if (hmax % c->hsf || vmax % c->vsf) {
            return fail(why, "sampling factors that do not divide the maximum are not supported");
        }
        img->comp[i].hsf = c->hsf;
        img->comp[i].vsf = c->vsf;
        img->comp[i].pw = pad_up(c->x, 8u * c->hsf);  /* decode.c:108-113 */
        img->comp[i].ph = pad_up(c->y, 8u * c->vsf);
        memcpy(img->qt[i], p->qts.tables[c->qtid].table, sizeof img->qt[i]);
    }
    /* MCU grid from the first scan component, as write_mcu does (decode.c:475-479) */
    img->mcus_x = img->comp[0].pw / (8 * img->comp[0].hsf);
    img->mcus_y = img->comp[0].ph / (8 * img->comp[0].vsf);
    for (int i = 1; i < fh->ncs; i++) {
        if (img->comp[i].pw / (8 * img->comp[i].hsf) != img->mcus_x ||
            img->comp[i].ph / (8 * img->comp[i].vsf) != img->mcus_y) {
            return fail(why, "components disagree on the MCU grid");
        }
    }
    img->n_mcu = img->mcus_x * img->mcus_y;

    /* block slots of one MCU in scan order, and the Huffman tables they use: AC tables
     * take the first indices, DC tables follow */
    uint32_t bpm = 0;
    int dc_slot[4] = {-1, -1, -1, -1}, ac_slot[4] = {-1, -1, -1, -1};
    for (int s = 0; s < sh->nics; s++) {
        int td = sh->ics[s].dc, ta = sh->ics[s].ac;
        if (td > 3 || ta > 3) {
            return fail(why, "bad Huffman table selector");
        }
        if (ac_slot[ta] < 0) {
            ac_slot[ta] = (int) nac++;
        }
    }
    ntab = nac;
    for (int s = 0; s < sh->nics; s++) {
        int td = sh->ics[s].dc;
        if (dc_slot[td] < 0) {
            dc_slot[td] = (int) ntab++;
        }
    }
    if (set) {
        set->nac = nac;
        set->ntab = ntab;
    }
    for (int t = 0; t < 4 && set; t++) {
        if (ac_slot[t] >= 0) {
            mj_flatten_huff(&p->hts.DCAC[TC_AC][t], 1, &set->tab[ac_slot[t]], &set->walk[ac_slot[t]], &set->run[ac_slot[t]],
                            &set->next[ac_slot[t]]);
        }
        if (dc_slot[t] >= 0) {
            mj_flatten_huff(&p->hts.DCAC[TC_DC][t], 0, &set->tab[dc_slot[t]], &set->walk[dc_slot[t]], NULL, NULL);
        }
    }
    for (int s = 0; s < sh->nics; s++) {
        int ci = -1;
        for (int j = 0; j < fh->ncs; j++) {
            if (fh->cs[j].id == sh->ics[s].sc) { /* decode.c:595-604 */
                ci = j;
                break;
            }
        }
        if (ci < 0) {
            return fail(why, "scan names an unknown component");
        }
        int td = sh->ics[s].dc, ta = sh->ics[s].ac;
        const Component *c = &fh->cs[ci];
        for (int v = 0; v < c->vsf; v++) {
            for (int h = 0; h < c->hsf; h++) {
                if (bpm >= MJ_MAX_BPM) {
                    return fail(why, "more than 10 blocks per MCU");
                }
                img->blk_comp[bpm] = (uint8_t) ci;
                img->blk_dc[bpm] = (uint8_t) dc_slot[td];
                img->blk_ac[bpm] = (uint8_t) ac_slot[ta];
                img->blk_h[bpm] = (uint8_t) h;
                img->blk_v[bpm] = (uint8_t) v;
                bpm++;
            }
        }
    }
    img->bpm = bpm;
    img->ri = p->ri ? p->ri : img->n_mcu;
    img->n_ivl = (img->n_mcu + img->ri - 1) / img->ri;
    img->raw_len = file_len - p->ent_pos;
    mj_choose_split(img->raw_len, img->n_ivl, img->n_mcu, 0, 0, &img->sub_bits, &img->lane_subs);   /* the caller may choose again once it knows the batch */
    img->pitch = (3 * img->X + 3u) & ~3u; /* rows start 4-byte aligned */
    return 0;
}
