/*
 * mj_device.h -- descriptors shared by the host side (C, flatten.c) and the HIP
 * kernels.  Plain C structs with fixed-width members; the host fills them, they are
 * copied to HBM once per batch, and every kernel indexes them by image.
 */
#ifndef MJ_DEVICE_H
#define MJ_DEVICE_H

#include <stdint.h>

#define MJ_MAX_COMP 4
#define MJ_MAX_BPM 10          /* T.81 B.2.3: at most 10 blocks per MCU */
#define MJ_MAX_TABS 8          /* Huffman tables one scan can reference */
#define MJ_LUT_BITS 10
#define MJ_LUT_SIZE (1 << MJ_LUT_BITS)
#define MJ_SEQ_LANES 64        /* lanes per wavefront; a sequence is MJ_SEQ_LANES * lane_subs subsequences */
#define MJ_MAX_LANE_SUBS 8
#define MJ_SCAN_TILE 8192      /* raw bytes per scan-kernel workgroup (a multiple of 4096) */
#define MJ_STREAM_PAD 64       /* readable zero bytes after every compacted stream */
#define MJ_NO_TERMINATOR 0xFFFFFFFFu

#define MJ_SUB_BITS (16 - MJ_LUT_BITS)
#define MJ_SUB_SIZE (1 << MJ_SUB_BITS)
#define MJ_SUB_MAX 8           /* second-level tables per Huffman table */
/* Lookup entry, 16 bits, what one code means for the decoder:
 *   bits 15-11  code length 1..16
 *   bits 10-7   size: value bits that follow the code
 *   bits 6-0    how far the zigzag index moves: run + 1 for an AC (run, size) symbol,
 *               64 for end of block, 1 for a DC category
 * A first-level entry with length 0 is not a code: */
#define MJ_E_LEN(e) ((uint32_t) (e) >> 11)
#define MJ_LUT_LONG 0x0400u    /* code longer than MJ_LUT_BITS; low bits = second-level table */
#define MJ_LUT_WALK 0x0200u    /* longer code that found no second-level table: canonical walk */

/* natural (row-major) position of the coefficient with zigzag index k (quant_table.c:25-34) */
#define MJ_NATPOS_INIT                                                                                                   \
    {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, \
     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63}

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MJ_DEV_INLINE static inline __host__ __device__
#else
#define MJ_DEV_INLINE static inline
#endif

MJ_DEV_INLINE uint32_t mj_lut_entry(uint32_t len, uint32_t sym, int is_ac) {
    const uint32_t kinc = !is_ac ? 1u : (sym == 0 ? 64u : (sym >> 4) + 1u);
    return (len << 11) | ((sym & 15u) << 7) | kinc;
}

/* One Huffman table, flattened from the reference's HuffTable (huff_table.h:9-14):
 * a direct lookup on the next MJ_LUT_BITS bits for short codes; for longer codes a
 * second-level lookup on the following MJ_SUB_BITS bits, selected by the first-level
 * entry; and the canonical ranges (decode.c:674-681 compares against max_codes per
 * length) as the fallback for tables that need more second-level tables than fit. */
typedef struct {
    uint16_t lut[MJ_LUT_SIZE]; /* entry; MJ_LUT_LONG | n; MJ_LUT_WALK */
    uint16_t sub[MJ_SUB_MAX][MJ_SUB_SIZE]; /* entry */
} MjHuffDev;                   /* the part the kernels keep in LDS */

typedef struct {
    int32_t maxcode[17];       /* [len] last code of that length (right-justified), -1 if none */
    int32_t valoff[17];        /* [len] index into vals of code 0 of that length: valptr - mincode */
    uint8_t vals[256];
    uint32_t pad_[2];
} MjHuffWalk;                  /* canonical ranges, read from HBM only on MJ_LUT_WALK entries */

/* Runs of AC symbols that one first-level lookup resolves at once, for the passes that
 * only count (no coefficient values): entry i describes the longest run of two or more
 * whole symbols whose codes lie inside the MJ_RUN_BITS bits i (value bits may stick
 * out), ending at the latest with an end of block:
 *   bits 15-11  bits the run consumes
 *   bits 10-7   zigzag steps of all its symbols but the last (at most 15)
 *   bits 6-0    zigzag step of the last symbol (64 for end of block)
 * 0 = no such run.  The run may be taken only if the block does not end before its
 * last symbol, i.e. k + (bits 10-7) < 64. */
#define MJ_RUN_BITS 12
#define MJ_RUN_SIZE (1 << MJ_RUN_BITS)
typedef struct {
    uint16_t run[MJ_RUN_SIZE];    /* indexed by the next MJ_RUN_BITS bits */
} MjHuffRun;

typedef struct {
    uint16_t pair[MJ_LUT_SIZE];   /* indexed like the first-level lookup */
} MjHuffPair;

/* The tables one scan uses.  AC tables come first (indices 0 .. nac-1), DC tables after
 * them, so the per-AC-table side tables below are dense. */
typedef struct {
    uint32_t ntab;
    uint32_t nac;
    uint32_t pad_[2];
    MjHuffDev tab[MJ_MAX_TABS];
    MjHuffWalk walk[MJ_MAX_TABS];
    MjHuffRun run[MJ_MAX_TABS / 2];     /* counting passes: runs of whole AC symbols */
    MjHuffPair next[MJ_MAX_TABS / 2];   /* writing pass: lookup entry of the AC symbol that follows the one the index
                                           starts with, when its code lies inside the index bits too; 0 = none */
} MjHuffSet;

typedef struct {
    uint32_t hsf, vsf;   /* sampling factors */
    uint32_t pw, ph;     /* plane size padded to whole MCUs (decode.c:108-115) */
    uint64_t plane_off;  /* byte offset of the plane in the batch plane buffer */
} MjCompDev;

typedef struct {
    /* frame */
    uint32_t X, Y, ncs, hmax, vmax;
    uint32_t mcus_x, mcus_y, n_mcu, bpm;
    uint32_t ri;         /* MCUs per restart interval; n_mcu when the file has no DRI */
    uint32_t n_ivl;      /* ceil(n_mcu / ri) */
    uint32_t sub_bits;   /* subsequence length in bits, multiple of 32: one record per subsequence */
    uint32_t huff_set;   /* index into the batch's MjHuffSet array */
    uint32_t pitch;      /* RGB row pitch in bytes */
    /* per block slot of one MCU, in scan order (decode.c:591-611) */
    uint8_t blk_comp[MJ_MAX_BPM + 2];   /* frame component of the slot */
    uint8_t blk_dc[MJ_MAX_BPM + 2];     /* table index (into MjHuffSet.tab) for the DC symbol */
    uint8_t blk_ac[MJ_MAX_BPM + 2];     /* ... for AC symbols */
    uint8_t blk_h[MJ_MAX_BPM + 2];      /* block column inside the MCU for its component */
    uint8_t blk_v[MJ_MAX_BPM + 2];      /* block row inside the MCU */
    MjCompDev comp[MJ_MAX_COMP];
    uint16_t qt[MJ_MAX_COMP][64];       /* per component, natural order (quant_table.c:108-114) */
    /* placement inside the batch buffers */
    uint64_t raw_off;    /* entropy-coded bytes (stuffed, with RSTn) in the raw buffer */
    uint64_t raw_len;
    uint64_t cmp_off;    /* unstuffed, marker-free stream; 16-byte aligned */
    uint64_t coef_off;   /* first 8x8 block in the coefficient buffer */
    uint64_t rgb_off;    /* first byte in the RGB buffer */
    uint32_t tile_off;   /* first scan tile; n_tiles = ceil(raw_len / MJ_SCAN_TILE) */
    uint32_t n_tiles;
    uint32_t ivl_off;    /* first entry in the interval-start table (n_ivl + 1 entries) */
    uint32_t seq_off;    /* first sequence slot */
    uint32_t seq_cap;    /* sequence slots reserved (upper bound) */
    uint32_t wg_first;   /* unused */
    uint32_t multi_cap;  /* upper bound on intervals longer than one sequence */
    uint32_t lane_subs;  /* consecutive subsequences one lane decodes (1..MJ_MAX_LANE_SUBS) */
    uint32_t rec_off;    /* first subsequence record of the image; sequence q starts at rec_off + q * 64 * lane_subs */
    uint32_t idct_tiles_x;    /* tiles of the IDCT + colour kernel per MCU row */
    uint32_t idct_tile_mcus;  /* MCUs per tile (the last tile of a row may hold fewer) */
    uint32_t pad_;
} MjImageDev;

/* Planes -> RGB for one frame: where its component planes lie and where its RGB goes.  Used for
 * frames coded in one scan per component (the planes their scans were decoded into; the last
 * kernel of the decode combines them) and for ycbcr_rgb / y_rgb on planes the caller filled. */
typedef struct {
    uint64_t plane_off[3];   /* byte offsets in the batch plane buffer, by component */
    uint64_t rgb_off;        /* first byte in the RGB buffer */
    uint32_t pw[3];          /* plane strides */
    uint32_t hsf[3], vsf[3]; /* sampling factors of the frame */
    uint32_t hmax, vmax;
    uint32_t X, Y, pitch;
    uint32_t ncs;            /* 3, or 1: grey (R = G = B = component 0) */
} MjCombine;

/* What the IDCT + colour kernel needs to know about an image, in one 64-byte record that a
 * wavefront reads with a single scalar load. */
typedef struct {
    uint64_t coef_off;   /* first 8x8 block in the coefficient buffer */
    uint64_t rgb_off;    /* first byte in the RGB buffer */
    uint32_t X, Y;
    uint32_t pitch;      /* RGB row pitch in bytes */
    uint32_t mcus_x;
    uint32_t geo;        /* bpm | MCUs per tile << 8 | hmax << 16 | vmax << 24 */
    uint32_t slot_comp;  /* frame component of block slot i of the MCU in bits 2i, 2i + 1 */
    uint32_t cls;        /* colour layout (MJ_CL_*) | first block slot of component 0 << 8 | of 1 << 16 | of 2 << 24 */
    uint32_t sf;         /* hsf | vsf << 4 of component c in bits 8c .. 8c + 7 */
    uint32_t first_tile; /* index of the image's first tile in the batch */
    uint32_t n_tiles;    /* tiles per MCU row x MCU rows */
    uint32_t tiles_x;    /* tiles per MCU row */
    uint32_t pad_;
} MjIdctImg;
/* colour layouts: one component; three with luma at full resolution and both chroma
 * components subsampled by 2^SH x 2^SV with one block per MCU (MJ_CL_FAST + SH + 3 * SV);
 * anything else */
enum { MJ_CL_GREY = 0, MJ_CL_FAST = 1, MJ_CL_GENERIC = 7 };

/* Values the scan kernels compute per image (device-written). */
typedef struct {
    uint32_t term_pos;   /* raw offset of the marker that ends the segment, or MJ_NO_TERMINATOR */
    uint32_t cmp_len;    /* bytes in the compacted stream */
    uint32_t n_rst;      /* restart markers found before the terminator */
    uint32_t n_seq;      /* sequences laid out */
    uint32_t status;     /* bit 0 stream overrun, bit 1 missing restart markers */
    uint32_t spec_guessed; /* speculative synchronisation: hops of the image's true chains that were guesses */
    uint32_t pad_[2];
} MjImageDyn;

/* What a decoder knows at a symbol boundary; equal states decode identically
 * from there on.  p is a bit offset in the image's compacted stream. */
typedef struct {
    uint32_t p;
    uint32_t kc;         /* (c << 8) | k : block slot in MCU, zigzag index of next coefficient */
} MjState;

/* Result of decoding one subsequence from some entry state. */
typedef struct {
    MjState exit;
    uint32_t nb;          /* blocks completed */
    int32_t dcsum[MJ_MAX_COMP]; /* sum of DC differences decoded, per component, modulo 2^16 */
    uint32_t pad_;
} MjSubRec;               /* 32 bytes */

/* Per-sequence entry data produced by the cross-sequence fix-up. */
typedef struct {
    MjState entry;
    uint32_t blk;         /* blocks of the interval that precede this sequence */
    int32_t pred[MJ_MAX_COMP];
    uint32_t pad_;
} MjSeqEntry;             /* 32 bytes */

#endif /* MJ_DEVICE_H */
