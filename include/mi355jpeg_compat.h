/*
 * mi355jpeg_compat.h -- the reference decoder's host API, re-declared for the
 * MI355X library (libmi355jpeg.so).
 *
 * Struct layouts, field names, enumerators and function signatures are those of
 * the reference headers (cited per item, paths relative to the reference's src/),
 * because this is the boundary the library drops in behind: the reference's
 * jpeg_decoder.c calls decode_jpeg_buffer() and then y_rgb()/ycbcr_rgb()
 * (jpeg_decoder.c:55, :95-98) and frees with free_img()/free_frame_header()
 * (:102-104).  Here those entry points parse on the host and run the entropy
 * decode, dequantise+IDCT and upsample+colour stages as HIP kernels on the GPU.
 * There is no CPU decode path behind them: without a usable HIP device they fail
 * with -1 and a message on stderr.
 *
 * Error convention (same as the reference): 0 = ok, -1 = failure.
 * Threading: one host thread per mj context; the drop-in entry points below use a
 * process-wide default context guarded by a mutex.
 */
#ifndef MI355JPEG_COMPAT_H
#define MI355JPEG_COMPAT_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- bit/byte cursor: bitstream.h:4-22 --------------------------------- */
typedef struct {
    uint8_t *encoded_data; /* current byte */
    uint8_t offset;        /* bits already consumed in *encoded_data, 0..7 */
    uint32_t size;         /* bytes consumed by next_byte() so far */
} Bitstream;

uint8_t get_byte(Bitstream *bs);                    /* peek, bitstream.c:4-6 */
uint8_t next_byte(Bitstream *bs);                   /* read + advance, :8-14 */
uint8_t next_bit(Bitstream *bs);                    /* MSB first, FF00 unstuffing, :61-67 */
int16_t next_bit_size(Bitstream *bs, uint8_t size); /* n-bit magnitude, :69-78 */
uint8_t check_marker(Bitstream *bs);                /* 0 none / 1 RSTn / 2 other, :84-134 */
int next_byte_restart_marker(Bitstream *bs);        /* step past FF Dn, :142-176 */
int progress(Bitstream *bs, int len);               /* skip len bytes, :178-182 */

/* ---- frame header: frame_header.h:5-57 --------------------------------- */
typedef enum {
    BDCT = 0,     /* baseline DCT, Huffman */
    ESDCTHC = 1,  /* extended sequential DCT, Huffman */
    PDCTHC = 2,   /* progressive DCT, Huffman */
    LSHC = 3,     /* lossless, Huffman */
    DSDCTHC = 5,  /* differential variants */
    DPDCTHC = 6,
    DLSHC = 7,
    ESDCTAC = 9,  /* arithmetic-coded variants */
    PDCTAC = 10,
    LSAC = 11,
    DSDCTAC = 13,
    DPDCTAC = 14,
    DLSAC = 15,
} Encoding;

char *encoding_str(Encoding process);

typedef struct {
    uint8_t id;
    uint8_t hsf;  /* horizontal sampling factor */
    uint8_t vsf;  /* vertical sampling factor */
    uint16_t x;   /* component width in samples, ceil(X*hsf/hmax) */
    uint16_t y;   /* component height in samples */
    uint8_t qtid; /* quantisation table selector */
} Component;

typedef struct {
    Encoding process;
    uint8_t precision;
    uint16_t X;
    uint16_t Y;
    uint8_t ncs;   /* number of components */
    Component *cs; /* heap array, released by free_frame_header() */
} FrameHeader;

int32_t decode_frame_header(Encoding encoding_process, Bitstream *bs, FrameHeader *fh);
int32_t decode_number_of_lines(Bitstream *bs, FrameHeader *fh);
int32_t free_frame_header(FrameHeader *fh);
void print_frame_header(FrameHeader *fh);
void print_component(Component *comp, int len);

/* ---- scan header: scan_header.h:5-24 ----------------------------------- */
typedef struct {
    uint8_t sc; /* component selector */
    uint8_t dc; /* DC table selector */
    uint8_t ac; /* AC table selector */
} ImageComponent;

typedef struct {
    uint8_t nics; /* components in this scan */
    ImageComponent ics[4];
    uint8_t ss;
    uint8_t se;
    uint8_t ah;
    uint8_t al;
} ScanHeader;

int32_t decode_scan_header(Bitstream *bs, ScanHeader *sh);
void print_scan_header(ScanHeader *sh);
void print_image_component(ImageComponent *ics, int len);

/* ---- Huffman tables: huff_table.h:7-55 --------------------------------- */
#define MAX_CODE_LEN 16
#define MAX_TABLES 4

typedef struct {
    int16_t min_codes[MAX_CODE_LEN]; /* first code of each length, -1 if none */
    int16_t max_codes[MAX_CODE_LEN]; /* last code of each length, -1 if none */
    uint8_t len[MAX_CODE_LEN];       /* number of codes of each length */
    uint8_t *symbols[MAX_CODE_LEN];   /* per-length symbol lists (heap) */
} HuffTable;

enum TableClass { TC_DC, TC_AC, TC_NUM };

typedef struct {
    uint8_t nDCAC;                      /* usable table ids per class */
    HuffTable DCAC[TC_NUM][MAX_TABLES]; /* [0][] DC tables, [1][] AC tables */
} HuffTables;

int new_huff_tables(Encoding process, HuffTables *hts);
int32_t decode_huff_tables(Bitstream *bs, HuffTables *hts);
int32_t free_huff_tables(HuffTables *hts);
void print_huff_tables(HuffTables *hts);
void print_huff_table(HuffTable *ht);

/* ---- quantisation tables: quant_table.h:5-30 --------------------------- */
extern const uint8_t zigzag[]; /* zigzag position -> natural index, quant_table.c:25-34 */

typedef struct {
    uint8_t precision;
    uint16_t table[64]; /* natural (row-major) order */
} QuantTable;

typedef struct {
    QuantTable tables[4];
} QuantTables;

int32_t decode_quant_table(Bitstream *bs, QuantTables *qts);
/* Host-side single-block helper kept for API completeness (quant_table.c:131-152);
 * the decode path dequantises on the GPU inside the IDCT kernel and never calls it. */
int32_t dequant_data_unit(QuantTable *qt, int16_t *du);

/* ---- restart interval: restart_interval.h:4-7 -------------------------- */
typedef uint16_t RestartInterval;
int32_t decode_restart_interval(Bitstream *bs, RestartInterval *ri);

/* ---- image + top-level decode: decode.h:12-21 -------------------------- */
typedef struct {
    uint8_t **buf; /* one plane per component, padded to whole MCUs */
    int *width;    /* plane strides */
    uint16_t mcu;  /* MCUs written (reference bookkeeping; saturates here) */
    uint8_t n_du;  /* number of components */
} Image;

Image *allocate_img(FrameHeader *fh);
void free_img(Image *img);

/* decode.c:138-424.  Caller sets fh->process = -1 first (jpeg_decoder.c:54).
 * On success *img is a heap Image whose planes hold the decoded samples and the
 * interleaved RGB stays resident on the GPU, keyed by *img, until free_img(). */
int decode_jpeg_buffer(uint8_t *buf, size_t len, FrameHeader *fh, Image **img);

/* The SOS seam, decode.c:535-543 (prototype decode.c:37-45): decodes one
 * interleaved baseline scan starting at bs into img.  On return bs sits on the
 * marker that ended the entropy-coded segment. */
int decode_scan(Bitstream *bs, Image *img, FrameHeader *fh, ScanHeader *sh, HuffTables *hts,
                QuantTables *qts, RestartInterval ri);

/* ---- upsample + colour: colour_conversion.h:4-28 ----------------------- */
/* pixels: caller-owned RGB24, `pitch` bytes per row (may exceed 3*width). */
int y_rgb(FrameHeader *fh, Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch);
int ycbcr_rgb(FrameHeader *fh, Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch);
/* colour_conversion.h:20-27 / colour_conversion.c:85-162: four-component (YCCK) files.
 * The reference's viewer names it (jpeg_decoder.c:99-101), so the symbol exists; the GPU
 * path decodes one- and three-component frames only, and this always returns -1 with a
 * message on stderr (decode_jpeg_buffer has already refused such a file by then). */
int yccb_rgb(FrameHeader *fh, Image *img, uint8_t *pixels, uint16_t width, uint16_t height, uint16_t pitch);
int get_max_sf(FrameHeader *fh, uint8_t *max_hsf, uint8_t *max_vsf);

#ifdef __cplusplus
}
#endif
#endif /* MI355JPEG_COMPAT_H */
