/*
 * mj_host.h -- internal host-side interface between the marker walk (decode_host.c),
 * the table flattening (flatten.c) and the HIP side (csrc/).
 */
#ifndef MJ_HOST_H
#define MJ_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "mi355jpeg_compat.h"
#include "../csrc/mj_device.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Everything the marker walk collects up to and including the first SOS. */
typedef struct {
    FrameHeader fh;
    ScanHeader sh;
    HuffTables hts;
    QuantTables qts;
    RestartInterval ri;
    int have_fh, have_hts, have_qts;
    size_t ent_pos;   /* offset of the first entropy-coded byte in the file */
} MjParsed;

/* decode.c:160-415: walk the markers of one file with the segment parsers and stop
 * after the scan header.  0 / -1; on failure *why names the reason. */
int mj_parse_headers(const uint8_t *buf, size_t len, MjParsed *out, const char **why);
void mj_parsed_free(MjParsed *p);

/* Frames coded in several scans: where the entropy-coded segment that starts at ent_pos ends
 * (offset of the next marker that is not a stuffed FF00, a fill byte or RSTn; len if none), and
 * the marker walk continued from there with the tables in force (DHT, DQT and DRI segments
 * between scans replace them) up to the next scan header.  0: p->sh and p->ent_pos describe
 * the next scan; 1: no further scan; -1: rejected (p is freed). */
size_t mj_scan_end(const uint8_t *buf, size_t len, size_t ent_pos);
int mj_parse_next_scan(const uint8_t *buf, size_t len, size_t from, MjParsed *p, const char **why);

/* huff_table.h:9-14 -> lookup form. */
void mj_flatten_huff(const HuffTable *ht, int is_ac, MjHuffDev *out, MjHuffWalk *walk, MjHuffRun *pair, MjHuffPair *next);
/* frees the calling thread's cache of flattened tables (threads that end call it) */
void mj_flatten_thread_done(void);

/* What the Huffman set of the scan depends on, as at most MJ_HUFF_KEY_BYTES bytes (returns
 * how many): images with equal keys share one set, and only the first needs it built. */
#define MJ_HUFF_KEY_BYTES (16 + 8 * (16 + 256))
size_t mj_huff_key(const MjParsed *p, uint8_t *key);

/* Geometry, per-slot table selection, quantiser tables and the Huffman set of one
 * image (set may be NULL when the caller already has the set for this image's key).
 * Batch placement fields are left zero.  0 / -1. */
int mj_flatten_image(const MjParsed *p, size_t file_len, MjImageDev *img, MjHuffSet *set, const char **why);

/* How an entropy segment of raw_len bytes in n_ivl restart intervals is cut up:
 * subsequence length (bits, multiple of 32) and subsequences per lane.  n_mcu: MCUs in the
 * segment; batch_bits: the
 * entropy-coded bits of the whole batch the image is decoded with, chip_lanes: how many
 * lanes give every SIMD of the device one wavefront; both 0 when not known. */
void mj_choose_split(uint64_t raw_len, uint32_t n_ivl, uint32_t n_mcu, uint64_t batch_bits, uint32_t chip_lanes, uint32_t *sub_bits,
                     uint32_t *lane_subs);

#ifdef __cplusplus
}
#endif
#endif
