/*
 * mj_entropy_core.h -- the symbol-level Huffman decoder shared by the entropy
 * kernels (HIP, one lane per subsequence) and by the host-side emulation used in
 * the CPU tests.  One function, mj_decode_span(), decodes from a given state up to
 * a bit position; it is what decode_data_unit() (decode.c:665-723) plus the bit
 * reader (bitstream.c:61-78) do, restated over an unstuffed, marker-free stream
 * and restartable at any symbol boundary.
 */
#ifndef MJ_ENTROPY_CORE_H
#define MJ_ENTROPY_CORE_H

#include <stdint.h>

#include "mj_device.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MJ_HD __host__ __device__ __forceinline__
#else
#define MJ_HD static inline
#endif

/* Wave-uniform description of the scan a lane is decoding. */
struct MjScanCtx {
    const uint32_t *stream;   /* the image's compacted stream as big-endian 32-bit words */
    const MjHuffDev *tabs;    /* the image's Huffman tables (LDS on the device) */
    int16_t *coefs;           /* block 0 of the image; 64 int16 per block, zigzag order */
    uint32_t bpm;             /* blocks per MCU */
    uint64_t dc_sel, ac_sel;  /* 4 bits per block slot: table index for DC / AC symbols */
    uint64_t comp_sel;        /* 4 bits per block slot: component */
};

MJ_HD uint32_t mj_be32(uint32_t v) {
    return __builtin_bswap32(v);
}

MJ_HD uint64_t mj_pack_sel(const uint8_t *v, uint32_t n) {
    uint64_t r = 0;
    for (uint32_t i = 0; i < n; i++) {
        r |= (uint64_t) (v[i] & 15u) << (4 * i);
    }
    return r;
}

/*
 * Decode symbols from state (p, k, c) until p >= end.
 *   p   bit offset in the stream; k  zigzag index of the next coefficient (0 = the
 *   block's DC symbol comes next); c  block slot inside the MCU.
 * nb is advanced by the number of blocks completed.  d0..d3 are per-component DC
 * accumulators: every DC difference is added to its component's accumulator
 * (decode.c:682-690; the int16 wrap is applied when the value is stored).  With
 * WRITE, coefficients are stored for blocks blk <= index < blk_end, where blk is
 * the block the entry state is in; the caller supplies zero-filled blocks
 * (decode.c:612) and the accumulators must hold the predictors on entry.
 */
template <bool WRITE>
MJ_HD void mj_decode_span(const MjScanCtx &cx, uint32_t &p, uint32_t &k, uint32_t &c, uint32_t end, uint32_t &nb,
                          int32_t &d0, int32_t &d1, int32_t &d2, int32_t &d3, uint32_t blk, uint32_t blk_end) {
    if (p >= end) {
        return;
    }
    uint32_t wi = p >> 5;
    uint64_t buf = ((uint64_t) mj_be32(cx.stream[wi]) << 32) | mj_be32(cx.stream[wi + 1]);
    buf <<= (p & 31);
    int32_t avail = 64 - (int32_t) (p & 31);
    wi += 2;
    uint32_t nxt = mj_be32(cx.stream[wi]);
    while (p < end) {
        if (avail <= 32) {
            buf |= (uint64_t) nxt << (32 - avail);
            avail += 32;
            wi++;
            nxt = mj_be32(cx.stream[wi]);
        }
        const bool isac = (k != 0);
        const uint32_t t = (uint32_t) ((isac ? cx.ac_sel : cx.dc_sel) >> (4 * c)) & 15u;
        const MjHuffDev *T = cx.tabs + t;
        const uint32_t w = (uint32_t) (buf >> 48);
        uint32_t e = T->lut[w >> (16 - MJ_LUT_BITS)];
        uint32_t len = e >> 8, sym = e & 255u;
        if (len == 0) {
            /* code longer than the direct table: canonical ranges per length */
            len = 16;
            sym = 0;
            for (uint32_t L = MJ_LUT_BITS + 1; L <= 16; L++) {
                int32_t code = (int32_t) (w >> (16 - L));
                if (code <= T->maxcode[L]) {
                    sym = T->vals[(uint32_t) (T->valoff[L] + code) & 255u];
                    len = L;
                    break;
                }
            }
        }
        const uint32_t size = sym & 15u;
        const uint32_t rest = (uint32_t) ((buf << len) >> 32);
        const uint32_t mag = size ? (rest >> (32 - size)) : 0u;
        /* EXTEND (decode.c:684-686, :716-718) */
        const int32_t val = (size && mag < (1u << (size - 1))) ? (int32_t) mag - (int32_t) (1u << size) + 1 : (int32_t) mag;
        const uint32_t used = len + size;
        buf <<= used;
        avail -= (int32_t) used;
        p += used;
        if (!isac) {
            const uint32_t comp = (uint32_t) (cx.comp_sel >> (4 * c)) & 15u;
            d0 += (comp == 0) ? val : 0;
            d1 += (comp == 1) ? val : 0;
            d2 += (comp == 2) ? val : 0;
            d3 += (comp == 3) ? val : 0;
            if (WRITE && blk < blk_end) {
                const int32_t dc = (comp == 0) ? d0 : (comp == 1) ? d1 : (comp == 2) ? d2 : d3;
                cx.coefs[(uint64_t) blk * 64] = (int16_t) dc;
            }
            k = 1;
        } else if (sym == 0) {
            k = 64; /* EOB */
        } else {
            k += sym >> 4; /* run; ZRL (0xF0) has size 0 and nets +16 */
            if (WRITE && size && k < 64 && blk < blk_end) {
                cx.coefs[(uint64_t) blk * 64 + k] = (int16_t) val;
            }
            k += 1;
        }
        if (k >= 64) {
            k = 0;
            c = (c + 1 == cx.bpm) ? 0 : c + 1;
            nb++;
            blk++;
        }
    }
}

/* Byte classes of the raw entropy-coded segment (bitstream.c:22-59, :84-176). */
enum { MJ_B_KEEP = 0, MJ_B_DROP = 1, MJ_B_RST = 2, MJ_B_TERM = 3 };

/* Classify byte `cur` given its neighbours.  A stuffed 00 after FF, both bytes of an
 * RSTn marker and FF fill bytes are dropped; the FF of an RSTn is reported as
 * MJ_B_RST; an FF followed by anything else (EOI, a table segment...) ends the
 * segment. */
MJ_HD int mj_classify(uint32_t prev, uint32_t cur, uint32_t next) {
    if (prev == 0xFF && (cur == 0x00 || (cur >= 0xD0 && cur <= 0xD7))) {
        return MJ_B_DROP;
    }
    if (cur == 0xFF) {
        if (next == 0x00) {
            return MJ_B_KEEP;
        }
        if (next >= 0xD0 && next <= 0xD7) {
            return MJ_B_RST;
        }
        if (next == 0xFF) {
            return MJ_B_DROP;
        }
        return MJ_B_TERM;
    }
    return MJ_B_KEEP;
}

#endif /* MJ_ENTROPY_CORE_H */
