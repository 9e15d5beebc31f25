/*
 * mj_entropy_core.h -- the symbol-level Huffman decoder shared by the entropy
 * kernels (HIP, one lane per subsequence) and by the host-side emulation used in
 * the CPU tests.  mj_step() decodes one symbol; it is what decode_data_unit()
 * (decode.c:665-723) plus the bit reader (bitstream.c:61-78) do per symbol,
 * restated over an unstuffed, marker-free stream and restartable at any symbol
 * boundary.  mj_decode_span() runs it from a given state up to a bit position.
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

/* Bit window over the stream: the next `avail` (>= 33 after a refill) bits are the
 * top bits of buf.  nxt_raw is the following word, loaded one refill ahead and left
 * byte-swapped-later so that the load's latency overlaps several symbols. */
struct MjBits {
    uint64_t buf;
    int32_t avail;
    uint32_t wi;
    uint32_t nxt_raw;
};

MJ_HD void mj_bits_init(MjBits &bw, const uint32_t *stream, uint32_t p) {
    uint32_t wi = p >> 5;
    bw.buf = (((uint64_t) mj_be32(stream[wi]) << 32) | mj_be32(stream[wi + 1])) << (p & 31);
    bw.avail = 64 - (int32_t) (p & 31);
    bw.wi = wi + 2;
    bw.nxt_raw = stream[bw.wi];
}

MJ_HD void mj_bits_refill(MjBits &bw, const uint32_t *stream) {
    if (bw.avail <= 32) {
        bw.buf |= (uint64_t) mj_be32(bw.nxt_raw) << (32 - bw.avail);
        bw.avail += 32;
        bw.wi++;
        bw.nxt_raw = stream[bw.wi];
    }
}

/* What one symbol did. */
struct MjSym {
    int32_t val;      /* DC difference, or AC coefficient value */
    uint32_t kpos;    /* zigzag position the value belongs to (0 for DC) */
    uint32_t comp;    /* component, valid when dc */
    bool dc;          /* the symbol was the block's DC */
    bool store;       /* val is a coefficient to store (DC always; AC when size != 0 and in range) */
    bool done;        /* the block ended with this symbol */
};

/*
 * Decode one symbol at state (p, k, c) and advance the state.
 *   p  bit offset in the stream; k  zigzag index of the next coefficient (0: the
 *   block's DC symbol comes next); c  block slot inside the MCU.
 * The caller has refilled bw (mj_bits_refill) since the previous step.
 */
MJ_HD MjSym mj_step(const MjScanCtx &cx, MjBits &bw, uint32_t &p, uint32_t &k, uint32_t &c) {
    MjSym s;
    const bool isac = (k != 0);
    const uint32_t t = (uint32_t) ((isac ? cx.ac_sel : cx.dc_sel) >> (4 * c)) & 15u;
    const MjHuffDev *T = cx.tabs + t;
    const uint32_t w = (uint32_t) (bw.buf >> 48);
    const uint32_t e = T->lut[w >> (16 - MJ_LUT_BITS)];
    uint32_t len = e >> 8, sym = e & 255u;
    if (len == 0) {
        /* code longer than the direct table: canonical ranges per length
         * (the reference compares against max_codes per length, decode.c:674-681) */
        len = 16;
        sym = 0;
        for (uint32_t L = MJ_LUT_BITS + 1; L <= 16; L++) {
            const int32_t code = (int32_t) (w >> (16 - L));
            if (code <= T->maxcode[L]) {
                sym = T->vals[(uint32_t) (T->valoff[L] + code) & 255u];
                len = L;
                break;
            }
        }
    }
    const uint32_t size = sym & 15u;
    const uint32_t rest = (uint32_t) ((bw.buf << len) >> 32);
    const uint32_t mag = size ? (rest >> (32 - size)) : 0u;
    /* EXTEND (decode.c:684-686, :716-718) */
    s.val = (size && mag < (1u << (size - 1))) ? (int32_t) mag - (int32_t) (1u << size) + 1 : (int32_t) mag;
    const uint32_t used = len + size;
    bw.buf <<= used;
    bw.avail -= (int32_t) used;
    p += used;
    s.dc = !isac;
    s.comp = (uint32_t) (cx.comp_sel >> (4 * c)) & 15u;
    if (!isac) {
        s.kpos = 0;
        s.store = true;
        k = 1;
    } else if (sym == 0) {
        s.kpos = 0;
        s.store = false;
        k = 64; /* EOB */
    } else {
        k += sym >> 4; /* run; ZRL (0xF0) has size 0 and nets +16 */
        s.kpos = k;
        s.store = (size != 0) && (k < 64);
        k += 1;
    }
    s.done = (k >= 64);
    if (s.done) {
        k = 0;
        c = (c + 1 == cx.bpm) ? 0 : c + 1;
    }
    return s;
}

/*
 * Decode symbols from state (p, k, c) until p >= end.  nb is advanced by the number
 * of blocks completed.  d0..d3 are per-component DC accumulators: every DC
 * difference is added to its component's accumulator (decode.c:682-690; the int16
 * wrap is applied when the value is stored).  With WRITE, coefficients are stored
 * straight to cx.coefs for blocks blk <= index < blk_end, where blk is the block the
 * entry state is in; the caller supplies zero-filled blocks (decode.c:612) and the
 * accumulators must hold the predictors on entry.
 */
template <bool WRITE>
MJ_HD void mj_decode_span(const MjScanCtx &cx, uint32_t &p, uint32_t &k, uint32_t &c, uint32_t end, uint32_t &nb,
                          int32_t &d0, int32_t &d1, int32_t &d2, int32_t &d3, uint32_t blk, uint32_t blk_end) {
    if (p >= end) {
        return;
    }
    MjBits bw;
    mj_bits_init(bw, cx.stream, p);
    while (p < end) {
        mj_bits_refill(bw, cx.stream);
        const MjSym s = mj_step(cx, bw, p, k, c);
        if (s.dc) {
            d0 += (s.comp == 0) ? s.val : 0;
            d1 += (s.comp == 1) ? s.val : 0;
            d2 += (s.comp == 2) ? s.val : 0;
            d3 += (s.comp == 3) ? s.val : 0;
            if (WRITE && blk < blk_end) {
                const int32_t dc = (s.comp == 0) ? d0 : (s.comp == 1) ? d1 : (s.comp == 2) ? d2 : d3;
                cx.coefs[(uint64_t) blk * 64] = (int16_t) dc;
            }
        } else if (WRITE && s.store && blk < blk_end) {
            cx.coefs[(uint64_t) blk * 64 + s.kpos] = (int16_t) s.val;
        }
        if (s.done) {
            nb++;
            blk++;
        }
    }
}

/*
 * One step of the writing pass of one lane.  The lane assembles the block it is in
 * inside blkbuf (64 int16, zero-filled by the caller) and reports `fin` when the
 * block is complete and should be copied to block index fin_blk of the image.
 *
 * Ownership: a block belongs to the lane that decodes its DC symbol.  A lane whose
 * entry state is in the middle of a block (k != 0) runs in `skip` mode until that
 * block ends -- the previous lane finishes it by decoding past its own subsequence
 * end (see mj_write_live) -- so every block is assembled by exactly one lane and
 * leaves the chip as one full 128-byte line.
 */
MJ_HD void mj_write_step(const MjScanCtx &cx, MjBits &bw, uint32_t &p, uint32_t &k, uint32_t &c, bool &skip,
                         uint32_t &blk, uint32_t blk_end, int32_t &d0, int32_t &d1, int32_t &d2, int32_t &d3,
                         int16_t *blkbuf, bool &fin, uint32_t &fin_blk) {
    mj_bits_refill(bw, cx.stream);
    const MjSym s = mj_step(cx, bw, p, k, c);
    const bool mine = !skip && blk < blk_end;
    if (s.dc) {
        d0 += (s.comp == 0) ? s.val : 0;
        d1 += (s.comp == 1) ? s.val : 0;
        d2 += (s.comp == 2) ? s.val : 0;
        d3 += (s.comp == 3) ? s.val : 0;
        if (mine) {
            const int32_t dc = (s.comp == 0) ? d0 : (s.comp == 1) ? d1 : (s.comp == 2) ? d2 : d3;
            blkbuf[0] = (int16_t) dc;
        }
    } else if (s.store && mine) {
        blkbuf[s.kpos] = (int16_t) s.val;
    }
    fin = false;
    if (s.done) {
        if (mine) {
            fin = true;
            fin_blk = blk;
        }
        skip = false;
        blk++;
    }
}

/* Is the lane still decoding in the writing pass?  Up to its subsequence end, and
 * beyond it while it is inside a block it owns (never past the interval's end). */
MJ_HD bool mj_write_live(uint32_t p, uint32_t k, bool skip, uint32_t end, uint32_t hard_end) {
    return (p < end) || (k != 0 && !skip && p < hard_end);
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
