/*
 * mj_entropy_core.h -- the symbol-level Huffman decoder shared by the entropy
 * kernels (HIP, one lane per subsequence) and by the host-side emulation used in
 * the CPU tests.  mj_step() decodes one symbol; it is what decode_data_unit()
 * (decode.c:665-723) plus the bit reader (bitstream.c:61-78) do per symbol,
 * restated over an unstuffed, marker-free stream and restartable at any symbol
 * boundary.
 *
 * The code is written for the GPU's instruction budget: every lane of a wavefront
 * executes every "rare" branch that any lane takes, so the per-symbol path is
 * branch-light, 32-bit only, and keeps per-block work (table selection) out of it.
 */
#ifndef MJ_ENTROPY_CORE_H
#define MJ_ENTROPY_CORE_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

#include "mj_device.h"

#if defined(__HIPCC__)
#define MJ_HD __host__ __device__ __forceinline__
#define MJ_HDM __host__ __device__ __forceinline__
#else
#define MJ_HD static inline
#define MJ_HDM inline
#endif

/* Per block slot of the MCU, eight bits: bits 0-1 component, bits 2-4 index of its DC
 * table, bits 5-7 index of its AC table (indices into the MjHuffDev array).  The
 * descriptors of all slots of an MCU (at most ten) travel as three wave-uniform words,
 * so switching block costs a few ALU operations and no memory access. */
MJ_HD uint32_t mj_slot_desc(uint32_t comp, uint32_t tdc, uint32_t tac) {
    return (comp & 3u) | ((tdc & 7u) << 2) | ((tac & 7u) << 5);
}

/* Wave-uniform description of the scan a lane is decoding. */
struct MjScanCtx {
    const uint32_t *stream;      /* the image's compacted stream, 32-bit words in file byte order; 16-byte aligned on the device */
    const unsigned char *tabs;   /* the image's MjHuffDev array (LDS on the device) */
    const MjHuffWalk *walk;      /* the matching canonical ranges (HBM on the device) */
    const unsigned char *pairs;  /* one side table per AC table (LDS on the device): MjHuffRun for the counting passes,
                                    MjHuffPair (second-symbol entries) for the writing pass */
    uint32_t pair_stride;        /* sizeof of that side table */
    uint32_t slotw0, slotw1, slotw2; /* mj_slot_desc() of slot i in byte i & 3 of word i >> 2 (separate members: never indexed) */
    int16_t *coefs;              /* block 0 of the image; 64 int16 per block, natural (row-major) order */
    const uint8_t *natpos;       /* [64] natural position of zigzag index k (LDS on the device) */
    uint32_t bpm;                /* blocks per MCU */
};

/* fill cx.slotw from per-slot component / table indices */
MJ_HD void mj_ctx_slots(MjScanCtx &cx, uint32_t bpm, const uint8_t *comp, const uint8_t *tdc, const uint8_t *tac) {
    uint32_t w0 = 0, w1 = 0, w2 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < MJ_MAX_BPM; i++) {
        const uint32_t d = (i < bpm) ? mj_slot_desc(comp[i], tdc[i], tac[i]) << (8 * (i & 3)) : 0u;
        if (i < 4) w0 |= d;
        else if (i < 8) w1 |= d;
        else w2 |= d;
    }
    cx.slotw0 = w0;
    cx.slotw1 = w1;
    cx.slotw2 = w2;
    cx.bpm = bpm;
}

MJ_HD uint32_t mj_slot_of(const MjScanCtx &cx, uint32_t c) {
    /* shifts, not a choice between the words themselves: a select between struct
     * members becomes a select between their addresses and sends the struct to memory */
    const uint64_t w01 = ((uint64_t) cx.slotw1 << 32) | cx.slotw0;
    const uint32_t a = (uint32_t) (w01 >> ((c & 7u) * 8u));
    const uint32_t b = cx.slotw2 >> ((c & 1u) * 8u);
    return ((c < 8) ? a : b) & 0xFFu;
}
MJ_HD uint32_t mj_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
static_assert(sizeof(MjHuffDev) == 3072, "table offsets below are index * 3 * 1024, by shifts");
MJ_HD uint32_t mj_slot_tdc(uint32_t slot) {   /* byte offset of the slot's DC table */
    const uint32_t k = (slot & 0x1Cu) << 8;   /* index * 1024 */
    return k + (k << 1);
}
MJ_HD uint32_t mj_slot_tac(uint32_t slot) {
    const uint32_t k = (slot & 0xE0u) << 5;
    return k + (k << 1);
}

MJ_HD uint32_t mj_be32(uint32_t v) {
    return __builtin_bswap32(v);
}

/* (hi:lo) >> sh, low 32 bits, for 1 <= sh <= 31 (v_alignbit_b32 on the device) */
MJ_HD uint32_t mj_align(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t) ((((uint64_t) hi << 32) | lo) >> sh);
#endif
}

/* `width` bits of v starting at bit `off` (width 0 gives 0; off + width <= 32) */
MJ_HD uint32_t mj_bfe(uint32_t v, uint32_t off, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    return width ? (v >> off) & (0xFFFFFFFFu >> (32u - width)) : 0u;
#endif
}

/* Four consecutive stream words (one 16-byte load per lane: scattered per-lane
 * dword loads cost one cache-line access each and saturate the vector memory path
 * long before the ALUs; 16 bytes per access cuts those accesses four-fold). */
struct MjQuad {
    uint32_t w0, w1, w2, w3;
};

MJ_HD MjQuad mj_load_quad(const MjScanCtx &cx, uint32_t block) {
    MjQuad q;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 v = reinterpret_cast<const uint4 *>(cx.stream)[block];
    q.w0 = v.x;
    q.w1 = v.y;
    q.w2 = v.z;
    q.w3 = v.w;
#else
    const uint32_t *s = cx.stream + 4u * (size_t) block;
    q.w0 = s[0];
    q.w1 = s[1];
    q.w2 = s[2];
    q.w3 = s[3];
#endif
    return q;
}

/* How often mj_checkpoint must run: at least once every this many symbols.  A symbol
 * takes at most one word from the queue, and the queue holds eight. */
#define MJ_CHECK_EVERY 4

/* Decoder state of one lane.  (p, k, c) is the state proper -- equal (p, k, c)
 * decode identically from there on; the rest is derived from it.
 *
 * Stream words reach the window through a queue of 16-byte blocks: qa and qb are what
 * mj_refill takes words from, qc follows them, nq is in flight.  Loads are issued only
 * in mj_checkpoint, which the caller runs every MJ_CHECK_EVERY symbols for the whole
 * wavefront at once, and what a checkpoint requests is not looked at before the next
 * one: the memory counter a wavefront waits on is in order, and any use of a register
 * still being loaded -- a copy the compiler inserts is enough -- stalls every lane. */
struct MjLane {
    uint32_t hi, lo;    /* the stream from bit p on, left-aligned in hi:lo, zero beyond the window */
    uint32_t p;         /* bit offset of the next symbol */
    uint32_t thr;       /* the window ends at bit thr + 32: more than 32 bits are left while p < thr */
    MjQuad qa, qb;      /* the 32 bytes the next words come from */
    uint32_t icur;      /* index (0..7) of the next word in qa:qb */
    MjQuad qc;          /* the following 16 bytes */
    MjQuad nq;          /* the 16 bytes after qc, requested at the last checkpoint */
    uint32_t blk16;     /* index of the 16-byte block held in qc */
    uint32_t k;         /* zigzag index of the next coefficient; 0: the DC symbol comes next */
    uint32_t c;         /* block slot inside the MCU */
    uint32_t slot;      /* mj_slot_of(c) */
    uint32_t tcur;      /* byte offset of the table the next symbol is decoded with */
    uint32_t pcur;      /* byte offset of the MjHuffPair of the slot's AC table */
};

MJ_HD uint32_t mj_pick4(const MjQuad &q, bool b0, bool b1) {
    const uint32_t x = b0 ? q.w1 : q.w0, y = b0 ? q.w3 : q.w2;
    return b1 ? y : x;
}

/* next stream word, byte-swapped to bit order */
MJ_HD uint32_t mj_take_word(MjLane &L) {
    const uint32_t i = L.icur;
    const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0;
    const uint32_t a = mj_pick4(L.qa, b0, b1), b = mj_pick4(L.qb, b0, b1);
    L.icur = i + 1;
    return mj_be32((i & 4u) ? b : a);
}

/* Keep the queue ahead of the window: once qa is used up, move on by 16 bytes; then
 * request the block after qc again, every lane, whether it moved or not -- a request
 * that depends on nothing can be issued without touching the previous one's data. */
MJ_HD void mj_checkpoint(const MjScanCtx &cx, MjLane &L) {
    if (L.icur >= 4) {
        L.qa = L.qb;
        L.qb = L.qc;
        L.qc = L.nq;
        L.icur -= 4;
        L.blk16++;
    }
    L.nq = mj_load_quad(cx, L.blk16 + 1);
}

MJ_HD void mj_lane_init(const MjScanCtx &cx, MjLane &L, uint32_t p, uint32_t k, uint32_t c) {
    const uint32_t wi = p >> 5, sh = p & 31u;
    L.blk16 = wi >> 2;
    L.qa = mj_load_quad(cx, L.blk16);
    L.qb = mj_load_quad(cx, L.blk16 + 1);
    L.blk16 += 2;
    L.qc = mj_load_quad(cx, L.blk16);
    L.nq = mj_load_quad(cx, L.blk16 + 1);
    L.icur = wi & 3u;
    const uint32_t w0 = mj_take_word(L), w1 = mj_take_word(L);
    L.hi = sh ? mj_align(w0, w1, 32u - sh) : w0;
    L.lo = w1 << sh;
    L.p = p;
    L.thr = (p & ~31u) + 32u;
    L.k = k;
    L.c = c;
    L.slot = mj_slot_of(cx, c);
    L.tcur = k ? mj_slot_tac(L.slot) : mj_slot_tdc(L.slot);
    L.pcur = mj_mul24(L.slot >> 5, cx.pair_stride);
}

/* Top the window up to more than 32 bits. */
MJ_HD void mj_refill(MjLane &L) {
    if (L.p >= L.thr) {
        const uint32_t w = mj_take_word(L);
        const uint32_t sh = L.p - L.thr;           /* 0..31: 32 - sh bits are left */
        L.hi |= (w >> 1) >> (31u - sh);
        L.lo = w << sh;
        L.thr += 32u;
    }
}

/* Per-component DC accumulators (decode.c:682-690; the int16 wrap is applied when a
 * value is stored).  Two homes: lane-private words in memory, acc[comp * stride], which
 * the host emulation uses; and two registers holding the four sums modulo 2^16 (MjAccPk
 * below), which the kernels use. */
struct MjAccMem {
    int32_t *p;
    uint32_t stride;
    MJ_HDM int32_t add(uint32_t comp, int32_t v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        (void) __hip_atomic_fetch_add(p + comp * stride, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        return 0;
#else
        p[comp * stride] += v;
        return p[comp * stride];
#endif
    }
};

/* The sums in registers, modulo 2^16, two to a word.  A predictor is only ever used
 * through the int16 it is stored as (decode.c:687-690), so every sum that feeds it may be
 * taken modulo 2^16, and with two independent 16-bit fields per word adding to one
 * component is a shift and a packed add instead of a four-way choice. */
struct MjAccPk {
    uint32_t a01, a23;
    MJ_HDM static MjAccPk of(int32_t d0, int32_t d1, int32_t d2, int32_t d3) {
        const MjAccPk a = {((uint32_t) d0 & 0xFFFFu) | ((uint32_t) d1 << 16), ((uint32_t) d2 & 0xFFFFu) | ((uint32_t) d3 << 16)};
        return a;
    }
    MJ_HDM static uint32_t add16x2(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b));
#else
        return ((a + b) & 0xFFFFu) | (((a >> 16) + (b >> 16)) << 16);
#endif
    }
    /* adds v to the component's sum; returns the new sum, right in its low 16 bits */
    MJ_HDM int32_t add(uint32_t comp, int32_t v) {
        const uint32_t sh = (comp & 1u) << 4;
        const bool upper = (comp & 2u) != 0;
        const int32_t sum = (int32_t) ((upper ? a23 : a01) >> sh) + v;
        const uint32_t field = ((uint32_t) v & 0xFFFFu) << sh;
        a01 = add16x2(a01, upper ? 0u : field);
        a23 = add16x2(a23, upper ? field : 0u);
        return sum;
    }
    MJ_HDM int32_t sum(uint32_t comp) const {        /* modulo 2^16, as a non-negative number */
        return (int32_t) ((((comp & 2u) ? a23 : a01) >> ((comp & 1u) << 4)) & 0xFFFFu);
    }
};

/* What one symbol did. */
struct MjSym {
    int32_t val;      /* WRITE only: the coefficient value (for DC: the predictor after the addition) */
    uint32_t kpos;    /* zigzag position the value belongs to (0 for DC) */
    bool store;       /* WRITE only: val is a coefficient to store (DC always; AC when size != 0 and in range) */
    int32_t val2;     /* WRITE only: a second AC symbol taken in the same step */
    uint32_t kpos2;
    bool store2;
    bool done;        /* the block ended with this step */
};

/* A first-level entry that is not a code: second-level table or canonical walk.  tcur is the
 * byte offset of the table in use, hi the stream from the symbol's first bit on. */
MJ_HD uint32_t mj_long_code_at(const MjScanCtx &cx, uint32_t tcur, uint32_t hi, int is_ac, uint32_t e) {
    const MjHuffDev *H = reinterpret_cast<const MjHuffDev *>(cx.tabs + tcur);
    const uint32_t w = hi >> 16;
    if (e & MJ_LUT_LONG) {
        /* the MJ_SUB_BITS bits after the first MJ_LUT_BITS index the second level */
        return H->sub[e & (MJ_SUB_MAX - 1)][w & (MJ_SUB_SIZE - 1)];
    }
    /* canonical ranges per length (the reference's own walk, decode.c:674-681) */
    const MjHuffWalk *W = cx.walk + tcur / (uint32_t) sizeof(MjHuffDev);
    for (uint32_t l = MJ_LUT_BITS + 1; l <= 16; l++) {
        const int32_t code = (int32_t) (w >> (16 - l));
        if (code <= W->maxcode[l]) {
            return mj_lut_entry(l, W->vals[(uint32_t) (W->valoff[l] + code) & 255u], is_ac);
        }
    }
    return mj_lut_entry(16, 0, is_ac);   /* no such code: 16 bits are consumed as an end-of-block */
}
MJ_HD uint32_t mj_long_code(const MjScanCtx &cx, const MjLane &L, uint32_t e) {
    return mj_long_code_at(cx, L.tcur, L.hi, L.k != 0, e);
}

/*
 * Decode one symbol and advance the lane: what decode_data_unit (decode.c:665-723)
 * does per symbol.  The caller has run mj_refill since the previous step.  A DC
 * difference is added to its component's accumulator acc[comp * stride]; nb counts
 * the blocks that end.  WRITE: also produce the coefficient to store (the writing
 * pass needs it, the synchronisation passes only need bit counts and DC sums).
 */
template <bool WRITE, class ACC>
MJ_HD MjSym mj_step(const MjScanCtx &cx, MjLane &L, ACC &acc, uint32_t &nb, uint32_t end) {
    MjSym s;
    const uint32_t idx2 = (L.hi >> (32 - MJ_LUT_BITS)) << 1;
    uint32_t e = *reinterpret_cast<const uint16_t *>(cx.tabs + L.tcur + idx2);
    const uint32_t d = *reinterpret_cast<const uint16_t *>(cx.pairs + L.pcur + (WRITE ? idx2 : ((L.hi >> (32 - MJ_RUN_BITS)) << 1)));
    if (MJ_E_LEN(e) == 0) {
        e = mj_long_code(cx, L, e);
    }
    const uint32_t size = mj_bfe(e, 7, 4);
    uint32_t used = MJ_E_LEN(e) + size;            /* 1..31 */
    uint32_t step = e & 127u;                      /* run + 1; 64 on end of block; ZRL nets 16 */
    const bool dc = (L.k == 0);
    s.val = 0;
    s.val2 = 0;
    s.kpos2 = 0;
    s.store2 = false;
    if (!WRITE) {
        /* a run of whole AC symbols in one go, when the block lasts up to the run's last
         * symbol and the run ends inside the span: exactly what that many single steps
         * would do, including where the span stops */
        const uint32_t run_used = d >> 11, run_before = mj_bfe(d, 7, 4);
        const bool run = (d != 0) && !dc && (L.k + run_before < 64u) && (L.p + run_used <= end);
        used = run ? run_used : used;
        step = run ? run_before + (d & 127u) : step;
    }
    if (WRITE || dc) {
        /* the `size` bits after the code, then EXTEND (decode.c:684-686, :716-718):
         * a leading 1 means the value as read, a leading 0 means value - 2^size + 1 */
        const uint32_t v = mj_bfe(L.hi, 32u - used, size);
        const uint32_t full = 1u << size;
        s.val = (v < (full >> 1)) ? (int32_t) (v + 1u - full) : (int32_t) v;
        if (dc) {
            s.val = acc.add(L.slot & 3u, s.val);
            L.tcur = mj_slot_tac(L.slot);
        }
    }
    const uint32_t k1 = L.k + step;
    s.kpos = dc ? 0u : k1 - 1u;
    s.store = dc || (size != 0 && k1 <= 64u);
    L.k = k1;
    if (WRITE) {
        /* the next AC symbol as well, when its code lay inside the lookup index and the
         * block goes on: two single steps in one */
        if ((d != 0) && !dc && (k1 < 64u)) {
            const uint32_t size2 = mj_bfe(d, 7, 4);
            used += MJ_E_LEN(d) + size2;           /* at most 10 + 15 */
            const uint32_t v = mj_bfe(L.hi, 32u - used, size2);
            const uint32_t full = 1u << size2;
            s.val2 = (v < (full >> 1)) ? (int32_t) (v + 1u - full) : (int32_t) v;
            const uint32_t k2 = k1 + (d & 127u);
            s.kpos2 = k2 - 1u;
            s.store2 = (size2 != 0 && k2 <= 64u);
            L.k = k2;
        }
    }
    L.hi = mj_align(L.hi, L.lo, 32u - used);
    L.lo <<= used;
    L.p += used;
    s.done = (L.k >= 64u);
    if (s.done) {
        L.k = 0;
        L.c = (L.c + 1 == cx.bpm) ? 0 : L.c + 1;
        L.slot = mj_slot_of(cx, L.c);
        L.tcur = mj_slot_tdc(L.slot);
        L.pcur = mj_mul24(L.slot >> 5, cx.pair_stride);
        nb++;
    }
    return s;
}

/*
 * Decode symbols with an initialised lane until its position reaches `end`, counting
 * only: nb is advanced by the number of blocks completed and every DC difference is
 * added to its component's accumulator (MjAccMem or MjAccPk).  The lane can go on to
 * the next span as it is.
 */
template <class ACC>
MJ_HD void mj_run_span(const MjScanCtx &cx, MjLane &L, uint32_t end, uint32_t &nb, ACC &acc) {
    while (L.p < end) {
        mj_checkpoint(cx, L);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < MJ_CHECK_EVERY; i++) {
            if (L.p < end) {
                mj_refill(L);
                (void) mj_step<false>(cx, L, acc, nb, end);
            }
        }
    }
}

/* The same from a bare state (p, k, c), which is updated. */
MJ_HD void mj_decode_span(const MjScanCtx &cx, uint32_t &p, uint32_t &k, uint32_t &c, uint32_t end, uint32_t &nb,
                          int32_t *acc, uint32_t stride) {
    if (p >= end) {
        return;
    }
    MjLane L;
    mj_lane_init(cx, L, p, k, c);
    MjAccMem am = {acc, stride};
    mj_run_span(cx, L, end, nb, am);
    p = L.p;
    k = L.k;
    c = L.c;
}

/*
 * One step of the writing pass of one lane.  The lane assembles the block it is in
 * inside blkbuf (64 int16 in natural order -- the de-zigzag of quant_table.c:131-152
 * happens here, at the store -- zero-filled by the caller) and reports `fin` when the
 * block is complete and should be copied to block index fin_blk of the image.
 *
 * Ownership: a block belongs to the lane that decodes its DC symbol.  A lane whose
 * entry state is in the middle of a block (k != 0) runs in `skip` mode until that
 * block ends -- the previous lane finishes it by decoding past its own subsequence
 * end (see mj_write_live) -- so every block is assembled by exactly one lane and
 * leaves the chip as one full 128-byte line.  acc holds the DC predictors; cx.pairs
 * must point at the second-symbol tables (MjHuffSet.next).  The caller
 * runs mj_checkpoint at least every MJ_CHECK_EVERY steps.
 */
MJ_HD void mj_write_step(const MjScanCtx &cx, MjLane &L, bool &skip, uint32_t &blk, uint32_t blk_end, MjAccPk &acc,
                         int16_t *blkbuf, bool &fin, uint32_t &fin_blk) {
    mj_refill(L);
    uint32_t ended = 0;
    const MjSym s = mj_step<true>(cx, L, acc, ended, 0);
    const bool mine = !skip && blk < blk_end;
    if (s.store && mine) {
        blkbuf[cx.natpos[s.kpos]] = (int16_t) s.val;
    }
    if (s.store2 && mine) {
        blkbuf[cx.natpos[s.kpos2]] = (int16_t) s.val2;
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

/* ---- sixteen bytes at a time -------------------------------------------------- */

/* Classes of sixteen consecutive bytes as byte masks: bit 7 of byte j of word i is set
 * when byte 4 i + j has the class; a byte flagged in none of the three is MJ_B_DROP. */
struct MjClass16 {
    uint32_t keep[4], rst[4], term[4];
};

/* bit 7 of every byte of v that is zero (exact per byte: no carries between bytes) */
MJ_HD uint32_t mj_zero_bytes(uint32_t v) {
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
}

/* the bytes of word i (of four) that are among the first n of the sixteen, as a mask */
MJ_HD uint32_t mj_bytes_below(uint32_t n, int i) {
    const uint32_t m = n > 4u * i ? n - 4u * i : 0u;
    return m >= 4 ? 0xFFFFFFFFu : (m ? (0xFFFFFFFFu >> (32u - 8u * m)) : 0u);
}

/* mj_classify for the sixteen bytes w[0..3] (little-endian words) given the byte before
 * and the byte after them */
MJ_HD MjClass16 mj_classify16(const uint32_t (&w)[4], uint32_t prev, uint32_t next) {
    uint32_t ff[4], z0[4], rs[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 4; i++) {
        ff[i] = mj_zero_bytes(~w[i]);
        z0[i] = mj_zero_bytes(w[i]);
        rs[i] = mj_zero_bytes((w[i] & 0xF8F8F8F8u) ^ 0xD0D0D0D0u);
    }
    MjClass16 c;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 4; i++) {
        const uint32_t prev_ff = (ff[i] << 8) | (i ? ff[i - 1] >> 24 : (prev == 0xFFu ? 0x80u : 0u));
        const uint32_t next_00 = (z0[i] >> 8) | (i < 3 ? z0[i + 1] << 24 : (next == 0x00u ? 0x80000000u : 0u));
        const uint32_t next_rs = (rs[i] >> 8) | (i < 3 ? rs[i + 1] << 24 : ((next & 0xF8u) == 0xD0u ? 0x80000000u : 0u));
        const uint32_t next_ff = (ff[i] >> 8) | (i < 3 ? ff[i + 1] << 24 : (next == 0xFFu ? 0x80000000u : 0u));
        const uint32_t drop1 = prev_ff & (z0[i] | rs[i]);
        c.keep[i] = ~drop1 & (~ff[i] | next_00) & 0x80808080u;
        c.rst[i] = ff[i] & next_rs;
        c.term[i] = ff[i] & ~(next_00 | next_rs | next_ff);
    }
    return c;
}

/* The classes of bytes pos .. pos + 15 of a segment of len bytes (pos a multiple of 16,
 * pos < len; twenty bytes are readable from the last four-byte boundary at or before
 * seg + pos).  The segment may start at any byte address.  Bytes at or past the end are
 * MJ_B_DROP and count as zero for their neighbours.  w gets the bytes. */
MJ_HD MjClass16 mj_classify16_at(const uint8_t *seg, uint32_t len, uint32_t pos, uint32_t (&w)[4]) {
    const uint8_t *at = seg + pos;
    const uint32_t mis = (uint32_t) ((uintptr_t) at & 3u);
#if defined(__HIP_DEVICE_COMPILE__)
    /* four-byte aligned loads (all the memory instructions ask for), then a byte shift */
    struct __attribute__((aligned(4))) Quad4 {
        uint32_t x, y, z, w;
    };
    const Quad4 v = *reinterpret_cast<const Quad4 *>(at - mis);
    if (mis == 0) {
        w[0] = v.x;
        w[1] = v.y;
        w[2] = v.z;
        w[3] = v.w;
    } else {
        const uint32_t t = *reinterpret_cast<const uint32_t *>(at - mis + 16);
        w[0] = __builtin_amdgcn_alignbyte(v.y, v.x, mis);
        w[1] = __builtin_amdgcn_alignbyte(v.z, v.y, mis);
        w[2] = __builtin_amdgcn_alignbyte(v.w, v.z, mis);
        w[3] = __builtin_amdgcn_alignbyte(t, v.w, mis);
    }
#else
    (void) mis;
    memcpy(w, at, 16);
#endif
    const uint32_t prev = pos ? seg[pos - 1] : 0u;
    const uint32_t next = (pos + 16 < len) ? seg[pos + 16] : 0u;
    const uint32_t nvalid = len - pos;     /* >= 1 */
    uint32_t valid[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (nvalid < 16) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 4; i++) {
            valid[i] = mj_bytes_below(nvalid, i);
            w[i] &= valid[i];
        }
    }
    MjClass16 c = mj_classify16(w, prev, next);
    if (nvalid < 16) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 4; i++) {
            c.keep[i] &= valid[i];
            c.rst[i] &= valid[i];
            c.term[i] &= valid[i];
        }
    }
    return c;
}

#endif /* MJ_ENTROPY_CORE_H */
