/*
 * mj_entropy_dev.h -- device code the entropy kernel files share (mj_k_entropy.hip: the
 * synchronisation rounds, the cross-sequence kernels, the writing pass; mj_k_spec.hip: the
 * speculative synchronisation): the geometry of an interval, the carve-up of the tables in
 * LDS, the counting decoders -- a lane's (run_nowrite) and a whole wavefront's for one
 * stream (solo_nowrite) -- and the record they leave per subsequence.
 */
#ifndef MJ_ENTROPY_DEV_H
#define MJ_ENTROPY_DEV_H

#include "mj_entropy_core.h"
#include "mj_internal.h"

namespace {

struct Geo {          /* wave-uniform geometry of one interval */
    uint32_t b0, b1;  /* bit range in the compacted stream */
    uint32_t nsub;
    uint32_t nseq;
};

__device__ __forceinline__ Geo interval_geo(const uint32_t *ivl, uint32_t iv, uint32_t S, uint32_t SQ) {
    Geo g;
    const uint32_t a = ivl[iv], e = ivl[iv + 1];
    g.b0 = a * 8u;
    g.b1 = (e > a ? e : a) * 8u;
    g.nsub = (g.b1 - g.b0 + S - 1) / S;
    if (g.nsub == 0) g.nsub = 1;
    g.nseq = (g.nsub + SQ - 1) / SQ;
    return g;
}

__device__ __forceinline__ uint32_t sub_end(uint32_t b0, uint32_t b1, uint32_t nsub, uint32_t j, uint32_t S) {
    return (j + 1 >= nsub) ? b1 : b0 + (j + 1) * S;
}

extern __shared__ __align__(16) unsigned char s_dyn[];

/* LDS carve-up shared by all entropy kernels:
 * [Huffman tables][16 slot words][DC accumulators: waves x 4 components x 64 lanes]... */
__device__ __forceinline__ int32_t *lds_dc(const MjDeviceBatch &b) {
    return reinterpret_cast<int32_t *>(s_dyn + b.max_ntab * sizeof(MjHuffDev) + 16 * sizeof(uint32_t));
}
constexpr uint32_t DC_WAVE_BYTES = 64u * MJ_MAX_COMP * sizeof(int32_t);
__device__ __forceinline__ unsigned char *lds_after_dc(const MjDeviceBatch &b, uint32_t waves) {
    return reinterpret_cast<unsigned char *>(lds_dc(b)) + waves * DC_WAVE_BYTES;
}

/* cooperative copy of the image's Huffman tables into LDS, and of one side table per AC
 * table into `pairs`: the symbol runs for the counting kernels, the second-symbol
 * entries for the writing kernel (the caller synchronises) */
__device__ __forceinline__ void load_tables(const MjDeviceBatch &b, const MjImageDev &im, unsigned char *pairs, bool runs) {
    const MjHuffSet &set = b.sets[im.huff_set];
    const uint32_t n = set.ntab * (uint32_t) (sizeof(MjHuffDev) / 16);
    const uint4 *src = reinterpret_cast<const uint4 *>(set.tab);
    uint4 *dst = reinterpret_cast<uint4 *>(s_dyn);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    const uint32_t np = set.nac * (uint32_t) ((runs ? sizeof(MjHuffRun) : sizeof(MjHuffPair)) / 16);
    const uint4 *psrc = runs ? reinterpret_cast<const uint4 *>(set.run) : reinterpret_cast<const uint4 *>(set.next);
    uint4 *pdst = reinterpret_cast<uint4 *>(pairs);
    for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) pdst[i] = psrc[i];
}

__device__ __forceinline__ MjScanCtx make_ctx(const MjDeviceBatch &b, const MjImageDev &im, const unsigned char *pairs, bool runs) {
    MjScanCtx cx;
    cx.stream = reinterpret_cast<const uint32_t *>(b.cmp + im.cmp_off);
    cx.tabs = s_dyn;
    cx.walk = b.sets[im.huff_set].walk;
    cx.pairs = pairs;
    cx.pair_stride = runs ? (uint32_t) sizeof(MjHuffRun) : (uint32_t) sizeof(MjHuffPair);
    mj_ctx_slots(cx, im.bpm, im.blk_comp, im.blk_dc, im.blk_ac);
    cx.coefs = b.coefs + im.coef_off * 64;
    cx.natpos = nullptr;    /* the kernels that store coefficients point it at their LDS copy */
    return cx;
}

struct LaneRec {      /* what a lane knows about one subsequence */
    uint32_t p, kc, nb;
    int32_t d0, d1, d2, d3;
};

/* Decode with an initialised lane up to `end`, counting blocks and DC differences */
__device__ __forceinline__ LaneRec run_nowrite(const MjScanCtx &cx, MjLane &L, uint32_t end) {
    uint32_t nb = 0;
    MjAccPk acc = {0, 0};
    mj_run_span(cx, L, end, nb, acc);
    LaneRec r;
    r.p = L.p;
    r.kc = (L.c << 8) | L.k;
    r.nb = nb;
    r.d0 = acc.sum(0);       /* modulo 2^16: all a predictor needs */
    r.d1 = acc.sum(1);
    r.d2 = acc.sum(2);
    r.d3 = acc.sum(3);
    return r;
}

/* the same from a bare state */
__device__ __forceinline__ LaneRec decode_nowrite(const MjScanCtx &cx, uint32_t p, uint32_t kc, uint32_t end) {
    if (p >= end) {
        const LaneRec r = {p, kc, 0, 0, 0, 0, 0};
        return r;
    }
    MjLane L;
    mj_lane_init(cx, L, p, kc & 0xFFu, kc >> 8);
    return run_nowrite(cx, L, end);
}

/* One decoder, the whole wavefront: what decode_nowrite returns, for a wave-uniform state.
 * A wavefront issues one instruction at a time however few of its lanes take part, and a
 * step of the lane-parallel decoder is a hundred instructions.  Here lane l looks up the
 * code that would start at bit l of a 64-bit window, all lanes at once, and packs where the
 * symbol after it starts and how far it moves the zigzag index into one word; the decoder
 * proper is a walk over those words in scalar registers, seven instructions per AC symbol.
 * Every lane's 32 bits come from three stream words, so a symbol may start at any of the 64
 * bits; the walk leaves the window when the next symbol starts past it (or, for a DC symbol,
 * so late that the lane its value bits start at does not exist), and the lanes past the
 * span's end answer "stop".  The words are made again when the window moves on and
 * when the block slot brings other tables.  Symbols are taken one at a time: the symbol runs
 * of the lane-parallel decoder are defined as what that many single steps do. */
__device__ __forceinline__ uint32_t uni(uint32_t v) {
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ uint32_t lane_of(uint32_t v, uint32_t l) {      /* v of lane l (wave-uniform l) */
    return (uint32_t) __builtin_amdgcn_readlane((int) v, (int) l);
}

#ifndef MJ_SOLO_SPAN
#define MJ_SOLO_SPAN 64
#endif
constexpr uint32_t SOLO_SPAN = MJ_SOLO_SPAN;   /* symbols that start in this many bits of a window are decoded from it (32 .. 64) */
constexpr uint32_t SOLO_STOP = 127u;   /* zigzag step of a word that is not a symbol: past the span, or a long code */

/* lane l's word for lookup entry e: bits 0-6 where the next symbol starts, 8-14 the zigzag
 * step, 16-20 the code length, 21-24 the number of value bits */
__device__ __forceinline__ uint32_t solo_word(uint32_t e, uint32_t l, uint32_t lim) {
    const uint32_t len = MJ_E_LEN(e), size = (e >> 7) & 15u;
    const uint32_t w = (l + len + size) | ((e & 127u) << 8) | (len << 16) | (size << 21);
    return (len == 0 || l >= lim) ? (l | (SOLO_STOP << 8)) : w;
}

__device__ __forceinline__ LaneRec solo_nowrite(const MjScanCtx &cx, uint32_t p_in, uint32_t kc_in, uint32_t end_in, uint32_t lane) {
    uint32_t p = uni(p_in);
    const uint32_t kc = uni(kc_in), end = uni(end_in);
    if (p >= end) {
        const LaneRec r = {p, kc, 0, 0, 0, 0, 0};
        return r;
    }
    /* the scan's description in scalar registers, so that the walk below stays in them */
    MjScanCtx sx = cx;
    sx.slotw0 = uni(cx.slotw0);
    sx.slotw1 = uni(cx.slotw1);
    sx.slotw2 = uni(cx.slotw2);
    const uint32_t bpm = uni(cx.bpm);
    uint32_t k = kc & 0xFFu, c = kc >> 8;
    uint32_t slot = mj_slot_of(sx, c);
    uint32_t nb = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    /* the span's stream words in one register, a word per lane: one memory access per 64 words */
    const uint32_t wlast = ((end - 1u) >> 5) + 2u;             /* the third word of the last window */
    uint32_t wi0 = p >> 5;
    uint32_t V = mj_be32((wi0 + lane <= wlast) ? cx.stream[wi0 + lane] : 0u);
    const uint32_t sh = lane & 31u;
    const bool up = lane >= 32u;
    for (;;) {
        /* the window starts at the word p is in; W: the 32 stream bits from bit `lane` of the window on */
        uint32_t j = (p >> 5) - wi0;
        if (j + 2u > 63u) {
            wi0 = p >> 5;
            V = mj_be32((wi0 + lane <= wlast) ? cx.stream[wi0 + lane] : 0u);
            j = 0;
        }
        const uint32_t w0 = lane_of(V, j), w1 = lane_of(V, j + 1), w2 = lane_of(V, j + 2);
        const uint32_t a = up ? w1 : w0, bb = up ? w2 : w1;
        const uint32_t W = (uint32_t) ((((((uint64_t) a) << 32) | bb) << sh) >> 32);
        const uint32_t idx2 = (W >> (32 - MJ_LUT_BITS)) << 1;
        const uint32_t base = p & ~31u;
        const uint32_t lim = min(SOLO_SPAN, end - base);       /* a symbol is decoded when it starts before this */
        uint32_t o = p - base;                                 /* 0 .. 31, < lim */
        uint32_t Edc = 0, Eac = 0, Ndc = 0, Nac = 0, have = 0xFFFFFFFFu;
        for (;;) {
            if (have != (slot >> 2)) {                         /* the slot's two tables */
                Edc = *reinterpret_cast<const uint16_t *>(cx.tabs + mj_slot_tdc(slot) + idx2);
                Eac = *reinterpret_cast<const uint16_t *>(cx.tabs + mj_slot_tac(slot) + idx2);
                Ndc = solo_word(Edc, lane, lim);
                Nac = solo_word(Eac, lane, lim);
                have = slot >> 2;
            }
            if (k == 0) {
                /* the DC symbol: the difference is `size` bits after the code, then EXTEND (decode.c:684-686);
                 * its value bits are taken from the lane the code ends at, which must exist */
                if (o + 16u > 63u) break;
                uint32_t x = lane_of(Ndc, o);
                if (((x >> 8) & 127u) == SOLO_STOP) {          /* o < lim here: a long code */
                    const uint32_t e = uni(mj_long_code_at(cx, mj_slot_tdc(slot), lane_of(W, o), 0, lane_of(Edc, o)));
                    x = solo_word(e, o, lim);
                }
                const uint32_t len = (x >> 16) & 31u, size = (x >> 21) & 15u;
                const uint32_t v = size ? lane_of(W, o + len) >> (32u - size) : 0u;
                const uint32_t full = 1u << size;
                const uint32_t val = (v < (full >> 1)) ? v + 1u - full : v;
                const uint32_t comp = slot & 3u;
                d0 += (comp == 0) ? val : 0u;
                d1 += (comp == 1) ? val : 0u;
                d2 += (comp == 2) ? val : 0u;
                d3 += (comp == 3) ? val : 0u;
                o = x & 127u;
                k = 1;
            }
            /* AC symbols up to the end of the block, of the span, or a long code */
            uint32_t x = 0;
            while ((k | o) < 64u) {                            /* the block goes on and the symbol starts inside the window */
                x = lane_of(Nac, o);
                k += (x >> 8) & 127u;
                o = x & 127u;
            }
            if (((x >> 8) & 127u) == SOLO_STOP) {
                k -= SOLO_STOP;                                /* not a symbol: o did not move */
                if (o >= lim) break;
                const uint32_t e = uni(mj_long_code_at(cx, mj_slot_tac(slot), lane_of(W, o), 1, lane_of(Eac, o)));
                o += MJ_E_LEN(e) + ((e >> 7) & 15u);
                k += e & 127u;
            }
            if (k >= 64u) {
                k = 0;
                c = (c + 1 == bpm) ? 0 : c + 1;
                slot = mj_slot_of(sx, c);
                nb++;
            }
            if (o >= lim) break;
        }
        p = base + o;
        if (p >= end) break;
    }
    LaneRec r;
    r.p = p;
    r.kc = (c << 8) | k;
    r.nb = nb;
    r.d0 = (int32_t) (d0 & 0xFFFFu);       /* modulo 2^16, like MjAccPk::sum */
    r.d1 = (int32_t) (d1 & 0xFFFFu);
    r.d2 = (int32_t) (d2 & 0xFFFFu);
    r.d3 = (int32_t) (d3 & 0xFFFFu);
    return r;
}

__device__ __forceinline__ void store_rec(MjSubRec *dst, const LaneRec &r) {
    MjSubRec o;
    o.exit.p = r.p;
    o.exit.kc = r.kc;
    o.nb = r.nb;
    o.dcsum[0] = r.d0;
    o.dcsum[1] = r.d1;
    o.dcsum[2] = r.d2;
    o.dcsum[3] = r.d3;
    o.pad_ = 0;
    *dst = o;
}

/* exclusive prefix over the lanes of a wave for the five counters of a record */
__device__ __forceinline__ void wave_excl_scan5(uint32_t &nb, int32_t &d0, int32_t &d1, int32_t &d2, int32_t &d3) {
    const int lane = threadIdx.x & 63;
    uint32_t inb = nb;
    int32_t i0 = d0, i1 = d1, i2 = d2, i3 = d3;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t a = __shfl_up(inb, d, 64);
        int32_t x0 = __shfl_up(i0, d, 64), x1 = __shfl_up(i1, d, 64), x2 = __shfl_up(i2, d, 64), x3 = __shfl_up(i3, d, 64);
        if (lane >= d) {
            inb += a;
            i0 += x0;
            i1 += x1;
            i2 += x2;
            i3 += x3;
        }
    }
    nb = inb - nb;
    d0 = i0 - d0;
    d1 = i1 - d1;
    d2 = i2 - d2;
    d3 = i3 - d3;
}

}  // namespace

/* dynamic LDS of the kernels that keep an image's tables there, before what each adds */
static inline size_t table_lds(const MjDeviceBatch &b, uint32_t waves) {
    return (size_t) b.max_ntab * sizeof(MjHuffDev) + 16 * sizeof(uint32_t) + (size_t) waves * DC_WAVE_BYTES;
}

#endif
