/*
 * mj_k_entropy.hip -- Huffman MCU decode on the GPU.
 *
 * Replaces the reference's scan loop and block decoder (decode_scan,
 * decode.c:535-663; decode_data_unit, decode.c:665-723), which walk the stream one
 * bit at a time on one thread.  A JPEG scan is one long serial code stream, so the
 * parallel units are:
 *   - restart intervals (restart_interval.c, decode.c:580-584): independent by
 *     construction, the predictors restart at zero;
 *   - inside an interval, fixed-size *subsequences* of sub_bits bits.  A decoder
 *     that starts at an arbitrary bit with a guessed state produces garbage at
 *     first but, because Huffman codes are self-synchronising, soon falls into
 *     step with the true decode.  Every lane decodes its own subsequence from a
 *     cold state, then keeps decoding into the following subsequences until the
 *     state it reaches at a subsequence boundary equals the state recorded there
 *     by the lane before it; the recorded per-subsequence results (exit state,
 *     blocks completed, DC-difference sums) are then exact, and an exclusive
 *     prefix sum over them gives every lane the block index and DC predictors it
 *     starts with for the final, writing pass.
 *
 * A lane covers lane_subs consecutive subsequences and leaves one record at the end
 * of each, so a decoder that continues past its own lane can stop at the first record
 * it agrees with.  One wavefront owns one *sequence* of up to 64 lanes of one
 * interval; the wavefronts of a workgroup share the image's Huffman lookup tables in
 * LDS.
 *   mj_k_entropy_sync   cold decode of every lane, then rounds: every decoder still
 *                       out of step decodes one more subsequence.  The decoders left
 *                       in a round are few and scattered, so they are queued in LDS
 *                       and packed into as few wavefronts as hold them; one 32-byte
 *                       record per subsequence in HBM;
 *   mj_k_entropy_fix    only for intervals longer than one sequence (scans without
 *                       restart markers): cross-sequence synchronisation and the
 *                       prefix over sequences;
 *   mj_k_entropy_write  prefix sums over the records, then the writing pass: every
 *                       lane assembles its current block in LDS and the wavefront
 *                       copies finished blocks out as whole 128-byte lines, so the
 *                       coefficient buffer is written exactly once, with no
 *                       read-modify-write and no zero-fill pass.
 *
 * The kernels are latency / issue bound (dependent LDS lookups per symbol); the
 * figure of merit is decoded bits per second, not an HBM roofline.
 */
#include <cstdlib>

#include "mj_entropy_core.h"
#include "mj_internal.h"
#include "mj_spec_core.h"

namespace {

constexpr int WAVES = 4;             /* writing, serial kernels */
constexpr int TPB = WAVES * 64;
/* The synchronisation kernel comes in three shapes.  SPW = 1: a sequence per wavefront, eight
 * wavefronts per workgroup, a lane takes the subsequences a lane of the writing pass takes.
 * SPW = 2: a sequence per half wavefront, four wavefronts per workgroup, a lane takes twice
 * as many subsequences in the cold pass; half as many decoders have to go on into the
 * rounds, and five workgroups fit a CU (by their LDS) instead of four (by their
 * wavefronts), so that all the workgroups of a large batch are on the chip from the start
 * instead of the last of them starting when the first have finished.  SPW = 4: a sequence per
 * quarter wavefront and two groups of sequences to a workgroup (still four wavefronts), a
 * lane takes four times the subsequences in the cold pass and a quarter of the decoders go
 * on into the rounds: less work, longer chains, for batches so large that the workgroups
 * take several turns on the chip and the chains of one turn hide behind the cold passes of
 * the next (C3: 2.40 -> 2.04 ms at 256 images, 1.43 -> 1.30 at 128; 0.88 -> 1.18 at 64,
 * where it is not used).  A batch too small to fill the chip is bound by the time a lane
 * takes and keeps the short lanes.  The records are per subsequence, so the passes need not
 * agree on the lanes. */
#ifndef MJ_SYNC4_MIN_GROUPS_PER_CU
#define MJ_SYNC4_MIN_GROUPS_PER_CU 8
#endif
#ifndef MJ_SOLO_MAX
#define MJ_SOLO_MAX 8
#endif
constexpr uint32_t SOLO_MAX = MJ_SOLO_MAX;    /* that many decoders or fewer: one wavefront each at a time */

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

/* zigzag index -> natural position, for the kernels that store coefficients */
__constant__ uint8_t c_natpos[64] = MJ_NATPOS_INIT;

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

/* ------------------------------------------------------------------------ */
/* kernel 1: cold decode + synchronisation rounds over a workgroup's        */
/* sequences                                                                */
/* ------------------------------------------------------------------------ */
struct SeqGeo {           /* one sequence of the workgroup, in LDS */
    uint32_t b0, b1;      /* bit range of its interval */
    uint32_t nsub;        /* subsequences in the interval */
    uint32_t first;       /* the sequence's first subsequence within the interval */
    uint32_t n;           /* subsequences in the sequence (0: no such sequence) */
    uint32_t rec_base;    /* index of its first record */
    uint32_t pad_[2];
};

/* A decoder that has to go on, in one word: bits 0-5 zigzag index, 6-9 block slot, 10-14
 * how many bits past the start of its next subsequence it stands (a symbol that crosses a
 * boundary ends at most 30 bits after it), 15-18 sequence within the workgroup, 19-28 the
 * subsequence (within the sequence) to decode next. */
__device__ __forceinline__ uint32_t pack_task(uint32_t kc, uint32_t over, uint32_t seq, uint32_t u) {
    return (kc & 63u) | ((kc >> 8) << 6) | (over << 10) | (seq << 15) | (u << 19);
}
__device__ __forceinline__ uint32_t task_seq(uint32_t t) {
    return (t >> 15) & 15u;
}
__device__ __forceinline__ uint32_t task_sub(uint32_t t) {
    return t >> 19;
}

__device__ __forceinline__ void push_task(bool have, uint32_t task, uint32_t *queue, uint32_t *count) {
    const unsigned long long mask = __ballot(have);
    if (mask == 0) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(count, (uint32_t) __popcll(mask));
    base = __shfl(base, 0, 64);
    if (have) queue[base + __popcll(mask & ((1ull << lane) - 1ull))] = task;
}

/* The sequence after sequence sw of the workgroup, if it continues the same interval (a decoder
 * that is still out of step at the end of its sequence goes on there: the boundaries between
 * the sequences of a workgroup are then settled by its rounds, and the cross-sequence kernels
 * are left with the boundaries between workgroups); sw itself if there is none. */
template <int SEQS>
__device__ __forceinline__ uint32_t next_seq_here(const SeqGeo *s_geo, uint32_t sw, const SeqGeo &g, uint32_t SQ) {
    /* only in intervals of four sequences or more (scans without restart markers): the one
     * boundary of an interval that spills a little into a second sequence is cheaper left to
     * the cross-sequence kernel, which has the whole chip for it, than made part of the chains
     * the workgroup waits for (4K files with a restart interval per MCU row: 1.5 % of this kernel) */
    if (sw + 1 >= (uint32_t) SEQS || g.nsub <= 3u * SQ) return sw;
    const SeqGeo h = s_geo[sw + 1];
    return (h.n != 0 && h.b0 == g.b0 && h.b1 == g.b1 && h.first == g.first + g.n) ? sw + 1 : sw;
}

constexpr int sync_groups(int spw) {       /* groups of MJ_WG_SEQS sequences one workgroup takes */
    return spw == 4 ? 2 : 1;
}
constexpr int sync_threads(int spw) {
    return sync_groups(spw) * MJ_WG_SEQS * (MJ_SEQ_LANES / spw);
}
template <int SYNC_SPW>
__global__ __launch_bounds__(sync_threads(SYNC_SPW), SYNC_SPW >= 2 ? 5 : 4) void mj_k_entropy_sync(MjDeviceBatch b) {
    constexpr int SYNC_SEQ_LANES = MJ_SEQ_LANES / SYNC_SPW;
    constexpr int SYNC_SEQS = sync_groups(SYNC_SPW) * MJ_WG_SEQS;
    constexpr int SYNC_TPB = sync_threads(SYNC_SPW);
    constexpr uint32_t SYNC_WAVES = SYNC_TPB / 64;
    /* the block's entry in the list of the groups that exist; with two groups to a block, the
     * second is the next of the same image (if it has one) */
    if (blockIdx.x >= b.multi_count[SYNC_SPW == 4 ? 3 : 2]) return;
    const uint2 ge = (SYNC_SPW == 4) ? b.pair_list[blockIdx.x] : b.grp_list[blockIdx.x];
    const uint32_t ii = ge.x;
    const MjImageDev &im = b.img[ii];
    const uint32_t n_seq = b.dyn[ii].n_seq;
    const uint32_t seq0 = ge.y * MJ_WG_SEQS;
    if (seq0 >= n_seq) return;

    /* dynamic LDS: [tables][sequence geometry][3 queue counters][two queues][symbol-run tables] */
    SeqGeo *s_geo = reinterpret_cast<SeqGeo *>(lds_after_dc(b, 0));
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_geo + SYNC_SEQS);
    uint32_t *s_queue = s_cnt + 4;
    unsigned char *s_pairs = reinterpret_cast<unsigned char *>(s_queue + 2 * SYNC_TPB);

    const uint32_t S = im.sub_bits, G = im.lane_subs, SQ = MJ_SEQ_LANES * G;
    load_tables(b, im, s_pairs, true);
    if (threadIdx.x < SYNC_SEQS) {
        SeqGeo sg = {0, 0, 0, 0, 0, 0, {0, 0}};
        const uint32_t seq = seq0 + threadIdx.x;
        if (seq < n_seq) {
            const uint32_t iv = b.seq_ivl[im.seq_off + seq];
            const Geo g = interval_geo(b.ivl_start + im.ivl_off, iv, S, SQ);
            sg.b0 = g.b0;
            sg.b1 = g.b1;
            sg.nsub = g.nsub;
            sg.first = b.seq_sub[im.seq_off + seq];
            sg.n = min(g.nsub - sg.first, SQ);
            sg.rec_base = im.rec_off + seq * SQ;
        }
        s_geo[threadIdx.x] = sg;
    }
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const MjScanCtx cx = make_ctx(b, im, s_pairs, true);
#if defined(MJ_SYNC_TRACE)
    const unsigned long long t_begin = wall_clock64();
#endif

    /* cold pass: the lane's subsequences back to back from a cold state at its first
     * (exact for the first lane of an interval), one record after each */
    {
        const uint32_t sw = wave * SYNC_SPW + lane / SYNC_SEQ_LANES;      /* the lane's sequence within the workgroup */
        const uint32_t GC = G * SYNC_SPW;                                  /* subsequences per lane in this pass */
        const SeqGeo g = s_geo[sw];
        const uint32_t u0 = (lane % SYNC_SEQ_LANES) * GC;
        bool more = false;
        uint32_t task = 0;
        if (u0 < g.n) {
            const uint32_t j0 = g.first + u0;
            MjLane L;
            mj_lane_init(cx, L, g.b0 + j0 * S, 0, 0);
            for (uint32_t q = 0; q < GC; q++) {
                if (u0 + q < g.n) {
                    const LaneRec r = run_nowrite(cx, L, sub_end(g.b0, g.b1, g.nsub, j0 + q, S));
                    store_rec(b.rec + g.rec_base + u0 + q, r);
                }
            }
            more = (u0 + GC < g.n);
            task = pack_task((L.c << 8) | L.k, L.p - (g.b0 + (j0 + GC) * S), sw, u0 + GC);
            if (!more) {
                /* the lane that ends its sequence goes on into the next one, where this workgroup has it */
                const uint32_t nsw = next_seq_here<SYNC_SEQS>(s_geo, sw, g, SQ);
                if (nsw != sw) {
                    more = true;
                    task = pack_task((L.c << 8) | L.k, L.p - (g.b0 + (g.first + g.n) * S), nsw, 0);
                }
            }
        }
        push_task(more, task, s_queue, &s_cnt[0]);
    }
    __syncthreads();

    /* rounds: every queued decoder decodes one more subsequence, replaces its record and
     * goes on while the state it reached differs from the one recorded there before */
#if defined(MJ_SYNC_TRACE)
    /* timing experiment: 100 MHz ticks of the cold pass and of the rounds by how many decoders they had
     * (slots 0 and 3 of the statistics also count in the scan kernels and are left alone) */
    unsigned long long t_mark = wall_clock64(), t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    t_acc[1] = t_mark - t_begin;
#endif
    uint32_t rounds = 0, tasks = 0;
    const uint32_t max_rounds = (MJ_DBG(b) >> 8) & 0xFFu;
    uint32_t ci = 0;                         /* counter of this round; next is ci + 1, the one to clear ci + 2 (mod 3) */
    for (;;) {
        const uint32_t total = s_cnt[ci];
        if (total == 0 || (MJ_DBG(b) & 2u) || (max_rounds && rounds >= max_rounds)) break;
        /* The number of decoders left never grows.  The last SOLO_MAX get a whole wavefront
         * each at a time (solo_nowrite), dealt out over the wavefronts of the workgroup;
         * before that they are lanes of as few wavefronts as hold them.  Wavefront w has
         * nothing more to do once no more than w decoders are left and leaves (a barrier
         * only counts the wavefronts that are still there).  Decoders are taken from one
         * queue and put back into the other, so a round needs one barrier, at its end. */
        const bool solo = total <= SOLO_MAX;
        if (total <= wave) return;
        const uint32_t cn = (ci == 2) ? 0 : ci + 1, cz = (cn == 2) ? 0 : cn + 1;
        if (threadIdx.x == 0) s_cnt[cz] = 0;        /* last read a round ago, next added to a round from now */
        const uint32_t *q_from = s_queue + (rounds & 1u) * SYNC_TPB;
        uint32_t *q_to = s_queue + ((rounds + 1u) & 1u) * SYNC_TPB;
        rounds++;
        tasks += total;
        if (solo) {
            for (uint32_t t = wave; t < total; t += SYNC_WAVES) {
                const uint32_t tw = uni(q_from[t]);
                const uint32_t sw = task_seq(tw), u = task_sub(tw);
                const uint32_t kc = (tw & 63u) | (((tw >> 6) & 15u) << 8);
                const SeqGeo g = s_geo[sw];
                MjSubRec *dst = b.rec + g.rec_base + u;
                const MjState old = dst->exit;
                const uint32_t start = g.b0 + (g.first + u) * S;
                const LaneRec r = solo_nowrite(cx, start + ((tw >> 10) & 31u), kc, sub_end(g.b0, g.b1, g.nsub, g.first + u, S), lane);
                if (lane == 0) {
                    store_rec(dst, r);
                    if (!(r.p == old.p && r.kc == old.kc)) {
                        const uint32_t nsw = (u + 1 < g.n) ? sw : next_seq_here<SYNC_SEQS>(s_geo, sw, g, SQ);
                        if (u + 1 < g.n || nsw != sw) {
                            q_to[atomicAdd(&s_cnt[cn], 1u)] = pack_task(r.kc, r.p - (start + S), nsw, nsw == sw ? u + 1 : 0u);
                        }
                    }
                }
            }
        } else if (wave * 64 < total) {
            bool more = false;
            uint32_t task = 0;
            if (threadIdx.x < total) {
                const uint32_t tk = q_from[threadIdx.x];
                const uint32_t sw = task_seq(tk), u = task_sub(tk);
                const uint32_t kc = (tk & 63u) | (((tk >> 6) & 15u) << 8);
                const SeqGeo g = s_geo[sw];
                MjSubRec *dst = b.rec + g.rec_base + u;
                const MjState old = dst->exit;
                const uint32_t start = g.b0 + (g.first + u) * S;
                const LaneRec r = decode_nowrite(cx, start + ((tk >> 10) & 31u), kc, sub_end(g.b0, g.b1, g.nsub, g.first + u, S));
                store_rec(dst, r);
                const bool differs = !(r.p == old.p && r.kc == old.kc);
                const uint32_t nsw = (u + 1 < g.n) ? sw : next_seq_here<SYNC_SEQS>(s_geo, sw, g, SQ);
                more = differs && (u + 1 < g.n || nsw != sw);
                task = pack_task(r.kc, r.p - (start + S), nsw, nsw == sw ? u + 1 : 0u);
            }
            push_task(more, task, q_to, &s_cnt[cn]);
        }
        __syncthreads();
        ci = cn;
#if defined(MJ_SYNC_TRACE)
        {
            const unsigned long long now = wall_clock64();
            const int bk = total > 64 ? -1 : (total > 8 ? 4 : 6);
            if (bk > 0) {
                t_acc[bk] += now - t_mark;
                t_acc[bk + 1] += 1;
            }
            t_mark = now;
        }
#endif
    }
#if defined(MJ_SYNC_TRACE)
    if (threadIdx.x == 0) {
        atomicAdd(&b.stats[1], t_acc[1]);
        atomicMax(&b.stats[2], wall_clock64() - t_begin);      /* the longest-lived workgroup */
        for (int i = 4; i < 8; i++) atomicAdd(&b.stats[i], t_acc[i]);
    }
    return;
#endif
    if (threadIdx.x == 0) {
        atomicAdd(&b.stats[MJ_ST_ROUNDS], (unsigned long long) rounds);
        atomicMax(&b.stats[MJ_ST_MAXROUNDS], (unsigned long long) rounds);
        atomicAdd(&b.stats[MJ_ST_TASKS], (unsigned long long) tasks);
        atomicAdd(&b.stats[MJ_ST_WGS], 1ull);
    }
}

/* intervals of more sequences than this have their sequences' totals summed by mj_k_entropy_fix_first
 * (one wavefront per sequence, chip-wide) instead of by the interval's one workgroup in mj_k_entropy_fix
 * (an 8K file without restart markers is a thousand sequences: 0.08 ms of 0.35; files of 18 sequences
 * lose 6 us to it) */
constexpr uint32_t FIX_TOTALS_EARLY = 64;

/* The totals of one sequence's records (blocks completed, DC-difference sums), by one wavefront:
 * what the prefix over the sequences of a long interval starts from.  Lane 0 parks them in
 * the sequence's entry; its entry state is filled in by mj_k_entropy_fix once the boundaries
 * are settled. */
__device__ __forceinline__ void sequence_totals(const MjSubRec *rec, uint32_t n, uint32_t lane, MjSeqEntry *dst) {
    uint32_t tnb = 0;
    int32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (uint32_t l = lane; l < n; l += 64) {
        const MjSubRec r = rec[l];
        tnb += r.nb;
        t0 += r.dcsum[0];
        t1 += r.dcsum[1];
        t2 += r.dcsum[2];
        t3 += r.dcsum[3];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        tnb += __shfl_xor(tnb, d, 64);
        t0 += __shfl_xor(t0, d, 64);
        t1 += __shfl_xor(t1, d, 64);
        t2 += __shfl_xor(t2, d, 64);
        t3 += __shfl_xor(t3, d, 64);
    }
    if (lane == 0) {
        dst->blk = tnb;
        dst->pred[0] = t0;
        dst->pred[1] = t1;
        dst->pred[2] = t2;
        dst->pred[3] = t3;
        dst->pad_ = 0;
    }
}

/* ------------------------------------------------------------------------ */
/* kernel 2a: the first cross-sequence pass, spread over the whole chip      */
/* ------------------------------------------------------------------------ */
/* Every sequence that is not the first of its interval is entered by the decoder that
 * left the sequence before it: that decoder goes on here, one wavefront per such boundary
 * (solo_nowrite: a lone decoder is several times faster as a wavefront than as a lane) and
 * eight boundaries per workgroup, so that the serial decodes of all the boundaries of a
 * long scan run on as many CUs as there are, not on the one CU of the interval's workgroup
 * in mj_k_entropy_fix.  That kernel then finds the entry states it would have used already
 * recorded in seq_seen, repeats only what changed since, and checks convergence as before. */
__global__ __launch_bounds__(MJ_WG_SEQS * 64) void mj_k_entropy_fix_first(MjDeviceBatch b) {
    if (blockIdx.x >= b.multi_count[2]) return;
    const uint2 ge = b.grp_list[blockIdx.x];
    const uint32_t ii = ge.x;
    const MjImageDev &im = b.img[ii];
    const uint32_t n_seq = b.dyn[ii].n_seq;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t sq0 = ge.y * MJ_WG_SEQS;           /* first sequence (of the image) of the workgroup */
    /* lane t of every wavefront looks at sequence t, so all wavefronts decide alike whether to stay:
     * a workgroup has work if any of its sequences belongs to an interval of several */
    bool some = false;
    if (lane < MJ_WG_SEQS && sq0 + lane < n_seq) {
        const uint32_t sub = b.seq_sub[im.seq_off + sq0 + lane];
        some = sub != 0;
        if (!some) {
            const uint32_t SQl = MJ_SEQ_LANES * im.lane_subs;
            some = interval_geo(b.ivl_start + im.ivl_off, b.seq_ivl[im.seq_off + sq0 + lane], im.sub_bits, SQl).nseq > FIX_TOTALS_EARLY;
        }
    }
    if (!__any(some)) return;
    unsigned char *s_pairs = lds_after_dc(b, 0);
    load_tables(b, im, s_pairs, true);
    __syncthreads();
    const uint32_t sq = sq0 + wave;
    if (sq >= n_seq) return;
    const MjScanCtx cx = make_ctx(b, im, s_pairs, true);
    const uint32_t S = im.sub_bits, SQ = MJ_SEQ_LANES * im.lane_subs;
    const uint32_t iv = b.seq_ivl[im.seq_off + sq], first = b.seq_sub[im.seq_off + sq];
    const Geo g = interval_geo(b.ivl_start + im.ivl_off, iv, S, SQ);
    if (g.nseq <= 1) return;
    MjSubRec *rec = b.rec + im.rec_off + (uint64_t) sq * SQ;       /* this sequence's records; the one before them is the entry */
    const uint32_t n = min(g.nsub - first, (uint32_t) SQ);
    if (lane == 0) b.seq_flag[im.seq_off + sq] = 0;
    if (first != 0) {
        const MjState e = rec[-1].exit;
        uint32_t p = uni(e.p), kc = uni(e.kc);
        /* records that come from the speculative synchronisation are one verified chain across
         * the sequences, except where the hop over the boundary was a guess or the mending
         * replaced the record before it (bit 31 of the sequence's counter): only those
         * boundaries are decoded across again */
#if defined(MJ_SPEC_ALWAYS_REDO)
        const uint32_t n_redo = n;
#else
        const uint32_t n_redo = (b.spec && !(b.spec_brk[im.seq_off + sq] & 0x80000000u)) ? 0u : n;
#endif
        if (lane == 0 && n_redo) b.seq_seen[im.seq_off + sq] = e;
        for (uint32_t l = 0; l < n_redo; l++) {
            const MjState old = rec[l].exit;
            const LaneRec o = solo_nowrite(cx, p, kc, sub_end(g.b0, g.b1, g.nsub, first + l, S), lane);
            if (lane == 0) store_rec(rec + l, o);
            p = o.p;
            kc = o.kc;
            if (uni(old.p) == o.p && uni(old.kc) == o.kc) break;
        }
    }
    /* the sequence's totals while the chip is at it, where the interval has so many sequences that
     * summing them would keep its one workgroup in mj_k_entropy_fix busy for long: that kernel
     * then sums again only the sequences it has to decode once more */
    if (g.nseq > FIX_TOTALS_EARLY) {
        if (first != 0) __threadfence_block();        /* lane 0's records, before the other lanes of the wavefront read them */
        sequence_totals(rec, n, lane, &b.seq_entry[im.seq_off + sq]);
    }
}

/* ------------------------------------------------------------------------ */
/* kernel 2: per long interval -- cross-sequence synchronisation            */
/* ------------------------------------------------------------------------ */
constexpr int FIX_TPB = 1024;

__global__ __launch_bounds__(FIX_TPB) void mj_k_entropy_fix(MjDeviceBatch b) {
    __shared__ int s_changed;
    __shared__ uint32_t s_part[FIX_TPB / 64];
    /* the grid is a bound (the host does not know how many intervals take several sequences): a
     * workgroup takes the entries blockIdx.x, blockIdx.x + gridDim.x, ... of the list */
    const uint32_t n_multi = min(*b.multi_count, b.n_multi_cap);
    for (uint32_t entry = blockIdx.x; entry < n_multi; entry += gridDim.x) {
    __syncthreads();            /* the last entry's uses of the shared words and tables are over */
    const uint2 mi = b.multi[entry];
    const MjImageDev &im = b.img[mi.x];
    const uint32_t iv = mi.y;
    const uint32_t S = im.sub_bits, SQ = MJ_SEQ_LANES * im.lane_subs;
    const Geo g = interval_geo(b.ivl_start + im.ivl_off, iv, S, SQ);
    const uint32_t seq_first = im.seq_off + b.ivl_seq[im.ivl_off + iv]; /* global slot of the interval's first sequence */
    unsigned char *s_pairs = lds_after_dc(b, 0);
    load_tables(b, im, s_pairs, true);
    __syncthreads();
    const MjScanCtx cx = make_ctx(b, im, s_pairs, true);
    MjSubRec *rec = b.rec + im.rec_off + (uint64_t) b.ivl_seq[im.ivl_off + iv] * SQ;

    /* Boundary q (1 <= q < nseq): the decoder that leaves sequence q-1 continues
     * into sequence q, overwriting its records, until it meets a recorded exit
     * state equal to its own.  Repeat while any boundary's entry state changed. */
    uint32_t iters = 0;
    for (;;) {
        if (threadIdx.x == 0) s_changed = 0;
        __syncthreads();
        for (uint32_t q = 1 + threadIdx.x; q < g.nseq; q += FIX_TPB) {
            const MjState e = rec[(uint64_t) q * SQ - 1].exit;
            const MjState seen = b.seq_seen[seq_first + q];
            const bool redo = e.p != seen.p || e.kc != seen.kc;   /* seen: what mj_k_entropy_fix_first, or the last iteration, started from */
            /* bit 0: decode again in this iteration; bit 1: decoded again at some point, so the totals on file are stale */
            b.seq_flag[seq_first + q] = (b.seq_flag[seq_first + q] & 2u) | (redo ? 3u : 0u);
            if (redo) {
                b.seq_seen[seq_first + q] = e;
                s_changed = 1;
            }
        }
        __syncthreads();
        if (!s_changed) break;
        iters++;
        /* one wavefront per boundary that has to be decoded again (solo_nowrite) */
        for (uint32_t q = 1 + (threadIdx.x >> 6); q < g.nseq; q += FIX_TPB / 64) {
            if (!(b.seq_flag[seq_first + q] & 1u)) continue;
            const MjState e = b.seq_seen[seq_first + q];
            uint32_t p = uni(e.p), kc = uni(e.kc);
            const uint32_t first = q * SQ;
            const uint32_t n = min(g.nsub - first, (uint32_t) SQ);
            for (uint32_t l = 0; l < n; l++) {
                MjSubRec *dst = rec + (uint64_t) first + l;
                const MjState old = dst->exit;
                const LaneRec o = solo_nowrite(cx, p, kc, sub_end(g.b0, g.b1, g.nsub, first + l, S), threadIdx.x & 63u);
                if ((threadIdx.x & 63u) == 0) store_rec(dst, o);
                p = o.p;
                kc = o.kc;
                if (uni(old.p) == o.p && uni(old.kc) == o.kc) break;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&b.stats[MJ_ST_MULTI], 1ull);
        atomicMax(&b.stats[MJ_ST_FIXITERS], (unsigned long long) iters);
    }

    /* Prefix over sequences: blocks and DC sums that precede each sequence.  The totals of every
     * sequence are in its entry (mj_k_entropy_fix_first); those of the sequences decoded again here
     * are summed again, one wavefront per sequence at a time (its records are one contiguous run:
     * coalesced 32-byte loads); then an exclusive scan over the sequences by the whole workgroup. */
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t q = wave; q < g.nseq; q += FIX_TPB / 64) {
        if (g.nseq > FIX_TOTALS_EARLY && !(b.seq_flag[seq_first + q] & 2u)) continue;
        const uint32_t first = q * SQ;
        sequence_totals(rec + (uint64_t) first, min(g.nsub - first, (uint32_t) SQ), lane, &b.seq_entry[seq_first + q]);
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < g.nseq; q += FIX_TPB) {
        MjState e;
        if (q == 0) {
            e.p = g.b0;
            e.kc = 0;
        } else {
            e = rec[(uint64_t) q * SQ - 1].exit;
        }
        b.seq_entry[seq_first + q].entry = e;
    }
    __syncthreads();
    const uint32_t blk_base = iv * im.ri * im.bpm;
    uint32_t carry[5] = {blk_base, 0, 0, 0, 0};
    for (uint32_t q0 = 0; q0 < g.nseq; q0 += FIX_TPB) {
        const uint32_t q = q0 + threadIdx.x;
        uint32_t v[5] = {0, 0, 0, 0, 0};
        if (q < g.nseq) {
            const MjSeqEntry se = b.seq_entry[seq_first + q];
            v[0] = se.blk;
            v[1] = (uint32_t) se.pred[0];
            v[2] = (uint32_t) se.pred[1];
            v[3] = (uint32_t) se.pred[2];
            v[4] = (uint32_t) se.pred[3];
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            /* inclusive scan inside the wavefront, then the wavefronts' totals through LDS */
            uint32_t inc = v[i];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 64);
                if (lane >= (uint32_t) d) inc += o;
            }
            __syncthreads();
            if (lane == 63) s_part[wave] = inc;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (uint32_t w = 0; w < FIX_TPB / 64; w++) {
                const uint32_t t = s_part[w];
                before += (w < wave) ? t : 0u;
                total += t;
            }
            v[i] = carry[i] + before + inc - v[i];
            carry[i] += total;
        }
        if (q < g.nseq) {
            b.seq_entry[seq_first + q].blk = v[0];
            b.seq_entry[seq_first + q].pred[0] = (int32_t) v[1];
            b.seq_entry[seq_first + q].pred[1] = (int32_t) v[2];
            b.seq_entry[seq_first + q].pred[2] = (int32_t) v[3];
            b.seq_entry[seq_first + q].pred[3] = (int32_t) v[4];
        }
    }
    }   /* entries */
}

/* ------------------------------------------------------------------------ */
/* kernel 3: writing pass, all sequences                                    */
/* ------------------------------------------------------------------------ */
constexpr int BLK_STRIDE = 128;   /* bytes per lane block buffer (unpadded: LDS capacity decides the occupancy here) */

__global__ __launch_bounds__(TPB) void mj_k_entropy_write(MjDeviceBatch b) {
    /* dynamic LDS: [tables][block buffers: TPB x 128 B][flush lists: TPB x (4 + 4) B][second-symbol tables] */
    unsigned char *s_blk = lds_after_dc(b, 0);
    uint32_t *s_list_blk = reinterpret_cast<uint32_t *>(s_blk + TPB * BLK_STRIDE);
    uint32_t *s_list_lane = s_list_blk + TPB;

    /* two workgroups of WAVES sequences per group of MJ_WG_SEQS */
    const uint32_t grp = blockIdx.x / (MJ_WG_SEQS / WAVES), part = blockIdx.x % (MJ_WG_SEQS / WAVES);
    if (grp >= b.multi_count[2]) return;
    const uint2 ge = b.grp_list[grp];
    const uint32_t ii = ge.x;
    const MjImageDev &im = b.img[ii];
    const uint32_t n_seq = b.dyn[ii].n_seq;
    const uint32_t seq0 = ge.y * MJ_WG_SEQS + part * WAVES;
    if (seq0 >= n_seq) return;
    unsigned char *s_next = reinterpret_cast<unsigned char *>(s_list_lane + TPB);
    __shared__ uint8_t s_natpos[64];
    load_tables(b, im, s_next, false);
    if (threadIdx.x < 64) s_natpos[threadIdx.x] = c_natpos[threadIdx.x];
    {   /* zero the block buffers */
        uint4 *z = reinterpret_cast<uint4 *>(s_blk);
        for (uint32_t i = threadIdx.x; i < TPB * BLK_STRIDE / 16; i += TPB) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t seq = seq0 + wave;
    if (seq >= n_seq) return;

    const uint32_t S = im.sub_bits, G = im.lane_subs, SQ = MJ_SEQ_LANES * G;
    const uint32_t iv = b.seq_ivl[im.seq_off + seq];
    const uint32_t first = b.seq_sub[im.seq_off + seq];
    const Geo g = interval_geo(b.ivl_start + im.ivl_off, iv, S, SQ);
    const uint32_t n = min(g.nsub - first, SQ);
    MjScanCtx cx = make_ctx(b, im, s_next, false);
    cx.natpos = s_natpos;
    /* The lane's subsequences: u0 .. u0 + cnt - 1 of the sequence.  A sequence is written by one
     * wavefront, each lane taking lane_subs subsequences, or (gridDim.y > 1: small batches, where
     * what counts is how long one lane takes) by gridDim.y wavefronts of different blocks, each
     * taking a run of 64 shorter lanes; the records are per subsequence, so a wavefront finds
     * the blocks and DC differences that precede its run by summing the records before it. */
    const uint32_t Gw = G / gridDim.y, ub = blockIdx.y * MJ_SEQ_LANES * Gw;
    if (ub >= n) return;
    const uint32_t u0 = ub + lane * Gw;
    const bool valid = u0 < n;
    const uint32_t cnt = valid ? min(Gw, n - u0) : 0u;
    const MjSubRec *recs = b.rec + im.rec_off + (uint64_t) seq * SQ;
    uint32_t pre_nb = 0;
    int32_t pre0 = 0, pre1 = 0, pre2 = 0, pre3 = 0;
    if (ub) {
        for (uint32_t t = lane; t < ub; t += 64) {
            const MjSubRec r = recs[t];
            pre_nb += r.nb;
            pre0 += r.dcsum[0];
            pre1 += r.dcsum[1];
            pre2 += r.dcsum[2];
            pre3 += r.dcsum[3];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            pre_nb += __shfl_xor(pre_nb, d, 64);
            pre0 += __shfl_xor(pre0, d, 64);
            pre1 += __shfl_xor(pre1, d, 64);
            pre2 += __shfl_xor(pre2, d, 64);
            pre3 += __shfl_xor(pre3, d, 64);
        }
    }
    uint32_t nb = 0, last_p = 0, last_kc = 0;
    int32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (uint32_t q = 0; q < cnt; q++) {
        const MjSubRec r = recs[u0 + q];
        nb += r.nb;
        d0 += r.dcsum[0];
        d1 += r.dcsum[1];
        d2 += r.dcsum[2];
        d3 += r.dcsum[3];
        last_p = r.exit.p;
        last_kc = r.exit.kc;
    }

    /* where the sequence starts: the interval start, or what the fix-up computed */
    const uint32_t n_blocks = im.n_mcu * im.bpm;
    const uint32_t blk_base = iv * im.ri * im.bpm;
    const uint32_t blk_end = min(blk_base + im.ri * im.bpm, n_blocks);
    MjSeqEntry se;
    se.entry.p = g.b0;
    se.entry.kc = 0;
    se.blk = blk_base;
    se.pred[0] = se.pred[1] = se.pred[2] = se.pred[3] = 0;
    if (g.nseq > 1) se = b.seq_entry[im.seq_off + seq];
    if (ub) {
        se.entry = recs[ub - 1].exit;
        se.blk += pre_nb;
        se.pred[0] += pre0;
        se.pred[1] += pre1;
        se.pred[2] += pre2;
        se.pred[3] += pre3;
    }

    /* entry state: where the lane before stopped */
    uint32_t p = __shfl_up(last_p, 1, 64), kc = __shfl_up(last_kc, 1, 64);
    if (lane == 0) {
        p = se.entry.p;
        kc = se.entry.kc;
    }
    const uint32_t own_nb = nb;
    wave_excl_scan5(nb, d0, d1, d2, d3);
    /* every block of the interval must be accounted for by its last sequence */
    if (first + n == g.nsub && valid && u0 + cnt == n && se.blk + nb + own_nb < blk_end) atomicOr(&b.dyn[ii].status, 1u);
    if (MJ_DBG(b) & 1u) return;

    uint32_t k = kc & 0xFFu, c = kc >> 8;
    uint32_t blk = se.blk + nb;
    MjAccPk acc = MjAccPk::of(d0 + se.pred[0], d1 + se.pred[1], d2 + se.pred[2], d3 + se.pred[3]);
    bool skip = (k != 0);
    const uint32_t end = sub_end(g.b0, g.b1, g.nsub, first + u0 + (cnt ? cnt - 1 : 0), S);
    int16_t *blkbuf = reinterpret_cast<int16_t *>(s_blk + (wave * 64 + lane) * BLK_STRIDE);
    uint32_t *list_blk = s_list_blk + wave * 64, *list_lane = s_list_lane + wave * 64;
    const unsigned char *wave_blk = s_blk + wave * 64 * BLK_STRIDE;
    unsigned char *coef_bytes = reinterpret_cast<unsigned char *>(cx.coefs);
    const uint32_t store_end = (MJ_DBG(b) & 4u) ? 0u : blk_end;

    MjLane L;
    bool live = valid && mj_write_live(p, k, skip, end, g.b1);
    if (live) mj_lane_init(cx, L, p, k, c);
    while (__any(live)) {
        if (live) mj_checkpoint(cx, L);
#pragma unroll
        for (int step = 0; step < MJ_CHECK_EVERY; step++) {
            bool fin = false;
            uint32_t fin_blk = 0;
            if (live) {
                mj_write_step(cx, L, skip, blk, blk_end, acc, blkbuf, fin, fin_blk);
                live = mj_write_live(L.p, L.k, skip, end, g.b1);
            }
            const unsigned long long mask = __ballot(fin);
            if (mask) {
                /* finished blocks leave as whole lines: 8 lanes x 16 bytes per block */
                const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
                if (fin) {
                    list_blk[rank] = fin_blk;
                    list_lane[rank] = lane;
                }
                const uint32_t cnt = __popcll(mask);
                for (uint32_t i0 = 0; i0 < cnt; i0 += 8) {
                    const uint32_t idx = i0 + (lane >> 3);
                    if (idx < cnt) {
                        const uint32_t src_lane = list_lane[idx], dst_blk = list_blk[idx];
                        uint4 *src = reinterpret_cast<uint4 *>(const_cast<unsigned char *>(wave_blk) + src_lane * BLK_STRIDE + (lane & 7) * 16);
                        const uint4 v = *src;
                        *src = make_uint4(0, 0, 0, 0);
                        if (dst_blk < store_end) {
                            /* not read again before the next kernel, by which time gigabytes have gone
                             * through the caches: stored past them, so that the stream bytes the lanes
                             * come back to stay (4K x 256: 3.45 -> 3.35 ms) */
                            typedef uint32_t nt4 __attribute__((ext_vector_type(4)));
                            const nt4 nv = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(nv, reinterpret_cast<nt4 *>(coef_bytes + (uint64_t) dst_blk * 128 + (lane & 7) * 16));
                        }
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* speculative synchronisation: the records without the rounds              */
/* ------------------------------------------------------------------------ */
/* For batches that leave most of the chip idle (a file or a few dozen small ones), where the
 * rounds of mj_k_entropy_sync are a chain of some tens of serial decodes that nothing runs
 * beside.  mj_spec_core.h has the idea; here:
 *   mj_k_spec_decode   every lane once per block slot ("variant"), its own G subsequences and
 *                      X more, a record after each (grid: groups of sequences x variants);
 *   mj_k_spec_maps     per subsequence end the map "candidate here -> candidate at the next
 *                      end", composed into prefix maps from the start of the sequence and one
 *                      total map per sequence (a scan inside every wavefront);
 *   mj_k_spec_resolve  the candidate the true decode is at when it enters the sequence (the
 *                      totals of the sequences before it, folded), then through the prefix
 *                      maps the true candidate at every subsequence end, whose record is
 *                      copied to where the other kernels expect the records;
 *   mj_k_spec_mend     where the chain had to guess (nothing that goes on stood where the true
 *                      decode stood when its trajectory ended): decode on from the last
 *                      verified state until the records agree, as a round would.
 * The cross-sequence kernels then run as for any batch: every sequence boundary is decoded
 * across once more, and whatever the mending left open across boundaries is settled there. */
constexpr int SPEC_TPB = MJ_WG_SEQS * 64;

struct SpecSeq {          /* what a wavefront knows about its sequence */
    Geo g;
    uint32_t first, n;    /* first subsequence within the interval, subsequences in the sequence */
    uint32_t rec_base;    /* record slot of its first subsequence */
    uint32_t lane_base;   /* lane slot (batch-wide) of the interval's first lane */
    uint32_t seq_slot;    /* sequence slot (batch-wide) */
};

__device__ __forceinline__ SpecSeq spec_seq(const MjDeviceBatch &b, const MjImageDev &im, uint32_t seq) {
    SpecSeq s;
    const uint32_t G = im.lane_subs, SQ = MJ_SEQ_LANES * G;
    const uint32_t iv = b.seq_ivl[im.seq_off + seq];
    s.g = interval_geo(b.ivl_start + im.ivl_off, iv, im.sub_bits, SQ);
    s.first = b.seq_sub[im.seq_off + seq];
    s.n = min(s.g.nsub - s.first, SQ);
    s.rec_base = im.rec_off + seq * SQ;
    s.lane_base = im.rec_off / G + b.ivl_seq[im.ivl_off + iv] * MJ_SEQ_LANES;
    s.seq_slot = im.seq_off + seq;
    return s;
}

/* the records of the trajectory of lane `lane` (of the interval) in variant v */
__device__ __forceinline__ MjSubRec *spec_nodes(const MjDeviceBatch &b, const SpecSeq &s, uint32_t lane, uint32_t v, uint32_t len) {
    return b.spec_rec + ((uint64_t) (s.lane_base + lane) * b.spec_stride + v * len);
}

/* The exit states once more, where the map kernel finds those of neighbouring lanes next to each
 * other (it compares every lane's with those of the lanes after it: out of the 32-byte records,
 * 960 bytes apart from lane to lane, that is a cache line per lane and load -- 160 MB of lines for
 * 64 small files; from here a wavefront's load is four lines): [variant][subsequence][lane slot] */
__device__ __forceinline__ MjState *spec_state_at(const MjDeviceBatch &b, const SpecSeq &s, uint32_t lane, uint32_t v, uint32_t a, uint32_t len) {
    return b.spec_state + ((uint64_t) (v * len + a) * b.spec_lanes + s.lane_base + lane);
}

/* LDS written by some lanes of a wavefront and read by others of the same wavefront */
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/* The sequences of a block: a group of MJ_WG_SEQS sequences is taken by one block of as many
 * wavefronts or by several smaller ones (a batch of a few sequences is done sooner with its
 * wavefronts spread over many compute units than with eight to a unit).  seq0: the first
 * sequence (of the image) of the block; false when there is none. */
__device__ __forceinline__ bool spec_my_seq(const MjDeviceBatch &b, uint32_t &ii, uint32_t &seq0, uint32_t &n_seq) {
    const uint32_t wpb = blockDim.x >> 6, parts = MJ_WG_SEQS / wpb;
    const uint32_t grp = blockIdx.x / parts, part = blockIdx.x % parts;
    if (grp >= b.multi_count[2]) return false;
    const uint2 ge = b.grp_list[grp];
    ii = ge.x;
    n_seq = b.dyn[ii].n_seq;
    seq0 = ge.y * MJ_WG_SEQS + part * wpb;
    return seq0 < n_seq;
}

__global__ __launch_bounds__(SPEC_TPB) void mj_k_spec_decode(MjDeviceBatch b) {
    uint32_t ii, seq0, n_seq;
    if (!spec_my_seq(b, ii, seq0, n_seq)) return;
    const MjImageDev &im = b.img[ii];
    const uint32_t v = blockIdx.y;
    if (v >= im.bpm) return;
    unsigned char *s_pairs = lds_after_dc(b, 0);
    load_tables(b, im, s_pairs, true);
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seq0 + wave >= n_seq) return;
    const SpecSeq s = spec_seq(b, im, seq0 + wave);
    const uint32_t S = im.sub_bits, G = im.lane_subs, LEN = mj_spec_len(G, b.spec);
    const uint32_t u0 = lane * G;
    if (u0 >= s.n) return;
    const uint32_t j0 = s.first + u0;
    if (j0 == 0 && v != 0) return;            /* the first lane of an interval knows its slot */
    const MjScanCtx cx = make_ctx(b, im, s_pairs, true);
    MjLane L;
    mj_lane_init(cx, L, s.g.b0 + j0 * S, 0, v);
    MjSubRec *dst = spec_nodes(b, s, lane + (s.first / G), v, LEN);
    for (uint32_t a = 0; a < LEN; a++) {
        if (j0 + a < s.g.nsub) {
            const LaneRec r = run_nowrite(cx, L, sub_end(s.g.b0, s.g.b1, s.g.nsub, j0 + a, S));
            store_rec(dst + a, r);
            MjState st;
            st.p = r.p;
            st.kc = r.kc;
            *spec_state_at(b, s, lane + (s.first / G), v, a, LEN) = st;
        }
    }
}

/* out = second after first; maps are MJ_SPEC_MAXC bytes in LDS, read and written a word at a
 * time, the first nw words of them (the candidates a batch can have: wave-uniform) */
__device__ __forceinline__ void spec_compose(const uint8_t *first, const uint8_t *second, uint8_t *out, uint32_t nw) {
    const uint32_t *f = reinterpret_cast<const uint32_t *>(first);
    uint32_t *o = reinterpret_cast<uint32_t *>(out);
#pragma unroll
    for (int w = 0; w < MJ_SPEC_MAXC / 4; w++) {
        if ((uint32_t) w < nw) {
            const uint32_t x = f[w];
            o[w] = (uint32_t) second[x & MJ_SPEC_IDX] | ((uint32_t) second[(x >> 8) & MJ_SPEC_IDX] << 8) |
                   ((uint32_t) second[(x >> 16) & MJ_SPEC_IDX] << 16) | ((uint32_t) second[(x >> 24) & MJ_SPEC_IDX] << 24);
        }
    }
}
__device__ __forceinline__ void spec_identity(uint8_t *m) {
    uint32_t *o = reinterpret_cast<uint32_t *>(m);
#pragma unroll
    for (int w = 0; w < MJ_SPEC_MAXC / 4; w++) o[w] = 0x03020100u + 0x04040404u * (uint32_t) w;
}
__device__ __forceinline__ void spec_copy(const uint8_t *src, uint8_t *dst) {       /* both 16-byte aligned */
    const uint4 *a = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    d[0] = a[0];
    d[1] = a[1];
}
/* words of a map that hold candidates: (lanes alive at a position) x (block slots), in fours */
__device__ __forceinline__ uint32_t spec_words(const MjDeviceBatch &b, uint32_t G) {
    return ((mj_spec_back(G, b.spec) + 1u) * b.spec_bpm + 3u) / 4u;
}

/* Per sequence, by one wavefront.  Lane l has the positions l * G .. l * G + G - 1 of the
 * sequence.  It builds their maps one after the other in LDS -- the entries the indices decide,
 * and at the one position of the lane where trajectories end (those of the lane mj_spec_back
 * lanes before it) the entries of the variants that end, from the recorded states -- and
 * composes them as it goes: what is stored per position is the map from the candidate at the
 * lane's first position to the candidate there.  A scan over the lanes' totals then gives
 * every lane the map from the sequence's first position to its own first position, and the
 * sequence its total.  (mj_k_spec_resolve applies the two maps one after the other.) */
__global__ __launch_bounds__(SPEC_TPB) void mj_k_spec_maps(MjDeviceBatch b) {
    /* dynamic LDS: three maps per thread */
    uint32_t ii, seq0, n_seq;
    if (!spec_my_seq(b, ii, seq0, n_seq)) return;
    const MjImageDev &im = b.img[ii];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seq0 + wave >= n_seq) return;         /* no workgroup barrier below: every wavefront works alone */
    const SpecSeq s = spec_seq(b, im, seq0 + wave);
    const uint32_t G = im.lane_subs, X = b.spec, LEN = mj_spec_len(G, X), bpm = im.bpm;
    uint8_t *maps = b.spec_map + (uint64_t) s.rec_base * MJ_SPEC_MAXC;
    auto M = [&](uint32_t k, uint32_t l) -> uint8_t * { return s_dyn + (size_t) (k * blockDim.x + (wave << 6) + l) * MJ_SPEC_MAXC; };
    const uint32_t lg = 31u - (uint32_t) __clz((int) G);       /* lane_subs is a power of two */
    const uint32_t back = mj_spec_back(G, X);
    const uint32_t nw = spec_words(b, G);

    /* The position of the lane where trajectories end (LEN - 1, LEN - 1 + G, ... of the
     * interval), if it has a map at all: the states of the candidates that go on there -- those
     * of the lanes after the one that ends -- and of the variants that end, all loads at once. */
    const uint32_t q_end = (LEN - 1u) & (G - 1u);
    const uint32_t m_end = s.first + lane * G + q_end;
    const bool ends = lane * G + q_end < s.n && m_end + 1 < s.g.nsub && m_end >= LEN - 1u;
    const uint32_t r_end = (LEN - 1u - q_end) >> lg;
    uint32_t hop[(MJ_MAX_BPM + 3) / 4] = {0, 0, 0};       /* the entries of the variants that end, a byte each */
    if (ends) {
        const uint32_t l0 = m_end >> lg;
        uint64_t theirs[MJ_SPEC_MAX_BACK][MJ_MAX_BPM], mine[MJ_MAX_BPM];
        uint32_t their_next[MJ_SPEC_MAX_BACK][MJ_MAX_BPM];
#pragma unroll
        for (uint32_t r2 = 0; r2 < MJ_SPEC_MAX_BACK; r2++) {
#pragma unroll
            for (uint32_t v2 = 0; v2 < MJ_MAX_BPM; v2++) {
                const bool on = r2 < r_end && v2 < bpm && mj_spec_valid(G, X, l0, q_end, r2, v2);
                their_next[r2][v2] = on ? mj_spec_next(G, X, bpm, q_end, r2, v2) : (uint32_t) MJ_SPEC_END;
                theirs[r2][v2] = 0;
                if (on) {
                    const MjState st = *spec_state_at(b, s, l0 - r2, v2, mj_spec_age(G, q_end, r2), LEN);
                    theirs[r2][v2] = mj_spec_state(st.p, st.kc);
                }
            }
        }
#pragma unroll
        for (uint32_t v = 0; v < MJ_MAX_BPM; v++) {
            mine[v] = 0;
            if (v < bpm && mj_spec_valid(G, X, l0, q_end, r_end, v)) {
                const MjState st = *spec_state_at(b, s, l0 - r_end, v, LEN - 1u, LEN);
                mine[v] = mj_spec_state(st.p, st.kc);
            }
        }
        /* nearly always some candidate stands exactly where the one that ends stands: look for
         * that first (the youngest lane first), and rank the others only for the variants
         * that found none */
        uint32_t none = 0;
#pragma unroll
        for (uint32_t v = 0; v < MJ_MAX_BPM; v++) {
            if (v < bpm) {
                uint32_t found = MJ_SPEC_END;
#pragma unroll
                for (uint32_t r2 = 0; r2 < MJ_SPEC_MAX_BACK; r2++) {
#pragma unroll
                    for (uint32_t v2 = 0; v2 < MJ_MAX_BPM; v2++) {
                        const bool hit = theirs[r2][v2] == mine[v] && their_next[r2][v2] != MJ_SPEC_END && found == MJ_SPEC_END;
                        found = hit ? their_next[r2][v2] : found;
                    }
                }
                none |= (found == MJ_SPEC_END && mj_spec_valid(G, X, l0, q_end, r_end, v)) ? 1u << v : 0u;
                hop[v / 4] |= (found & 0xFFu) << (8 * (v % 4));
            }
        }
        if (none) {
#pragma unroll
            for (uint32_t v = 0; v < MJ_MAX_BPM; v++) {
                if (none & (1u << v)) {
                    uint32_t rank = 0;
                    uint8_t e = mj_spec_guess(G, X, bpm, q_end);
#pragma unroll
                    for (uint32_t r2 = 0; r2 < MJ_SPEC_MAX_BACK; r2++) {
#pragma unroll
                        for (uint32_t v2 = 0; v2 < MJ_MAX_BPM; v2++) {
                            if (their_next[r2][v2] != MJ_SPEC_END) mj_spec_better(mine[v], theirs[r2][v2], their_next[r2][v2], &rank, &e);
                        }
                    }
                    hop[v / 4] = (hop[v / 4] & ~(0xFFu << (8 * (v % 4)))) | ((uint32_t) e << (8 * (v % 4)));
                }
            }
        }
    }

    /* the lane's positions one after the other.  All lanes do the same steps (a position without
     * a map counts as the identity), so the running map is in the same buffer in every lane: 0
     * and 2 in turn, 1 holds the map being applied. */
    uint32_t a = 0;
    spec_identity(M(a, lane));
    for (uint32_t q = 0; q < G; q++) {
        const uint32_t u = lane * G + q, m = s.first + u;
        if (u < s.n) spec_copy(M(a, lane), maps + (uint64_t) u * MJ_SPEC_MAXC);      /* candidate at the lane's first position -> candidate here */
        uint8_t *t = M(1, lane);
        if (u < s.n && m + 1 < s.g.nsub) {
            const uint32_t l0 = m >> lg, a0 = q;
            const uint32_t guess = mj_spec_guess(G, X, bpm, a0) * 0x01010101u;
#pragma unroll
            for (int w = 0; w < MJ_SPEC_MAXC / 4; w++) reinterpret_cast<uint32_t *>(t)[w] = guess;
            for (uint32_t r = 0; r <= back; r++) {
                for (uint32_t v = 0; v < bpm; v++) t[r * bpm + v] = mj_spec_entry(G, X, bpm, l0, a0, r, v);
            }
            if (q == q_end && ends) {
#pragma unroll
                for (uint32_t v = 0; v < MJ_MAX_BPM; v++) {
                    if (v < bpm && mj_spec_valid(G, X, l0, a0, r_end, v)) t[r_end * bpm + v] = (uint8_t) (hop[v / 4] >> (8 * (v % 4)));
                }
            }
        } else {
            spec_identity(t);
        }
        spec_compose(M(a, lane), t, M(2 - a, lane), nw);
        a = 2 - a;
    }
    /* inclusive scan over the lanes: after the step with distance d lane l holds what the lanes
     * l - 2d + 1 .. l do, the earlier ones first */
    for (uint32_t d = 1; d < 64; d <<= 1) {
        wave_lds_sync();
        if (lane >= d) {
            spec_compose(M(a, lane - d), M(a, lane), M(2 - a, lane), nw);
        } else {
            spec_copy(M(a, lane), M(2 - a, lane));
        }
        a = 2 - a;
    }
    wave_lds_sync();
    /* what the whole sequence does (the lanes past its end count as the identity) */
    if (lane == 63) spec_copy(M(a, 63), b.spec_tot + (uint64_t) s.seq_slot * MJ_SPEC_MAXC);
    /* exclusive: what the lanes before this one do -- the candidate at the sequence's first
     * position -> the candidate at the lane's first position */
    if (lane * G < s.n) {
        uint8_t *dst = b.spec_lane + (uint64_t) (s.rec_base / G + lane) * MJ_SPEC_MAXC;
        if (lane) {
            spec_copy(M(a, lane - 1), dst);
        } else {
            spec_identity(M(1, lane));
            spec_copy(M(1, lane), dst);
        }
    }
}

/* Intervals of more than SPEC_FOLD_SELF sequences: the candidate the true decode is at where it
 * enters each of their sequences, by one workgroup per interval -- an inclusive scan over the
 * sequences' total maps, 1024 at a time, and candidate 0 taken through it.  (A wavefront of
 * mj_k_spec_resolve folds the maps before its sequence by itself when they are few.) */
constexpr uint32_t SPEC_FOLD_SELF = 64;
constexpr int SPEC_CARRY_TPB = 512;
constexpr uint32_t SPEC_CARRY_EACH = 4;       /* sequences a thread composes by itself before the scan over the threads */

__global__ __launch_bounds__(SPEC_CARRY_TPB) void mj_k_spec_carry(MjDeviceBatch b) {
    /* dynamic LDS: three maps per thread */
    auto M = [&](uint32_t k, uint32_t t) -> uint8_t * { return s_dyn + (size_t) (k * SPEC_CARRY_TPB + t) * MJ_SPEC_MAXC; };
    const uint32_t n_multi = min(*b.multi_count, b.n_multi_cap);
    const uint32_t t = threadIdx.x;
    for (uint32_t entry = blockIdx.x; entry < n_multi; entry += gridDim.x) {
        const uint2 mi = b.multi[entry];
        const MjImageDev &im = b.img[mi.x];
        const uint32_t SQ = MJ_SEQ_LANES * im.lane_subs;
        const Geo g = interval_geo(b.ivl_start + im.ivl_off, mi.y, im.sub_bits, SQ);
        if (g.nseq <= SPEC_FOLD_SELF) continue;
        const uint32_t nw = spec_words(b, im.lane_subs);
        const uint32_t seq_first = im.seq_off + b.ivl_seq[im.ivl_off + mi.y];
        const uint8_t *tot = b.spec_tot + (uint64_t) seq_first * MJ_SPEC_MAXC;
        uint32_t c = 0;                                   /* the candidate where sequence q0 is entered */
        for (uint32_t q0 = 0; q0 < g.nseq; q0 += SPEC_CARRY_TPB * SPEC_CARRY_EACH) {
            __syncthreads();
            /* what the thread's own sequences do, composed in order (the last sequence of the
             * interval leads nowhere; a thread without sequences holds the identity, and so does
             * every thread after it) */
            const uint32_t mine0 = q0 + t * SPEC_CARRY_EACH;
            uint32_t a = 0;
            spec_identity(M(a, t));
            for (uint32_t j = 0; j < SPEC_CARRY_EACH; j++) {
                if (mine0 + j + 1 < g.nseq) {
                    spec_copy(tot + (uint64_t) (mine0 + j) * MJ_SPEC_MAXC, M(1, t));
                    spec_compose(M(a, t), M(1, t), M(2 - a, t), nw);
                    a = 2 - a;
                }
            }
            /* every thread's running map in buffer 0, then the inclusive scan over the threads between 0 and 2 */
            if (a != 0) spec_copy(M(2, t), M(0, t));
            a = 0;
            for (uint32_t d = 1; d < SPEC_CARRY_TPB; d <<= 1) {
                __syncthreads();
                if (t >= d) {
                    spec_compose(M(a, t - d), M(a, t), M(2 - a, t), nw);
                } else {
                    spec_copy(M(a, t), M(2 - a, t));
                }
                a = 2 - a;
            }
            __syncthreads();
            /* the thread's first sequence is entered where the threads before it take c; then on through its own */
            uint32_t e = t ? M(a, t - 1)[c & MJ_SPEC_IDX] : c;
            for (uint32_t j = 0; j < SPEC_CARRY_EACH; j++) {
                if (mine0 + j < g.nseq) {
                    b.spec_carry[seq_first + mine0 + j] = (uint8_t) e;
                    if (mine0 + j + 1 < g.nseq) e = tot[(uint64_t) (mine0 + j) * MJ_SPEC_MAXC + (e & MJ_SPEC_IDX)];
                }
            }
            c = M(a, SPEC_CARRY_TPB - 1)[c & MJ_SPEC_IDX];
        }
    }
}

__global__ __launch_bounds__(SPEC_TPB) void mj_k_spec_resolve(MjDeviceBatch b) {
    uint8_t (*s_fold)[MJ_SPEC_MAXC] = reinterpret_cast<uint8_t (*)[MJ_SPEC_MAXC]>(s_dyn);      /* dynamic LDS: a map per thread */
    uint32_t ii, seq0, n_seq;
    if (!spec_my_seq(b, ii, seq0, n_seq)) return;
    const MjImageDev &im = b.img[ii];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seq0 + wave >= n_seq) return;
    const SpecSeq s = spec_seq(b, im, seq0 + wave);
    const uint32_t G = im.lane_subs, X = b.spec, LEN = mj_spec_len(G, X), bpm = im.bpm, SQ = MJ_SEQ_LANES * G;
    (void) X;
    const uint32_t lg = 31u - (uint32_t) __clz((int) G);       /* lane_subs is a power of two */
    const uint32_t nw = spec_words(b, G);
    /* The candidate the true decode is at where it enters the sequence: candidate 0 at the
     * interval's first position, taken through the total maps of the sequences before this one,
     * 64 of them at a time (a tree of compositions inside the wavefront). */
    uint32_t c = 0;
    uint32_t before = s.first / SQ;
#if defined(MJ_SPEC_NO_CARRY)
    if (false) {
#else
    if (s.g.nseq > SPEC_FOLD_SELF) {              /* mj_k_spec_carry has been here */
#endif
        c = b.spec_carry[s.seq_slot];
        before = 0;
    }
    const uint8_t *tot = b.spec_tot + (uint64_t) (s.seq_slot - before) * MJ_SPEC_MAXC;
    uint8_t *mine = s_fold[threadIdx.x];
    /* 64 maps at a time, the first time fewer: those sit in the last lanes, behind identities (an
     * identity composed after a map would drop the map's "guess" bit; composed before it, it
     * changes nothing) */
    const uint32_t lead = (64u - (before & 63u)) & 63u;
    for (uint32_t q0 = 0; q0 < before + lead; q0 += 64) {
        wave_lds_sync();
        if (q0 + lane >= lead) {
            spec_copy(tot + (uint64_t) (q0 + lane - lead) * MJ_SPEC_MAXC, mine);
        } else {
            spec_identity(mine);
        }
        for (uint32_t d = 1; d < 64; d <<= 1) {
            wave_lds_sync();
            if ((lane & (2 * d - 1)) == 0) spec_compose(mine, s_fold[threadIdx.x + d], mine, nw);
        }
        wave_lds_sync();
        c = s_fold[wave << 6][c & MJ_SPEC_IDX];
    }
    /* the true candidate at each of the lane's positions -- through the map of the lanes before
     * this one, then through the map of the lane's positions before this one -- and its record
     * (the "guess" bit is that of the last hop: of the second map where it is not the identity) */
    const uint8_t *maps = b.spec_map + (uint64_t) s.rec_base * MJ_SPEC_MAXC;
    uint32_t guessed = 0;
    uint32_t at_lane = c;
    if (lane && lane * G < s.n) at_lane = b.spec_lane[(uint64_t) (s.rec_base / G + lane) * MJ_SPEC_MAXC + (c & MJ_SPEC_IDX)];
    for (uint32_t q = 0; q < G; q++) {
        const uint32_t u = lane * G + q, m = s.first + u;
        if (u >= s.n) break;
        const uint32_t e = q ? maps[(uint64_t) u * MJ_SPEC_MAXC + (at_lane & MJ_SPEC_IDX)] : at_lane;
        const uint32_t idx = e & MJ_SPEC_IDX, r = idx / bpm, v = idx % bpm;
        MjSubRec rec = spec_nodes(b, s, (m >> lg) - r, v, LEN)[mj_spec_age(G, m & (G - 1u), r)];
        rec.pad_ = (e & MJ_SPEC_UNVERIFIED) ? 1u : 0u;
        guessed += rec.pad_;
        b.rec[s.rec_base + u] = rec;
        /* the state the next sequence of the interval is entered with, as its records assume it
         * (the cross-sequence kernels decode across the boundary again when it is no longer that) */
        if (u + 1 == s.n && m + 1 < s.g.nsub) b.seq_seen[s.seq_slot + 1] = rec.exit;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) guessed += __shfl_xor(guessed, d, 64);
    if (lane == 0) {
        /* bit 31: the boundary before this sequence has to be decoded across again
         * (mj_k_entropy_fix_first): here because the hop over it was a guess */
        b.spec_brk[s.seq_slot] = guessed | ((s.first != 0 && (c & MJ_SPEC_UNVERIFIED)) ? 0x80000000u : 0u);
        if (guessed) {
            atomicAdd(&b.stats[MJ_ST_ROUNDS], (unsigned long long) guessed);
            atomicAdd(&b.dyn[ii].spec_guessed, guessed);
        }
    }
}

/* Where the chain guessed: the decoder that stands at the last verified state goes on, replacing
 * the records, until it meets a record it agrees with that is not itself followed by a guess.
 * One wavefront per sequence, its guesses in ascending order, one decoder at a time
 * (solo_nowrite).  After every subsequence the decoder also looks whether another trajectory
 * alive there stands where it stands (some variant of a younger lane usually does within a
 * subsequence or two: that is what the speculation rests on) and then copies that trajectory's
 * records for as long as it lives instead of decoding them.  A decoder stops at the end of its
 * sequence, where the cross-sequence kernels take over. */
__global__ __launch_bounds__(SPEC_TPB) void mj_k_spec_mend(MjDeviceBatch b) {
    uint32_t ii, seq0, n_seq;
    if (!spec_my_seq(b, ii, seq0, n_seq)) return;
    const MjImageDev &im = b.img[ii];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    /* lane t of every wavefront looks at sequence t of the block, so all decide alike whether to stay */
    const bool some = lane < (blockDim.x >> 6) && seq0 + lane < n_seq && (b.spec_brk[im.seq_off + seq0 + lane] & 0x7FFFFFFFu) != 0;
    if (!__any(some)) return;
    unsigned char *s_pairs = lds_after_dc(b, 0);
    load_tables(b, im, s_pairs, true);
    __syncthreads();
    if (seq0 + wave >= n_seq) return;
    const SpecSeq s = spec_seq(b, im, seq0 + wave);
    if ((b.spec_brk[s.seq_slot] & 0x7FFFFFFFu) == 0) return;
    const MjScanCtx cx = make_ctx(b, im, s_pairs, true);
    const uint32_t S = im.sub_bits, G = im.lane_subs, X = b.spec, LEN = mj_spec_len(G, X), bpm = im.bpm;
    const uint32_t lg = 31u - (uint32_t) __clz((int) G), back = mj_spec_back(G, X);
    MjSubRec *rec = b.rec + s.rec_base;
    uint32_t cursor = 0, decoded = 0;
    for (;;) {
        uint32_t mine = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < G; q++) {
            const uint32_t u = lane * G + q;
            if (u >= cursor && u < s.n && mine == 0xFFFFFFFFu && rec[u].pad_ != 0) mine = u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mine = min(mine, (uint32_t) __shfl_xor(mine, d, 64));
        const uint32_t u0 = uni(mine);
        if (u0 == 0xFFFFFFFFu) break;
        if (s.first + u0 == 0) {                /* cannot be: the first record of an interval is exact */
            cursor = 1;
            continue;
        }
        const MjState e = rec[(int64_t) u0 - 1].exit;     /* u0 == 0: the last record of the sequence before, same interval */
        uint32_t p = uni(e.p), kc = uni(e.kc), l = u0, last = u0;       /* last: the last record replaced */
        bool done = false;
        while (l < s.n && !done) {
            const MjState old = rec[l].exit;
            const uint32_t next_guessed = (l + 1 < s.n) ? rec[l + 1].pad_ : 0u;
            const LaneRec o = solo_nowrite(cx, p, kc, sub_end(s.g.b0, s.g.b1, s.g.nsub, s.first + l, S), lane);
            if (lane == 0) store_rec(rec + l, o);
            last = l;
            decoded++;
            p = o.p;
            kc = o.kc;
            if (uni(old.p) == o.p && uni(old.kc) == o.kc && uni(next_guessed) == 0) break;
            /* Not the chain's record -- but one of the other trajectories alive here may stand
             * exactly where the decoder stands (lane i looks at candidate i): its records are then
             * the true ones for as long as it lives, and are copied instead of decoded. */
            const uint32_t m = s.first + l, l0 = m >> lg, a0 = m & (G - 1u);
            const uint32_t r2 = lane / bpm, v2 = lane % bpm;
            bool hit = false;
            if (r2 <= back && mj_spec_valid(G, X, l0, a0, r2, v2) && mj_spec_next(G, X, bpm, a0, r2, v2) != MJ_SPEC_END) {
                const MjState st = spec_nodes(b, s, l0 - r2, v2, LEN)[mj_spec_age(G, a0, r2)].exit;
                hit = st.p == o.p && st.kc == o.kc;
            }
            const unsigned long long hits = __ballot(hit);
            l++;
            if (hits) {
                const uint32_t j = (uint32_t) __ffsll((long long) hits) - 1u;      /* the youngest lane first */
                const uint32_t rj = j / bpm, vj = j % bpm;
                const MjSubRec *nodes = spec_nodes(b, s, l0 - rj, vj, LEN);
                for (uint32_t a = mj_spec_age(G, a0, rj) + 1u; a < LEN && l < s.n; a++, l++) {
                    const MjState was = rec[l].exit;
                    const uint32_t after_guessed = (l + 1 < s.n) ? rec[l + 1].pad_ : 0u;
                    MjSubRec r = nodes[a];
                    r.pad_ = 0;
                    if (lane == 0) rec[l] = r;
                    last = l;
                    p = uni(r.exit.p);
                    kc = uni(r.exit.kc);
                    if (uni(was.p) == p && uni(was.kc) == kc && uni(after_guessed) == 0) {
                        done = true;
                        break;
                    }
                }
            }
        }
        __threadfence_block();                  /* lane 0's records, before the other lanes look at the flags again */
        /* the last record of the sequence replaced: the sequence after it (same interval) is entered from another state */
        if (last + 1 >= s.n && s.first + s.n < s.g.nsub && lane == 0) atomicOr(&b.spec_brk[s.seq_slot + 1], 0x80000000u);
        cursor = last + 1;
    }
    if (lane == 0 && decoded) atomicAdd(&b.stats[MJ_ST_TASKS], (unsigned long long) decoded);
}

/* ------------------------------------------------------------------------ */
/* diagnostic: one lane per restart interval, no subsequence machinery      */
/* ------------------------------------------------------------------------ */
__global__ __launch_bounds__(TPB) void mj_k_entropy_serial(MjDeviceBatch b) {
    const uint32_t ii = blockIdx.x;
    const MjImageDev &im = b.img[ii];
    unsigned char *s_next = lds_after_dc(b, WAVES);
    __shared__ uint8_t s_natpos[64];
    load_tables(b, im, s_next, false);
    if (threadIdx.x < 64) s_natpos[threadIdx.x] = c_natpos[threadIdx.x];
    __syncthreads();
    MjScanCtx cx = make_ctx(b, im, s_next, false);
    cx.natpos = s_natpos;
    const uint32_t *ivl = b.ivl_start + im.ivl_off;
    const uint32_t n_blocks = im.n_mcu * im.bpm;
    for (uint32_t iv = threadIdx.x; iv < im.n_ivl; iv += TPB) {
        uint32_t blk = iv * im.ri * im.bpm;
        const uint32_t blk_end = min(blk + im.ri * im.bpm, n_blocks);
        const uint32_t p0 = ivl[iv] * 8u, end = ivl[iv + 1] * 8u;
        if (p0 < end) {
            MjLane L;
            mj_lane_init(cx, L, p0, 0, 0);
            MjAccPk acc = {0, 0};
            while (L.p < end && blk < blk_end) {
                mj_checkpoint(cx, L);
                mj_refill(L);
                const uint32_t at = blk;
                const MjSym s = mj_step<true>(cx, L, acc, blk, 0);
                if (s.store) cx.coefs[(uint64_t) at * 64 + cx.natpos[s.kpos]] = (int16_t) s.val;
                if (s.store2) cx.coefs[(uint64_t) at * 64 + cx.natpos[s.kpos2]] = (int16_t) s.val2;
            }
        }
        /* the stream ended before the interval's blocks did */
        if (blk < blk_end) atomicOr(&b.dyn[ii].status, 1u);
    }
}

}  // namespace

static size_t table_lds(const MjDeviceBatch &b, uint32_t waves) {
    return (size_t) b.max_ntab * sizeof(MjHuffDev) + 16 * sizeof(uint32_t) + (size_t) waves * DC_WAVE_BYTES;
}

hipError_t mj_launch_entropy(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_entropy_wg == 0) return hipSuccess;
#if defined(MJ_EXPERIMENT)
    static const size_t pad = getenv("MJ_TUNE_LDS_PAD") ? (size_t) atol(getenv("MJ_TUNE_LDS_PAD")) : 0;   /* occupancy experiments */
    static const int force_spw = getenv("MJ_TUNE_SYNC_SPW") ? atoi(getenv("MJ_TUNE_SYNC_SPW")) : 0;
#else
    const size_t pad = 0;
    const int force_spw = 0;
#endif
    uint32_t cus = 0;
    const hipError_t e = mj_device_cus(&cus);
    if (e != hipSuccess) return e;
    /* long lanes once the short ones would put four wavefronts or more on every SIMD; longer
     * ones still, two groups of sequences to a workgroup, once the groups that hold work
     * (the host knows a lower bound) are several times what the chip holds at once: the
     * ends of the chains then hide behind other workgroups' cold passes, and what counts is
     * how few decoders go on into the rounds */
    const int spw = force_spw ? force_spw : (b.n_busy_groups >= MJ_SYNC4_MIN_GROUPS_PER_CU * cus) ? 4 : (b.n_entropy_wg >= 2 * cus) ? 2 : 1;
    const uint32_t tpb = spw == 4 ? sync_threads(4) : spw == 2 ? sync_threads(2) : sync_threads(1);
    const uint32_t seqs = (spw == 4 ? sync_groups(4) : 1) * MJ_WG_SEQS;
    const size_t lds = table_lds(b, 0) + seqs * sizeof(SeqGeo) + 4 * sizeof(uint32_t) + 2 * tpb * sizeof(uint32_t) +
                       (size_t) b.max_nac * sizeof(MjHuffRun) + pad;
    if (spw == 4) {
        hipLaunchKernelGGL(mj_k_entropy_sync<4>, dim3(b.n_sync_pairs), dim3(tpb), lds, s, b);
    } else if (spw == 2) {
        hipLaunchKernelGGL(mj_k_entropy_sync<2>, dim3(b.n_entropy_wg), dim3(tpb), lds, s, b);
    } else {
        hipLaunchKernelGGL(mj_k_entropy_sync<1>, dim3(b.n_entropy_wg), dim3(tpb), lds, s, b);
    }
    return hipGetLastError();
}

hipError_t mj_launch_spec(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_entropy_wg == 0) return hipSuccess;
    uint32_t cus = 0;
    const hipError_t e = mj_device_cus(&cus);
    if (e != hipSuccess) return e;
    /* few sequences: small blocks, so that they spread over the chip (the decode keeps two
     * wavefronts to a block, which share the tables in LDS) */
    const bool few = (uint64_t) b.n_entropy_wg * MJ_WG_SEQS <= 4ull * cus;
    /* the kernels without tables: a wavefront per block while that is a handful per compute unit
     * (maps, resolve) */
    const bool few_o = (uint64_t) b.n_entropy_wg * MJ_WG_SEQS <= 16ull * cus;
    const uint32_t wd = few ? 2u : (uint32_t) MJ_WG_SEQS, wo = few_o ? 1u : (uint32_t) MJ_WG_SEQS;
    const size_t lds = table_lds(b, 0) + (size_t) b.max_nac * sizeof(MjHuffRun);
    hipLaunchKernelGGL(mj_k_spec_decode, dim3(b.n_entropy_wg * (MJ_WG_SEQS / wd), b.spec_bpm), dim3(wd * 64), lds, s, b);
    hipLaunchKernelGGL(mj_k_spec_maps, dim3(b.n_entropy_wg * (MJ_WG_SEQS / wo)), dim3(wo * 64), (size_t) 3 * wo * 64 * MJ_SPEC_MAXC, s, b);
    if (b.spec_long && b.n_multi_cap) {
        hipLaunchKernelGGL(mj_k_spec_carry, dim3(b.n_multi_cap < cus ? b.n_multi_cap : cus), dim3(SPEC_CARRY_TPB),
                           (size_t) 3 * SPEC_CARRY_TPB * MJ_SPEC_MAXC, s, b);
    }
    hipLaunchKernelGGL(mj_k_spec_resolve, dim3(b.n_entropy_wg * (MJ_WG_SEQS / wo)), dim3(wo * 64), (size_t) wo * 64 * MJ_SPEC_MAXC, s, b);
    const uint32_t wm = few ? 1u : (uint32_t) MJ_WG_SEQS;         /* it loads the tables: many threads to a block unless the blocks are very few */
    hipLaunchKernelGGL(mj_k_spec_mend, dim3(b.n_entropy_wg * (MJ_WG_SEQS / wm)), dim3(wm * 64), lds, s, b);
    return hipGetLastError();
}

hipError_t mj_launch_entropy_fix(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_multi_cap == 0 || b.n_entropy_wg == 0) return hipSuccess;
    hipLaunchKernelGGL(mj_k_entropy_fix_first, dim3(b.n_entropy_wg), dim3(MJ_WG_SEQS * 64), table_lds(b, 0) + (size_t) b.max_nac * sizeof(MjHuffRun), s, b);
    uint32_t cus = 0;
    const hipError_t e = mj_device_cus(&cus);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mj_k_entropy_fix, dim3(b.n_multi_cap < 2 * cus ? b.n_multi_cap : 2 * cus), dim3(FIX_TPB),
                       table_lds(b, 0) + (size_t) b.max_nac * sizeof(MjHuffRun), s, b);
    return hipGetLastError();
}

hipError_t mj_launch_entropy_write(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_entropy_wg == 0) return hipSuccess;
    const size_t lds = table_lds(b, 0) + (size_t) TPB * BLK_STRIDE + (size_t) TPB * 8 + (size_t) b.max_nac * sizeof(MjHuffPair);
    /* a lane per subsequence where the sequences are so few (of those that took the speculative
     * synchronisation, which all have one lane_subs) that a wavefront has its SIMD nearly to itself:
     * 64 files of 512 x 512 0.085 -> 0.061 ms; one of 7680 x 4320, which fills the chip, 0.088 -> 0.100 */
    uint32_t cus = 0;
    const hipError_t e = mj_device_cus(&cus);
    if (e != hipSuccess) return e;
    const uint32_t split = (b.spec && (uint64_t) b.n_entropy_wg * MJ_WG_SEQS <= 7ull * cus) ? b.spec_g : 1u;
    hipLaunchKernelGGL(mj_k_entropy_write, dim3(b.n_entropy_wg * (MJ_WG_SEQS / WAVES), split), dim3(TPB), lds, s, b);
    return hipGetLastError();
}

hipError_t mj_launch_entropy_serial(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_img == 0) return hipSuccess;
    hipLaunchKernelGGL(mj_k_entropy_serial, dim3(b.n_img), dim3(TPB), table_lds(b, WAVES) + (size_t) b.max_nac * sizeof(MjHuffPair), s, b);
    return hipGetLastError();
}
