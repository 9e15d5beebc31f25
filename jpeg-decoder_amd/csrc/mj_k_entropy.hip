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
 * One wavefront owns one *sequence* of up to 64 consecutive subsequences of one
 * interval; the four wavefronts of a workgroup share the image's Huffman lookup
 * tables in LDS.
 *   mj_k_entropy_sync   cold decode + lock-step synchronisation rounds (cross-lane
 *                       shuffles), one 32-byte record per subsequence to HBM;
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
#include "mj_entropy_core.h"
#include "mj_internal.h"

namespace {

constexpr int WAVES = 4;
constexpr int TPB = WAVES * 64;

struct Geo {          /* wave-uniform geometry of one interval */
    uint32_t b0, b1;  /* bit range in the compacted stream */
    uint32_t nsub;
    uint32_t nseq;
};

__device__ __forceinline__ Geo interval_geo(const uint32_t *ivl, uint32_t iv, uint32_t S) {
    Geo g;
    const uint32_t a = ivl[iv], e = ivl[iv + 1];
    g.b0 = a * 8u;
    g.b1 = (e > a ? e : a) * 8u;
    g.nsub = (g.b1 - g.b0 + S - 1) / S;
    if (g.nsub == 0) g.nsub = 1;
    g.nseq = (g.nsub + MJ_SEQ_SUBS - 1) / MJ_SEQ_SUBS;
    return g;
}

__device__ __forceinline__ uint32_t sub_end(const Geo &g, uint32_t j, uint32_t S) {
    return (j + 1 >= g.nsub) ? g.b1 : g.b0 + (j + 1) * S;
}

extern __shared__ __align__(16) unsigned char s_dyn[];

/* LDS carve-up shared by all entropy kernels: [Huffman tables][16 slot words]... */
__device__ __forceinline__ uint32_t *lds_slots(const MjDeviceBatch &b) {
    return reinterpret_cast<uint32_t *>(s_dyn + b.max_ntab * sizeof(MjHuffDev));
}
/* per-lane DC accumulators: [wave][component][lane] */
__device__ __forceinline__ int32_t *lds_dc(const MjDeviceBatch &b) {
    return reinterpret_cast<int32_t *>(s_dyn + b.max_ntab * sizeof(MjHuffDev) + 16 * sizeof(uint32_t));
}
constexpr uint32_t DC_LDS_BYTES = 4u * 64u * MJ_MAX_COMP * sizeof(int32_t);   /* up to 4 waves */
__device__ __forceinline__ unsigned char *lds_after_dc(const MjDeviceBatch &b) {
    return reinterpret_cast<unsigned char *>(lds_dc(b)) + DC_LDS_BYTES;
}

/* cooperative copy of the image's Huffman tables and per-slot table selection into
 * LDS (the caller synchronises) */
__device__ __forceinline__ void load_tables(const MjDeviceBatch &b, const MjImageDev &im) {
    const MjHuffSet &set = b.sets[im.huff_set];
    const uint32_t n = set.ntab * (uint32_t) (sizeof(MjHuffDev) / 4);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(set.tab);
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_dyn);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x < im.bpm) {
        lds_slots(b)[threadIdx.x] = mj_slot_pack(im.blk_comp[threadIdx.x], im.blk_dc[threadIdx.x], im.blk_ac[threadIdx.x]);
    }
}

__device__ __forceinline__ MjScanCtx make_ctx(const MjDeviceBatch &b, const MjImageDev &im) {
    MjScanCtx cx;
    cx.stream = reinterpret_cast<const uint32_t *>(b.cmp + im.cmp_off);
    cx.tabs = s_dyn;
    cx.walk = b.sets[im.huff_set].walk;
    cx.slots = lds_slots(b);
    cx.coefs = b.coefs + im.coef_off * 64;
    cx.bpm = im.bpm;
    cx.wi_mask = (b.dbg & 32u) ? 15u : 0xFFFFFFFFu;
    return cx;
}

struct LaneRec {      /* what a lane knows about one subsequence */
    uint32_t p, kc, nb;
    int32_t d0, d1, d2, d3;
};

/* acc: this lane's accumulators, acc[comp * 64] */
__device__ __forceinline__ LaneRec decode_nowrite(const MjScanCtx &cx, uint32_t p, uint32_t kc, uint32_t end, int32_t *acc) {
    uint32_t k = kc & 0xFFu, c = kc >> 8, nb = 0;
    acc[0] = 0;
    acc[64] = 0;
    acc[128] = 0;
    acc[192] = 0;
    mj_decode_span(cx, p, k, c, end, nb, acc, 64);
    LaneRec r;
    r.p = p;
    r.kc = (c << 8) | k;
    r.nb = nb;
    r.d0 = acc[0];
    r.d1 = acc[64];
    r.d2 = acc[128];
    r.d3 = acc[192];
    return r;
}

__device__ __forceinline__ LaneRec shfl_up_rec(const LaneRec &r, int d) {
    LaneRec o;
    o.p = __shfl_up(r.p, d, 64);
    o.kc = __shfl_up(r.kc, d, 64);
    o.nb = __shfl_up(r.nb, d, 64);
    o.d0 = __shfl_up(r.d0, d, 64);
    o.d1 = __shfl_up(r.d1, d, 64);
    o.d2 = __shfl_up(r.d2, d, 64);
    o.d3 = __shfl_up(r.d3, d, 64);
    return o;
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
/* kernel 1: per sequence -- cold decode, lock-step synchronisation rounds, */
/* and, for single-sequence intervals, prefix sums + the writing pass       */
/* ------------------------------------------------------------------------ */
__global__ __launch_bounds__(TPB) void mj_k_entropy_sync(MjDeviceBatch b) {
    const uint32_t ii = b.wg_img[blockIdx.x];
    const MjImageDev &im = b.img[ii];
    const uint32_t n_seq = b.dyn[ii].n_seq;
    const uint32_t seq0 = (blockIdx.x - im.wg_first) * WAVES;
    if (seq0 >= n_seq) return;
    load_tables(b, im);
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t seq = seq0 + wave;
    if (seq >= n_seq) return;

    const uint32_t S = im.sub_bits;
    const uint32_t iv = b.seq_ivl[im.seq_off + seq];
    const uint32_t first = b.seq_sub[im.seq_off + seq];
    const Geo g = interval_geo(b.ivl_start + im.ivl_off, iv, S);
    const uint32_t n = min(g.nsub - first, (uint32_t) MJ_SEQ_SUBS);
    const MjScanCtx cx = make_ctx(b, im);

    const bool valid = lane < n;
    const uint32_t j = first + lane;
    const uint32_t start = g.b0 + j * S;
    const uint32_t end_own = sub_end(g, j, S);

    /* round 0: cold start at the subsequence boundary (exact for the interval's first) */
    LaneRec rec = {start, 0, 0, 0, 0, 0, 0};
    int32_t *acc = lds_dc(b) + wave * 64 * MJ_MAX_COMP + lane;
    if (valid) rec = decode_nowrite(cx, start, 0, end_own, acc);
    uint32_t carry_p = rec.p, carry_kc = rec.kc;
    bool active = valid && (lane + 1 < n);
    uint32_t rounds = 0;
    for (uint32_t r = 1; r < n; r++) {
        if (!__any(active) || (b.dbg & 2u)) break;
        rounds++;
        LaneRec out = rec;
        if (active) out = decode_nowrite(cx, carry_p, carry_kc, sub_end(g, j + r, S), acc);
        /* lane m = l + r takes what lane l produced for subsequence m */
        const LaneRec in = shfl_up_rec(out, (int) r);
        const int in_active = __shfl_up((int) active, (int) r, 64);
        int synced = 0;
        if (lane >= r && in_active) {
            synced = (in.p == rec.p && in.kc == rec.kc);
            rec = in;
        }
        const int told = __shfl_down(synced, (int) r, 64);
        if (active) {
            carry_p = out.p;
            carry_kc = out.kc;
            if (told || lane + r + 1 >= n) active = false;
        }
    }
    if (lane == 0) {
        atomicAdd(&b.stats[MJ_ST_ROUNDS], (unsigned long long) rounds);
        atomicMax(&b.stats[MJ_ST_MAXROUNDS], (unsigned long long) rounds);
    }

    /* one record per subsequence for the fix-up / writing kernels */
    MjSubRec o;
    o.exit.p = rec.p;
    o.exit.kc = rec.kc;
    o.nb = valid ? rec.nb : 0;
    o.dcsum[0] = valid ? rec.d0 : 0;
    o.dcsum[1] = valid ? rec.d1 : 0;
    o.dcsum[2] = valid ? rec.d2 : 0;
    o.dcsum[3] = valid ? rec.d3 : 0;
    o.pad_ = 0;
    b.rec[(uint64_t) (im.seq_off + seq) * MJ_SEQ_SUBS + lane] = o;
}

/* ------------------------------------------------------------------------ */
/* kernel 2: per long interval -- cross-sequence synchronisation            */
/* ------------------------------------------------------------------------ */
constexpr int FIX_TPB = 1024;

__global__ __launch_bounds__(FIX_TPB) void mj_k_entropy_fix(MjDeviceBatch b) {
    __shared__ int s_changed;
    __shared__ uint32_t s_part[FIX_TPB * 5];
    const uint32_t n_multi = min(*b.multi_count, b.n_multi_cap);
    if (blockIdx.x >= n_multi) return;
    const uint2 mi = b.multi[blockIdx.x];
    const MjImageDev &im = b.img[mi.x];
    const uint32_t iv = mi.y;
    const uint32_t S = im.sub_bits;
    const Geo g = interval_geo(b.ivl_start + im.ivl_off, iv, S);
    const uint32_t seq_first = im.seq_off + b.ivl_seq[im.ivl_off + iv]; /* global slot of the interval's first sequence */
    load_tables(b, im);
    __syncthreads();
    const MjScanCtx cx = make_ctx(b, im);
    int32_t *fix_acc = lds_dc(b) + (threadIdx.x >> 6) * 64 * MJ_MAX_COMP + (threadIdx.x & 63);
    MjSubRec *rec = b.rec + (uint64_t) seq_first * MJ_SEQ_SUBS;

    /* Boundary q (1 <= q < nseq): the decoder that leaves sequence q-1 continues
     * into sequence q, overwriting its records, until it meets a recorded exit
     * state equal to its own.  Repeat while any boundary's entry state changed. */
    uint32_t iters = 0;
    for (;;) {
        if (threadIdx.x == 0) s_changed = 0;
        __syncthreads();
        for (uint32_t q = 1 + threadIdx.x; q < g.nseq; q += FIX_TPB) {
            const MjState e = rec[(uint64_t) q * MJ_SEQ_SUBS - 1].exit;
            const MjState seen = b.seq_seen[seq_first + q];
            const bool redo = (iters == 0) || e.p != seen.p || e.kc != seen.kc;
            b.seq_flag[seq_first + q] = redo ? 1u : 0u;
            if (redo) {
                b.seq_seen[seq_first + q] = e;
                s_changed = 1;
            }
        }
        __syncthreads();
        if (!s_changed) break;
        iters++;
        for (uint32_t q = 1 + threadIdx.x; q < g.nseq; q += FIX_TPB) {
            if (!b.seq_flag[seq_first + q]) continue;
            const MjState e = b.seq_seen[seq_first + q];
            uint32_t p = e.p, kc = e.kc;
            const uint32_t first = q * MJ_SEQ_SUBS;
            const uint32_t n = min(g.nsub - first, (uint32_t) MJ_SEQ_SUBS);
            for (uint32_t l = 0; l < n; l++) {
                const LaneRec o = decode_nowrite(cx, p, kc, sub_end(g, first + l, S), fix_acc);
                MjSubRec *dst = rec + (uint64_t) first + l;
                const MjState old = dst->exit;
                MjSubRec w;
                w.exit.p = o.p;
                w.exit.kc = o.kc;
                w.nb = o.nb;
                w.dcsum[0] = o.d0;
                w.dcsum[1] = o.d1;
                w.dcsum[2] = o.d2;
                w.dcsum[3] = o.d3;
                w.pad_ = 0;
                *dst = w;
                p = o.p;
                kc = o.kc;
                if (old.p == o.p && old.kc == o.kc) break;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&b.stats[MJ_ST_MULTI], 1ull);
        atomicMax(&b.stats[MJ_ST_FIXITERS], (unsigned long long) iters);
    }

    /* Prefix over sequences: blocks and DC sums that precede each sequence.
     * Each thread owns a contiguous chunk of sequences. */
    const uint32_t per = (g.nseq + FIX_TPB - 1) / FIX_TPB;
    const uint32_t q_lo = min(threadIdx.x * per, g.nseq), q_hi = min(q_lo + per, g.nseq);
    uint32_t tnb = 0;
    int32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (uint32_t q = q_lo; q < q_hi; q++) {
        const uint32_t first = q * MJ_SEQ_SUBS;
        const uint32_t n = min(g.nsub - first, (uint32_t) MJ_SEQ_SUBS);
        for (uint32_t l = 0; l < n; l++) {
            const MjSubRec &r = rec[(uint64_t) first + l];
            tnb += r.nb;
            t0 += r.dcsum[0];
            t1 += r.dcsum[1];
            t2 += r.dcsum[2];
            t3 += r.dcsum[3];
        }
    }
    s_part[threadIdx.x * 5 + 0] = tnb;
    s_part[threadIdx.x * 5 + 1] = (uint32_t) t0;
    s_part[threadIdx.x * 5 + 2] = (uint32_t) t1;
    s_part[threadIdx.x * 5 + 3] = (uint32_t) t2;
    s_part[threadIdx.x * 5 + 4] = (uint32_t) t3;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a[5] = {0, 0, 0, 0, 0};
        for (uint32_t t = 0; t < FIX_TPB; t++) {
            for (int i = 0; i < 5; i++) {
                const uint32_t v = s_part[t * 5 + i];
                s_part[t * 5 + i] = a[i];
                a[i] += v;
            }
        }
    }
    __syncthreads();
    uint32_t anb = s_part[threadIdx.x * 5 + 0];
    int32_t a0 = (int32_t) s_part[threadIdx.x * 5 + 1], a1 = (int32_t) s_part[threadIdx.x * 5 + 2];
    int32_t a2 = (int32_t) s_part[threadIdx.x * 5 + 3], a3 = (int32_t) s_part[threadIdx.x * 5 + 4];
    const uint32_t blk_base = iv * im.ri * im.bpm;
    for (uint32_t q = q_lo; q < q_hi; q++) {
        MjSeqEntry se;
        if (q == 0) {
            se.entry.p = g.b0;
            se.entry.kc = 0;
        } else {
            se.entry = rec[(uint64_t) q * MJ_SEQ_SUBS - 1].exit;
        }
        se.blk = blk_base + anb;
        se.pred[0] = a0;
        se.pred[1] = a1;
        se.pred[2] = a2;
        se.pred[3] = a3;
        se.pad_ = 0;
        b.seq_entry[seq_first + q] = se;
        const uint32_t first = q * MJ_SEQ_SUBS;
        const uint32_t n = min(g.nsub - first, (uint32_t) MJ_SEQ_SUBS);
        for (uint32_t l = 0; l < n; l++) {
            const MjSubRec &r = rec[(uint64_t) first + l];
            anb += r.nb;
            a0 += r.dcsum[0];
            a1 += r.dcsum[1];
            a2 += r.dcsum[2];
            a3 += r.dcsum[3];
        }
    }
}

/* ------------------------------------------------------------------------ */
/* kernel 3: writing pass, all sequences                                    */
/* ------------------------------------------------------------------------ */
constexpr int BLK_STRIDE = 128;   /* bytes per lane block buffer (unpadded: LDS capacity decides the occupancy here) */

__global__ __launch_bounds__(TPB) void mj_k_entropy_write(MjDeviceBatch b) {
    /* dynamic LDS: [tables][slots][block buffers: TPB x 144 B][flush lists: TPB x (4 + 4) B] */
    unsigned char *s_blk = lds_after_dc(b);
    uint32_t *s_list_blk = reinterpret_cast<uint32_t *>(s_blk + TPB * BLK_STRIDE);
    uint32_t *s_list_lane = s_list_blk + TPB;

    const uint32_t ii = b.wg_img[blockIdx.x];
    const MjImageDev &im = b.img[ii];
    const uint32_t n_seq = b.dyn[ii].n_seq;
    const uint32_t seq0 = (blockIdx.x - im.wg_first) * WAVES;
    if (seq0 >= n_seq) return;
    load_tables(b, im);
    {   /* zero the block buffers */
        uint4 *z = reinterpret_cast<uint4 *>(s_blk);
        for (uint32_t i = threadIdx.x; i < TPB * BLK_STRIDE / 16; i += TPB) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t seq = seq0 + wave;
    if (seq >= n_seq) return;

    const uint32_t S = im.sub_bits;
    const uint32_t iv = b.seq_ivl[im.seq_off + seq];
    const uint32_t first = b.seq_sub[im.seq_off + seq];
    const Geo g = interval_geo(b.ivl_start + im.ivl_off, iv, S);
    const uint32_t n = min(g.nsub - first, (uint32_t) MJ_SEQ_SUBS);
    const MjScanCtx cx = make_ctx(b, im);
    const bool valid = lane < n;
    const MjSubRec r = b.rec[(uint64_t) (im.seq_off + seq) * MJ_SEQ_SUBS + lane];

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

    uint32_t p = __shfl_up(r.exit.p, 1, 64), kc = __shfl_up(r.exit.kc, 1, 64);
    if (lane == 0) {
        p = se.entry.p;
        kc = se.entry.kc;
    }
    uint32_t nb = valid ? r.nb : 0;
    int32_t d0 = valid ? r.dcsum[0] : 0, d1 = valid ? r.dcsum[1] : 0, d2 = valid ? r.dcsum[2] : 0,
            d3 = valid ? r.dcsum[3] : 0;
    const uint32_t own_nb = nb;
    wave_excl_scan5(nb, d0, d1, d2, d3);
    /* every block of the interval must be accounted for by its last sequence */
    if (first + n == g.nsub && lane == n - 1 && se.blk + nb + own_nb < blk_end) atomicOr(&b.dyn[ii].status, 1u);
    if (b.dbg & 1u) return;

    uint32_t k = kc & 0xFFu, c = kc >> 8;
    uint32_t blk = se.blk + nb;
    int32_t *acc = lds_dc(b) + wave * 64 * MJ_MAX_COMP + lane;
    acc[0] = d0 + se.pred[0];
    acc[64] = d1 + se.pred[1];
    acc[128] = d2 + se.pred[2];
    acc[192] = d3 + se.pred[3];
    bool skip = (k != 0);
    const uint32_t end = sub_end(g, first + lane, S);
    int16_t *blkbuf = reinterpret_cast<int16_t *>(s_blk + (wave * 64 + lane) * BLK_STRIDE);
    uint32_t *list_blk = s_list_blk + wave * 64, *list_lane = s_list_lane + wave * 64;
    const unsigned char *wave_blk = s_blk + wave * 64 * BLK_STRIDE;
    unsigned char *coef_bytes = reinterpret_cast<unsigned char *>(cx.coefs);
    const uint32_t store_end = (b.dbg & 4u) ? 0u : blk_end;

    MjLane L;
    bool live = valid && mj_write_live(p, k, skip, end, g.b1);
    if (live) mj_lane_init(cx, L, p, k, c);
    while (__any(live)) {
        bool fin = false;
        uint32_t fin_blk = 0;
        if (live) {
            mj_write_step(cx, L, skip, blk, blk_end, acc, 64, blkbuf, fin, fin_blk);
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
                        *reinterpret_cast<uint4 *>(coef_bytes + (uint64_t) dst_blk * 128 + (lane & 7) * 16) = v;
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* diagnostic: one lane per restart interval, no subsequence machinery      */
/* ------------------------------------------------------------------------ */
__global__ __launch_bounds__(TPB) void mj_k_entropy_serial(MjDeviceBatch b) {
    const uint32_t ii = blockIdx.x;
    const MjImageDev &im = b.img[ii];
    load_tables(b, im);
    __syncthreads();
    const MjScanCtx cx = make_ctx(b, im);
    const uint32_t *ivl = b.ivl_start + im.ivl_off;
    const uint32_t n_blocks = im.n_mcu * im.bpm;
    for (uint32_t iv = threadIdx.x; iv < im.n_ivl; iv += TPB) {
        uint32_t blk = iv * im.ri * im.bpm;
        const uint32_t blk_end = min(blk + im.ri * im.bpm, n_blocks);
        const uint32_t p0 = ivl[iv] * 8u, end = ivl[iv + 1] * 8u;
        if (p0 >= end) continue;
        MjLane L;
        mj_lane_init(cx, L, p0, 0, 0);
        int32_t *acc = lds_dc(b) + (threadIdx.x >> 6) * 64 * MJ_MAX_COMP + (threadIdx.x & 63);
        acc[0] = acc[64] = acc[128] = acc[192] = 0;
        while (L.p < end && blk < blk_end) {
            mj_refill(cx, L);
            const MjSym s = mj_step<true>(cx, L);
            int32_t v = s.val;
            if (s.dc) v = mj_dc_add(acc, 64, s.comp, s.val);
            if (s.store) cx.coefs[(uint64_t) blk * 64 + s.kpos] = (int16_t) v;
            blk += s.done ? 1u : 0u;
        }
    }
}

}  // namespace

static size_t table_lds(const MjDeviceBatch &b, uint32_t waves) {
    return (size_t) b.max_ntab * sizeof(MjHuffDev) + 16 * sizeof(uint32_t) + (size_t) waves * 64 * MJ_MAX_COMP * sizeof(int32_t);
}

hipError_t mj_launch_entropy(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_entropy_wg == 0) return hipSuccess;
    hipLaunchKernelGGL(mj_k_entropy_sync, dim3(b.n_entropy_wg), dim3(TPB), table_lds(b, WAVES), s, b);
    return hipGetLastError();
}

hipError_t mj_launch_entropy_fix(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_multi_cap == 0 || b.n_entropy_wg == 0) return hipSuccess;
    hipLaunchKernelGGL(mj_k_entropy_fix, dim3(b.n_multi_cap), dim3(FIX_TPB), table_lds(b, FIX_TPB / 64), s, b);
    return hipGetLastError();
}

hipError_t mj_launch_entropy_write(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_entropy_wg == 0) return hipSuccess;
    const size_t lds = table_lds(b, WAVES) + (size_t) TPB * BLK_STRIDE + (size_t) TPB * 8;
    hipLaunchKernelGGL(mj_k_entropy_write, dim3(b.n_entropy_wg), dim3(TPB), lds, s, b);
    return hipGetLastError();
}

hipError_t mj_launch_entropy_serial(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_img == 0) return hipSuccess;
    hipLaunchKernelGGL(mj_k_entropy_serial, dim3(b.n_img), dim3(TPB), table_lds(b, WAVES), s, b);
    return hipGetLastError();
}
