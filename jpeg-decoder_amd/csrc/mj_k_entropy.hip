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

#include "mj_entropy_dev.h"

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

/* zigzag index -> natural position, for the kernels that store coefficients */
__constant__ uint8_t c_natpos[64] = MJ_NATPOS_INIT;

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
