/*
 * mj_k_spec.hip -- the speculative synchronisation of the Huffman decode: the subsequence
 * records of mj_k_entropy.hip without its rounds, for batches that leave the chip idle.
 * Replaces, like mj_k_entropy_sync, the part of decode_scan (decode.c:535-663) that a serial
 * decoder does not have: finding the state every lane of the parallel decode starts from.
 */
#include "mj_entropy_dev.h"
#include "mj_spec_core.h"

namespace {

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

}  // namespace

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

