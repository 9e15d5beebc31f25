/*
 * mj_emu.cpp -- host-side emulation of the entropy kernels' algorithm, lane by lane
 * and round by round, built on the same mj_decode_span()/mj_classify() the kernels
 * use.  It exists so the subsequence-synchronisation logic (which only the GPU runs
 * in production) has CPU test coverage, and to measure how many synchronisation
 * rounds real streams need.  It is not reachable from the decode entry points.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../host/mj_host.h"
#include "mj_colour_core.h"
#include "mj_entropy_core.h"
#include "mj_spec_core.h"

namespace {

struct Layout {
    std::vector<uint8_t> cmp;         // compacted stream (+ padding)
    std::vector<uint32_t> ivl_start;  // n_ivl + 1 byte offsets
    uint32_t n_rst = 0;
    uint32_t term = MJ_NO_TERMINATOR;
};

// What the scan kernels compute: unstuffed stream + interval starts.
void emu_scan(const uint8_t *raw, size_t n, uint32_t n_ivl, Layout &L) {
    L.cmp.clear();
    L.ivl_start.assign(n_ivl + 1, 0);
    uint32_t k = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t prev = i ? raw[i - 1] : 0, cur = raw[i], next = (i + 1 < n) ? raw[i + 1] : 0;
        int cls = mj_classify(prev, cur, next);
        if (cls == MJ_B_TERM) {
            L.term = (uint32_t) i;
            break;
        }
        if (cls == MJ_B_KEEP) {
            L.cmp.push_back((uint8_t) cur);
        } else if (cls == MJ_B_RST) {
            k++;
            if (k < n_ivl) {
                L.ivl_start[k] = (uint32_t) L.cmp.size();
            }
        }
    }
    L.n_rst = k;
    for (uint32_t j = k + 1; j <= n_ivl; j++) {
        L.ivl_start[j] = (uint32_t) L.cmp.size();
    }
    L.ivl_start[n_ivl] = (uint32_t) L.cmp.size();
    L.cmp.resize(L.cmp.size() + MJ_STREAM_PAD + 8, 0);
    // word alignment of the buffer end
    while (L.cmp.size() % 4) L.cmp.push_back(0);
}

struct Rec {
    uint32_t p, k, c, nb;
    int32_t d[4];
};

Rec span(const MjScanCtx &cx, uint32_t p, uint32_t k, uint32_t c, uint32_t end) {
    Rec r{p, k, c, 0, {0, 0, 0, 0}};
    mj_decode_span(cx, r.p, r.k, r.c, end, r.nb, r.d, 1);
    return r;
}

bool same_state(const Rec &a, const Rec &b) {
    return a.p == b.p && a.k == b.k && a.c == b.c;
}

}  // namespace

extern "C" {

/* The sixteen-at-a-time byte classifier of the scan kernels, byte by byte for the tests:
 * out[i] = class (MJ_B_*) of byte pos + i.  buf must be readable up to pos + 16. */
void mj_emu_classify16(const uint8_t *buf, uint32_t len, uint32_t pos, uint8_t *out) {
    uint32_t w[4];
    const MjClass16 c = mj_classify16_at(buf, len, pos, w);
    for (int i = 0; i < 16; i++) {
        const uint32_t bit = 1u << (8 * (i & 3) + 7);
        out[i] = (c.keep[i >> 2] & bit) ? MJ_B_KEEP : (c.rst[i >> 2] & bit) ? MJ_B_RST : (c.term[i >> 2] & bit) ? MJ_B_TERM : MJ_B_DROP;
    }
}

/* The integer colour transform of the IDCT + colour kernel (mj_colour_core.h), one pixel
 * per (y, cb, cr) triple: out gets R, G, B per pixel, as the oracle's row functions do. */
int mj_emu_ycbcr_row(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int n, uint8_t *out) {
    static constexpr MjChromaTable tab = mj_chroma_table();
    for (int i = 0; i < n; i++) {
        const uint32_t px = mj_ycc_to_rgb(y[i], cb[i], cr[i], tab.t[cb[i]], tab.t[256 + cr[i]]);
        out[3 * i] = (uint8_t) px;
        out[3 * i + 1] = (uint8_t) (px >> 8);
        out[3 * i + 2] = (uint8_t) (px >> 16);
    }
    return 0;
}

/* How many (Cb, Cr) pairs have the G-share sum of the pair whose G wraps: the kernel tells
 * that pair by its sum, so the answer must be one. */
int mj_emu_wrap_sum_pairs(void) {
    constexpr int32_t want = mj_cb_term(MJ_WRAP_CB).g + mj_cr_term(MJ_WRAP_CR).g;
    int n = 0;
    for (uint32_t cb = 0; cb < 256; cb++) {
        for (uint32_t cr = 0; cr < 256; cr++) n += (mj_cb_term(cb).g + mj_cr_term(cr).g == want);
    }
    return n;
}

/* mj_spec_core.h checked against a brute-force walk: for lanes of G subsequences with X more,
 * every candidate that exists at a position and goes on is a candidate at the next position
 * with the index mj_spec_next says, one subsequence older, and the candidates that exist are
 * exactly the lanes whose trajectories cover the position.  Returns 0, or the line that failed. */
int mj_emu_spec_selftest(void) {
    for (uint32_t G = 1; G <= 8; G *= 2) {
        for (uint32_t X = 1; X <= 6; X++) {
            for (uint32_t bpm = 1; bpm <= 10; bpm++) {
                const uint32_t x = mj_spec_fit_x(G, bpm, X);
                if (x == 0) return __LINE__;
                if ((mj_spec_back(G, x) + 1) * bpm > MJ_SPEC_MAXC || mj_spec_back(G, x) > MJ_SPEC_MAX_BACK) return __LINE__;
                if (x != X) continue;
                for (uint32_t m = 0; m < 5 * G + 7; m++) {
                    const uint32_t l0 = m / G, a0 = m % G;
                    for (uint32_t r = 0; r <= mj_spec_back(G, X) + 1; r++) {
                        for (uint32_t v = 0; v < bpm; v++) {
                            /* lane l0 - r starts at subsequence (l0 - r) * G and covers G + X of them */
                            const bool covers = r <= l0 && m < (l0 - r) * G + G + X && (l0 != r || v == 0);
                            if (covers != mj_spec_valid(G, X, l0, a0, r, v)) return __LINE__;
                            if (!covers) continue;
                            if (mj_spec_age(G, a0, r) != m - (l0 - r) * G) return __LINE__;
                            const uint32_t nx = mj_spec_next(G, X, bpm, a0, r, v);
                            const bool goes_on = m + 1 < (l0 - r) * G + G + X;
                            if (goes_on != (nx != MJ_SPEC_END)) return __LINE__;
                            if (goes_on) {
                                const uint32_t l1 = (m + 1) / G, r1 = nx / bpm, v1 = nx % bpm;
                                if (v1 != v || l1 - r1 != l0 - r) return __LINE__;
                                if (!mj_spec_valid(G, X, l1, (m + 1) % G, r1, v1)) return __LINE__;
                            }
                        }
                    }
                    /* the guess is a candidate of the next position that goes on from this one */
                    const uint32_t gs = mj_spec_guess(G, X, bpm, a0) & MJ_SPEC_IDX;
                    if (!mj_spec_valid(G, X, (m + 1) / G, (m + 1) % G, gs / bpm, gs % bpm)) return __LINE__;
                }
            }
        }
    }
    /* composition: the later map's flag survives, the earlier one's does not */
    uint8_t a[MJ_SPEC_MAXC], b2[MJ_SPEC_MAXC], c[MJ_SPEC_MAXC];
    for (uint32_t i = 0; i < MJ_SPEC_MAXC; i++) {
        a[i] = (uint8_t) (((i * 7 + 3) % MJ_SPEC_MAXC) | ((i & 1) ? MJ_SPEC_UNVERIFIED : 0));
        b2[i] = (uint8_t) (((i * 5 + 1) % MJ_SPEC_MAXC) | ((i % 3 == 0) ? MJ_SPEC_UNVERIFIED : 0));
    }
    mj_spec_compose(a, b2, c);
    for (uint32_t i = 0; i < MJ_SPEC_MAXC; i++) {
        if (c[i] != b2[a[i] & MJ_SPEC_IDX]) return __LINE__;
    }
    if (mj_spec_rank(mj_spec_state(100, 0x0305), mj_spec_state(100, 0x0305)) != 3) return __LINE__;
    if (mj_spec_rank(mj_spec_state(100, 0x0305), mj_spec_state(100, 0x0309)) != 2) return __LINE__;
    if (mj_spec_rank(mj_spec_state(100, 0x0305), mj_spec_state(101, 0x0305)) != 1) return __LINE__;
    if (mj_spec_rank(mj_spec_state(100, 0x0305), mj_spec_state(100, 0x0405)) != 0) return __LINE__;
    return 0;
}

/* the byte-at-a-time rule, for comparison */
int mj_emu_classify(uint32_t prev, uint32_t cur, uint32_t next) {
    return mj_classify(prev, cur, next);
}

/*
 * Decode the coefficients of one JPEG file with the emulated kernel algorithm.
 * coefs: n_blocks*64 int16 (caller-allocated, zeroed here).  force_sub_bits: 0 =
 * the library's own choice.  stats[0] subsequences, [1] rounds summed over
 * sequences, [2] max rounds, [3] sequences, [4] multi-sequence intervals,
 * [5] fix-up iterations (max over intervals), [6] sub_bits used, [7] restart markers
 * found, [8] subsequences per lane, [9] subsequence decodes spent in the rounds.
 * [10] hops of the speculative synchronisation that were guesses, [11] subsequences the
 * mending pass decoded, [12] the most it decoded from one guess.
 * force_sub_bits: low 16 bits = subsequence length, bits 16-23 = subsequences per lane,
 * bits 24-31 = X: when not 0 the records come from the speculative synchronisation
 * (mj_spec_core.h: every lane once per block slot, X subsequences past its own) instead of
 * the rounds.
 * Returns the number of blocks, or -1.
 */
int64_t mj_emu_decode_coefs(const uint8_t *file, size_t len, int16_t *coefs, int64_t cap_blocks, uint32_t force_sub_bits,
                            int64_t *stats) {
    MjParsed P;
    const char *why = nullptr;
    if (mj_parse_headers(file, len, &P, &why) != 0) {
        return -1;
    }
    MjImageDev img;
    MjHuffSet *set = (MjHuffSet *) calloc(1, sizeof(MjHuffSet));
    if (mj_flatten_image(&P, len, &img, set, &why) != 0) {
        mj_parsed_free(&P);
        free(set);
        return -1;
    }
    const int64_t n_blocks = (int64_t) img.n_mcu * img.bpm;
    if (n_blocks > cap_blocks) {
        mj_parsed_free(&P);
        free(set);
        return -1;
    }
    memset(coefs, 0, (size_t) n_blocks * 64 * sizeof(int16_t));
    if (force_sub_bits & 0xFFFFu) {
        img.sub_bits = ((force_sub_bits & 0xFFFFu) + 31) / 32 * 32;
    }
    if ((force_sub_bits >> 16) & 0xFFu) {
        img.lane_subs = (force_sub_bits >> 16) & 0xFFu;
    }
    uint32_t spec_x = force_sub_bits >> 24;
    if (spec_x) {
        spec_x = mj_spec_fit_x(img.lane_subs, img.bpm, spec_x);
        if (spec_x == 0) {
            mj_parsed_free(&P);
            free(set);
            return -1;
        }
    }
    const uint32_t S = img.sub_bits, G = img.lane_subs, SQ = MJ_SEQ_LANES * G;

    Layout L;
    emu_scan(file + P.ent_pos, img.raw_len, img.n_ivl, L);

    MjScanCtx cx;
    cx.stream = (const uint32_t *) L.cmp.data();
    mj_ctx_slots(cx, img.bpm, img.blk_comp, img.blk_dc, img.blk_ac);
    cx.tabs = (const unsigned char *) set->tab;
    cx.walk = set->walk;
    cx.pairs = (const unsigned char *) set->run;
    cx.pair_stride = (uint32_t) sizeof(MjHuffRun);
    cx.coefs = coefs;
    static const uint8_t natpos[64] = MJ_NATPOS_INIT;
    cx.natpos = natpos;
    MjScanCtx wcx = cx;   /* the writing pass looks up second symbols, not runs */
    wcx.pairs = (const unsigned char *) set->next;
    wcx.pair_stride = (uint32_t) sizeof(MjHuffPair);

    int64_t st_subs = 0, st_rounds = 0, st_maxr = 0, st_seqs = 0, st_multi = 0, st_fix = 0, st_tasks = 0, st_guess = 0, st_mend = 0, st_mendmax = 0;
    // MJ_EMU_HIST=1: print, per round, how many decoders are still out of step and by which part of the state
    int64_t hist_store[64] = {0};
    int64_t *hist = getenv("MJ_EMU_HIST") ? hist_store : nullptr;
    int64_t mism[32][3] = {{0}};

    struct Task {
        uint32_t m;   // subsequence to decode next (index within the interval)
        Rec from;     // state to decode it from
    };

    for (uint32_t iv = 0; iv < img.n_ivl; iv++) {
        const uint32_t b0 = L.ivl_start[iv] * 8, b1 = L.ivl_start[iv + 1] * 8;
        const uint32_t bits = b1 - b0;
        uint32_t nsub = (bits + S - 1) / S;
        if (nsub == 0) nsub = 1;
        const uint32_t nseq = (nsub + SQ - 1) / SQ;
        st_subs += nsub;
        st_seqs += nseq;
        auto sub_end = [&](uint32_t j) { return (j + 1 == nsub) ? b1 : b0 + (j + 1) * S; };
        std::vector<Rec> rec(nsub);

        if (spec_x) {
            // ---- speculative synchronisation: every lane in every variant, G + X subsequences each ----
            const uint32_t X = spec_x, bpm = img.bpm, len_t = mj_spec_len(G, X);
            const uint32_t nlanes = (nsub + G - 1) / G;
            std::vector<Rec> node((size_t) nlanes * bpm * len_t, Rec{0xFFFFFFFFu, 0, 0, 0, {0, 0, 0, 0}});
            for (uint32_t l = 0; l < nlanes; l++) {
                for (uint32_t v = 0; v < (l ? bpm : 1u); v++) {
                    Rec cur{b0 + l * G * S, 0, v, 0, {0, 0, 0, 0}};
                    for (uint32_t a = 0; a < len_t && l * G + a < nsub; a++) {
                        cur = span(cx, cur.p, cur.k, cur.c, sub_end(l * G + a));
                        node[((size_t) l * bpm + v) * len_t + a] = cur;
                    }
                }
            }
            auto node_at = [&](uint32_t m, uint32_t r, uint32_t v) -> const Rec & {
                return node[((size_t) (m / G - r) * bpm + v) * len_t + mj_spec_age(G, m % G, r)];
            };
            // the chain: candidate 0 at position 0, then the maps (the kernels compose them as a scan)
            std::vector<char> guessed(nsub, 0);
            uint32_t at = 0;
            for (uint32_t m = 0; m < nsub; m++) {
                const uint32_t r = (at & MJ_SPEC_IDX) / bpm, v = (at & MJ_SPEC_IDX) % bpm;
                rec[m] = node_at(m, r, v);
                guessed[m] = (at & MJ_SPEC_UNVERIFIED) != 0;
                st_guess += guessed[m];
                if (m + 1 < nsub) {
                    uint8_t g[MJ_SPEC_MAXC];
                    mj_spec_map(G, X, bpm, m / G, m % G, [&](uint32_t r2, uint32_t v2) {
                        const Rec &n = node_at(m, r2, v2);
                        return mj_spec_state(n.p, (n.c << 8) | n.k);
                    }, g);
                    at = g[at & MJ_SPEC_IDX];
                }
            }
            // mending: from the last verified state on until the records are met again
            for (uint32_t m = 1; m < nsub; m++) {
                if (!guessed[m]) continue;
                Rec cur = rec[m - 1];
                int64_t steps = 0;
                for (uint32_t l = m; l < nsub; l++) {
                    steps++;
                    if (steps > st_mendmax) st_mendmax = steps;
                    const Rec o = span(cx, cur.p, cur.k, cur.c, sub_end(l));
                    const bool synced = same_state(o, rec[l]);
                    rec[l] = o;
                    cur = o;
                    st_mend++;
                    if (synced) break;
                }
            }
        }
        // ---- kernel A: per sequence, one lane per G consecutive subsequences ------
        for (uint32_t q = 0; q < nseq && !spec_x; q++) {
            const uint32_t first = q * SQ;
            const uint32_t n = (nsub - first < SQ) ? nsub - first : SQ;
            std::vector<Task> tasks;
            // cold pass: lane l decodes its G subsequences back to back from a cold state
            // (exact for the interval's first lane), leaving one record per subsequence
            for (uint32_t l = 0; l * G < n; l++) {
                Rec cur{b0 + (first + l * G) * S, 0, 0, 0, {0, 0, 0, 0}};
                for (uint32_t g = 0; g < G && l * G + g < n; g++) {
                    const uint32_t j = first + l * G + g;
                    rec[j] = span(cx, cur.p, cur.k, cur.c, sub_end(j));
                    cur = rec[j];
                }
                if ((l + 1) * G < n) tasks.push_back(Task{first + (l + 1) * G, cur});
            }
            // rounds: every decoder still out of step decodes one more subsequence, replaces
            // its record and stops when the state it reaches equals the one recorded there before
            int64_t rounds = 0;
            while (!tasks.empty()) {
                rounds++;
                st_tasks += (int64_t) tasks.size();
                if (hist && rounds < 32) {
                    hist[rounds] += (int64_t) tasks.size();
                    hist[32 + rounds]++;
                }
                std::vector<Task> next;
                for (const Task &t : tasks) {
                    const Rec out = span(cx, t.from.p, t.from.k, t.from.c, sub_end(t.m));
                    const bool synced = same_state(out, rec[t.m]);
                    if (hist && !synced && rounds < 32) {
                        /* which part of the state differs: bit position; else zigzag index; else block slot */
                        const int what = out.p != rec[t.m].p ? 0 : (out.k != rec[t.m].k ? 1 : 2);
                        mism[rounds][what]++;
                    }
                    rec[t.m] = out;
                    if (!synced && t.m + 1 < first + n) next.push_back(Task{t.m + 1, out});
                }
                tasks.swap(next);
            }
            st_rounds += rounds;
            if (rounds > st_maxr) st_maxr = rounds;
        }

        // ---- kernel B: cross-sequence fix-up (intervals longer than one sequence) --
        if (nseq > 1 && !spec_x) {
            st_multi++;
            std::vector<Rec> seen(nseq);
            std::vector<char> have(nseq, 0);
            int64_t iters = 0;
            for (;;) {
                bool changed = false;
                iters++;
                // phase 1: every boundary reads its entry state
                std::vector<Rec> entry(nseq);
                for (uint32_t q = 1; q < nseq; q++) entry[q] = rec[q * SQ - 1];
                // phase 2: boundaries whose entry changed re-decode into their sequence
                for (uint32_t q = 1; q < nseq; q++) {
                    if (have[q] && same_state(seen[q], entry[q])) continue;
                    seen[q] = entry[q];
                    have[q] = 1;
                    changed = true;
                    Rec cur = entry[q];
                    const uint32_t first = q * SQ;
                    const uint32_t n = (nsub - first < SQ) ? nsub - first : SQ;
                    for (uint32_t l = 0; l < n; l++) {
                        Rec o = span(cx, cur.p, cur.k, cur.c, sub_end(first + l));
                        const bool synced = same_state(o, rec[first + l]);
                        rec[first + l] = o;
                        cur = o;
                        if (synced) break;
                    }
                }
                if (!changed) break;
            }
            if (iters > st_fix) st_fix = iters;
        }

        // ---- prefix sums + write pass (mj_k_entropy_write) --------------------------
        // lane = G consecutive subsequences: entry state = exit of the record before its
        // first one, counts = sums over its own records
        const uint32_t blk_base = iv * img.ri * img.bpm;
        uint32_t blk_end = blk_base + img.ri * img.bpm;
        if (blk_end > (uint32_t) n_blocks) blk_end = (uint32_t) n_blocks;
        uint32_t blk0 = blk_base;
        int32_t pred[4] = {0, 0, 0, 0};
        for (uint32_t j = 0; j < nsub; j += G) {
            const uint32_t jl = (j + G < nsub) ? j + G - 1 : nsub - 1;   // the lane's last subsequence
            uint32_t p = j ? rec[j - 1].p : b0, k = j ? rec[j - 1].k : 0, c = j ? rec[j - 1].c : 0;
            uint32_t blk = blk0;
            MjAccPk d = MjAccPk::of(pred[0], pred[1], pred[2], pred[3]);
            bool skip = (k != 0);
            int16_t buf[64];
            memset(buf, 0, sizeof buf);
            const uint32_t end = sub_end(jl);
            if (mj_write_live(p, k, skip, end, b1)) {
                MjLane L;
                mj_lane_init(wcx, L, p, k, c);
                while (mj_write_live(L.p, L.k, skip, end, b1)) {
                    mj_checkpoint(wcx, L);
                    bool fin = false;
                    uint32_t fin_blk = 0;
                    mj_write_step(wcx, L, skip, blk, blk_end, d, buf, fin, fin_blk);
                    if (fin) {
                        /* the kernels leave blocks in natural order; this function reports them
                         * in zigzag order, as the reference hands them to dequant_data_unit */
                        for (int z = 0; z < 64; z++) coefs[(size_t) fin_blk * 64 + z] = buf[natpos[z]];
                        memset(buf, 0, sizeof buf);
                    }
                }
            }
            // what the kernel uses: the recorded counts, not the write pass's own
            for (uint32_t u = j; u <= jl; u++) {
                blk0 += rec[u].nb;
                for (int i = 0; i < 4; i++) pred[i] += rec[u].d[i];
            }
        }
    }
    if (hist) {
        for (int r = 1; r < 32 && hist[32 + r]; r++) {
            fprintf(stderr, "round %d: %lld decoders in %lld sequences; still out of step by bit position %lld, by zigzag index only %lld, by block slot only %lld\n",
                    r, (long long) hist[r], (long long) hist[32 + r], (long long) mism[r][0], (long long) mism[r][1], (long long) mism[r][2]);
        }
    }
    if (stats) {
        stats[0] = st_subs;
        stats[1] = st_rounds;
        stats[2] = st_maxr;
        stats[3] = st_seqs;
        stats[4] = st_multi;
        stats[5] = st_fix;
        stats[6] = S;
        stats[7] = L.n_rst;
        stats[8] = G;
        stats[9] = st_tasks;
        stats[10] = st_guess;
        stats[11] = st_mend;
        stats[12] = st_mendmax;
    }
    mj_parsed_free(&P);
    free(set);
    return n_blocks;
}

}  // extern "C"
