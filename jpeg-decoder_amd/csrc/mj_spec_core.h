/*
 * mj_spec_core.h -- the bookkeeping of the speculative synchronisation (host + device).
 *
 * The rounds of mj_k_entropy_sync are a chain: a decoder that is out of step decodes
 * subsequence after subsequence until it happens to fall into step, and the kernel waits
 * for the unluckiest one.  What keeps a decoder out of step is, most of the time, the one
 * part of its state that the stream does not give away: which block of the MCU it is in
 * (decode.c:591-611 walks the block slots in a fixed order; the slot picks the tables).
 * A decoder that starts at a lane boundary knowing the slot is in step within a
 * subsequence or two nearly always.  Where the chip has room for several times the work,
 * every lane is therefore decoded once per possible slot ("variant"), each for its own G
 * subsequences plus X more, all at once, with a record after every subsequence; then the
 * true decode is *looked up* instead of decoded: it is the chain that starts with the
 * interval's first lane (whose state is known) and, wherever the trajectory it follows
 * ends, goes on with a younger trajectory that has reached the same state at the same
 * subsequence boundary (equal states decode identically from there on).
 *
 * Positions are subsequence ends, m = 0 .. nsub - 1 within an interval.  The candidates at
 * position m are the trajectories alive there: lane l0 = m / G and the lanes before it whose
 * extra subsequences reach m, each in all its variants; candidate (r, v) -- r lanes back,
 * variant v -- has the local index r * bpm + v (at most MJ_SPEC_MAXC of them).  The map
 * g_m takes every candidate at m to the candidate at m + 1 that continues it: itself, or
 * where it ends the youngest other candidate with the same state; where there is none the
 * chain takes a guess and marks the hop MJ_SPEC_UNVERIFIED, and a mending pass decodes on
 * from the last verified state until it meets the records again (the guessed chain is
 * still a chain of real decodes, each from the state before it, so a decoder that meets
 * one of its states may stop, as in the rounds).  Following the maps from candidate 0 at
 * position 0 is a prefix composition over positions, which is a scan, not a chain.
 */
#ifndef MJ_SPEC_CORE_H
#define MJ_SPEC_CORE_H

#include <stdint.h>

#include "mj_device.h"

#if defined(__HIPCC__)
#define MJ_SPEC_HD static inline __host__ __device__
#else
#define MJ_SPEC_HD static inline
#endif

#define MJ_SPEC_MAXC 32          /* candidates per position; a map is that many bytes */
#define MJ_SPEC_IDX 31u          /* the index bits of a map entry */
#define MJ_SPEC_UNVERIFIED 0x80u /* the hop into the next position was a guess */
#define MJ_SPEC_END 0xFFu        /* mj_spec_next: the trajectory ends at this position */

#define MJ_SPEC_MAX_BACK 3u      /* lanes before the youngest whose trajectories may still be alive at a position */

/* how many lanes back a trajectory can still be alive: candidate lanes are r = 0 .. this */
MJ_SPEC_HD uint32_t mj_spec_back(uint32_t G, uint32_t X) {
    return (G + X - 1u) / G;
}
/* subsequences one trajectory decodes */
MJ_SPEC_HD uint32_t mj_spec_len(uint32_t G, uint32_t X) {
    return G + X;
}
/* the largest X (at least 1, at most want) whose candidates fit a map; 0: the slots do not fit at all */
MJ_SPEC_HD uint32_t mj_spec_fit_x(uint32_t G, uint32_t bpm, uint32_t want) {
    uint32_t x = want;
    while (x >= 1u && ((mj_spec_back(G, x) + 1u) * bpm > MJ_SPEC_MAXC || mj_spec_back(G, x) > MJ_SPEC_MAX_BACK)) x--;
    return x;
}

/* A position m is given as its lane l0 = m / G and its place in the lane a0 = m % G.
 * Is (r, v) a candidate there: its lane exists, it is alive, and the first lane of an interval
 * has the one variant that is exact */
MJ_SPEC_HD bool mj_spec_valid(uint32_t G, uint32_t X, uint32_t l0, uint32_t a0, uint32_t r, uint32_t v) {
    if (r > l0) return false;
    if (a0 + r * G >= G + X) return false;
    return (l0 - r) != 0u || v == 0u;
}
/* index at the next position of the candidate (r, v) of this one, or MJ_SPEC_END */
MJ_SPEC_HD uint32_t mj_spec_next(uint32_t G, uint32_t X, uint32_t bpm, uint32_t a0, uint32_t r, uint32_t v) {
    if (a0 + r * G + 1u >= G + X) return MJ_SPEC_END;
    return (r + (a0 + 1u == G ? 1u : 0u)) * bpm + v;
}
/* which record of its trajectory candidate r holds for the position */
MJ_SPEC_HD uint32_t mj_spec_age(uint32_t G, uint32_t a0, uint32_t r) {
    return a0 + r * G;
}
/* the guess: variant 0 of the youngest lane, which is alive at the next position wherever this one is (X >= 1) */
MJ_SPEC_HD uint8_t mj_spec_guess(uint32_t G, uint32_t X, uint32_t bpm, uint32_t a0) {
    return (uint8_t) (mj_spec_next(G, X, bpm, a0, 0, 0) | MJ_SPEC_UNVERIFIED);
}

/* Entry (r, v) of the map of a position, as far as the indices alone decide it: where the
 * candidate goes on, it goes on as itself; everything else (candidates that do not exist, and
 * those that end here until mj_spec_better has looked for a successor) takes the guess. */
MJ_SPEC_HD uint8_t mj_spec_entry(uint32_t G, uint32_t X, uint32_t bpm, uint32_t l0, uint32_t a0, uint32_t r, uint32_t v) {
    if (!mj_spec_valid(G, X, l0, a0, r, v)) return mj_spec_guess(G, X, bpm, a0);
    const uint32_t nx = mj_spec_next(G, X, bpm, a0, r, v);
    return nx == MJ_SPEC_END ? mj_spec_guess(G, X, bpm, a0) : (uint8_t) nx;
}

/* A state as one word: bit position in the low half, zigzag index in bits 32-39, block slot above. */
MJ_SPEC_HD uint64_t mj_spec_state(uint32_t p, uint32_t kc) {
    return (uint64_t) p | ((uint64_t) kc << 32);
}
/* How good a successor a candidate that goes on (state `theirs`) is for one that ends (state
 * `mine`): 3 it stands where the other stands (a verified hop); 2 at the same bit in the same
 * block slot (it has the zigzag index wrong and will be right when the block ends); 1 in the
 * same slot; 0 no better than the guess. */
MJ_SPEC_HD uint32_t mj_spec_rank(uint64_t mine, uint64_t theirs) {
    const uint64_t d = mine ^ theirs;
    return d == 0 ? 3u : (d >> 40) != 0 ? 0u : (uint32_t) d == 0u ? 2u : 1u;
}
/* The entry of a candidate that ends at a position, improved by one candidate that goes on
 * (asked for the candidates youngest lane first, variants in order): the first of the highest
 * rank wins.  rank and entry start as 0 and the guess. */
MJ_SPEC_HD void mj_spec_better(uint64_t mine, uint64_t theirs, uint32_t their_next, uint32_t *rank, uint8_t *entry) {
    const uint32_t rk = mj_spec_rank(mine, theirs);
    if (rk > *rank) {
        *rank = rk;
        *entry = (uint8_t) (their_next | (rk == 3u ? 0u : MJ_SPEC_UNVERIFIED));
    }
}

/* The map of position l0 * G + a0 (the kernels build it in the same two steps, entry by entry).
 * state(r, v) returns mj_spec_state of candidate (r, v) there and is only asked for candidates
 * that exist. */
template <typename StateFn>
MJ_SPEC_HD void mj_spec_map(uint32_t G, uint32_t X, uint32_t bpm, uint32_t l0, uint32_t a0, StateFn state, uint8_t *g) {
    const uint32_t back = mj_spec_back(G, X);
    for (uint32_t i = 0; i < MJ_SPEC_MAXC; i++) g[i] = mj_spec_guess(G, X, bpm, a0);
    for (uint32_t r = 0; r <= back; r++) {
        for (uint32_t v = 0; v < bpm; v++) {
            g[r * bpm + v] = mj_spec_entry(G, X, bpm, l0, a0, r, v);
            if (!mj_spec_valid(G, X, l0, a0, r, v) || mj_spec_next(G, X, bpm, a0, r, v) != MJ_SPEC_END) continue;
            const uint64_t mine = state(r, v);
            uint32_t rank = 0;
            for (uint32_t r2 = 0; r2 < r; r2++) {          /* the lanes after it are the ones that go on */
                for (uint32_t v2 = 0; v2 < bpm; v2++) {
                    if (!mj_spec_valid(G, X, l0, a0, r2, v2)) continue;
                    const uint32_t nx2 = mj_spec_next(G, X, bpm, a0, r2, v2);
                    if (nx2 == MJ_SPEC_END) continue;
                    mj_spec_better(mine, state(r2, v2), nx2, &rank, &g[r * bpm + v]);
                }
            }
        }
    }
}

/* out = second after first (out may be neither of them) */
MJ_SPEC_HD void mj_spec_compose(const uint8_t *first, const uint8_t *second, uint8_t *out) {
    for (uint32_t i = 0; i < MJ_SPEC_MAXC; i++) out[i] = second[first[i] & MJ_SPEC_IDX];
}

#endif
