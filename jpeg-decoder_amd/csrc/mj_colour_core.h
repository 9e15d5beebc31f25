/*
 * mj_colour_core.h -- the YCbCr -> RGB step of colour_conversion.c:71-79 in integers,
 * shared by the IDCT + colour kernel and by the host-side sweep that pins it against
 * the compiled reference for all 2^24 (Y, Cb, Cr) triples.
 *
 * The reference computes, in double and left to right, rounds to float, clamps and
 * truncates:
 *   R = Y + 1.402 (Cr - 128)
 *   G = Y - 0.34414 (Cb - 128) - 0.71414 (Cr - 128)
 *   B = Y + 1.772 (Cb - 128)
 * Every exact value is Y plus a multiple of 1/500, 1/50000 or 1/250.  A float near 256
 * has a spacing of 1.5e-5 below and 3e-5 above it, and the double arithmetic is good to
 * 1e-13, so the rounded result sits on the same side of every integer as the exact value,
 * or on the integer itself when the exact value is one: truncation of a non-negative
 * result is floor of the exact value, negative results clamp to 0, results above 256 to
 * 255.  The one exception is the clamp's own edge: a result that rounds to exactly
 * 256.0f is not "above 256", and gcc converts it through a 32-bit integer to the byte 0.
 * Only G can get there, and only for (Y, Cb, Cr) = (219, 228, 28).
 *
 * So: R = clamp(Y + floor(1.402 d_r)), B = clamp(Y + floor(1.772 d_b)),
 * G = clamp(Y + floor(-0.34414 d_b - 0.71414 d_r)) with the one triple patched.  The G
 * floor is taken from two per-chroma terms in 2^-20 fixed point, each rounded up by at
 * most one unit: the sum overshoots the exact value by at most 2^-19 < 2e-6, which is
 * less than the distance (at least 2e-5) of any non-integer exact value to the next
 * integer, and an exact integer stays itself.
 */
#ifndef MJ_COLOUR_CORE_H
#define MJ_COLOUR_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MJ_CC constexpr __host__ __device__
#else
#define MJ_CC constexpr
#endif

#define MJ_G_FRAC_BITS 20
#define MJ_WRAP_Y 219u
#define MJ_WRAP_CB 228u
#define MJ_WRAP_CR 28u

MJ_CC int32_t mj_floor_div(int64_t num, int64_t den) {   /* den > 0 */
    return (int32_t) (num / den - ((num % den != 0 && num < 0) ? 1 : 0));
}

/* Per chroma sample value: its share of G in 2^-20 fixed point, and the integer offset of
 * its own channel in the low 16 bits of a word (two's complement, upper half zero: the
 * form the packed 16-bit adds of the kernel take). */
struct MjChromaTerm {
    int32_t g;     /* floor(-coef * d * 2^20) + 1 */
    uint32_t o;    /* floor(1.772 d) for Cb, floor(1.402 d) for Cr, & 0xFFFF */
};

MJ_CC MjChromaTerm mj_cb_term(uint32_t cb) {
    const int32_t d = (int32_t) cb - 128;
    return MjChromaTerm{mj_floor_div((int64_t) -34414 * d * (1 << MJ_G_FRAC_BITS), 100000) + 1,
                        (uint32_t) mj_floor_div((int64_t) 443 * d, 250) & 0xFFFFu};
}

MJ_CC MjChromaTerm mj_cr_term(uint32_t cr) {
    const int32_t d = (int32_t) cr - 128;
    return MjChromaTerm{mj_floor_div((int64_t) -71414 * d * (1 << MJ_G_FRAC_BITS), 100000) + 1,
                        (uint32_t) mj_floor_div((int64_t) 701 * d, 500) & 0xFFFFu};
}

/* both tables, as the kernel keeps them in LDS: Cb terms, then Cr terms */
struct MjChromaTable {
    MjChromaTerm t[512];
};
MJ_CC MjChromaTable mj_chroma_table() {
    MjChromaTable tab{};
    for (uint32_t v = 0; v < 256; v++) {
        tab.t[v] = mj_cb_term(v);
        tab.t[256 + v] = mj_cr_term(v);
    }
    return tab;
}

MJ_CC int32_t mj_clamp255(int32_t v) {
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

/* one pixel, 0x00BBGGRR */
MJ_CC uint32_t mj_ycc_to_rgb(uint32_t y, uint32_t cb, uint32_t cr, MjChromaTerm tb, MjChromaTerm tr) {
    const int32_t qg = (tb.g + tr.g) >> MJ_G_FRAC_BITS;
    const uint32_t R = (uint32_t) mj_clamp255((int32_t) y + (int16_t) tr.o);
    const uint32_t G = (y == MJ_WRAP_Y && cb == MJ_WRAP_CB && cr == MJ_WRAP_CR) ? 0u : (uint32_t) mj_clamp255((int32_t) y + qg);
    const uint32_t B = (uint32_t) mj_clamp255((int32_t) y + (int16_t) tb.o);
    return R | (G << 8) | (B << 16);
}

#endif /* MJ_COLOUR_CORE_H */
