/*
 * mj_k_idct_colour.hip -- dequantise + 8x8 inverse DCT + plane assembly + chroma
 * upsample + YCbCr->RGB, fused: coefficients are read once from HBM and interleaved
 * RGB24 is written once.
 *
 * Replaces dequant_data_unit (quant_table.c:131-152), fast_2didct / fast_idct_new
 * (dct.c:158-204, :296-341), write_mcu / write_data_unit (decode.c:436-533) and
 * ycbcr_rgb / y_rgb (colour_conversion.c:5-83).
 *
 * Work decomposition: one *wavefront* per tile of consecutive MCUs of one MCU row, at
 * most 24 blocks (4 MCUs of 4:2:0, 64 x 16 pixels).  In scan order those MCUs' blocks are
 * contiguous in the coefficient buffer (luma and chroma of the same pixels sit next to
 * each other), and the entropy stage leaves every block in natural order, so:
 *   1. eight lanes share one 8x8 block, 8 blocks per pass: lane r loads row r (16 bytes;
 *      the eight lanes of a block read one 128-byte line), dequantises it (int16 wrap),
 *      runs the row butterfly, the eight lanes transpose through LDS, lane x runs the
 *      column butterfly, they transpose back, and lane r level-shifts, clamps, truncates
 *      and stores row r of the samples to the wavefront's sample buffer in LDS, sixteen
 *      bits per sample;
 *   2. the wavefront walks its tile in patches of 8 x 2^v pixels, one per lane:
 *      nearest-neighbour chroma lookup, colour transform in packed 16-bit integers,
 *      saturating pack, and 24-byte stores of RGB24.
 * Wavefronts are persistent and independent of each other -- no workgroup barrier after
 * the start -- so at any time some are in the arithmetic of step 1 while others wait
 * for their stores of step 2, and the load on HBM is even.  Every wavefront takes one
 * contiguous run of the batch's tiles (it walks along MCU rows, so the 192-byte pieces it
 * writes to an RGB row complete each other's cache lines, and it keeps the 64-byte record
 * of its image in scalar registers until the run crosses into the next image).  The
 * coefficient rows of a wavefront's next tile are requested as soon as the butterflies
 * have consumed the current ones and arrive during the colour stage.
 *
 * The IDCT arithmetic follows the reference's rounding points exactly: float storage,
 * double constants, no fused multiply-add (the file is compiled with
 * -ffp-contract=off), truncating conversions.  The colour transform is the integer form
 * of mj_colour_core.h, equal to the reference's on all 2^24 triples.  Bound: HBM (2*S
 * bytes of coefficients in + 3 bytes of RGB out per pixel, S = samples per pixel); MFMA
 * is not applicable (no dense contraction at this precision).
 */
#include "mj_colour_core.h"
#include "mj_internal.h"

namespace {

constexpr int TPB = MJ_IDCT_THREADS;
constexpr int WAVES = TPB / 64;                /* independent wavefronts per workgroup */
constexpr int MAXB = MJ_IDCT_TILE_BLOCKS;      /* blocks per tile */
constexpr int PASSES = (MAXB + 7) / 8;         /* eight blocks per pass */
/* The 8x8 transpose buffer of a block: rows ten floats apart, blocks eighty.  With that the
 * eight-byte row accesses of the eight lanes of two neighbouring blocks (one LDS pass of
 * sixteen lanes) fall on sixteen different bank pairs, and so do those of four blocks on
 * the 64-bank read path; the four-byte column accesses of a block are eight consecutive
 * banks, two-way shared between blocks g and g + 2, which a store hides and a load pays
 * with one cycle. */
constexpr int TR_ROW = 10;
constexpr int TR_GROUP = 8 * TR_ROW;
constexpr int TR_BYTES = 8 * TR_GROUP * 4;     /* per wavefront */
constexpr int SMP_BYTES = MAXB * 128 + MJ_IDCT_TILE_MCUS_MAX * 16;   /* per wavefront: 128 bytes per block, 16 between MCUs */
constexpr int QT_COMPS = 3;                    /* images of more components are refused when the batch is built */
constexpr int QT_BYTES = QT_COMPS * 64 * 2;    /* per wavefront: the quantiser tables of its image */
constexpr int WAVE_BYTES = SMP_BYTES + TR_BYTES + QT_BYTES;
static_assert(3 * (WAVES * WAVE_BYTES + 512 * 8) <= 160 * 1024, "three workgroups per compute unit");
constexpr uint32_t BLK_BYTES = 128;            /* a block of samples in LDS: 8 rows of 8 x 16 bits */

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

/* both chroma tables (Cb terms, then Cr terms), copied to LDS by every workgroup */
__device__ const MjChromaTable c_chroma = mj_chroma_table();

/* the one sum of G shares that belongs to the chroma pair whose G wraps (with Y = 219);
 * tests/test_oracle_vs_ref.py checks that no other pair has it */
constexpr int32_t WRAP_GSUM = mj_cb_term(MJ_WRAP_CB).g + mj_cr_term(MJ_WRAP_CR).g;

/* (float) (C * (double) x) -- a product with a double constant, rounded to double and
 * then to float, as the C source's mixed types imply -- without leaving single
 * precision: split C = hi + lo with hi = (float) C and lo = (float) (C - hi); then
 * fma(hi, x, lo * x), two roundings of which the first is far below the result's last
 * bit, gives the same float.  That is a property of the constant, checked for the two
 * mantissas used here (0.707106781 and 1.414213562, and their multiples by powers of
 * two) against the double computation for every one of the 2^23 float mantissas of x;
 * signs and exponents do not matter, and (hi, lo) = (1, 0) leaves x as it is.  Two
 * values at a time, it is two packed instructions instead of six conversions and
 * multiplications. */
struct SplitConst {
    float hi, lo;
};
constexpr SplitConst split_const(double c) {
    return SplitConst{(float) c, (float) (c - (double) (float) c)};
}
__device__ __forceinline__ float mulc(float x, SplitConst c) {
    return __builtin_fmaf(c.hi, x, c.lo * x);
}
__device__ __forceinline__ f2 mulc(f2 x, SplitConst c) {
    const f2 hi = {c.hi, c.hi}, lo = {c.lo, c.lo};
    return __builtin_elementwise_fma(hi, x, lo * x);
}

/* 8-point inverse butterfly (dct.c:296-341).
 *
 * The reference halves after every add ("0.5 * (a + b)").  Multiplying by a power
 * of two is exact and commutes with every rounding, so the halvings are carried as
 * an implicit scale instead (capital letters = 2x or 4x the reference's variable)
 * and folded into the constants: the results are bit-identical and twelve
 * multiplies per butterfly disappear.  Products of one value with one constant go
 * through mulc; the two-term rotations keep the reference's double arithmetic. */
/* D = (A[SA] + B[SB], A[SA] + B[SB]) with the B term of the low (NL) or high (NH) half
 * negated: one packed add for a butterfly's sum and difference, operands taken from
 * whichever half of a register pair they are in. */
#define MJ_SUMDIFF(D, A, SA, B, SB, NL, NH)                                                                              \
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[" #SA "," #SB "] op_sel_hi:[" #SA "," #SB "] neg_lo:[0," #NL "] neg_hi:[0," #NH "]" \
        : "=v"(D)                                                                                                        \
        : "v"(A), "v"(B))

/* In: the eight values as the pairs (d0, d4), (d1, d7), (d2, d6), (d3, d5).  Out: the pairs
 * (d0, d7), (d1, d6), (d2, d5), (d3, d4) of the result -- throughout, values stay in the
 * register pairs in which the packed instructions take and leave them. */
struct Idct8Out {
    f2 o07, o16, o25, o34;
};
template <bool FINAL>
__device__ __forceinline__ Idct8Out idct8(f2 d04, f2 d17, f2 d26, f2 d35) {
    /* FINAL: the caller's next step is 0.25 * x + 128 (dct.c:192); the exact factor
     * 0.25 is folded into the last constant as well */
    constexpr SplitConst C = split_const(FINAL ? 0.25 * 0.707106781 : 0.707106781);
    constexpr double C2 = 2.0 * 0.707106781;
    static_assert(C2 == 1.414213562, "one split serves both constants");
    constexpr SplitConst S2 = split_const(C2);
    /* (t0, t1) = (1.414213562 d0, d4): the second half multiplies by one */
    const f2 t01 = __builtin_elementwise_fma((f2){S2.hi, 1.0f}, d04, (f2){S2.lo, 0.0f} * d04);
    const f2 T56 = mulc(d35, S2);                        /* 2 t5, 2 t6 */
    f2 T47, U01, U46, U57, W03, W12;
    MJ_SUMDIFF(T47, d17, 0, d17, 1, 1, 0);               /* 2 t4 = d1 - d7, 2 t7 = d1 + d7 */
    MJ_SUMDIFF(U01, t01, 0, t01, 1, 0, 1);               /* 2 u0 = t0 + t1, 2 u1 = t0 - t1 */
    const double t2d = (double) d26.x, t3d = (double) d26.y;
    f2 U23;
    U23.x = (float) (C2 * (0.38268343236 * t2d + -0.92387953251 * t3d));   /* 2 u2 */
    U23.y = (float) (C2 * (0.92387953251 * t2d + 0.38268343236 * t3d));    /* 2 u3 */
    MJ_SUMDIFF(U46, T47, 0, T56, 1, 0, 1);               /* 4 u4 = T4 + T6, 4 u6 = T4 - T6 */
    MJ_SUMDIFF(U57, T47, 1, T56, 0, 1, 0);               /* 4 u5 = T7 - T5, 4 u7 = T7 + T5 */

    MJ_SUMDIFF(W03, U01, 0, U23, 1, 0, 1);               /* 4 w0 = U0 + U3, 4 w3 = U0 - U3 */
    MJ_SUMDIFF(W12, U01, 1, U23, 0, 0, 1);               /* 4 w1 = U1 + U2, 4 w2 = U1 - U2 */
    const double u4d = (double) U46.x, u5d = (double) U57.x, u6d = (double) U46.y, u7d = (double) U57.y;
    f2 W47, W56;
    W47.x = (float) (0.8314696123 * u4d + -0.55557023302 * u7d);           /* 4 w4 */
    W56.x = (float) (0.9807852804 * u5d + -0.19509032201 * u6d);
    W56.y = (float) (0.19509032201 * u5d + 0.9807852804 * u6d);
    W47.y = (float) (0.55557023302 * u4d + 0.8314696123 * u7d);

    /* reference: 1.414213562 * 2 * (w + w'); here the sum carries a factor 4 */
    f2 s07, s16, s25, s34;
    MJ_SUMDIFF(s07, W03, 0, W47, 1, 0, 1);
    MJ_SUMDIFF(s16, W12, 0, W56, 1, 0, 1);
    MJ_SUMDIFF(s25, W12, 1, W56, 0, 0, 1);
    MJ_SUMDIFF(s34, W03, 1, W47, 0, 0, 1);
    Idct8Out o;
    o.o07 = mulc(s07, C);
    o.o16 = mulc(s16, C);
    o.o25 = mulc(s25, C);
    o.o34 = mulc(s34, C);
    return o;
}

/* dct.c:187-193 with CLAMP_8 (dct.c:6): clamp(0.25*x + 128) in double, then truncate.
 * q is 0.25*x already (see idct8<true>).  q + 128 is exact in double, and for a sum
 * that is not negative truncation is floor(q) + 128; a negative sum clamps to 0 either
 * way.  floor(q) + 128 is an integer, exact in float wherever it is inside [0, 255], and
 * the conversion instruction saturates to that range, so the double arithmetic reduces,
 * bit for bit, to a floor, an add and a saturating conversion.  Two samples at a time:
 * the result holds them as two 16-bit fields. */
__device__ __forceinline__ uint32_t finish2(float qa, float qb) {
    const f2 v = (f2){floorf(qa), floorf(qb)} + (f2){128.0f, 128.0f};
    return __builtin_amdgcn_cvt_pk_u8_f32(v.y, 2, __builtin_amdgcn_cvt_pk_u8_f32(v.x, 0, 0));
}

/* D.byte[i] = byte sel.byte[i] of the eight bytes {lo: 0-3, hi: 4-7}; 0x0c gives 0x00, 0x0d gives 0xff */
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    return __builtin_amdgcn_perm(hi, lo, sel);
}

/* two signed 16-bit fields, each clamped to 0..255, as the two low bytes of the result */
__device__ __forceinline__ uint32_t sat_pk(uint32_t v) {
    uint32_t d;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(v));
    return d;
}

/* {y.lo + o.lo, y.hi + o.lo}: two luma samples plus the one offset they share */
__device__ __forceinline__ uint32_t add_pair_shared(uint32_t y, uint32_t o) {
    uint32_t d;
    asm("v_pk_add_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(y), "v"(o));
    return d;
}
/* {y.lo + o.lo, y.hi + o.hi} */
__device__ __forceinline__ uint32_t add_pair(uint32_t y, uint32_t o) {
    uint32_t d;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(y), "v"(o));
    return d;
}

/* Four pixels from their clamped channel pairs (r01 = R0 | R1 << 8 in its low half, ...) to
 * `o`: twelve bytes of RGB24 (4-byte aligned), or sixteen of R, G, B, 255. */
struct __attribute__((aligned(4))) Rgb4 {
    uint32_t a, b, c;
};
template <bool RGBA>
__device__ __forceinline__ void store4(uint8_t *o, uint32_t r01, uint32_t g01, uint32_t b01, uint32_t r23, uint32_t g23,
                                       uint32_t b23) {
    const uint32_t gb01 = perm(b01, g01, 0x05040100u);   /* G0 G1 B0 B1 */
    const uint32_t rg23 = perm(g23, r23, 0x05040100u);   /* R2 R3 G2 G3 */
    if (RGBA) {
        *reinterpret_cast<uint4 *>(o) = make_uint4(perm(gb01, r01, 0x0d060400u), perm(gb01, r01, 0x0d070501u),
                                                   perm(rg23, b23, 0x0d000604u), perm(rg23, b23, 0x0d010705u));
    } else {
        Rgb4 v;
        v.a = perm(gb01, r01, 0x01060400u);              /* R0 G0 B0 R1 */
        v.b = perm(gb01, rg23, 0x02000705u);             /* G1 B1 R2 G2 */
        v.c = perm(rg23, b23, 0x01070500u);              /* B2 R3 G3 B3 */
        *reinterpret_cast<Rgb4 *>(o) = v;
    }
}

/* pixel k of four (0x00BBGGRR each) to `o`, a byte at a time: edge pixels and the generic paths */
__device__ __forceinline__ void store_px(uint8_t *o, uint32_t k, uint32_t px, bool rgba) {
    if (rgba) {
        reinterpret_cast<uint32_t *>(o)[k] = px | 0xFF000000u;
    } else {
        o[3 * k] = (uint8_t) px;
        o[3 * k + 1] = (uint8_t) (px >> 8);
        o[3 * k + 2] = (uint8_t) (px >> 16);
    }
}

#ifndef MJ_IDCT_WAVES
#define MJ_IDCT_WAVES 6
#endif
#define MJ_IDCT_WG_PER_CU 3

/* Timing experiments only (-DMJ_IDCT_TRACE): the first thread of every workgroup adds the
 * shader-clock time it spends in each phase of a tile to the batch's statistics words,
 * over what the entropy kernels left there. */
#if defined(MJ_IDCT_TRACE)
#define MJ_STAMP(i) \
    do { \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        acc_[i] += now_ - stamp_; \
        stamp_ = now_; \
    } while (0)
#else
#define MJ_STAMP(i) do { } while (0)
#endif

/* Where the samples of the tile live after the butterflies: block `blk` of MCU `mcu` of
 * the tile (blk counted over the whole tile) keeps its 8x8 samples row-major, sixteen
 * bits each, at blk * 128 + mcu * 16. */
__device__ __forceinline__ uint32_t sample_off(uint32_t mcu, uint32_t blk, uint32_t x, uint32_t y) {
    return __umul24(blk, BLK_BYTES) + mcu * 16u + (y & 7u) * 16u + (x & 7u) * 2u;
}

struct TileGeo {
    uint32_t bpm, s0, s1, s2;          /* blocks per MCU; first block slot of each component inside the MCU */
    uint32_t tile_w, tile_h;           /* pixels */
    uint32_t px0, py0;                 /* the tile's first pixel in the image */
    uint32_t X, Y, pitch;
    bool rgba;
    bool fold, nostore;                /* timing experiments only */
    uint32_t nostore_lane;
};

/*
 * Colour stage for the common layouts: luma at full resolution, both chroma
 * components subsampled by 2^SH horizontally and 2^SV vertically with one block each
 * per MCU (4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1).  A thread converts a patch of
 * 8 x 2^SV pixels per step -- one row of a luma block per pixel row -- so every chroma
 * sample and its two table rows are fetched once per patch.  Per pair of pixels and
 * channel: one packed 16-bit add of the two luma samples and the channel's offset, one
 * saturating pack; byte permutes put four pixels' channels in RGB order.
 */
template <int SH, int SV, bool RGBA>
__device__ __forceinline__ void colour_fast(const uint8_t *s_smp, const MjChromaTerm *s_tab, const TileGeo &g, uint8_t *rgb, uint32_t lane) {
    constexpr int NC = 8 >> SH, ROWS = 1 << SV;          /* chroma samples per patch row; pixel rows per patch */
    constexpr int NCW = (NC + 1) / 2;                    /* words that hold them */
    constexpr uint32_t bpp = RGBA ? 4u : 3u;
    const uint32_t cols = g.tile_w / 8, prow_n = g.tile_h >> SV;
    const uint32_t dy = 64 / cols, dq = 64 - dy * cols;
#if defined(MJ_EXPERIMENT)
    /* timing experiment: every tile's rows land in the same 64 KiB */
    uint8_t *rgb_tile = g.fold ? rgb + ((threadIdx.x >> 6) << 16) : rgb + (uint64_t) g.py0 * g.pitch;
    const uint32_t fold_mask = g.fold ? 0xFFFFu : 0xFFFFFFFFu;
#else
    uint8_t *rgb_tile = rgb + (uint64_t) g.py0 * g.pitch;
    constexpr uint32_t fold_mask = 0xFFFFFFFFu;
#endif
    uint32_t pr = lane / cols;
    uint32_t qx = lane - pr * cols;
    for (; pr < prow_n; pr += dy, qx += dq) {
        if (qx >= cols) {
            qx -= cols;
            pr++;
            if (pr >= prow_n) break;
        }
        const uint32_t x = qx * 8, gx = g.px0 + x;
        const uint32_t y = pr << SV, gy = g.py0 + y;
        if (gy >= g.Y || gx >= g.X) continue;
        /* chroma samples of the patch: chroma pixels (x >> SH ...) of row pr of MCU x >> (3 + SH) */
        const uint32_t cx = x >> SH, mcu = x >> (3 + SH);
        const uint32_t mblk = __umul24(mcu, g.bpm);
        const uint8_t *cbp = s_smp + sample_off(mcu, mblk + g.s1, cx, pr);
        const uint8_t *crp = s_smp + sample_off(mcu, mblk + g.s2, cx, pr);
        uint32_t cbw[NCW], crw[NCW];
        if (NC == 8) {
            const uint4 vb = *reinterpret_cast<const uint4 *>(cbp), vr = *reinterpret_cast<const uint4 *>(crp);
            cbw[0] = vb.x, cbw[1] = vb.y, cbw[2 % NCW] = vb.z, cbw[3 % NCW] = vb.w;
            crw[0] = vr.x, crw[1] = vr.y, crw[2 % NCW] = vr.z, crw[3 % NCW] = vr.w;
        } else if (NC == 4) {
            const uint2 vb = *reinterpret_cast<const uint2 *>(cbp), vr = *reinterpret_cast<const uint2 *>(crp);
            cbw[0] = vb.x, cbw[1 % NCW] = vb.y;
            crw[0] = vr.x, crw[1 % NCW] = vr.y;
        } else {
            cbw[0] = *reinterpret_cast<const uint32_t *>(cbp);
            crw[0] = *reinterpret_cast<const uint32_t *>(crp);
        }
        /* luma block: column (x >> 3) & (2^SH - 1), row ly >> 3 of the MCU's luma blocks */
        const uint32_t lcol = (x >> 3) & ((1u << SH) - 1u);
        uint4 yw[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const uint32_t ly = y + r;
            const uint32_t lblk = mblk + g.s0 + ((ly >> 3) << SH) + lcol;
            yw[r] = *reinterpret_cast<const uint4 *>(s_smp + sample_off(mcu, lblk, 0, ly));
        }
        const bool whole = (gx + 8 <= g.X) && (gy + ROWS <= g.Y);
        /* four pixels of every row at a time: the chroma samples they share, then the rows */
#pragma unroll
        for (int h = 0; h < 2; h++) {
            constexpr int NCH = (NC >= 2) ? NC / 2 : 1;      /* chroma samples per half row */
            uint32_t oR[NCH], oG[NCH], oB[NCH];
            bool wraps = false;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int ci = (NC >= 2) ? h * NCH + i : 0;
                const uint32_t cb = (cbw[ci >> 1] >> (16 * (ci & 1))) & 0xFFFFu, cr = (crw[ci >> 1] >> (16 * (ci & 1))) & 0xFFFFu;
                const MjChromaTerm tb = s_tab[cb], tr = s_tab[256 + cr];
                const int32_t gs = tb.g + tr.g;
                wraps |= (gs == WRAP_GSUM);
                oR[i] = tr.o;
                oG[i] = (uint32_t) (gs >> MJ_G_FRAC_BITS);
                oB[i] = tb.o;
            }
            if (whole && !wraps) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const uint32_t yp[2] = {h ? yw[r].z : yw[r].x, h ? yw[r].w : yw[r].y};   /* the two pixel pairs */
                    uint32_t cr8[2], cg8[2], cb8[2];
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if (SH == 0) {
                            cr8[j] = sat_pk(add_pair(yp[j], perm(oR[(2 * j + 1) % NCH], oR[(2 * j) % NCH], 0x05040100u)));
                            cg8[j] = sat_pk(add_pair(yp[j], perm(oG[(2 * j + 1) % NCH], oG[(2 * j) % NCH], 0x05040100u)));
                            cb8[j] = sat_pk(add_pair(yp[j], perm(oB[(2 * j + 1) % NCH], oB[(2 * j) % NCH], 0x05040100u)));
                        } else {
                            const int i = (SH == 1) ? j : 0;
                            cr8[j] = sat_pk(add_pair_shared(yp[j], oR[i]));
                            cg8[j] = sat_pk(add_pair_shared(yp[j], oG[i]));
                            cb8[j] = sat_pk(add_pair_shared(yp[j], oB[i]));
                        }
                    }
                    /* rows of one tile are at most a few megabytes apart: 32-bit offsets from the
                     * tile's first row */
#if defined(MJ_EXPERIMENT)
                    if (g.nostore && lane != g.nostore_lane) continue;   /* timing experiment: the arithmetic without the stores */
#endif
                    store4<RGBA>(rgb_tile + ((__umul24(y + r, g.pitch) + bpp * (gx + 4 * h)) & fold_mask), cr8[0], cg8[0], cb8[0], cr8[1], cg8[1], cb8[1]);
                }
            } else {
                /* pixels cut off by the image edge, and the one chroma pair for which a luma value
                 * exists that wraps G (colour_conversion.c:78): pixel by pixel */
                for (uint32_t r = 0; r < (uint32_t) ROWS && gy + r < g.Y; r++) {
                    const uint32_t ly = y + r;
                    const uint32_t lblk = mblk + g.s0 + ((ly >> 3) << SH) + lcol;
                    uint8_t *o = rgb_tile + (__umul24(ly, g.pitch) + bpp * (gx + 4 * h));
                    for (uint32_t k = 0; k < 4 && gx + 4 * h + k < g.X; k++) {
                        const uint32_t kk = 4 * h + k;
                        const uint32_t Y = *reinterpret_cast<const uint16_t *>(s_smp + sample_off(mcu, lblk, kk, ly));
                        const uint32_t cb = *reinterpret_cast<const uint16_t *>(cbp + 2 * (kk >> SH));
                        const uint32_t cr = *reinterpret_cast<const uint16_t *>(crp + 2 * (kk >> SH));
                        store_px(o, k, mj_ycc_to_rgb(Y, cb, cr, s_tab[cb], s_tab[256 + cr]), RGBA);
                    }
                }
            }
        }
    }
}

/* one component: R = G = B = the sample (y_rgb, colour_conversion.c:5-28), four pixels per lane and step */
template <bool RGBA>
__device__ __forceinline__ void colour_grey(const uint8_t *s_smp, const TileGeo &g, uint32_t hs0, uint8_t *rgb, uint32_t lane) {
    constexpr uint32_t bpp = RGBA ? 4u : 3u;
    const uint32_t bpm = g.bpm, sb0 = g.s0;
    const uint32_t quads = g.tile_w / 4;
    const uint32_t dy = 64 / quads, dq = 64 - dy * quads;
    uint32_t y = lane / quads;
    uint32_t qx = lane - y * quads;
        for (; y < g.tile_h; y += dy, qx += dq) {
            if (qx >= quads) {
                qx -= quads;
                y++;
                if (y >= g.tile_h) break;
            }
            const uint32_t x = qx * 4, gy = g.py0 + y, gx = g.px0 + x;
            if (gy >= g.Y || gx >= g.X) continue;
            const uint32_t bx = x >> 3, m = bx / hs0, h = bx - m * hs0;
            const uint2 v = *reinterpret_cast<const uint2 *>(s_smp + sample_off(m, m * bpm + sb0 + (y >> 3) * hs0 + h, x, y));
            uint8_t *o = rgb + (uint64_t) gy * g.pitch + bpp * gx;
            if (gx + 4 <= g.X) {
                if (RGBA) {
                    *reinterpret_cast<uint4 *>(o) = make_uint4(perm(0, v.x, 0x0d000000u), perm(0, v.x, 0x0d020202u),
                                                               perm(0, v.y, 0x0d000000u), perm(0, v.y, 0x0d020202u));
                } else {
                    Rgb4 w;
                    w.a = perm(0, v.x, 0x02000000u);         /* Y0 Y0 Y0 Y1 */
                    w.b = perm(v.y, v.x, 0x04040202u);       /* Y1 Y1 Y2 Y2 */
                    w.c = perm(0, v.y, 0x02020200u);         /* Y2 Y3 Y3 Y3 */
                    *reinterpret_cast<Rgb4 *>(o) = w;
                }
            } else {
                const uint32_t yy[4] = {v.x & 0xFFu, (v.x >> 16) & 0xFFu, v.y & 0xFFu, (v.y >> 16) & 0xFFu};
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    if (gx + k < g.X) store_px(o, k, yy[k] * 0x010101u, RGBA);
                }
            }
        }
}

/* the colour stage of one tile, by the layout of its image */
__device__ __forceinline__ void colour_tile(const uint8_t *s_smp, const MjChromaTerm *s_tab, const MjIdctImg &im, const TileGeo &g,
                                            uint8_t *rgb, uint32_t lane) {
    const uint32_t bpm = g.bpm, sb0 = g.s0, sb1 = g.s1, sb2 = g.s2;
    const uint32_t cl = im.cls & 0xFFu, hmax = (im.geo >> 16) & 0xFFu, vmax = im.geo >> 24;
    /* sampling factors of the three components (1 where there is none) */
    const uint32_t hs0 = max(im.sf & 15u, 1u), hs1 = max((im.sf >> 8) & 15u, 1u), hs2 = max((im.sf >> 16) & 15u, 1u);
    const uint32_t vs0 = max((im.sf >> 4) & 15u, 1u), vs1 = max((im.sf >> 12) & 15u, 1u), vs2 = max((im.sf >> 20) & 15u, 1u);
    /* sample (x, y) of component c, in tile coordinates of that component */
    auto smp = [&](uint32_t sb, uint32_t hs, uint32_t x, uint32_t y) -> uint32_t {
        const uint32_t bx = x >> 3, m = bx / hs, h = bx - m * hs;
        return *reinterpret_cast<const uint16_t *>(s_smp + sample_off(m, m * bpm + sb + (y >> 3) * hs + h, x, y));
    };
    const uint32_t bpp = g.rgba ? 4u : 3u;
    /* The generic paths number the tile's quads (4 horizontally adjacent pixels) row-major;
     * a lane takes quad lane, lane + 64, ... and tracks (row, quad-in-row) incrementally,
     * so the loop has no division. */
    const uint32_t quads = g.tile_w / 4;
    const uint32_t dy = 64 / quads, dq = 64 - dy * quads;
    uint32_t y = lane / quads;
    uint32_t qx = lane - y * quads;
    if (cl == MJ_CL_GREY) {
        if (g.rgba) colour_grey<true>(s_smp, g, hs0, rgb, lane);
        else colour_grey<false>(s_smp, g, hs0, rgb, lane);
        return;
    }
#define MJ_FAST(N, SH, SV)                                                \
    case MJ_CL_FAST + N:                                                  \
        if (g.rgba) colour_fast<SH, SV, true>(s_smp, s_tab, g, rgb, lane); \
        else colour_fast<SH, SV, false>(s_smp, s_tab, g, rgb, lane);       \
        return
    switch (cl) {
        MJ_FAST(0, 0, 0);
        MJ_FAST(1, 1, 0);
        MJ_FAST(2, 2, 0);
        MJ_FAST(3, 0, 1);
        MJ_FAST(4, 1, 1);
        MJ_FAST(5, 2, 1);
        default: break;
    }
#undef MJ_FAST
    /* Source sample of output (i, j) in component c is (uint32)(ratio * i) with
     * ratio = (float) sf_c / max_sf (colour_conversion.c:44-45, :62-69).  Sampling
     * factors divide the maximum here, so that is i / (max_sf / sf_c). */
    const uint32_t qh0 = hmax / hs0, qh1 = hmax / hs1, qh2 = hmax / hs2;
    const uint32_t qv0 = vmax / vs0, qv1 = vmax / vs1, qv2 = vmax / vs2;
    for (; y < g.tile_h; y += dy, qx += dq) {
        if (qx >= quads) {
            qx -= quads;
            y++;
            if (y >= g.tile_h) break;
        }
        const uint32_t x = qx * 4, gy = g.py0 + y, gx = g.px0 + x;
        if (gy >= im.Y || gx >= im.X) continue;
        uint8_t *o = rgb + (uint64_t) gy * im.pitch + bpp * gx;
        for (uint32_t k = 0; k < 4 && gx + k < im.X; k++) {
            const uint32_t xx = x + k;
            const uint32_t Y = smp(sb0, hs0, xx / qh0, y / qv0), cb = smp(sb1, hs1, xx / qh1, y / qv1), cr = smp(sb2, hs2, xx / qh2, y / qv2);
            store_px(o, k, mj_ycc_to_rgb(Y, cb, cr, s_tab[cb], s_tab[256 + cr]), g.rgba);
        }
    }
}

/* row r of every block this lane works on in a tile of nblk blocks at src: all loads of a
 * tile are issued at once */
__device__ __forceinline__ void load_rows(const u4 *src, uint32_t nblk, uint32_t lane, u4 (&cv)[PASSES]) {
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
        cv[p] = (u4){0, 0, 0, 0};
        if (p * 8 + (lane >> 3) < nblk) cv[p] = __builtin_nontemporal_load(src + p * 64 + lane);
    }
}

/* the image's record, through the scalar cache */
__device__ __forceinline__ MjIdctImg load_record(const MjDeviceBatch &b, uint32_t ii) {
    return b.idct_img[ii];
}

__device__ __forceinline__ const u4 *tile_rows(const MjDeviceBatch &b, const MjIdctImg &d, uint32_t mrow, uint32_t m0) {
    return reinterpret_cast<const u4 *>(b.coefs + (d.coef_off + ((uint64_t) mrow * d.mcus_x + m0) * (d.geo & 0xFFu)) * 64);
}

__global__ __launch_bounds__(TPB, MJ_IDCT_WAVES) void mj_k_idct_colour(MjDeviceBatch b) {
    /* LDS: the two chroma tables, shared; per wavefront the tile's samples, the transpose
     * buffers of the butterflies and the quantiser tables of the image it is working on */
    __shared__ __align__(16) MjChromaTerm s_tab[512];
    __shared__ __align__(16) uint8_t s_wave[WAVES][WAVE_BYTES];
    static_assert(sizeof(MjChromaTable) == TPB * sizeof(MjChromaTerm), "one table row per thread");
    s_tab[threadIdx.x] = c_chroma.t[threadIdx.x];
    __syncthreads();       /* the only one: from here on every wavefront is on its own */

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *s_smp = s_wave[wave];
    uint8_t *s_tr = s_smp + SMP_BYTES;
    uint16_t (*s_qt)[64] = reinterpret_cast<uint16_t (*)[64]>(s_tr + TR_BYTES);
    /* this wavefront's share: tiles t .. t_end - 1 of the batch, one contiguous run, so
     * that it stays in one image (whose record it holds in scalar registers) and along one
     * MCU row for as long as possible */
    const uint32_t n_waves = gridDim.x * WAVES;
    const uint32_t share = (b.n_idct_tiles + n_waves - 1) / n_waves;
    uint32_t t = (blockIdx.x * WAVES + wave) * share;
    const uint32_t t_end = min(t + share, b.n_idct_tiles);
    if (t >= t_end) return;
    /* the image of tile t: the last one whose first tile is not past t (a bisection: a walk
     * over the records costs a dependent load apiece, and a batch may hold thousands) */
    uint32_t ii = 0;
    for (uint32_t hi = b.n_img - 1; ii < hi;) {
        const uint32_t mid = (ii + hi + 1) >> 1;
        if (b.idct_img[mid].first_tile <= t) {
            ii = mid;
        } else {
            hi = mid - 1;
        }
    }
    MjIdctImg d = load_record(b, ii);
    uint32_t mrow = (t - d.first_tile) / d.tiles_x;
    uint32_t m0 = ((t - d.first_tile) - mrow * d.tiles_x) * ((d.geo >> 8) & 0xFFu);
    u4 cv[PASSES];
    load_rows(tile_rows(b, d, mrow, m0), min((d.geo >> 8) & 0xFFu, d.mcus_x - m0) * (d.geo & 0xFFu), threadIdx.x & 63u, cv);
    uint32_t qt_of = 0xFFFFFFFFu;                        /* image whose quantiser tables are in s_qt */

#if defined(MJ_IDCT_TRACE)
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime(), acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        /* Per-lane values are derived again for every tile from a lane index the compiler
         * cannot see through: hoisted out of the loop they would be held in registers (or
         * spilled) across all of it, and they cost a handful of instructions. */
        uint32_t lane = threadIdx.x & 63u;
        asm volatile("" : "+v"(lane));
        const uint32_t r = lane & 7u;                        /* the lane's row, then column, then row again */
        const uint32_t grp = lane >> 3;                      /* block within the pass */
        float *tr = reinterpret_cast<float *>(s_tr) + grp * TR_GROUP;
        /* the reference scales the first row and the first column by 1/sqrt(2) before the
         * butterflies (dct.c:164-167), the corner twice: lane 0 scales its whole row, the
         * others multiply by one, which changes nothing */
        constexpr SplitConst RS = split_const(0.707106781);
        const SplitConst row_scale = {(r == 0) ? RS.hi : 1.0f, (r == 0) ? RS.lo : 0.0f};
        /* where element j of a row sits after the row butterfly: 0 7 1 6 2 5 3 4 hold positions 0..7 */
        const uint32_t tpos = (r < 4) ? 2 * r : 15 - 2 * r;
        const MjIdctImg im = d;                              /* the current tile's image (d moves on to the next tile's below) */
        const uint32_t cur_ii = ii, cur_mrow = mrow, cur_m0 = m0;
        const uint32_t bpm = im.geo & 0xFFu, tile_mcus = (im.geo >> 8) & 0xFFu;
        const uint32_t nm = min(tile_mcus, im.mcus_x - cur_m0), nblk = nm * bpm;
        const uint32_t slot_comp = im.slot_comp;
        if (qt_of != cur_ii) {
            const MjImageDev &full = b.img[cur_ii];
            for (uint32_t i = lane; i < QT_COMPS * 64; i += 64) s_qt[i >> 6][i & 63] = full.qt[i >> 6][i & 63];
            qt_of = cur_ii;
        }
        /* a wavefront runs its LDS operations in order: what the colour stage of the tile
         * before still reads is read before anything below is written */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        MJ_STAMP(0);
#if defined(MJ_IDCT_TRACE)
        __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0): the tile's rows, and the stores of the tile before */
        MJ_STAMP(1);
#endif

        /* ---- 1. eight lanes per block, eight blocks per pass -------------------- */
        if (!(MJ_DBG(b) & 16u)) {
            const float rcp_bpm = 1.0f / (float) bpm;
#pragma unroll
            for (int p = 0; p < PASSES; p++) {
                const uint32_t blk = p * 8 + grp;
                if (blk >= nblk) continue;                       /* whole groups of eight lanes drop out together */
                /* blk + 0.5 over bpm is at least 0.05 away from an integer: the rounding of the
                 * reciprocal and of the product cannot reach one */
                const uint32_t mcu = (uint32_t) (((float) blk + 0.5f) * rcp_bpm);
                const uint32_t slot = blk - mcu * bpm;
                const uint32_t comp = (slot_comp >> (2 * slot)) & 3u;
                float c[8];
                {
                    /* int16 wrap of the product (quant_table.c:137): the low halves of the 16-bit products */
                    const us8 q = *reinterpret_cast<const us8 *>(&s_qt[comp][8 * r]);
                    const us8 z = __builtin_bit_cast(us8, cv[p]);
                    const us8 zq = z * q;
#pragma unroll
                    for (int j = 0; j < 8; j++) c[j] = (float) (int16_t) zq[j];
                }
                c[0] = mulc(c[0], RS);                           /* first column */
                /* first row, two values at a time, in the pairs the butterfly takes */
                const f2 c04 = mulc((f2){c[0], c[4]}, row_scale), c17 = mulc((f2){c[1], c[7]}, row_scale);
                const f2 c26 = mulc((f2){c[2], c[6]}, row_scale), c35 = mulc((f2){c[3], c[5]}, row_scale);
                const Idct8Out ro = idct8<false>(c04, c17, c26, c35);
                /* rows -> columns (the reference transposes, runs rows again, and transposes
                 * back on the store: dct.c:173-193).  A row is stored in the order the
                 * butterfly leaves it: 0 7 1 6 2 5 3 4. */
                *reinterpret_cast<f2 *>(tr + r * TR_ROW) = ro.o07;
                *reinterpret_cast<f2 *>(tr + r * TR_ROW + 2) = ro.o16;
                *reinterpret_cast<f2 *>(tr + r * TR_ROW + 4) = ro.o25;
                *reinterpret_cast<f2 *>(tr + r * TR_ROW + 6) = ro.o34;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float *col = tr + tpos;                    /* element r of every row */
                const Idct8Out co = idct8<true>((f2){col[0], col[4 * TR_ROW]}, (f2){col[1 * TR_ROW], col[7 * TR_ROW]},
                                                (f2){col[2 * TR_ROW], col[6 * TR_ROW]}, (f2){col[3 * TR_ROW], col[5 * TR_ROW]});
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                tr[0 * TR_ROW + r] = co.o07.x;
                tr[7 * TR_ROW + r] = co.o07.y;
                tr[1 * TR_ROW + r] = co.o16.x;
                tr[6 * TR_ROW + r] = co.o16.y;
                tr[2 * TR_ROW + r] = co.o25.x;
                tr[5 * TR_ROW + r] = co.o25.y;
                tr[3 * TR_ROW + r] = co.o34.x;
                tr[4 * TR_ROW + r] = co.o34.y;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const f2 *row = reinterpret_cast<const f2 *>(tr + r * TR_ROW);
                const f2 s01 = row[0], s23 = row[1], s45 = row[2], s67 = row[3];
                /* row r of the samples */
                *reinterpret_cast<uint4 *>(s_smp + sample_off(mcu, blk, 0, r)) =
                    make_uint4(finish2(s01.x, s01.y), finish2(s23.x, s23.y), finish2(s45.x, s45.y), finish2(s67.x, s67.y));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        MJ_STAMP(2);
        /* the next tile's rows, on their way during the colour stage */
        const bool more = ++t < t_end;
        if (more) {
            m0 += tile_mcus;
            if (m0 >= im.mcus_x) {
                m0 = 0;
                mrow++;
                if (t >= d.first_tile + d.n_tiles) {
                    d = load_record(b, ++ii);
                    mrow = 0;
                }
            }
            load_rows(tile_rows(b, d, mrow, m0), min((d.geo >> 8) & 0xFFu, d.mcus_x - m0) * (d.geo & 0xFFu), lane, cv);
        }
        MJ_STAMP(3);

        /* ---- optional: the padded component planes (Image.buf, decode.c:517-531) */
        if (b.want_planes) {
            const MjImageDev &full = b.img[cur_ii];
            for (uint32_t i = lane; i < nblk * 8; i += 64) {
                const uint32_t blk = i >> 3, row = i & 7u;
                const uint32_t m = blk / bpm, slot = blk - m * bpm;
                const uint32_t cc = full.blk_comp[slot];
                const uint32_t hs = full.comp[cc].hsf, vs = full.comp[cc].vsf;
                const uint64_t py = (uint64_t) cur_mrow * vs * 8 + full.blk_v[slot] * 8u + row;
                const uint64_t px = ((uint64_t) (cur_m0 + m) * hs + full.blk_h[slot]) * 8;
                const uint4 v = *reinterpret_cast<const uint4 *>(s_smp + sample_off(m, blk, 0, row));
                *reinterpret_cast<uint2 *>(b.planes + full.comp[cc].plane_off + py * full.comp[cc].pw + px) =
                    make_uint2(perm(v.y, v.x, 0x06040200u), perm(v.w, v.z, 0x06040200u));
            }
        }

        /* ---- 2. upsample + colour ---------------------------------------------- */
        if (!(MJ_DBG(b) & 8u)) {
            const uint32_t hmax = (im.geo >> 16) & 0xFFu, vmax = im.geo >> 24;
            TileGeo g;
            g.bpm = bpm;
            g.s0 = (im.cls >> 8) & 0xFFu;
            g.s1 = (im.cls >> 16) & 0xFFu;
            g.s2 = im.cls >> 24;
            g.tile_w = nm * hmax * 8;
            g.tile_h = vmax * 8;
            g.px0 = cur_m0 * hmax * 8;
            g.py0 = cur_mrow * g.tile_h;
            g.X = im.X;
            g.Y = im.Y;
            g.pitch = im.pitch;
            g.rgba = b.rgba != 0;
            g.fold = (MJ_DBG(b) & 32u) != 0;
            g.nostore = (MJ_DBG(b) & 64u) != 0;
            g.nostore_lane = 64u + (b.n_img >> 30);
            colour_tile(s_smp, s_tab, im, g, b.rgb + im.rgb_off, lane);
        }
        MJ_STAMP(4);
#if defined(MJ_IDCT_TRACE)
        acc_[7]++;
#endif
        if (!more) break;
    }
#if defined(MJ_IDCT_TRACE)
    if ((threadIdx.x & 63u) == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&b.stats[i], acc_[i]);
    }
#endif
}

/* ------------------------------------------------------------------------ */
/* planes -> RGB                                                            */
/* ------------------------------------------------------------------------ */
/* The upsample + colour step on its own (colour_conversion.c:30-83: nearest neighbour, source
 * index = (uint32)(ratio * i) with ratio = (float) sf / max_sf; :5-28 for one component), from
 * planar samples to RGB.  Two callers: frames coded in one scan per component, whose scans were
 * decoded as grey images with their planes kept, and ycbcr_rgb / y_rgb on planes the caller
 * filled.  A lane takes four neighbouring pixels of a row and stores them as twelve (RGBA:
 * sixteen) bytes; a component at full horizontal resolution is read four samples at a time. */
constexpr uint32_t PLANES_TPB = 256;

template <bool RGBA>
__global__ __launch_bounds__(PLANES_TPB) void mj_k_planes_rgb(const MjCombine *comb, const uint8_t *planes, uint8_t *rgb) {
    __shared__ MjChromaTerm s_tab[512];
    s_tab[threadIdx.x] = c_chroma.t[threadIdx.x];
    s_tab[256 + threadIdx.x] = c_chroma.t[256 + threadIdx.x];
    __syncthreads();
    const MjCombine &cb = comb[blockIdx.z];
    const uint32_t x0 = (blockIdx.x * PLANES_TPB + threadIdx.x) * 4u, y = blockIdx.y;
    if (x0 >= cb.X || y >= cb.Y) return;
    const uint32_t ncs = cb.ncs;
    uint32_t v[3][4];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if ((uint32_t) c >= ncs) {
            v[c][0] = v[c][1] = v[c][2] = v[c][3] = 128u;
            continue;
        }
        const float hr = (float) cb.hsf[c] / (float) cb.hmax, vr = (float) cb.vsf[c] / (float) cb.vmax;
        const uint32_t sy = (uint32_t) ((float) y * vr);
        const uint8_t *row = planes + cb.plane_off[c] + (uint64_t) sy * cb.pw[c];
        if (cb.hsf[c] == cb.hmax && ((cb.plane_off[c] | cb.pw[c]) & 3u) == 0 && x0 + 4 <= cb.pw[c]) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(row + x0);
            v[c][0] = w & 0xFFu;
            v[c][1] = (w >> 8) & 0xFFu;
            v[c][2] = (w >> 16) & 0xFFu;
            v[c][3] = w >> 24;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t x = x0 + k < cb.X ? x0 + k : cb.X - 1;
                v[c][k] = row[(uint32_t) (hr * (float) x)];
            }
        }
    }
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        px[k] = ncs == 1 ? v[0][k] * 0x010101u : mj_ycc_to_rgb(v[0][k], v[1][k], v[2][k], s_tab[v[1][k]], s_tab[256 + v[2][k]]);
    }
    uint8_t *o = rgb + cb.rgb_off + (uint64_t) y * cb.pitch;
    const bool aligned = ((cb.rgb_off | cb.pitch) & 3u) == 0;
    if (x0 + 4 <= cb.X && (RGBA || aligned)) {
        if (RGBA) {
            *reinterpret_cast<uint4 *>(o + 16u * (x0 >> 2)) =
                make_uint4(px[0] | 0xFF000000u, px[1] | 0xFF000000u, px[2] | 0xFF000000u, px[3] | 0xFF000000u);
        } else {
            Rgb4 w;
            w.a = px[0] | (px[1] << 24);            /* R0 G0 B0 R1 */
            w.b = (px[1] >> 8) | (px[2] << 16);     /* G1 B1 R2 G2 */
            w.c = (px[2] >> 16) | (px[3] << 8);     /* B2 R3 G3 B3 */
            *reinterpret_cast<Rgb4 *>(o + 3u * x0) = w;
        }
    } else {
        for (uint32_t k = 0; k < 4 && x0 + k < cb.X; k++) store_px(o, x0 + k, px[k], RGBA);
    }
}

}  // namespace

hipError_t mj_launch_idct_colour(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_idct_tiles == 0) return hipSuccess;
    /* as many workgroups as the chip holds at once (three per compute unit, by their LDS) */
    uint32_t cus = 0;
    const hipError_t e = mj_device_cus(&cus);
    if (e != hipSuccess) return e;
    const uint32_t resident = cus * MJ_IDCT_WG_PER_CU;
    const uint32_t need = (b.n_idct_tiles + WAVES - 1) / WAVES;
    hipLaunchKernelGGL(mj_k_idct_colour, dim3(need < resident ? need : resident), dim3(TPB), 0, s, b);
    return hipGetLastError();
}

hipError_t mj_launch_planes_rgb(const MjCombine *comb, uint32_t n, uint32_t max_x, uint32_t max_y, const uint8_t *planes, uint8_t *rgb,
                                bool rgba, hipStream_t s) {
    if (n == 0 || max_x == 0 || max_y == 0) return hipSuccess;
    if (n > 65535u || max_y > 65535u) return hipErrorInvalidValue;      /* the grid's y and z */
    const dim3 grid((max_x + 4 * PLANES_TPB - 1) / (4 * PLANES_TPB), max_y, n);
    if (rgba) {
        hipLaunchKernelGGL(mj_k_planes_rgb<true>, grid, dim3(PLANES_TPB), 0, s, comb, planes, rgb);
    } else {
        hipLaunchKernelGGL(mj_k_planes_rgb<false>, grid, dim3(PLANES_TPB), 0, s, comb, planes, rgb);
    }
    return hipGetLastError();
}

hipError_t mj_launch_combine(const MjDeviceBatch &b, hipStream_t s) {
    return mj_launch_planes_rgb(b.comb, b.n_comb, b.comb_max_x, b.comb_max_y, b.planes, b.rgb, b.rgba != 0, s);
}
