/*
 * mj_k_idct_colour.hip -- dequantise + 8x8 inverse DCT + plane assembly + chroma
 * upsample + YCbCr->RGB, fused: coefficients are read once from HBM and interleaved
 * RGB24 is written once.
 *
 * Replaces dequant_data_unit (quant_table.c:131-152), fast_2didct / fast_idct_new
 * (dct.c:158-204, :296-341), write_mcu / write_data_unit (decode.c:436-533) and
 * ycbcr_rgb / y_rgb (colour_conversion.c:5-83).
 *
 * Work decomposition: one workgroup per *tile* of T consecutive MCUs of one MCU
 * row.  In scan order those MCUs' blocks are contiguous in the coefficient buffer
 * (luma and chroma of the same pixels sit next to each other), so the tile is one
 * contiguous run of T*bpm*128 bytes:
 *   1. the workgroup copies it to LDS with 16-byte coalesced loads;
 *   2. eight lanes share one 8x8 block, 64 blocks per pass: lane r de-zigzags and
 *      dequantises row r (int16 wrap), runs the row butterfly, the eight lanes
 *      transpose through LDS, lane x runs the column butterfly, they transpose back,
 *      and lane r level-shifts, clamps, truncates and stores row r of the samples
 *      over the block's own coefficients.  A lane holds eight values, not sixty-four:
 *      the kernel runs at six wavefronts per SIMD instead of three, which is what
 *      keeps HBM busy while other wavefronts are in the butterflies;
 *   3. the workgroup walks the tile's pixel rows: nearest-neighbour chroma lookup,
 *      colour transform, clamp, and 12-byte (4-pixel) coalesced stores of RGB24.
 *
 * Arithmetic follows the reference's rounding points exactly: float storage, double
 * constants, no fused multiply-add (the file is compiled with -ffp-contract=off),
 * truncating conversions.  Bound: HBM (2*S bytes of coefficients in + 3 bytes of RGB
 * out per pixel, S = samples per pixel); MFMA is not applicable (no dense
 * contraction at this precision).
 */
#include "mj_internal.h"

namespace {

constexpr int TPB = MJ_IDCT_THREADS;
constexpr int MAXB = MJ_IDCT_MAX_BLOCKS;
constexpr int COEF_STRIDE = 136;               /* bytes per block in LDS: 128 + 8, spreads the 2-byte de-zigzag reads over the banks */
constexpr int TR_GROUP = 72;                   /* floats per 8x8 transpose buffer: 64 + 8, conflict-free strided access */
constexpr int TR_BYTES = (TPB / 8) * TR_GROUP * 4;

/* zigzag position of natural-order coefficient n (inverse of quant_table.c:25-34) */
__constant__ uint8_t c_zzinv[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                    41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                    46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

typedef float f2 __attribute__((ext_vector_type(2)));

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
template <bool FINAL>
__device__ __forceinline__ void idct8(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5, float &d6,
                                      float &d7) {
    /* FINAL: the caller's next step is 0.25 * x + 128 (dct.c:192); the exact factor
     * 0.25 is folded into the last constant as well */
    constexpr SplitConst C = split_const(FINAL ? 0.25 * 0.707106781 : 0.707106781);
    constexpr double C2 = 2.0 * 0.707106781;
    static_assert(C2 == 1.414213562, "one split serves both constants");
    constexpr SplitConst S2 = split_const(C2);
    const float t0 = mulc(d0, S2);
    const float t1 = d4, t2 = d2, t3 = d6;
    const f2 T56 = mulc((f2){d3, d5}, S2);               /* 2 t5, 2 t6 */
    const float T4 = d1 - d7;                            /* 2 t4 */
    const float T5 = T56.x, T6 = T56.y;
    const float T7 = d1 + d7;                            /* 2 t7 */

    const float U0 = t0 + t1;                            /* 2 u0 */
    const float U1 = t0 - t1;                            /* 2 u1 */
    const double t2d = (double) t2, t3d = (double) t3;
    const float U2 = (float) (C2 * (0.38268343236 * t2d + -0.92387953251 * t3d));   /* 2 u2 */
    const float U3 = (float) (C2 * (0.92387953251 * t2d + 0.38268343236 * t3d));    /* 2 u3 */
    const float U4 = T4 + T6;                            /* 4 u4 */
    const float U5 = T7 - T5;                            /* 4 u5 */
    const float U6 = T4 - T6;                            /* 4 u6 */
    const float U7 = T5 + T7;                            /* 4 u7 */

    const float W0 = U0 + U3;                            /* 4 w0 */
    const float W1 = U1 + U2;
    const float W2 = U1 - U2;
    const float W3 = U0 - U3;
    const double u4d = (double) U4, u5d = (double) U5, u6d = (double) U6, u7d = (double) U7;
    const float W4 = (float) (0.8314696123 * u4d + -0.55557023302 * u7d);           /* 4 w4 */
    const float W5 = (float) (0.9807852804 * u5d + -0.19509032201 * u6d);
    const float W6 = (float) (0.19509032201 * u5d + 0.9807852804 * u6d);
    const float W7 = (float) (0.55557023302 * u4d + 0.8314696123 * u7d);

    /* reference: 1.414213562 * 2 * (w + w'); here the sum carries a factor 4 */
    const f2 o07 = mulc((f2){W0, W0} + (f2){W7, -W7}, C);
    const f2 o16 = mulc((f2){W1, W1} + (f2){W6, -W6}, C);
    const f2 o25 = mulc((f2){W2, W2} + (f2){W5, -W5}, C);
    const f2 o34 = mulc((f2){W3, W3} + (f2){W4, -W4}, C);
    d0 = o07.x;
    d1 = o16.x;
    d2 = o25.x;
    d3 = o34.x;
    d4 = o34.y;
    d5 = o25.y;
    d6 = o16.y;
    d7 = o07.y;
}

/* dct.c:187-193 with CLAMP_8 (dct.c:6): clamp(0.25*x + 128) in double, then truncate.
 * q is 0.25*x already (see idct8<true>).  q + 128 is exact in double, and for a sum
 * that is not negative truncation is floor(q) + 128; a negative sum clamps to 0 either
 * way.  floor(q) + 128 is an integer, exact in float wherever it is inside [0, 255], and
 * the conversion instruction saturates to that range, so the double arithmetic reduces,
 * bit for bit, to a floor, an add and a saturating conversion -- which also places the
 * byte: the result is `word` with byte `pos` replaced by the sample. */
__device__ __forceinline__ uint32_t finish(float q, uint32_t pos, uint32_t word) {
    return __builtin_amdgcn_cvt_pk_u8_f32(floorf(q) + 128.0f, pos, word);
}

/* colour_conversion.c:76-79 as compiled by gcc 11.4 -O2: below zero -> 0, above
 * 256 -> 255, otherwise truncate through a 32-bit integer (so exactly 256.0f, which
 * one (Y,Cb,Cr) triple reaches in G, wraps to 0).  Truncation of a value that is not
 * negative is floor, and the conversion saturates; the result replaces byte `pos` of
 * `word`. */
__device__ __forceinline__ uint32_t store8(float v, uint32_t pos, uint32_t word) {
    const uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(floorf(v), pos, word);
    return (v == 256.0f) ? (word & ~(0xFFu << (8 * pos))) : r;
}

/* colour_conversion.c:71-74, per chroma sample value, from per-workgroup LDS tables:
 *   cb -> g = 0.34414 * (cb - 128) in double, o = floor(1.772 * (cb - 128))
 *   cr -> g = 0.71414 * (cr - 128) in double, o = floor(1.402 * (cr - 128))
 * G keeps the reference's arithmetic: (y - g_cb) - g_cr in double, left to right,
 * rounded to float, truncated.  R = y + 1.402 (cr - 128) and B = y + 1.772 (cb - 128)
 * are an integer plus a multiple of 1/500 resp. 1/250: such a sum is never close enough
 * to an integer from below for the double or the float rounding to reach it, so
 * truncating the rounded sum equals y + floor(product) for every non-negative result,
 * and negative results clamp to 0 either way.  (Checked against the compiled reference
 * for all 65,536 (y, chroma) pairs per channel.) */
struct __attribute__((aligned(16))) ChromaRow {
    double g;
    int32_t o;
    int32_t pad_;
};

__device__ __forceinline__ int32_t floor_div(int32_t num, int32_t den) {   /* den > 0 */
    const int32_t q = num / den;
    return q - ((num % den != 0 && num < 0) ? 1 : 0);
}

__device__ __forceinline__ uint32_t ycc_to_rgb24(uint32_t Y, const ChromaRow tcb, const ChromaRow tcr) {
    const double y = (double) Y;
    const uint32_t R = (uint32_t) min(max((int32_t) Y + tcr.o, 0), 255);
    const float G = (float) ((y - tcb.g) - tcr.g);
    const uint32_t B = (uint32_t) min(max((int32_t) Y + tcb.o, 0), 255);
    return store8(G, 1, R | (B << 16));
}

/* 12-byte store of four RGB24 pixels (4-byte aligned) */
struct __attribute__((aligned(4))) Rgb4 {
    uint32_t a, b, c;
};

/* four pixels (0x00BBGGRR each) to `o`: twelve bytes of RGB24, or sixteen of R, G, B, 255 */
__device__ __forceinline__ void store_quad(uint8_t *o, const uint32_t px[4], uint32_t gx, uint32_t X, bool rgba) {
    if (rgba) {
        if (gx + 4 <= X) {
            *reinterpret_cast<uint4 *>(o) = make_uint4(px[0] | 0xFF000000u, px[1] | 0xFF000000u, px[2] | 0xFF000000u, px[3] | 0xFF000000u);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                if (gx + k < X) reinterpret_cast<uint32_t *>(o)[k] = px[k] | 0xFF000000u;
            }
        }
        return;
    }
    if (gx + 4 <= X) {
        Rgb4 v;
        v.a = px[0] | (px[1] << 24);
        v.b = (px[1] >> 8) | (px[2] << 16);
        v.c = (px[2] >> 16) | (px[3] << 8);
        *reinterpret_cast<Rgb4 *>(o) = v;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {      /* constant indices: px stays in registers */
            if (gx + k < X) {
                o[3 * k] = (uint8_t) px[k];
                o[3 * k + 1] = (uint8_t) (px[k] >> 8);
                o[3 * k + 2] = (uint8_t) (px[k] >> 16);
            }
        }
    }
}

#ifndef MJ_IDCT_WAVES
#define MJ_IDCT_WAVES 6
#endif

/* Where the samples of the tile live after the butterflies: block `blk` of the tile
 * keeps its 8x8 bytes, row-major, in the first 64 bytes of its LDS slot. */
__device__ __forceinline__ uint32_t sample_off(uint32_t blk, uint32_t x, uint32_t y) {
    return __umul24(blk, COEF_STRIDE) + (y & 7u) * 8u + (x & 7u);
}

/*
 * Colour stage for the common layouts: luma at full resolution, both chroma
 * components subsampled by 2^SH horizontally and 2^SV vertically with one block each
 * per MCU (4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1).  A thread converts a patch of
 * 8 x 2^SV pixels per step -- one row of a luma block per pixel row -- so every chroma
 * sample and its two table rows are fetched once per patch and the address arithmetic
 * is paid once per eight or sixteen pixels.  s0, s1, s2: first block slot of each
 * component inside the MCU.
 */
template <int SH, int SV>
__device__ __forceinline__ void colour_fast(const uint8_t *s_smp, const ChromaRow *s_tcb, const ChromaRow *s_tcr, uint32_t bpm,
                                            uint32_t s0, uint32_t s1, uint32_t s2, uint32_t tile_w, uint32_t tile_h,
                                            uint32_t px0, uint32_t py0, uint32_t X, uint32_t Y, uint32_t pitch,
                                            uint8_t *rgb, bool rgba) {
    const uint32_t bpp = rgba ? 4u : 3u;
    constexpr int NC = 8 >> SH, HC = 4 >> SH, ROWS = 1 << SV;   /* chroma samples per patch row, per half of it */
    const uint32_t cols = tile_w / 8, prow_n = tile_h >> SV;
    const uint32_t dy = TPB / cols, dq = TPB - dy * cols;
    uint8_t *rgb_tile = rgb + (uint64_t) py0 * pitch;
    uint32_t pr = threadIdx.x / cols;
    uint32_t qx = threadIdx.x - pr * cols;
    for (; pr < prow_n; pr += dy, qx += dq) {
        if (qx >= cols) {
            qx -= cols;
            pr++;
            if (pr >= prow_n) break;
        }
        const uint32_t x = qx * 8, gx = px0 + x;
        const uint32_t y = pr << SV, gy = py0 + y;
        if (gy >= Y || gx >= X) continue;
        /* chroma samples of the patch: chroma pixels (x >> SH ...) of row pr of MCU x >> (3 + SH) */
        const uint32_t cx = x >> SH, mcu = x >> (3 + SH);
        const uint32_t mblk = __umul24(mcu, bpm);
        const uint8_t *cbp = s_smp + sample_off(mblk + s1, cx, pr);
        const uint8_t *crp = s_smp + sample_off(mblk + s2, cx, pr);
        uint32_t cbw[2] = {0, 0}, crw[2] = {0, 0};
        if (NC == 8) {
            const uint2 vb = *reinterpret_cast<const uint2 *>(cbp), vr = *reinterpret_cast<const uint2 *>(crp);
            cbw[0] = vb.x;
            cbw[1] = vb.y;
            crw[0] = vr.x;
            crw[1] = vr.y;
        } else if (NC == 4) {
            const uint32_t vb = *reinterpret_cast<const uint32_t *>(cbp), vr = *reinterpret_cast<const uint32_t *>(crp);
            cbw[0] = vb & 0xFFFFu;
            cbw[1] = vb >> 16;
            crw[0] = vr & 0xFFFFu;
            crw[1] = vr >> 16;
        } else {
            const uint32_t vb = *reinterpret_cast<const uint16_t *>(cbp), vr = *reinterpret_cast<const uint16_t *>(crp);
            cbw[0] = vb & 0xFFu;
            cbw[1] = vb >> 8;
            crw[0] = vr & 0xFFu;
            crw[1] = vr >> 8;
        }
        /* luma block: column (x >> 3) & (2^SH - 1), row r >> 3 of the MCU's luma blocks */
        const uint32_t lcol = (x >> 3) & ((1u << SH) - 1u);
        uint2 yw[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const uint32_t ly = y + r;
            const uint32_t lblk = mblk + s0 + ((ly >> 3) << SH) + lcol;
            yw[r] = *reinterpret_cast<const uint2 *>(s_smp + sample_off(lblk, 0, ly));
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {                   /* four pixels of every row at a time */
            if (gx + 4 * h >= X) break;
            ChromaRow tb[HC], tr[HC];
#pragma unroll
            for (int i = 0; i < HC; i++) {
                tb[i] = s_tcb[(cbw[h] >> (8 * i)) & 0xFFu];
                tr[i] = s_tcr[(crw[h] >> (8 * i)) & 0xFFu];
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                if (gy + r >= Y) break;
                const uint32_t w = h ? yw[r].y : yw[r].x;
                uint32_t px[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    px[k] = ycc_to_rgb24((w >> (8 * k)) & 0xFFu, tb[k >> SH], tr[k >> SH]);
                }
                /* rows of one tile are at most a few megabytes apart: 32-bit offsets from the
                 * tile's first row */
                store_quad(rgb_tile + (__umul24(y + r, pitch) + bpp * (gx + 4 * h)), px, gx + 4 * h, X, rgba);
            }
        }
    }
}

__global__ __launch_bounds__(TPB, MJ_IDCT_WAVES) void mj_k_idct_colour(MjDeviceBatch b) {
    /* LDS: the tile's blocks (coefficients, then -- in place -- samples); the transpose
     * buffers of the butterflies, which later hold the two chroma product tables; the
     * quantiser tables */
    __shared__ __align__(16) uint8_t s_blk[MAXB * COEF_STRIDE];
    __shared__ __align__(16) uint8_t s_tr[TR_BYTES];
    __shared__ __align__(16) uint16_t s_qt[MJ_MAX_COMP][64];
    ChromaRow *s_tcb = reinterpret_cast<ChromaRow *>(s_tr);
    ChromaRow *s_tcr = s_tcb + 256;
    static_assert(2 * 256 * sizeof(ChromaRow) <= TR_BYTES, "chroma tables reuse the transpose buffers");

    const uint32_t ii = b.idct_wg_img[blockIdx.x];
    const MjImageDev &im = b.img[ii];
    const uint32_t bpm = im.bpm;
    const uint32_t tiles_x = mj_idct_tiles_x(im.mcus_x, bpm, im.hmax);
    const uint32_t T = mj_idct_tile_mcus(im.mcus_x, bpm, im.hmax);
    const uint32_t wg = blockIdx.x - b.idct_wg_first[ii];
    const uint32_t mrow = wg / tiles_x;
    const uint32_t m0 = (wg % tiles_x) * T;
    const uint32_t nm = min(T, im.mcus_x - m0);     /* MCUs in this tile */
    const uint32_t nblk = nm * bpm;

    /* ---- 1. coefficients of the tile -> LDS (coalesced 16-byte loads) ------ */
    const uint4 *src = reinterpret_cast<const uint4 *>(b.coefs + (im.coef_off + ((uint64_t) mrow * im.mcus_x + m0) * bpm) * 64);
    for (uint32_t i = threadIdx.x; i < nblk * 8; i += TPB) {
        const uint4 v = src[i];
        uint2 *d = reinterpret_cast<uint2 *>(s_blk + (i >> 3) * COEF_STRIDE + (i & 7) * 16);
        d[0] = make_uint2(v.x, v.y);
        d[1] = make_uint2(v.z, v.w);
    }
    for (uint32_t i = threadIdx.x; i < im.ncs * 64; i += TPB) {
        s_qt[i >> 6][i & 63] = im.qt[i >> 6][i & 63];
    }
    __syncthreads();

    /* ---- 2. eight lanes per block, 64 blocks per pass ---------------------- */
    if (!(MJ_DBG(b) & 16u)) {
        const uint32_t r = threadIdx.x & 7u;                 /* the lane's row, then column, then row again */
        const uint32_t grp = threadIdx.x >> 3;               /* block within the pass */
        float *tr = reinterpret_cast<float *>(s_tr) + grp * TR_GROUP;
        /* zigzag positions of the coefficients of row r */
        const uint2 zr = reinterpret_cast<const uint2 *>(c_zzinv)[r];
        /* the reference scales the first row and the first column by 1/sqrt(2) before the
         * butterflies (dct.c:164-167), the corner twice: lane 0 scales its whole row, the
         * others multiply by one, which changes nothing */
        constexpr SplitConst RS = split_const(0.707106781);
        const SplitConst row_scale = {(r == 0) ? RS.hi : 1.0f, (r == 0) ? RS.lo : 0.0f};
        uint32_t slot = grp % bpm;                           /* block slot inside the MCU, kept up incrementally */
        const uint32_t slot_step = (TPB / 8) % bpm;
        for (uint32_t b0 = 0; b0 < nblk; b0 += TPB / 8, slot += slot_step) {
            const uint32_t blk = b0 + grp;
            if (blk >= nblk) continue;                       /* whole groups of eight lanes drop out together */
            slot = (slot >= bpm) ? slot - bpm : slot;
            const uint32_t comp = im.blk_comp[slot];
            uint8_t *bp = s_blk + __umul24(blk, COEF_STRIDE);
            float c[8];
            {
                const uint4 qv = *reinterpret_cast<const uint4 *>(&s_qt[comp][8 * r]);
                const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t zi = (((j < 4) ? zr.x : zr.y) >> (8 * (j & 3))) & 0xFFu;
                    const int32_t z = *reinterpret_cast<const int16_t *>(bp + 2 * zi);
                    const int32_t q = (int32_t) ((j & 1) ? (qw[j >> 1] >> 16) : (qw[j >> 1] & 0xFFFFu));
                    /* int16 wrap of the product (quant_table.c:137) */
                    c[j] = (float) (int16_t) (z * q);
                }
            }
            c[0] = mulc(c[0], RS);                           /* first column */
#pragma unroll
            for (int j = 0; j < 8; j += 2) {                 /* first row */
                const f2 v = mulc((f2){c[j], c[j + 1]}, row_scale);
                c[j] = v.x;
                c[j + 1] = v.y;
            }
            idct8<false>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
            /* rows -> columns (the reference transposes, runs rows again, and transposes
             * back on the store: dct.c:173-193).  The eight lanes of a block sit in one
             * wavefront, which runs its LDS operations in order. */
            *reinterpret_cast<float4 *>(tr + r * 8) = make_float4(c[0], c[1], c[2], c[3]);
            *reinterpret_cast<float4 *>(tr + r * 8 + 4) = make_float4(c[4], c[5], c[6], c[7]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int y = 0; y < 8; y++) c[y] = tr[y * 8 + r];
            idct8<true>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int y = 0; y < 8; y++) tr[y * 8 + r] = c[y];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float4 lo4 = *reinterpret_cast<const float4 *>(tr + r * 8), hi4 = *reinterpret_cast<const float4 *>(tr + r * 8 + 4);
            const uint32_t lo = finish(lo4.w, 3, finish(lo4.z, 2, finish(lo4.y, 1, finish(lo4.x, 0, 0))));
            const uint32_t hi = finish(hi4.w, 3, finish(hi4.z, 2, finish(hi4.y, 1, finish(hi4.x, 0, 0))));
            /* row r of the samples, over the block's own coefficients: all eight lanes
             * have read theirs by now */
            *reinterpret_cast<uint2 *>(bp + r * 8) = make_uint2(lo, hi);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();   /* all samples are in place; the transpose buffers are free */
    {
        const int32_t vi = (int32_t) (threadIdx.x & 255u) - 128;
        const double v = (double) vi;
        ChromaRow row;
        row.pad_ = 0;
        if (threadIdx.x < 256) {
            row.g = 0.34414 * v;
            row.o = floor_div(443 * vi, 250);    /* 1.772 */
            s_tcb[threadIdx.x] = row;
        } else {
            row.g = 0.71414 * v;
            row.o = floor_div(701 * vi, 500);    /* 1.402 */
            s_tcr[threadIdx.x - 256] = row;
        }
    }
    const uint8_t *s_smp = s_blk;

    /* first block slot of each component inside the MCU (scan order: all blocks of a
     * component are consecutive, decode.c:591-611) */
    uint32_t sb0 = 0, sb1 = 0, sb2 = 0;
    for (uint32_t i = bpm; i-- > 0;) {
        const uint32_t cc = im.blk_comp[i];
        if (cc == 0) sb0 = i;
        else if (cc == 1) sb1 = i;
        else if (cc == 2) sb2 = i;
    }
    const uint32_t hs0 = im.comp[0].hsf, hs1 = (im.ncs > 1) ? im.comp[1].hsf : 1u, hs2 = (im.ncs > 2) ? im.comp[2].hsf : 1u;
    /* sample (x, y) of component c, in tile coordinates of that component */
    auto smp = [&](uint32_t sb, uint32_t hs, uint32_t x, uint32_t y) -> uint32_t {
        const uint32_t bx = x >> 3, m = bx / hs, h = bx - m * hs;
        return sample_off(m * bpm + sb + (y >> 3) * hs + h, x, y);
    };
    __syncthreads();

    /* ---- optional: the padded component planes (Image.buf, decode.c:517-531) */
    if (b.want_planes) {
        for (uint32_t i = threadIdx.x; i < nblk * 8; i += TPB) {
            const uint32_t blk = i >> 3, row = i & 7u;
            const uint32_t m = blk / bpm, slot = blk - m * bpm;
            const uint32_t cc = im.blk_comp[slot];
            const uint32_t hs = im.comp[cc].hsf, vs = im.comp[cc].vsf;
            const uint64_t py = (uint64_t) mrow * vs * 8 + im.blk_v[slot] * 8u + row;
            const uint64_t px = ((uint64_t) (m0 + m) * hs + im.blk_h[slot]) * 8;
            *reinterpret_cast<uint2 *>(b.planes + im.comp[cc].plane_off + py * im.comp[cc].pw + px) =
                *reinterpret_cast<const uint2 *>(s_smp + blk * COEF_STRIDE + row * 8);
        }
    }

    if (MJ_DBG(b) & 8u) return;
    /* ---- 3. upsample + colour, 4 pixels (12 bytes) per thread per step ----- */
    /* The tile's quads (4 horizontally adjacent pixels) are numbered row-major; a
     * thread takes quad tid, tid + TPB, ... and tracks (row, quad-in-row)
     * incrementally, so the loop has no division. */
    const uint32_t tile_w = nm * im.hmax * 8, tile_h = im.vmax * 8;
    const uint32_t px0 = m0 * im.hmax * 8, py0 = mrow * tile_h;
    const uint32_t quads = tile_w / 4;
    const uint32_t dy = TPB / quads, dq = TPB - dy * quads;
    uint32_t y = threadIdx.x / quads;
    uint32_t qx = threadIdx.x - y * quads;
    uint8_t *rgb = b.rgb + im.rgb_off;
    const bool rgba = b.rgba != 0;
    const uint32_t bpp = rgba ? 4u : 3u;
    if (im.ncs == 1) {
        for (; y < tile_h; y += dy, qx += dq) {
            if (qx >= quads) {
                qx -= quads;
                y++;
                if (y >= tile_h) break;
            }
            const uint32_t x = qx * 4, gy = py0 + y, gx = px0 + x;
            if (gy >= im.Y || gx >= im.X) continue;
            const uint32_t v = *reinterpret_cast<const uint32_t *>(s_smp + smp(sb0, hs0, x, y));
            const uint32_t a = v & 0xFF, bb = (v >> 8) & 0xFF, cc = (v >> 16) & 0xFF, d = v >> 24;
            const uint32_t px[4] = {a * 0x010101u, bb * 0x010101u, cc * 0x010101u, d * 0x010101u};
            store_quad(rgb + (uint64_t) gy * im.pitch + bpp * gx, px, gx, im.X, rgba);
        }
        return;
    }
    {
        const uint32_t ch = im.hmax / im.comp[1].hsf, cv = im.vmax / im.comp[1].vsf;
        const bool same = im.comp[1].hsf == 1 && im.comp[1].vsf == 1 && im.comp[2].hsf == 1 && im.comp[2].vsf == 1 &&
                          im.comp[0].hsf == im.hmax && im.comp[0].vsf == im.vmax;
        if (same && ch != 3 && cv <= 2) {
#define MJ_FAST(SH, SV)                                                                                          \
    colour_fast<SH, SV>(s_smp, s_tcb, s_tcr, bpm, sb0, sb1, sb2, tile_w, tile_h, px0, py0, im.X, im.Y, im.pitch, rgb, rgba)
            if (cv == 1) {
                if (ch == 1) MJ_FAST(0, 0);
                else if (ch == 2) MJ_FAST(1, 0);
                else MJ_FAST(2, 0);
            } else {
                if (ch == 1) MJ_FAST(0, 1);
                else if (ch == 2) MJ_FAST(1, 1);
                else MJ_FAST(2, 1);
            }
#undef MJ_FAST
            return;
        }
    }
    /* Source sample of output (i, j) in component c is (uint32)(ratio * i) with
     * ratio = (float) sf_c / max_sf (colour_conversion.c:44-45, :62-69).  Sampling
     * factors divide the maximum here, so that is i / (max_sf / sf_c). */
    const uint32_t qh0 = im.hmax / im.comp[0].hsf, qh1 = im.hmax / im.comp[1].hsf, qh2 = im.hmax / im.comp[2].hsf;
    const uint32_t qv0 = im.vmax / im.comp[0].vsf, qv1 = im.vmax / im.comp[1].vsf, qv2 = im.vmax / im.comp[2].vsf;
    for (; y < tile_h; y += dy, qx += dq) {
        if (qx >= quads) {
            qx -= quads;
            y++;
            if (y >= tile_h) break;
        }
        const uint32_t x = qx * 4, gy = py0 + y, gx = px0 + x;
        if (gy >= im.Y || gx >= im.X) continue;
        uint32_t px[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t xx = x + k;
            px[k] = ycc_to_rgb24(s_smp[smp(sb0, hs0, xx / qh0, y / qv0)], s_tcb[s_smp[smp(sb1, hs1, xx / qh1, y / qv1)]],
                                 s_tcr[s_smp[smp(sb2, hs2, xx / qh2, y / qv2)]]);
        }
        store_quad(rgb + (uint64_t) gy * im.pitch + bpp * gx, px, gx, im.X, rgba);
    }
}

}  // namespace

hipError_t mj_launch_idct_colour(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_idct_wg == 0) return hipSuccess;
    hipLaunchKernelGGL(mj_k_idct_colour, dim3(b.n_idct_wg), dim3(TPB), 0, s, b);
    return hipGetLastError();
}
