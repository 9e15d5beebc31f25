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
 *   2. each thread takes one 8x8 block: de-zigzag + dequantise (int16 wrap) into
 *      registers, sixteen 8-point butterflies, level shift, clamp, truncate, and
 *      stores the 64 samples into the tile's component planes in LDS;
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
constexpr int COEF_STRIDE = 144;               /* bytes per block in LDS: 128 + 16 pad, conflict-free b128 reads */
constexpr int PLANE_BYTES = MAXB * 64;         /* every sample of every block of the tile, all components */

/* compile-time zigzag table for fully unrolled register indexing */
struct ZZ {
    static constexpr int v[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
};

/* 8-point inverse butterfly (dct.c:296-341).  (float)(C * (double)x) is a product
 * rounded to double and then to float, as the C source's mixed types imply.
 *
 * The reference halves after every add ("0.5 * (a + b)").  Multiplying by a power
 * of two is exact and commutes with every rounding, so the halvings are carried as
 * an implicit scale instead (capital letters = 2x or 4x the reference's variable)
 * and folded into the double constants: the results are bit-identical and twelve
 * multiplies per butterfly disappear. */
template <bool FINAL>
__device__ __forceinline__ void idct8(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5, float &d6,
                                      float &d7) {
    /* FINAL: the caller's next step is 0.25 * x + 128 (dct.c:192); the exact factor
     * 0.25 is folded into the last constant as well */
    constexpr double C = FINAL ? 0.25 * 0.707106781 : 0.707106781, C2 = 2.0 * 0.707106781;
    const float t0 = (float) (1.414213562 * (double) d0);
    const float t1 = d4, t2 = d2, t3 = d6;
    const float T4 = d1 - d7;                            /* 2 t4 */
    const float T5 = (float) (C2 * (double) d3);         /* 2 t5 */
    const float T6 = (float) (C2 * (double) d5);         /* 2 t6 */
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
    d0 = (float) (C * (double) (W0 + W7));
    d1 = (float) (C * (double) (W1 + W6));
    d2 = (float) (C * (double) (W2 + W5));
    d3 = (float) (C * (double) (W3 + W4));
    d4 = (float) (C * (double) (W3 - W4));
    d5 = (float) (C * (double) (W2 - W5));
    d6 = (float) (C * (double) (W1 - W6));
    d7 = (float) (C * (double) (W0 - W7));
}

/* dct.c:187-193 with CLAMP_8 (dct.c:6): clamp(0.25*x + 128) in double, then truncate.
 * q is 0.25*x already (see idct8<true>).  q + 128 is exact in double, and for a sum
 * that is not negative truncation is floor(q) + 128; a negative sum clamps to 0 either
 * way.  So the double arithmetic reduces, bit for bit, to a float floor, a clamp to
 * [-128, 127] (which also keeps the integer conversion in range) and an integer add. */
__device__ __forceinline__ uint32_t finish(float q) {
    const float f = fminf(fmaxf(floorf(q), -128.0f), 127.0f);
    return (uint32_t) ((int32_t) f + 128);
}

/* colour_conversion.c:76-79 as compiled by gcc 11.4 -O2: below zero -> 0, above
 * 256 -> 255, otherwise truncate through a 32-bit integer (so exactly 256.0f, which
 * one (Y,Cb,Cr) triple reaches in G, wraps to 0). */
__device__ __forceinline__ uint32_t store8(float v) {
    const int32_t i = (int32_t) v;
    const uint32_t r = (uint32_t) min(max(i, 0), 255);
    return (v == 256.0f) ? 0u : r;
}

/* colour_conversion.c:71-74.  The products with the chroma samples come from
 * per-workgroup LDS tables indexed by the sample value:
 *   cr -> (1.402 * (cr - 128), 0.71414 * (cr - 128)),  cb -> (0.34414 * (cb - 128), 1.772 * (cb - 128))
 * computed once per value in double exactly as the expression does; the adds keep
 * the source's left-to-right order. */
__device__ __forceinline__ uint32_t ycc_to_rgb24(uint32_t Y, const double2 tcb, const double2 tcr) {
    const double y = (double) Y;
    const float R = (float) (y + tcr.x);
    const float G = (float) ((y - tcb.x) - tcr.y);
    const float B = (float) (y + tcb.y);
    return store8(R) | (store8(G) << 8) | (store8(B) << 16);
}

/* 12-byte store of four RGB24 pixels (4-byte aligned) */
struct __attribute__((aligned(4))) Rgb4 {
    uint32_t a, b, c;
};

__device__ __forceinline__ void store_quad(uint8_t *o, const uint32_t px[4], uint32_t gx, uint32_t X) {
    if (gx + 4 <= X) {
        Rgb4 v;
        v.a = px[0] | (px[1] << 24);
        v.b = (px[1] >> 8) | (px[2] << 16);
        v.c = (px[2] >> 16) | (px[3] << 8);
        *reinterpret_cast<Rgb4 *>(o) = v;
    } else {
        for (uint32_t k = 0; k < 4 && gx + k < X; k++) {
            o[3 * k] = (uint8_t) px[k];
            o[3 * k + 1] = (uint8_t) (px[k] >> 8);
            o[3 * k + 2] = (uint8_t) (px[k] >> 16);
        }
    }
}

/*
 * Colour stage for the common layouts: luma at full resolution, both chroma
 * components subsampled by 2^SH horizontally and 2^SV vertically (4:4:4, 4:2:2,
 * 4:2:0, 4:4:0, 4:1:1).  A thread converts a patch of 4 x 2^SV pixels per step, so
 * every chroma sample and its two table rows are fetched once per patch.
 */
template <int SH, int SV>
__device__ __forceinline__ void colour_fast(const uint8_t *s_plane, const double2 *s_tcb, const double2 *s_tcr,
                                            uint32_t po1, uint32_t po2, uint32_t pw0, uint32_t pwc, uint32_t tile_w,
                                            uint32_t tile_h, uint32_t px0, uint32_t py0, uint32_t X, uint32_t Y,
                                            uint32_t pitch, uint8_t *rgb) {
    constexpr int NC = 4 >> SH, ROWS = 1 << SV;
    const uint32_t quads = tile_w / 4, prow_n = tile_h >> SV;
    const uint32_t dy = TPB / quads, dq = TPB - dy * quads;
    uint32_t pr = threadIdx.x / quads;
    uint32_t qx = threadIdx.x - pr * quads;
    for (; pr < prow_n; pr += dy, qx += dq) {
        if (qx >= quads) {
            qx -= quads;
            pr++;
            if (pr >= prow_n) break;
        }
        const uint32_t x = qx * 4, gx = px0 + x;
        const uint32_t y = pr << SV, gy = py0 + y;
        if (gy >= Y || gx >= X) continue;
        /* chroma samples of the patch and their product rows */
        const uint8_t *cbp = s_plane + po1 + pr * pwc + (x >> SH);
        const uint8_t *crp = s_plane + po2 + pr * pwc + (x >> SH);
        uint32_t cbw, crw;
        if (SH == 0) {
            cbw = *reinterpret_cast<const uint32_t *>(cbp);
            crw = *reinterpret_cast<const uint32_t *>(crp);
        } else if (SH == 1) {
            cbw = *reinterpret_cast<const uint16_t *>(cbp);
            crw = *reinterpret_cast<const uint16_t *>(crp);
        } else {
            cbw = *cbp;
            crw = *crp;
        }
        double2 tb[NC], tr[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) {
            tb[i] = s_tcb[(cbw >> (8 * i)) & 0xFFu];
            tr[i] = s_tcr[(crw >> (8 * i)) & 0xFFu];
        }
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if (gy + r >= Y) break;
            const uint32_t yw = *reinterpret_cast<const uint32_t *>(s_plane + (y + r) * pw0 + x);
            uint32_t px[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                px[k] = ycc_to_rgb24((yw >> (8 * k)) & 0xFFu, tb[k >> SH], tr[k >> SH]);
            }
            store_quad(rgb + (uint64_t) (gy + r) * pitch + 3 * gx, px, gx, X);
        }
    }
}

#ifndef MJ_IDCT_WAVES
#define MJ_IDCT_WAVES 3
#endif
__global__ __launch_bounds__(TPB, MJ_IDCT_WAVES) void mj_k_idct_colour(MjDeviceBatch b) {
    /* One LDS region used twice: first the tile's coefficients (144 B per block),
     * then -- once every thread holds its block in registers -- the tile's sample
     * planes (64 B per block) followed by the two chroma product tables. */
    __shared__ __align__(16) uint8_t s_mem[MAXB * COEF_STRIDE];
    __shared__ __align__(16) uint16_t s_qt[MJ_MAX_COMP][64];
    uint8_t *s_coef = s_mem;
    uint8_t *s_plane = s_mem;
    double2 *s_tcb = reinterpret_cast<double2 *>(s_mem + PLANE_BYTES);
    double2 *s_tcr = s_tcb + 256;
    static_assert(PLANE_BYTES + 2 * 256 * sizeof(double2) <= MAXB * COEF_STRIDE, "LDS reuse");

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
        *reinterpret_cast<uint4 *>(s_coef + (i >> 3) * COEF_STRIDE + (i & 7) * 16) = v;
    }
    for (uint32_t i = threadIdx.x; i < im.ncs * 64; i += TPB) {
        s_qt[i >> 6][i & 63] = im.qt[i >> 6][i & 63];
    }
    __syncthreads();

    /* plane layout inside s_plane: component c occupies rows of width w_c = nm*hsf*8,
     * height vsf*8, one after the other (scalars + selects: no indexed arrays) */
    const uint32_t pw0 = nm * im.comp[0].hsf * 8;
    const uint32_t pw1 = (im.ncs > 1) ? nm * im.comp[1].hsf * 8 : 0;
    const uint32_t pw2 = (im.ncs > 2) ? nm * im.comp[2].hsf * 8 : 0;
    const uint32_t po0 = 0;
    const uint32_t po1 = pw0 * im.comp[0].vsf * 8;
    const uint32_t po2 = po1 + ((im.ncs > 1) ? pw1 * im.comp[1].vsf * 8 : 0);
    auto pl_w = [&](uint32_t c) { return c == 0 ? pw0 : (c == 1 ? pw1 : pw2); };
    auto pl_off = [&](uint32_t c) { return c == 0 ? po0 : (c == 1 ? po1 : po2); };

    /* ---- 2. one block per thread ------------------------------------------ */
    const bool has_block = threadIdx.x < nblk;
    uint32_t zw[32];
    if (has_block) {
        const uint4 *cz = reinterpret_cast<const uint4 *>(s_coef + threadIdx.x * COEF_STRIDE);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint4 v = cz[i];
            zw[4 * i + 0] = v.x;
            zw[4 * i + 1] = v.y;
            zw[4 * i + 2] = v.z;
            zw[4 * i + 3] = v.w;
        }
    }
    __syncthreads();   /* every block is in registers: the region becomes planes + tables */
    {
        const double v = (double) (float) threadIdx.x - 128.0;   /* TPB == 256 sample values */
        s_tcb[threadIdx.x] = make_double2(0.34414 * v, 1.772 * v);
        s_tcr[threadIdx.x] = make_double2(1.402 * v, 0.71414 * v);
    }
    if (has_block && !(b.dbg & 16u)) {
        const uint32_t t = threadIdx.x / bpm, slot = threadIdx.x % bpm;
        const uint32_t comp = im.blk_comp[slot];
        const uint16_t *q = s_qt[comp];
        float c[64];
#pragma unroll
        for (int i = 0; i < 64; i++) {
            const int n = ZZ::v[i];
            const int32_t z = (int32_t) (int16_t) ((i & 1) ? (zw[i >> 1] >> 16) : (zw[i >> 1] & 0xFFFFu));
            /* int16 wrap of the product (quant_table.c:137) */
            c[n] = (float) (int16_t) (z * (int32_t) q[n]);
        }
        /* dct.c:164-167: first row and first column scaled; c[0] twice */
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c[i] = (float) ((double) c[i] * 0.707106781);
            c[8 * i] = (float) ((double) c[8 * i] * 0.707106781);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            idct8<false>(c[8 * r + 0], c[8 * r + 1], c[8 * r + 2], c[8 * r + 3], c[8 * r + 4], c[8 * r + 5], c[8 * r + 6],
                         c[8 * r + 7]);
        }
        /* second pass down the columns (the reference transposes, runs rows, and
         * transposes back on the store: dct.c:173-193) */
#pragma unroll
        for (int x = 0; x < 8; x++) {
            idct8<true>(c[x], c[8 + x], c[16 + x], c[24 + x], c[32 + x], c[40 + x], c[48 + x], c[56 + x]);
        }
        const uint32_t x0 = (t * im.comp[comp].hsf + im.blk_h[slot]) * 8, y0 = im.blk_v[slot] * 8u;
        uint8_t *dst = s_plane + pl_off(comp) + y0 * pl_w(comp) + x0;
#pragma unroll
        for (int y = 0; y < 8; y++) {
            uint32_t lo = finish(c[8 * y + 0]) | (finish(c[8 * y + 1]) << 8) | (finish(c[8 * y + 2]) << 16) |
                          (finish(c[8 * y + 3]) << 24);
            uint32_t hi = finish(c[8 * y + 4]) | (finish(c[8 * y + 5]) << 8) | (finish(c[8 * y + 6]) << 16) |
                          (finish(c[8 * y + 7]) << 24);
            *reinterpret_cast<uint2 *>(dst + y * pl_w(comp)) = make_uint2(lo, hi);
        }
    }
    __syncthreads();

    /* ---- optional: the padded component planes (Image.buf, decode.c:517-531) */
    if (b.want_planes) {
        for (uint32_t c = 0; c < im.ncs; c++) {
            const uint32_t w = pl_w(c), h = im.comp[c].vsf * 8;
            uint8_t *pl = b.planes + im.comp[c].plane_off + (uint64_t) (mrow * h) * im.comp[c].pw + m0 * im.comp[c].hsf * 8;
            for (uint32_t i = threadIdx.x; i < (w / 4) * h; i += TPB) {
                const uint32_t y = i / (w / 4), x4 = i % (w / 4);
                *reinterpret_cast<uint32_t *>(pl + (uint64_t) y * im.comp[c].pw + x4 * 4) =
                    *reinterpret_cast<const uint32_t *>(s_plane + pl_off(c) + y * w + x4 * 4);
            }
        }
    }

    if (b.dbg & 8u) return;
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
    if (im.ncs == 1) {
        for (; y < tile_h; y += dy, qx += dq) {
            if (qx >= quads) {
                qx -= quads;
                y++;
                if (y >= tile_h) break;
            }
            const uint32_t x = qx * 4, gy = py0 + y, gx = px0 + x;
            if (gy >= im.Y || gx >= im.X) continue;
            const uint32_t v = *reinterpret_cast<const uint32_t *>(s_plane + y * pw0 + x);
            const uint32_t a = v & 0xFF, bb = (v >> 8) & 0xFF, cc = (v >> 16) & 0xFF, d = v >> 24;
            uint8_t *o = rgb + (uint64_t) gy * im.pitch + 3 * gx;
            if (gx + 4 <= im.X) {
                uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
                o32[0] = a | (a << 8) | (a << 16) | (bb << 24);
                o32[1] = bb | (bb << 8) | (cc << 16) | (cc << 24);
                o32[2] = cc | (d << 8) | (d << 16) | (d << 24);
            } else {
                const uint32_t px[4] = {a, bb, cc, d};
                for (uint32_t k = 0; k < 4 && gx + k < im.X; k++) {
                    o[3 * k] = o[3 * k + 1] = o[3 * k + 2] = (uint8_t) px[k];
                }
            }
        }
        return;
    }
    {
        const uint32_t ch = im.hmax / im.comp[1].hsf, cv = im.vmax / im.comp[1].vsf;
        const bool same = im.comp[1].hsf == im.comp[2].hsf && im.comp[1].vsf == im.comp[2].vsf &&
                          im.comp[0].hsf == im.hmax && im.comp[0].vsf == im.vmax;
        if (same && ch != 3 && cv <= 2) {
            const uint32_t pwc = pw1;
#define MJ_FAST(SH, SV)                                                                                          \
    colour_fast<SH, SV>(s_plane, s_tcb, s_tcr, po1, po2, pw0, pwc, tile_w, tile_h, px0, py0, im.X, im.Y, im.pitch, rgb)
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
     * factors divide the maximum here, so that is i / (max_sf / sf_c): a shift when
     * the quotient is 1, 2 or 4, a division by 3 otherwise. */
    const uint32_t qh0 = im.hmax / im.comp[0].hsf, qh1 = im.hmax / im.comp[1].hsf, qh2 = im.hmax / im.comp[2].hsf;
    const uint32_t qv0 = im.vmax / im.comp[0].vsf, qv1 = im.vmax / im.comp[1].vsf, qv2 = im.vmax / im.comp[2].vsf;
    const bool pow2 = qh0 != 3 && qh1 != 3 && qh2 != 3 && qv0 != 3 && qv1 != 3 && qv2 != 3;
    const uint32_t sh0 = qh0 >> 1, sh1 = qh1 >> 1, sh2 = qh2 >> 1;   /* 1,2,4 -> 0,1,2 */
    const uint32_t sv0 = qv0 >> 1, sv1 = qv1 >> 1, sv2 = qv2 >> 1;
    for (; y < tile_h; y += dy, qx += dq) {
        if (qx >= quads) {
            qx -= quads;
            y++;
            if (y >= tile_h) break;
        }
        const uint32_t x = qx * 4, gy = py0 + y, gx = px0 + x;
        if (gy >= im.Y || gx >= im.X) continue;
        const uint8_t *r0, *r1, *r2;
        uint32_t px[4];
        if (pow2) {
            r0 = s_plane + po0 + (y >> sv0) * pw0;
            r1 = s_plane + po1 + (y >> sv1) * pw1;
            r2 = s_plane + po2 + (y >> sv2) * pw2;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t xx = x + k;
                px[k] = ycc_to_rgb24(r0[xx >> sh0], s_tcb[r1[xx >> sh1]], s_tcr[r2[xx >> sh2]]);
            }
        } else {
            r0 = s_plane + po0 + (y / qv0) * pw0;
            r1 = s_plane + po1 + (y / qv1) * pw1;
            r2 = s_plane + po2 + (y / qv2) * pw2;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t xx = x + k;
                px[k] = ycc_to_rgb24(r0[xx / qh0], s_tcb[r1[xx / qh1]], s_tcr[r2[xx / qh2]]);
            }
        }
        store_quad(rgb + (uint64_t) gy * im.pitch + 3 * gx, px, gx, im.X);
    }
}

}  // namespace

hipError_t mj_launch_idct_colour(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_idct_wg == 0) return hipSuccess;
    hipLaunchKernelGGL(mj_k_idct_colour, dim3(b.n_idct_wg), dim3(TPB), 0, s, b);
    return hipGetLastError();
}
