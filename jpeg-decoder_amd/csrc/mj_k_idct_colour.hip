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
constexpr int MAX_TILE_W = 64 * 8 * 4;         /* widest luma tile row in samples (T <= 64 MCUs, hmax <= 4)... */
constexpr int PLANE_BYTES = MAXB * 64;         /* every sample of every block of the tile, all components */

__constant__ uint8_t c_zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* compile-time zigzag table for fully unrolled register indexing */
struct ZZ {
    static constexpr int v[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
};

/* 8-point inverse butterfly (dct.c:296-341).  (float)(C * (double)x) is a product
 * rounded to double and then to float, as the C source's mixed types imply. */
__device__ __forceinline__ void idct8(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5, float &d6,
                                      float &d7) {
    const float t0 = (float) (1.414213562 * (double) d0);
    const float t1 = d4, t2 = d2, t3 = d6;
    const float t4 = 0.5f * (d1 - d7);
    const float t5 = (float) (0.707106781 * (double) d3);
    const float t6 = (float) (0.707106781 * (double) d5);
    const float t7 = 0.5f * (d1 + d7);

    const float u0 = 0.5f * (t0 + t1);
    const float u1 = 0.5f * (t0 - t1);
    const double t2d = (double) t2, t3d = (double) t3;
    const float u2 = (float) (0.707106781 * (0.38268343236 * t2d + -0.92387953251 * t3d));
    const float u3 = (float) (0.707106781 * (0.92387953251 * t2d + 0.38268343236 * t3d));
    const float u4 = 0.5f * (t4 + t6);
    const float u5 = 0.5f * (t7 - t5);
    const float u6 = 0.5f * (t4 - t6);
    const float u7 = 0.5f * (t5 + t7);

    const float w0 = 0.5f * (u0 + u3);
    const float w1 = 0.5f * (u1 + u2);
    const float w2 = 0.5f * (u1 - u2);
    const float w3 = 0.5f * (u0 - u3);
    const double u4d = (double) u4, u5d = (double) u5, u6d = (double) u6, u7d = (double) u7;
    const float w4 = (float) (0.8314696123 * u4d + -0.55557023302 * u7d);
    const float w5 = (float) (0.9807852804 * u5d + -0.19509032201 * u6d);
    const float w6 = (float) (0.19509032201 * u5d + 0.9807852804 * u6d);
    const float w7 = (float) (0.55557023302 * u4d + 0.8314696123 * u7d);

    const double K = 1.414213562 * 2;
    d0 = (float) (K * (double) (w0 + w7));
    d1 = (float) (K * (double) (w1 + w6));
    d2 = (float) (K * (double) (w2 + w5));
    d3 = (float) (K * (double) (w3 + w4));
    d4 = (float) (K * (double) (w3 - w4));
    d5 = (float) (K * (double) (w2 - w5));
    d6 = (float) (K * (double) (w1 - w6));
    d7 = (float) (K * (double) (w0 - w7));
}

/* dct.c:187-193 with CLAMP_8 (dct.c:6): 0.25*x + 128 in double, clamp, truncate */
__device__ __forceinline__ uint32_t finish(float x) {
    const double v = 0.25 * (double) x + 128.0;
    const double c = (v > 255.0) ? 255.0 : ((v < 0.0) ? 0.0 : v);
    return (uint32_t) (int32_t) c;
}

/* colour_conversion.c:76-79 as compiled by gcc 11.4 -O2: below zero -> 0, above
 * 256 -> 255, otherwise truncate (through a 32-bit integer) */
__device__ __forceinline__ uint32_t store8(float v) {
    if (v < 0.0f) return 0u;
    if (v > 256.0f) return 255u;
    return (uint32_t) (int32_t) v & 0xFFu;
}

/* colour_conversion.c:71-74: double arithmetic left to right, rounded to float */
__device__ __forceinline__ uint32_t ycc_to_rgb24(uint32_t Y, uint32_t Cb, uint32_t Cr) {
    const double y = (double) Y;
    const double cb = (double) (float) Cb - 128.0, cr = (double) (float) Cr - 128.0;
    const float R = (float) (y + 1.402 * cr);
    const float G = (float) ((y - 0.34414 * cb) - 0.71414 * cr);
    const float B = (float) (y + 1.772 * cb);
    return store8(R) | (store8(G) << 8) | (store8(B) << 16);
}

__global__ __launch_bounds__(TPB) void mj_k_idct_colour(MjDeviceBatch b) {
    __shared__ __align__(16) uint8_t s_coef[MAXB * COEF_STRIDE];
    __shared__ __align__(16) uint8_t s_plane[PLANE_BYTES];
    __shared__ __align__(16) uint16_t s_qt[MJ_MAX_COMP][64];

    const uint32_t ii = b.idct_wg_img[blockIdx.x];
    const MjImageDev &im = b.img[ii];
    const uint32_t bpm = im.bpm;
    const uint32_t tiles_x = mj_idct_tiles_x(im.mcus_x, bpm);
    const uint32_t T = mj_idct_tile_mcus(im.mcus_x, bpm);
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
    if (threadIdx.x < nblk) {
        const uint32_t t = threadIdx.x / bpm, slot = threadIdx.x % bpm;
        const uint32_t comp = im.blk_comp[slot];
        const uint16_t *q = s_qt[comp];
        const uint4 *cz = reinterpret_cast<const uint4 *>(s_coef + threadIdx.x * COEF_STRIDE);
        uint32_t zw[32];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint4 v = cz[i];
            zw[4 * i + 0] = v.x;
            zw[4 * i + 1] = v.y;
            zw[4 * i + 2] = v.z;
            zw[4 * i + 3] = v.w;
        }
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
            idct8(c[8 * r + 0], c[8 * r + 1], c[8 * r + 2], c[8 * r + 3], c[8 * r + 4], c[8 * r + 5], c[8 * r + 6],
                  c[8 * r + 7]);
        }
        /* second pass down the columns (the reference transposes, runs rows, and
         * transposes back on the store: dct.c:173-193) */
#pragma unroll
        for (int x = 0; x < 8; x++) {
            idct8(c[x], c[8 + x], c[16 + x], c[24 + x], c[32 + x], c[40 + x], c[48 + x], c[56 + x]);
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

    /* ---- 3. upsample + colour, 4 pixels (12 bytes) per thread per step ----- */
    const uint32_t tile_w = nm * im.hmax * 8, tile_h = im.vmax * 8;
    const uint32_t px0 = m0 * im.hmax * 8, py0 = mrow * tile_h;
    const uint32_t quads = tile_w / 4;
    uint8_t *rgb = b.rgb + im.rgb_off;
    if (im.ncs == 1) {
        for (uint32_t i = threadIdx.x; i < quads * tile_h; i += TPB) {
            const uint32_t y = i / quads, x = (i % quads) * 4;
            const uint32_t gy = py0 + y, gx = px0 + x;
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
    /* integer form of (uint32)(ratio * i), ratio = (float)sf / max_sf
     * (colour_conversion.c:44-45, :62-69); exact for the factor pairs accepted */
    const uint32_t h0 = im.comp[0].hsf, h1 = im.comp[1].hsf, h2 = im.comp[2].hsf;
    const uint32_t v0 = im.comp[0].vsf, v1 = im.comp[1].vsf, v2 = im.comp[2].vsf;
    const uint32_t hm = im.hmax, vm = im.vmax;
    for (uint32_t i = threadIdx.x; i < quads * tile_h; i += TPB) {
        const uint32_t y = i / quads, x = (i % quads) * 4;
        const uint32_t gy = py0 + y, gx = px0 + x;
        if (gy >= im.Y || gx >= im.X) continue;
        const uint8_t *r0 = s_plane + po0 + (y * v0 / vm) * pw0;
        const uint8_t *r1 = s_plane + po1 + (y * v1 / vm) * pw1;
        const uint8_t *r2 = s_plane + po2 + (y * v2 / vm) * pw2;
        uint32_t px[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t xx = x + k;
            px[k] = ycc_to_rgb24(r0[xx * h0 / hm], r1[xx * h1 / hm], r2[xx * h2 / hm]);
        }
        uint8_t *o = rgb + (uint64_t) gy * im.pitch + 3 * gx;
        if (gx + 4 <= im.X) {
            uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
            o32[0] = px[0] | (px[1] << 24);
            o32[1] = (px[1] >> 8) | (px[2] << 16);
            o32[2] = (px[2] >> 16) | (px[3] << 8);
        } else {
            for (uint32_t k = 0; k < 4 && gx + k < im.X; k++) {
                o[3 * k] = (uint8_t) px[k];
                o[3 * k + 1] = (uint8_t) (px[k] >> 8);
                o[3 * k + 2] = (uint8_t) (px[k] >> 16);
            }
        }
    }
}

}  // namespace

hipError_t mj_launch_idct_colour(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_idct_wg == 0) return hipSuccess;
    hipLaunchKernelGGL(mj_k_idct_colour, dim3(b.n_idct_wg), dim3(TPB), 0, s, b);
    return hipGetLastError();
}
