/*
 * mj_k_scan.hip -- the byte-level front end of the entropy stage, on the GPU.
 *
 * Replaces, for the whole entropy-coded segment at once, what the reference does a
 * byte at a time inside its bit reader: stepping over the 00 stuffed after a data FF
 * (next_byte_for_bits, bitstream.c:22-59), noticing RSTn / end markers
 * (check_marker, bitstream.c:84-134) and jumping past RSTn
 * (next_byte_restart_marker, bitstream.c:142-176).
 *
 *   mj_k_scan_count   per 4 KiB tile: surviving bytes, RSTn count, first end marker
 *   mj_k_scan_prefix  per image: exclusive prefix over its tiles
 *   mj_k_scan_write   per tile: write the unstuffed, marker-free stream and the byte
 *                     offset at which every restart interval starts in it
 *   mj_k_layout       per image: cut every interval into subsequences of sub_bits
 *                     bits and groups of 64 subsequences (one wavefront each)
 *
 * All four are streaming byte kernels: HBM/L2-bound, no MFMA, ~0.25 B per pixel.
 */
#include "mj_entropy_core.h"
#include "mj_internal.h"

namespace {

constexpr int TPB = 256;
constexpr int GR = MJ_SCAN_TILE / (TPB * 16);     /* groups of sixteen consecutive raw bytes per thread */
static_assert(GR >= 1 && TPB * 16 * GR == MJ_SCAN_TILE, "a thread takes whole groups of 16 raw bytes");

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

/* exclusive scan over the 256 threads of a block; *total gets the block sum */
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_w /*[4]*/, uint32_t *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < TPB / 64; i++) {
        uint32_t t = s_w[i];
        if (i < w) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

/* Classes of this thread's 16 bytes with everything at or past the terminating marker
 * (raw offset term) removed; kept and rst get the counts of what is left. */
__device__ __forceinline__ void cut_at(MjClass16 &c, uint32_t pos, uint32_t term, uint32_t &kept, uint32_t &rst) {
    const uint32_t nlive = term > pos ? min(term - pos, 16u) : 0u;
    kept = rst = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t live = mj_bytes_below(nlive, i);
        c.keep[i] &= live;
        c.rst[i] &= live;
        kept += __popc(c.keep[i]);
        rst += __popc(c.rst[i]);
    }
}

__global__ __launch_bounds__(TPB) void mj_k_scan_count(MjDeviceBatch b) {
    __shared__ uint32_t s_term;
    __shared__ uint32_t s_sum[2];
    const uint32_t tile = blockIdx.x;
    const uint32_t ii = b.tile_img[tile];
    const MjImageDev &im = b.img[ii];
    const uint32_t len = (uint32_t) im.raw_len;
    const uint8_t *seg = b.raw + im.raw_off;
    const uint32_t pos0 = (tile - im.tile_off) * MJ_SCAN_TILE + threadIdx.x * (16 * GR);
    if (threadIdx.x == 0) {
        s_term = MJ_NO_TERMINATOR;
        s_sum[0] = s_sum[1] = 0;
    }
    /* the decode's counters start at zero: nothing reads or adds to them before the layout kernel */
    if (blockIdx.x == 0) {
        if (threadIdx.x < MJ_ST_COUNT) b.stats[threadIdx.x] = 0ull;
        if (threadIdx.x < 4) b.multi_count[threadIdx.x] = 0u;
    }
    __syncthreads();
    MjClass16 cls[GR];
    uint32_t my_term = MJ_NO_TERMINATOR;
#pragma unroll
    for (int g = GR - 1; g >= 0; g--) {
        const uint32_t pos = pos0 + 16 * g;
        cls[g] = MjClass16{};
        if (pos < len) {
            uint32_t w[4];
            cls[g] = mj_classify16_at(seg, len, pos, w);
            /* first terminating marker in the tile */
#pragma unroll
            for (int i = 3; i >= 0; i--) {
                if (cls[g].term[i]) my_term = pos + 4 * i + ((uint32_t) __builtin_ctz(cls[g].term[i]) >> 3);
            }
        }
    }
    if (my_term != MJ_NO_TERMINATOR) atomicMin(&s_term, my_term);
    __syncthreads();
    const uint32_t term = s_term;
    uint32_t kept = 0, rst = 0;
#pragma unroll
    for (int g = 0; g < GR; g++) {
        uint32_t k1, r1;
        cut_at(cls[g], pos0 + 16 * g, term, k1, r1);
        kept += k1;
        rst += r1;
    }
    /* block sums: wave reduce then LDS atomics (4 waves) */
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        kept += __shfl_xor(kept, d, 64);
        rst += __shfl_xor(rst, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_sum[0], kept);
        atomicAdd(&s_sum[1], rst);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MjTileCnt c;
        c.kept = s_sum[0];
        c.rst = s_sum[1];
        c.term = term;
        c.pad_ = 0;
        b.tile_cnt[tile] = c;
    }
}

__global__ __launch_bounds__(TPB) void mj_k_scan_prefix(MjDeviceBatch b) {
    __shared__ uint32_t s_w[TPB / 64];
    __shared__ uint32_t s_term;
    const uint32_t ii = blockIdx.x;
    const MjImageDev &im = b.img[ii];
    const MjTileCnt *cnt = b.tile_cnt + im.tile_off;
    MjTileBase *base = b.tile_base + im.tile_off;
    if (threadIdx.x == 0) s_term = MJ_NO_TERMINATOR;
    __syncthreads();
    uint32_t t = MJ_NO_TERMINATOR;
    for (uint32_t i = threadIdx.x; i < im.n_tiles; i += TPB) t = min(t, cnt[i].term);
    if (t != MJ_NO_TERMINATOR) atomicMin(&s_term, t);
    __syncthreads();
    const uint32_t term = s_term;
    uint32_t carry_k = 0, carry_r = 0;
    for (uint32_t i0 = 0; i0 < im.n_tiles; i0 += TPB) {
        const uint32_t i = i0 + threadIdx.x;
        uint32_t k = 0, r = 0;
        if (i < im.n_tiles && (uint64_t) i * MJ_SCAN_TILE < term) {
            k = cnt[i].kept;
            r = cnt[i].rst;
        }
        uint32_t tk, tr;
        const uint32_t ek = block_excl_scan(k, s_w, &tk);
        const uint32_t er = block_excl_scan(r, s_w, &tr);
        if (i < im.n_tiles) {
            MjTileBase o;
            o.kept = carry_k + ek;
            o.rst = carry_r + er;
            base[i] = o;
        }
        carry_k += tk;
        carry_r += tr;
    }
    if (threadIdx.x == 0) {
        MjImageDyn d;
        d.term_pos = term;
        d.cmp_len = carry_k;
        d.n_rst = carry_r;
        d.n_seq = 0;
        d.status = 0;
        d.spec_guessed = 0;
        d.pad_[0] = d.pad_[1] = 0;
        b.dyn[ii] = d;
    }
}

__global__ __launch_bounds__(TPB) void mj_k_scan_write(MjDeviceBatch b) {
    __shared__ uint32_t s_w[TPB / 64];
    /* the tile's surviving bytes, placed so that LDS offset i goes to the 16-byte aligned
     * destination address plus i: whole 16-byte pieces then leave with one store */
    __shared__ __align__(16) uint8_t s_out[MJ_SCAN_TILE + 32];
    const uint32_t tile = blockIdx.x;
    const uint32_t ii = b.tile_img[tile];
    const MjImageDev &im = b.img[ii];
    const uint32_t term = b.dyn[ii].term_pos;
    const uint32_t tile_pos = (tile - im.tile_off) * MJ_SCAN_TILE;
    if (tile_pos >= term) return; /* whole block: nothing of this tile survives */
    const uint32_t len = (uint32_t) im.raw_len;
    const uint8_t *seg = b.raw + im.raw_off;
    const uint32_t pos0 = tile_pos + threadIdx.x * (16 * GR);
    MjClass16 cls[GR];
    uint32_t w[GR][4];
    uint32_t kept = 0, rst = 0;
#pragma unroll
    for (int g = 0; g < GR; g++) {
        const uint32_t pos = pos0 + 16 * g;
        cls[g] = MjClass16{};
        w[g][0] = w[g][1] = w[g][2] = w[g][3] = 0;
        if (pos < len) cls[g] = mj_classify16_at(seg, len, pos, w[g]);
        uint32_t k1, r1;
        cut_at(cls[g], pos, term, k1, r1);
        kept += k1;
        rst += r1;
    }
    uint32_t total;
    const uint32_t ex = block_excl_scan(kept | (rst << 16), s_w, &total);
    const MjTileBase tb = b.tile_base[tile];
    uint8_t *dst = b.cmp + im.cmp_off;          /* 16-byte aligned */
    const uint32_t out0 = tb.kept;               /* first output byte of the tile */
    const uint32_t skew = out0 & 15u;
    uint32_t out = skew + (ex & 0xFFFFu);        /* LDS offset of this thread's first surviving byte */
    uint32_t ridx = tb.rst + (ex >> 16);
    if (kept | rst) {
        uint32_t *ivl = b.ivl_start + im.ivl_off;
#pragma unroll
        for (int g = 0; g < GR; g++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t bit = 1u << (8 * (i & 3) + 7);
                if (cls[g].keep[i >> 2] & bit) {
                    s_out[out++] = (uint8_t) ((w[g][i >> 2] >> (8 * (i & 3))) & 0xFFu);
                } else if (cls[g].rst[i >> 2] & bit) {
                    ridx++;
                    if (ridx < im.n_ivl) ivl[ridx] = out0 - skew + out; /* interval ridx starts at the next surviving byte */
                }
            }
        }
    }
    __syncthreads();
    /* copy out: LDS [skew, skew + n) -> dst [out0, out0 + n); piece j covers LDS [16j, 16j + 16) */
    const uint32_t n = total & 0xFFFFu;
    const uint32_t lo = skew, hi = skew + n;
    uint8_t *base = dst + (out0 - skew);
    for (uint32_t j = threadIdx.x; j * 16 < hi; j += TPB) {
        const uint32_t a = j * 16, e = a + 16;
        if (a >= lo && e <= hi) {
            *reinterpret_cast<uint4 *>(base + a) = *reinterpret_cast<const uint4 *>(s_out + a);
        } else {
            for (uint32_t k = (a > lo ? a : lo); k < (e < hi ? e : hi); k++) base[k] = s_out[k];
        }
    }
}

/* Per image: interval table tail, then the subsequence / sequence layout. */
__global__ __launch_bounds__(TPB) void mj_k_layout(MjDeviceBatch b) {
    __shared__ uint32_t s_w[TPB / 64];
    __shared__ uint4 s_big[TPB];
    __shared__ uint32_t s_nbig;
    if (threadIdx.x == 0) s_nbig = 0;
    const uint32_t ii = blockIdx.x;
    const MjImageDev &im = b.img[ii];
    MjImageDyn &dy = b.dyn[ii];
    uint32_t *ivl = b.ivl_start + im.ivl_off;
    uint32_t *ivl_seq = b.ivl_seq + im.ivl_off;
    const uint32_t cmp_len = dy.cmp_len;
    const uint32_t n_rst = dy.n_rst;
    const uint32_t S = im.sub_bits, SQ = MJ_SEQ_LANES * im.lane_subs;
    /* interval 0 starts at 0; intervals whose RSTn never came are empty */
    for (uint32_t k = threadIdx.x; k <= im.n_ivl; k += TPB) {
        if (k == 0) ivl[0] = 0;
        else if (k > n_rst || k == im.n_ivl) ivl[k] = cmp_len;
    }
    __syncthreads();
    uint32_t carry = 0;
    uint32_t status = (n_rst + 1 < im.n_ivl) ? 2u : 0u;
    for (uint32_t i0 = 0; i0 < im.n_ivl; i0 += TPB) {
        const uint32_t iv = i0 + threadIdx.x;
        uint32_t nseq = 0, nsub = 0;
        if (iv < im.n_ivl) {
            const uint32_t a = ivl[iv], e = ivl[iv + 1];
            const uint32_t bits = (e > a) ? (e - a) * 8u : 0u;
            nsub = (bits + S - 1) / S;
            if (nsub == 0) nsub = 1;
            nseq = (nsub + SQ - 1) / SQ;
        }
        uint32_t tot;
        const uint32_t first = carry + block_excl_scan(nseq, s_w, &tot);
        if (iv < im.n_ivl) {
            ivl_seq[iv] = first;
            if (nseq <= 4) {
                for (uint32_t q = 0; q < nseq; q++) {
                    if (first + q < im.seq_cap) {
                        b.seq_ivl[im.seq_off + first + q] = iv;
                        b.seq_sub[im.seq_off + first + q] = q * SQ;
                    }
                }
            } else {
                /* a long interval (a scan without restart markers is one): its sequences
                 * are filled in by the whole workgroup below */
                const uint32_t slot = atomicAdd(&s_nbig, 1u);
                s_big[slot] = make_uint4(iv, first, nseq, 0);
            }
            if (nseq > 1) {
                const uint32_t slot = atomicAdd(b.multi_count, 1u);
                if (slot < b.n_multi_cap) b.multi[slot] = make_uint2(ii, iv);
            }
            atomicAdd(&b.stats[MJ_ST_SUBS], (unsigned long long) nsub);
        }
        carry += tot;
        __syncthreads();
        const uint32_t nbig = s_nbig;
        for (uint32_t e = 0; e < nbig; e++) {
            const uint4 big = s_big[e];
            for (uint32_t q = threadIdx.x; q < big.z; q += TPB) {
                if (big.y + q < im.seq_cap) {
                    b.seq_ivl[im.seq_off + big.y + q] = big.x;
                    b.seq_sub[im.seq_off + big.y + q] = q * SQ;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_nbig = 0;
    }
    if (threadIdx.x == 0) {
        if (carry > im.seq_cap) {
            status |= 1u;
            carry = im.seq_cap;
        }
        dy.n_seq = carry;
        dy.status = status;
        atomicAdd(&b.stats[MJ_ST_SEQS], (unsigned long long) carry);
    }

    /* The workgroup that finishes last lists the groups of MJ_WG_SEQS sequences that exist, image
     * after image in the host's order, and every two groups of one image: block i of an
     * entropy grid takes entry i, so the blocks with work are the first of the grid and sit
     * next to each other (the grids are sized by an upper bound on the sequences: the blocks
     * past the end of the list leave at once). */
    __shared__ uint32_t s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&b.multi_count[1], 1u) == gridDim.x - 1u) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    __shared__ uint32_t s_img[TPB], s_g[TPB], s_eg[TPB], s_ep[TPB];
    uint32_t carry_g = 0, carry_p = 0;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i0 = 0; i0 < b.n_img; i0 += TPB) {
        const uint32_t i = i0 + threadIdx.x;
        uint32_t img = 0, g = 0;
        if (i < b.n_img) {
            img = b.img_order[i];
            /* written by another workgroup of this kernel: read past this compute unit's cache */
            const uint32_t nseq = __hip_atomic_load(&b.dyn[img].n_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g = (nseq + MJ_WG_SEQS - 1) / MJ_WG_SEQS;
        }
        const uint32_t p = (g + 1) / 2;
        uint32_t tg, tp;
        const uint32_t eg = carry_g + block_excl_scan(g, s_w, &tg);
        const uint32_t ep = carry_p + block_excl_scan(p, s_w, &tp);
        /* an image of a few groups is listed by its own thread; one of many (a file without
         * restart markers may have a thousand) by a wavefront, a lane per entry */
        constexpr uint32_t OWN = 48;
        if (g <= OWN) {
            for (uint32_t k = 0; k < g; k++) {
                if (eg + k < b.n_entropy_wg) b.grp_list[eg + k] = make_uint2(img, k);
                if (!(k & 1u) && ep + (k >> 1) < b.n_sync_pairs) b.pair_list[ep + (k >> 1)] = make_uint2(img, k);
            }
        }
        const unsigned long long big = __ballot(g > OWN);
        if (big) {
            s_img[threadIdx.x] = img;
            s_g[threadIdx.x] = g;
            s_eg[threadIdx.x] = eg;
            s_ep[threadIdx.x] = ep;
            for (unsigned long long m = big; m; m &= m - 1) {          /* this wavefront's own large images, one after the other */
                const uint32_t t = (threadIdx.x & ~63u) + (uint32_t) __builtin_ctzll(m);
                const uint32_t gi = s_g[t], im_i = s_img[t], eg_i = s_eg[t], ep_i = s_ep[t];
                for (uint32_t k = lane; k < gi; k += 64) {
                    if (eg_i + k < b.n_entropy_wg) b.grp_list[eg_i + k] = make_uint2(im_i, k);
                    if (!(k & 1u) && ep_i + (k >> 1) < b.n_sync_pairs) b.pair_list[ep_i + (k >> 1)] = make_uint2(im_i, k);
                }
            }
        }
        carry_g += tg;
        carry_p += tp;
    }
    if (threadIdx.x == 0) {
        b.multi_count[2] = min(carry_g, b.n_entropy_wg);
        b.multi_count[3] = min(carry_p, b.n_sync_pairs);
    }
}

}  // namespace

hipError_t mj_launch_scan(const MjDeviceBatch &b, hipStream_t s) {
    if (b.n_tiles == 0 || b.n_img == 0) {
        /* nothing to scan: the counters the first scan kernel would have cleared */
        hipError_t e = hipMemsetAsync(b.stats, 0, MJ_ST_COUNT * sizeof(unsigned long long), s);
        if (e == hipSuccess) e = hipMemsetAsync(b.multi_count, 0, 16, s);
        return e;
    }
    hipLaunchKernelGGL(mj_k_scan_count, dim3(b.n_tiles), dim3(TPB), 0, s, b);
    hipLaunchKernelGGL(mj_k_scan_prefix, dim3(b.n_img), dim3(TPB), 0, s, b);
    hipLaunchKernelGGL(mj_k_scan_write, dim3(b.n_tiles), dim3(TPB), 0, s, b);
    hipLaunchKernelGGL(mj_k_layout, dim3(b.n_img), dim3(TPB), 0, s, b);
    return hipGetLastError();
}
