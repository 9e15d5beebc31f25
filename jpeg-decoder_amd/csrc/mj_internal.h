/*
 * mj_internal.h -- what the C-ABI layer (mj_abi.hip) and the kernel files share:
 * the per-batch set of device buffers and one launch function per stage.
 */
#ifndef MJ_INTERNAL_H
#define MJ_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mj_device.h"

/* per scan tile (MJ_SCAN_TILE raw bytes) */
struct MjTileCnt {
    uint32_t kept;   /* bytes that survive unstuffing, before the terminator */
    uint32_t rst;    /* RSTn markers */
    uint32_t term;   /* raw offset (in the image's segment) of the first terminating marker, or MJ_NO_TERMINATOR */
    uint32_t pad_;
};
struct MjTileBase {
    uint32_t kept;   /* compacted bytes before this tile */
    uint32_t rst;    /* RSTn markers before this tile */
};

/* entropy statistics (device counters) */
enum { MJ_ST_SUBS = 0, MJ_ST_ROUNDS, MJ_ST_MAXROUNDS, MJ_ST_SEQS, MJ_ST_MULTI, MJ_ST_FIXITERS, MJ_ST_TASKS, MJ_ST_WGS, MJ_ST_COUNT = 8 };

/* sequences (wavefronts) per workgroup of the synchronisation kernel; the writing
 * kernel runs two workgroups of half that per group */
#define MJ_WG_SEQS 8

struct MjDeviceBatch {
    uint32_t n_img;
    uint32_t n_tiles;        /* scan tiles over the whole batch */
    uint32_t n_entropy_wg;   /* upper bound on the groups of MJ_WG_SEQS sequences (the host knows how many an image has at most) */
    uint32_t n_multi_cap;    /* capacity of the multi-sequence interval list */
    uint32_t n_idct_tiles;   /* tiles of the dequant+IDCT+colour kernel (one wavefront works on a tile) */
    uint32_t want_planes;
    uint32_t rgba;           /* 1: four bytes per pixel (R, G, B, 255) */
    uint32_t max_ntab;       /* largest Huffman table set in the batch (sizes the kernels' LDS) */
    uint32_t max_nac;        /* most AC tables in one set */
    uint32_t dbg;            /* always 0 in the product; see MJ_DBG */

    const MjImageDev *img;   /* [n_img] */
    MjImageDyn *dyn;         /* [n_img] */
    const MjHuffSet *sets;
    const uint8_t *raw;
    uint8_t *cmp;
    const uint32_t *tile_img;   /* [n_tiles] image of each scan tile */
    MjTileCnt *tile_cnt;
    MjTileBase *tile_base;
    uint32_t *ivl_start;     /* per image n_ivl+1 byte offsets into its compacted stream */
    uint32_t *ivl_seq;       /* per interval: image-local index of its first sequence */
    uint32_t *seq_ivl;       /* per sequence slot: interval (image-local) */
    uint32_t *seq_sub;       /* per sequence slot: first subsequence within the interval */
    MjSubRec *rec;           /* per image from rec_off: 64 * lane_subs subsequence records per sequence slot */
    MjSeqEntry *seq_entry;   /* per sequence slot */
    MjState *seq_seen;       /* per sequence slot: entry state the fix-up last decoded from */
    uint32_t *seq_flag;      /* per sequence slot: scratch for the fix-up */
    uint2 *multi;            /* (image, interval) of intervals longer than one sequence */
    uint32_t *multi_count;   /* four counters, cleared before every decode: [0] entries in `multi`, [1] workgroups of the layout kernel
                                that are through, [2] groups of sequences that exist (entries in grp_list), [3] entries in pair_list */
    const uint32_t *img_order; /* [n_img] the images in the order their groups of sequences are listed (longest lanes first) */
    uint2 *grp_list;         /* [n_entropy_wg at most] (image, group within the image) of every group of MJ_WG_SEQS sequences that exists,
                                written by the layout kernel: block i of an entropy grid takes entry i */
    uint2 *pair_list;        /* [n_sync_pairs at most] (image, first group) of every two groups of one image (its last may be a single one) */
    uint32_t n_sync_pairs;   /* upper bound, like n_entropy_wg */
    uint32_t n_busy_groups;  /* lower bound on the workgroups that will hold work: per image the larger of its restart intervals and its bytes over a sequence's, in eights */
    /* speculative synchronisation (mj_spec_core.h), for batches that leave the chip idle: every image has the same lane_subs then */
    uint32_t spec;           /* 0: the rounds of mj_k_entropy_sync; else X, the subsequences a lane decodes past its own */
    uint32_t spec_stride;    /* records per lane slot in spec_rec: most block slots per MCU in the batch x (lane_subs + X) */
    uint32_t spec_bpm;       /* most block slots per MCU in the batch */
    uint32_t spec_g;         /* the lane_subs every image of the batch has then */
    MjSubRec *spec_rec;      /* per lane slot (rec_off / lane_subs + 64 x sequence + lane), variant and subsequence of the trajectory */
    MjState *spec_state;     /* the exit states of spec_rec once more, [variant][subsequence of the trajectory][lane slot] */
    uint32_t spec_lanes;     /* lane slots in the batch */
    uint8_t *spec_map;       /* per record slot: MJ_SPEC_MAXC bytes, candidate at the first position of the slot's lane -> candidate here */
    uint8_t *spec_lane;      /* per lane slot: MJ_SPEC_MAXC bytes, candidate at the sequence's first position -> candidate at the lane's first */
    uint8_t *spec_tot;       /* per sequence slot: MJ_SPEC_MAXC bytes, what the whole sequence does to a candidate */
    uint32_t *spec_brk;      /* per sequence slot: hops of the true chain in it that were guesses; bit 31: the boundary before it is to be decoded across again */
    uint8_t *spec_carry;     /* per sequence slot, intervals of many sequences only: the candidate the true chain enters the sequence at */
    uint32_t spec_long;      /* some interval may have so many sequences that mj_k_spec_carry is launched */
    const MjIdctImg *idct_img; /* [n_img] what the IDCT + colour kernel reads per image */
    const MjCombine *comb;   /* [n_comb] frames coded in one scan per component */
    uint32_t n_comb, comb_max_x, comb_max_y;
    int16_t *coefs;
    uint8_t *planes;
    uint8_t *rgb;
    unsigned long long *stats; /* [MJ_ST_COUNT] */
};

/* Timing experiments (skip a pass, cap the rounds) exist only in builds with
 * -DMJ_EXPERIMENT; they produce wrong output by design.  In the product build MJ_DBG is
 * the constant 0 and the branches disappear. */
#if defined(MJ_EXPERIMENT)
#define MJ_DBG(b) ((b).dbg)
#else
#define MJ_DBG(b) 0u
#endif

/* Stage launchers; all asynchronous on `s`.  Return hipSuccess or the launch error. */
hipError_t mj_launch_scan(const MjDeviceBatch &b, hipStream_t s);
hipError_t mj_launch_entropy(const MjDeviceBatch &b, hipStream_t s);
hipError_t mj_launch_spec(const MjDeviceBatch &b, hipStream_t s);
hipError_t mj_launch_entropy_fix(const MjDeviceBatch &b, hipStream_t s);
hipError_t mj_launch_entropy_write(const MjDeviceBatch &b, hipStream_t s);
hipError_t mj_launch_entropy_serial(const MjDeviceBatch &b, hipStream_t s);
hipError_t mj_launch_idct_colour(const MjDeviceBatch &b, hipStream_t s);
hipError_t mj_launch_combine(const MjDeviceBatch &b, hipStream_t s);
hipError_t mj_launch_planes_rgb(const MjCombine *comb, uint32_t n, uint32_t max_x, uint32_t max_y, const uint8_t *planes, uint8_t *rgb,
                                bool rgba, hipStream_t s);

/* Compute units of the current device (asked once per device). */
static inline hipError_t mj_device_cus(uint32_t *out) {
    static int cus[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    int n = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
    if (n == 0) {
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (n <= 0) n = 1;
        __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
    }
    *out = (uint32_t) n;
    return hipSuccess;
}

/* Tile geometry of the IDCT kernel (host + device).  A tile is a run of consecutive MCUs
 * of one MCU row that one wavefront takes from coefficients to RGB: at most
 * MJ_IDCT_TILE_BLOCKS blocks (eight lanes work on a block: three passes of eight blocks). */
#define MJ_IDCT_THREADS 512
#define MJ_IDCT_TILE_BLOCKS 24
#define MJ_IDCT_TILE_MCUS_MAX 16
static inline __host__ __device__ uint32_t mj_idct_tile_mcus(uint32_t bpm) {
    const uint32_t t = MJ_IDCT_TILE_BLOCKS / (bpm ? bpm : 1u);
    return t > MJ_IDCT_TILE_MCUS_MAX ? MJ_IDCT_TILE_MCUS_MAX : (t ? t : 1u);
}

#endif
