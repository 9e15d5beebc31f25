/*
 * mj_abi.hip -- the C ABI declared in include/mi355jpeg.h: device context, batch
 * layout in HBM, uploads, stage launches, result copies and HIP-event timing.  Host code
 * in the reference's language (C) calls only these entry points; nothing HIP-specific
 * appears in their signatures.
 *
 * A batch's device memory is four regions:
 *   meta   what the host computes per batch (image descriptors, Huffman sets, workgroup
 *          -> image maps); mirrored byte for byte in page-locked host memory and sent
 *          with one asynchronous copy;
 *   state  small device-written tables (tile counts, interval and sequence tables,
 *          counters); cleared with one asynchronous memset per upload;
 *   work   the entropy-coded bytes, the unstuffed streams and the subsequence records;
 *          never cleared: every kernel writes what a later one reads;
 *   out    coefficients, RGB and optional planes.
 * mj_batch_create and the pipeline put all four in one allocation per batch (or
 * pipeline slot); mj_decode_static puts the first three in a file-scope device array
 * and takes `out` from the caller.
 */
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <algorithm>
#include <system_error>
#include <thread>
#include <vector>

#include "../host/mj_host.h"
#include "mi355jpeg.h"
#include "mj_internal.h"
#include "mj_spec_core.h"

/* Work memory of mj_decode_static: one copy per GPU, held from the moment the library's
 * code is loaded on that GPU until the process ends.  In host code this name is only a
 * handle for hipGetSymbolAddress (see mj_create); kernels get the address as an argument. */
static __device__ __align__(256) unsigned char scratch[MJ_STATIC_SCRATCH_BYTES];

namespace {

thread_local char g_err[512] = "";

int set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            return set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                  \
    } while (0)

uint64_t align_up(uint64_t v, uint64_t a) {
    return (v + a - 1) / a * a;
}

/* One decode call's stage boundaries: the interval ev[i]..ev[i+1] is stage i in
 * launch order, filed under kernel slot kSlotOf[i]. */
struct EventSet {
    hipEvent_t ev[MJ_T_COUNT + 1];
};
constexpr int kSlotOf[MJ_T_COUNT] = {MJ_T_MEMSET, MJ_T_SCAN, MJ_T_ENTROPY, MJ_T_ENTROPY_FIX, MJ_T_IDCT_COLOUR};

/* Upload streams per context.  (Spreading the per-image copies of a batch over four
 * streams was measured and is slower than one: 4.5 against 4.1 ms per 64 images of 2 MB.) */
constexpr int kCopyStreams = 1;

constexpr size_t kStaticMetaBytes = 4u << 20;   /* page-locked mirror of the meta region of mj_decode_static */

/* How batches choose between the rounds and the speculative synchronisation (mj_set_sync_policy). */
#ifndef MJ_SPEC_MAX_WORK_BITS_PER_CU
#define MJ_SPEC_MAX_WORK_BITS_PER_CU 4000000u   /* 1.0 Gbit of counting decode on 256 CUs, 0.3 ms: where the rounds take as long (128 files of 512 x 512, or one of 7680 x 4320) */
#endif
#ifndef MJ_SPEC_LONG_INTERVAL_BITS
#define MJ_SPEC_LONG_INTERVAL_BITS 262144u
#endif
#ifndef MJ_SPEC_MAX_BITS_PER_BLOCK
#define MJ_SPEC_MAX_BITS_PER_BLOCK 160u
#endif
#ifndef MJ_SPEC_MAX_BITS_PER_BLOCK_SUBSAMPLED
#define MJ_SPEC_MAX_BITS_PER_BLOCK_SUBSAMPLED 280u     /* MCUs of four blocks or more */
#endif
#ifndef MJ_SPEC_FINE_BITS_PER_CU
#define MJ_SPEC_FINE_BITS_PER_CU 10000u              /* batches up to this size get subsequences finer than MJ_SPEC_SUB_BITS */
#endif
#ifndef MJ_SPEC_SUB_BITS_FINE
#define MJ_SPEC_SUB_BITS_FINE 160u
#endif
#ifndef MJ_SPEC_SUB_BITS
#define MJ_SPEC_SUB_BITS 256u
#endif
/* Subsequences a lane decodes past its own, by lane_subs: what matters is that a lane's
 * variants have some 768 bits in which to fall into step before the trajectory of the lane
 * before them ends, and the more lanes start, the more chances there are (counted on 512 x 512
 * files at 2 bit per pixel: lanes of one subsequence and three more never had to guess, lanes
 * of two and three more once in four files, lanes of four and four more once or twice a file). */
static inline uint32_t spec_default_extra(uint32_t lane_subs) {
    return lane_subs <= 2 ? 3u : 4u;
}
struct SyncPolicy {
    uint32_t mode = 0, sub_bits = 0, lane_subs = 0, extra = 0;
} g_sync;

/* offsets of every device buffer inside its region (all multiples of 256) */
struct Offsets {
    /* meta */
    size_t img, sets, tile_img, img_order, idct_img, comb, meta_bytes;
    /* state */
    size_t dyn, tile_cnt, tile_base, ivl_start, ivl_seq, seq_ivl, seq_sub, seq_entry, seq_seen, seq_flag, multi, multi_count,
        grp_list, pair_list, stats, state_bytes;
    /* work */
    size_t raw, cmp, rec, spec_rec, spec_state, spec_map, spec_lane, spec_tot, spec_brk, spec_carry, work_bytes;
    /* out */
    size_t coefs, rgb, planes, out_bytes;
};

struct Carve {
    size_t off = 0;
    size_t take(uint64_t bytes) {
        const size_t o = off;
        off += (size_t) align_up(bytes ? bytes : 16, 256);
        return o;
    }
};

}  // namespace

struct mj_ctx {
    int device;
    hipStream_t own = nullptr;   /* the context's compute stream */
    hipStream_t copy[kCopyStreams];   /* uploads; [0] also carries the memset and the descriptors */
    hipStream_t stream;   /* where decodes are enqueued: own, or the caller's (mj_set_stream) */
    /* page-locked staging for entropy-coded bytes handed over in pageable memory and for
     * result read-back; kept and grown from one batch to the next */
    uint8_t *stage = nullptr;
    size_t stage_cap = 0;
    /* the device memory and the page-locked meta mirror of the last batch destroyed, for the
     * next one to move into: a caller that decodes batch after batch (or image after image
     * through the drop-in entry points) allocates once */
    void *arena = nullptr;
    size_t arena_cap = 0;
    uint8_t *meta_host = nullptr;
    size_t meta_cap = 0;
    /* mj_planes_to_rgb: descriptor + planes + RGB of one frame, kept and grown from call to call */
    uint8_t *conv = nullptr;
    size_t conv_cap = 0;
    /* mj_decode_static: this device's copy of `scratch`, and the one batch object that lives in it */
    uint8_t *scratch_dev = nullptr;
    mj_batch *sbatch = nullptr;
};

/* A three-component frame coded in several scans (T.81 A.2.2, A.2.3: one component each, or one
 * alone and two interleaved): every scan is decoded as an image of its own (a "unit": grey for a
 * scan of one component, two components for a scan of two), and a last kernel combines the
 * planes into the RGB of the file.  The unit that holds component 0 sits at the file's own
 * index, the others after all the files. */
struct MjMultiScan {
    uint32_t file;
    uint32_t n_scans;        /* 2 or 3 */
    uint32_t scan_unit[3];   /* unit of every scan, in the order of the scans in the file */
    uint32_t unit[3];        /* by frame component: the unit that codes it ... */
    uint32_t unit_comp[3];   /* ... and which of the unit's components it is */
    uint32_t hsf[3], vsf[3];
    uint32_t X, Y;
};

struct mj_batch {
    mj_ctx *ctx = nullptr;
    uint32_t flags = 0;
    std::vector<MjImageDev> img;
    std::vector<mj_image_info> info;
    std::vector<MjHuffSet> sets;
    std::vector<uint32_t> tile_img, img_order;
    uint32_t n_pairs_cap = 0;           /* upper bound on the pairs of sequence groups (one image's, two at a time) */
    uint32_t n_busy_groups = 0;
    std::vector<MjIdctImg> idct_img;
    size_t n_files = 0;                /* img, info, idct_img also hold the extra units of files with several scans */
    std::vector<MjMultiScan> multi;
    std::vector<MjCombine> comb;
    Offsets o{};
    uint64_t raw_bytes = 0;
    const uint8_t *span_src = nullptr;   /* non-null: the images' bytes lie close together in page-locked memory and leave as one copy from here */
    uint64_t span_bytes = 0;
    uint8_t *arena = nullptr;      /* meta | state | work | out, one allocation (null for the static batch) */
    size_t arena_cap = 0;
    uint8_t *meta_host = nullptr;  /* page-locked mirror of the meta region */
    size_t meta_cap = 0;
    uint8_t *stage = nullptr;      /* pipeline slots: own staging for pageable input (a one-shot batch uses the context's) */
    size_t stage_cap = 0;
    bool borrowed = false;         /* static batch: no device memory of its own to free */
    bool owned_by_pipe = false;    /* a pipeline slot: only mj_pipe_destroy frees it */
    MjDeviceBatch d{};
    uint64_t coef_bytes = 0, rgb_bytes = 0, plane_bytes = 0;
    hipStream_t stream = nullptr;  /* the stream the last decode was enqueued on */
    hipEvent_t ev_up[kCopyStreams] = {nullptr}, ev_done = nullptr;
    int n_up = 0;                  /* copy streams the last upload used: ev_up[0 .. n_up) were recorded */
    bool uploaded_async = false;   /* those events are pending */
    bool decoded = false;          /* ev_done has been recorded */
    std::vector<EventSet> pending, pool;
    float ms[MJ_T_COUNT] = {0};
    int launches[MJ_T_COUNT] = {0};
};

struct mj_pipe {
    mj_ctx *ctx = nullptr;
    uint32_t flags = 0;
    std::vector<mj_batch *> slots;
    hipStream_t alt = nullptr;         /* second decode stream: consecutive batches alternate (see mj_pipe_submit) */
    uint64_t next = 0;
    double wait_ms = 0, host_ms = 0;   /* per submit: waiting for the slot; parse + layout + enqueue */
};

/* One image ready for placement: flattened descriptor, which of the call's Huffman sets
 * it uses, and its entropy bytes. */
struct MjPrepared {
    MjImageDev im;
    uint32_t set;
    const uint8_t *ent;
};

/* The images of one call and the distinct Huffman sets among them (mostly one: an
 * encoder's default tables).  A set is built for the first image with a given key only. */
struct MjPrep {
    size_t n_files = 0;
    std::vector<MjMultiScan> multi;
    std::vector<MjPrepared> img;
    std::vector<MjHuffSet> sets;
    std::vector<uint8_t> keys;       /* MJ_HUFF_KEY_BYTES per set */
    std::vector<size_t> key_len;
};

namespace {

int host_reserve(uint8_t **buf, size_t *cap, size_t bytes, const char *what) {
    if (*cap >= bytes) return 0;
    if (*buf) (void) hipHostFree(*buf);
    *buf = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc(reinterpret_cast<void **>(buf), want, hipHostMallocDefault) != hipSuccess) {
        *buf = nullptr;
        return set_err("cannot allocate %zu bytes of page-locked memory for %s", want, what);
    }
    *cap = want;
    return 0;
}

int collect_timing(mj_batch *b) {
    if (b->pending.empty()) return 0;
    if (b->stream) HIP_TRY(hipStreamSynchronize(b->stream));
    for (EventSet &es : b->pending) {
        for (int t = 0; t < MJ_T_COUNT; t++) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, es.ev[t], es.ev[t + 1]));
            b->ms[kSlotOf[t]] += ms;
            b->launches[kSlotOf[t]] += 1;
        }
        b->pool.push_back(es);
    }
    b->pending.clear();
    return 0;
}

mj_batch *batch_new(mj_ctx *ctx) {
    mj_batch *b = new mj_batch();
    b->ctx = ctx;
    bool ok = hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < kCopyStreams && ok; i++) ok = hipEventCreateWithFlags(&b->ev_up[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        for (hipEvent_t e : b->ev_up) {
            if (e) (void) hipEventDestroy(e);
        }
        if (b->ev_done) (void) hipEventDestroy(b->ev_done);
        delete b;
        set_err("cannot create HIP events");
        return nullptr;
    }
    return b;
}

/* wait for everything enqueued for the batch: its upload and its last decode */
int batch_wait(mj_batch *b) {
    if (b->uploaded_async) {
        for (int i = 0; i < b->n_up; i++) HIP_TRY(hipEventSynchronize(b->ev_up[i]));
        b->uploaded_async = false;
    }
    if (b->decoded) HIP_TRY(hipEventSynchronize(b->ev_done));
    return 0;
}

/* index of the set with this key among those of the call, or -1 */
int find_set(const MjPrep &pp, const uint8_t *key, size_t len) {
    for (size_t i = 0; i < pp.key_len.size(); i++) {
        if (pp.key_len[i] == len && memcmp(&pp.keys[i * MJ_HUFF_KEY_BYTES], key, len) == 0) return (int) i;
    }
    return -1;
}

/* one scan's parsed headers -> descriptor, building its Huffman set unless the call has it */
int prepare_one(MjPrep &pp, MjPrepared &dst, const MjParsed &P, size_t len_after_headers_base, const char **why) {
    uint8_t key[MJ_HUFF_KEY_BYTES];
    const size_t klen = mj_huff_key(&P, key);
    int si = find_set(pp, key, klen);
    if (si >= 0) {
        if (mj_flatten_image(&P, len_after_headers_base, &dst.im, nullptr, why) != 0) return -1;
    } else {
        pp.sets.emplace_back();
        if (mj_flatten_image(&P, len_after_headers_base, &dst.im, &pp.sets.back(), why) != 0) {
            pp.sets.pop_back();
            return -1;
        }
        si = (int) pp.sets.size() - 1;
        pp.keys.resize(pp.sets.size() * MJ_HUFF_KEY_BYTES);
        memcpy(&pp.keys[(size_t) si * MJ_HUFF_KEY_BYTES], key, klen);
        pp.key_len.push_back(klen);
    }
    dst.set = (uint32_t) si;
    return 0;
}

/* A file whose first scan holds one or two of its three components: every scan becomes a unit
 * (decode.c:187-201 re-enters the scan loop for every SOS; the reference's own bookkeeping does
 * not survive that -- its MCU counter runs on, decode.c:125, :473, :504 -- so the behaviour
 * here is T.81's).  A scan of one component codes that component's own blocks in raster order
 * (A.2.2): a grey image of the component's size.  A scan of two codes the frame's MCUs with
 * those components' blocks only (A.2.3): an image of two components with the frame's sampling
 * factors, as large as the larger of the two. */
int prepare_scans(MjPrep &pp, size_t i, const uint8_t *buf, size_t len, MjParsed &P, const char **why, bool &freed) {
    const FrameHeader &fh = P.fh;
    if (fh.ncs != 3 || fh.precision != 8) {
        *why = "several scans: only three-component 8-bit frames are supported";
        return -1;
    }
    uint32_t hmax = 0, vmax = 0;
    for (int c = 0; c < 3; c++) {
        hmax = fh.cs[c].hsf > hmax ? fh.cs[c].hsf : hmax;
        vmax = fh.cs[c].vsf > vmax ? fh.cs[c].vsf : vmax;
    }
    if (fh.cs[0].hsf != hmax || fh.cs[0].vsf != vmax) {
        *why = "several scans: the first component must have the largest sampling factors";
        return -1;
    }
    MjMultiScan mm{};
    mm.file = (uint32_t) i;
    mm.X = fh.X;
    mm.Y = fh.Y;
    MjPrepared part[3];
    int scan_of[3] = {-1, -1, -1};      /* by component: the scan that codes it */
    int n_seen = 0, n_scans = 0;
    while (n_seen < 3) {
        const int nics = P.sh.nics;
        if (nics < 1 || nics > 2 || n_scans == 3) {
            *why = "several scans: a scan must hold one or two components that no earlier scan has coded";
            return -1;
        }
        Component sub[2];
        uint32_t uh = 0, uv = 0;
        for (int s = 0; s < nics; s++) {
            int ci = -1;
            for (int c = 0; c < 3; c++) {
                if (fh.cs[c].id == P.sh.ics[s].sc) ci = c;
            }
            if (ci < 0 || scan_of[ci] >= 0) {
                *why = "several scans: a scan names an unknown component or one already coded";
                return -1;
            }
            if (fh.cs[ci].hsf == 0 || fh.cs[ci].vsf == 0 || hmax % fh.cs[ci].hsf || vmax % fh.cs[ci].vsf) {
                *why = "sampling factors that do not divide the maximum are not supported";
                return -1;
            }
            scan_of[ci] = n_scans;
            n_seen++;
            mm.unit_comp[ci] = (uint32_t) s;
            mm.hsf[ci] = fh.cs[ci].hsf;
            mm.vsf[ci] = fh.cs[ci].vsf;
            sub[s] = fh.cs[ci];
            uh = sub[s].hsf > uh ? sub[s].hsf : uh;
            uv = sub[s].vsf > uv ? sub[s].vsf : uv;
        }
        MjParsed Q = P;                    /* shares the tables' heap data with P, which outlives it */
        Q.fh.cs = sub;
        Q.fh.ncs = (uint8_t) nics;
        if (nics == 1) {
            /* the scan as a grey file of the component's own size */
            sub[0].hsf = sub[0].vsf = 1;
            Q.fh.X = sub[0].x;
            Q.fh.Y = sub[0].y;
        } else {
            Q.fh.X = (uint16_t) (((uint32_t) fh.X * uh + hmax - 1) / hmax);
            Q.fh.Y = (uint16_t) (((uint32_t) fh.Y * uv + vmax - 1) / vmax);
        }
        const size_t ent_end = mj_scan_end(buf, len, P.ent_pos);
        if (prepare_one(pp, part[n_scans], Q, ent_end, why) != 0) return -1;
        part[n_scans].ent = buf + P.ent_pos;
        n_scans++;
        if (n_seen < 3) {
            const int r = mj_parse_next_scan(buf, len, ent_end, &P, why);
            if (r != 0) {
                if (r == 1) *why = "several scans: the file ends before every component was coded";
                freed = (r < 0);                 /* a rejecting parser has freed P */
                return -1;
            }
        }
    }
    /* the unit of component 0 takes the file's place, the others go to the end of the list */
    uint32_t unit_of_scan[3] = {0, 0, 0};
    for (int sc = 0; sc < n_scans; sc++) {
        if (sc == scan_of[0]) {
            pp.img[i] = part[sc];
            unit_of_scan[sc] = (uint32_t) i;
        } else {
            unit_of_scan[sc] = (uint32_t) pp.img.size();
            pp.img.push_back(part[sc]);
        }
        mm.scan_unit[sc] = unit_of_scan[sc];
    }
    mm.n_scans = (uint32_t) n_scans;
    for (int c = 0; c < 3; c++) mm.unit[c] = unit_of_scan[scan_of[c]];
    pp.multi.push_back(mm);
    return 0;
}

/* the cut of every image once more, now that the size of the batch is known */
int choose_for_batch(MjPrep &pp, const char *who) {
    uint32_t cus = 0;
    if (mj_device_cus(&cus) != hipSuccess) return set_err("%s: cannot query the device", who);
    uint64_t batch_bits = 0;
    for (const MjPrepared &u : pp.img) batch_bits += (uint64_t) u.im.raw_len * 8;
    for (MjPrepared &u : pp.img) {
        mj_choose_split(u.im.raw_len, u.im.n_ivl, u.im.n_mcu, batch_bits, cus * 4u * MJ_SEQ_LANES, &u.im.sub_bits, &u.im.lane_subs);
    }
    return 0;
}

/* one file of a call: parsed and flattened into pp.img[i]; a file of several scans is only
 * recognised here when `defer` is given (its units are appended to pp.img, which is not for
 * several threads at once) and its index left there */
int prepare_file(MjPrep &pp, MjPrepared &dst, size_t i, const uint8_t *buf, size_t len, bool *defer, MjPrep *units_into,
                 char *msg, size_t msg_cap) {
    MjParsed P;
    const char *why = "?";
    if (!buf) {
        snprintf(msg, msg_cap, "null buffer");
        return -1;
    }
    if (mj_parse_headers(buf, len, &P, &why) != 0) {
        snprintf(msg, msg_cap, "%s", why);       /* the parser freed what it had */
        return -1;
    }
    if (P.sh.nics < P.fh.ncs && P.fh.ncs == 3) {
        if (defer) {
            *defer = true;
            mj_parsed_free(&P);
            return 0;
        }
        bool freed = false;
        if (prepare_scans(*units_into, i, buf, len, P, &why, freed) != 0) {
            if (!freed) mj_parsed_free(&P);
            snprintf(msg, msg_cap, "%s", why);
            return -1;
        }
        mj_parsed_free(&P);
        return 0;
    }
    if (P.fh.ncs != 1 && P.fh.ncs != 3) {
        mj_parsed_free(&P);
        snprintf(msg, msg_cap, "only grey and three-component images are supported");
        return -1;
    }
    if (prepare_one(pp, dst, P, len, &why) != 0) {
        mj_parsed_free(&P);
        snprintf(msg, msg_cap, "%s", why);
        return -1;
    }
    dst.ent = buf + P.ent_pos;
    mj_parsed_free(&P);
    return 0;
}

/* host: headers -> descriptors (decode.c:160-415 marker walk + table flattening).  Files are
 * independent, and a file with Huffman tables of its own costs tens of microseconds (the
 * lookup tables are built from them): a call of many files is spread over a few threads,
 * each with the table sets it meets, which are merged by key afterwards; files of several
 * scans, whose units go to the end of the list, are taken last, in order, by the caller's
 * thread. */
constexpr int kPrepareThreads = 4;
constexpr int kPrepareMinFilesPerThread = 16;

int prepare(const uint8_t *const *bufs, const size_t *lens, int n, MjPrep &pp, const char *who) {
    pp.img.resize(n);
    pp.n_files = (size_t) n;
    char msg[256];
    const int n_thr = std::min(kPrepareThreads, n / kPrepareMinFilesPerThread);
    if (n_thr < 2) {
        for (int i = 0; i < n; i++) {
            if (prepare_file(pp, pp.img[i], (size_t) i, bufs[i], lens[i], nullptr, &pp, msg, sizeof msg) != 0) {
                return set_err("%s: image %d: %s", who, i, msg);
            }
        }
        return choose_for_batch(pp, who);
    }
    struct Share {
        MjPrep local;                 /* the table sets this thread met; img unused */
        std::vector<int> deferred;    /* files of several scans */
        int bad = -1;                 /* first file that was refused */
        char msg[256];
    };
    std::vector<Share> share((size_t) n_thr);
    auto work = [&](int t) {
        Share &sh = share[(size_t) t];
        MjPrep &sets_of = t == 0 ? pp : sh.local;     /* the caller's thread fills the call's own list: nothing to merge */
        const int i0 = (int) ((int64_t) n * t / n_thr), i1 = (int) ((int64_t) n * (t + 1) / n_thr);
        for (int i = i0; i < i1; i++) {
            bool defer = false;
            if (prepare_file(sets_of, pp.img[i], (size_t) i, bufs[i], lens[i], &defer, nullptr, sh.msg, sizeof sh.msg) != 0) {
                sh.bad = i;
                return;
            }
            if (defer) sh.deferred.push_back(i);
        }
    };
    {
        std::vector<std::thread> pool;
        pool.reserve((size_t) n_thr);
        for (int t = 1; t < n_thr; t++) {
            try {
                pool.emplace_back([&work, t]() {
                    work(t);
                    mj_flatten_thread_done();      /* the thread ends here: its table cache with it */
                });
            } catch (const std::system_error &) {
                work(t);                           /* no thread to be had: the caller's does the share */
            }
        }
        work(0);
        for (std::thread &th : pool) th.join();
    }
    for (const Share &sh : share) {        /* shares are in file order: the first refusal is reported */
        if (sh.bad >= 0) return set_err("%s: image %d: %s", who, sh.bad, sh.msg);
    }
    /* one list of sets: those of every share looked up by key, new ones appended */
    for (int t = 1; t < n_thr; t++) {
        Share &sh = share[(size_t) t];
        std::vector<uint32_t> global(sh.local.sets.size());
        for (size_t k = 0; k < sh.local.sets.size(); k++) {
            const uint8_t *key = &sh.local.keys[k * MJ_HUFF_KEY_BYTES];
            const size_t klen = sh.local.key_len[k];
            int si = find_set(pp, key, klen);
            if (si < 0) {
                pp.sets.push_back(sh.local.sets[k]);
                si = (int) pp.sets.size() - 1;
                pp.keys.resize(pp.sets.size() * MJ_HUFF_KEY_BYTES);
                memcpy(&pp.keys[(size_t) si * MJ_HUFF_KEY_BYTES], key, klen);
                pp.key_len.push_back(klen);
            }
            global[k] = (uint32_t) si;
        }
        const int i0 = (int) ((int64_t) n * t / n_thr), i1 = (int) ((int64_t) n * (t + 1) / n_thr);
        size_t d = 0;
        for (int i = i0; i < i1; i++) {
            if (d < sh.deferred.size() && sh.deferred[d] == i) {
                d++;
                continue;
            }
            pp.img[i].set = global[pp.img[i].set];
        }
    }
    for (int t = 0; t < n_thr; t++) {
        for (int i : share[(size_t) t].deferred) {
            if (prepare_file(pp, pp.img[i], (size_t) i, bufs[i], lens[i], nullptr, &pp, msg, sizeof msg) != 0) {
                return set_err("%s: image %d: %s", who, i, msg);
            }
        }
    }
    return choose_for_batch(pp, who);
}

/* host: where every image and every table goes; no HIP call */
/* Do the entropy-coded segments lie in ascending order inside one stretch of page-locked
 * memory with little else between them (the files of a batch loaded back to back)?  Then
 * the stretch is copied whole -- one DMA instead of one per image, each of which costs
 * about as much again in set-up as it takes to move 2 MB -- and every image is read where
 * it lands: the scan kernels accept any byte alignment. */
bool find_span(const std::vector<MjPrepared> &prep, const uint8_t **src, uint64_t *bytes) {
    if (prep.size() < 2) return false;
    uint64_t sum = 0;
    for (size_t i = 0; i < prep.size(); i++) {
        sum += prep[i].im.raw_len;
        if (i && prep[i].ent < prep[i - 1].ent + prep[i - 1].im.raw_len) return false;
    }
    const uint8_t *first = prep.front().ent, *last = prep.back().ent + prep.back().im.raw_len;
    const uint64_t span = (uint64_t) (last - first);
    if (span > sum + sum / 16 + 4096 * prep.size() || span >= (1ull << 40)) return false;
    hipPointerAttribute_t a, z;
    memset(&a, 0, sizeof a);
    memset(&z, 0, sizeof z);
    if (hipPointerGetAttributes(&a, first) != hipSuccess || hipPointerGetAttributes(&z, last - 1) != hipSuccess) {
        (void) hipGetLastError();
        return false;
    }
    if (a.type != hipMemoryTypeHost || z.type != hipMemoryTypeHost || a.hostPointer == nullptr || z.hostPointer == nullptr) return false;
    /* one allocation: the distance between the two host addresses equals that between their device views */
    if ((const uint8_t *) z.devicePointer - (const uint8_t *) a.devicePointer != (last - 1) - first) return false;
    *src = first;
    *bytes = span;
    return true;
}

/* the IDCT + colour kernel's record of an image whose placement (coef_off, rgb_off, pitch,
 * idct_tiles_x, idct_tile_mcus) is final */
MjIdctImg idct_record(const MjImageDev &im, uint32_t first_tile) {
    MjIdctImg r{};
    r.coef_off = im.coef_off;
    r.rgb_off = im.rgb_off;
    r.X = im.X;
    r.Y = im.Y;
    r.pitch = im.pitch;
    r.mcus_x = im.mcus_x;
    r.geo = im.bpm | (im.idct_tile_mcus << 8) | (im.hmax << 16) | (im.vmax << 24);
    uint32_t first[3] = {0, 0, 0};
    for (uint32_t i = im.bpm; i-- > 0;) {
        r.slot_comp |= (uint32_t) (im.blk_comp[i] & 3u) << (2 * i);
        if (im.blk_comp[i] < 3) first[im.blk_comp[i]] = i;
    }
    for (uint32_t c = 0; c < im.ncs && c < 3; c++) r.sf |= (im.comp[c].hsf | (im.comp[c].vsf << 4)) << (8 * c);
    uint32_t cl = MJ_CL_GENERIC;
    if (im.ncs == 1) {
        cl = MJ_CL_GREY;
    } else if (im.ncs == 3 && im.comp[1].hsf == 1 && im.comp[1].vsf == 1 && im.comp[2].hsf == 1 && im.comp[2].vsf == 1 &&
               im.comp[0].hsf == im.hmax && im.comp[0].vsf == im.vmax && im.hmax != 3 && im.vmax <= 2) {
        cl = MJ_CL_FAST + (im.hmax == 4 ? 2u : im.hmax - 1u) + 3u * (im.vmax - 1u);
    }
    r.cls = cl | (first[0] << 8) | (first[1] << 16) | (first[2] << 24);
    r.first_tile = first_tile;
    r.n_tiles = im.mcus_y * im.idct_tiles_x;
    r.tiles_x = im.idct_tiles_x;
    return r;
}

int layout(mj_batch *b, MjPrep &pp, uint32_t flags, bool may_span, bool may_spec) {
    const std::vector<MjPrepared> &prep = pp.img;
    const int n = (int) prep.size();
    b->span_src = nullptr;
    b->span_bytes = 0;
    if (may_span && !find_span(prep, &b->span_src, &b->span_bytes)) b->span_src = nullptr;
    b->flags = flags;
    b->img.resize(n);
    b->info.resize(n);
    b->sets.swap(pp.sets);          /* taken over, not copied: a set is 70 KB and a mixed batch has a hundred */
    b->tile_img.clear();
    b->img_order.clear();
    b->n_pairs_cap = 0;
    b->n_busy_groups = 0;
    b->idct_img.assign((size_t) n, MjIdctImg{});
    b->n_files = pp.n_files;
    b->multi = pp.multi;
    b->comb.clear();

    /* How the subsequences are synchronised.  The rounds are a chain, and in scans without
     * restart markers a long one (some thirty serial decodes, and as many again across the
     * sequences); a batch that leaves most of the chip idle waits for little else.  The
     * speculative form (mj_spec_core.h) decodes every lane once per block slot and a little
     * further -- fifteen to twenty-four times the work for 4:2:0 -- and has no chain.  It is
     * taken where that work is what the chip does in a fraction of a millisecond and the chains
     * are long: most of the batch in intervals of 256 kbit or more (files with a restart
     * interval per MCU row synchronise in a few rounds as they are) and next to none of it in
     * files of very many bits per block, lanes of one subsequence if the work allows and of
     * two otherwise (longer lanes start too seldom: too many guesses). */
    uint32_t spec_x = 0, spec_g = 0, spec_s = 0, spec_bpm = 1;
    bool spec_fine = false;
    if (may_spec && !(flags & MJ_ENTROPY_SERIAL) && g_sync.mode != 1) {
        uint64_t bits = 0, long_bits = 0, dense_bits = 0;
        for (int i = 0; i < n; i++) {
            const MjImageDev &im = prep[i].im;
            bits += im.raw_len * 8;
            if (im.raw_len * 8 >= (uint64_t) MJ_SPEC_LONG_INTERVAL_BITS * im.n_ivl) {
                /* blocks of many bits are blocks that fill up without an end-of-block symbol, which
                 * is where a decoder gets its zigzag index right: there the variants of a lane
                 * stay out of step whatever their slot, and the chain guesses (files of 190 bits
                 * per block: at 0.4 % of the lane boundaries, of 280: at 7 %) */
                const uint64_t blocks = (uint64_t) im.n_mcu * im.bpm;
                /* (measured on 640 x 480 files: 4:4:4 at 190 bits per block is already slower than by rounds,
                 * 4:2:0 at 270 and 4:2:2 at 180 still a third faster) */
                if (im.raw_len * 8 > (uint64_t) (im.bpm <= 3 ? MJ_SPEC_MAX_BITS_PER_BLOCK : MJ_SPEC_MAX_BITS_PER_BLOCK_SUBSAMPLED) * blocks) dense_bits += im.raw_len * 8;
                else long_bits += im.raw_len * 8;
            }
            spec_bpm = im.bpm > spec_bpm ? im.bpm : spec_bpm;
        }
        uint32_t cus = 0;
        if (mj_device_cus(&cus) != hipSuccess) return set_err("cannot query the device");
        const uint64_t limit = (uint64_t) MJ_SPEC_MAX_WORK_BITS_PER_CU * cus;
        const bool chains = spec_bpm >= 2 && 2 * long_bits > bits && 16 * dense_bits <= bits;
        for (uint32_t g = 1; g <= MJ_MAX_LANE_SUBS && !spec_x; g *= 2) {
            if (g_sync.lane_subs ? g != g_sync.lane_subs : (g_sync.mode != 2 && g > 2)) continue;
            const uint32_t x = mj_spec_fit_x(g, spec_bpm, g_sync.extra ? g_sync.extra : spec_default_extra(g));
            if (x == 0) continue;
            const uint64_t work = bits * spec_bpm * (g + x) / g;
            /* lanes of one subsequence only while their work is a third of the limit: past that the
             * half again that lanes of two save is worth more than the guesses they avoid */
            if (g_sync.mode == 2 || (chains && work <= (g == 1 ? limit / 3 : limit))) {
                spec_x = x;
                spec_g = g;
            }
        }
        spec_s = g_sync.sub_bits ? (g_sync.sub_bits + 31u) / 32u * 32u : MJ_SPEC_SUB_BITS;
        /* a file or two on their own: every wavefront has its SIMD to itself and the decode takes as
         * long as one lane does, so shorter lanes, as short as the files' blocks allow (below) */
        spec_fine = bits <= (uint64_t) MJ_SPEC_FINE_BITS_PER_CU * cus;
    }

    uint64_t raw_total = 0, cmp_total = 0, coef_blocks = 0, rgb_total = 0, plane_total = 0;
    uint64_t n_tiles = 0, n_ivl = 0, n_seq = 0, n_rec = 0, n_wg = 0, n_multi = 0, n_idct = 0;
    for (int i = 0; i < n; i++) {
        MjImageDev &im = b->img[i];
        im = prep[i].im;
        if (spec_x) {
            /* a variant falls into step block by block, the sooner the more blocks the lane's few
             * subsequences hold: subsequences of at least two blocks of this file (640 x 480 files
             * of 120 to 130 bits per block guess at one lane boundary in a thousand with 256-bit
             * subsequences, files of 190 at one in thirty, and at one in five hundred with 512;
             * lanes twice as long take twice as long, so no longer than that) */
            uint32_t sb = spec_s;
            const uint64_t blocks = (uint64_t) im.n_mcu * im.bpm;
            const uint64_t block_bits = blocks ? im.raw_len * 8 / blocks : 0;
            if (!g_sync.sub_bits) {
                if (spec_fine && 2u * block_bits <= MJ_SPEC_SUB_BITS) {
                    /* (a 512 x 512 file alone: 187 us with 256 bits, 171 with 192, 165 with 160 -- and 184
                     * with 128, where the guesses begin; eight of them: 203, 211, 224) */
                    sb = (uint32_t) ((2u * block_bits + 31u) / 32u * 32u);
                    sb = sb < MJ_SPEC_SUB_BITS_FINE ? MJ_SPEC_SUB_BITS_FINE : sb;
                } else {
                    while (sb < 2048u && (uint64_t) sb < 2u * block_bits) sb *= 2u;
                }
            }
            im.sub_bits = sb;
            im.lane_subs = spec_g;
        }
        if (im.raw_len >= (1ull << 28)) {
            return set_err("image %d: entropy segment of %llu bytes is too large", i, (unsigned long long) im.raw_len);
        }
        im.huff_set = prep[i].set;      /* images with identical tables share one set */

        if (b->span_src) {
            im.raw_off = (uint64_t) (prep[i].ent - b->span_src);   /* where the image lands: any alignment */
            raw_total = align_up(b->span_bytes + 32, 16);
        } else {
            im.raw_off = raw_total;
            raw_total = align_up(raw_total + im.raw_len + 32, 16);
        }
        im.cmp_off = cmp_total;
        cmp_total = align_up(cmp_total + im.raw_len + MJ_STREAM_PAD, 16);
        im.coef_off = coef_blocks;
        coef_blocks += (uint64_t) im.n_mcu * im.bpm;
        if (flags & MJ_OUT_RGBA) im.pitch = 4 * im.X;
        im.rgb_off = rgb_total;
        rgb_total = align_up(rgb_total + (uint64_t) im.pitch * im.Y, 256);
        for (uint32_t c = 0; c < im.ncs; c++) {
            im.comp[c].plane_off = plane_total;
            plane_total = align_up(plane_total + (uint64_t) im.comp[c].pw * im.comp[c].ph, 256);
        }
        im.tile_off = (uint32_t) n_tiles;
        im.n_tiles = (uint32_t) ((im.raw_len + MJ_SCAN_TILE - 1) / MJ_SCAN_TILE);
        n_tiles += im.n_tiles;
        im.ivl_off = (uint32_t) n_ivl;
        n_ivl += im.n_ivl + 1;
        const uint64_t seq_subs = (uint64_t) MJ_SEQ_LANES * im.lane_subs;
        const uint64_t seq_bits = seq_subs * im.sub_bits;
        im.seq_cap = (uint32_t) ((im.raw_len * 8 + seq_bits - 1) / seq_bits + im.n_ivl);
        im.seq_off = (uint32_t) n_seq;
        n_seq += im.seq_cap;
        im.rec_off = (uint32_t) n_rec;
        n_rec += im.seq_cap * seq_subs;
        uint64_t mc = im.raw_len * 8 / seq_bits;
        im.multi_cap = (uint32_t) (mc < im.n_ivl ? mc : im.n_ivl);
        n_multi += im.multi_cap;
        im.idct_tile_mcus = mj_idct_tile_mcus(im.bpm);
        im.idct_tiles_x = (im.mcus_x + im.idct_tile_mcus - 1) / im.idct_tile_mcus;
        const uint32_t iw = im.mcus_y * im.idct_tiles_x;
        b->idct_img[i] = idct_record(im, (uint32_t) n_idct);
        n_idct += iw;
        b->tile_img.insert(b->tile_img.end(), im.n_tiles, (uint32_t) i);
        {
            const uint64_t by_bits = im.raw_len * 8 / seq_bits;
            const uint64_t seqs_lb = by_bits > im.n_ivl ? by_bits : im.n_ivl;
            b->n_busy_groups += (uint32_t) ((seqs_lb + MJ_WG_SEQS - 1) / MJ_WG_SEQS);
        }
        if (n_tiles >= (1ull << 31) || n_seq >= (1ull << 31) || n_rec >= (1ull << 31) || n_idct >= (1ull << 31) || n_ivl >= (1ull << 31)) {
            return set_err("batch too large");
        }

        mj_image_info &inf = b->info[i];
        memset(&inf, 0, sizeof inf);
        inf.X = (int32_t) im.X;
        inf.Y = (int32_t) im.Y;
        inf.ncs = (int32_t) im.ncs;
        for (uint32_t c = 0; c < im.ncs; c++) {
            inf.hsf[c] = (int32_t) im.comp[c].hsf;
            inf.vsf[c] = (int32_t) im.comp[c].vsf;
            inf.plane_w[c] = (int32_t) im.comp[c].pw;
            inf.plane_h[c] = (int32_t) im.comp[c].ph;
        }
        inf.mcus_x = (int32_t) im.mcus_x;
        inf.mcus_y = (int32_t) im.mcus_y;
        inf.blocks_per_mcu = (int32_t) im.bpm;
        inf.restart_interval = (im.ri == im.n_mcu && im.n_ivl == 1) ? 0 : (int32_t) im.ri;
        inf.n_intervals = (int32_t) im.n_ivl;
        inf.n_blocks = (int64_t) im.n_mcu * im.bpm;
        inf.entropy_bytes = (int64_t) im.raw_len;
        inf.rgb_pitch = im.pitch;
        inf.coef_offset = (int64_t) im.coef_off;
        inf.rgb_offset = (int64_t) im.rgb_off;
    }

    /* frames coded in one scan per component: the file's entry describes the frame, its RGB is
     * where the unit of component 0 (same size) has its own, and the planes are always kept */
    uint32_t comb_max_x = 0, comb_max_y = 0;
    if (b->multi.size() > 65535) return set_err("batch too large: more than 65,535 files coded in several scans");
    for (const MjMultiScan &mm : b->multi) {
        MjCombine cb{};
        mj_image_info &inf = b->info[mm.file];
        inf.ncs = 3;
        inf.n_blocks = 0;
        inf.entropy_bytes = 0;
        inf.n_intervals = 0;
        for (int c = 0; c < 3; c++) {
            const MjImageDev &u = b->img[mm.unit[c]];
            const MjCompDev &uc = u.comp[mm.unit_comp[c]];
            cb.plane_off[c] = uc.plane_off;
            cb.pw[c] = uc.pw;
            cb.hsf[c] = mm.hsf[c];
            cb.vsf[c] = mm.vsf[c];
            inf.hsf[c] = (int32_t) mm.hsf[c];
            inf.vsf[c] = (int32_t) mm.vsf[c];
            inf.plane_w[c] = (int32_t) uc.pw;
            inf.plane_h[c] = (int32_t) uc.ph;
        }
        bool all_single = true;
        for (uint32_t sc = 0; sc < mm.n_scans; sc++) {
            const MjImageDev &u = b->img[mm.scan_unit[sc]];
            inf.n_blocks += (int64_t) u.n_mcu * u.bpm;
            inf.entropy_bytes += (int64_t) u.raw_len;
            inf.n_intervals += (int32_t) u.n_ivl;
            all_single = all_single && u.ncs == 1;
        }
        const MjImageDev &y = b->img[mm.unit[0]];
        cb.ncs = 3;
        cb.hmax = mm.hsf[0];
        cb.vmax = mm.vsf[0];
        cb.X = mm.X;
        cb.Y = mm.Y;
        cb.pitch = y.pitch;
        cb.rgb_off = y.rgb_off;
        inf.mcus_x = (int32_t) ((mm.X + 8 * cb.hmax - 1) / (8 * cb.hmax));
        inf.mcus_y = (int32_t) ((mm.Y + 8 * cb.vmax - 1) / (8 * cb.vmax));
        inf.blocks_per_mcu = all_single ? 1 : 0;      /* one block at a time in every scan, or it differs from scan to scan */
        comb_max_x = mm.X > comb_max_x ? mm.X : comb_max_x;
        comb_max_y = mm.Y > comb_max_y ? mm.Y : comb_max_y;
        b->comb.push_back(cb);
    }

    /* Entropy workgroups, in the order the grids take them: the images whose lanes are longest
     * first (a workgroup lasts about as long as its lanes are; in a mixed batch the long ones
     * started last would end the kernels alone), files of one lane length in the order given.
     * The host only knows an upper bound on an image's groups of sequences (seq_cap); the layout
     * kernel, which knows the number, lists the groups that exist in this order of the images
     * (mj_k_layout), and the entropy grids are sized by the bound. */
    {
        b->img_order.resize((size_t) n);
        for (int i = 0; i < n; i++) b->img_order[(size_t) i] = (uint32_t) i;
        std::stable_sort(b->img_order.begin(), b->img_order.end(), [&](uint32_t x, uint32_t y) {
            return (uint64_t) b->img[x].sub_bits * b->img[x].lane_subs > (uint64_t) b->img[y].sub_bits * b->img[y].lane_subs;
        });
        uint64_t n_pairs = 0;
        for (int i = 0; i < n; i++) {
            const uint32_t wgs = (b->img[i].seq_cap + MJ_WG_SEQS - 1) / MJ_WG_SEQS;
            n_wg += wgs;
            n_pairs += (wgs + 1) / 2;
        }
        if (n_wg >= (1ull << 31)) return set_err("batch too large");
        b->n_pairs_cap = (uint32_t) n_pairs;
    }

    MjDeviceBatch &d = b->d;
    memset(&d, 0, sizeof d);
    d.n_img = (uint32_t) n;
    d.n_comb = (uint32_t) b->comb.size();
    d.comb_max_x = comb_max_x;
    d.comb_max_y = comb_max_y;
    d.n_tiles = (uint32_t) n_tiles;
    d.n_entropy_wg = (uint32_t) n_wg;
    d.n_multi_cap = (uint32_t) n_multi;
    d.n_idct_tiles = (uint32_t) n_idct;
    d.want_planes = ((flags & MJ_WANT_PLANES) || !b->multi.empty()) ? 1u : 0u;
    d.rgba = (flags & MJ_OUT_RGBA) ? 1u : 0u;
#if defined(MJ_EXPERIMENT)
    d.dbg = (flags >> 16) & 0xFFFFu;
#else
    d.dbg = 0;
#endif
    d.spec = spec_x;
    d.spec_bpm = spec_bpm;
    d.spec_g = spec_g;
    d.spec_lanes = spec_x ? (uint32_t) (n_rec / spec_g) : 0;
    d.spec_long = 0;
    for (int i = 0; i < n && spec_x; i++) d.spec_long |= (b->img[i].seq_cap > 64u) ? 1u : 0u;      /* SPEC_FOLD_SELF of the kernels */
    d.spec_stride = spec_x ? spec_bpm * mj_spec_len(spec_g, spec_x) : 0;
    d.max_ntab = 1;
    d.max_nac = 1;
    for (const MjHuffSet &hs : b->sets) {
        d.max_ntab = hs.ntab > d.max_ntab ? hs.ntab : d.max_ntab;
        d.max_nac = hs.nac > d.max_nac ? hs.nac : d.max_nac;
    }
    b->coef_bytes = coef_blocks * 128;
    b->rgb_bytes = rgb_total;
    b->plane_bytes = d.want_planes ? plane_total : 0;
    b->raw_bytes = raw_total + 64;

    Offsets &o = b->o;
    Carve m, st, w, out;
    o.img = m.take((uint64_t) n * sizeof(MjImageDev));
    o.sets = m.take(b->sets.size() * sizeof(MjHuffSet));
    o.tile_img = m.take(n_tiles * 4);
    o.img_order = m.take((uint64_t) n * 4);
    o.idct_img = m.take((uint64_t) n * sizeof(MjIdctImg));
    o.comb = m.take(b->comb.size() * sizeof(MjCombine));
    o.meta_bytes = m.off;
    o.dyn = st.take((uint64_t) n * sizeof(MjImageDyn));
    o.tile_cnt = st.take(n_tiles * sizeof(MjTileCnt));
    o.tile_base = st.take(n_tiles * sizeof(MjTileBase));
    o.ivl_start = st.take(n_ivl * 4);
    o.ivl_seq = st.take(n_ivl * 4);
    o.seq_ivl = st.take(n_seq * 4);
    o.seq_sub = st.take(n_seq * 4);
    o.seq_entry = st.take(n_seq * sizeof(MjSeqEntry));
    o.seq_seen = st.take(n_seq * sizeof(MjState));
    o.seq_flag = st.take(n_seq * 4);
    o.multi = st.take(n_multi * sizeof(uint2));
    o.multi_count = st.take(16);
    o.grp_list = st.take(n_wg * sizeof(uint2));
    o.pair_list = st.take((uint64_t) b->n_pairs_cap * sizeof(uint2));
    o.stats = st.take(MJ_ST_COUNT * sizeof(unsigned long long));
    o.state_bytes = st.off;
    o.raw = w.take(b->raw_bytes);
    o.cmp = w.take(cmp_total + 64);
    o.rec = w.take(n_rec * sizeof(MjSubRec));
    o.spec_rec = w.take(spec_x ? n_rec / spec_g * d.spec_stride * sizeof(MjSubRec) : 0);
    o.spec_state = w.take(spec_x ? n_rec / spec_g * d.spec_stride * sizeof(MjState) : 0);
    o.spec_map = w.take(spec_x ? n_rec * MJ_SPEC_MAXC : 0);
    o.spec_lane = w.take(spec_x ? n_rec / spec_g * MJ_SPEC_MAXC : 0);
    o.spec_tot = w.take(spec_x ? n_seq * MJ_SPEC_MAXC : 0);
    o.spec_brk = w.take(spec_x ? n_seq * 4 : 0);
    o.spec_carry = w.take(spec_x ? n_seq : 0);
    o.work_bytes = w.off;
    o.coefs = out.take(b->coef_bytes);
    o.rgb = out.take(rgb_total);
    o.planes = out.take(b->plane_bytes);
    o.out_bytes = out.off;
    return 0;
}

/* device pointers of every buffer, given where the regions live */
void bind(mj_batch *b, uint8_t *meta, uint8_t *state, uint8_t *work, uint8_t *coefs, uint8_t *rgb, uint8_t *planes) {
    const Offsets &o = b->o;
    MjDeviceBatch &d = b->d;
    d.img = reinterpret_cast<const MjImageDev *>(meta + o.img);
    d.sets = reinterpret_cast<const MjHuffSet *>(meta + o.sets);
    d.tile_img = reinterpret_cast<const uint32_t *>(meta + o.tile_img);
    d.img_order = reinterpret_cast<const uint32_t *>(meta + o.img_order);
    d.n_sync_pairs = b->n_pairs_cap;
    d.n_busy_groups = b->n_busy_groups;
    d.idct_img = reinterpret_cast<const MjIdctImg *>(meta + o.idct_img);
    d.comb = reinterpret_cast<const MjCombine *>(meta + o.comb);
    d.dyn = reinterpret_cast<MjImageDyn *>(state + o.dyn);
    d.tile_cnt = reinterpret_cast<MjTileCnt *>(state + o.tile_cnt);
    d.tile_base = reinterpret_cast<MjTileBase *>(state + o.tile_base);
    d.ivl_start = reinterpret_cast<uint32_t *>(state + o.ivl_start);
    d.ivl_seq = reinterpret_cast<uint32_t *>(state + o.ivl_seq);
    d.seq_ivl = reinterpret_cast<uint32_t *>(state + o.seq_ivl);
    d.seq_sub = reinterpret_cast<uint32_t *>(state + o.seq_sub);
    d.seq_entry = reinterpret_cast<MjSeqEntry *>(state + o.seq_entry);
    d.seq_seen = reinterpret_cast<MjState *>(state + o.seq_seen);
    d.seq_flag = reinterpret_cast<uint32_t *>(state + o.seq_flag);
    d.multi = reinterpret_cast<uint2 *>(state + o.multi);
    d.multi_count = reinterpret_cast<uint32_t *>(state + o.multi_count);
    d.grp_list = reinterpret_cast<uint2 *>(state + o.grp_list);
    d.pair_list = reinterpret_cast<uint2 *>(state + o.pair_list);
    d.stats = reinterpret_cast<unsigned long long *>(state + o.stats);
    d.raw = work + o.raw;
    d.cmp = work + o.cmp;
    d.rec = reinterpret_cast<MjSubRec *>(work + o.rec);
    d.spec_rec = reinterpret_cast<MjSubRec *>(work + o.spec_rec);
    d.spec_state = reinterpret_cast<MjState *>(work + o.spec_state);
    d.spec_map = work + o.spec_map;
    d.spec_lane = work + o.spec_lane;
    d.spec_tot = work + o.spec_tot;
    d.spec_brk = reinterpret_cast<uint32_t *>(work + o.spec_brk);
    d.spec_carry = work + o.spec_carry;
    d.coefs = reinterpret_cast<int16_t *>(coefs);
    d.rgb = rgb;
    d.planes = planes;
}

/* the batch's own allocation: take over what is there if it is large enough */
int arena_reserve(mj_batch *b, size_t need) {
    mj_ctx *ctx = b->ctx;
    if (b->arena && b->arena_cap >= need) return 0;
    if (b->arena) {
        (void) hipFree(b->arena);
        b->arena = nullptr;
        b->arena_cap = 0;
    }
    if (ctx->arena && ctx->arena_cap >= need) {          /* move into what the last batch left */
        b->arena = static_cast<uint8_t *>(ctx->arena);
        b->arena_cap = ctx->arena_cap;
        ctx->arena = nullptr;
        ctx->arena_cap = 0;
        return 0;
    }
    if (ctx->arena) {
        (void) hipFree(ctx->arena);
        ctx->arena = nullptr;
        ctx->arena_cap = 0;
    }
    /* a little more than asked: the next batch of the same shape rarely compresses to the
     * very same size, and growing means freeing and allocating gigabytes */
    const size_t want = need + need / 16;
    void *p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) return set_err("cannot allocate %zu bytes of device memory", want);
    b->arena = static_cast<uint8_t *>(p);
    b->arena_cap = want;
    return 0;
}

bool is_page_locked(const void *p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void) hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

bool all_page_locked(const std::vector<MjPrepared> &prep) {
    for (const MjPrepared &p : prep) {
        if (!is_page_locked(p.ent)) return false;
    }
    return true;
}

/* Enqueue on the streams ss[0 .. ns): clear the state region, send the meta region and
 * the entropy-coded bytes.  meta/state: device addresses of the regions.  With
 * stage == nullptr every image's bytes go straight from the caller's memory, one copy per
 * image, round-robin over the streams; otherwise they are gathered in `stage`
 * (page-locked, at least raw_bytes) first and leave as one copy.  Returns the number of
 * streams used through *used.  Allocates nothing. */
int upload(mj_batch *b, const std::vector<MjPrepared> &prep, uint8_t *meta, uint8_t *state, uint8_t *stage, const hipStream_t *ss,
           int ns, int *used) {
    const Offsets &o = b->o;
    hipStream_t s = ss[0];
    *used = 1;
    HIP_TRY(hipMemsetAsync(state, 0, o.state_bytes, s));
    uint8_t *mh = b->meta_host;
    const size_t n = b->img.size();
    memcpy(mh + o.img, b->img.data(), n * sizeof(MjImageDev));
    memcpy(mh + o.sets, b->sets.data(), b->sets.size() * sizeof(MjHuffSet));
    memcpy(mh + o.tile_img, b->tile_img.data(), b->tile_img.size() * 4);
    memcpy(mh + o.img_order, b->img_order.data(), b->img_order.size() * 4);
    memcpy(mh + o.idct_img, b->idct_img.data(), n * sizeof(MjIdctImg));
    if (!b->comb.empty()) memcpy(mh + o.comb, b->comb.data(), b->comb.size() * sizeof(MjCombine));
    HIP_TRY(hipMemcpyAsync(meta, mh, o.meta_bytes, hipMemcpyHostToDevice, s));
    uint8_t *raw = const_cast<uint8_t *>(b->d.raw);
    if (b->span_src) {
        HIP_TRY(hipMemcpyAsync(raw, b->span_src, b->span_bytes, hipMemcpyHostToDevice, s));
    } else if (stage) {
        for (size_t i = 0; i < n; i++) memcpy(stage + b->img[i].raw_off, prep[i].ent, b->img[i].raw_len);
        HIP_TRY(hipMemcpyAsync(raw, stage, b->raw_bytes, hipMemcpyHostToDevice, s));
    } else {
        for (size_t i = 0; i < n; i++) {
            const int k = (int) (i % (size_t) ns);
            if (k + 1 > *used) *used = k + 1;
            if (b->img[i].raw_len) HIP_TRY(hipMemcpyAsync(raw + b->img[i].raw_off, prep[i].ent, b->img[i].raw_len, hipMemcpyHostToDevice, ss[k]));
        }
    }
    return 0;
}

int take_events(mj_batch *b, EventSet *es) {
    if (!b->pool.empty()) {
        *es = b->pool.back();
        b->pool.pop_back();
    } else {
        for (hipEvent_t &ev : es->ev) HIP_TRY(hipEventCreate(&ev));
    }
    return 0;
}

/* the whole hot path of the batch on stream s (asynchronous) */
int enqueue_decode(mj_batch *b, hipStream_t s) {
    const bool timing = (b->flags & MJ_TIMING) != 0;
    EventSet es{};
    if (timing) {
        if (b->pending.size() >= 256 && collect_timing(b)) return -1;
        if (take_events(b, &es)) return -1;
    }
    auto mark = [&](int i) -> hipError_t { return timing ? hipEventRecord(es.ev[i], s) : hipSuccess; };
    b->stream = s;
    if (b->uploaded_async) {
        for (int i = 0; i < b->n_up; i++) HIP_TRY(hipStreamWaitEvent(s, b->ev_up[i], 0));
    }

    HIP_TRY(mark(0));
    /* the per-decode counters are cleared by the first scan kernel */
    if (b->flags & MJ_ENTROPY_SERIAL) {
        /* the diagnostic decoder stores only non-zero coefficients (decode.c:612 zeroes each block first) */
        HIP_TRY(hipMemsetAsync(b->d.coefs, 0, b->coef_bytes ? b->coef_bytes : 16, s));
    }
    HIP_TRY(mark(1));
    HIP_TRY(mj_launch_scan(b->d, s));
    HIP_TRY(mark(2));
    if (b->flags & MJ_ENTROPY_SERIAL) {
        HIP_TRY(mj_launch_entropy_serial(b->d, s));
        HIP_TRY(mark(3));
        HIP_TRY(mark(4));
    } else {
        HIP_TRY(b->d.spec ? mj_launch_spec(b->d, s) : mj_launch_entropy(b->d, s));
        HIP_TRY(mark(3));
        HIP_TRY(mj_launch_entropy_fix(b->d, s));
        HIP_TRY(mj_launch_entropy_write(b->d, s));
        HIP_TRY(mark(4));
    }
    HIP_TRY(mj_launch_idct_colour(b->d, s));
    HIP_TRY(mj_launch_combine(b->d, s));
    HIP_TRY(mark(5));
    HIP_TRY(hipEventRecord(b->ev_done, s));
    b->decoded = true;
    if (timing) b->pending.push_back(es);
    return 0;
}

void batch_free(mj_batch *b, bool park) {
    for (EventSet &es : b->pending) b->pool.push_back(es);
    for (EventSet &es : b->pool) {
        for (hipEvent_t ev : es.ev) (void) hipEventDestroy(ev);
    }
    mj_ctx *c = b->ctx;
    if (b->arena && !b->borrowed) {
        if (park && c->arena && c->arena_cap < b->arena_cap) {
            (void) hipFree(c->arena);
            c->arena = nullptr;
        }
        if (park && !c->arena) {
            c->arena = b->arena;
            c->arena_cap = b->arena_cap;
        } else {
            (void) hipFree(b->arena);
        }
    }
    if (b->meta_host) {
        if (park && c->meta_host && c->meta_cap < b->meta_cap) {
            (void) hipHostFree(c->meta_host);
            c->meta_host = nullptr;
        }
        if (park && !c->meta_host) {
            c->meta_host = b->meta_host;
            c->meta_cap = b->meta_cap;
        } else {
            (void) hipHostFree(b->meta_host);
        }
    }
    if (b->stage) (void) hipHostFree(b->stage);
    for (hipEvent_t e : b->ev_up) {
        if (e) (void) hipEventDestroy(e);
    }
    if (b->ev_done) (void) hipEventDestroy(b->ev_done);
    delete b;
}

/* lay the images out in the batch's own allocation and send them on the copy stream;
 * with `wait` the call returns when the copies are through (the caller's buffers are
 * then free) */
int fill_own(mj_batch *b, MjPrep &pp, uint32_t flags, bool own_stage, bool wait) {
    mj_ctx *ctx = b->ctx;
    const std::vector<MjPrepared> &prep = pp.img;
    if (layout(b, pp, flags, true, true)) return -1;
    const Offsets &o = b->o;
    const size_t need = o.meta_bytes + o.state_bytes + o.work_bytes + o.out_bytes;
    if (arena_reserve(b, need)) return -1;
    if (!b->meta_host && ctx->meta_host) {               /* the mirror the last batch left */
        b->meta_host = ctx->meta_host;
        b->meta_cap = ctx->meta_cap;
        ctx->meta_host = nullptr;
        ctx->meta_cap = 0;
    }
    if (host_reserve(&b->meta_host, &b->meta_cap, o.meta_bytes, "the batch descriptors")) return -1;
    uint8_t *meta = b->arena, *state = meta + o.meta_bytes, *work = state + o.state_bytes, *out = work + o.work_bytes;
    bind(b, meta, state, work, out + o.coefs, out + o.rgb, b->plane_bytes ? out + o.planes : nullptr);
    uint8_t *stage = nullptr;
    if (!b->span_src && !all_page_locked(prep)) {
        if (own_stage) {
            if (host_reserve(&b->stage, &b->stage_cap, b->raw_bytes, "staging the entropy-coded bytes")) return -1;
            stage = b->stage;
        } else {
            if (host_reserve(&ctx->stage, &ctx->stage_cap, b->raw_bytes, "staging the entropy-coded bytes")) return -1;
            stage = ctx->stage;
        }
    }
    if (upload(b, prep, meta, state, stage, ctx->copy, kCopyStreams, &b->n_up)) return -1;
    for (int i = 0; i < b->n_up; i++) HIP_TRY(hipEventRecord(b->ev_up[i], ctx->copy[i]));
    b->uploaded_async = true;
    b->decoded = false;
    if (wait && batch_wait(b)) return -1;
    return 0;
}

int check_flags(uint32_t flags, const char *who) {
#if !defined(MJ_EXPERIMENT)
    if (flags & ~MJ_FLAGS_ALL) return set_err("%s: unknown flag bits 0x%x", who, flags & ~MJ_FLAGS_ALL);
#else
    (void) flags;
    (void) who;
#endif
    return 0;
}

int batch_build(mj_ctx *ctx, MjPrep &prep, uint32_t flags, mj_batch **out, const char *who) {
    if (check_flags(flags, who)) return -1;
    HIP_TRY(hipSetDevice(ctx->device));
    mj_batch *b = batch_new(ctx);
    if (!b) return -1;
    if (fill_own(b, prep, flags, false, true) != 0) {
        char msg[sizeof g_err];
        snprintf(msg, sizeof msg, "%s", g_err);
        batch_free(b, true);
        return set_err("%s: %s", who, msg);
    }
    *out = b;
    return 0;
}

}  // namespace

extern "C" {

const char *mj_last_error(void) {
    return g_err;
}

int mj_create(int device, mj_ctx **out) {
    if (!out) return set_err("mj_create: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        return set_err("mj_create: no HIP device available (%s); this library has no CPU decode path",
                       e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n) return set_err("mj_create: device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    mj_ctx *c = new mj_ctx();
    c->device = device;
    for (hipStream_t &cs : c->copy) cs = nullptr;
    e = hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking);
    for (int i = 0; i < kCopyStreams && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&c->copy[i], hipStreamNonBlocking);
    if (e != hipSuccess) {
        if (c->own) (void) hipStreamDestroy(c->own);
        for (hipStream_t cs : c->copy) {
            if (cs) (void) hipStreamDestroy(cs);
        }
        delete c;
        return set_err("mj_create: hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    c->stream = c->own;
    /* this device's copy of the static work memory: with the device current, ask the
     * runtime where it is; from here on only that address is used, never the array's name */
    void *sp = nullptr;
    e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
    if (e == hipSuccess && sp) {
        c->scratch_dev = static_cast<uint8_t *>(sp);
        c->sbatch = batch_new(c);
        if (c->sbatch) {
            c->sbatch->borrowed = true;
            if (hipHostMalloc(reinterpret_cast<void **>(&c->sbatch->meta_host), kStaticMetaBytes, hipHostMallocDefault) == hipSuccess) {
                c->sbatch->meta_cap = kStaticMetaBytes;
            } else {
                c->sbatch->meta_host = nullptr;
                batch_free(c->sbatch, false);
                c->sbatch = nullptr;
            }
        }
    }
    if (!c->sbatch) {
        /* the context still serves every other entry point; mj_decode_static reports this */
        c->scratch_dev = nullptr;
        (void) hipGetLastError();
    }
    *out = c;
    return 0;
}

void mj_destroy(mj_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipStreamSynchronize(ctx->own);
    for (hipStream_t cs : ctx->copy) (void) hipStreamSynchronize(cs);
    if (ctx->sbatch) batch_free(ctx->sbatch, false);
    (void) hipStreamDestroy(ctx->own);
    for (hipStream_t cs : ctx->copy) (void) hipStreamDestroy(cs);
    if (ctx->stage) (void) hipHostFree(ctx->stage);
    if (ctx->meta_host) (void) hipHostFree(ctx->meta_host);
    if (ctx->arena) (void) hipFree(ctx->arena);
    if (ctx->conv) (void) hipFree(ctx->conv);
    delete ctx;
}

int mj_set_sync_policy(int mode, int sub_bits, int lane_subs, int extra) {
    if (mode < 0 || mode > 2) return set_err("mj_set_sync_policy: mode %d (0, 1 or 2)", mode);
    if (sub_bits < 0 || sub_bits > 32768) return set_err("mj_set_sync_policy: sub_bits %d", sub_bits);
    if (lane_subs != 0 && lane_subs != 1 && lane_subs != 2 && lane_subs != 4 && lane_subs != 8) {
        return set_err("mj_set_sync_policy: lane_subs %d (0, 1, 2, 4 or 8)", lane_subs);
    }
    if (extra < 0 || extra > 8) return set_err("mj_set_sync_policy: extra %d (0 .. 8)", extra);
    g_sync.mode = (uint32_t) mode;
    g_sync.sub_bits = (uint32_t) sub_bits;
    g_sync.lane_subs = (uint32_t) lane_subs;
    g_sync.extra = (uint32_t) extra;
    return 0;
}

int mj_device_pci_bus_id(int device, char *buf, size_t cap) {
    if (!buf || cap < 13) return set_err("mj_device_pci_bus_id: bad arguments (the text needs 13 bytes)");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        (void) hipGetLastError();
        return set_err("mj_device_pci_bus_id: no HIP device %d", device);
    }
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int) cap, device));
    for (char *c = buf; *c; c++) {
        if (*c >= 'A' && *c <= 'F') *c = (char) (*c - 'A' + 'a');      /* as sysfs spells it */
    }
    return 0;
}

int mj_set_stream(mj_ctx *ctx, void *hip_stream) {
    if (!ctx) return set_err("mj_set_stream: null context");
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own;
    return 0;
}

void *mj_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        set_err("mj_host_alloc: cannot allocate %zu bytes of page-locked memory", bytes);
        return nullptr;
    }
    return p;
}

void mj_host_free(void *p) {
    if (p) (void) hipHostFree(p);
}

void mj_batch_destroy(mj_batch *b) {
    /* the view mj_decode_static hands out belongs to the context, a batch that mj_pipe_submit
     * returned to its pipeline, which uses the slot again */
    if (!b || b->borrowed || b->owned_by_pipe) return;
    (void) hipSetDevice(b->ctx->device);
    /* the stream every decode of this batch went to, whatever the context's current one is */
    (void) batch_wait(b);
    if (b->stream) (void) hipStreamSynchronize(b->stream);
    batch_free(b, true);
}

int mj_batch_create(mj_ctx *ctx, const uint8_t *const *bufs, const size_t *lens, int n, uint32_t flags,
                    mj_batch **out) {
    if (!ctx || !bufs || !lens || !out || n <= 0) return set_err("mj_batch_create: bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));      /* the cut of the lanes follows this device's size */
    MjPrep prep;
    if (prepare(bufs, lens, n, prep, "mj_batch_create")) return -1;
    return batch_build(ctx, prep, flags, out, "mj_batch_create");
}

int mj_batch_create_scan(mj_ctx *ctx, const uint8_t *entropy, size_t entropy_len, const FrameHeader *fh,
                         const ScanHeader *sh, const HuffTables *hts, const QuantTables *qts, RestartInterval ri,
                         uint32_t flags, mj_batch **out) {
    if (!ctx || !entropy || !fh || !sh || !hts || !qts || !out) return set_err("mj_batch_create_scan: bad arguments");
    if (fh->ncs != 1 && fh->ncs != 3) return set_err("mj_batch_create_scan: only grey and three-component images are supported");
    HIP_TRY(hipSetDevice(ctx->device));
    MjParsed P;
    memset(&P, 0, sizeof P);
    P.fh = *fh;   /* shallow views: nothing here is freed through P */
    P.sh = *sh;
    P.hts = *hts;
    P.qts = *qts;
    P.ri = ri;
    P.ent_pos = 0;
    MjPrep prep;
    prep.img.resize(1);
    prep.n_files = 1;
    const char *why = "?";
    if (prepare_one(prep, prep.img[0], P, entropy_len, &why) != 0) {
        return set_err("mj_batch_create_scan: %s", why);
    }
    prep.img[0].ent = entropy;
    if (choose_for_batch(prep, "mj_batch_create_scan")) return -1;
    return batch_build(ctx, prep, flags, out, "mj_batch_create_scan");
}

int mj_batch_decode(mj_batch *b) {
    if (!b) return set_err("mj_batch_decode: null batch");
    HIP_TRY(hipSetDevice(b->ctx->device));
    return enqueue_decode(b, b->ctx->stream);
}

int mj_batch_sync(mj_batch *b) {
    if (!b) return set_err("mj_batch_sync: null batch");
    HIP_TRY(hipSetDevice(b->ctx->device));
    return batch_wait(b);
}

/* ---- pipeline: batch k+1 is parsed and uploaded while batch k decodes ------------ */

int mj_pipe_create(mj_ctx *ctx, int depth, uint32_t flags, mj_pipe **out) {
    if (!ctx || !out || depth < 2 || depth > 8) return set_err("mj_pipe_create: bad arguments (depth 2..8)");
    if (check_flags(flags, "mj_pipe_create")) return -1;
    HIP_TRY(hipSetDevice(ctx->device));
    mj_pipe *p = new mj_pipe();
    p->ctx = ctx;
    p->flags = flags;
    for (int i = 0; i < depth; i++) {
        mj_batch *b = batch_new(ctx);
        if (!b) {
            for (mj_batch *s : p->slots) batch_free(s, false);
            delete p;
            return -1;
        }
        b->owned_by_pipe = true;
        p->slots.push_back(b);
    }
    if (hipStreamCreateWithFlags(&p->alt, hipStreamNonBlocking) != hipSuccess) {
        for (mj_batch *s : p->slots) batch_free(s, false);
        delete p;
        return set_err("mj_pipe_create: cannot create the second decode stream");
    }
    *out = p;
    return 0;
}

void mj_pipe_destroy(mj_pipe *p) {
    if (!p) return;
    (void) hipSetDevice(p->ctx->device);
    for (mj_batch *b : p->slots) {
        (void) batch_wait(b);
        if (b->stream) (void) hipStreamSynchronize(b->stream);
        batch_free(b, false);
    }
    if (p->alt) (void) hipStreamDestroy(p->alt);
    delete p;
}

int mj_pipe_submit(mj_pipe *p, const uint8_t *const *bufs, const size_t *lens, int n, mj_batch **out) {
    if (!p || !bufs || !lens || !out || n <= 0) return set_err("mj_pipe_submit: bad arguments");
    mj_ctx *ctx = p->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    mj_batch *b = p->slots[p->next % p->slots.size()];
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    /* the slot's previous occupant: its decode must be over before its memory (device and
     * page-locked mirrors) is written again */
    if (batch_wait(b)) return -1;
    const clk::time_point t1 = clk::now();
    MjPrep prep;
    if (prepare(bufs, lens, n, prep, "mj_pipe_submit")) return -1;
    if (fill_own(b, prep, p->flags, true, false) != 0) return -1;
    /* Consecutive batches decode on two streams in turn: the last synchronisation rounds and
     * the ends of the other kernels of one batch, which leave most of the chip idle, run
     * beside the kernels of the next.  Batches are independent (each slot has its own device
     * memory); a caller who has put the context on a stream of its own keeps strict order. */
    hipStream_t s = (ctx->stream == ctx->own && (p->next & 1u)) ? p->alt : ctx->stream;
    if (enqueue_decode(b, s) != 0) return -1;
    const clk::time_point t2 = clk::now();
    p->wait_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
    p->host_ms += std::chrono::duration<double, std::milli>(t2 - t1).count();
    p->next++;
    *out = b;
    return 0;
}

int mj_pipe_host_ms(mj_pipe *p, double *wait_ms, double *work_ms, int64_t *submits, int reset) {
    if (!p) return set_err("mj_pipe_host_ms: null pipeline");
    if (wait_ms) *wait_ms = p->wait_ms;
    if (work_ms) *work_ms = p->host_ms;
    if (submits) *submits = (int64_t) p->next;
    if (reset) {
        p->wait_ms = p->host_ms = 0;
    }
    return 0;
}

/* ---- the entry point that allocates nothing --------------------------------------- */

int mj_decode_static(mj_ctx *ctx, const uint8_t *const *bufs, const size_t *lens, int n, uint32_t flags, void *dev_coefs,
                     size_t coefs_bytes, void *dev_rgb, size_t rgb_bytes, mj_batch **view) {
    if (!ctx || !bufs || !lens || n <= 0 || !dev_coefs || !dev_rgb) return set_err("mj_decode_static: bad arguments");
    if (flags & ~(uint32_t) MJ_OUT_RGBA) return set_err("mj_decode_static: only MJ_OUT_RGBA is accepted here (flags 0x%x)", flags);
    if (!ctx->sbatch || !ctx->scratch_dev) return set_err("mj_decode_static: the static work memory could not be located on this device");
    if (((uintptr_t) dev_coefs & 15u) || ((uintptr_t) dev_rgb & 15u)) return set_err("mj_decode_static: output buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    mj_batch *b = ctx->sbatch;
    /* every call on this GPU uses the same work memory: the previous call must be over */
    if (batch_wait(b)) return -1;
    MjPrep prep;
    if (prepare(bufs, lens, n, prep, "mj_decode_static")) return -1;
    if (!prep.multi.empty()) {
        return set_err("mj_decode_static: image %u is coded in several scans, which need plane memory this entry point has none of",
                       prep.multi.front().file);
    }
    if (layout(b, prep, flags, true, false)) return -1;      /* the work memory is fixed: no room for speculation */
    const Offsets &o = b->o;
    const size_t need = o.meta_bytes + o.state_bytes + o.work_bytes;
    if (need > (size_t) MJ_STATIC_SCRATCH_BYTES || o.meta_bytes > b->meta_cap) {
        b->decoded = false;
        return set_err("mj_decode_static: this batch needs %zu bytes of work memory (%zu of them descriptors); the static limits are %zu and %zu",
                       need, o.meta_bytes, (size_t) MJ_STATIC_SCRATCH_BYTES, b->meta_cap);
    }
    if (b->coef_bytes > coefs_bytes || b->rgb_bytes > rgb_bytes) {
        b->decoded = false;
        return set_err("mj_decode_static: outputs need %llu bytes of coefficients and %llu of RGB; %zu and %zu were passed",
                       (unsigned long long) b->coef_bytes, (unsigned long long) b->rgb_bytes, coefs_bytes, rgb_bytes);
    }
    uint8_t *meta = ctx->scratch_dev, *state = meta + o.meta_bytes, *work = state + o.state_bytes;
    bind(b, meta, state, work, static_cast<uint8_t *>(dev_coefs), static_cast<uint8_t *>(dev_rgb), nullptr);
    hipStream_t s = ctx->stream;
    /* uploads and kernels in order on one stream; the state region is cleared there first */
    b->uploaded_async = false;
    b->decoded = false;
    if (upload(b, prep.img, meta, state, nullptr, &s, 1, &b->n_up)) return -1;
    HIP_TRY(hipEventRecord(b->ev_up[0], s));
    if (enqueue_decode(b, s)) return -1;
    if (view) *view = b;
    /* the caller's buffers and the descriptor mirror are free once the upload is through;
     * the kernels run on */
    HIP_TRY(hipEventSynchronize(b->ev_up[0]));
    return 0;
}

/* ---- planes -> RGB on its own (colour_conversion.c:5-83) ---------------------------- */

int mj_planes_to_rgb(mj_ctx *ctx, int ncs, const uint8_t *const *planes, const int32_t *strides, const int32_t *hsf,
                     const int32_t *vsf, int width, int height, uint8_t *pixels, size_t pitch) {
    if (!ctx || !planes || !strides || !hsf || !vsf || !pixels) return set_err("mj_planes_to_rgb: bad arguments");
    if (ncs != 1 && ncs != 3) return set_err("mj_planes_to_rgb: %d components (1 or 3 are converted)", ncs);
    if (width <= 0 || height <= 0 || width > 65535 || height > 65535) return set_err("mj_planes_to_rgb: bad size %d x %d", width, height);
    if (pitch < (size_t) 3 * (size_t) width) return set_err("mj_planes_to_rgb: pitch smaller than a row");
    HIP_TRY(hipSetDevice(ctx->device));
    MjCombine cb{};
    cb.ncs = (uint32_t) ncs;
    cb.X = (uint32_t) width;
    cb.Y = (uint32_t) height;
    for (int c = 0; c < ncs; c++) {
        if (!planes[c] || strides[c] <= 0 || hsf[c] <= 0 || vsf[c] <= 0 || hsf[c] > 4 || vsf[c] > 4) {
            return set_err("mj_planes_to_rgb: component %d: bad plane, stride or sampling factors", c);
        }
        cb.hmax = (uint32_t) hsf[c] > cb.hmax ? (uint32_t) hsf[c] : cb.hmax;
        cb.vmax = (uint32_t) vsf[c] > cb.vmax ? (uint32_t) vsf[c] : cb.vmax;
    }
    /* what the loops of colour_conversion.c:61-69 read of every plane: rows 0 .. (uint32)((height - 1) * vratio),
     * of the last of them the bytes up to (uint32)(hratio * (width - 1)) */
    size_t bytes[3] = {0, 0, 0}, off = 256;
    for (int c = 0; c < ncs; c++) {
        const float hr = (float) hsf[c] / (float) cb.hmax, vr = (float) vsf[c] / (float) cb.vmax;
        const uint32_t last_row = (uint32_t) ((float) (height - 1) * vr), last_col = (uint32_t) (hr * (float) (width - 1));
        if (last_col >= (uint32_t) strides[c]) return set_err("mj_planes_to_rgb: component %d: stride %d is smaller than the row the frame needs", c, strides[c]);
        bytes[c] = (size_t) last_row * (size_t) strides[c] + last_col + 1;
        cb.hsf[c] = (uint32_t) hsf[c];
        cb.vsf[c] = (uint32_t) vsf[c];
        cb.pw[c] = (uint32_t) strides[c];
        cb.plane_off[c] = off;
        off = (size_t) align_up(off + bytes[c] + 4, 256);      /* + 4: the kernel reads a full-resolution row four bytes at a time */
    }
    cb.pitch = (uint32_t) ((3u * (uint32_t) width + 3u) & ~3u);
    cb.rgb_off = off;
    const size_t rgb_bytes = (size_t) cb.pitch * (size_t) height, total = off + rgb_bytes;
    if (ctx->conv_cap < total) {
        if (ctx->conv) (void) hipFree(ctx->conv);
        ctx->conv = nullptr;
        ctx->conv_cap = 0;
        void *p = nullptr;
        if (hipMalloc(&p, total + total / 8) != hipSuccess) return set_err("mj_planes_to_rgb: cannot allocate %zu bytes of device memory", total);
        ctx->conv = static_cast<uint8_t *>(p);
        ctx->conv_cap = total + total / 8;
    }
    if (host_reserve(&ctx->stage, &ctx->stage_cap, total, "the planes and their RGB")) return -1;
    memcpy(ctx->stage, &cb, sizeof cb);
    for (int c = 0; c < ncs; c++) memcpy(ctx->stage + cb.plane_off[c], planes[c], bytes[c]);
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->conv, ctx->stage, off, hipMemcpyHostToDevice, s));
    HIP_TRY(mj_launch_planes_rgb(reinterpret_cast<const MjCombine *>(ctx->conv), 1, cb.X, cb.Y, ctx->conv, ctx->conv, false, s));
    HIP_TRY(hipMemcpyAsync(ctx->stage + off, ctx->conv + off, rgb_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int y = 0; y < height; y++) memcpy(pixels + (size_t) y * pitch, ctx->stage + off + (size_t) y * cb.pitch, (size_t) 3 * (size_t) width);
    return 0;
}

/* ---- results ----------------------------------------------------------------------- */

/* the record of file i if it is coded in one scan per component, else null */
static const MjMultiScan *multi_of(const mj_batch *b, int i) {
    for (const MjMultiScan &mm : b->multi) {
        if (mm.file == (uint32_t) i) return &mm;
    }
    return nullptr;
}

int mj_batch_size(const mj_batch *b) {
    return b ? (int) b->n_files : -1;
}

int mj_batch_image_info(const mj_batch *b, int i, mj_image_info *info) {
    if (!b || !info || i < 0 || i >= (int) b->n_files) return set_err("mj_batch_image_info: bad arguments");
    *info = b->info[i];
    return 0;
}

int mj_batch_status(mj_batch *b, int i, uint32_t *status) {
    if (!b || !status || i < 0 || i >= (int) b->n_files) return set_err("mj_batch_status: bad arguments");
    if (mj_batch_sync(b)) return -1;
    MjImageDyn dy;
    HIP_TRY(hipMemcpy(&dy, b->d.dyn + i, sizeof dy, hipMemcpyDeviceToHost));
    *status = dy.status;
    if (const MjMultiScan *mm = multi_of(b, i)) {        /* any of its scans */
        for (uint32_t sc = 0; sc < mm->n_scans; sc++) {
            HIP_TRY(hipMemcpy(&dy, b->d.dyn + mm->scan_unit[sc], sizeof dy, hipMemcpyDeviceToHost));
            *status |= dy.status;
        }
    }
    return 0;
}

int mj_batch_scan_end(mj_batch *b, int i, size_t *offset) {
    if (!b || !offset || i < 0 || i >= (int) b->n_files) return set_err("mj_batch_scan_end: bad arguments");
    if (multi_of(b, i)) return set_err("mj_batch_scan_end: image %d has several scans", i);
    if (mj_batch_sync(b)) return -1;
    MjImageDyn dy;
    HIP_TRY(hipMemcpy(&dy, b->d.dyn + i, sizeof dy, hipMemcpyDeviceToHost));
    *offset = (dy.term_pos == MJ_NO_TERMINATOR) ? (size_t) b->img[i].raw_len : (size_t) dy.term_pos;
    return 0;
}

void *mj_batch_device_rgb(const mj_batch *b) {
    return b ? b->d.rgb : nullptr;
}

void *mj_batch_device_coefs(const mj_batch *b) {
    return b ? b->d.coefs : nullptr;
}

int mj_batch_copy_rgb(mj_batch *b, int i, uint8_t *dst, size_t dst_pitch) {
    if (!b || !dst || i < 0 || i >= (int) b->n_files) return set_err("mj_batch_copy_rgb: bad arguments");
    const MjImageDev &im = b->img[i];
    const size_t row = (size_t) (b->d.rgba ? 4 : 3) * im.X;
    if (dst_pitch < row) return set_err("mj_batch_copy_rgb: pitch smaller than a row");
    if (mj_batch_sync(b)) return -1;
    /* one contiguous copy into the context's page-locked buffer, then rows to the caller's
     * pitch on the host (a pitched copy into pageable memory is several times slower) */
    const size_t bytes = (size_t) im.pitch * im.Y;
    if (bytes == 0) return 0;
    mj_ctx *ctx = b->ctx;
    if (host_reserve(&ctx->stage, &ctx->stage_cap, bytes, "reading results back")) return -1;
    HIP_TRY(hipMemcpy(ctx->stage, b->d.rgb + im.rgb_off, bytes, hipMemcpyDeviceToHost));
    for (uint32_t y = 0; y < im.Y; y++) memcpy(dst + (size_t) y * dst_pitch, ctx->stage + (size_t) y * im.pitch, row);
    return 0;
}

int mj_batch_copy_coefs(mj_batch *b, int i, int16_t *dst) {
    if (!b || !dst || i < 0 || i >= (int) b->n_files) return set_err("mj_batch_copy_coefs: bad arguments");
    if (mj_batch_sync(b)) return -1;
    size_t n_blocks = 0;
    if (const MjMultiScan *mm = multi_of(b, i)) {
        /* scan after scan, in the order of the file */
        for (uint32_t sc = 0; sc < mm->n_scans; sc++) {
            const MjImageDev &u = b->img[mm->scan_unit[sc]];
            const size_t nb = (size_t) u.n_mcu * u.bpm;
            HIP_TRY(hipMemcpy(dst + n_blocks * 64, b->d.coefs + u.coef_off * 64, nb * 128, hipMemcpyDeviceToHost));
            n_blocks += nb;
        }
    } else {
        const MjImageDev &im = b->img[i];
        n_blocks = (size_t) im.n_mcu * im.bpm;
        HIP_TRY(hipMemcpy(dst, b->d.coefs + im.coef_off * 64, n_blocks * 128, hipMemcpyDeviceToHost));
    }
    /* HBM holds the blocks in natural order; the caller gets them in zigzag order, as the
     * reference hands them to dequant_data_unit (quant_table.c:131) */
    static const uint8_t natpos[64] = MJ_NATPOS_INIT;
    for (size_t k = 0; k < n_blocks; k++) {
        int16_t *blk = dst + k * 64, t[64];
        memcpy(t, blk, sizeof t);
        for (int z = 0; z < 64; z++) blk[z] = t[natpos[z]];
    }
    return 0;
}

int mj_batch_copy_plane(mj_batch *b, int i, int comp, uint8_t *dst) {
    if (!b || !dst || i < 0 || i >= (int) b->n_files) return set_err("mj_batch_copy_plane: bad arguments");
    if (!b->d.want_planes) return set_err("mj_batch_copy_plane: batch was created without MJ_WANT_PLANES");
    if (const MjMultiScan *mm = multi_of(b, i)) {        /* the plane of the component's own scan */
        if (comp < 0 || comp >= 3) return set_err("mj_batch_copy_plane: no such component");
        i = (int) mm->unit[comp];
        comp = (int) mm->unit_comp[comp];
    }
    const MjImageDev &im = b->img[i];
    if (comp < 0 || comp >= (int) im.ncs) return set_err("mj_batch_copy_plane: no such component");
    if (mj_batch_sync(b)) return -1;
    HIP_TRY(hipMemcpy(dst, b->d.planes + im.comp[comp].plane_off, (size_t) im.comp[comp].pw * im.comp[comp].ph,
                      hipMemcpyDeviceToHost));
    return 0;
}

int mj_batch_kernel_ms(mj_batch *b, float ms[MJ_T_COUNT], int launches[MJ_T_COUNT]) {
    if (!b || !ms || !launches) return set_err("mj_batch_kernel_ms: bad arguments");
    if (collect_timing(b)) return -1;
    for (int t = 0; t < MJ_T_COUNT; t++) {
        ms[t] = b->ms[t];
        launches[t] = b->launches[t];
    }
    return 0;
}

int mj_batch_reset_timing(mj_batch *b) {
    if (!b) return set_err("mj_batch_reset_timing: null batch");
    if (collect_timing(b)) return -1;
    for (int t = 0; t < MJ_T_COUNT; t++) {
        b->ms[t] = 0.f;
        b->launches[t] = 0;
    }
    return 0;
}

int mj_batch_entropy_stats(mj_batch *b, int64_t stats[8]) {
    if (!b || !stats) return set_err("mj_batch_entropy_stats: bad arguments");
    if (mj_batch_sync(b)) return -1;
    unsigned long long h[MJ_ST_COUNT];
    HIP_TRY(hipMemcpy(h, b->d.stats, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) stats[i] = (i < MJ_ST_COUNT) ? (int64_t) h[i] : 0;
    return 0;
}

}  // extern "C"
