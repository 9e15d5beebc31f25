/*
 * mj_abi.hip -- the C ABI declared in include/mi355jpeg.h: device context, batch
 * layout in HBM, stage launches, result copies and HIP-event timing.  Host code in
 * the reference's language (C) calls only these entry points; nothing HIP-specific
 * appears in their signatures.
 */
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../host/mj_host.h"
#include "mi355jpeg.h"
#include "mj_internal.h"

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

}  // namespace

struct mj_ctx {
    int device;
    hipStream_t own;
    hipStream_t stream;
    /* page-locked staging for the entropy-coded bytes of the batch being built: the copy to
     * HBM then runs at the bus rate; kept and grown from one batch to the next */
    uint8_t *stage = nullptr;
    size_t stage_cap = 0;
    /* the device memory of the last batch destroyed, for the next one to move into: a
     * batch is one allocation, and a caller that decodes batch after batch (or image after
     * image through the drop-in entry points) allocates once */
    void *arena = nullptr;
    size_t arena_cap = 0;
};

struct mj_batch {
    mj_ctx *ctx = nullptr;
    uint32_t flags = 0;
    std::vector<MjImageDev> img;
    std::vector<mj_image_info> info;
    std::vector<MjHuffSet> sets;
    uint8_t *arena = nullptr;      /* every device buffer of the batch is a piece of this one allocation */
    size_t arena_cap = 0, arena_used = 0;
    bool measuring = false;        /* first pass over the buffer list: sizes only */
    MjDeviceBatch d{};
    uint64_t coef_bytes = 0, rgb_bytes = 0, plane_bytes = 0;
    std::vector<EventSet> pending, pool;
    float ms[MJ_T_COUNT] = {0};
    int launches[MJ_T_COUNT] = {0};
};

namespace {

/* the context's page-locked staging buffer, at least `bytes` long (contents are not kept) */
int stage_reserve(mj_ctx *ctx, size_t bytes) {
    if (ctx->stage_cap >= bytes) return 0;
    if (ctx->stage) (void) hipHostFree(ctx->stage);
    ctx->stage = nullptr;
    ctx->stage_cap = 0;
    const size_t cap = bytes + bytes / 4;
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->stage), cap, hipHostMallocDefault) != hipSuccess) {
        return set_err("cannot allocate %zu bytes of page-locked staging memory", cap);
    }
    ctx->stage_cap = cap;
    return 0;
}

template <typename T>
int dev_alloc(mj_batch *b, T **out, uint64_t count) {
    uint64_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    bytes = align_up(bytes, 256);
    *out = nullptr;
    if (!b->measuring) {
        if (b->arena_used + bytes > b->arena_cap) return set_err("mj_batch_create: internal error: arena too small");
        *out = reinterpret_cast<T *>(b->arena + b->arena_used);
    }
    b->arena_used += bytes;
    return 0;
}

template <typename T>
int dev_upload(mj_batch *b, const T **out, const std::vector<T> &v) {
    T *p = nullptr;
    if (dev_alloc(b, &p, v.size())) return -1;
    if (p && !v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = p;
    return 0;
}

int collect_timing(mj_batch *b) {
    if (b->pending.empty()) return 0;
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
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
    e = hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return set_err("mj_create: hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    c->stream = c->own;
    *out = c;
    return 0;
}

void mj_destroy(mj_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipStreamDestroy(ctx->own);
    if (ctx->stage) (void) hipHostFree(ctx->stage);
    if (ctx->arena) (void) hipFree(ctx->arena);
    delete ctx;
}

int mj_set_stream(mj_ctx *ctx, void *hip_stream) {
    if (!ctx) return set_err("mj_set_stream: null context");
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own;
    return 0;
}

void mj_batch_destroy(mj_batch *b) {
    if (!b) return;
    (void) hipSetDevice(b->ctx->device);
    (void) hipStreamSynchronize(b->ctx->stream);
    for (EventSet &es : b->pending) b->pool.push_back(es);
    for (EventSet &es : b->pool) {
        for (hipEvent_t ev : es.ev) (void) hipEventDestroy(ev);
    }
    if (b->arena) {
        mj_ctx *c = b->ctx;
        if (c->arena && c->arena_cap < b->arena_cap) {
            (void) hipFree(c->arena);
            c->arena = nullptr;
        }
        if (!c->arena) {
            c->arena = b->arena;
            c->arena_cap = b->arena_cap;
        } else {
            (void) hipFree(b->arena);
        }
    }
    delete b;
}

/* One image ready for placement: flattened descriptor + tables + its entropy bytes. */
struct MjPrepared {
    MjImageDev im;
    MjHuffSet set;
    const uint8_t *ent;
};

static int batch_build(mj_ctx *ctx, std::vector<MjPrepared> &prep, uint32_t flags, mj_batch **out);

int mj_batch_create(mj_ctx *ctx, const uint8_t *const *bufs, const size_t *lens, int n, uint32_t flags,
                    mj_batch **out) {
    if (!ctx || !bufs || !lens || !out || n <= 0) return set_err("mj_batch_create: bad arguments");
    std::vector<MjPrepared> prep(n);
    for (int i = 0; i < n; i++) {
        MjParsed P;
        const char *why = "?";
        if (mj_parse_headers(bufs[i], lens[i], &P, &why) != 0) {
            mj_parsed_free(&P);
            return set_err("mj_batch_create: image %d: %s", i, why);
        }
        if (mj_flatten_image(&P, lens[i], &prep[i].im, &prep[i].set, &why) != 0) {
            mj_parsed_free(&P);
            return set_err("mj_batch_create: image %d: %s", i, why);
        }
        prep[i].ent = bufs[i] + P.ent_pos;
        mj_parsed_free(&P);
    }
    return batch_build(ctx, prep, flags, out);
}

int mj_batch_create_scan(mj_ctx *ctx, const uint8_t *entropy, size_t entropy_len, const FrameHeader *fh,
                         const ScanHeader *sh, const HuffTables *hts, const QuantTables *qts, RestartInterval ri,
                         uint32_t flags, mj_batch **out) {
    if (!ctx || !entropy || !fh || !sh || !hts || !qts || !out) return set_err("mj_batch_create_scan: bad arguments");
    MjParsed P;
    memset(&P, 0, sizeof P);
    P.fh = *fh;   /* shallow views: nothing here is freed through P */
    P.sh = *sh;
    P.hts = *hts;
    P.qts = *qts;
    P.ri = ri;
    P.ent_pos = 0;
    std::vector<MjPrepared> prep(1);
    const char *why = "?";
    if (mj_flatten_image(&P, entropy_len, &prep[0].im, &prep[0].set, &why) != 0) {
        return set_err("mj_batch_create_scan: %s", why);
    }
    prep[0].ent = entropy;
    return batch_build(ctx, prep, flags, out);
}

static int batch_build(mj_ctx *ctx, std::vector<MjPrepared> &prep, uint32_t flags, mj_batch **out) {
    const int n = (int) prep.size();
    HIP_TRY(hipSetDevice(ctx->device));
    mj_batch *b = new mj_batch();
    b->ctx = ctx;
    b->flags = flags;
    b->img.resize(n);
    b->info.resize(n);

    uint64_t raw_total = 0, cmp_total = 0, coef_blocks = 0, rgb_total = 0, plane_total = 0;
    uint64_t n_tiles = 0, n_ivl = 0, n_seq = 0, n_rec = 0, n_wg = 0, n_multi = 0, n_idct = 0;
    std::vector<uint32_t> tile_img, wg_img, idct_img, idct_first(n);

    for (int i = 0; i < n; i++) {
        MjImageDev &im = b->img[i];
        im = prep[i].im;
        const MjHuffSet &set = prep[i].set;
        if (im.raw_len >= (1ull << 28)) {
            mj_batch_destroy(b);
            return set_err("mj_batch_create: image %d: entropy segment of %llu bytes is too large", i,
                           (unsigned long long) im.raw_len);
        }
        /* images with identical tables share one set */
        uint32_t si = 0;
        for (; si < b->sets.size(); si++) {
            if (memcmp(&b->sets[si], &set, sizeof set) == 0) break;
        }
        if (si == b->sets.size()) b->sets.push_back(set);
        im.huff_set = si;

        im.raw_off = raw_total;
        raw_total = align_up(raw_total + im.raw_len + 32, 16);
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
        im.wg_first = (uint32_t) n_wg;
        const uint32_t wgs = (im.seq_cap + MJ_WG_SEQS - 1) / MJ_WG_SEQS;
        n_wg += wgs;
        uint64_t mc = im.raw_len * 8 / seq_bits;
        im.multi_cap = (uint32_t) (mc < im.n_ivl ? mc : im.n_ivl);
        n_multi += im.multi_cap;
        const uint32_t iw = im.mcus_y * mj_idct_tiles_x(im.mcus_x, im.bpm, im.hmax);
        idct_first[i] = (uint32_t) n_idct;
        n_idct += iw;
        tile_img.insert(tile_img.end(), im.n_tiles, (uint32_t) i);
        wg_img.insert(wg_img.end(), wgs, (uint32_t) i);
        idct_img.insert(idct_img.end(), iw, (uint32_t) i);
        if (n_tiles >= (1ull << 31) || n_seq >= (1ull << 31) || n_rec >= (1ull << 31) || n_idct >= (1ull << 31) || n_ivl >= (1ull << 31)) {
            mj_batch_destroy(b);
            return set_err("mj_batch_create: batch too large");
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

    MjDeviceBatch &d = b->d;
    d.n_img = (uint32_t) n;
    d.n_tiles = (uint32_t) n_tiles;
    d.n_entropy_wg = (uint32_t) n_wg;
    d.n_multi_cap = (uint32_t) n_multi;
    d.n_idct_wg = (uint32_t) n_idct;
    d.want_planes = (flags & MJ_WANT_PLANES) ? 1u : 0u;
    d.rgba = (flags & MJ_OUT_RGBA) ? 1u : 0u;
    d.dbg = (flags >> 16) & 0xFFFFu;
    d.max_ntab = 1;
    d.max_nac = 1;
    for (const MjHuffSet &hs : b->sets) {
        d.max_ntab = hs.ntab > d.max_ntab ? hs.ntab : d.max_ntab;
        d.max_nac = hs.nac > d.max_nac ? hs.nac : d.max_nac;
    }
    b->coef_bytes = coef_blocks * 128;
    b->rgb_bytes = rgb_total;
    b->plane_bytes = plane_total;

    /* raw entropy bytes: staged on the host, one copy to HBM */
    const size_t raw_bytes = (size_t) raw_total + 64;
    if (stage_reserve(ctx, raw_bytes)) {
        mj_batch_destroy(b);
        return -1;
    }
    {
        uint8_t *raw = ctx->stage;
        uint64_t at = 0;
        for (int i = 0; i < n; i++) {       /* the images' bytes, zeros in the padding between them */
            const MjImageDev &im = b->img[i];
            memset(raw + at, 0, (size_t) (im.raw_off - at));
            memcpy(raw + im.raw_off, prep[i].ent, im.raw_len);
            at = im.raw_off + im.raw_len;
        }
        memset(raw + at, 0, raw_bytes - (size_t) at);
    }

    /* the buffer list runs twice: once to add up the sizes, once to hand out the pieces */
    auto place = [&]() -> int {
        int rc = 0;
        rc |= dev_upload(b, &d.img, b->img);
        rc |= dev_upload(b, &d.sets, b->sets);
        {
            uint8_t *p = nullptr;
            rc |= dev_alloc(b, &p, raw_bytes);
            if (!rc && p && hipMemcpy(p, ctx->stage, raw_bytes, hipMemcpyHostToDevice) != hipSuccess) rc = -1;
            d.raw = p;
        }
        rc |= dev_upload(b, &d.tile_img, tile_img);
        rc |= dev_upload(b, &d.wg_img, wg_img);
        rc |= dev_upload(b, &d.idct_wg_img, idct_img);
        rc |= dev_upload(b, &d.idct_wg_first, idct_first);
        rc |= dev_alloc(b, &d.dyn, (uint64_t) n);
        rc |= dev_alloc(b, &d.cmp, cmp_total + 64);
        rc |= dev_alloc(b, &d.tile_cnt, n_tiles);
        rc |= dev_alloc(b, &d.tile_base, n_tiles);
        rc |= dev_alloc(b, &d.ivl_start, n_ivl);
        rc |= dev_alloc(b, &d.ivl_seq, n_ivl);
        rc |= dev_alloc(b, &d.seq_ivl, n_seq);
        rc |= dev_alloc(b, &d.seq_sub, n_seq);
        rc |= dev_alloc(b, &d.rec, n_rec);
        rc |= dev_alloc(b, &d.seq_entry, n_seq);
        rc |= dev_alloc(b, &d.seq_seen, n_seq);
        rc |= dev_alloc(b, &d.seq_flag, n_seq);
        rc |= dev_alloc(b, &d.multi, n_multi);
        rc |= dev_alloc(b, &d.multi_count, 4);
        rc |= dev_alloc(b, &d.coefs, coef_blocks * 64);
        rc |= dev_alloc(b, &d.rgb, rgb_total);
        rc |= dev_alloc(b, &d.planes, d.want_planes ? plane_total : 0);
        rc |= dev_alloc(b, &d.stats, (uint64_t) MJ_ST_COUNT);
        return rc;
    };
    b->measuring = true;
    b->arena_used = 0;
    int rc = place();
    if (!rc) {
        const size_t need = b->arena_used;
        if (ctx->arena && ctx->arena_cap >= need) {          /* move into what the last batch left */
            b->arena = static_cast<uint8_t *>(ctx->arena);
            b->arena_cap = ctx->arena_cap;
            ctx->arena = nullptr;
            ctx->arena_cap = 0;
        } else {
            if (ctx->arena) {
                (void) hipFree(ctx->arena);
                ctx->arena = nullptr;
                ctx->arena_cap = 0;
            }
            void *p = nullptr;
            if (hipMalloc(&p, need) != hipSuccess) {
                mj_batch_destroy(b);
                return set_err("mj_batch_create: cannot allocate %zu bytes of device memory", need);
            }
            b->arena = static_cast<uint8_t *>(p);
            b->arena_cap = need;
        }
        /* everything starts at zero, whoever used the memory before */
        if (hipMemset(b->arena, 0, need) != hipSuccess) rc = -1;
        b->measuring = false;
        b->arena_used = 0;
        if (!rc) rc = place();
    }
    if (rc) {
        mj_batch_destroy(b);
        return -1;
    }
    *out = b;
    return 0;
}

static int take_events(mj_batch *b, EventSet *es) {
    if (!b->pool.empty()) {
        *es = b->pool.back();
        b->pool.pop_back();
    } else {
        for (hipEvent_t &ev : es->ev) HIP_TRY(hipEventCreate(&ev));
    }
    return 0;
}

int mj_batch_decode(mj_batch *b) {
    if (!b) return set_err("mj_batch_decode: null batch");
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t s = b->ctx->stream;
    const bool timing = (b->flags & MJ_TIMING) != 0;
    EventSet es{};
    if (timing) {
        if (b->pending.size() >= 256 && collect_timing(b)) return -1;
        if (take_events(b, &es)) return -1;
    }
    auto mark = [&](int i) -> hipError_t { return timing ? hipEventRecord(es.ev[i], s) : hipSuccess; };

    HIP_TRY(mark(0));
    HIP_TRY(hipMemsetAsync(b->d.stats, 0, MJ_ST_COUNT * sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(b->d.multi_count, 0, 16, s));
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
        HIP_TRY(mj_launch_entropy(b->d, s));
        HIP_TRY(mark(3));
        HIP_TRY(mj_launch_entropy_fix(b->d, s));
        HIP_TRY(mj_launch_entropy_write(b->d, s));
        HIP_TRY(mark(4));
    }
    HIP_TRY(mj_launch_idct_colour(b->d, s));
    HIP_TRY(mark(5));
    if (timing) b->pending.push_back(es);
    return 0;
}

int mj_batch_sync(mj_batch *b) {
    if (!b) return set_err("mj_batch_sync: null batch");
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

int mj_batch_size(const mj_batch *b) {
    return b ? (int) b->img.size() : -1;
}

int mj_batch_image_info(const mj_batch *b, int i, mj_image_info *info) {
    if (!b || !info || i < 0 || i >= (int) b->info.size()) return set_err("mj_batch_image_info: bad arguments");
    *info = b->info[i];
    return 0;
}

int mj_batch_status(mj_batch *b, int i, uint32_t *status) {
    if (!b || !status || i < 0 || i >= (int) b->img.size()) return set_err("mj_batch_status: bad arguments");
    if (mj_batch_sync(b)) return -1;
    MjImageDyn dy;
    HIP_TRY(hipMemcpy(&dy, b->d.dyn + i, sizeof dy, hipMemcpyDeviceToHost));
    *status = dy.status;
    return 0;
}

int mj_batch_scan_end(mj_batch *b, int i, size_t *offset) {
    if (!b || !offset || i < 0 || i >= (int) b->img.size()) return set_err("mj_batch_scan_end: bad arguments");
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
    if (!b || !dst || i < 0 || i >= (int) b->img.size()) return set_err("mj_batch_copy_rgb: bad arguments");
    const MjImageDev &im = b->img[i];
    const size_t row = (size_t) (b->d.rgba ? 4 : 3) * im.X;
    if (dst_pitch < row) return set_err("mj_batch_copy_rgb: pitch smaller than a row");
    if (mj_batch_sync(b)) return -1;
    /* one contiguous copy into the context's page-locked buffer, then rows to the caller's
     * pitch on the host (a pitched copy into pageable memory is several times slower) */
    const size_t bytes = (size_t) im.pitch * im.Y;
    if (bytes == 0) return 0;
    if (stage_reserve(b->ctx, bytes)) return -1;
    HIP_TRY(hipMemcpy(b->ctx->stage, b->d.rgb + im.rgb_off, bytes, hipMemcpyDeviceToHost));
    if (dst_pitch == im.pitch) {
        memcpy(dst, b->ctx->stage, (size_t) im.pitch * (im.Y - 1) + row);
    } else {
        for (uint32_t y = 0; y < im.Y; y++) memcpy(dst + (size_t) y * dst_pitch, b->ctx->stage + (size_t) y * im.pitch, row);
    }
    return 0;
}

int mj_batch_copy_coefs(mj_batch *b, int i, int16_t *dst) {
    if (!b || !dst || i < 0 || i >= (int) b->img.size()) return set_err("mj_batch_copy_coefs: bad arguments");
    const MjImageDev &im = b->img[i];
    if (mj_batch_sync(b)) return -1;
    HIP_TRY(hipMemcpy(dst, b->d.coefs + im.coef_off * 64, (size_t) im.n_mcu * im.bpm * 128, hipMemcpyDeviceToHost));
    return 0;
}

int mj_batch_copy_plane(mj_batch *b, int i, int comp, uint8_t *dst) {
    if (!b || !dst || i < 0 || i >= (int) b->img.size()) return set_err("mj_batch_copy_plane: bad arguments");
    const MjImageDev &im = b->img[i];
    if (!b->d.want_planes) return set_err("mj_batch_copy_plane: batch was created without MJ_WANT_PLANES");
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
