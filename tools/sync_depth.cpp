// How soon is a decoder that starts cold at a subsequence boundary in step with the true decode --
// with one decoder per boundary (block slot 0, as the synchronisation rounds start them), with one
// per block slot of the MCU (the speculative synchronisation), with one per table class?
//   make -C jpeg-decoder_amd libmjemu.so
//   g++ -std=c++17 -O2 -Iinclude -Ijpeg-decoder_amd/host -Ijpeg-decoder_amd/csrc -o /tmp/sync_depth tools/sync_depth.cpp \
//       jpeg-decoder_amd/host/bitcursor.o jpeg-decoder_amd/host/segments.o jpeg-decoder_amd/host/flatten.o -lm
//   /tmp/sync_depth file.jpg <subsequence bits> <starts every G subsequences> [most subsequences followed]
// Prints, per strategy, for what share of the subsequence ends the youngest start that is in step
// there lies 1, 2, 3 ... subsequences back ("depth"), and how many subsequences the best variant of a
// start needs.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "mj_host.h"
#include "mj_colour_core.h"
#include "mj_entropy_core.h"

struct St { uint32_t p, k, c; };
static bool eq(const St &a, const St &b) { return a.p == b.p && a.k == b.k && a.c == b.c; }

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: file S G [maxage]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); size_t len = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> file(len);
    if (fread(file.data(), 1, len, f) != len) return 2;
    fclose(f);
    const uint32_t S = atoi(argv[2]), G = atoi(argv[3]);
    const uint32_t MAXA = argc > 4 ? atoi(argv[4]) : 80;
    MjParsed P; const char *why = nullptr;
    if (mj_parse_headers(file.data(), len, &P, &why) != 0) { fprintf(stderr, "parse: %s\n", why); return 1; }
    MjImageDev img; MjHuffSet *set = (MjHuffSet *) calloc(1, sizeof(MjHuffSet));
    if (mj_flatten_image(&P, len, &img, set, &why) != 0) { fprintf(stderr, "flatten: %s\n", why); return 1; }
    // scan
    std::vector<uint8_t> cmp; std::vector<uint32_t> ivl(img.n_ivl + 1, 0);
    {
        const uint8_t *raw = file.data() + P.ent_pos; size_t n = img.raw_len; uint32_t k = 0;
        for (size_t i = 0; i < n; i++) {
            uint32_t prev = i ? raw[i - 1] : 0, cur = raw[i], next = (i + 1 < n) ? raw[i + 1] : 0;
            int cls = mj_classify(prev, cur, next);
            if (cls == MJ_B_TERM) break;
            if (cls == MJ_B_KEEP) cmp.push_back((uint8_t) cur);
            else if (cls == MJ_B_RST) { k++; if (k < img.n_ivl) ivl[k] = (uint32_t) cmp.size(); }
        }
        for (uint32_t j = k + 1; j <= img.n_ivl; j++) ivl[j] = (uint32_t) cmp.size();
        ivl[img.n_ivl] = (uint32_t) cmp.size();
        cmp.resize(cmp.size() + MJ_STREAM_PAD + 8, 0);
        while (cmp.size() % 4) cmp.push_back(0);
    }
    MjScanCtx cx;
    cx.stream = (const uint32_t *) cmp.data();
    mj_ctx_slots(cx, img.bpm, img.blk_comp, img.blk_dc, img.blk_ac);
    cx.tabs = (const unsigned char *) set->tab;
    cx.walk = set->walk;
    cx.pairs = (const unsigned char *) set->run;
    cx.pair_stride = (uint32_t) sizeof(MjHuffRun);
    cx.coefs = nullptr;
    static const uint8_t natpos[64] = MJ_NATPOS_INIT;
    cx.natpos = natpos;
    const uint32_t bpm = img.bpm;
    fprintf(stderr, "bpm %u n_ivl %u S %u G %u\n", bpm, img.n_ivl, S, G);

    auto span = [&](St s, uint32_t end) {
        uint32_t nb = 0; int32_t d[4] = {0, 0, 0, 0};
        mj_decode_span(cx, s.p, s.k, s.c, end, nb, d, 1);
        return s;
    };

    // strategies: 0 = {c=0}; 1 = all slots; 2 = one slot per table class change (0, and first of each class)
    const int NSTRAT = 3;
    std::vector<uint32_t> depth_hist[NSTRAT];
    for (auto &h : depth_hist) h.assign(MAXA + 2, 0);
    uint64_t nsub_total = 0;
    uint32_t gmax[NSTRAT] = {0, 0, 0};
    // per group of 8 sequences (8*64*G subsequences): max depth -> mean over groups
    double wg_sum[NSTRAT] = {0, 0, 0}; uint64_t wg_n = 0;
    const uint32_t n_ivl_lim = std::min<uint32_t>(img.n_ivl, 40);
    for (uint32_t iv = 0; iv < n_ivl_lim; iv++) {
        const uint32_t b0 = ivl[iv] * 8, b1 = ivl[iv + 1] * 8;
        uint32_t nsub = (b1 - b0 + S - 1) / S; if (!nsub) nsub = 1;
        auto sub_end = [&](uint32_t j) { return (j + 1 == nsub) ? b1 : b0 + (j + 1) * S; };
        std::vector<St> T(nsub);
        St cur{b0, 0, 0};
        for (uint32_t j = 0; j < nsub; j++) { cur = span(cur, sub_end(j)); T[j] = cur; }
        // merge age per (start, variant): a = subsequences decoded until equal to truth (MAXA+1: never within MAXA)
        std::vector<uint32_t> age((size_t) nsub * bpm, MAXA + 1);
        for (uint32_t s = 0; s < nsub; s += G) {
            for (uint32_t v = 0; v < bpm; v++) {
                if (s == 0) { age[(size_t) s * bpm + v] = (v == 0) ? 1 : MAXA + 1; if (v) continue; else continue; }
                St x{b0 + s * S, 0, v};
                for (uint32_t a = 1; a <= MAXA && s + a - 1 < nsub; a++) {
                    x = span(x, sub_end(s + a - 1));
                    if (eq(x, T[s + a - 1])) { age[(size_t) s * bpm + v] = a; break; }
                }
            }
        }
        {   // per start: after how many subsequences is its best variant true
            static uint64_t hist[16] = {0}, n = 0;
            for (uint32_t st = G; st < nsub; st += G) {
                uint32_t best = MAXA + 1;
                for (uint32_t v = 0; v < bpm; v++) best = std::min(best, age[(size_t) st * bpm + v]);
                hist[std::min<uint32_t>(best, 15)]++; n++;
            }
            if (iv + 1 == n_ivl_lim) { printf("best variant of a start true after a subsequences: "); uint64_t acc = n; for (int a = 1; a < 8; a++) { printf(" >%d:%.5f", a - 1, (double) acc / n); acc -= hist[a]; } printf("\n"); }
        }
        // depth(j) per strategy
        std::vector<uint32_t> dep[NSTRAT];
        for (auto &d : dep) d.assign(nsub, 0);
        for (uint32_t j = 0; j < nsub; j++) {
            for (int st = 0; st < NSTRAT; st++) {
                uint32_t best = MAXA + 1;
                for (uint32_t s = (j / G) * G;; s -= G) {
                    const uint32_t need = j - s + 1;
                    if (need > MAXA) break;
                    bool ok = false;
                    for (uint32_t v = 0; v < bpm && !ok; v++) {
                        if (st == 0 && v != 0) continue;
                        if (st == 2 && !(v == 0 || img.blk_ac[v] != img.blk_ac[v - 1] || img.blk_dc[v] != img.blk_dc[v - 1])) continue;
                        ok = age[(size_t) s * bpm + v] <= need;
                    }
                    if (ok) { best = need; break; }
                    if (s == 0) break;
                }
                dep[st][j] = best;
                depth_hist[st][best]++;
                gmax[st] = std::max(gmax[st], best);
            }
        }
        nsub_total += nsub;
        const uint32_t WG = 8 * 64 * G;
        for (uint32_t w = 0; w * WG < nsub; w++) {
            for (int st = 0; st < NSTRAT; st++) {
                uint32_t m = 0;
                for (uint32_t j = w * WG; j < std::min(nsub, (w + 1) * WG); j++) m = std::max(m, dep[st][j]);
                wg_sum[st] += m;
            }
            wg_n++;
        }
    }
    for (int st = 0; st < NSTRAT; st++) {
        printf("strategy %d: max depth %u, mean over workgroups of max depth %.2f (rounds = depth - G = %.2f); tail:", st, gmax[st], wg_sum[st] / wg_n,
               wg_sum[st] / wg_n - G);
        uint64_t acc = 0;
        for (uint32_t a = MAXA + 1; a >= 1; a--) {
            acc += depth_hist[st][a];
            if (a <= 12 || a % 8 == 0) printf(" >=%u:%.5f", a, (double) acc / nsub_total);
        }
        printf("\n");
    }
    return 0;
}
