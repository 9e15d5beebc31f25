// Damage fuzzer for the host side: the marker parser, the table flattening and the host
// emulation of the entropy kernels' algorithm (same mj_entropy_core.h as the device code).
// Built with -fsanitize=address,undefined by tests/test_host_cpu.py; any report aborts.
//
//   fuzz_host <iterations per file> file.jpg...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mj_host.h"

extern "C" long long mj_emu_decode_coefs(const uint8_t *data, size_t len, int16_t *out, long long cap, uint32_t force,
                                         long long *stats);
// decode_jpeg_buffer (mj_host.h -> mi355jpeg_compat.h) is the drop-in walker (decode_host.c) over stub_abi.c:
// it runs up to the scan, where the stub refuses

// Every input is handed over in an allocation of exactly its size, so that a read one byte
// past the end lands in a sanitizer red zone (a shrunk std::vector keeps its capacity).
static long long run_exact(const std::vector<uint8_t> &d, int16_t *out, long long cap, uint32_t force, long long *st) {
    uint8_t *p = (uint8_t *) malloc(d.size() ? d.size() : 1);
    if (!p) abort();
    memcpy(p, d.data(), d.size());
    const long long n = mj_emu_decode_coefs(p, d.size(), out, cap, force, st);
    {   // the marker walk over every scan of the file, as the batch front end does it for frames coded in several scans
        MjParsed P;
        const char *why = "?";
        if (mj_parse_headers(p, d.size(), &P, &why) == 0) {
            bool freed = false;
            for (int k = 0; k < 6; k++) {
                const size_t e = mj_scan_end(p, d.size(), P.ent_pos);
                if (e > d.size() || e < P.ent_pos) abort();
                const int r = mj_parse_next_scan(p, d.size(), e, &P, &why);
                if (r != 0) {
                    freed = (r < 0);
                    break;
                }
                if (P.ent_pos > d.size()) abort();
            }
            if (!freed) mj_parsed_free(&P);
        }
    }
    FrameHeader fh;
    memset(&fh, 0, sizeof fh);
    fh.process = (Encoding) -1;
    Image *img = nullptr;
    // a file cut before its scan "decodes" to blank planes, as in the reference (decode.c:417-423)
    if (decode_jpeg_buffer(p, d.size(), &fh, &img) == 0) {
        free_img(img);
        free_frame_header(&fh);
    }
    free(p);
    return n;
}

// segments [pos, pos + 2 + len) of the marker `m` in the headers of a good file
static std::vector<size_t> find_segments(const std::vector<uint8_t> &g, size_t hdr, uint8_t m) {
    std::vector<size_t> v;
    size_t i = 2;
    while (i + 4 <= hdr) {
        if (g[i] != 0xFF) break;
        const size_t len = ((size_t) g[i + 2] << 8) | g[i + 3];
        if (g[i + 1] == m) v.push_back(i);
        i += 2 + len;
    }
    return v;
}

static std::vector<uint8_t> slurp(const char *p) {
    std::vector<uint8_t> v;
    FILE *f = fopen(p, "rb");
    if (!f) return v;
    uint8_t b[4096];
    size_t n;
    while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n);
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int iters = atoi(argv[1]);
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    int runs = 0, rejected = 0;
    std::vector<int16_t> out((size_t) 20000 * 64);
    long long st[16];
    for (int a = 2; a < argc; a++) {
        const std::vector<uint8_t> good = slurp(argv[a]);
        if (good.size() < 64) return 3;
        size_t sos = 0;
        for (size_t i = 0; i + 1 < good.size(); i++) {
            if (good[i] == 0xFF && good[i + 1] == 0xDA) {
                sos = i;
                break;
            }
        }
        const size_t hdr = sos + 2 + ((good[sos + 2] << 8) | good[sos + 3]);
        const size_t ent = good.size() - hdr;
        for (int v = 0; v < iters; v++) {
            std::vector<uint8_t> d = good;
            switch (v % 11) {
                case 8: {   // a table segment is the last thing in the file, its counts / precision damaged
                    const std::vector<size_t> segs = find_segments(good, hdr, (rnd() & 1) ? 0xC4 : 0xDB);
                    if (segs.empty()) break;
                    const size_t at = segs[rnd() % segs.size()];
                    const size_t len = ((size_t) good[at + 2] << 8) | good[at + 3];
                    d.resize(at + 2 + len);
                    const size_t k = at + 4 + rnd() % (len - 2);
                    d[k] = (rnd() & 1) ? 0xFF : (uint8_t) rnd();
                    if (rnd() & 1) d[at + 4] = (uint8_t) (0x10 | (d[at + 4] & 0x0F));   // Pq = 1 / class AC
                    break;
                }
                case 9: {   // ... and claims to be shorter or longer than it is
                    const std::vector<size_t> segs = find_segments(good, hdr, (rnd() & 1) ? 0xC4 : 0xDB);
                    if (segs.empty()) break;
                    const size_t at = segs[rnd() % segs.size()];
                    const size_t len = ((size_t) good[at + 2] << 8) | good[at + 3];
                    const size_t keep = 4 + rnd() % (len - 1);
                    d.resize(at + keep);
                    const size_t claim = 2 + rnd() % (keep - 1);
                    d[at + 2] = (uint8_t) (claim >> 8);
                    d[at + 3] = (uint8_t) claim;
                    for (size_t k = at + 4; k < d.size() && k < at + 21; k++) if (rnd() & 1) d[k] = 0xFF;
                    break;
                }
                case 10: {  // the smallest such file: SOI and one table segment whose counts are all 0xFF
                    const bool dht = (rnd() & 1) != 0;
                    const size_t body = dht ? 17 + rnd() % 4 : 1 + rnd() % 130;
                    d.assign(4 + 2 + body, 0xFF);
                    d[1] = 0xD8;
                    d[3] = dht ? 0xC4 : 0xDB;
                    d[4] = (uint8_t) ((body + 2) >> 8);
                    d[5] = (uint8_t) (body + 2);
                    d[6] = dht ? (uint8_t) (rnd() & 0x11) : (uint8_t) ((rnd() % 3) << 4);
                    break;
                }
                case 0: d[2 + rnd() % (hdr - 2)] = (uint8_t) rnd(); break;                      // a header byte
                case 1: d[2 + rnd() % (hdr - 2)] ^= (uint8_t) (1u << (rnd() % 8)); break;       // a header bit
                case 2: d.resize(2 + rnd() % hdr); break;                                       // cut inside the headers
                case 3: { size_t p = 2 + rnd() % (hdr - 4); d[p] = 0xFF; d[p + 1] = (uint8_t) (0xC0 + rnd() % 0x3F); break; }
                case 4: d.resize(hdr + rnd() % ent); break;                                     // cut inside the scan
                case 5: for (int k = 0; k < 3; k++) d[hdr + rnd() % (ent - 2)] ^= (uint8_t) (1u << (rnd() % 8)); break;
                case 6: for (int k = 0; k < 3; k++) d[hdr + rnd() % (ent - 2)] = 0xFF; break;    // stray markers / stuffing
                default: { size_t p = hdr + rnd() % (ent - 4); d[p] = 0xFF; d[p + 1] = (uint8_t) (0xD0 + rnd() % 8); break; }
            }
            // the library's own cut, two forced ones, and two forms of the speculative synchronisation
            // (bits 24-31: subsequences a lane decodes past its own)
            static const uint32_t forces[5] = {0u, 128u | (1u << 16), 256u | (2u << 16), 64u | (1u << 16) | (1u << 24),
                                               256u | (2u << 16) | (3u << 24)};
            const uint32_t force = forces[v % 5];
            const long long n = run_exact(d, out.data(), 20000, force, st);
            runs++;
            if (n < 0) rejected++;
        }
    }
    printf("runs %d rejected %d\n", runs, rejected);
    return (runs > 0 && rejected > 0 && rejected < runs) ? 0 : 4;
}
