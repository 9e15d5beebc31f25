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

extern "C" long long mj_emu_decode_coefs(const uint8_t *data, size_t len, int16_t *out, long long cap, uint32_t force,
                                         long long *stats);

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
            switch (v % 8) {
                case 0: d[2 + rnd() % (hdr - 2)] = (uint8_t) rnd(); break;                      // a header byte
                case 1: d[2 + rnd() % (hdr - 2)] ^= (uint8_t) (1u << (rnd() % 8)); break;       // a header bit
                case 2: d.resize(2 + rnd() % hdr); break;                                       // cut inside the headers
                case 3: { size_t p = 2 + rnd() % (hdr - 4); d[p] = 0xFF; d[p + 1] = (uint8_t) (0xC0 + rnd() % 0x3F); break; }
                case 4: d.resize(hdr + rnd() % ent); break;                                     // cut inside the scan
                case 5: for (int k = 0; k < 3; k++) d[hdr + rnd() % (ent - 2)] ^= (uint8_t) (1u << (rnd() % 8)); break;
                case 6: for (int k = 0; k < 3; k++) d[hdr + rnd() % (ent - 2)] = 0xFF; break;    // stray markers / stuffing
                default: { size_t p = hdr + rnd() % (ent - 4); d[p] = 0xFF; d[p + 1] = (uint8_t) (0xD0 + rnd() % 8); break; }
            }
            const uint32_t force = (v % 3 == 0) ? 0u : (v % 3 == 1) ? (128u | (1u << 16)) : (256u | (2u << 16));
            const long long n = mj_emu_decode_coefs(d.data(), d.size(), out.data(), 20000, force, st);
            runs++;
            if (n < 0) rejected++;
        }
    }
    printf("runs %d rejected %d\n", runs, rejected);
    return (runs > 0 && rejected > 0 && rejected < runs) ? 0 : 4;
}
