// Fuzz driver for the host-only game-line formatter (ataxxzero_amd/csrc/json.cpp), built with -fsanitize=address,undefined by
// tests/test_json_format.py: well-formed and damaged finished-game records, exact-size heap copies (an over-read is a report),
// small and large output buffers.  The GPU boxes admit no sanitizer runs; this part of the library needs no GPU.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
extern "C" int azh_format_record_json(const uint32_t *rec, int64_t words, int32_t with_ids, char *buf, int64_t cap, int64_t *used);
int azh_fail(int code, const char *fmt, ...) { (void)fmt; return code; }
int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 rng(12345);
    std::vector<char> buf(1 << 20);
    long ok = 0, refused = 0, small = 0;
    for (int it = 0; it < iters; it++) {
        // a well-formed record ...
        std::vector<uint32_t> r = {0x415A4847u, (uint32_t)(rng() % 4096), (uint32_t)rng(), 0, (uint32_t)(1 + rng() % 2), 0, (uint32_t)(rng() % 3 == 0 ? 1 + rng() % 120 : 0), 0};
        const int plies = (int)(rng() % 12);
        for (int p = 0; p < plies; p++) {
            const uint32_t nd = (uint32_t)(rng() % 20);
            r.push_back((uint32_t)rng()); r.push_back((uint32_t)(rng() & 0x1FFFF)); r.push_back((uint32_t)rng()); r.push_back((uint32_t)(rng() & 0x1FFFF));
            r.push_back((uint32_t)((rng() % 49) | ((rng() % 49) << 8) | (nd << 16))); r.push_back(0);
            for (uint32_t j = 0; j < nd; j++)
                r.push_back((uint32_t)((rng() % 49) | ((rng() % 49) << 8) | ((rng() % 60001) << 16)));
        }
        r[3] = (uint32_t)plies;
        r[5] = (uint32_t)r.size();
        // ... then, two times out of three, damaged: random words overwritten, the length cut or the header lied about
        const int mode = (int)(rng() % 6);
        std::vector<uint32_t> m = r;
        if (mode == 1) for (int k = 0; k < 3; k++) m[rng() % m.size()] = (uint32_t)rng();
        if (mode == 2) m.resize(rng() % (m.size() + 1));
        if (mode == 3) m[3] = (uint32_t)rng();
        if (mode == 4) m[5] = (uint32_t)rng();
        if (mode == 5) m[8 + (m.size() > 13 ? 4 : 0) < m.size() ? 8 + 4 : 0] |= 0xFFFF0000u;
        // exact-size heap copy so that any over-read is an ASan report
        uint32_t *heap = (uint32_t *)malloc(m.size() * 4 + 4);
        memcpy(heap, m.data(), m.size() * 4);
        int64_t used = -1;
        const int64_t cap = (rng() % 8 == 0) ? (int64_t)(rng() % 64) : (int64_t)buf.size();
        const int rc = azh_format_record_json(heap, (int64_t)m.size(), (int)(rng() & 1), buf.data(), cap, &used);
        free(heap);
        if (rc == 0) ok++; else if (rc == -6) small++; else refused++;
        if (rc == 0 && (used <= 0 || used > cap)) { fprintf(stderr, "bad used\n"); return 1; }
    }
    printf("formatted %ld, refused %ld, buffer too small %ld\n", ok, refused, small);
    return 0;
}
