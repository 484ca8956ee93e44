// CPU check of the shade kernel's work-item decode divisions (RenderParams::div_*, wavefront.hip: wf_make_fastdiv / wf_fastdiv):
// the multiplier and shift the library computes for a divisor d give EXACTLY x / d for every dividend x < 2^30.  Links against
// libmcpt_hip.so (host symbol, no GPU needed).  Prints "ok <cases>" or the first counter-example.
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
void wf_make_fastdiv(uint32_t d, uint32_t& m, uint32_t& s);
int main() {
    std::mt19937_64 rng(7);
    std::vector<uint32_t> ds;
    for (uint32_t d = 1; d <= 6000; d++) ds.push_back(d);                                 // tiles_x, widths, small tile counts
    for (int b = 1; b < 32; b++) for (int k = -2; k <= 2; k++) { const int64_t v = (int64_t(1) << b) + k; if (v >= 1 && v <= 0xffffffffll) ds.push_back(uint32_t(v)); }
    ds.push_back(0x7fffffffu); ds.push_back(0xffffffffu);                                 // probe mode's "one row of tiles"
    for (int i = 0; i < 20000; i++) ds.push_back(uint32_t(rng() >> (32 + rng() % 31)) | 1u);
    const uint32_t X = (1u << 30) - 1u;
    unsigned long long cases = 0;
    for (uint32_t d : ds) {
        uint32_t m, s; wf_make_fastdiv(d, m, s);
        if (s > 63u) { std::printf("shift %u out of range for d=%u\n", s, d); return 1; }
        auto check = [&](uint64_t x64) {
            if (x64 > X) return true;
            const uint32_t x = uint32_t(x64), q = uint32_t((uint64_t(x) * m) >> s);
            cases++;
            if (q != x / d) { std::printf("d=%u x=%u: got %u want %u (m=%u s=%u)\n", d, x, q, x / d, m, s); return false; }
            return true;
        };
        if (!check(0) || !check(X) || !check(X - 1)) return 1;
        for (int i = 0; i < 40; i++) {
            const uint64_t k = rng() % (uint64_t(X) / d + 1);                             // multiples of d and their neighbours: where a wrong multiplier shows
            if (!check(k * d) || !check(k * d + d - 1) || (k * d > 0 && !check(k * d - 1)) || !check(rng() % (uint64_t(X) + 1))) return 1;
        }
        const uint64_t top = (uint64_t(X) / d) * d;                                        // the largest multiple below the bound
        if (!check(top) || (top > 0 && !check(top - 1))) return 1;
    }
    std::printf("ok %llu\n", cases);
    return 0;
}
