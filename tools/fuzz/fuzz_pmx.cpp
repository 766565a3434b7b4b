// Robustness check of the PMX loader (renderer_amd/host/pmx_loader.cpp) + rig compiler under ASan /
// UBSan: a valid model (argv[1], written by tests/pmx_writer.py) is mutated — random bytes, spliced
// extreme integers, truncations — and every variant must load or be rejected with an error code.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../renderer_amd/csrc/rig_compile.hpp"
#include "../../renderer_amd/host/host_capi.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: fuzz_pmx file.pmx iterations [seed]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    std::vector<uint8_t> base;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) base.insert(base.end(), buf, buf + n);
    fclose(f);
    const long iters = atol(argv[2]);
    if (argc > 3) rng_state ^= strtoull(argv[3], nullptr, 0) * 0x2545F4914F6CDD1Dull;
    long ok = 0, err = 0;
    for (long it = 0; it < iters; ++it) {
        std::vector<uint8_t> m = base;
        switch (rnd() % 4) {
            case 0: for (int k = 0, c = 1 + (int)(rnd() % 6); k < c; ++k) m[rnd() % m.size()] = (uint8_t)rnd(); break;
            case 1: { static const uint32_t vals[] = {0xffffffffu, 0x7fffffffu, 0x80000000u, 0, 0xfffffffeu, 1u << 30};
                      const uint32_t v = vals[rnd() % 6]; if (m.size() > 8) memcpy(&m[rnd() % (m.size() - 4)], &v, 4); } break;
            case 2: m.resize(rnd() % (m.size() + 1)); break;
            default: for (int k = 0, c = 1 + (int)(rnd() % 3); k < c; ++k) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
        }
        skinrt_host_rig* rig = nullptr;
        const int rc = skinrt_host_load_pmx(m.data(), m.size(), &rig);
        if (rc == SKINRT_OK && rig) {
            skinrt::CompiledRig compiled;
            std::string why;
            (void)skinrt::compile_rig(skinrt_host_rig_desc(rig), nullptr, compiled, why);
            ++ok;
        } else {
            ++err;
        }
        if (rig) skinrt_host_free_rig(rig);
    }
    printf("fuzz_pmx: %ld variants, %ld loaded, %ld rejected\n", iters, ok, err);
    return 0;
}
