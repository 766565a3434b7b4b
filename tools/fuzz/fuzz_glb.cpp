// Robustness check of the host side (glb loader -> flattener -> rig validation/compilation) under
// AddressSanitizer + UBSan: a valid .glb (argv[1], written by tests/glb_writer.py) is mutated —
// random byte flips in the JSON chunk, in the binary chunk, truncations, spliced integers — and
// every variant must come back as SKINRT_OK or an error code, never as a crash, an out-of-bounds
// read or a hang.  Host code only; no GPU.  Built and run by tools/fuzz/run.sh.
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
    if (argc < 3) { fprintf(stderr, "usage: fuzz_glb file.glb iterations [seed]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    std::vector<uint8_t> base;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) base.insert(base.end(), buf, buf + n);
    fclose(f);
    const long iters = atol(argv[2]);
    if (argc > 3) rng_state ^= strtoull(argv[3], nullptr, 0) * 0x2545F4914F6CDD1Dull;
    const uint32_t json_len = base.size() >= 20 ? (uint32_t)base[12] | (uint32_t)base[13] << 8 | (uint32_t)base[14] << 16 | (uint32_t)base[15] << 24 : 0;
    long ok = 0, err = 0;
    for (long it = 0; it < iters; ++it) {
        std::vector<uint8_t> m = base;
        const int kind = (int)(rnd() % 6);
        const size_t json_beg = 20, json_end = std::min<size_t>(m.size(), 20 + (size_t)json_len);
        if (kind == 0) {  // byte flips anywhere
            for (int k = 0, c = 1 + (int)(rnd() % 8); k < c; ++k) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));
        } else if (kind == 1 && json_end > json_beg) {  // a digit in the JSON becomes another digit
            for (int tries = 0; tries < 64; ++tries) {
                const size_t p = json_beg + rnd() % (json_end - json_beg);
                if (m[p] >= '0' && m[p] <= '9') { m[p] = (uint8_t)('0' + rnd() % 10); if (rnd() % 2) break; }
            }
        } else if (kind == 2) {  // truncate
            m.resize(rnd() % (m.size() + 1));
        } else if (kind == 3 && json_end > json_beg) {  // a huge number spliced over digits
            for (int tries = 0; tries < 64; ++tries) {
                const size_t p = json_beg + rnd() % (json_end - json_beg);
                if (m[p] >= '0' && m[p] <= '9' && p + 10 < json_end) { memcpy(&m[p], "4294967295", 10); break; }
            }
        } else if (kind == 4) {  // header fields
            if (m.size() > 20) m[8 + rnd() % 12] = (uint8_t)rnd();
        } else {  // random bytes in the binary chunk
            if (m.size() > json_end + 16) for (int k = 0; k < 16; ++k) m[json_end + 8 + rnd() % (m.size() - json_end - 8)] = (uint8_t)rnd();
        }
        skinrt_host_rig* rig = nullptr;
        const int rc = skinrt_host_load_glb(m.data(), m.size(), -1, &rig);
        if (rc == SKINRT_OK && rig) {
            const skinrt_rig_desc* desc = skinrt_host_rig_desc(rig);
            {
                skinrt::CompiledRig compiled;  // what skinrt_upload_rig / skinrt_validate_rig run on the host
                std::string why;
                (void)skinrt::compile_rig(desc, nullptr, compiled, why);
            }
            // second phase: the descriptor itself (the C ABI's input) with hostile VALUES — indices,
            // parents, channel targets, key times — while the arrays keep their declared lengths
            auto hostile = [&]() -> uint32_t {
                switch (rnd() % 5) { case 0: return 0; case 1: return 0xffffffffu; case 2: return 0x7fffffffu; case 3: return desc->node_count; default: return (uint32_t)rnd(); }
            };
            for (int k = 0, c = 1 + (int)(rnd() % 4); k < c; ++k) {
                switch (rnd() % 9) {
                    case 7: if (desc->skin_count) {  // long form (ABI 6): a skin's joint -> node table
                                const skinrt_skin_desc* sk = &desc->skins[rnd() % desc->skin_count];
                                if (sk->joint_count) const_cast<uint32_t*>(sk->joint_node)[rnd() % sk->joint_count] = hostile();
                            } break;
                    case 8: if (desc->section_count) {  // a section's node / skin / vertex range / target range
                                skinrt_section_desc* sc = const_cast<skinrt_section_desc*>(&desc->sections[rnd() % desc->section_count]);
                                switch (rnd() % 6) { case 0: sc->node = hostile(); break; case 1: sc->skin = (int32_t)hostile(); break;
                                                     case 2: sc->first_vertex = hostile(); break; case 3: sc->vertex_count = hostile(); break;
                                                     case 4: sc->first_target = hostile(); break; default: sc->target_count = hostile(); }
                            } break;
                    case 0: if (desc->node_count) const_cast<int32_t*>(desc->parent)[rnd() % desc->node_count] = (int32_t)hostile(); break;
                    case 1: if (desc->joint_count) const_cast<uint32_t*>(desc->joint_node)[rnd() % desc->joint_count] = hostile(); break;
                    case 2: if (desc->vertex_count && desc->set_count) const_cast<uint16_t*>(desc->joints[rnd() % desc->set_count])[rnd() % (desc->vertex_count * 4)] = (uint16_t)hostile(); break;
                    case 3: if (desc->vertex_count && desc->set_count) { uint32_t b = hostile(); float w; memcpy(&w, &b, 4); const_cast<float*>(desc->weights[rnd() % desc->set_count])[rnd() % (desc->vertex_count * 4)] = w; } break;
                    case 4: if (desc->clip_count && desc->clips[0].channel_count) {
                                skinrt_channel_desc* ch = const_cast<skinrt_channel_desc*>(&desc->clips[0].channels[rnd() % desc->clips[0].channel_count]);
                                switch (rnd() % 4) { case 0: ch->target_node = hostile(); break; case 1: ch->path = (uint8_t)(3 + rnd() % 253); break;           /* invalid enum values only: a valid other path */
                                                     case 2: ch->interpolation = (uint8_t)(3 + rnd() % 253); break;  /* would change the declared length of values[] */ default: if (ch->key_count) { uint32_t b = hostile(); float t; memcpy(&t, &b, 4); const_cast<float*>(ch->times)[rnd() % ch->key_count] = t; } }
                            } break;
                    case 5: if (desc->target_count) {
                                const skinrt_morph_target_desc* t = &desc->targets[rnd() % desc->target_count];
                                if (t->indices && t->count) const_cast<uint32_t*>(t->indices)[rnd() % t->count] = hostile();
                            } break;
                    default: if (desc->vertex_count) { uint32_t b = hostile(); float x; memcpy(&x, &b, 4); const_cast<float*>(desc->position)[rnd() % (desc->vertex_count * 3)] = x; } break;
                }
            }
            {
                skinrt::CompiledRig compiled;
                std::string why;
                (void)skinrt::compile_rig(desc, nullptr, compiled, why);
            }
            ++ok;
        } else {
            ++err;
        }
        if (rig) skinrt_host_free_rig(rig);
    }
    printf("fuzz_glb: %ld variants, %ld loaded, %ld rejected\n", iters, ok, err);
    return 0;
}
