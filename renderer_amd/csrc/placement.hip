// placement.hip — where the three output arrays (positions / normals / tangents) live in HBM.
//
// Measured on MI355X (tools/ubench/place_probe{,2,3}.hip, place_scan.hip; profiles/r02_placement.txt,
// profiles/r03_place_scan.txt): the 288 GiB of HBM3E behave as THREE regions of one third of the physical address
// space each (96 GiB; 36 GB stacks are three ranks deep).  Write streams confined to one region top out at ~5.4 TB/s
// whatever the store shape, occupancy or offsets inside the region; the same streams spread over two regions reach
// 6.7-6.9 TB/s and over three ~7.0 TB/s.  The skin stage writes three equal streams, so the natural layout is one array
// per region.  hipMalloc gives no control over physical placement and what the VRAM manager does differs between
// boxes of one pool (round 2 steered by arithmetic on hipMemGetInfo — "lowest-first, bytes in use = the frontier" —
// which worked on the boxes of that round and did nothing at all on boxes of round 3: spread_gain 1.00 behind 177 GiB
// of filler).  So nothing is assumed any more: the placement is SEARCHED BY MEASUREMENT.  Positions are allocated
// first; further candidates of the same size are allocated one after the other — each pushes the allocator on — and
// every candidate is timed TOGETHER with the arrays already chosen, with the stage's own store pattern: a pair in two
// regions writes ~1.3x faster than twice one array alone.  The first candidate that is "far" from positions becomes
// normals, the first that is far from both becomes tangents; the candidates that were not taken (and the stride
// fillers between them) are freed before the call returns.  The result is then measured once more, all three
// together against one alone, and that measurement is what skinrt_get_output_placement reports.
#include "device_types.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace skinrt {

namespace {

typedef float float3v __attribute__((ext_vector_type(3)));

// The skin stage's output address pattern with no arithmetic: block = (tile of 256 vertices, range of
// 80 instances), consecutive blocks round-robin over the 8 XCDs, per instance every wave writes
// 768 contiguous bytes to each array (global_store_dwordx3, streaming).
template <int ARRAYS>
__global__ __launch_bounds__(256) void k_store_probe(char* a0, char* a1, char* a2, uint32_t V, uint32_t T, uint32_t I,
                                                     uint32_t R, uint32_t ranges) {
    const uint32_t Tx = (T + 7) / 8, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const uint32_t tile = xcd * Tx + k % Tx, rng = k / Tx;
    if (tile >= T || rng >= ranges) return;
    if (tile * kTileVerts + threadIdx.x >= V) return;
    const uint32_t off = (tile * kTileVerts + threadIdx.x) * 12;
    const float3v d = {0.0f, 0.0f, 0.0f};
    const uint32_t i1 = min(I, (rng + 1) * R);
    const size_t istride = (size_t)V * 12;
    for (uint32_t i = rng * R; i < i1; ++i) {
        char* p = a0 + (size_t)i * istride;
        asm volatile("global_store_dwordx3 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(d), "s"(p) : "memory");
        if (ARRAYS >= 2) {
            char* n = a1 + (size_t)i * istride;
            asm volatile("global_store_dwordx3 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(d), "s"(n) : "memory");
        }
        if (ARRAYS == 3) {
            char* t = a2 + (size_t)i * istride;
            asm volatile("global_store_dwordx3 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(d), "s"(t) : "memory");
        }
    }
}

hipError_t time_probe(int arrays, char* a0, char* a1, char* a2, uint32_t V, uint32_t I, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                      float* best_ms, int reps = 3) {
    const uint32_t T = (V + kTileVerts - 1) / kTileVerts, R = 80, ranges = (I + R - 1) / R;
    const dim3 grid(8u * ((T + 7) / 8) * ranges), block(256);
    *best_ms = 3.0e38f;
    for (int rep = 0; rep < reps; ++rep) {  // the first run also warms the page tables
        hipError_t e = hipEventRecord(e0, s);
        if (e != hipSuccess) return e;
        if (arrays == 1) hipLaunchKernelGGL(k_store_probe<1>, grid, block, 0, s, a0, a1, a2, V, T, I, R, ranges);
        else if (arrays == 2) hipLaunchKernelGGL(k_store_probe<2>, grid, block, 0, s, a0, a1, a2, V, T, I, R, ranges);
        else hipLaunchKernelGGL(k_store_probe<3>, grid, block, 0, s, a0, a1, a2, V, T, I, R, ranges);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) return e;
        if (rep) *best_ms = std::min(*best_ms, ms);
    }
    return hipSuccess;
}

constexpr size_t kSteerMinSetBytes = (size_t)1 << 30;     // smaller output sets are not worth placing
constexpr size_t kSearchStride = (size_t)6 << 30;         // the allocator is pushed on by at least this much per candidate
constexpr size_t kSearchReserve = (size_t)3 << 30;        // never take the device's last bytes for the search
constexpr uint32_t kSearchMaxCandidates = 64;
constexpr float kFar = 1.15f;                             // 2 * alone / pair: 1.0 = one region, ~1.3 = two
constexpr float kSpreadConfirmed = 1.15f;                 // 3 * alone / together: 1.0 = one region, 1.25-1.3 = three

bool device_has_regions(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return false; }
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

}  // namespace

hipError_t allocate_outputs(int device, size_t bytes_each, uint32_t V, uint32_t I, uint32_t mode, uint32_t max_filler_gib, hipStream_t s,
                            void** pos, void** nrm, void** tan, skinrt_placement_info* info) {
    std::memset(info, 0, sizeof *info);
    *pos = *nrm = *tan = nullptr;
    if (bytes_each == 0) return hipSuccess;
    auto release_all = [&]() {
        for (void** p : {pos, nrm, tan}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    };
    auto plain = [&]() -> hipError_t {
        for (void** p : {pos, nrm, tan}) {
            hipError_t e = hipMalloc(p, bytes_each);
            if (e != hipSuccess) { release_all(); return e; }
        }
        return hipSuccess;
    };
    const bool search = mode == 0 && 3 * bytes_each >= kSteerMinSetBytes && device_has_regions(device);
    if (!search) return plain();

    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e != hipSuccess) { if (e0) (void)hipEventDestroy(e0); return e; }
    auto done_events = [&]() { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); };

    // ---- the search ------------------------------------------------------------------------------------
    const size_t cap = max_filler_gib ? (size_t)max_filler_gib << 30 : ~(size_t)0;
    const size_t stride_fill = bytes_each < kSearchStride ? kSearchStride - bytes_each : 0;
    std::vector<void*> held;      // candidates not taken + stride fillers: freed below
    size_t held_bytes = 0;
    bool no_room = false, searched = false;
    float alone = 0.0f;
    e = hipMalloc(pos, bytes_each);
    if (e == hipSuccess) e = time_probe(1, (char*)*pos, nullptr, nullptr, V, I, s, e0, e1, &alone, 2);
    auto far_from = [&](void* a, void* b, bool* far) -> hipError_t {
        float pair = 0.0f;
        hipError_t pe = time_probe(2, (char*)a, (char*)b, nullptr, V, I, s, e0, e1, &pair, 2);
        *far = pe == hipSuccess && pair > 0.0f && 2.0f * alone / pair >= kFar;
        if (*far) {  // a candidate is only taken on a second opinion (other work on the device can make one run look fast or slow)
            float again = 0.0f, alone2 = 0.0f;
            pe = time_probe(2, (char*)a, (char*)b, nullptr, V, I, s, e0, e1, &again, 3);
            if (pe == hipSuccess) pe = time_probe(1, (char*)a, nullptr, nullptr, V, I, s, e0, e1, &alone2, 3);
            *far = pe == hipSuccess && again > 0.0f && 2.0f * alone2 / again >= kFar;
        }
        return pe;
    };
    std::vector<void*> near_pos;  // candidates in positions' own region, in allocation order (fallback choices)
    for (uint32_t k = 0; e == hipSuccess && !*tan && k < kSearchMaxCandidates; ++k) {
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); break; }
        if (f < bytes_each + kSearchReserve || held_bytes + bytes_each > cap) { no_room = true; break; }
        void* cand = nullptr;
        if (hipMalloc(&cand, bytes_each) != hipSuccess) { (void)hipGetLastError(); no_room = true; break; }
        searched = true;
        bool far_pos = false, far_nrm = false;
        e = far_from(*pos, cand, &far_pos);
        if (e == hipSuccess && *nrm && far_pos) e = far_from(*nrm, cand, &far_nrm);
        if (e != hipSuccess) { (void)hipFree(cand); break; }
        if (!*nrm && far_pos) { *nrm = cand; continue; }
        if (*nrm && far_pos && far_nrm) { *tan = cand; break; }
        held.push_back(cand);
        held_bytes += bytes_each;
        if (!far_pos) near_pos.push_back(cand);
        // push the allocator on by a stride even when the arrays are small
        if (stride_fill && f >= bytes_each + stride_fill + kSearchReserve && held_bytes + stride_fill <= cap) {
            void* fill = nullptr;
            if (hipMalloc(&fill, stride_fill) == hipSuccess) { held.push_back(fill); held_bytes += stride_fill; }
            else (void)hipGetLastError();
        }
    }
    // what the search did not find is taken from what it holds (two regions are still better than one), else plainly
    auto take_held = [&](void** dst) {
        if (*dst) return;
        void* pick = nullptr;
        for (void* h : near_pos) if (std::find(held.begin(), held.end(), h) != held.end()) { pick = h; break; }
        if (!pick) return;
        held.erase(std::find(held.begin(), held.end(), pick));
        near_pos.erase(std::find(near_pos.begin(), near_pos.end(), pick));
        *dst = pick;
    };
    if (e == hipSuccess) { take_held(nrm); take_held(tan); }
    info->filler_bytes = held_bytes;
    for (void* h : held) (void)hipFree(h);
    held.clear();
    if (e == hipSuccess && !*nrm) e = hipMalloc(nrm, bytes_each);
    if (e == hipSuccess && !*tan) e = hipMalloc(tan, bytes_each);
    if (e != hipSuccess) {  // no room for any of it: take what a plain allocation gives
        (void)hipGetLastError();
        release_all();
        done_events();
        info->filler_bytes = 0;
        return plain();
    }

    // measure what came out of it
    float together = 0.0f;
    e = time_probe(3, (char*)*pos, (char*)*nrm, (char*)*tan, V, I, s, e0, e1, &together);
    if (e == hipSuccess) e = time_probe(1, (char*)*pos, nullptr, nullptr, V, I, s, e0, e1, &alone);
    done_events();
    if (e != hipSuccess) { release_all(); return e; }
    info->probe_together_ms = together;
    info->probe_alone_ms = alone;
    info->spread_gain = together > 0.0f ? 3.0f * alone / together : 0.0f;
    info->probe_GBps = together > 0.0f ? (float)(3.0 * (double)bytes_each / (together * 1e-3) / 1e9) : 0.0f;
    // what happened, for the caller (skinrt_placement_info.strategy): the measurement decides, not the intention
    if (info->spread_gain >= kSpreadConfirmed) info->strategy = 1u;
    else info->strategy = no_room ? 3u : (searched ? 2u : 3u);
    return hipSuccess;
}

}  // namespace skinrt
