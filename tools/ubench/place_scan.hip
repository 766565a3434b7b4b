// place_scan — what does THIS box's VRAM allocator do?  (round 3: on some boxes of the pool the filler steering of
// placement.hip has no effect at all: spread_gain 1.00 with 177 GiB of filler.)  Allocates chunks until the device is
// nearly full and probes, with the skin stage's store pattern, every chunk PAIRED with chunk 0, with the middle chunk and
// with the last chunk: pairs in one "region" write at ~5.3 TB/s, pairs in two regions at ~6.7 TB/s.  Prints the map.
//   hipcc --offload-arch=gfx950 -O3 -o place_scan place_scan.hip ; ./place_scan [chunk_GiB=4]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void st3(char* base, unsigned off, f3 d) {
    asm volatile("global_store_dwordx3 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base) : "memory");
}
template <int ARRAYS>
__global__ __launch_bounds__(256) void k_pat(char* a0, char* a1, char* a2, unsigned V, unsigned T, unsigned I, unsigned R, unsigned ranges) {
    const unsigned Tx = (T + 7) / 8, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const unsigned tile = xcd * Tx + k % Tx, rng = k / Tx;
    if (tile >= T || rng >= ranges) return;
    if (tile * 256 + threadIdx.x >= V) return;
    const unsigned off = (tile * 256 + threadIdx.x) * 12;
    const f3 d = {0.f, 0.f, 0.f};
    const unsigned i1 = min(I, (rng + 1) * R);
    const size_t istride = (size_t)V * 12;
    for (unsigned i = rng * R; i < i1; ++i) {
        st3(a0 + (size_t)i * istride, off, d);
        if (ARRAYS > 1) st3(a1 + (size_t)i * istride, off, d);
        if (ARRAYS > 2) st3(a2 + (size_t)i * istride, off, d);
    }
}
static const unsigned V = 60000, R = 80;
static unsigned I = 0;
static float probe(int arrays, char* a, char* b, char* c, int reps = 3) {
    const unsigned T = (V + 255) / 256, ranges = (I + R - 1) / R, grid = 8 * ((T + 7) / 8) * ranges;
    static hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0));
        if (arrays == 1) hipLaunchKernelGGL(k_pat<1>, dim3(grid), dim3(256), 0, 0, a, b, c, V, T, I, R, ranges);
        else if (arrays == 2) hipLaunchKernelGGL(k_pat<2>, dim3(grid), dim3(256), 0, 0, a, b, c, V, T, I, R, ranges);
        else hipLaunchKernelGGL(k_pat<3>, dim3(grid), dim3(256), 0, 0, a, b, c, V, T, I, R, ranges);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) best = std::min(best, ms);
    }
    return best;
}
int main(int argc, char** argv) {
    const double GiB = 1073741824.0;
    const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 4) << 30;
    I = (unsigned)(chunk / ((size_t)V * 12));
    const double arr = (double)I * V * 12;
    size_t free_b, total_b; CK(hipMemGetInfo(&free_b, &total_b));
    printf("# start: used %.2f GiB of %.1f; chunk %.0f GiB = %u instances x %u vertices\n", (total_b - free_b) / GiB, total_b / GiB, chunk / GiB, I, V);
    std::vector<char*> ch; std::vector<double> at;
    while (true) {
        CK(hipMemGetInfo(&free_b, &total_b));
        if (free_b < chunk + ((size_t)3 << 30)) break;
        char* p = nullptr;
        at.push_back((total_b - free_b) / GiB);
        if (hipMalloc((void**)&p, chunk) != hipSuccess) { (void)hipGetLastError(); at.pop_back(); break; }
        ch.push_back(p);
    }
    const int n = (int)ch.size();
    printf("# %d chunks\n", n);
    const int refs[3] = {0, n / 2, n - 1};
    printf("# chunk  used-before  address          alone(GB/s)   pair with chunk %d / %d / %d (GB/s)\n", refs[0], refs[1], refs[2]);
    std::vector<std::vector<float>> rate(n, std::vector<float>(3, 0.f));
    for (int c = 0; c < n; ++c) {
        const float a = probe(1, ch[c], ch[c], ch[c]);
        printf("%4d  %8.1f GiB  %p  %8.0f   ", c, at[c], (void*)ch[c], arr / a / 1e6);
        for (int r = 0; r < 3; ++r) {
            const float ms = probe(2, ch[refs[r]], ch[c], ch[c]);
            rate[c][r] = (float)(2.0 * arr / ms / 1e6);
            printf(" %8.0f", rate[c][r]);
        }
        printf("\n");
        fflush(stdout);
    }
    // triples: the best pairwise-far triple by the table, and consecutive ones
    auto tri = [&](int a, int b, int c, const char* what) {
        const float ms = probe(3, ch[a], ch[b], ch[c]);
        printf("triple (%3d,%3d,%3d) %-22s : %6.3f ms %7.0f GB/s\n", a, b, c, what, ms, 3.0 * arr / ms / 1e6);
    };
    tri(0, 1, 2, "first three");
    tri(0, n / 2, n - 1, "first / middle / last");
    tri(0, n / 3, 2 * n / 3, "thirds by order");
    tri(n / 6, n / 2, 5 * n / 6, "centres of thirds");
    // free every other chunk and allocate again: does a freed hole get reused, and where does a new chunk land?
    for (int c = 1; c < n; c += 2) { CK(hipFree(ch[c])); ch[c] = nullptr; }
    char* again = nullptr;
    CK(hipMalloc((void**)&again, chunk));
    printf("# after freeing the odd chunks a new chunk lands at %p; pair with chunk 0: %.0f GB/s, with chunk %d: %.0f GB/s\n", (void*)again,
           2.0 * arr / probe(2, ch[0], again, again) / 1e6, (n - 1) & ~1, 2.0 * arr / probe(2, ch[(n - 1) & ~1], again, again) / 1e6);
    return 0;
}
