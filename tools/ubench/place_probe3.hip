// Test of the "three 96 GiB regions" model of MI355X HBM placement (see place_probe.hip / DESIGN.md):
// a write stream confined to one third of the physical address space runs at ~5.4 TB/s; streams
// spread over two or three thirds reach 6.7 - 7.0 TB/s.  Part 1 picks chunk triples by predicted
// region; part 2 rehearses the library's steering: pos, filler, nrm, filler, tan with filler sizes
// predicted from hipMemGetInfo, fillers freed afterwards.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
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
    const bool valid = tile * 256 + threadIdx.x < V;
    const unsigned off = (tile * 256 + threadIdx.x) * 12;
    float x = (float)threadIdx.x, y = x + 1.f, z = x + 2.f;
    const unsigned i1 = min(I, (rng + 1) * R);
    const size_t istride = (size_t)V * 12;
    for (unsigned i = rng * R; i < i1; ++i) {
        x += 1.f; y += 2.f; z += 3.f;
        if (!valid) continue;
        st3(a0 + (size_t)i * istride, off, f3{x, y, z});
        if (ARRAYS > 1) st3(a1 + (size_t)i * istride, off, f3{y, z, x});
        if (ARRAYS > 2) st3(a2 + (size_t)i * istride, off, f3{z, x, y});
    }
}
static const unsigned V = 60000, I = 10000, R = 80;
static float probe(int arrays, char* a, char* b, char* c, int reps = 3) {
    const unsigned T = (V + 255) / 256, ranges = (I + R - 1) / R, grid = 8 * ((T + 7) / 8) * ranges;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
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
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}
int main() {
    const double GiB = 1073741824.0;
    const size_t chunk = (size_t)8 << 30, arr = (size_t)I * V * 12;
    size_t free_b, total_b; CK(hipMemGetInfo(&free_b, &total_b));
    printf("# start: used %.2f GiB of %.1f\n", (total_b - free_b) / GiB, total_b / GiB);
    {
        std::vector<char*> ch; std::vector<double> at;
        while (true) {
            CK(hipMemGetInfo(&free_b, &total_b));
            if (free_b < chunk + ((size_t)6 << 30)) break;
            char* p = nullptr;
            at.push_back((total_b - free_b) / GiB);
            if (hipMalloc((void**)&p, chunk) != hipSuccess) { (void)hipGetLastError(); break; }
            ch.push_back(p);
        }
        printf("# %zu chunks; predicted start (GiB used before the chunk):", ch.size());
        for (size_t c = 0; c < ch.size(); ++c) printf(" %zu:%.1f", c, at[c]);
        printf("\n");
        auto tri = [&](int a, int b, int c, const char* what) {
            printf("triple (%2d,%2d,%2d) %-26s : %6.3f ms %7.1f GB/s\n", a, b, c, what, probe(3, ch[a], ch[b], ch[c]), 3.0 * arr / probe(3, ch[a], ch[b], ch[c]) / 1e6);
        };
        auto pair = [&](int a, int b, const char* what) {
            const float ms = probe(2, ch[a], ch[b], ch[b]);
            printf("pair   (%2d,%2d)    %-26s : %6.3f ms %7.1f GB/s\n", a, b, what, ms, 2.0 * arr / ms / 1e6);
        };
        tri(0, 1, 2, "R0 R0 R0"); tri(0, 5, 9, "R0 R0 R0"); tri(13, 16, 20, "R1 R1 R1"); tri(25, 29, 33, "R2 R2 R2");
        tri(0, 1, 13, "R0 R0 R1"); tri(0, 13, 14, "R0 R1 R1"); tri(0, 1, 25, "R0 R0 R2"); tri(13, 25, 26, "R1 R2 R2");
        tri(0, 13, 25, "R0 R1 R2"); tri(5, 17, 29, "R0 R1 R2"); tri(9, 20, 33, "R0 R1 R2"); tri(25, 13, 0, "R2 R1 R0");
        pair(0, 1, "R0 R0"); pair(0, 13, "R0 R1"); pair(0, 25, "R0 R2"); pair(13, 25, "R1 R2"); pair(13, 14, "R1 R1");
        for (int c = 8; c <= 14; ++c) pair(0, c, "R0 vs chunk c (boundary scan)");
        for (int c = 20; c <= 26; ++c) pair(13, c, "R1 vs chunk c (boundary scan)");
        for (char* p : ch) CK(hipFree(p));
    }
    // part 2: steering by fillers sized from hipMemGetInfo
    for (int rep = 0; rep < 2; ++rep) {
        const size_t region = total_b / 3;
        char *pos, *nrm, *tan, *f1 = nullptr, *f2 = nullptr, *pre = nullptr;
        if (rep == 1) CK(hipMalloc((void**)&pre, (size_t)13 << 30));  // something else already lives on the device
        CK(hipMemGetInfo(&free_b, &total_b));
        const size_t a0 = total_b - free_b;
        CK(hipMalloc((void**)&pos, arr));
        const size_t r0 = a0 / region;
        CK(hipMemGetInfo(&free_b, &total_b));
        size_t front = total_b - free_b;
        size_t want = (r0 + 1) * region + ((size_t)256 << 20);
        if (want > front) CK(hipMalloc((void**)&f1, want - front));
        CK(hipMalloc((void**)&nrm, arr));
        CK(hipMemGetInfo(&free_b, &total_b));
        front = total_b - free_b;
        want = (r0 + 2) * region + ((size_t)256 << 20);
        if (want > front && want + arr < total_b) CK(hipMalloc((void**)&f2, want - front));
        CK(hipMalloc((void**)&tan, arr));
        printf("steer rep %d: a0 %.2f GiB, r0 %zu; filler1 %s filler2 %s\n", rep, a0 / GiB, r0, f1 ? "yes" : "no", f2 ? "yes" : "no");
        printf("  with fillers held : 3 arrays %6.3f ms | pos alone %6.3f ms\n", probe(3, pos, nrm, tan), probe(1, pos, pos, pos));
        if (f1) CK(hipFree(f1));
        if (f2) CK(hipFree(f2));
        printf("  fillers freed     : 3 arrays %6.3f ms\n", probe(3, pos, nrm, tan));
        // a plain allocation of the same three arrays right after, for contrast
        char *p2, *n2, *t2;
        CK(hipMalloc((void**)&p2, arr)); CK(hipMalloc((void**)&n2, arr)); CK(hipMalloc((void**)&t2, arr));
        printf("  plain set after   : 3 arrays %6.3f ms\n", probe(3, p2, n2, t2));
        CK(hipFree(p2)); CK(hipFree(n2)); CK(hipFree(t2));
        CK(hipFree(pos)); CK(hipFree(nrm)); CK(hipFree(tan));
        if (pre) CK(hipFree(pre));
    }
    return 0;
}
