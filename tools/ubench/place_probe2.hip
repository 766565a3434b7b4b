// Which RELATIVE placement of the three output arrays is fast?  One 72 GiB allocation; the skin stage's
// output pattern (10,000 x 60,000 x 12 B per array, block = tile x 80 instances, XCD-chunked) is
// timed with pos at base + s, nrm at base + s + D1, tan at base + s + D2 for swept s, D1, D2.
// place_probe.hip showed: each array alone always writes at the same rate; three arrays 8 GiB + 2 MiB
// apart are mostly 25 % slower than three arrays at irregular distances.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void st3(char* base, unsigned off, f3 d) {
    asm volatile("global_store_dwordx3 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base) : "memory");
}
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
        st3(a1 + (size_t)i * istride, off, f3{y, z, x});
        st3(a2 + (size_t)i * istride, off, f3{z, x, y});
    }
}
static const unsigned V = 60000, I = 10000, R = 80;
static char* g_base;
static float probe(size_t s, size_t D1, size_t D2, int reps = 3) {
    const unsigned T = (V + 255) / 256, ranges = (I + R - 1) / R, grid = 8 * ((T + 7) / 8) * ranges;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_pat, dim3(grid), dim3(256), 0, 0, g_base + s, g_base + s + D1, g_base + s + D2, V, T, I, R, ranges);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) best = std::min(best, ms);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}
int main() {
    const size_t MiB = 1u << 20, GiB = 1u << 30;
    const size_t total = 72 * GiB;
    CK(hipMalloc((void**)&g_base, total));
    const size_t arr = (size_t)I * V * 12;
    const size_t A = (arr + 2 * MiB - 1) / (2 * MiB) * (2 * MiB);
    printf("# base %p, array %zu B, A = %zu (%zu pages of 2 MiB)\n", (void*)g_base, arr, A, A / (2 * MiB));
    printf("# sweep 1: D1 = A + k*2MiB, D2 = 2*D1\n");
    for (unsigned k = 0; k < 96; ++k) { const size_t D1 = A + k * 2 * MiB; printf("s1 k %3u : %6.3f ms\n", k, probe(0, D1, 2 * D1)); }
    printf("# sweep 2: D1 = A + k*256MiB, D2 = 2*D1\n");
    for (unsigned k = 0; k < 40; ++k) { const size_t D1 = A + k * 256 * MiB; printf("s2 k %3u : %6.3f ms\n", k, probe(0, D1, 2 * D1)); }
    printf("# sweep 3: D1 = A, D2 = 2A + k*2MiB\n");
    for (unsigned k = 0; k < 64; ++k) printf("s3 k %3u : %6.3f ms\n", k, probe(0, A, 2 * A + k * 2 * MiB));
    printf("# sweep 4: D1 = A + d, D2 = 2A  (d in bytes)\n");
    for (size_t d = 64; d <= 1 * MiB; d *= 2) printf("s4 d %8zu : %6.3f ms\n", d, probe(0, A + d, 2 * A));
    printf("# sweep 5: D1 = A, D2 = 2A + d\n");
    for (size_t d = 64; d <= 1 * MiB; d *= 2) printf("s5 d %8zu : %6.3f ms\n", d, probe(0, A, 2 * A + d));
    printf("# sweep 6: absolute shift s = k*2MiB, D1 = A, D2 = 2A\n");
    for (unsigned k = 0; k < 48; ++k) printf("s6 k %3u : %6.3f ms\n", k, probe(k * 2 * MiB, A, 2 * A));
    printf("# sweep 7: absolute shift s = k*1GiB, D1 = A, D2 = 2A\n");
    for (unsigned k = 0; k < 40; ++k) printf("s7 k %3u : %6.3f ms\n", k, probe(k * GiB, A, 2 * A));
    printf("# sweep 8: D1 = k*1GiB (k >= 7), D2 = 2*D1\n");
    for (unsigned k = 7; k < 30; ++k) printf("s8 k %3u : %6.3f ms\n", k, probe(0, k * GiB, 2 * k * GiB));
    printf("# sweep 9: D1 = 8GiB + k*2MiB, D2 = 2*D1\n");
    for (unsigned k = 0; k < 64; ++k) { const size_t D1 = 8 * GiB + k * 2 * MiB; printf("s9 k %3u : %6.3f ms\n", k, probe(0, D1, 2 * D1)); }
    CK(hipFree(g_base));
    return 0;
}
