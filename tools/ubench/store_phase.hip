// Microbenchmark: the skin stage's store stream (C3 shape: 3 arrays, 235 tiles x 10,000 instances, blocks of 256 vertices x 80
// instances, non-temporal dwordx3 per lane, k_skin2's block -> (XCD, tile, range) mapping) with NO arithmetic, and the
// question the library's ablations raised: does it matter whether neighbouring tiles walk a range's instances in phase?
//   rot = 0: every tile stores instance i0, i0+1, ... in the same order; rot = r: tile t starts at (t * r) % 80
//   waves = 1: the block's four waves start a quarter of the order apart as well
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float float3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void store3(float* base, unsigned byte_off, float x, float y, float z) {
    float3v d = {x, y, z};
    asm volatile("global_store_dwordx3 %0, %1, %2 nt\n\ts_nop 1" : : "v"(byte_off), "v"(d), "s"(base) : "memory");
}
__global__ __launch_bounds__(256) void k(float* pos, float* nrm, float* tan, unsigned V, unsigned T, unsigned I, unsigned R,
                                         unsigned ranges, unsigned rot, unsigned waves, unsigned fmas) {
    const unsigned tpx = (T + 7) / 8, xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
    const unsigned tile = (kb % tpx) * 8 + xcd, rng = kb / tpx;
    if (tile >= T || rng >= ranges) return;
    const unsigned v = tile * 256 + threadIdx.x, i0 = rng * R, n = min(I, i0 + R) - i0;
    const unsigned start = __builtin_amdgcn_readfirstlane((tile * rot + (waves ? (threadIdx.x >> 6) * (n / 4) : 0u)) % n);
    float x = (float)threadIdx.x, y = x + 1.f, z = x + 2.f;
    const unsigned olb = v * 12;
    for (unsigned q = 0; q < n; ++q) {
        unsigned i = start + q; if (i >= n) i -= n;
        for (unsigned f = 0; f < fmas; ++f) { x = __builtin_fmaf(x, 1.0001f, y); y = __builtin_fmaf(y, 0.9999f, z); z = __builtin_fmaf(z, 1.0002f, x); }
        const size_t o = (size_t)(i0 + i) * V * 3;
        if (v < V) { store3(pos + o, olb, x, y, z); store3(nrm + o, olb, y, z, x); store3(tan + o, olb, z, x, y); }
    }
}
int main() {
    const unsigned V = 60000, T = (V + 255) / 256, I = 10000, R = 80, ranges = (I + R - 1) / R;
    const size_t n = (size_t)I * V * 3;
    float *pos, *nrm, *tan;
    CK(hipMalloc(&pos, n * 4)); CK(hipMalloc(&nrm, n * 4)); CK(hipMalloc(&tan, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = 8 * ((T + 7) / 8) * ranges;
    for (unsigned fmas : {0u, 50u})
        for (unsigned waves : {0u, 1u})
            for (unsigned rot : {0u, 1u, 7u, 23u, 40u}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(a));
                    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, pos, nrm, tan, V, T, I, R, ranges, rot, waves, fmas);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (rep) best = std::min(best, ms);
                }
                printf("fmas %3u  waves %s  rot %2u : %7.3f ms  %6.2f TB/s\n", fmas * 3, waves ? "apart   " : "together", rot, best, n * 12.0 / best / 1e9);
            }
    return 0;
}
