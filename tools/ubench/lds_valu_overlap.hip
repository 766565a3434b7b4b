// Microbenchmark: do LDS gathers (ds_read_b128) and fp32 VALU work of DIFFERENT waves overlap on a gfx950 CU, or do
// their times add?  The skin kernels' inner loop is "6 x ds_read_b128 of a palette entry, then 21 FMAs on it" per
// influence; their measured time is close to (LDS-array time) + (VALU time), not to the larger of the two.
//   mode 0: LDS only   — R rounds of 6 x ds_read_b128 (per-lane entry, 96-byte stride like the palette), waited, not used
//   mode 1: VALU only  — R rounds of F v_fma_f32 on 12 + 9 accumulators
//   mode 2: both       — each round's FMAs consume the round's loads (the real dependency)
//   mode 3: both, software-pipelined — round r's FMAs run on round r-1's registers while round r's loads are in flight
// Blocks of 256 threads, W blocks per CU (W waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load6(f4 (&r)[6], uint32_t addr) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
                 "ds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]) : "v"(addr));
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int F>  // F = FMAs per round in units of 21
__device__ __forceinline__ void fmas(float (&S)[21], const f4 (&r)[6], float w) {
    const float m[21] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w, r[2].x, r[2].y, r[2].z, r[2].w,
                         r[3].x, r[3].y, r[3].z, r[4].x, r[4].y, r[4].z, r[5].x, r[5].y, r[5].z};
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int x = 0; x < 21; ++x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(S[x]) : "v"(w), "v"(m[x]));
}

template <int MODE, int F>
__global__ __launch_bounds__(256) void k(float* out, int rounds, float seed) {
    extern __shared__ f4 lds[];
    for (uint32_t c = threadIdx.x; c < 128 * 6; c += 256) lds[c] = f4{seed, seed * 0.5f, seed * 0.25f, (float)c};
    __syncthreads();
    // lane -> palette entry: a few distinct joints per 16-lane group, like a real tile (conflict-free: entries 0-7)
    uint32_t addr = ((threadIdx.x * 5u + (threadIdx.x >> 4)) & 7u) * 96u;
    float S[21];
#pragma unroll
    for (int x = 0; x < 21; ++x) S[x] = seed * x;
    const float w = seed + threadIdx.x * 1e-3f;
    f4 r[6], q[6];
#pragma unroll
    for (int x = 0; x < 6; ++x) { r[x] = f4{seed, seed, seed, seed}; q[x] = r[x]; }
    for (int it = 0; it < rounds; ++it) {
        if (MODE == 0) {
            load6(r, addr);
            wait_lds();
#pragma unroll
            for (int x = 0; x < 6; ++x) asm volatile("" ::"v"(r[x]));
        } else if (MODE == 1) {
            fmas<F>(S, r, w);
        } else if (MODE == 2) {
            load6(r, addr);
            wait_lds();
            fmas<F>(S, r, w);
        } else {
            load6(r, addr);  // two rounds per trip, the register sets swap roles: no copies
            fmas<F>(S, q, w);
            wait_lds();
            load6(q, addr ^ (96u * 8u));
            fmas<F>(S, r, w);
            wait_lds();
            ++it;
        }
        addr ^= 96u * 8u;  // alternate between two sets of entries
    }
    float s = 0;
#pragma unroll
    for (int x = 0; x < 21; ++x) s += S[x];
    out[blockIdx.x * 256 + threadIdx.x] = s + r[0].x;
}

template <int MODE, int F>
double run(float* out, int W) {
    const int rounds = 20000, grid = 256 * W;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // LDS per block sized so that exactly W blocks fit a CU (160 KiB)
    const size_t lds = W == 1 ? 98304 : (160 * 1024 / W) & ~1023u;
    CK(hipFuncSetAttribute((const void*)k<MODE, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k<MODE, F>), dim3(grid), dim3(256), lds, 0, out, 100, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, F>), dim3(grid), dim3(256), lds, 0, out, rounds, 1.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e6 / rounds;  // ns per round of the whole CU (W blocks = 4 W waves do one round each)
}

template <int F>
void sweep(float* out, int W) {
    const double l = run<0, F>(out, W), v = run<1, F>(out, W), both = run<2, F>(out, W), pipe = run<3, F>(out, W);
    printf("W=%d waves/SIMD, %3d FMAs per 6 b128: LDS only %7.1f ns/round, VALU only %7.1f, dependent %7.1f, pipelined %7.1f   "
           "(max %7.1f, sum %7.1f)\n", W, 21 * F, l, v, both, pipe, l > v ? l : v, l + v);
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int W : {1, 2, 4, 5}) {
        sweep<1>(out, W);
        sweep<2>(out, W);
        sweep<4>(out, W);
    }
    return 0;
}
