// Microbenchmark: does a ds_read_b128 (or a v_fma_f32) cost less when part of the wave's lanes are switched off?
// (Would per-lane skipping of zero-weight influences — lanes sorted by influence count inside a wave, so the active
// lanes are a prefix — save LDS or VALU time?)  6 x ds_read_b128 per round as in lds_valu_overlap.hip, under an EXEC
// mask: a prefix of `nact` lanes, or every other lane.  5 blocks of 256 threads per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: LDS reads, 1: FMAs
__global__ __launch_bounds__(256) void k(float* out, int rounds, float seed, unsigned nact, unsigned pattern) {
    extern __shared__ f4 lds[];
    for (uint32_t c = threadIdx.x; c < 128 * 6; c += 256) lds[c] = f4{seed, seed * 0.5f, seed * 0.25f, (float)c};
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    uint32_t addr = ((threadIdx.x * 5u + (threadIdx.x >> 4)) & 7u) * 96u;
    f4 r[6];
    for (int x = 0; x < 6; ++x) r[x] = f4{seed, seed, seed, seed};
    float S[21];
    for (int x = 0; x < 21; ++x) S[x] = seed * x;
    const bool on = pattern == 0 ? lane < nact : (pattern == 1 ? (lane & 1u) == 0 : (lane & 15u) < nact / 4);
    if (on) {
        for (int it = 0; it < rounds; ++it) {
            if (MODE == 0) {
                asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
                             "ds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]) : "v"(addr) : "memory");
            } else {
#pragma unroll
                for (int x = 0; x < 21; ++x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(S[x]) : "v"(seed), "v"(r[x % 6].x));
            }
            addr ^= 96u * 8u;
        }
    }
    float s = r[0].x + r[5].w;
    for (int x = 0; x < 21; ++x) s += S[x];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
double run(float* out, unsigned nact, unsigned pattern) {
    const int rounds = 20000, W = 5, grid = 256 * W;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t lds = (160 * 1024 / W) & ~1023u;
    CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), lds, 0, out, 100, 1.0f, nact, pattern);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), lds, 0, out, rounds, 1.0f, nact, pattern);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e6 / rounds;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (unsigned nact : {64u, 48u, 32u, 16u, 8u})
        printf("prefix of %2u lanes            : 6 x ds_read_b128 %7.1f ns per CU-round, 21 x v_fma_f32 %7.1f\n", nact, run<0>(out, nact, 0), run<1>(out, nact, 0));
    printf("every other lane (32 lanes)   : 6 x ds_read_b128 %7.1f ns per CU-round, 21 x v_fma_f32 %7.1f\n", run<0>(out, 32, 1), run<1>(out, 32, 1));
    for (unsigned nact : {32u, 16u})
        printf("%2u lanes, first of every row  : 6 x ds_read_b128 %7.1f ns per CU-round, 21 x v_fma_f32 %7.1f\n", nact, run<0>(out, nact, 2), run<1>(out, nact, 2));
    return 0;
}
