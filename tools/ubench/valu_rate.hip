// Microbenchmark: issue cost of scalar vs packed fp32 VALU on gfx950, as the skinning blend uses them:
// 12 accumulators, weight operand shared, matrix operands from 12 other registers.
//   mode 0: 12 x v_fma_f32            (12 FMAs)
//   mode 1:  6 x v_pk_fma_f32         (12 FMAs), weight pair {w,w}
//   mode 2:  6 x v_pk_fma_f32 with op_sel broadcasting the weight's low half (no duplicated register)
//   mode 3: 12 x (v_mul_f32 + v_add_f32)   (the contraction-free form, 24 instructions)
//   mode 4:  6 x (v_pk_mul_f32 + v_pk_add_f32)   (12 instructions)
// Every wave runs `iters` rounds; W waves per SIMD (grid = 256 CUs x 4 SIMDs x W waves, blocks of 64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed) {
    float S[12], m[12];
    float w = seed + threadIdx.x * 1e-3f;
#pragma unroll
    for (int x = 0; x < 12; ++x) { S[x] = seed * x; m[x] = seed + x; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (MODE == 0) {
#pragma unroll
                for (int x = 0; x < 12; ++x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(S[x]) : "v"(w), "v"(m[x]));
            } else if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int x = 0; x < 12; x += 2) {
                    f2 acc = {S[x], S[x + 1]}, mm = {m[x], m[x + 1]}, ww = {w, w};
                    if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(ww), "v"(mm));
                    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(ww), "v"(mm));
                    S[x] = acc.x; S[x + 1] = acc.y;
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int x = 0; x < 12; ++x) {
                    float p;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(w), "v"(m[x]));
                    asm volatile("v_add_f32 %0, %1, %0" : "+v"(S[x]) : "v"(p));
                }
            } else {
#pragma unroll
                for (int x = 0; x < 12; x += 2) {
                    f2 acc = {S[x], S[x + 1]}, mm = {m[x], m[x + 1]}, ww = {w, w}, p;
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(ww), "v"(mm));
                    asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc) : "v"(p));
                    S[x] = acc.x; S[x + 1] = acc.y;
                }
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int x = 0; x < 12; ++x) r += S[x];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int MODE>
void run(float* out, int W, const char* name) {
    const int iters = 20000, grid = 256 * 4 * W;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 100, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double fmas_per_wave = (double)iters * 8 * 12;
    // ns per 12-FMA round per SIMD (W waves share it)
    const double ns_round = ms * 1e6 / ((double)iters * 8 * W);
    printf("%-46s W=%d : %8.3f ms, %6.2f ns per 12-FMA round per SIMD, %6.1f TFLOP/s\n", name, W, ms, ns_round,
           fmas_per_wave * 64 * 2 * grid / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 4 * 8 * 64 * 4));
    for (int W : {1, 2, 5, 8}) {
        run<0>(out, W, "12 x v_fma_f32");
        run<1>(out, W, "6 x v_pk_fma_f32 {w,w}");
        run<2>(out, W, "6 x v_pk_fma_f32 op_sel_hi:[0,1,1]");
        run<3>(out, W, "12 x (v_mul_f32, v_add_f32)");
        run<4>(out, W, "6 x (v_pk_mul_f32, v_pk_add_f32)");
    }
    return 0;
}
