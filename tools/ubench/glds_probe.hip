// Probe: semantics of __builtin_amdgcn_global_load_lds (16-byte form) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k(const float* src, float* out, float* junk) {
    __shared__ float lds[4][3][256];   // per wave: 3 pieces of 1 KiB
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* g = src + wave * 576 + lane * 4;         // wave's share: 576 floats
    // a few stores first, to see whether the waits below have to cover them
    junk[threadIdx.x] = 1.0f; junk[256 + threadIdx.x] = 2.0f; junk[512 + threadIdx.x] = 3.0f;
    __builtin_amdgcn_global_load_lds(g, &lds[wave][0][0], 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g + 256, &lds[wave][1][0], 16, 0, 0);
    if (lane < 16) __builtin_amdgcn_global_load_lds(g + 512, &lds[wave][2][0], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int k = 0; k < 9; ++k) out[wave * 576 + k * 64 + lane] = lds[wave][0][k * 64 + lane];
}
int main() {
    std::vector<float> h(4 * 576 + 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *s, *o, *j; CK(hipMalloc(&s, h.size() * 4)); CK(hipMalloc(&o, 4 * 576 * 4)); CK(hipMalloc(&j, 4096));
    CK(hipMemcpy(s, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, o, j);
    CK(hipDeviceSynchronize());
    std::vector<float> r(4 * 576);
    CK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t i = 0; i < r.size(); ++i) if (r[i] != h[i]) { if (bad < 5) printf("mismatch at %zu: %f vs %f\n", i, r[i], h[i]); ++bad; }
    printf("glds probe: %d mismatches of %zu\n", bad, r.size());
    return bad != 0;
}
