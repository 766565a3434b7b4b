// Microbenchmark: write rate of the skinning kernel's OUTPUT ADDRESS PATTERN, with no math.
// Block = (tile of 256 vertices, range of R instances); per instance each wave stores 3 x 768 B
// (pos/nrm/tan of 64 vertices) at offset (i*V + tile*256 + wave*64)*12 of three arrays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// order 0: block id -> tile fastest (b % T), range = b / T
// order 1: XCD-chunked (tile chunks per XCD) as libskinrt does
__global__ __launch_bounds__(256) void k_pat(float* pos, float* nrm, float* tan, unsigned V, unsigned T, unsigned I,
                                            unsigned R, unsigned ranges, int order, unsigned sleep) {
    unsigned tile, rng;
    if (order == 0) { tile = blockIdx.x % T; rng = blockIdx.x / T; }
    else {
        const unsigned Tx = (T + 7) / 8, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        tile = xcd * Tx + k % Tx; rng = k / Tx;
    }
    if (tile >= T || rng >= ranges) return;
    const unsigned v = tile * 256 + threadIdx.x;
    float x = (float)threadIdx.x, y = x + 1.f, z = x + 2.f;
    const unsigned i1 = min(I, (rng + 1) * R);
    for (unsigned i = rng * R; i < i1; ++i) {
        x += 1.f; y += 2.f; z += 3.f;
        if (sleep) __builtin_amdgcn_s_sleep(8);
        const size_t o = ((size_t)i * V + v) * 3;
        if (v < V) {
            *reinterpret_cast<float3*>(pos + o) = make_float3(x, y, z);
            *reinterpret_cast<float3*>(nrm + o) = make_float3(y, z, x);
            *reinterpret_cast<float3*>(tan + o) = make_float3(z, x, y);
        }
    }
}

int main() {
    const unsigned V = 60000, T = (V + 255) / 256, I = 10000;
    const size_t n = (size_t)I * V * 3;
    float *pos, *nrm, *tan;
    CK(hipMalloc(&pos, n * 4)); CK(hipMalloc(&nrm, n * 4)); CK(hipMalloc(&tan, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int order = 0; order < 2; ++order)
        for (unsigned R : {16u, 64u, 256u, 1024u})
            for (unsigned sleep : {0u, 1u}) {
                const unsigned ranges = (I + R - 1) / R;
                const unsigned grid = order == 0 ? T * ranges : 8 * ((T + 7) / 8) * ranges;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(a));
                    hipLaunchKernelGGL(k_pat, dim3(grid), dim3(256), 0, 0, pos, nrm, tan, V, T, I, R, ranges, order, sleep);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                }
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                printf("order %d range %4u sleep %u : %7.3f ms  %7.1f GB/s\n", order, R, sleep, ms, n * 4.0 / ms / 1e6);
            }
    return 0;
}
