// Microbenchmark: what does it cost the skinning kernel's output stream if a tile's 256 vertices are dealt to the
// lanes in another order than the mesh's (rig compiler sorting a tile's vertices by influence count so that whole
// waves can stop early)?  Same block shape and addresses as store_pattern.hip (block = tile x range of instances,
// three arrays, non-temporal dwordx3 per lane), but lane -> vertex inside the tile goes through a permutation:
//   mode 0  identity                      (a wave's store covers 768 contiguous bytes)
//   mode 1  random over the tile          (a wave's 64 x 12 B pieces are spread over the tile's 3072 bytes)
//   mode 2  interleaved: vertex = 4 * lane + wave   (every wave touches every line of the tile)
//   mode 3  random inside each 64-vertex quarter    (same lines as identity, other lane order)
//   mode 4  groups of 4 consecutive vertices dealt at random over the tile (48-byte pieces)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float float3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void store3(float* base, unsigned byte_off, float x, float y, float z) {
    float3v d = {x, y, z};
    asm volatile("global_store_dwordx3 %0, %1, %2 nt\n\ts_nop 1" : : "v"(byte_off), "v"(d), "s"(base) : "memory");
}

__global__ __launch_bounds__(256) void k_pat(float* pos, float* nrm, float* tan, const unsigned char* perm, unsigned V, unsigned T,
                                            unsigned I, unsigned R, unsigned ranges, unsigned fmas) {
    const unsigned tile = blockIdx.x % T, rng = blockIdx.x / T;
    if (tile >= T || rng >= ranges) return;
    const unsigned v = tile * 256 + perm[threadIdx.x];
    float x = (float)threadIdx.x, y = x + 1.f, z = x + 2.f;
    const unsigned i1 = min(I, (rng + 1) * R);
    const unsigned olb = v * 12;
    for (unsigned i = rng * R; i < i1; ++i) {
        for (unsigned f = 0; f < fmas; ++f) { x = __builtin_fmaf(x, 1.0001f, y); y = __builtin_fmaf(y, 0.9999f, z); z = __builtin_fmaf(z, 1.0002f, x); }
        const size_t o = (size_t)i * V * 3;
        if (v < V) {
            store3(pos + o, olb, x, y, z);
            store3(nrm + o, olb, y, z, x);
            store3(tan + o, olb, z, x, y);
        }
    }
}

int main() {
    const unsigned V = 60000, T = (V + 255) / 256, I = 10000, R = 80;
    const size_t n = (size_t)I * V * 3;
    float *pos, *nrm, *tan;
    CK(hipMalloc(&pos, n * 4)); CK(hipMalloc(&nrm, n * 4)); CK(hipMalloc(&tan, n * 4));
    unsigned char* dperm; CK(hipMalloc(&dperm, 256));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::mt19937 rng(12345);
    static const char* names[5] = {"identity", "random over the tile", "vertex = 4*lane + wave", "random inside each quarter", "groups of 4 at random"};
    for (unsigned fmas : {0u, 40u})
        for (int mode = 0; mode < 5; ++mode) {
            std::vector<unsigned char> p(256);
            for (int k = 0; k < 256; ++k) p[k] = (unsigned char)k;
            if (mode == 1) std::shuffle(p.begin(), p.end(), rng);
            if (mode == 2) for (int k = 0; k < 256; ++k) p[k] = (unsigned char)(4 * (k & 63) + (k >> 6));
            if (mode == 3) for (int q = 0; q < 4; ++q) std::shuffle(p.begin() + 64 * q, p.begin() + 64 * q + 64, rng);
            if (mode == 4) {
                std::vector<int> g(64);
                for (int k = 0; k < 64; ++k) g[k] = k;
                std::shuffle(g.begin(), g.end(), rng);
                for (int k = 0; k < 256; ++k) p[k] = (unsigned char)(4 * g[k / 4] + (k & 3));
            }
            CK(hipMemcpy(dperm, p.data(), 256, hipMemcpyHostToDevice));
            const unsigned ranges = (I + R - 1) / R;
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_pat, dim3(T * ranges), dim3(256), 0, 0, pos, nrm, tan, dperm, V, T, I, R, ranges, fmas);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep) best = std::min(best, ms);
            }
            printf("fmas %2u  %-28s : %7.3f ms  %7.1f GB/s\n", fmas * 3, names[mode], best, n * 4.0 / best / 1e6);
        }
    return 0;
}
