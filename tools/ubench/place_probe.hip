// Where in HBM is a write stream fast?  Allocates the device's memory in 8 GiB chunks and times, per
// chunk, (a) a linear float4 fill and (b) the skin stage's output pattern on ONE array inside the chunk
// (10,000 instances x 60,000 vertices x 12 B = 7.2 GB; block = tile x 80 instances, XCD-chunked).
// Then the three-array pattern on triples of chunks.  Purpose: the skin stage's time depends on where
// hipMalloc puts its output arrays (DESIGN.md "output placement"); this looks for the rule.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_fill(f4* p, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p[i] = f4{1.f, 2.f, 3.f, 4.f};
}

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

static float time_it(void (*fn)(void*), void* arg, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0)); fn(arg); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) best = std::min(best, ms);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

struct PatArg { char* a[3]; int arrays; };
static const unsigned V = 60000, I = 10000, R = 80;
static void launch_pat(void* p) {
    PatArg* q = (PatArg*)p;
    const unsigned T = (V + 255) / 256, ranges = (I + R - 1) / R, grid = 8 * ((T + 7) / 8) * ranges;
    if (q->arrays == 1) hipLaunchKernelGGL(k_pat<1>, dim3(grid), dim3(256), 0, 0, q->a[0], q->a[1], q->a[2], V, T, I, R, ranges);
    else hipLaunchKernelGGL(k_pat<3>, dim3(grid), dim3(256), 0, 0, q->a[0], q->a[1], q->a[2], V, T, I, R, ranges);
}
struct FillArg { f4* p; size_t n4; };
static void launch_fill(void* p) { FillArg* q = (FillArg*)p; hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, 0, q->p, q->n4); }

int main(int argc, char** argv) {
    const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 8) << 30;
    const int order_desc = argc > 2 ? atoi(argv[2]) : 0;
    size_t free_b, total_b; CK(hipMemGetInfo(&free_b, &total_b));
    printf("# free %.1f GiB of %.1f GiB, chunk %.0f GiB\n", free_b / 1073741824.0, total_b / 1073741824.0, chunk / 1073741824.0);
    std::vector<char*> chunks;
    while (true) {
        CK(hipMemGetInfo(&free_b, &total_b));
        if (free_b < chunk + ((size_t)6 << 30)) break;
        char* p = nullptr;
        if (hipMalloc((void**)&p, chunk) != hipSuccess) { (void)hipGetLastError(); break; }
        chunks.push_back(p);
    }
    printf("# %zu chunks\n", chunks.size());
    const size_t arr = (size_t)I * V * 12;
    std::vector<float> fill_ms(chunks.size()), pat_ms(chunks.size());
    for (size_t c = 0; c < chunks.size(); ++c) {
        FillArg fa{(f4*)chunks[c], arr / 16};
        fill_ms[c] = time_it(launch_fill, &fa, 3);
        PatArg pa{{chunks[c], chunks[c], chunks[c]}, 1};
        pat_ms[c] = time_it(launch_pat, &pa, 3);
        printf("chunk %2zu at %p : fill %6.3f ms %7.1f GB/s | pattern(1 array) %6.3f ms %7.1f GB/s\n", c, (void*)chunks[c], fill_ms[c],
               arr / fill_ms[c] / 1e6, pat_ms[c], arr / pat_ms[c] / 1e6);
        fflush(stdout);
    }
    // triples: consecutive chunks, then the three fastest and the three slowest single-array chunks
    std::vector<size_t> idx(chunks.size());
    for (size_t c = 0; c < idx.size(); ++c) idx[c] = c;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return pat_ms[a] < pat_ms[b]; });
    auto triple = [&](size_t a, size_t b, size_t c, const char* what) {
        PatArg pa{{chunks[a], chunks[b], chunks[c]}, 3};
        const float ms = time_it(launch_pat, &pa, 3);
        printf("triple (%2zu,%2zu,%2zu) %-10s : %6.3f ms %7.1f GB/s\n", a, b, c, what, ms, 3.0 * arr / ms / 1e6);
        fflush(stdout);
    };
    for (size_t c = 0; c + 2 < chunks.size(); c += (order_desc ? 1 : 3)) triple(c, c + 1, c + 2, "consecutive");
    if (chunks.size() >= 6) {
        triple(idx[0], idx[1], idx[2], "fastest");
        triple(idx[idx.size() - 1], idx[idx.size() - 2], idx[idx.size() - 3], "slowest");
        triple(idx[0], idx[idx.size() / 2], idx[idx.size() - 1], "mixed");
    }
    // the three arrays inside ONE chunk are not possible at 8 GiB; same chunk for all three = same bytes rewritten
    triple(0, 0, 0, "same chunk");
    for (char* p : chunks) CK(hipFree(p));
    return 0;
}
