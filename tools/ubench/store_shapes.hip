// Microbenchmark: HBM write rate of the store shapes the skinning epilogue could use.
// Each wave writes consecutive 768-byte chunks (64 vertices x 3 floats) of a big buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// mode 0: dwordx3 per lane (lane stride 12 B)
// mode 1: 48 lanes x dwordx4 (768 B contiguous), data via LDS transpose
// mode 2: 64 lanes x dwordx4 (1024 B contiguous)  [reference]
// mode 3: 3 x dword per lane at stride 12 B
template <int MODE>
__global__ __launch_bounds__(256) void k_store(float* out, size_t chunks_per_wave, int reps) {
    __shared__ float lds[4][192];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gw = (size_t)blockIdx.x * 4 + wave;
    const size_t stride_bytes = MODE == 2 ? 1024 : 768;
    char* base = (char*)out + gw * chunks_per_wave * stride_bytes;
    float x = (float)lane, y = x + 1.f, z = x + 2.f;
    for (int r = 0; r < reps; ++r)
    for (size_t c = 0; c < chunks_per_wave; ++c) {
        char* p = base + c * stride_bytes;
        x += 1.f; y += 1.f; z += 1.f;
        if (MODE == 0) {
            *reinterpret_cast<float3*>(p + lane * 12) = make_float3(x, y, z);
        } else if (MODE == 1) {
            lds[wave][lane * 3 + 0] = x; lds[wave][lane * 3 + 1] = y; lds[wave][lane * 3 + 2] = z;
            __builtin_amdgcn_wave_barrier();
            if (lane < 48) *reinterpret_cast<float4*>(p + lane * 16) = *reinterpret_cast<float4*>(&lds[wave][lane * 4]);
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == 2) {
            *reinterpret_cast<float4*>(p + lane * 16) = make_float4(x, y, z, x);
        } else {
            float* q = reinterpret_cast<float*>(p + lane * 12);
            q[0] = x; q[1] = y; q[2] = z;
        }
    }
}

template <int MODE>
void run(float* buf, size_t bytes, const char* name) {
    const int blocks = 256 * 8;
    const size_t stride = MODE == 2 ? 1024 : 768;
    const size_t chunks = bytes / ((size_t)blocks * 4 * stride);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_store<MODE>, dim3(blocks), dim3(256), 0, 0, buf, chunks, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_store<MODE>, dim3(blocks), dim3(256), 0, 0, buf, chunks, 1);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double written = (double)chunks * blocks * 4 * stride;
    printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, written / ms / 1e6);
}

int main() {
    const size_t bytes = 8ull << 30;
    float* buf; CK(hipMalloc(&buf, bytes));
    run<2>(buf, bytes, "64 lanes x dwordx4 (1 KiB/instr)");
    run<0>(buf, bytes, "dwordx3 per lane (768 B/instr)");
    run<1>(buf, bytes, "LDS transpose -> 48 lanes x dwordx4");
    run<3>(buf, bytes, "3 x dword per lane, stride 12 B");
    run<2>(buf, bytes, "64 lanes x dwordx4 (again)");
    return 0;
}
