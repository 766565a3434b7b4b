// Can a gfx950 workgroup take more than 64 KiB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize)?  prints what launches.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, unsigned n) {
    extern __shared__ float s[];
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    float a = 0.0f;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) a += s[n - 1 - i];
    if (a == -1.0f) out[0] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = s[n - 1];
}
int main() {
    float* d; hipMalloc(&d, 64);
    for (unsigned kb : {48u, 64u, 72u, 80u, 96u, 128u, 160u}) {
        const unsigned bytes = kb * 1024, n = bytes / 4;
        hipError_t a = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(k, dim3(512), dim3(256), bytes, 0, d, n);
        hipError_t l = hipGetLastError();
        hipError_t s = hipDeviceSynchronize();
        float h[2] = {0, 0}; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%3u KiB: set %s, launch %s, sync %s, last element %.0f (want %u)\n", kb, hipGetErrorName(a), hipGetErrorName(l), hipGetErrorName(s), h[1], n - 1);
    }
    return 0;
}
