// Calibration of rocprofv3 FETCH_SIZE / TCC_EA0_RDREQ on gfx950 for sparse line accesses
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").
// Each kernel touches a KNOWN number of distinct 128-B lines of a 512 MiB buffer (larger than
// L2 + Infinity Cache, so every line comes from HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

// one dword per STRIDE bytes
template <int STRIDE>
__global__ void sparse_read(const float* __restrict__ src, float* __restrict__ dst, size_t nlines)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) acc += src[i * (STRIDE / 4)];
    if (acc == 123.25f) dst[0] = acc;
}
__global__ void stream_read(const float4* __restrict__ src, float* __restrict__ dst, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.25f) dst[0] = acc;
}
int main()
{
    const size_t bytes = 512ull << 20;
    float *buf, *dst;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&dst, 256));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(sparse_read<128>, dim3(4096), dim3(256), 0, 0, buf, dst, bytes / 128);   // 4 Mi lines, 512 MiB of lines
    hipLaunchKernelGGL(sparse_read<256>, dim3(4096), dim3(256), 0, 0, buf, dst, bytes / 256);   // 2 Mi lines, 256 MiB of lines
    hipLaunchKernelGGL(sparse_read<64>, dim3(4096), dim3(256), 0, 0, buf, dst, bytes / 64);     // every 64-B half line
    hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, 0, (const float4*)buf, dst, bytes / 16);
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
