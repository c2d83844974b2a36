// GPU-side duration of an (almost) empty kernel by grid shape (experiment, not product code): run under
// rocprofv3 --kernel-trace --stats and read the per-kernel averages.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T> __global__ __launch_bounds__(T) void empty_k(int* out) { if (out == nullptr && threadIdx.x == 0) out[0] = T; }
template <int T> void run(int* out, int wgs)
{
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_k<T>, dim3(wgs), dim3(T), 0, 0, out);
    (void)hipDeviceSynchronize();
}
int main()
{
    int* out; (void)hipMalloc(&out, 4);
    run<1024>(out, 220); run<512>(out, 440); run<256>(out, 880); run<128>(out, 1760); run<64>(out, 3520);
    run<256>(out, 256); run<256>(out, 1760);
    return 0;
}
