// Host cost of hipLaunchKernelGGL as a function of the kernel-argument size (experiment, not product code).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Args { int v[N]; };
template <int N> __global__ void k(Args<N> a, int* out) { if (a.v[0] == 12345 && threadIdx.x == 0) out[0] = a.v[N - 1]; }
template <int N> void run(int* out)
{
    Args<N> a{}; 
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<N>, dim3(880), dim3(256), 0, 0, a, out);
    hipDeviceSynchronize();
    const int n = 20000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<N>, dim3(880), dim3(256), 0, 0, a, out);
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("%4d-byte args: host %.2f us per launch, incl. drain %.2f us\n", N * 4,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
}
int main()
{
    int* out; hipMalloc(&out, 4);
    run<4>(out); run<16>(out); run<64>(out); run<128>(out); run<208>(out); run<256>(out);
    return 0;
}
