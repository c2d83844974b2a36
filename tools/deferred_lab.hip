// Experiment harness (not product code): times corr_splat_fold_kernel / corr_record_kernel at a training shape, with the
// ablations of bwd_deferred.hip's RC_LAB_DEF_* switches (tools/Makefile builds one binary per switch).
//   usage: deferred_lab.bin [B H W K]        default 6 50 90 12 (BASELINE configs[3])
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../raft_amd/csrc/bwd_deferred.hip"

namespace raftcorr {
int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return code; }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return -3; } return 0; }
}
using namespace raftcorr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename F> float timeit(F launch, int iters = 20)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main(int argc, char** argv)
{
    const int B = argc > 4 ? atoi(argv[1]) : 6, H = argc > 4 ? atoi(argv[2]) : 50, W = argc > 4 ? atoi(argv[3]) : 90;
    const int K = argc > 4 ? atoi(argv[4]) : 12, L = 4, r = 4, D = 256, N = H * W, C = L * 81;
    const size_t S0 = (size_t)level_geom(H, W, 0).slab;
    std::vector<float> h((size_t)B * N * C);
    std::vector<float*> rec(K), co(K);
    float* dout;
    CK(hipMalloc(&dout, h.size() * 4));
    for (int k = 0; k < K; ++k) {
        for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2 - 1;
        CK(hipMalloc(&rec[k], h.size() * 4));
        CK(hipMemcpy(rec[k], h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> c((size_t)B * 2 * N);
        for (int b = 0; b < B; ++b)
            for (int q = 0; q < N; ++q) {
                c[((size_t)b * 2) * N + q] = (q % W) + 3.0f * k * ((rand() / (float)RAND_MAX) * 2 - 1);
                c[((size_t)b * 2 + 1) * N + q] = (q / W) + 3.0f * k * ((rand() / (float)RAND_MAX) * 2 - 1);
            }
        CK(hipMalloc(&co[k], c.size() * 4));
        CK(hipMemcpy(co[k], c.data(), c.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(dout, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    float *d0, *wgmax;
    CK(hipMalloc(&d0, (size_t)B * N * S0 * 4));
    CK(hipMalloc(&wgmax, 32768 * 4));
    unsigned nwg = 0;
    float us = timeit([&]() { launch_splat_fold(B, H, W, L, r, RAFTCORR_MODE_CANONICAL, D, K, co.data(), rec.data(), d0, 0, wgmax, &nwg, 0); });
    printf("splat_fold  B=%d %dx%d K=%d: %8.1f us  (%u workgroups, %.2f us per query and workgroup; d0 %.0f MB -> %.2f TB/s)\n", B, H, W, K, us,
           nwg, us * nwg / ((double)B * N), B * N * S0 * 4 / 1e6, B * N * S0 * 4 / us / 1e6);
    us = timeit([&]() { launch_lookup_bwd_record(dout, B, N, C, rec[0], 0); });
    printf("record      %8.1f us  (%.0f MB in + out -> %.2f TB/s)\n", us, 2.0 * h.size() * 4 / 1e6, 2.0 * h.size() * 4 / us / 1e6);
    return 0;
}
