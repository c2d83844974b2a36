// Host cost of raftcorr_lookup_fwd through the C ABI (experiment, not product code).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include "../include/raftcorr.h"
int main()
{
    const int B = 1, H = 55, W = 128, L = 4, r = 4, N = H * W;
    int32_t hl[4], wl[4]; int64_t slab[4];
    raftcorr_level_layout(H, W, L, hl, wl, slab);
    float* lv[4];
    for (int l = 0; l < L; ++l) { (void)hipMalloc(&lv[l], (size_t)N * slab[l] * 4); (void)hipMemset(lv[l], 0, (size_t)N * slab[l] * 4); }
    float *coords, *out;
    (void)hipMalloc(&coords, (size_t)2 * N * 4); (void)hipMemset(coords, 0, (size_t)2 * N * 4);
    (void)hipMalloc(&out, (size_t)L * 81 * N * 4);
    for (int i = 0; i < 200; ++i) raftcorr_lookup_fwd(lv, B, H, W, L, r, coords, 1, out, nullptr);
    (void)hipDeviceSynchronize();
    const int n = 3000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) raftcorr_lookup_fwd(lv, B, H, W, L, r, coords, 1, out, nullptr);
    auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("raftcorr_lookup_fwd: host %.2f us per call (GPU-bound if close to %.2f us incl. drain)\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    return 0;
}
