// Experiment harness (not product code): the PRODUCT lookup kernel with in-kernel time stamps.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_stamps;
#define RC_STAMP(i) do { if ((i) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (threadIdx.x == 0) g_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#include "../raft_amd/csrc/lookup.hip"
namespace raftcorr {
int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return code; }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return -3; } return 0; }
}
using namespace raftcorr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main()
{
    const int H = 55, W = 128, N = H * W, L = 4, NC = 32, r = 4;
    LevelSet lv;
    for (int l = 0; l < RAFTCORR_MAX_LEVELS; ++l) { lv.ptr[l] = nullptr; lv.g[l] = LevelGeom{0, 0, 0, 0, 0, 4, 0}; }
    for (int l = 0; l < L; ++l) {
        lv.g[l] = level_geom(H, W, l);
        size_t n = (size_t)N * lv.g[l].slab;
        CK(hipMalloc(&lv.ptr[l], n * 4));
        std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)((unsigned)(i * 2654435761u) >> 20) * 1e-3f;
        CK(hipMemcpy(lv.ptr[l], h.data(), n * 4, hipMemcpyHostToDevice));
    }
    std::vector<float*> dco(NC);
    std::vector<float> drift(2 * N, 0.f);
    srand(1);
    auto gauss = []() { float u = (rand() + 1.f) / ((float)RAND_MAX + 2.f), v = (rand() + 1.f) / ((float)RAND_MAX + 2.f); return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v); };
    for (int k = 0; k < NC; ++k) {
        std::vector<float> c(2 * N);
        for (int q = 0; q < N; ++q) { c[q] = q % W + drift[q]; c[N + q] = q / W + drift[N + q]; }
        CK(hipMalloc(&dco[k], 2 * N * 4)); CK(hipMemcpy(dco[k], c.data(), 2 * N * 4, hipMemcpyHostToDevice));
        for (int i = 0; i < 2 * N; ++i) drift[i] += 3.f * gauss();
    }
    float* out; CK(hipMalloc(&out, (size_t)L * 81 * N * 4));
    const int nb = ((N + 31) / 32) * L;
    unsigned long long* ds; CK(hipMalloc(&ds, nb * 4 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ds, sizeof(ds)));
    for (int mode = 1; mode >= 0; --mode)
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 6; ++i) launch_lookup_fwd(lv, 1, H, W, L, r, dco[i], mode, out, 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs(nb * 4);
        CK(hipMemcpy(hs.data(), ds, nb * 4 * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t3 = 0;
        for (int i = 0; i < nb; ++i) { t0 = std::min(t0, hs[i * 4]); t3 = std::max(t3, hs[i * 4 + 3]); }
        double sm[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
        for (int i = 0; i < nb; ++i) for (int k = 0; k < 4; ++k) { double v = (hs[i * 4 + k] - t0) * 0.01; sm[k] += v; mx[k] = fmax(mx[k], v); }
        printf("mode %d: start avg %.2f max %.2f | tables avg %.2f max %.2f | taps avg %.2f max %.2f | stores avg %.2f max %.2f | span %.2f us\n", mode,
               sm[0] / nb, mx[0], sm[1] / nb, mx[1], sm[2] / nb, mx[2], sm[3] / nb, mx[3], (t3 - t0) * 0.01);
    }
    return 0;
}
