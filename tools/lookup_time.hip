// Experiment harness (not product code): per-launch time of the PRODUCT lookup kernel (HIP events over many launches).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include <hip/hip_runtime.h>
#include "../raft_amd/csrc/lookup.hip"
namespace raftcorr {
int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return code; }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return -3; } return 0; }
}
using namespace raftcorr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv)
{
    const int H = argc > 1 ? atoi(argv[1]) : 55, W = argc > 2 ? atoi(argv[2]) : 128, B = argc > 3 ? atoi(argv[3]) : 1;
    const int N = H * W, L = 4, NC = 32, r = 4;
    LevelSet lv;
    for (int l = 0; l < RAFTCORR_MAX_LEVELS; ++l) { lv.ptr[l] = nullptr; lv.g[l] = LevelGeom{0, 0, 0, 0, 0, 4, 0}; }
    for (int l = 0; l < L; ++l) {
        lv.g[l] = level_geom(H, W, l);
        size_t n = (size_t)B * N * lv.g[l].slab;
        CK(hipMalloc(&lv.ptr[l], n * 4));
        CK(hipMemset(lv.ptr[l], 0, n * 4));
    }
    std::vector<float*> dco(NC);
    std::vector<float> drift(2 * (size_t)B * N, 0.f);
    srand(1);
    auto gauss = []() { float u = (rand() + 1.f) / ((float)RAND_MAX + 2.f), v = (rand() + 1.f) / ((float)RAND_MAX + 2.f); return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v); };
    for (int k = 0; k < NC; ++k) {
        std::vector<float> c(2 * (size_t)B * N);
        for (int b = 0; b < B; ++b)
            for (int q = 0; q < N; ++q) { c[(size_t)b * 2 * N + q] = q % W + drift[(size_t)b * 2 * N + q]; c[(size_t)b * 2 * N + N + q] = q / W + drift[(size_t)b * 2 * N + N + q]; }
        CK(hipMalloc(&dco[k], c.size() * 4)); CK(hipMemcpy(dco[k], c.data(), c.size() * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < drift.size(); ++i) drift[i] += 3.f * gauss();
    }
    float* out; CK(hipMalloc(&out, (size_t)B * L * 81 * N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 1; mode >= 0; --mode)
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < NC; ++i) launch_lookup_fwd(lv, B, H, W, L, r, dco[i], mode, out, 0);
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < 10; ++it)
                for (int i = 0; i < NC; ++i) launch_lookup_fwd(lv, B, H, W, L, r, dco[i], mode, out, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%dx%d B=%d mode %d: %.2f us per lookup\n", H, W, B, mode, ms * 1e3 / (10 * NC));
        }
    return 0;
}
