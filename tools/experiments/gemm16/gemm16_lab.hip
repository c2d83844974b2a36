// Lab (not product code): the conversion-free fp16x2 GEMM of raft_amd/csrc/gemm16.hip on random operands, checked on
// sampled entries against a double-precision sum on the host, and timed.   usage: gemm16_lab.bin [B D H W]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../../../raft_amd/csrc/common.h"
namespace raftcorr { int fail(int c, const char* f, ...) { fprintf(stderr, "fail %d: %s\n", c, f); return c; }
int check_launch(const char* w) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", w, hipGetErrorString(e)); return 1; } return 0; } }
#include "gemm16.hip"
using namespace raftcorr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void maxabs128(const float* x, size_t n, float* out) {     // out[128] partial maxima (one block of 128)
    float m = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 128) m = fmaxf(m, fabsf(x[i]));
    out[threadIdx.x] = m;
}

int main(int argc, char** argv) {
    const int B = argc > 4 ? atoi(argv[1]) : 6, D = argc > 4 ? atoi(argv[2]) : 256, H = argc > 4 ? atoi(argv[3]) : 50, W = argc > 4 ? atoi(argv[4]) : 90;
    const int N = H * W, S0 = (int)level_geom(H, W, 0).slab, Np = (N + 31) / 32 * 32;
    printf("B %d D %d N %d S0 %d Np %d\n", B, D, N, S0, Np);
    std::vector<float> hF((size_t)B * D * S0), hG((size_t)B * N * S0), hF1((size_t)B * D * N);
    srand(1);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : hF) v = rnd();
    for (auto& v : hF1) v = rnd() * 3.f;
    for (auto& v : hG) v = rnd() * rnd() * 0.01f;
    float *dF, *dG, *dF1, *mx, *C1, *C2;
    void *pF, *pG, *pF1;
    CK(hipMalloc(&dF, hF.size() * 4)); CK(hipMalloc(&dG, hG.size() * 4)); CK(hipMalloc(&dF1, hF1.size() * 4));
    CK(hipMalloc(&pF, hF.size() * 4)); CK(hipMalloc(&pG, (size_t)B * ((N + 127) / 128 * 128) * S0 * 4)); CK(hipMalloc(&pF1, (size_t)B * D * Np * 4));
    CK(hipMalloc(&mx, 3 * 128 * 4)); CK(hipMalloc(&C1, (size_t)B * D * N * 4)); CK(hipMalloc(&C2, (size_t)B * D * S0 * 4));
    CK(hipMemcpy(dF, hF.data(), hF.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dG, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dF1, hF1.data(), hF1.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(maxabs128, dim3(1), dim3(128), 0, 0, dF, hF.size(), mx);
    hipLaunchKernelGGL(maxabs128, dim3(1), dim3(128), 0, 0, dG, hG.size(), mx + 128);
    hipLaunchKernelGGL(maxabs128, dim3(1), dim3(128), 0, 0, dF1, hF1.size(), mx + 256);
    // operands as pk16 rows (the gradient volume too: in the library corr_splat_fold_kernel writes it that way)
    if (launch_pk16_rows(dF, (size_t)B * D, S0, S0, mx, pF, 0)) return 1;
    if (launch_pk16_rows(dG, (size_t)B * N, S0, S0, mx + 128, pG, 0, getenv("LAB_TILEMAJOR") ? N : 0)) return 1;
    if (launch_pk16_rows(dF1, (size_t)B * D, N, Np, mx + 256, pF1, 0)) return 1;
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const bool tm = getenv("LAB_TILEMAJOR") != nullptr;
    const int QT = (N + 127) / 128;
    const bool smallb = getenv("LAB_SMALLB") != nullptr;     // every workgroup reads the first rows of the volume: L2 hits
    const size_t sBb = smallb ? 0 : tm ? (size_t)QT * 128 * S0 * 4 : (size_t)N * S0 * 4;
    const int rB = smallb ? 32 : N;
    auto g1 = [&]() { return launch_gemm_pk16(pF, pG, C1, D, N, S0, 0, S0, S0, rB, N, (size_t)D * S0 * 4, sBb, (size_t)D * N, B, mx, mx + 128, 0, tm); };
    auto g2 = [&]() { return launch_gemm_pk16(pF1, pG, C2, D, S0, N, 1, Np, S0, rB, S0, (size_t)D * Np * 4, sBb, (size_t)D * S0, B, mx + 256, mx + 128, 0, tm); };
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 3; ++i) if (which ? g2() : g1()) return 1;
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        const int reps = 10;
        for (int i = 0; i < reps; ++i) which ? g2() : g1();
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * B * D * (double)N * S0;
        printf("gemm%d: %.1f us  (%.0f algorithmic TF/s, executed %.3f of 2.5 PF)\n", which + 1, ms * 1e3 / reps, fl / (ms / reps * 1e-3) / 1e12,
               3 * fl / (ms / reps * 1e-3) / 2.5e15);
    }
    std::vector<float> h1((size_t)B * D * N), h2((size_t)B * D * S0);
    CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
    double worst1 = 0, worst2 = 0, big1 = 0, big2 = 0;
    for (int it = 0; it < 4000; ++it) {
        const int b = rand() % B, d = rand() % D, q = it < 50 ? N - 1 - it : rand() % N, pcol = it < 50 ? S0 - 1 - it : rand() % S0;
        double s1 = 0, s2 = 0;
        for (int p = 0; p < S0; ++p) s1 += (double)hF[((size_t)b * D + d) * S0 + p] * hG[((size_t)b * N + q) * S0 + p];
        for (int k = 0; k < N; ++k) s2 += (double)hF1[((size_t)b * D + d) * N + k] * hG[((size_t)b * N + k) * S0 + pcol];
        worst1 = fmax(worst1, fabs(s1 - h1[((size_t)b * D + d) * N + q])); big1 = fmax(big1, fabs(s1));
        worst2 = fmax(worst2, fabs(s2 - h2[((size_t)b * D + d) * S0 + pcol])); big2 = fmax(big2, fabs(s2));
    }
    printf("gemm1: max err %.3g of max %.3g (%.2g)\n", worst1, big1, worst1 / big1);
    printf("gemm2: max err %.3g of max %.3g (%.2g)\n", worst2, big2, worst2 / big2);
    return (worst1 / big1 < 1e-5 && worst2 / big2 < 1e-5) ? 0 : 2;
}
