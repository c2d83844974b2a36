// Experiment harness (not product code): times variants of the pyramid build kernel.
// Compiled twice (with and without -DRC_LAB_SKIP_EPILOGUE) by tools/Makefile.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../raft_amd/csrc/build_pyramid.hip"

namespace raftcorr {
int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return code; }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return -3; } return 0; }
}
using namespace raftcorr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename F> float timeit(F launch, int iters = 30)
{
    if (getenv("LAB_ITERS")) iters = atoi(getenv("LAB_ITERS"));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main(int argc, char** argv)
{
    const int B = 1, D = 256, H = argc > 2 ? atoi(argv[1]) : 55, W = argc > 2 ? atoi(argv[2]) : 128, N = H * W, L = getenv("LAB_L") ? atoi(getenv("LAB_L")) : 4;
    BuildArgs a; a.B = B; a.N = N; a.L = L;
    a.tilesM = (N + BM - 1) / BM; a.tilesN = patches_h(H) * patches_w(W); a.nblk_per_b = a.tilesM * a.tilesN;
    a.inv_sqrt_d = 1.f / 16; a.sqrt_d = 16;
    for (int l = 0; l < 4; ++l) {
        LevelGeom g = level_geom(H, W, l); a.slab[l] = g.slab;
        CK(hipMalloc(&a.lvl[l], (size_t)N * g.slab * 4));
    }
    std::vector<float> h((size_t)D * N);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2 - 1;
    float *f1, *f2, *ws;
    CK(hipMalloc(&f1, h.size() * 4)); CK(hipMalloc(&f2, h.size() * 4));
    CK(hipMemcpy(f1, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(f2, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ws, build_workspace_bytes(B, D, H, W)));
    a.PW = patches_w(W); a.group_m = getenv("LAB_GROUP_M") ? atoi(getenv("LAB_GROUP_M")) : GROUP_M; a.Dpad = D;
    dim3 grid(a.nblk_per_b), block(256);
    const double flops = 2.0 * N * (double)N * D;
#define RUN(name, ...) { float us = timeit([&]() { hipLaunchKernelGGL((__VA_ARGS__), grid, block, 0, 0, a); }); printf("%-40s %8.1f us  %6.1f TF (algorithmic)\n", name, us, flops / us * 1e-6); }
    {   // fp32 operands
        float* f1p = ws; float* f2p = ws + (size_t)B * a.tilesM * D * 128;
        a.f1p = f1p; a.f2p = f2p;
        launch_pack_operands(f1, f2, f1p, f2p, B, D, D, H, W, a.tilesM, a.tilesN, 0);
        RUN("fp32 (product instantiation)", corr_build_f32_kernel<true, BK, WPS>);
        if (getenv("LAB_OCC")) {        // occupancy sensitivity: pad the workgroup's LDS with dynamic shared memory
            for (int extra : {0, 24 * 1024, 37 * 1024, 64 * 1024, 140 * 1024}) {
                float us = timeit([&]() { hipLaunchKernelGGL((corr_build_f32_kernel<true, BK, WPS>), grid, block, extra, 0, a); });
                printf("fp32 product kernel + %3d KB dynamic LDS (%d workgroups/CU by LDS): %8.1f us\n", extra / 1024,
                       160 * 1024 / (16 * 1024 + extra), us);
            }
        }
        if (getenv("LAB_SWEEP")) {
            RUN("fp32 BK=16, 2 waves/SIMD", corr_build_f32_kernel<true, 16, 2>);
            RUN("fp32 BK=16, 4 waves/SIMD", corr_build_f32_kernel<true, 16, 4>);
            RUN("fp32 BK=32, 2 waves/SIMD", corr_build_f32_kernel<true, 32, 2>);
            RUN("fp32 BK=8, 4 waves/SIMD", corr_build_f32_kernel<true, 8, 4>);
            RUN("fp32 BK=8, 3 waves/SIMD", corr_build_f32_kernel<true, 8, 3>);
            RUN("fp32 BK=8, 5 waves/SIMD", corr_build_f32_kernel<true, 8, 5>);
            RUN("fp32 BK=8, 6 waves/SIMD", corr_build_f32_kernel<true, 8, 6>);
        }
        printf("pack fp32: %.1f us\n", timeit([&]() { launch_pack_operands(f1, f2, f1p, f2p, B, D, D, H, W, a.tilesM, a.tilesN, 0); }));
    }
    PackArgs pk{f1, f2, ws, nullptr, D, D, H, W, a.tilesM, a.tilesN, nullptr};
    a.maxpart = nullptr;
    dim3 pgrid(a.tilesM + a.tilesN, PACK_KSPLIT, B);
    {   // bf16 operands
        pk.f2p = ws + (size_t)B * a.tilesM * D * 64; a.f1p = pk.f1p; a.f2p = pk.f2p;
        hipLaunchKernelGGL((corr_pack_bf16_kernel<1, false>), pgrid, block, 0, 0, pk);
        RUN("bf16 (1 piece)", corr_build_bf16_kernel<true, 1, false>);
        RUN("bf16 (1 piece) -> bf16 volume", corr_build_bf16_kernel<true, 1, false, 2>);
        printf("pack bf16: %.1f us\n", timeit([&]() { hipLaunchKernelGGL((corr_pack_bf16_kernel<1, false>), pgrid, block, 0, 0, pk); }));
    }
    {   // 3 x bf16 operands
        pk.f2p = ws + (size_t)B * a.tilesM * D * 64 * 3; a.f1p = pk.f1p; a.f2p = pk.f2p;
        hipLaunchKernelGGL((corr_pack_bf16_kernel<3, false>), pgrid, block, 0, 0, pk);
        RUN("bf16x3 (3 pieces, 6 products)", corr_build_bf16_kernel<true, 3, false>);
        printf("pack bf16x3: %.1f us\n", timeit([&]() { hipLaunchKernelGGL((corr_pack_bf16_kernel<3, false>), pgrid, block, 0, 0, pk); }));
    }
    {   // 2 x fp16 operands (range scaling skipped: inputs are in [-1, 1]; maxpart = null -> unscale 1)
        pk.f2p = ws + (size_t)B * a.tilesM * D * 64 * 2; a.f1p = pk.f1p; a.f2p = pk.f2p;
        float* mp; CK(hipMalloc(&mp, 2 * MAXPARTS * 4));
        hipLaunchKernelGGL(corr_maxabs_kernel, dim3(MAXPARTS, 2), block, 0, 0, f1, f2, (size_t)D * N, mp, (const float*)nullptr, (size_t)0);
        pk.maxpart = mp; a.maxpart = mp;
        hipLaunchKernelGGL((corr_pack_bf16_kernel<2, true>), pgrid, block, 0, 0, pk);
        RUN("fp16x2 (2 pieces, 3 products)", corr_build_bf16_kernel<true, 2, true>);
        RUN("fp16x2 -> fp16 volume", corr_build_bf16_kernel<true, 2, true, 1>);
        printf("pack fp16x2: %.1f us, max|x| pass: %.1f us\n",
               timeit([&]() { hipLaunchKernelGGL((corr_pack_bf16_kernel<2, true>), pgrid, block, 0, 0, pk); }),
               timeit([&]() { hipLaunchKernelGGL(corr_maxabs_kernel, dim3(MAXPARTS, 2), block, 0, 0, f1, f2, (size_t)D * N, mp, (const float*)nullptr, (size_t)0); }));
    }
    return 0;
}
