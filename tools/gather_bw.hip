// Microbenchmark (not product code): what bandwidth does the lookup's ACCESS PATTERN allow?
// Reads whole 128-byte lines at pseudo-random positions of a volume-sized buffer (one line per
// 8 lanes, float4 per lane -- the lookup's 2x2-block loads) in runs of `run` consecutive lines,
// and streams out 9.1 MB like the lookup does.  run = 1: isolated lines (today's layout: every
// line of a window belongs to one query's slab); run = 32: 4 KiB runs (what a query-tile-blocked
// layout would give 32 neighbouring queries with coherent flow).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256)
void gather_kernel(const float4* __restrict__ vol, size_t lines, int run, int lines_per_thread_group, float* __restrict__ out,
                   size_t out_floats, unsigned seed)
{
    const size_t grp = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 3;      // 8 lanes share a line
    const int sub = threadIdx.x & 7;
    float acc = 0.f;
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < lines_per_thread_group) {
            const size_t id = grp * lines_per_thread_group + i;            // global line request index
            const size_t r = id / run, within = id - r * run;
            size_t h = (r + seed) * 0x9E3779B97F4A7C15ull;                 // hash -> start of the run
            h ^= h >> 29;
            const size_t line = ((h % (lines / run)) * run + within);
            v[i] = vol[line * 8 + sub];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < lines_per_thread_group) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    for (size_t o = tid; o < out_floats; o += nth) __builtin_nontemporal_store(acc, &out[o]);
}

int main()
{
    const size_t vol_bytes = 268ull << 20, lines = vol_bytes / 128;
    const size_t read_lines = 147000;                    // 18.8 MB, what one Sintel lookup fetches
    const size_t out_floats = 7040ull * 324;             // 9.1 MB
    float4* vol; float* out;
    CK(hipMalloc(&vol, vol_bytes)); CK(hipMalloc(&out, out_floats * 4));
    CK(hipMemset(vol, 0, vol_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int scale : {1, 4, 16}) {                       // 1x = one lookup's traffic; larger = amortised launch
        for (int run : {1, 2, 4, 8, 32}) {
            const int lpg = 5;                           // lines per 8-lane group (the lookup issues ~5 float4 per lane)
            const size_t groups = read_lines * scale / lpg, threads = groups * 8;
            const unsigned blocks = (unsigned)((threads + 255) / 256);
            auto launch = [&](unsigned seed) {
                hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, 0, vol, lines, run, lpg, out,
                                   out_floats * scale > (64ull << 20) / 4 ? out_floats : out_floats, seed);
            };
            for (int i = 0; i < 5; ++i) launch(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int iters = 50;
            for (int i = 0; i < iters; ++i) launch(100 + i);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            const double bytes = (double)groups * lpg * 128 + out_floats * 4.0;
            printf("traffic x%-2d run %2d lines: %7.2f us  %6.2f TB/s (reads %.1f MB + writes %.1f MB)\n", scale, run, us,
                   bytes / us * 1e-6, groups * lpg * 128 / 1e6, out_floats * 4 / 1e6);
        }
    }
    return 0;
}
