// Probe (not product code): semantics of ds_read_b64_tr_b16 on gfx950, as used by alt_bwd.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(s4* out) {
    __shared__ __attribute__((aligned(16))) short img[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (short)i;           // img[r][c] = r * 64 + c
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const int row0 = 8 * g, col0 = 16 * g;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(img + (row0 + qq) * 64 + col0 + 4 * pp));
    out[lane] = v;
}
int main() {
    s4* d; hipMalloc(&d, 64 * sizeof(s4));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15, row0 = 8 * g, col0 = 16 * g;
        for (int e = 0; e < 4; ++e) { const int want = (row0 + e) * 64 + col0 + i; if (h[lane * 4 + e] != want) { if (bad < 8) printf("lane %d e %d got %d want %d\n", lane, e, h[lane*4+e], want); ++bad; } }
    }
    printf("tr_probe: %d mismatches\n", bad);
    return bad != 0;
}
