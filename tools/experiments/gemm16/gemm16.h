// Device helpers of the "pk16" operand format (gemm16.hip): fp16 hi / lo pieces of range-scaled fp32 values, 32 columns
// per 128-byte block [32 x hi | 32 x lo].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace raftcorr {

// exponent e such that max * 2^e lies in [2^14, 2^15); 0 when there is no finite positive maximum
__device__ __forceinline__ int pk16_exp_of_max(float m)
{
    if (!(m > 0.f) || !(m < 3.0e38f)) return 0;
    int e;
    frexpf(m, &e);
    return 15 - e;
}

// the same from 128 partial maxima (every lane of the calling wave gets the result)
__device__ __forceinline__ int pk16_range_exp(const float* part)
{
    const int lane = threadIdx.x & 63;
    float m = fmaxf(part[lane], part[lane + 64]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return pk16_exp_of_max(m);
}

// 8-byte index (inside a pk16 row) of the hi pieces of columns c .. c + 3 (c a multiple of 4); the lo pieces are 8 further
__device__ __forceinline__ int pk16_u2(int c) { return (c >> 5) * 16 + ((c & 31) >> 2); }

// tile-major pk16 ("pk16t"): rows in tiles of 128; line index (128-byte units) of column block j of row q inside an
// image of nblk column blocks per row:  tile (q >> 7) holds nblk chunks of [128 rows][128 B]
__host__ __device__ __forceinline__ size_t pk16_tile_line(int q, int j, int nblk)
{
    return ((size_t)(q >> 7) * nblk + j) * 128 + (q & 127);
}

// v * scale -> hi = RNE fp16, lo = fp16 of the exact remainder
__device__ __forceinline__ void pk16_split4(const float (&v)[4], float scale, uint2& hi, uint2& lo)
{
    unsigned short h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = v[j] * scale;
        const _Float16 a = (_Float16)x;
        const _Float16 b = (_Float16)(x - (float)a);
        h[j] = __builtin_bit_cast(unsigned short, a);
        l[j] = __builtin_bit_cast(unsigned short, b);
    }
    hi = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
    lo = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
}

}  // namespace raftcorr
