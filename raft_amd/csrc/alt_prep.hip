// Frame-pair set-up and tear-down of the on-the-fly correlation block (core/corr.py:64-92), each as ONE host call:
//
//   launch_alt_prepare       ranges of the two maps, the fmap2 pyramid of corr.py:69-73 (avg_pool2d chain) and every packed
//                            copy the lookup kernels read (alt_fwd16.hip, alt_bwd.hip) -- what the block's constructor used
//                            to do with 8 pooling calls and 11 library calls from Python (about 20 small kernels with host
//                            gaps between them: 240 us per pair at KITTI size, now launches back to back)
//   launch_alt_features_bwd  adjoint of that set-up: the NHWC gradient sums of the lookups (d fmap1; d fmap2 per pyramid
//                            level) -> NCHW gradients of the two maps, the pooled levels folded back through the
//                            avg_pool2d chain -- one kernel instead of five layout copies, three pooling backward
//                            kernels and three additions.
//
// HBM-bound byte shuffling, a few MB per pair.
#include "common.h"

#include <algorithm>

namespace raftcorr {
namespace {

// avg_pool2d(x, 2, stride=2) of a (BC, Hi, Wi) map: the sum runs row by row as in torch's kernel, then one exact
// division by 4: pooled levels equal torch's bit for bit.  grid ceil(BC * Ho * Wo / 256)
__global__ __launch_bounds__(256)
void alt_pool_kernel(const float* __restrict__ in, float* __restrict__ out, long long total, int Hi, int Wi)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int Ho = Hi >> 1, Wo = Wi >> 1;
    const int x = (int)(idx % Wo), y = (int)((idx / Wo) % Ho);
    const long long bc = idx / ((long long)Wo * Ho);
    const float* src = in + (bc * Hi + 2 * y) * Wi + 2 * x;
    const float2 a = *(const float2*)src, b = *(const float2*)(src + Wi);        // Wi even, base 8-byte aligned ...
    out[idx] = (((a.x + a.y) + b.x) + b.y) * 0.25f;
}

__global__ __launch_bounds__(256)
void alt_pool_odd_kernel(const float* __restrict__ in, float* __restrict__ out, long long total, int Hi, int Wi)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;             // ... scalar loads otherwise
    if (idx >= total) return;
    const int Ho = Hi >> 1, Wo = Wi >> 1;
    const int x = (int)(idx % Wo), y = (int)((idx / Wo) % Ho);
    const long long bc = idx / ((long long)Wo * Ho);
    const float* src = in + (bc * Hi + 2 * y) * Wi + 2 * x;
    out[idx] = (((src[0] + src[1]) + src[Wi]) + src[Wi + 1]) * 0.25f;
}

struct FeatBwdArgs {
    const float* d1;                              // (B, H, W, C)
    const float* d2[RAFTCORR_MAX_LEVELS];         // level l: (B, H >> l, W >> l, C)
    float* df1;                                   // (B, C, H, W)
    float* df2;                                   // (B, C, H, W)
    int B, C, H, W, L;
};

// grid (ceil(H*W / 64), ceil(C / 64), 2 B): z < B: d fmap1 (layout change only); z >= B: d fmap2 = level 0 + the pooled
// levels, each passed down the chain as avg_pool2d's backward does (a quarter of the parent cell, nothing for the odd
// last row / column that the pooling dropped), in that order: t_l = d2_l + t_(l+1) / 4.
// Read: lane = channel (NHWC rows are channel-contiguous), written after an LDS transpose with lane = pixel.
__global__ __launch_bounds__(256)
void alt_features_bwd_kernel(FeatBwdArgs p)
{
    __shared__ float tile[64][65];
    const int HW = p.H * p.W;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const bool second = (int)blockIdx.z >= p.B;
    const int b = second ? blockIdx.z - p.B : blockIdx.z;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = c0 + lane;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int pp = w + 4 * i, pix = p0 + pp;
        float t = 0.f;
        if (pix < HW && c < p.C) {
            if (!second) {
                t = p.d1[((size_t)b * HW + pix) * p.C + c];
            } else {
                const int y = pix / p.W, x = pix - y * p.W;
                float v[RAFTCORR_MAX_LEVELS];                // all levels' loads first: they do not depend on each other
#pragma unroll
                for (int l = 0; l < RAFTCORR_MAX_LEVELS; ++l) {
                    const int Hl = p.H >> l, Wl = p.W >> l, yl = y >> l, xl = x >> l;
                    v[l] = (l < p.L && yl < Hl && xl < Wl) ? p.d2[l][(((size_t)b * Hl + yl) * Wl + xl) * p.C + c] : 0.f;
                }
#pragma unroll
                for (int l = RAFTCORR_MAX_LEVELS - 1; l >= 0; --l) {
                    const int Hl = p.H >> l, Wl = p.W >> l, yl = y >> l, xl = x >> l;
                    // the running sum lives at level l + 1; it reaches this level-l cell only if the pooling kept the cell
                    // (a cell outside its level has v = 0 and no parent either: (yl >> 1) >= (Hl >> 1) then)
                    t = ((yl >> 1) < (Hl >> 1) && (xl >> 1) < (Wl >> 1)) ? t * 0.25f : 0.f;
                    t = (yl < Hl && xl < Wl) ? t + v[l] : 0.f;
                }
            }
        }
        tile[pp][lane] = t;
    }
    __syncthreads();
    float* out = (second ? p.df2 : p.df1) + (size_t)b * p.C * HW;
    const int pix = p0 + lane;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int cc = w + 4 * i;
        if (pix < HW && c0 + cc < p.C) out[(size_t)(c0 + cc) * HW + pix] = tile[lane][cc];
    }
}

size_t level_floats(int B, int C, int H, int W, int l) { return (size_t)B * C * (H >> l) * (W >> l); }

}  // namespace

size_t alt_prepare_workspace_bytes(int B, int C, int H, int W, int L)
{
    size_t n = 0;
    for (int l = 1; l < L; ++l) n += (level_floats(B, C, H, W, l) + 63) / 64 * 64;
    return std::max<size_t>(n * sizeof(float), 256);
}

int launch_alt_prepare(const float* fmap1, const float* fmap2, int B, int C, int H, int W, int L, float* scales,
                       void* packed1, void* const* packed2, void* tiles1, void* const* tiles2,
                       void* workspace, size_t workspace_bytes, hipStream_t s)
{
    if (!workspace || workspace_bytes < alt_prepare_workspace_bytes(B, C, H, W, L))
        return fail(RAFTCORR_EINVAL, "alt_prepare: workspace of %zu bytes needed, %zu given",
                    alt_prepare_workspace_bytes(B, C, H, W, L), workspace_bytes);
    if (((uintptr_t)workspace & 255) != 0) return fail(RAFTCORR_EINVAL, "alt_prepare: workspace must be 256-byte aligned");
    int rc = launch_maxabs(fmap1, fmap2, (size_t)B * C * H * W, scales, s);
    if (rc) return rc;
    const float* level[RAFTCORR_MAX_LEVELS] = {fmap2};
    float* next = (float*)workspace;
    for (int l = 1; l < L; ++l) {
        const int Hi = H >> (l - 1), Wi = W >> (l - 1);
        const long long total = (long long)level_floats(B, C, H, W, l);
        if (total >= (1ll << 31) * 256) return fail(RAFTCORR_EUNSUPPORTED, "alt_prepare: level %d too large", l);
        const dim3 grid((unsigned)((total + 255) / 256));
        if (Wi % 2 == 0 && ((uintptr_t)level[l - 1] & 7) == 0) hipLaunchKernelGGL(alt_pool_kernel, grid, dim3(256), 0, s, level[l - 1], next, total, Hi, Wi);
        else             hipLaunchKernelGGL(alt_pool_odd_kernel, grid, dim3(256), 0, s, level[l - 1], next, total, Hi, Wi);
        rc = check_launch("alt_pool_kernel");
        if (rc) return rc;
        level[l] = next;
        next += ((size_t)total + 63) / 64 * 64;
    }
    // fmap1 and the L levels of fmap2: one launch per packed format
    const float* maps[RAFTCORR_MAX_LEVELS + 1] = {fmap1};
    const float* mp[RAFTCORR_MAX_LEVELS + 1] = {scales};
    void* dst[RAFTCORR_MAX_LEVELS + 1] = {packed1};
    void* dstt[RAFTCORR_MAX_LEVELS + 1] = {tiles1};
    int hs[RAFTCORR_MAX_LEVELS + 1] = {H}, ws[RAFTCORR_MAX_LEVELS + 1] = {W};
    for (int l = 0; l < L; ++l) {
        maps[l + 1] = level[l]; mp[l + 1] = scales + 128; dst[l + 1] = packed2[l]; hs[l + 1] = H >> l; ws[l + 1] = W >> l;
        if (tiles2) dstt[l + 1] = tiles2[l];
    }
    rc = launch_alt_pack_set(maps, hs, ws, mp, dst, L + 1, B, C, s);
    if (rc || !tiles2) return rc;
    // tile-major copies for the backward pass: the fmap2 pyramid, and fmap1 when asked for (the window-group backward
    // gathers its fmap1 images from packed1 per call and does not read a tile-major copy)
    if (tiles1) return launch_alt_pack_tiles_set(maps, hs, ws, mp, dstt, L + 1, B, C, s);
    return launch_alt_pack_tiles_set(maps + 1, hs + 1, ws + 1, mp + 1, dstt + 1, L, B, C, s);
}

int launch_alt_features_bwd(const float* d1, const float* const* d2, int B, int C, int H, int W, int L,
                            float* df1, float* df2, hipStream_t s)
{
    const long long HW = (long long)H * W;
    if (2ll * B > 65535 || (C + 63) / 64 > 65535 || HW >= (1ll << 31) - 64)
        return fail(RAFTCORR_EUNSUPPORTED, "alt_features_bwd: %d x %d x %d x %d is too large", B, C, H, W);
    FeatBwdArgs a = {};
    a.d1 = d1; a.df1 = df1; a.df2 = df2;
    for (int l = 0; l < L; ++l) a.d2[l] = d2[l];
    a.B = B; a.C = C; a.H = H; a.W = W; a.L = L;
    hipLaunchKernelGGL(alt_features_bwd_kernel, dim3((unsigned)((HW + 63) / 64), (C + 63) / 64, 2 * B), dim3(256), 0, s, a);
    return check_launch("alt_features_bwd_kernel");
}

}  // namespace raftcorr
