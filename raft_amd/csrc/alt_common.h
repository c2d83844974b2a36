// Shared by the on-the-fly kernels (alt_fwd16.hip, alt_bwd.hip): the 8x8 query tile, its window origins and the
// bounding box of its windows at one level.
#pragma once
#include "common.h"
#include <limits.h>

namespace raftcorr {
namespace {

constexpr int QW = 8, QH = 8, TQ = QW * QH;       // query tile
constexpr int MAXPARTS16 = 128;                   // partial maxima per map (corr_maxabs_kernel, build_pyramid.hip)

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wave_max128(float m)       // m: max of this lane's two partial maxima
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}

// power-of-two factor that maps max|x| into [2^14, 2^15) (as build_pyramid.hip:range_exponent)
__device__ __forceinline__ int range_exp16(float m)
{
    if (!(m > 0.f) || !(m < 3.0e38f)) return 0;
    int e;
    frexpf(m, &e);
    return 15 - e;
}

// direct-to-LDS load, 64 lanes x 16 B: per-lane global source, LDS destination = chunk base + lane * 16
__device__ __forceinline__ void glds16(const uint4* gsrc, uint4* lds_chunk)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_chunk, 16, 0, 0);
}

// up to 1 + RAFTCORR_MAX_LEVELS maps of one batch packed by one launch (fmap1 and the fmap2 pyramid): blockIdx.x in
// [blk0[m], blk0[m + 1]) works on map m
struct PackSet {
    const float* src[RAFTCORR_MAX_LEVELS + 1];
    uint4* dst[RAFTCORR_MAX_LEVELS + 1];
    const float* maxpart[RAFTCORR_MAX_LEVELS + 1];
    int h[RAFTCORR_MAX_LEVELS + 1], w[RAFTCORR_MAX_LEVELS + 1], blk0[RAFTCORR_MAX_LEVELS + 2];
    int n, C, Cp;
};

__device__ __forceinline__ int packset_map(const PackSet& ps, int bx)
{
    int m = 0;
    while (m + 1 < ps.n && bx >= ps.blk0[m + 1]) ++m;
    return m;
}

struct TileQ {
    int wx0[TQ], wy0[TQ];                         // window origin (x0 - r, y0 - r) per query
    float fx[TQ], fy[TQ];
    int qpix[TQ];                                 // query pixel index, -1 outside the map
    int box[4];
};

// Args needs: hl[], wl[], H, W, L, r, band (tiles per XCD band).  Workgroup id -> (XCD band, tile,
// level, batch): each XCD (id mod 8) owns a band of tiles, whose part of the target maps stays in that XCD's L2.
struct Tile16 {
    int b, l, ty0, tx0, Hl, Wl, HWl;
    bool live;
    int xmin, ymin, bw, bh, U;
};

// window origins of the tile's queries at level t.l + the bounding box of their in-level parts (wave 0: one lane
// per query).  t.b, t.l, t.ty0, t.tx0 must be set; contains a barrier (every thread of the workgroup calls it).
// cb: the coordinates of batch item t.b, (2, H, W).  qsrc: nullptr = the 8x8 pixel tile at (t.ty0, t.tx0); else 64
// query pixel indices (-1 = empty slot).
template <class Args>
__device__ __forceinline__ void tile_level_setup(const Args& p, TileQ& sh, Tile16& t, const float* __restrict__ cb,
                                                 const int* __restrict__ qsrc = nullptr)
{
    t.Hl = p.hl[t.l]; t.Wl = p.wl[t.l]; t.HWl = t.Hl * t.Wl;
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int qi = tid, gy = t.ty0 + (qi >> 3), gx = t.tx0 + (qi & 7);
        const int pix = qsrc ? qsrc[qi] : ((gy < p.H && gx < p.W) ? gy * p.W + gx : -1);
        const bool valid = pix >= 0;
        const int T = 2 * p.r + 2;
        int wx0 = -(1 << 24), wy0 = -(1 << 24);
        float fx = 0.f, fy = 0.f;
        if (valid) {
            const size_t P = (size_t)p.H * p.W, q = (size_t)pix;
            const float* c = cb + q;
            const float s = 1.0f / (float)(1 << t.l);        // exact power of two == coords / 2**l (corr.py:86)
            int x0, y0;
            split_coord(c[0] * s, x0, fx);
            split_coord(c[P] * s, y0, fy);
            wx0 = x0 - p.r; wy0 = y0 - p.r;
        }
        const int qx0 = max(wx0, 0), qx1 = min(wx0 + T - 1, t.Wl - 1);
        const int qy0 = max(wy0, 0), qy1 = min(wy0 + T - 1, t.Hl - 1);
        const bool has = valid && qx0 <= qx1 && qy0 <= qy1;
        int mnx = has ? qx0 : INT_MAX, mxx = has ? qx1 : INT_MIN;
        int mny = has ? qy0 : INT_MAX, mxy = has ? qy1 : INT_MIN;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, m)); mxx = max(mxx, __shfl_xor(mxx, m));
            mny = min(mny, __shfl_xor(mny, m)); mxy = max(mxy, __shfl_xor(mxy, m));
        }
        sh.wx0[qi] = wx0; sh.wy0[qi] = wy0; sh.fx[qi] = fx; sh.fy[qi] = fy;
        sh.qpix[qi] = pix;
        if (tid == 0) { sh.box[0] = mnx; sh.box[1] = mxx; sh.box[2] = mny; sh.box[3] = mxy; }
    }
    __syncthreads();
    const int mnx = sh.box[0], mxx = sh.box[1], mny = sh.box[2], mxy = sh.box[3];
    if (mnx > mxx || mny > mxy) { t.xmin = 0; t.ymin = 0; t.bw = 1; t.bh = 0; t.U = 0; }
    else { t.xmin = mnx; t.ymin = mny; t.bw = mxx - mnx + 1; t.bh = mxy - mny + 1; t.U = t.bw * t.bh; }
}

}  // namespace
}  // namespace raftcorr
