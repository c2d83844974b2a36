// Radius-r bilinear window lookup over the correlation pyramid, forward and backward.
//
// Replaces CorrBlock.__call__ + utils.bilinear_sampler (core/corr.py:29-51,
// core/utils/utils.py:57-65): per level ~10 tiny launches (linspace, meshgrid, stack, add,
// split, 2 affine maps, cat, grid_sample, view) plus a final cat/contiguous copy.  Here every
// level of every query is produced by one launch.
//
// All (2r+1)^2 samples of one (query, level) share the same fractional offsets, so they
// are a 2x2 box filter over the (2r+2)^2 integer taps around floor(coords):
//   out(a, b) = w00*T[b][a] + w01*T[b][a+1] + w10*T[b+1][a] + w11*T[b+1][a+1]
// with a = x offset (slow output index), b = y offset (fast), T[ty][tx] the volume entry at
// (y0 - r + ty, x0 - r + tx) or 0 outside the level (grid_sample's zero padding).
//
// A block handles QB = 32 consecutive queries of one level (grid.y = level):
//   phase 0  the tap address splits into a y-only and an x-only part (patch-tile order,
//            layout.h); they are tabulated per (query, window row) / (query, window column),
//            2 x 32 x (2r+2) entries, each thread loading the coordinate it needs  -> LDS
//   phase 1  all threads gather the 32 x (2r+2)^2 taps.  The kernel is latency- and
//            issue-bound (one block = three dependent memory round trips, one wave per
//            SIMD), so all of a thread's loads are issued back to back into registers,
//            branch-free (out-of-range taps read a valid dummy address and are zeroed by a
//            select), offset = ytab + xtab, level kind resolved at compile time    -> LDS
//   phase 2  all threads blend 4 LDS taps per output and store coalesced: along the
//            channel dim for the (B,N,C) layout, along the query dim for (B,C,H,W).
//            Output stores are non-temporal: the result is consumed by another kernel and
//            should not evict volume lines that the next iteration's windows re-touch.
// The backward pass mirrors it: dOut tile -> LDS, each tap's gradient is the 2x2 box
// filter of dOut, added to d(level) with a plain read-modify-write: a (query, tap) cell
// belongs to exactly one thread of one block, so no atomics are needed.
#include "common.h"
#include <algorithm>

#ifndef RC_STAMP            // tools/lookup_lab2.hip defines it for diagnostic (timing-stamp) builds
#define RC_STAMP(i)
#endif

namespace raftcorr {

constexpr int QB = 32;

struct LookupArgs {
    LevelSet lv;        // volume levels (fwd: read; bwd: read for dcoords only)
    LevelSet dlv;       // bwd: gradient levels, accumulated
    const float* coords;
    const float* dout;  // bwd
    float* out;         // fwd
    float* dcoords;     // bwd, optional
    int B, N, L, r;
    int mode;           // RAFTCORR_MODE_*
    float rcp_w[RAFTCORR_MAX_LEVELS], rcp_h[RAFTCORR_MAX_LEVELS];   // RN(1 / (W_l - 1)), RN(1 / (H_l - 1)) (ref_sample_pos)
};

static void set_reciprocals(LookupArgs& a)
{
    for (int l = 0; l < RAFTCORR_MAX_LEVELS; ++l) {
        a.rcp_w[l] = a.lv.g[l].w > 1 ? 1.0f / (float)(a.lv.g[l].w - 1) : 0.f;      // host division: correctly rounded
        a.rcp_h[l] = a.lv.g[l].h > 1 ? 1.0f / (float)(a.lv.g[l].h - 1) : 0.f;
    }
}

__device__ __forceinline__ void load_xy(const LookupArgs& p, int b, int q, int l, float& x, float& y)
{
    if (p.mode == RAFTCORR_MODE_FORK) {
        const float* c = p.coords + ((size_t)b * p.N + q) * 2;
        x = c[0]; y = c[1];
    } else {
        const float* c = p.coords + (size_t)b * 2 * p.N + q;
        const float s = 1.0f / (float)(1 << l);          // exact power of two == coords / 2**l
        x = c[0] * s; y = c[p.N] * s;
    }
}

// level_pos (layout.h) with the level kind known at compile time
template <int KIND>
__device__ __forceinline__ int pos_k(int pw, int w, int y, int x)
{
    if (KIND == 0) {
        const int patch = (y >> 3) * pw + (x >> 4);
        const int n = (((y >> 2) & 1) << 1) | ((x >> 3) & 1);
        const int dy = y & 3, dx = x & 7;
        const int j = (dx & 1) | ((dy & 1) << 1) | ((dy >> 1) << 2) | (((dx >> 1) & 1) << 3) | ((dx >> 2) << 4);
        return patch * 128 + n * 32 + j;
    } else if (KIND == 1) {
        const int patch = (y >> 2) * pw + (x >> 3);
        const int n = (((y >> 1) & 1) << 1) | ((x >> 2) & 1);
        return patch * 32 + n * 8 + ((x & 1) | ((y & 1) << 1) | (((x >> 1) & 1) << 2));
    } else if (KIND == 2) {
        const int patch = (y >> 1) * pw + (x >> 2);
        return patch * 8 + ((x & 1) | ((y & 1) << 1) | (((x >> 1) & 1) << 2));
    } else if (KIND == 3) {
        return ((y >> 1) * pw + (x >> 1)) * 4 + (x & 1) + (y & 1) * 2;
    }
    return y * w + x;
}

// pos_k splits into a y-only and an x-only part (the bit fields are disjoint), so a block can
// tabulate them once per (query, window row) and (query, window column):
//   pos_k<KIND>(pw, w, y, x) == ypart_k<KIND>(pw, w, y) + xpart_k<KIND>(x)
template <int KIND>
__device__ __forceinline__ int ypart_k(int pw, int w, int y)
{
    if (KIND == 0) return (y >> 3) * pw * 128 + ((y >> 2) & 1) * 64 + (y & 1) * 2 + ((y >> 1) & 1) * 4;
    if (KIND == 1) return (y >> 2) * pw * 32 + ((y >> 1) & 1) * 16 + (y & 1) * 2;
    if (KIND == 2) return (y >> 1) * pw * 8 + (y & 1) * 2;
    if (KIND == 3) return (y >> 1) * pw * 4 + (y & 1) * 2;
    return y * w;
}
template <int KIND>
__device__ __forceinline__ int xpart_k(int x)
{
    if (KIND == 0) return (x >> 4) * 128 + ((x >> 3) & 1) * 32 + (x & 1) + ((x >> 1) & 1) * 8 + ((x >> 2) & 1) * 16;
    if (KIND == 1) return (x >> 3) * 32 + ((x >> 2) & 1) * 8 + (x & 1) + ((x >> 1) & 1) * 4;
    if (KIND == 2) return (x >> 2) * 8 + (x & 1) + ((x >> 1) & 1) * 4;
    if (KIND == 3) return (x >> 1) * 4 + (x & 1);
    return x;
}

constexpr int TAB_INVALID = -(1 << 30);    // marks a window row / column outside the level

// Where the reference actually samples.  bilinear_sampler (core/utils/utils.py:61-65) maps every
// sample position p = coords + delta (core/corr.py:42-44: the window offset is added FIRST, in fp32)
// to [-1, 1] and grid_sample(align_corners=True) maps it back:
//     n = 2 p / (dim - 1);  g = n - 1;  ix = (g + 1) * ((dim - 1) / 2)        (each step rounded to fp32)
// ((dim-1)/2 is exact, so ATen's CPU form above and its CUDA form ((g+1)/2)*(dim-1) round identically.)
// The round trip moves a sample by O(1e-7 * dim) pixels -- 2e-6 of max|out| at W = 40, but 1e-5 at
// W = 240 (fixtures K1 / P1 under tests/golden) -- and differently for each of the 2r+1
// window offsets: the samples of a window do not share one fractional offset exactly.  This returns
// ix for window offset `a`, every step one correctly rounded fp32 operation (contraction off), i.e.
// bit for bit the reference's.  The division uses the host's correctly rounded reciprocal rcp =
// RN(1 / (dim-1)): q0 = RN(n rcp), r = n - q0 (dim-1) (exact, FMA), q = RN(q0 + r rcp) is the correctly
// rounded quotient (Markstein); three instructions instead of the IEEE division sequence.
// Levels one pixel wide / high divide by zero in the reference (NaN); there the pixel position is used.
__device__ __forceinline__ float ref_sample_pos(float c, int a_minus_r, int dim, float rcp)
{
#pragma clang fp contract(off)
    const float p = c + (float)a_minus_r;
    if (dim <= 1) return p;
    const float dm1 = (float)(dim - 1);
    const float n2 = 2.0f * p;
    const float q0 = n2 * rcp;
    const float rem = __builtin_fmaf(-q0, dm1, n2);
    const float n = __builtin_fmaf(rem, rcp, q0);
    const float g = n - 1.0f;
    const float u = g + 1.0f;
    return u * (0.5f * dm1);
}

// (low tap, fraction) of window offset a on one axis, relative to `origin` (the cell index of tap 0 of
// the staged tile): the low tap is floor(ix) and nearly always origin-relative index `nominal`; when
// the round trip carried ix across an integer it is one off, and when that tap pair is not part of the
// staged tile (j outside [0, jmax]) the nearest staged pair is used with a fraction slightly outside
// [0, 1] (linear extrapolation over <= 3e-5 px: an error of second order).
__device__ __forceinline__ void ref_sample_tap(float c, bool sane, int a, int r, int dim, float rcp, int origin,
                                               int nominal, float fc, int jmax, int& j, float& t)
{
    if (!sane) { j = nominal; t = fc; return; }
    const float ix = ref_sample_pos(c, a - r, dim, rcp);
    const int i0 = (int)floorf(ix);
    j = min(max(i0 - origin, max(nominal - 1, 0)), min(nominal + 1, jmax));
    t = ix - (float)(origin + j);
}

struct LookupSmem {
    float* tile;   // fwd: [QB][ST] taps;  bwd: [QB][SG] dOut
    float* sfx; float* sfy;
    int* xtab;     // [QB][T]  x part of the tap offset, TAB_INVALID outside
    int* ytab;     // [QB][T]  y part + query slab offset (relative to the block's first query)
};

__device__ __forceinline__ LookupSmem carve(float* smem, int r, bool bwd)
{
    const int T = 2 * r + 2, rd = 2 * r + 1;
    const int stride = bwd ? ((rd * rd) | 1) : ((T * T) | 1);
    LookupSmem s;
    s.tile = smem;
    s.sfx = smem + QB * stride;
    s.sfy = s.sfx + QB;
    s.xtab = (int*)(s.sfy + QB);
    s.ytab = s.xtab + QB * T;
    return s;
}

// phase 0: every thread loads the coordinate it needs itself (the 2r+2 lanes of a query hit the
// same address) and writes one table entry; one barrier later all address parts are in LDS.
template <int R, int KIND, int NT = 256>
__device__ __forceinline__ void build_tables(const LookupArgs& p, const LookupSmem& s, const LevelGeom& g,
                                             int b, int q0, int l, int r)
{
    const int T = 2 * r + 2;
    const bool fork = p.mode == RAFTCORR_MODE_FORK;
    const float cscale = fork ? 1.0f : 1.0f / (float)(1 << l);       // exact power of two == coords / 2**l
    // coordinate component behind table entry e, unscaled.  Branch-free (clamped, always valid
    // address): with a branch per mode hipcc waits vmcnt(0) after every load.
    auto load_c = [&](int e) -> float {
        const bool isy = e >= QB * T;
        const int ee = isy ? e - QB * T : e;
        const int q = min(q0 + min(ee / T, QB - 1), p.N - 1);
        const size_t idx = fork ? ((size_t)b * p.N + q) * 2 + (isy ? 1 : 0)
                                : ((size_t)b * 2 + (isy ? 1 : 0)) * p.N + q;
        return p.coords[idx];
    };
    auto emit = [&](int e, float c) {
        if (e >= 2 * QB * T) return;
        const bool isy = e >= QB * T;
        const int ee = isy ? e - QB * T : e;
        const int qi = ee / T, k = ee - qi * T;
        int c0; float fc;
        split_coord(c * cscale, c0, fc);
        const int v = c0 - r + k;
        if (isy) {
            const bool ok = q0 + qi < p.N && v >= 0 && v < g.h;
            s.ytab[ee] = ok ? ypart_k<KIND>(g.pw, g.w, v) + qi * (int)g.slab : TAB_INVALID;
            if (k == 0) s.sfy[qi] = fc;
        } else {
            const bool ok = v >= 0 && v < g.w;
            s.xtab[ee] = ok ? xpart_k<KIND>(v) : TAB_INVALID;
            if (k == 0) s.sfx[qi] = fc;
        }
    };
    if (R >= 0) {
        constexpr int Rc = R >= 0 ? R : 0;
        constexpr int NTAB = (2 * QB * (2 * Rc + 2) + NT - 1) / NT;
        float c[NTAB];
#pragma unroll
        for (int i = 0; i < NTAB; ++i) c[i] = load_c(threadIdx.x + i * NT);     // loads in flight together
#pragma unroll
        for (int i = 0; i < NTAB; ++i) emit(threadIdx.x + i * NT, c[i]);
    } else {
        for (int e = threadIdx.x; e < 2 * QB * T; e += NT) emit(e, load_c(e));
    }
}

// ---- forward, fused levels (kinds 0..3), compile-time radius ---------------------------------
// The window's (2r+2)^2 taps are fetched as even-aligned 2x2 blocks: layout.h makes every such
// block of every fused level one 16-byte aligned float4 (inner order x&1 | (y&1)<<1), so a
// 10x10 window costs 36 loads and address computations instead of 100.  Per query the blocks
// cover a TW x TW = (2r+4)^2 cell tile starting at the even-aligned origin; the window sits at
// parity offset (px, py) inside it.
template <int R>
struct BlockGeom {
    static constexpr int T = 2 * R + 2;          // taps per window side
    static constexpr int NB = T / 2 + 1;         // 2x2 blocks per tile side
    static constexpr int TW = 2 * NB;            // tile side in cells
    static constexpr int STQ = (TW * TW) | 1;    // LDS stride per query (odd: conflict-free)
    static constexpr int RD = 2 * R + 1;
    // floats of LDS per 32-query block: tile, block-offset tables (x, y), masks (x, y), per-sample
    // tables: x (low tap, fraction), y (4 row weights)
    static constexpr int SMEM = QB * STQ + 2 * QB * NB + 2 * QB + 2 * QB * RD + 4 * QB * RD;
};

template <int R>
__device__ __forceinline__ size_t fwd4_smem_floats()
{
    return (size_t)BlockGeom<R>::SMEM;
}

// xsink == nullptr: results go to p.out.  Otherwise (fused consumer, canonical mode) the block's
// results are left in LDS as xsink[(l * rd2 + channel) * XS + query] for a GEMM that follows.
constexpr int XS = QB + 1;

__device__ __forceinline__ unsigned short bf16_rne(float x)
{
    unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// Volume element access by type (RAFTCORR_VOL_*: 0 float32, 1 fp16, 2 bf16).  `vol` is the level's base as a byte
// pointer's worth of float* (api.hip hands 16-bit levels through the same LevelSet); offsets are in ELEMENTS.
template <int VOL>
__device__ __forceinline__ float vol_elem(unsigned short h)
{
    if (VOL == 1) return (float)__builtin_bit_cast(_Float16, h);
    return __uint_as_float((unsigned)h << 16);
}
template <int VOL>
__device__ __forceinline__ const float* vol_base(const float* level, size_t elems)
{
    return VOL == 0 ? level + elems : (const float*)((const unsigned short*)level + elems);
}
template <int VOL>
__device__ __forceinline__ float vol_load1(const float* vol, int o)
{
    if (VOL == 0) return vol[o];
    return vol_elem<VOL>(((const unsigned short*)vol)[o]);
}
template <int VOL>
__device__ __forceinline__ float4 vol_load_block(const float* vol, int o)        // one even-aligned 2x2 block
{
    if (VOL == 0) return *(const float4*)(vol + o);
    const uint2 w = *(const uint2*)((const unsigned short*)vol + o);             // 8 bytes
    return make_float4(vol_elem<VOL>((unsigned short)w.x), vol_elem<VOL>((unsigned short)(w.x >> 16)),
                       vol_elem<VOL>((unsigned short)w.y), vol_elem<VOL>((unsigned short)(w.y >> 16)));
}

// NT: threads that work on the block's 32 queries.  The stand-alone kernel runs QB * RD of them (R = 4: 288):
// one thread per (query, window column) in the blend, and the block loads (QB * NB^2), the table entries and
// the per-sample entries all divide evenly over them.  The fused kernel gives each level a 256-thread group.
template <int R, int KIND, int NT, int VOL = 0>
__device__ __forceinline__ void lookup_fwd_blocks(const LookupArgs& p, float* smem, int l, int b, int q0, int tid,
                                                  float* xsink = nullptr)
{
    typedef BlockGeom<R> G;
    constexpr int RD = G::RD, TW = G::TW, NB = G::NB;
    __builtin_assume(tid >= 0 && tid < NT);
    float4* wy = (float4*)smem;                          // [QB][RD] weights of tile rows (py + b - 1 .. py + b + 2)
    float* tile = smem + 4 * QB * RD;                    // [QB][STQ]
    int* bxtab = (int*)(tile + QB * G::STQ);             // [QB][NB] x part of the block offset
    int* bytab = bxtab + QB * NB;                        // [QB][NB] y part + query slab offset
    int* xmask = bytab + QB * NB;                        // validity bits of the TW tile columns
    int* ymask = xmask + QB;                             // validity bits of the TW tile rows; bit 16: parity py
    int* sjx = ymask + QB;                               // [QB][RD] low tap of window column a, tile-relative
    float* stx = (float*)(sjx + QB * RD);                // [QB][RD] fraction of window column a

    const LevelGeom g = p.lv.g[l];
    const int N = p.N;
    const float* __restrict__ vol = vol_base<VOL>(p.lv.ptr[l], ((size_t)b * N + q0) * g.slab);   // block's first slab

    RC_STAMP(0);
    // phase 0: block-offset tables + masks (2 QB NB entries).  The per-sample taps / fractions / row weights
    // (2 QB RD entries, ref_sample_tap) are computed in phase 1, between the issue of the tap loads and their
    // arrival -- arithmetic that needs only the coordinates, under memory latency the wave would otherwise
    // sit out.
    constexpr int NBE = 2 * QB * NB, NSE = 2 * QB * RD;
    constexpr int RB = (NBE + NT - 1) / NT, RS = (NSE + NT - 1) / NT;    // rounds of block / sample entries
    const bool fork = p.mode == RAFTCORR_MODE_FORK;
    const float cscale = fork ? 1.0f : 1.0f / (float)(1 << l);           // exact power of two == coords / 2**l
    // coordinate behind entry e of a kind with `per` entries per (axis, query).  Branch-free: a clamped
    // (always valid) address, every load issued before any is used -- with a branch per mode hipcc
    // waits vmcnt(0) after each load (serial round trips)
    auto load_c = [&](int e, int per) -> float {
        const bool isy = e >= QB * per;
        const int ee = isy ? e - QB * per : e;
        const int q = min(q0 + min(ee / per, QB - 1), N - 1);
        const size_t idx = fork ? ((size_t)b * N + q) * 2 + (isy ? 1 : 0)
                                : ((size_t)b * 2 + (isy ? 1 : 0)) * N + q;
        return p.coords[idx];
    };
    auto block_entry = [&](int e, float cc) {
        if (e >= NBE) return;
        const bool isy = e >= QB * NB;
        const int ee = isy ? e - QB * NB : e;
        const int qi = ee / NB, k = ee - qi * NB;
        int c0; float fc;
        split_coord(cc, c0, fc);
        const int cs = c0 - R;                       // first tap of the window
        const int ca = cs & ~1;                      // even-aligned tile origin
        const int dim = isy ? g.h : g.w, dimp = isy ? g.hp : g.wp;
        const int v = ca + 2 * k;                    // first cell of block k
        const bool ok = v >= 0 && v < dimp && (!isy || q0 + qi < N);
        if (isy) bytab[ee] = ok ? ypart_k<KIND>(g.pw, g.w, v) + qi * (int)g.slab : TAB_INVALID;
        else     bxtab[ee] = ok ? xpart_k<KIND>(v) : TAB_INVALID;
        if (k == 0) {
            const int lo = max(0, -ca), hi = min(TW, dim - ca);              // valid cells: [lo, hi)
            const int m = hi > lo ? (((1 << hi) - 1) & ~((1 << lo) - 1)) : 0;
            if (isy) ymask[qi] = m | ((cs & 1) << 16); else xmask[qi] = m;
        }
    };
    auto sample_entry = [&](int e, float cc) {
        if (e >= NSE) return;
        const bool isy = e >= QB * RD;
        const int ee = isy ? e - QB * RD : e;
        const int qi = ee / RD, k = ee - qi * RD;
        int c0; float fc;
        split_coord(cc, c0, fc);
        const int cs = c0 - R, ca = cs & ~1;
        const int n = (cs & 1) + k;                  // nominal low tap of window offset k, tile-relative
        int j; float t;
#if defined(RC_LAB_E1)
        j = n; t = fc;
#else
        ref_sample_tap(cc, c0 > -(1 << 23), k, R, isy ? g.h : g.w, isy ? p.rcp_h[l] : p.rcp_w[l], ca, n, fc,
                       TW - 2, j, t);
#endif
        if (!isy) {
            sjx[ee] = j; stx[ee] = t;
        } else {
            // weights of the tile rows n-1 .. n+2: the sample's tap pair is (n, n+1) unless the round
            // trip carried it across an integer (d = -1 / +1)
            const int d = j - n;
            float4 w;
            w.x = d < 0 ? 1.f - t : 0.f;
            w.y = d < 0 ? t : (d == 0 ? 1.f - t : 0.f);
            w.z = d == 0 ? t : (d > 0 ? 1.f - t : 0.f);
            w.w = d > 0 ? t : 0.f;
            wy[ee] = w;
        }
    };
    float cb[RB], cs_[RS];
#pragma unroll
    for (int i = 0; i < RB; ++i) cb[i] = load_c(tid + i * NT, NB);
#pragma unroll
    for (int i = 0; i < RS; ++i) cs_[i] = load_c(tid + i * NT, RD);
#pragma unroll
    for (int i = 0; i < RB; ++i) block_entry(tid + i * NT, cb[i] * cscale);
    __syncthreads();
    RC_STAMP(1);

    {   // phase 1: 2x2 blocks -> LDS tile, all loads of a thread in flight together
        constexpr int NBLK = NB * NB;
        constexpr int NIT = (QB * NBLK + NT - 1) / NT;
        constexpr bool EVEN = NT % NBLK == 0;             // a thread keeps its block (by, bx), its queries step by NT / NBLK
        float4 v[NIT];
        int off[NIT], xmk[NIT], ymk[NIT];
        const int qi0 = tid / NBLK, blk0 = tid - qi0 * NBLK;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + i * NT;
            const int qi = EVEN ? min(qi0 + i * (NT / NBLK), QB - 1) : min(e / NBLK, QB - 1);
            const int blk = EVEN ? blk0 : e - (e / NBLK) * NBLK;
            const int by = blk / NB, bx = blk - by * NB;
            const int o = bytab[qi * NB + by] + bxtab[qi * NB + bx];
            off[i] = (e < QB * NBLK && o >= 0) ? o : -1;
            v[i] = vol_load_block<VOL>(vol, max(o, 0));   // absent blocks read a valid dummy address
            xmk[i] = xmask[qi]; ymk[i] = ymask[qi];       // read now: no LDS round trip once the taps have landed
        }
        __builtin_amdgcn_sched_barrier(0);                // the loads are in flight: now the sample tables
#pragma unroll
        for (int i = 0; i < RS; ++i) sample_entry(tid + i * NT, cs_[i] * cscale);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + i * NT;
            if (e >= QB * NBLK) continue;
            const int qi = EVEN ? qi0 + i * (NT / NBLK) : e / NBLK;
            const int blk = EVEN ? blk0 : e - (e / NBLK) * NBLK;
            const int by = blk / NB, bx = blk - by * NB;
            const int xm = off[i] >= 0 ? (xmk[i] >> (2 * bx)) : 0;
            const int ym = (ymk[i] & 0xffff) >> (2 * by);
            float* t = tile + qi * G::STQ + (2 * by) * TW + 2 * bx;
            t[0] = ((xm & 1) && (ym & 1)) ? v[i].x : 0.f;
            t[1] = ((xm & 2) && (ym & 1)) ? v[i].y : 0.f;
            t[TW] = ((xm & 1) && (ym & 2)) ? v[i].z : 0.f;
            t[TW + 1] = ((xm & 2) && (ym & 2)) ? v[i].w : 0.f;
        }
    }
    __syncthreads();
    RC_STAMP(2);

    constexpr int rd2 = RD * RD;
    const int C = p.L * rd2;
    // one sample, the general way: 4 weighted tile rows x 2 columns.  row(i): tile row py + bb - 1 + i,
    // clamped at 0 (the weight of a row above the tile is zero by construction)
    auto sample = [&](int qi, int a, int bb) -> float {
        const float fx = stx[qi * RD + a];
        const float4 w = wy[qi * RD + bb];
        const int py = ymask[qi] >> 16;
        const float* tp = tile + qi * G::STQ + sjx[qi * RD + a];
        const float* r0 = tp + max(py + bb - 1, 0) * TW;
        const float* r1 = tp + (py + bb) * TW;
        const float h0 = r0[0] + fx * (r0[1] - r0[0]), h1 = r1[0] + fx * (r1[1] - r1[0]);
        const float h2 = r1[TW] + fx * (r1[TW + 1] - r1[TW]), h3 = r1[2 * TW] + fx * (r1[2 * TW + 1] - r1[2 * TW]);
        return w.x * h0 + w.y * h1 + w.z * h2 + w.w * h3;
    };
    if (p.mode == RAFTCORR_MODE_FORK) {
        // out (B, N, C): lanes run along the channel dim
        float* __restrict__ out = p.out + ((size_t)b * N + q0) * C + l * rd2;
#pragma unroll 2
        for (int e = tid; e < QB * rd2; e += NT) {        // (few samples in flight: each one is 14 LDS reads)
            const int qi = e / rd2, ch = e - qi * rd2;
            const int a = ch / RD;
            const float o = sample(qi, a, ch - a * RD);
            if (q0 + qi < N) __builtin_nontemporal_store(o, &out[(size_t)qi * C + ch]);
        }
    } else {
        // out (B, C, N): lanes run along the query dim.  A query's NT / QB threads each take one window column
        // a: the column is blended horizontally once for each of the RD + 3 tile rows its samples can touch (2
        // LDS reads per row), then every output is four weighted rows (one 16-byte LDS read of weights) -- no
        // selects, no dynamic register indices.  Columns beyond NT / QB (fused kernel, RD = 9: one) are dealt
        // out by row.  (Non-temporal global stores; the same stores through a buffer descriptor with a scalar
        // channel offset cost 1.4 us more per launch at Sintel size, tools/lookup_time.hip.)
        constexpr int TPQ = NT / QB;
        constexpr int NCOL = RD < TPQ ? RD : TPQ;             // columns handled whole
        constexpr int NEX = (RD - NCOL) * ((RD + TPQ - 1) / TPQ);   // outputs of the leftover columns, per thread
        const int qi = tid & (QB - 1), g8 = tid / QB;
        const bool vq = q0 + qi < N;
        const int a0 = min(g8, NCOL - 1);
        const float fx = stx[qi * RD + a0];
        const int py = ymask[qi] >> 16;
        const float* tcol = tile + qi * G::STQ + sjx[qi * RD + a0] + py * TW;
        float h[RD + 3];                                      // tile rows py - 1 .. py + RD + 1
        {
            const float* r0 = py ? tcol - TW : tcol;          // row above the tile: weight zero, any finite value
            h[0] = r0[0] + fx * (r0[1] - r0[0]);
        }
#pragma unroll
        for (int k = 1; k < RD + 3; ++k) {
            const float* rk = tcol + (k - 1) * TW;
            h[k] = rk[0] + fx * (rk[1] - rk[0]);
        }
        float ex[NEX > 0 ? NEX : 1];
#pragma unroll
        for (int i = 0; i < NEX; ++i)
            ex[i] = sample(qi, NCOL + i / ((RD + TPQ - 1) / TPQ), min(g8 + TPQ * (i % ((RD + TPQ - 1) / TPQ)), RD - 1));
        float* __restrict__ out = p.out + ((size_t)b * C + l * rd2) * N + q0;
        auto emit = [&](int ch, float v) {
            if (xsink) xsink[(l * rd2 + ch) * XS + qi] = vq ? v : 0.f;
            else if (vq) __builtin_nontemporal_store(v, &out[(size_t)ch * N + qi]);
        };
        if (g8 < NCOL) {
#pragma unroll
            for (int bb = 0; bb < RD; ++bb) {
                const float4 w = wy[qi * RD + bb];
                emit(g8 * RD + bb, w.x * h[bb] + w.y * h[bb + 1] + w.z * h[bb + 2] + w.w * h[bb + 3]);
            }
        }
#pragma unroll
        for (int i = 0; i < NEX; ++i) {
            const int a = NCOL + i / ((RD + TPQ - 1) / TPQ), bb = g8 + TPQ * (i % ((RD + TPQ - 1) / TPQ));
            if (bb < RD) emit(a * RD + bb, ex[i]);
        }
    }
    RC_STAMP(3);
}

template <int R, int KIND, int NT, int VOL = 0>
__device__ __forceinline__ void lookup_fwd_body(const LookupArgs& p, float* smem)
{
    const int r = (R >= 0) ? R : p.r;
    const int T = 2 * r + 2, TT = T * T, rd = 2 * r + 1, rd2 = rd * rd;
    const int ST = TT | 1;                       // odd stride: conflict-free query-major reads
    const LookupSmem s = carve(smem, r, false);
    float* taps = s.tile;
    int* sjx = s.ytab + QB * T;                  // [QB][rd] low tap of window column a (window-relative)
    int* sjy = sjx + QB * rd;
    float* stx = (float*)(sjy + QB * rd);        // [QB][rd] fraction of window column a
    float* sty = stx + QB * rd;

    const int l = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * QB;
    const int tid = threadIdx.x;
    const LevelGeom g = p.lv.g[l];
    const int N = p.N;
    const float* __restrict__ vol = vol_base<VOL>(p.lv.ptr[l], ((size_t)b * N + q0) * g.slab);   // block's first slab

    RC_STAMP(0);
    build_tables<R, KIND, NT>(p, s, g, b, q0, l, r);
    {   // per-sample (low tap, fraction): where the reference samples (ref_sample_tap)
        const bool fork = p.mode == RAFTCORR_MODE_FORK;
        const float cscale = fork ? 1.0f : 1.0f / (float)(1 << l);
        for (int e = tid; e < 2 * QB * rd; e += NT) {
            const bool isy = e >= QB * rd;
            const int ee = isy ? e - QB * rd : e;
            const int qi = ee / rd, a = ee - qi * rd;
            const int q = min(q0 + qi, N - 1);
            const size_t idx = fork ? ((size_t)b * N + q) * 2 + (isy ? 1 : 0)
                                    : ((size_t)b * 2 + (isy ? 1 : 0)) * N + q;
            const float c = p.coords[idx] * cscale;
            int c0; float fc;
            split_coord(c, c0, fc);
            int j; float t;
            ref_sample_tap(c, c0 > -(1 << 23), a, r, isy ? g.h : g.w, isy ? p.rcp_h[l] : p.rcp_w[l], c0 - r, a, fc, T - 2, j, t);
            if (isy) { sjy[ee] = j; sty[ee] = t; } else { sjx[ee] = j; stx[ee] = t; }
        }
    }
    __syncthreads();
    RC_STAMP(1);

    if (R >= 0) {
        constexpr int Rc = R >= 0 ? R : 0;
        constexpr int Tc = 2 * Rc + 2, TTc = Tc * Tc;
        constexpr int NIT = (QB * TTc + NT - 1) / NT;
        float v[NIT];
        int off[NIT];
        int qi = tid / TTc, t = tid - qi * TTc;   // tap e = tid + NT*i, walked incrementally
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int qc = min(qi, QB - 1);
            const int ty = t / Tc, tx = t - ty * Tc;
            const int o = s.ytab[qc * Tc + ty] + s.xtab[qc * Tc + tx];
            off[i] = (qi < QB && o >= 0) ? o : -1;
            v[i] = vol_load1<VOL>(vol, max(o, 0));   // invalid taps read a valid dummy address
            t += NT % TTc; qi += NT / TTc;
            if (t >= TTc) { t -= TTc; ++qi; }
        }
        qi = tid / TTc; t = tid - qi * TTc;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (qi < QB) taps[qi * ST + t] = off[i] >= 0 ? v[i] : 0.f;
            t += NT % TTc; qi += NT / TTc;
            if (t >= TTc) { t -= TTc; ++qi; }
        }
    } else {
        for (int e = tid; e < QB * TT; e += NT) {
            const int qi = e / TT;
            const int t = e - qi * TT;
            const int ty = t / T, tx = t - ty * T;
            const int o = s.ytab[qi * T + ty] + s.xtab[qi * T + tx];
            taps[qi * ST + t] = o >= 0 ? vol_load1<VOL>(vol, o) : 0.f;
        }
    }
    __syncthreads();
    RC_STAMP(2);

    const int C = p.L * rd2;
    auto sample = [&](int qi, int a, int bb) -> float {
        const float fx = stx[qi * rd + a], fy = sty[qi * rd + bb];
        const float* tp = taps + qi * ST + sjy[qi * rd + bb] * T + sjx[qi * rd + a];
        return (tp[0] * (1.f - fx) + tp[1] * fx) * (1.f - fy) + (tp[T] * (1.f - fx) + tp[T + 1] * fx) * fy;
    };
    if (p.mode == RAFTCORR_MODE_FORK) {
        // out (B, N, C): lanes run along the channel dim
        float* __restrict__ out = p.out + ((size_t)b * N + q0) * C + l * rd2;
        for (int e = tid; e < QB * rd2; e += NT) {
            const int qi = e / rd2;
            const int ch = e - qi * rd2;
            const int a = ch / rd;
            if (q0 + qi >= N) break;
            __builtin_nontemporal_store(sample(qi, a, ch - a * rd), &out[(size_t)qi * C + ch]);
        }
    } else {
        // out (B, C, N): lanes run along the query dim
        float* __restrict__ out = p.out + ((size_t)b * C + l * rd2) * N + q0;
        const int qi = tid & (QB - 1);
        const bool vq = q0 + qi < N;
        for (int ch = tid / QB; ch < rd2; ch += NT / QB) {
            const int a = ch / rd;
            const float o = sample(qi, a, ch - a * rd);
            if (vq) __builtin_nontemporal_store(o, &out[(size_t)ch * N + qi]);
        }
    }
    RC_STAMP(3);
}

// threads per block of the forward kernel: one per (query, window column) for the specialised radii
template <int R>
#if defined(RC_LAB_E3)
struct FwdThreads { static constexpr int NT = 256; };
#else
struct FwdThreads { static constexpr int NT = R >= 0 ? QB * (2 * R + 1) : 256; };
#endif

template <int R, int VOL = 0>
__global__ __launch_bounds__(FwdThreads<R>::NT)
void corr_lookup_fwd_kernel(LookupArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = FwdThreads<R>::NT;
    // level l < 4 has layout kind l, deeper levels are row-major (layout.h): the level index alone
    // selects the code path and, as a literal, makes the level's geometry a fixed kernarg offset --
    // no dependent scalar load in front of the first vector instruction
    const int kind = min((int)blockIdx.y, 4);    // block-uniform
    if constexpr (R >= 0) {
        switch (kind) {
        case 0: lookup_fwd_blocks<R, 0, NT, VOL>(p, smem, 0, blockIdx.z, blockIdx.x * QB, threadIdx.x); return;
        case 1: lookup_fwd_blocks<R, 1, NT, VOL>(p, smem, 1, blockIdx.z, blockIdx.x * QB, threadIdx.x); return;
        case 2: lookup_fwd_blocks<R, 2, NT, VOL>(p, smem, 2, blockIdx.z, blockIdx.x * QB, threadIdx.x); return;
        case 3: lookup_fwd_blocks<R, 3, NT, VOL>(p, smem, 3, blockIdx.z, blockIdx.x * QB, threadIdx.x); return;
        default: lookup_fwd_body<R, 4, NT, VOL>(p, smem); return;
        }
    } else {
        switch (kind) {
        case 0: lookup_fwd_body<R, 0, NT, VOL>(p, smem); break;
        case 1: lookup_fwd_body<R, 1, NT, VOL>(p, smem); break;
        case 2: lookup_fwd_body<R, 2, NT, VOL>(p, smem); break;
        case 3: lookup_fwd_body<R, 3, NT, VOL>(p, smem); break;
        default: lookup_fwd_body<R, 4, NT, VOL>(p, smem); break;
        }
    }
}

template <int R, int KIND>
__device__ __forceinline__ void lookup_bwd_body(const LookupArgs& p, float* smem)
{
    const int r = (R >= 0) ? R : p.r;
    const int T = 2 * r + 2, TT = T * T, rd = 2 * r + 1, rd2 = rd * rd;
    const int SG = rd2 | 1;
    const LookupSmem s = carve(smem, r, true);
    float* g = s.tile;                           // [QB][SG]  dOut tile

    const int l = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * QB;
    const int tid = threadIdx.x;
    const LevelGeom geo = p.dlv.g[l];
    const int N = p.N;
    float* __restrict__ dvol = p.dlv.ptr[l] + ((size_t)b * N + q0) * geo.slab;     // block's first slab
    const int C = p.L * rd2;

    build_tables<R, KIND>(p, s, geo, b, q0, l, r);
    // dOut tile -> LDS.  Compile-time radius: clamped (always valid) addresses, every load of the
    // thread issued before the first is used; a "valid ? load : 0" per element makes hipcc branch
    // around each load and wait for it -- eleven serial round trips per thread.
    const int qlast = N - 1 - q0;                                   // last valid query of the block, >= 0
    if (p.mode == RAFTCORR_MODE_FORK) {
        const float* __restrict__ d = p.dout + ((size_t)b * N + q0) * C + l * rd2;
        if (R >= 0) {
            constexpr int Rc = R >= 0 ? R : 0;
            constexpr int RD2 = (2 * Rc + 1) * (2 * Rc + 1), NLD = (QB * RD2 + 255) / 256;
            float dv[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = min(tid + i * 256, QB * RD2 - 1);
                const int qi = e / RD2, ch = e - qi * RD2;
                dv[i] = d[(size_t)min(qi, qlast) * C + ch];
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + i * 256;
                const int qi = e / RD2, ch = e - qi * RD2;
                if (e < QB * RD2) g[qi * SG + ch] = (qi <= qlast) ? dv[i] : 0.f;
            }
        } else {
            for (int e = tid; e < QB * rd2; e += 256) {
                const int qi = e / rd2;
                const int ch = e - qi * rd2;
                g[qi * SG + ch] = (q0 + qi < N) ? d[(size_t)qi * C + ch] : 0.f;
            }
        }
    } else {
        const float* __restrict__ d = p.dout + ((size_t)b * C + l * rd2) * N + q0;
        const int qi = tid & (QB - 1);
        const bool vq = q0 + qi < N;
        if (R >= 0) {
            constexpr int Rc = R >= 0 ? R : 0;
            constexpr int RD2 = (2 * Rc + 1) * (2 * Rc + 1), NLD = (RD2 + 256 / QB - 1) / (256 / QB);
            float dv[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                dv[i] = d[(size_t)min(tid / QB + i * (256 / QB), RD2 - 1) * N + min(qi, qlast)];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int ch = tid / QB + i * (256 / QB);
                if (ch < RD2) g[qi * SG + ch] = vq ? dv[i] : 0.f;
            }
        } else {
            for (int ch = tid / QB; ch < rd2; ch += 256 / QB)
                g[qi * SG + ch] = vq ? d[(size_t)ch * N + qi] : 0.f;
        }
    }
    __syncthreads();

    auto tap_grad = [&](int qi, int tx, int ty) -> float {
        const float fx = s.sfx[qi], fy = s.sfy[qi];
        const float* gq = g + qi * SG;           // channel = a*rd + b, a <-> x, b <-> y
        float acc = 0.f;
        if (tx < rd && ty < rd) acc += gq[tx * rd + ty] * ((1.f - fx) * (1.f - fy));
        if (tx > 0 && ty < rd)  acc += gq[(tx - 1) * rd + ty] * (fx * (1.f - fy));
        if (tx < rd && ty > 0)  acc += gq[tx * rd + ty - 1] * ((1.f - fx) * fy);
        if (tx > 0 && ty > 0)   acc += gq[(tx - 1) * rd + ty - 1] * (fx * fy);
        return acc;
    };

    if (R >= 0) {
        constexpr int Rc = R >= 0 ? R : 0;
        constexpr int Tc = 2 * Rc + 2, TTc = Tc * Tc;
        constexpr int NIT = (QB * TTc + 255) / 256;
        float old[NIT];
        int off[NIT];
        int qi = tid / TTc, t = tid - qi * TTc;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {          // all read-modify-write loads in flight together
            const int qc = min(qi, QB - 1);
            const int ty = t / Tc, tx = t - ty * Tc;
            const int o = s.ytab[qc * Tc + ty] + s.xtab[qc * Tc + tx];
            off[i] = (qi < QB && o >= 0) ? o : -1;
            old[i] = dvol[max(o, 0)];
            t += 256 % TTc; qi += 256 / TTc;
            if (t >= TTc) { t -= TTc; ++qi; }
        }
        qi = tid / TTc; t = tid - qi * TTc;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int ty = t / Tc, tx = t - ty * Tc;
            if (off[i] >= 0) dvol[off[i]] = old[i] + tap_grad(qi, tx, ty);
            t += 256 % TTc; qi += 256 / TTc;
            if (t >= TTc) { t -= TTc; ++qi; }
        }
    } else {
        for (int e = tid; e < QB * TT; e += 256) {
            const int qi = e / TT;
            const int t = e - qi * TT;
            const int ty = t / T, tx = t - ty * T;
            const int o = s.ytab[qi * T + ty] + s.xtab[qi * T + tx];
            if (o >= 0) dvol[o] += tap_grad(qi, tx, ty);
        }
    }
}

template <int R>
__global__ __launch_bounds__(256)
void corr_lookup_bwd_kernel(LookupArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    switch (min((int)blockIdx.y, 4)) {          // level l < 4 has layout kind l (layout.h)
    case 0: lookup_bwd_body<R, 0>(p, smem); break;
    case 1: lookup_bwd_body<R, 1>(p, smem); break;
    case 2: lookup_bwd_body<R, 2>(p, smem); break;
    case 3: lookup_bwd_body<R, 3>(p, smem); break;
    default: lookup_bwd_body<R, 4>(p, smem); break;
    }
}

// d(out)/d(coords): one block = 32 queries, all levels; 8 lanes per query reduce by shuffles.
__global__ __launch_bounds__(256)
void corr_lookup_dcoords_kernel(LookupArgs p)
{
    const int r = p.r, rd = 2 * r + 1, rd2 = rd * rd;
    const int b = blockIdx.z;
    const int q = blockIdx.x * QB + (threadIdx.x >> 3);
    const int sub = threadIdx.x & 7;
    const int N = p.N, C = p.L * rd2;
    float gx = 0.f, gy = 0.f;
    if (q < N) {
        for (int l = 0; l < p.L; ++l) {
            const LevelGeom g = p.lv.g[l];
            const int Hl = g.h, Wl = g.w;
            const float* __restrict__ vol = p.lv.ptr[l] + ((size_t)b * N + q) * g.slab;
            float x, y;
            load_xy(p, b, q, l, x, y);
            int x0, y0; float fx, fy;
            split_coord(x, x0, fx);
            split_coord(y, y0, fy);
            float sx = 0.f, sy = 0.f;
            for (int ch = sub; ch < rd2; ch += 8) {
                const int a = ch / rd, bb = ch - a * rd;
                const int xi = x0 - r + a, yi = y0 - r + bb;
                auto tap = [&](int yy, int xx) -> float {
                    return (xx >= 0 && xx < Wl && yy >= 0 && yy < Hl) ? vol[level_pos(g, yy, xx)] : 0.f;
                };
                const float t00 = tap(yi, xi), t01 = tap(yi, xi + 1);
                const float t10 = tap(yi + 1, xi), t11 = tap(yi + 1, xi + 1);
                const float d = (p.mode == RAFTCORR_MODE_FORK)
                                    ? p.dout[((size_t)b * N + q) * C + l * rd2 + ch]
                                    : p.dout[((size_t)b * C + l * rd2 + ch) * N + q];
                sx += d * ((1.f - fy) * (t01 - t00) + fy * (t11 - t10));
                sy += d * ((1.f - fx) * (t10 - t00) + fx * (t11 - t01));
            }
            const float sc = (p.mode == RAFTCORR_MODE_FORK) ? 1.f : 1.0f / (float)(1 << l);
            gx += sx * sc; gy += sy * sc;
        }
    }
    for (int m = 1; m < 8; m <<= 1) { gx += __shfl_xor(gx, m); gy += __shfl_xor(gy, m); }
    if (q < N && sub == 0) {
        if (p.mode == RAFTCORR_MODE_FORK) {
            float* c = p.dcoords + ((size_t)b * N + q) * 2;
            c[0] = gx; c[1] = gy;
        } else {
            float* c = p.dcoords + (size_t)b * 2 * N + q;
            c[0] = gx; c[N] = gy;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Lookup fused with its only consumer in RAFT: the motion encoder's 1x1 convolution
// convc1 (L*(2r+1)^2 -> Cout) + bias + ReLU (core/update.py:83,90).  The (B, L*rd2, H, W) lookup
// result is never written: a 1024-thread block runs the four levels' lookups side by side (one
// 256-thread group per level, same code as the stand-alone kernel), leaves the 32 x Cin result tile
// in LDS and multiplies it with the weight on the matrix cores:
//   out[c][q] = relu( sum_k W[c][k] * X[k][q] + bias[c] ),  rows = 32 channels, columns = the 32 queries.
// fp32-grade arithmetic at 16-bit matrix-core speed: X and W are split into three bf16 pieces each
// and the six products of weight >= 2^-16 are accumulated in fp32 (as the bf16x3 build,
// build_pyramid.hip): 6 x v_mfma_f32_32x32x16_bf16 per 16 k instead of 8 x v_mfma_f32_32x32x2_f32
// (an fp32-MFMA version of this kernel spent 20 of its 32 us in the product).
// The weight is pre-split and pre-packed once (raftcorr_conv1x1_pack_weight) into A-fragment order,
// so every fragment is one coalesced 16-byte load per lane.
// ---------------------------------------------------------------------------------------------
struct ConvArgs {
    const uint4* wpacked;   // [Cout/32 tiles][ceil(Cin/16) k-steps][3 pieces][64 lanes] x 8 bf16
    const float* bias;      // [Cout] or null
    float* out;             // (B, Cout, N)
    int Cin, Cout, relu;
};

typedef float f32x16l __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8l __attribute__((ext_vector_type(8)));

__host__ __device__ inline int conv_ksteps(int Cin) { return (Cin + 15) / 16; }
__host__ __device__ inline int conv_xkp(int Cin) { return conv_ksteps(Cin) * 16 + 8; }   // bf16 elements; 16-byte rows whose
                                                                                         // b128 reads are bank-conflict-free

__global__ __launch_bounds__(256)
void conv1x1_pack_weight_kernel(const float* __restrict__ w, uint4* __restrict__ packed, int Cout, int Cin)
{
    const int ks = conv_ksteps(Cin), tiles = (Cout + 31) / 32;
    const size_t total = (size_t)tiles * ks * 64;                    // one thread = one lane's fragment, all 3 pieces
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int lane = (int)(e & 63);
        const size_t ts = e >> 6;
        const int s = (int)(ts % ks), mt = (int)(ts / ks);
        const int c = mt * 32 + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (c < Cout && k0 + j < Cin) ? w[(size_t)c * Cin + k0 + j] : 0.f;
#pragma unroll
        for (int piece = 0; piece < 3; ++piece) {
            unsigned short h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { h[j] = bf16_rne(x[j]); x[j] -= __uint_as_float((unsigned)h[j] << 16); }
            uint4 v;
            v.x = h[0] | ((unsigned)h[1] << 16); v.y = h[2] | ((unsigned)h[3] << 16);
            v.z = h[4] | ((unsigned)h[5] << 16); v.w = h[6] | ((unsigned)h[7] << 16);
            packed[((size_t)(mt * ks + s) * 3 + piece) * 64 + lane] = v;
        }
    }
}

template <int R>
__global__ __launch_bounds__(1024)
void corr_lookup_conv1x1_kernel(LookupArgs p, ConvArgs cv)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef BlockGeom<R> G;
    constexpr int GF = G::SMEM;                                      // floats per level group
    constexpr int GFA = (GF + 3) & ~3;
    // LDS: region A = the four level groups' tables and tap tiles, later the bf16 pieces of X;
    //      region B = X in fp32 as the lookups leave it ([k][query]), later the partial sums.
    const int xkp = conv_xkp(cv.Cin), ks = conv_ksteps(cv.Cin);
    const int tiles = (cv.Cout + 31) / 32;
    float* X32 = smem + 4 * GFA;                                     // [Cin][XS]
    unsigned short* X = (unsigned short*)smem;                       // [3 pieces][QB][xkp] bf16
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int b = blockIdx.z, q0 = blockIdx.x * QB;
    float* gs = smem + grp * GFA;
    if (grp < p.L) {
        switch (grp) {                                               // level l < 4 has layout kind l
        case 0: lookup_fwd_blocks<R, 0, 256>(p, gs, 0, b, q0, tid, X32); break;
        case 1: lookup_fwd_blocks<R, 1, 256>(p, gs, 1, b, q0, tid, X32); break;
        case 2: lookup_fwd_blocks<R, 2, 256>(p, gs, 2, b, q0, tid, X32); break;
        default: lookup_fwd_blocks<R, 3, 256>(p, gs, 3, b, q0, tid, X32); break;
        }
    } else {
        __syncthreads();                                             // the two barriers of lookup_fwd_blocks
        __syncthreads();
    }
    __syncthreads();
    // X (fp32, [k][query]) -> three bf16 pieces in B-fragment order ([piece][query][k], k contiguous);
    // lanes run along k on both sides: stride-33 reads and consecutive 2-byte writes, conflict-free
    for (int e = threadIdx.x; e < QB * xkp; e += 1024) {
        const int qi = e / xkp, k = e - qi * xkp;
        float o = k < cv.Cin ? X32[k * XS + qi] : 0.f;               // K padding is zero
#pragma unroll
        for (int piece = 0; piece < 3; ++piece) {
            const unsigned short h = bf16_rne(o);
            X[(size_t)(piece * QB + qi) * xkp + k] = h;
            o -= __uint_as_float((unsigned)h << 16);
        }
    }
    __syncthreads();

    // GEMM: (channel tile, K split) pairs over the 16 waves; partial products meet in LDS (the level
    // groups' tiles are free now), then every thread finishes a few outputs: bias, ReLU, coalesced store
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int nsplit = max(1, min(16 / tiles, ks));                  // tiles * nsplit <= 16 slabs (host sizes region B for it)
    const int kper = (ks + nsplit - 1) / nsplit;
    // partial sums [nsplit][tiles * 32][XS]: behind the bf16 pieces (the level groups' tiles and X32 are dead by now)
    float* red = smem + ((3 * QB * xkp / 2 + 3) & ~3);
    for (int pr = wave; pr < tiles * nsplit; pr += 16) {
        const int mt = pr % tiles, sp = pr / tiles;
        const int s0 = sp * kper, s1 = min(ks, s0 + kper);
        const uint4* __restrict__ wp = cv.wpacked + (size_t)mt * ks * 3 * 64 + lane;
        // B fragment (k-step s, piece c): 8 bf16 at X[(c * QB + query) * xkp + 16 s + 8 kh]
        const uint4* xb = (const uint4*)(X + (size_t)l31 * xkp + 8 * kh);
        const int xpiece = QB * xkp / 8;                             // uint4 stride between pieces
        f32x16l acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // weight fragments: a ring of PD k-steps in flight (one k-step of MFMAs is shorter than an L2 round trip)
        constexpr int PD = 4;
        uint4 a[PD][3];
#pragma unroll
        for (int u = 0; u < PD; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) a[u][c] = wp[(size_t)(min(s0 + u, ks - 1) * 3 + c) * 64];
        for (int sb = s0; sb < s1; sb += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int s = sb + u;
                if (s < s1) {                                        // wave-uniform
                    uint4 xf[3], af[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { xf[c] = xb[c * xpiece + 2 * s]; af[c] = a[u][c]; }
#pragma unroll
                    for (int c = 0; c < 3; ++c) a[u][c] = wp[(size_t)(min(s + PD, ks - 1) * 3 + c) * 64];
                    // small products first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                    const int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8l, af[pa[t]]),
                                                                      __builtin_bit_cast(bf16x8l, xf[pb[t]]), acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            red[((sp * tiles + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh) * XS + l31] = acc[i];
    }
    __syncthreads();
    {
        const int qi = threadIdx.x & 31, q = q0 + qi;
        if (q < p.N)
            for (int c = threadIdx.x >> 5; c < cv.Cout; c += 32) {
                float v = cv.bias ? cv.bias[c] : 0.f;
                for (int sp = 0; sp < nsplit; ++sp) v += red[(sp * tiles * 32 + c) * XS + qi];   // fixed order
                if (cv.relu) v = fmaxf(v, 0.f);
                cv.out[((size_t)b * cv.Cout + c) * p.N + q] = v;
            }
    }
}

// host mirror of BlockGeom<R>::SMEM
static size_t fwd_blocks_smem_floats(int r)
{
    const int T = 2 * r + 2, NB = T / 2 + 1, TW = 2 * NB, rd = 2 * r + 1;
    return (size_t)QB * ((TW * TW) | 1) + 2 * QB * NB + 2 * QB + 6 * (size_t)QB * rd;
}

size_t conv1x1_packed_bytes(int Cout, int Cin)
{
    return sizeof(uint4) * (size_t)((Cout + 31) / 32) * conv_ksteps(Cin) * 3 * 64;
}

int launch_conv1x1_pack_weight(const float* w, int Cout, int Cin, float* packed, hipStream_t s)
{
    const size_t total = conv1x1_packed_bytes(Cout, Cin) / sizeof(float);
    hipLaunchKernelGGL(conv1x1_pack_weight_kernel, dim3((unsigned)std::min<size_t>((total / 48 + 255) / 256 + 1, 1024)), dim3(256),
                       0, s, w, (uint4*)packed, Cout, Cin);
    return check_launch("conv1x1_pack_weight_kernel");
}

int launch_lookup_conv1x1(const LevelSet& lv, int B, int H, int W, int L, int r, const float* coords,
                          const float* wpacked, const float* bias, int Cout, int relu, float* out, hipStream_t s)
{
    if (L > 4) return fail(RAFTCORR_EUNSUPPORTED, "lookup_conv1x1: more than 4 levels");
    if (r != 4 && r != 3) return fail(RAFTCORR_EUNSUPPORTED, "lookup_conv1x1: radius %d (3 and 4 are built)", r);
    if (Cout > 384) return fail(RAFTCORR_EUNSUPPORTED, "lookup_conv1x1: Cout = %d > 384", Cout);
    LookupArgs a = {};
    a.lv = lv; a.coords = coords; a.out = nullptr;
    a.B = B; a.N = H * W; a.L = L; a.r = r; a.mode = RAFTCORR_MODE_CANONICAL;
    set_reciprocals(a);
    const int rd = 2 * r + 1;
    ConvArgs c = {(const uint4*)wpacked, bias, out, L * rd * rd, Cout, relu};
    const size_t gf = (fwd_blocks_smem_floats(r) + 3) & ~(size_t)3;
    const int tiles = (Cout + 31) / 32;
    // LDS: [4 level groups | X32 (fp32 lookup results)] during the lookups, then
    //      [bf16 pieces of X | partial sums] during the product
    const size_t pieces = sizeof(float) * ((3 * QB * (size_t)conv_xkp(c.Cin) / 2 + 3) & ~(size_t)3);
    const size_t red = sizeof(float) * XS * (size_t)std::max(tiles, 16 / std::max(tiles, 1) * tiles) * 32;
    // the kernel places X32 at 4 * gf floats: the pieces (written while X32 is read) must end before it
    if (pieces > sizeof(float) * 4 * gf) return fail(RAFTCORR_EUNSUPPORTED, "lookup_conv1x1: Cin = %d too large", c.Cin);
    const size_t sm = std::max(sizeof(float) * (4 * gf + (size_t)XS * c.Cin), pieces + red);
    if (sm > 160 * 1024) return fail(RAFTCORR_EUNSUPPORTED, "lookup_conv1x1: needs %zu bytes of LDS", sm);
    dim3 grid((a.N + QB - 1) / QB, 1, B), block(1024);
    // more than 64 KiB of dynamic LDS needs opting in, once per kernel and device context
    static thread_local int allowed_tab[64][2] = {};            // per device (the attribute is per device)
    int devid = 0;
    hipError_t e = hipGetDevice(&devid);
    int* allowed = allowed_tab[devid & 63];
    const int which = r == 4 ? 0 : 1;
    if (e == hipSuccess && allowed[which] < (int)sm) {
        e = r == 4 ? hipFuncSetAttribute((const void*)corr_lookup_conv1x1_kernel<4>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                   : hipFuncSetAttribute((const void*)corr_lookup_conv1x1_kernel<3>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) allowed[which] = 160 * 1024;
    }
    if (e == hipSuccess) {
        if (r == 4) hipLaunchKernelGGL(corr_lookup_conv1x1_kernel<4>, grid, block, sm, s, a, c);
        else        hipLaunchKernelGGL(corr_lookup_conv1x1_kernel<3>, grid, block, sm, s, a, c);
    }
    if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "lookup_conv1x1: %s", hipGetErrorString(e));
    return check_launch("corr_lookup_conv1x1_kernel");
}

static size_t lookup_smem(int r, bool bwd)
{
    const int T = 2 * r + 2, rd = 2 * r + 1;
    const int stride = bwd ? ((rd * rd) | 1) : ((T * T) | 1);
    size_t fl = (size_t)QB * stride + 2 * QB + 2 * QB * T;
    if (!bwd) {                                   // block-load path (compile-time radius)
        fl += 4 * (size_t)QB * rd;                // per-sample (tap, fraction) tables of the generic path
        const size_t f4 = fwd_blocks_smem_floats(r);
        if (f4 > fl) fl = f4;
    }
    return sizeof(float) * fl;
}

int launch_lookup_fwd(const LevelSet& lv, int B, int H, int W, int L, int r, const float* coords,
                      int mode, float* out, hipStream_t s, int vol)
{
    LookupArgs a = {};
    a.lv = lv; a.coords = coords; a.out = out;
    a.B = B; a.N = H * W; a.L = L; a.r = r; a.mode = mode;
    set_reciprocals(a);
    const size_t sm = lookup_smem(r, false);
    if (sm > 64 * 1024) return fail(RAFTCORR_EUNSUPPORTED, "lookup_fwd: radius %d too large", r);
    dim3 grid((a.N + QB - 1) / QB, L, B);
#define RC_LAUNCH_FWD(VOL)                                                                                          \
    do {                                                                                                            \
        if (r == 4)      hipLaunchKernelGGL((corr_lookup_fwd_kernel<4, VOL>), grid, dim3(FwdThreads<4>::NT), sm, s, a); \
        else if (r == 3) hipLaunchKernelGGL((corr_lookup_fwd_kernel<3, VOL>), grid, dim3(FwdThreads<3>::NT), sm, s, a); \
        else             hipLaunchKernelGGL((corr_lookup_fwd_kernel<-1, VOL>), grid, dim3(256), sm, s, a);              \
    } while (0)
    if (vol == RAFTCORR_VOL_FP16)      RC_LAUNCH_FWD(1);
    else if (vol == RAFTCORR_VOL_BF16) RC_LAUNCH_FWD(2);
    else                               RC_LAUNCH_FWD(0);
#undef RC_LAUNCH_FWD
    return check_launch("corr_lookup_fwd_kernel");
}

int launch_lookup_bwd(const LevelSet& lv, const LevelSet& dlv, bool have_dlevels, int B, int H, int W,
                      int L, int r, const float* coords, int mode, const float* dout,
                      float* dcoords, hipStream_t s)
{
    LookupArgs a = {};
    a.lv = lv; a.dlv = dlv; a.coords = coords; a.dout = dout; a.dcoords = dcoords;
    a.B = B; a.N = H * W; a.L = L; a.r = r; a.mode = mode;
    dim3 grid((a.N + QB - 1) / QB, L, B), block(256);
    int rc = RAFTCORR_OK;
    if (have_dlevels) {
        const size_t sm = lookup_smem(r, true);
        if (sm > 64 * 1024) return fail(RAFTCORR_EUNSUPPORTED, "lookup_bwd: radius %d too large", r);
        if (r == 4)      hipLaunchKernelGGL(corr_lookup_bwd_kernel<4>, grid, block, sm, s, a);
        else if (r == 3) hipLaunchKernelGGL(corr_lookup_bwd_kernel<3>, grid, block, sm, s, a);
        else             hipLaunchKernelGGL(corr_lookup_bwd_kernel<-1>, grid, block, sm, s, a);
        rc = check_launch("corr_lookup_bwd_kernel");
        if (rc) return rc;
    }
    if (dcoords) {
        dim3 g2((a.N + QB - 1) / QB, 1, B);
        hipLaunchKernelGGL(corr_lookup_dcoords_kernel, g2, block, 0, s, a);
        rc = check_launch("corr_lookup_dcoords_kernel");
    }
    return rc;
}

}  // namespace raftcorr
