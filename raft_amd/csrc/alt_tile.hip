// On-the-fly ("alternate") correlation -- the memory-lean path that never materialises the
// all-pairs volume -- as TILE GEMMS on the matrix cores.  Forward and backward; one level with
// the reference's pybind calling convention (alt_cuda_corr/correlation_kernel.cu:19-119 forward,
// :123-256 backward, launched from :260-324) and all levels fused (AlternateCorrBlock.__call__,
// core/corr.py:75-92: per-level permute + kernel call + stack + 1/sqrt(C)).
//
// The reference kernel runs 32-thread blocks (<1 wave per CU on this part), re-reads fmap2
// through a 32-channel shared-memory chunk with two barriers per tap and does 3200 global
// read-modify-writes per pixel.  Here:
//
// Neighbouring queries look at overlapping windows: a 4x8 tile of queries whose flow is locally
// smooth touches, at level l, a bounding box of only ~(8/2^l + 2r + 2) x (4/2^l + 2r + 2) targets,
// while its 32 windows hold 32 x (2r+2)^2 taps.  So a workgroup
//   1. finds the bounding box of its 32 windows (clipped to the level; any flow field is legal:
//      a box larger than UC targets is walked in chunks of UC),
//   2. computes the dense product  S[q][u] = <fmap1[:, q], fmap2_l[:, u]>  of its 32 queries with
//      the UC box targets on MFMA (fp32 32x32x2, exact fp32 products/accumulation),
//      operands staged channel-chunk by channel-chunk through LDS from the NHWC maps (a target's
//      32-channel chunk is exactly one 128-byte line, so no fetched byte is wasted),
//   3. lets every query pick its (2r+2)^2 taps out of S in LDS and blend them bilinearly.
// fmap2 is read once per tile instead of once per query, and nothing is reduced across lanes.
//
// Backward runs the same tiling with the tap-gradient matrix G[q][u] (the 2x2 box filter of
// d(out), scattered to box positions) as the shared operand of two GEMMs per channel chunk:
//   d(fmap1)[c][q] = sum_u fmap2[c][u] G[q][u]      d(fmap2)[c][u] = sum_q fmap1[c][q] G[q][u]
// d(fmap2) of a tile is added to memory with one float atomic per (channel, box target) instead
// of one per (channel, query, tap) as in the reference (:239-250), 128 contiguous bytes per
// wave-instruction; d(fmap1) belongs to the workgroup that owns the query tile and is summed in LDS (no atomics).
//
// Workgroups are numbered so that each XCD (workgroup id mod 8) owns a band of tile rows: the
// band's part of fmap2 stays in that XCD's L2 while neighbouring tiles re-read it.
#include "common.h"
#include <limits.h>
#include <algorithm>

namespace raftcorr {

namespace {

constexpr int TQW = 8, TQH = 4, TQ = TQW * TQH;   // query tile
constexpr int UC = 256;                           // box targets per chunk (4 waves x 2 MFMA column tiles)
constexpr int KC = 32;                            // channels per staged chunk
constexpr int LDU = UC + 1;                       // odd strides: row- and column-indexed reads conflict-free
constexpr int LDQ = TQ + 1;

typedef float v16f __attribute__((ext_vector_type(16)));

struct AltTileArgs {
    const float* f1;            // (B, H, W, C)
    LevelSet f2;                // level l: (B, H_l, W_l, C)
    LevelSet df2;               // bwd: accumulated with atomics
    const float* coords;        // planar: (B, 2, H, W);  else (B, Nc, H, W, 2)
    const float* dout;          // bwd: (B*Nc, L*rd2, H, W)
    float* out;                 // fwd: (B*Nc, L*rd2, H, W)
    float* df1;                 // bwd: (B, H, W, C), accumulated with atomics (zeroed by the launcher)
    int B, Nc, C, H, W, L, r;   // Nc coordinate sets per image (the reference's N, correlation_kernel.cu:264)
    int planar;                 // coords layout, see above (planar needs Nc == 1)
    int level_scale;            // 1: level l samples at coords / 2**l (corr.py:82); 0: coords as given
    int band;                   // tiles per XCD band
    int vec;                    // C % 4 == 0 and 16-byte aligned maps: float4 staging (host-side dispatch)
    float out_div;
};

struct Tile {
    int b, n, l, ty0, tx0, Hl, Wl, HWl;   // n: coordinate set
    bool live;
    int xmin, ymin, bw, U;
};

struct TileShared {
    int wx0[TQ], wy0[TQ];       // window origin (x0 - r, y0 - r) per query
    float fx[TQ], fy[TQ];
    int box[4];
    int toff[UC];               // offset (pixels) of the chunk's box targets in the level, -1 beyond the box
};

// queries' window origins + the bounding box of their in-level parts
// nset >= 0: the grid has no coordinate-set factor and this call sets up set `nset` (backward: one workgroup walks
// the sets of its query tile in order)
__device__ __forceinline__ Tile tile_setup(const AltTileArgs& p, TileShared& sh, int nset = -1)
{
    Tile t;
    const int tiles_x = (p.W + TQW - 1) / TQW, tiles = tiles_x * ((p.H + TQH - 1) / TQH);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int j = idx % p.band, bl = idx / p.band;
    const int tile = xcd * p.band + j;
    t.l = bl % p.L;
    const int bn = bl / p.L;
    if (nset >= 0) { t.b = bn; t.n = nset; }
    else { t.b = bn / p.Nc; t.n = bn - t.b * p.Nc; }
    t.live = tile < tiles;
    if (!t.live) return t;                                       // whole workgroup
    t.ty0 = (tile / tiles_x) * TQH;
    t.tx0 = (tile % tiles_x) * TQW;
    t.Hl = p.f2.g[t.l].h; t.Wl = p.f2.g[t.l].w; t.HWl = t.Hl * t.Wl;
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int qi = tid & 31, gy = t.ty0 + (qi >> 3), gx = t.tx0 + (qi & 7);
        const bool valid = tid < 32 && gy < p.H && gx < p.W;
        const int T = 2 * p.r + 2;
        int wx0 = -(1 << 24), wy0 = -(1 << 24);
        float fx = 0.f, fy = 0.f;
        if (valid) {
            const size_t P = (size_t)p.H * p.W, q = (size_t)gy * p.W + gx;
            float x, y;
            if (p.planar) { const float* c = p.coords + (size_t)t.b * 2 * P + q; x = c[0]; y = c[P]; }
            else { const float* c = p.coords + (((size_t)t.b * p.Nc + t.n) * P + q) * 2; x = c[0]; y = c[1]; }
            const float s = p.level_scale ? 1.0f / (float)(1 << t.l) : 1.0f;
            int x0, y0;
            split_coord(x * s, x0, fx);
            split_coord(y * s, y0, fy);
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
        if (tid < 32) { sh.wx0[qi] = wx0; sh.wy0[qi] = wy0; sh.fx[qi] = fx; sh.fy[qi] = fy; }
        if (tid == 0) { sh.box[0] = mnx; sh.box[1] = mxx; sh.box[2] = mny; sh.box[3] = mxy; }
    }
    __syncthreads();
    const int mnx = sh.box[0], mxx = sh.box[1], mny = sh.box[2], mxy = sh.box[3];
    if (mnx > mxx || mny > mxy) { t.xmin = 0; t.ymin = 0; t.bw = 1; t.U = 0; }
    else { t.xmin = mnx; t.ymin = mny; t.bw = mxx - mnx + 1; t.U = t.bw * (mxy - mny + 1); }
    return t;
}

// offset of box target u inside a level plane, or -1 beyond the box
__device__ __forceinline__ int box_offset(const Tile& t, int u)
{
    if (u >= t.U) return -1;
    const int uy = u / t.bw, ux = u - uy * t.bw;
    return (t.ymin + uy) * t.Wl + t.xmin + ux;
}

// chunk-relative box index of level cell (y, x), or -1 when outside the box / the chunk
__device__ __forceinline__ int box_index(const Tile& t, int bh, int y, int x, int ubase)
{
    const int ry = y - t.ymin, rx = x - t.xmin;
    const int u = ry * t.bw + rx - ubase;
    const bool ok = (unsigned)rx < (unsigned)t.bw && (unsigned)ry < (unsigned)bh && (unsigned)u < (unsigned)UC;
    return ok ? u : -1;
}

struct Stage {                   // one channel chunk of both operands, in flight in registers
    float4 rb[UC / 32];          // fmap2: 8 box targets x 4 channels
    float4 ra;                   // fmap1: 1 query x 4 channels
};

// Branch-free by construction (all loads of a stage must be in flight together): buffer loads,
// where an out-of-range offset returns zero instead of faulting, so "no such target / channel"
// is just the offset OOB.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;        // maps are < 2 GiB per image (checked on the host)

template <bool VEC>
__device__ __forceinline__ float4 load_quad(__amdgpu_buffer_rsrc_t rsrc, int off, int c, int C)
{
    const unsigned base = ((unsigned)off * (unsigned)C + (unsigned)c) * 4u;
    float4 v;
    if (VEC) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (off < 0 || c >= C) ? OOB : base, 0, 0);
        v = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    } else {
        v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (off < 0 || c >= C) ? OOB : base, 0, 0));
        v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (off < 0 || c + 1 >= C) ? OOB : base + 4, 0, 0));
        v.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (off < 0 || c + 2 >= C) ? OOB : base + 8, 0, 0));
        v.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (off < 0 || c + 3 >= C) ? OOB : base + 12, 0, 0));
    }
    return v;
}

// thread (kq = tid & 7, ut = tid >> 3): channels c0 + 4kq .. +3 of box targets ut + 32 i and of
// query ut -- eight lanes read one 128-byte line
template <bool VEC>
__device__ __forceinline__ void stage_load(Stage& st, __amdgpu_buffer_rsrc_t f1b, __amdgpu_buffer_rsrc_t f2b,
                                           int c0, int C, const int* toff, int q_off_stage)
{
    const int tid = threadIdx.x, c = c0 + 4 * (tid & 7), ut = tid >> 3;
#pragma unroll
    for (int i = 0; i < UC / 32; ++i) st.rb[i] = load_quad<VEC>(f2b, toff[ut + 32 * i], c, C);
    st.ra = load_quad<VEC>(f1b, q_off_stage, c, C);
}

// transposed into [channel][target] / [channel][query]; bank = (4kq + j + ut) mod 32: conflict-free
__device__ __forceinline__ void stage_store(const Stage& st, float* F2s, float* F1s)
{
    const int tid = threadIdx.x, k = 4 * (tid & 7), ut = tid >> 3;
#pragma unroll
    for (int i = 0; i < UC / 32; ++i) {
        float* d = F2s + k * LDU + ut + 32 * i;
        d[0] = st.rb[i].x; d[LDU] = st.rb[i].y; d[2 * LDU] = st.rb[i].z; d[3 * LDU] = st.rb[i].w;
    }
    float* d = F1s + k * LDQ + ut;
    d[0] = st.ra.x; d[LDQ] = st.ra.y; d[2 * LDQ] = st.ra.z; d[3 * LDQ] = st.ra.w;
}

__device__ __forceinline__ int query_offset(const AltTileArgs& p, const Tile& t, int qi)
{
    const int gy = t.ty0 + (qi >> 3), gx = t.tx0 + (qi & 7);
    return (gy < p.H && gx < p.W) ? gy * p.W + gx : -1;
}

// ---------------------------------------------------------------------------------------------
// forward.  grid: 8 XCD bands x (tiles per band x L x B), block 256.
// NOUT = ceil(32 * rd2 / 256) outputs per thread.
template <int NOUT, bool VEC>
__global__ __launch_bounds__(256, NOUT <= 11 ? 3 : 2)
void alt_tile_fwd_kernel(AltTileArgs p)
{
    __shared__ float F2s[KC * LDU];          // staged fmap2 chunk [k][u]; later S[q][u]
    __shared__ float F1s[KC * LDQ];          // staged fmap1 chunk [k][q]
    __shared__ TileShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int r = p.r, rd = 2 * r + 1, rd2 = rd * rd;
    const Tile t = tile_setup(p, sh);
    if (!t.live) return;
    const int bh = t.U / t.bw;
    const size_t P = (size_t)p.H * p.W;
    const __amdgpu_buffer_rsrc_t f1b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.f1) + (size_t)t.b * P * p.C, 0, (int)(P * p.C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t f2b = __builtin_amdgcn_make_buffer_rsrc(
        p.f2.ptr[t.l] + (size_t)t.b * t.HWl * p.C, 0, t.HWl * p.C * 4, 0x00020000);
    const int q_off = query_offset(p, t, tid & 31);          // the query this thread writes outputs of
    const int q_stage = query_offset(p, t, tid >> 3);        // the query this thread stages
    float o[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) o[i] = 0.f;

    const int nk = (p.C + KC - 1) / KC;
    for (int ubase = 0; ubase < t.U; ubase += UC) {
        __syncthreads();
        sh.toff[tid] = box_offset(t, ubase + tid);
        __syncthreads();
        const bool live0 = ubase + wave * 64 < t.U, live1 = ubase + wave * 64 + 32 < t.U;   // wave-uniform
        v16f acc0 = {0}, acc1 = {0};
        Stage st;
        stage_load<VEC>(st, f1b, f2b, 0, p.C, sh.toff, q_stage);
        for (int kc = 0; kc < nk; ++kc) {
            __syncthreads();
            stage_store(st, F2s, F1s);
            __syncthreads();
            if (kc + 1 < nk) stage_load<VEC>(st, f1b, f2b, (kc + 1) * KC, p.C, sh.toff, q_stage);
            if (live1) {
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) {
                    const float a = F1s[(2 * s + kh) * LDQ + l31];
                    const float b0 = F2s[(2 * s + kh) * LDU + wave * 64 + l31];
                    const float b1 = F2s[(2 * s + kh) * LDU + wave * 64 + 32 + l31];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
                }
            } else if (live0) {
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) {
                    const float a = F1s[(2 * s + kh) * LDQ + l31];
                    const float b0 = F2s[(2 * s + kh) * LDU + wave * 64 + l31];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                }
            }
        }
        __syncthreads();
        float* S = F2s;                                  // S[q][u], stride LDU
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
            S[row * LDU + wave * 64 + l31] = acc0[i];
            S[row * LDU + wave * 64 + 32 + l31] = acc1[i];
        }
        __syncthreads();
        const int qi = tid & 31;
        const int wx = sh.wx0[qi], wy = sh.wy0[qi];
        const float fx = sh.fx[qi], fy = sh.fy[qi];
        const float* Sq = S + qi * LDU;
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int ch = (tid >> 5) + 8 * i;
            if (ch < rd2) {
                const int ai = ch / rd, bi = ch - ai * rd;           // ai <-> x offset (slow), bi <-> y offset
                const int x = wx + ai, y = wy + bi;
                const int u00 = box_index(t, bh, y, x, ubase), u01 = box_index(t, bh, y, x + 1, ubase);
                const int u10 = box_index(t, bh, y + 1, x, ubase), u11 = box_index(t, bh, y + 1, x + 1, ubase);
                const float v00 = u00 >= 0 ? Sq[u00] : 0.f, v01 = u01 >= 0 ? Sq[u01] : 0.f;
                const float v10 = u10 >= 0 ? Sq[u10] : 0.f, v11 = u11 >= 0 ? Sq[u11] : 0.f;
                o[i] += v00 * ((1.f - fx) * (1.f - fy)) + v01 * (fx * (1.f - fy)) +
                        v10 * ((1.f - fx) * fy) + v11 * (fx * fy);
            }
        }
    }
    if (q_off >= 0) {
        float* __restrict__ out = p.out + (((size_t)t.b * p.Nc + t.n) * p.L + t.l) * rd2 * P + q_off;
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int ch = (tid >> 5) + 8 * i;
            if (ch < rd2) out[(size_t)ch * P] = o[i] / p.out_div;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward.  Grid: 8 XCD bands x (tiles per band x L x B) -- no coordinate-set factor: a workgroup owns its 32 queries'
// d(fmap1) rows outright.  They are summed in LDS (acc1, dynamic: [TQ][C]) over the coordinate sets and the box chunks
// in program order, the four waves' partial products through one LDS plane each added in wave order, and written
// once with plain stores: d(fmap1) needs no zero fill, no atomics, and has the same bits on every run
// (ACC1 = false, C too large for the LDS accumulator: float atomics into zero-filled memory, as round 1-2 did).
// d(fmap2) is a scatter across tiles and keeps its float atomics (the all-level path alt_bwd.hip is the
// reproducible one).
template <bool VEC, bool ACC1>
__global__ __launch_bounds__(256)
void alt_tile_bwd_kernel(AltTileArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float acc1[];    // ACC1: [TQ][C]
    __shared__ float F2s[KC * LDU];          // staged fmap2 chunk [c][u]; before the chunk loop: d(out) tile [ch][q]
    __shared__ float F1s[KC * LDQ];          // staged fmap1 chunk [c][q]
    __shared__ float G[TQ * LDU];            // tap gradients [q][u]
    __shared__ float red[4][TQ * LDQ];       // d(fmap1) chunk [wave][q][c]
    __shared__ TileShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int r = p.r, T = 2 * r + 2, TT = T * T, rd = 2 * r + 1, rd2 = rd * rd;
    const size_t P = (size_t)p.H * p.W;
    const float inv_div = 1.0f / p.out_div;
    const int nk = (p.C + KC - 1) / KC;
    if (ACC1)
        for (int e = tid; e < TQ * p.C; e += 256) acc1[e] = 0.f;
    int q_stage = -1;
    float* __restrict__ df1b = nullptr;

    for (int nset = 0; nset < p.Nc; ++nset) {
        __syncthreads();                                            // the previous set's windows are no longer read
        const Tile t = tile_setup(p, sh, nset);
        if (!t.live) return;                                        // whole workgroup, every set alike
        const int bh = t.U / t.bw;
        const __amdgpu_buffer_rsrc_t f1b = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.f1) + (size_t)t.b * P * p.C, 0, (int)(P * p.C * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t f2b = __builtin_amdgcn_make_buffer_rsrc(
            p.f2.ptr[t.l] + (size_t)t.b * t.HWl * p.C, 0, t.HWl * p.C * 4, 0x00020000);
        df1b = p.df1 + (size_t)t.b * P * p.C;
        float* __restrict__ df2b = p.df2.ptr[t.l] + (size_t)t.b * t.HWl * p.C;
        const int q_off = query_offset(p, t, tid & 31);
        q_stage = query_offset(p, t, tid >> 3);
        const float* __restrict__ dout = p.dout + (((size_t)t.b * p.Nc + t.n) * p.L + t.l) * rd2 * P;

        for (int ubase = 0; ubase < t.U; ubase += UC) {
            __syncthreads();
            float* Ds = F2s;                                            // [ch][q], stride 32
            for (int e = tid; e < rd2 * TQ; e += 256)
                Ds[e] = q_off >= 0 ? dout[(size_t)(e >> 5) * P + q_off] : 0.f;   // (e & 31) == (tid & 31)
            for (int e = tid; e < TQ * LDU; e += 256) G[e] = 0.f;
            sh.toff[tid] = box_offset(t, ubase + tid);
            __syncthreads();
            for (int e = tid; e < TT * TQ; e += 256) {
                const int qi = e & 31, tp = e >> 5;
                const int ty = tp / T, tx = tp - ty * T;
                const int u = box_index(t, bh, sh.wy0[qi] + ty, sh.wx0[qi] + tx, ubase);
                if (u >= 0) {
                    const float fx = sh.fx[qi], fy = sh.fy[qi];
                    float g = 0.f;
                    if (tx < rd && ty < rd) g += Ds[(tx * rd + ty) * TQ + qi] * ((1.f - fx) * (1.f - fy));
                    if (tx > 0 && ty < rd)  g += Ds[((tx - 1) * rd + ty) * TQ + qi] * (fx * (1.f - fy));
                    if (tx < rd && ty > 0)  g += Ds[(tx * rd + ty - 1) * TQ + qi] * ((1.f - fx) * fy);
                    if (tx > 0 && ty > 0)   g += Ds[((tx - 1) * rd + ty - 1) * TQ + qi] * (fx * fy);
                    G[qi * LDU + u] = g * inv_div;
                }
            }
            const int ulive = min(t.U - ubase - wave * 64, 64);        // box targets of this wave, wave-uniform
            Stage st;
            stage_load<VEC>(st, f1b, f2b, 0, p.C, sh.toff, q_stage);
            for (int kc = 0; kc < nk; ++kc) {
                const int c0 = kc * KC;
                __syncthreads();
                stage_store(st, F2s, F1s);
                __syncthreads();
                if (kc + 1 < nk) stage_load<VEC>(st, f1b, f2b, c0 + KC, p.C, sh.toff, q_stage);
                v16f acc = {0};
                if (ulive > 0) {
                    // d(fmap1)[q][c] += sum over this wave's box targets   (rows q, lanes c)
                    const int ns = min((ulive + 1) >> 1, 32);
                    for (int s = 0; s < ns; ++s) {
                        const int u = wave * 64 + 2 * s + kh;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(G[l31 * LDU + u], F2s[l31 * LDU + u], acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)                          // (zeros from a wave without box targets)
                    red[wave][((i & 3) + 8 * (i >> 2) + 4 * kh) * LDQ + l31] = acc[i];
                if (ulive > 0) {
                    // d(fmap2)[u][c] for this wave's box targets, all 32 queries   (rows u, lanes c)
                    v16f acc0 = {0}, acc1v = {0};
#pragma unroll
                    for (int s = 0; s < TQ / 2; ++s) {
                        const float b = F1s[l31 * LDQ + 2 * s + kh];
                        const float a0 = G[(2 * s + kh) * LDU + wave * 64 + l31];
                        const float a1 = G[(2 * s + kh) * LDU + wave * 64 + 32 + l31];
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
                        acc1v = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1v, 0, 0, 0);
                    }
                    if (c0 + l31 < p.C) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
                            const int o0 = sh.toff[wave * 64 + row], o1 = sh.toff[wave * 64 + 32 + row];
                            if (o0 >= 0) atomicAdd(df2b + (size_t)o0 * p.C + c0 + l31, acc0[i]);
                            if (o1 >= 0) atomicAdd(df2b + (size_t)o1 * p.C + c0 + l31, acc1v[i]);
                        }
                    }
                }
                __syncthreads();
                {                                                     // thread (query tid >> 3, channel quad tid & 7)
                    const int q = tid >> 3;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = 4 * (tid & 7) + j;
                        const float v = ((red[0][q * LDQ + c] + red[1][q * LDQ + c]) + red[2][q * LDQ + c]) + red[3][q * LDQ + c];
                        if (c0 + c < p.C) {
                            if (ACC1) acc1[q * p.C + c0 + c] += v;
                            else if (q_stage >= 0) atomicAdd(df1b + (size_t)q_stage * p.C + c0 + c, v);
                        }
                    }
                }
            }
        }
    }
    if (ACC1) {
        __syncthreads();
        // (query offsets do not depend on the coordinate set: q_stage of the last set serves)
        if (q_stage >= 0)
            for (int c = tid & 7; c < p.C; c += 8) df1b[(size_t)q_stage * p.C + c] = acc1[(tid >> 3) * p.C + c];
    }
}

int tile_prepare(AltTileArgs& a, const char* who)
{
    if (a.r > 7) return fail(RAFTCORR_EUNSUPPORTED, "%s: radius %d > 7", who, a.r);
    int64_t cells = (int64_t)a.H * a.W;
    for (int l = 0; l < a.L; ++l) cells = std::max<int64_t>(cells, (int64_t)a.f2.g[l].h * a.f2.g[l].w);
    if (cells * a.C * 4 >= (int64_t)1 << 31)
        return fail(RAFTCORR_EUNSUPPORTED, "%s: a feature map of %lld x %d floats exceeds 2 GiB", who,
                    (long long)cells, a.C);
    const int tiles = ((a.W + TQW - 1) / TQW) * ((a.H + TQH - 1) / TQH);
    a.band = (tiles + 7) / 8;
    bool aligned = ((uintptr_t)a.f1 & 15) == 0;
    for (int l = 0; l < a.L; ++l) aligned = aligned && ((uintptr_t)a.f2.ptr[l] & 15) == 0;
    a.vec = (a.C % 4 == 0) && aligned;
    return RAFTCORR_OK;
}

dim3 tile_grid(const AltTileArgs& a) { return dim3(8u * a.band * a.L * a.B * a.Nc); }

int run_fwd(AltTileArgs& a, const char* who, hipStream_t s)
{
    int rc = tile_prepare(a, who);
    if (rc) return rc;
    const int rd = 2 * a.r + 1;
    const bool small = TQ * rd * rd <= 256 * 11;
    if (small && a.vec)       hipLaunchKernelGGL((alt_tile_fwd_kernel<11, true>), tile_grid(a), dim3(256), 0, s, a);
    else if (small)           hipLaunchKernelGGL((alt_tile_fwd_kernel<11, false>), tile_grid(a), dim3(256), 0, s, a);
    else if (a.vec)           hipLaunchKernelGGL((alt_tile_fwd_kernel<29, true>), tile_grid(a), dim3(256), 0, s, a);
    else                      hipLaunchKernelGGL((alt_tile_fwd_kernel<29, false>), tile_grid(a), dim3(256), 0, s, a);
    return check_launch("alt_tile_fwd_kernel");
}

// d(fmap1) is overwritten (LDS accumulator; zero fill + atomics only when C is too large for it); d(fmap2) of every
// level is zeroed when zero_df2 and accumulated with atomics
int run_bwd(AltTileArgs& a, bool zero_df2, const char* who, hipStream_t s)
{
    int rc = tile_prepare(a, who);
    if (rc) return rc;
    const size_t acc_bytes = sizeof(float) * (size_t)TQ * a.C;
    const bool acc1 = acc_bytes <= 64 * 1024;
    hipError_t e = hipSuccess;
    if (!acc1) e = hipMemsetAsync(a.df1, 0, sizeof(float) * (size_t)a.B * a.C * a.H * a.W, s);
    for (int l = 0; zero_df2 && l < a.L && e == hipSuccess; ++l)
        e = hipMemsetAsync(a.df2.ptr[l], 0, sizeof(float) * (size_t)a.B * a.C * a.df2.g[l].h * a.df2.g[l].w, s);
    if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "%s: memset: %s", who, hipGetErrorString(e));
    const dim3 grid(8u * a.band * a.L * a.B);                // one workgroup per query tile: it walks the Nc sets
    typedef void (*Kern)(AltTileArgs);
    const Kern kern = acc1 ? (a.vec ? alt_tile_bwd_kernel<true, true> : alt_tile_bwd_kernel<false, true>)
                           : (a.vec ? alt_tile_bwd_kernel<true, false> : alt_tile_bwd_kernel<false, false>);
    if (acc1) {                                              // static + dynamic LDS beyond 64 KiB: opt in
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "%s: %s", who, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), acc1 ? acc_bytes : 0, s, a);
    return check_launch("alt_tile_bwd_kernel");
}

}  // namespace

// ---- one level, the reference's pybind surface (unscaled, coords (B,N,H1,W1,2)) ----------------
int launch_alt_level_fwd(const float* f1, const float* f2, const float* coords, float* corr,
                         int B, int N, int H1, int W1, int H2, int W2, int C, int r, hipStream_t s)
{
    AltTileArgs a = {};
    a.f1 = f1; a.f2.ptr[0] = const_cast<float*>(f2); a.f2.g[0].h = H2; a.f2.g[0].w = W2;
    a.coords = coords; a.out = corr;
    a.B = B; a.Nc = N; a.C = C; a.H = H1; a.W = W1; a.L = 1; a.r = r;
    a.planar = 0; a.level_scale = 0; a.out_div = 1.0f;
    return run_fwd(a, "alt_forward", s);
}

int launch_alt_level_bwd(const float* f1, const float* f2, const float* coords, const float* cg,
                         float* g1, float* g2, float* gc,
                         int B, int N, int H1, int W1, int H2, int W2, int C, int r, hipStream_t s)
{
    if (gc) {
        hipError_t e = hipMemsetAsync(gc, 0, sizeof(float) * (size_t)B * N * H1 * W1 * 2, s);
        if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "alt_backward: memset: %s", hipGetErrorString(e));
    }
    AltTileArgs a = {};
    a.f1 = f1; a.f2.ptr[0] = const_cast<float*>(f2); a.f2.g[0].h = H2; a.f2.g[0].w = W2;
    a.df2.ptr[0] = g2; a.df2.g[0].h = H2; a.df2.g[0].w = W2;
    a.coords = coords; a.dout = cg; a.df1 = g1;
    a.B = B; a.Nc = N; a.C = C; a.H = H1; a.W = W1; a.L = 1; a.r = r;
    a.planar = 0; a.level_scale = 0; a.out_div = 1.0f;
    return run_bwd(a, true, "alt_backward", s);
}

}  // namespace raftcorr
