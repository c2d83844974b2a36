// On-the-fly ("alternate") correlation, forward, on the 16-bit matrix cores at fp32-grade accuracy.
// Replaces AlternateCorrBlock.__call__ (core/corr.py:75-92: per level permute + alt_cuda_corr.forward
// (correlation_kernel.cu:19-119) + stack + / sqrt(C)) in one launch for all levels.
//
// Same idea as alt_tile.hip (the fp32-MFMA version, which still serves the pybind drop-in): neighbouring
// queries look at overlapping windows, so a workgroup takes a TILE of queries, finds the bounding box of
// their windows at one level and computes the dense product  S[q][u] = <f1[q], f2_l[u]>  on the matrix
// cores; every query then picks its (2r+2)^2 taps out of S in LDS and blends them.  What is new:
//
//  * 8x8 = 64 queries per workgroup (two MFMA row blocks share every staged target), box walked in passes
//    of 192 targets (6 MFMA column blocks); a pass computes only the column blocks that hold box targets.
//  * The features are PACKED once per frame pair (raftcorr_alt_pack, the constructor of AlternateCorrBlock):
//    each map is scaled by a power of two into the top of fp16's range and every value split into two fp16
//    pieces x = hi + lo (22 significand bits).  The product is three v_mfma_f32_32x32x16_f16 terms
//    (hi*hi + hi*lo + lo*hi) accumulated in fp32 and unscaled exactly -- the fp16x2 scheme of
//    build_pyramid.hip, fp32-grade (2e-7 of max|S| at C = 256) at 16/3 of the fp32 MFMA rate.
//  * Packed layout: per pixel, per 32-channel chunk, one 128-byte line [piece][k-step][k-half][8 x fp16] --
//    an MFMA operand fragment is one 16-byte slot.  Target lines go global -> LDS by direct-to-LDS DMA
//    (global_load_lds_dwordx4, 8 lanes per line, no VGPR round trip, no conversion), double buffered, one
//    barrier per chunk; slots are XOR-swizzled by (row >> 1) & 7 on the way in (the DMA writes 1 KiB
//    contiguously, the swizzle is applied to the SOURCE slot each lane fetches) so that the ds_read_b128
//    fragment reads of 128-byte rows are bank-conflict-free.  The query rows travel the same way: read
//    straight from global memory as fragments (16 bytes per lane from 32 different lines per instruction,
//    every line touched by four instructions of two waves) they cost more L1 -> L2 requests than all the
//    targets together and made the loop request-bound (tools/alt_lab.py ablations: 15 of 50 us).
#include "common.h"
#include "alt_common.h"
#include <algorithm>

namespace raftcorr {

namespace {

constexpr int UP = 192;                           // box targets per pass (6 MFMA column blocks)
constexpr int NBW = UP / 64;                      // column blocks per wave
constexpr int STG = (TQ + UP) * 8;                // uint4 slots per staging buffer: 64 query rows + UP target rows of 128 B
constexpr int KC = 32;                            // channels per staged chunk = one 128-byte line per pixel
constexpr int LDS_S = UP + 1;                     // S[q][u] row stride (odd: the per-query gathers are conflict-free)

struct AltFwdArgs {
    const uint4* f1p;                             // packed fmap1: [B][H*W][Cp/32][8 slots of 16 B]
    const uint4* f2p[RAFTCORR_MAX_LEVELS];        // packed pooled fmap2, level l: [B][H_l*W_l][Cp/32][8]
    int hl[RAFTCORR_MAX_LEVELS], wl[RAFTCORR_MAX_LEVELS];
    const float* coords;                          // (B, 2, H, W)
    float* out;                                   // (B, L*rd*rd, H, W)
    const float* maxpart;                         // [2][MAXPARTS16]: partial maxima of |fmap1|, |fmap2| (scales)
    int B, C, Cp, H, W, L, r;
    int band;                                     // tiles per XCD band
    float sqrt_c;
    // window mode (qlist != nullptr): the tiles are the query groups of launch_alt_groups (alt_bwd.hip), the same
    // list for every level
    const int* qlist;                             // [B][gmax][64] query pixel indices, -1 = empty slot
    const int* ntiles;                            // [B] groups in use
    int np0, np1;                                 // workgroups per level-0 / level-1 box (bands of rows), >= 1 (np1 = 1 when L = 1)
    int gmax;
};

struct TileSh : TileQ {
    int toff[UP];                                 // pixel index of the pass' box targets in the level, -1 beyond the box
};

// ---------------------------------------------------------------------------------------------
// pack: (B, C, HW) fp32 NCHW -> [B][HW][Cp/32][piece 2][32 ch] fp16, scaled.  grid (sum over the maps of ceil(HW/256), Cp/8, B)
__global__ __launch_bounds__(256)
void alt_pack16_kernel(PackSet ps)
{
    const int m = packset_map(ps, blockIdx.x);
    const float* __restrict__ f = ps.src[m];
    uint4* __restrict__ packed = ps.dst[m];
    const float* __restrict__ maxpart = ps.maxpart[m];
    const int C = ps.C, Cp = ps.Cp, HW = ps.h[m] * ps.w[m];
    const int pix = (blockIdx.x - ps.blk0[m]) * 256 + threadIdx.x, c0 = blockIdx.y * 8, b = blockIdx.z;
    const float scale = ldexpf(1.0f, range_exp16(wave_max128(fmaxf(maxpart[threadIdx.x & 63], maxpart[(threadIdx.x & 63) + 64]))));
    if (pix >= HW) return;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = f[((size_t)b * C + min(c0 + j, C - 1)) * HW + pix];   // all 8 loads in flight
    unsigned short h[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = c0 + j < C ? x[j] * scale : 0.f;
        const _Float16 hi = (_Float16)v;                     // RNE
        const _Float16 lo = (_Float16)(v - (float)hi);       // exact remainder, rounded once
        h[0][j] = __builtin_bit_cast(unsigned short, hi);
        h[1][j] = __builtin_bit_cast(unsigned short, lo);
    }
    uint4* dst = packed + ((size_t)b * HW + pix) * (Cp / 32) * 8 + (c0 >> 5) * 8 + ((c0 & 31) >> 3);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        uint4 w;
        w.x = h[pc][0] | ((unsigned)h[pc][1] << 16); w.y = h[pc][2] | ((unsigned)h[pc][3] << 16);
        w.z = h[pc][4] | ((unsigned)h[pc][5] << 16); w.w = h[pc][6] | ((unsigned)h[pc][7] << 16);
        dst[pc * 4] = w;                                     // slot = piece * 4 + (channel % 32) / 8
    }
}

// ---------------------------------------------------------------------------------------------
// Query tiles by WINDOW position instead of pixel position.  A pixel tile's box is the bounding box of its 64 windows:
// 18 x 18 targets at level 0 when the flow is smooth, but it grows with every pixel of disagreement between neighbours
// (37 x 37 for per-pixel noise of sigma = 4), and the product covers the whole box for all 64 queries.  In window mode
// the tiles are the query groups of launch_alt_groups (alt_bwd.hip: queries dealt by the 8 x 8 cell of level-0 targets
// that holds their window origin, rows of cells laid end to end, 64 slots per group, built without atomics): at level
// l the windows of a group lie 2^l times closer, so one list serves every level, and the lookup's backward pass
// takes the same groups.  The lists are a function of the coordinates alone: window mode is bit-reproducible
// (rounds 1-2 bucketed per level with LDS atomics; tile membership followed the order of the atomics).

// ---------------------------------------------------------------------------------------------
// forward.  grid: 8 XCD bands x (tiles per band x L x B), block 256 = 4 waves.
// wave w: MFMA row block mb = w & 1 (32 of the 64 queries), column blocks (w >> 1) + 2 j, j < NBW.
// RMAX: largest radius the instantiation's register arrays hold.
template <int RMAX>
__global__ __launch_bounds__(256, 2)
void alt_fwd16_kernel(AltFwdArgs p)
{
    constexpr int RDM = 2 * RMAX + 1, NCOL = (RDM + 3) / 4;  // window columns per thread (4 threads per query)
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];     // 2 x [TQ + UP rows][8 slots] staging; later S[TQ][LDS_S]
    __shared__ TileSh sh;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = p.r, rd = 2 * r + 1, rd2 = rd * rd, T = 2 * r + 2;
#if defined(RC_LAB_STAMPS)
    unsigned long long* stamps = (unsigned long long*)p.out + (size_t)blockIdx.x * 16;
    int nst = 0;
#define ALT_STAMP() do { if (threadIdx.x == 0 && nst < 16) stamps[nst] = __builtin_amdgcn_s_memrealtime(); ++nst; } while (0)
#else
#define ALT_STAMP()
#endif
    ALT_STAMP();
    // partial maxima behind the unscale factor: fetched now, used after the last pass (a dependent fetch there
    // would add a memory round trip to every workgroup's critical path)
    const float mp1 = fmaxf(p.maxpart[threadIdx.x & 63], p.maxpart[(threadIdx.x & 63) + 64]);
    const float mp2 = fmaxf(p.maxpart[MAXPARTS16 + (threadIdx.x & 63)], p.maxpart[MAXPARTS16 + (threadIdx.x & 63) + 64]);
    // Workgroup id -> (XCD band, tile, slot, batch); slot -> (level, part).  A tile's level-0 box is by far its largest (the
    // window is 2r + 2 wide at every level, the spread of the tile's queries halves with each level), and where the
    // tiles do not fill the chip the launch lasts as long as its level-0 workgroups.  So level 0 has np0 slots, the
    // first ones of the launch: a box of three passes and more is cut into np0 bands of rows that share their border
    // rows, so that the two tap rows of every output element lie in one band and each element has one owner -- no
    // partial sums to combine.
    Tile16 t;
    const int nslots = p.L + p.np0 - 1 + p.np1 - 1;
    int part, np;
    {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, j = idx % p.band, bl = idx / p.band;
        const int slot = bl % nslots;
        if (slot < p.np0)              { t.l = 0; part = slot; np = p.np0; }
        else if (slot < p.np0 + p.np1) { t.l = 1; part = slot - p.np0; np = p.np1; }
        else                           { t.l = slot - p.np0 - p.np1 + 2; part = 0; np = 1; }
        t.b = bl / nslots;
        if (p.qlist) {                                       // tiles = query groups, bands over the groups in use
            const int nt = p.ntiles[t.b], bandl = (nt + 7) >> 3, tile = xcd * bandl + j;
            t.live = j < bandl && tile < nt;
            if (!t.live) return;                             // whole workgroup
            t.ty0 = t.tx0 = 0;
            tile_level_setup(p, sh, t, p.coords + (size_t)t.b * 2 * p.H * p.W, p.qlist + ((size_t)t.b * p.gmax + tile) * 64);
        } else {
            const int tiles_x = (p.W + QW - 1) / QW, tile = xcd * p.band + j;
            t.live = tile < tiles_x * ((p.H + QH - 1) / QH);
            if (!t.live) return;                             // whole workgroup
            t.ty0 = (tile / tiles_x) * QH;
            t.tx0 = (tile % tiles_x) * QW;
            tile_level_setup(p, sh, t, p.coords + (size_t)t.b * 2 * p.H * p.W);
        }
    }
    const int nkc = p.Cp / KC;
    const size_t P = (size_t)p.H * p.W;
    const int mb = wave & 1, nb0 = wave >> 1;
    const int qi = tid & 63, g4 = wave;                      // outputs of this thread: query qi, window columns g4 + 4 i
    float* S = (float*)smem;
    // rows of first taps this workgroup owns: [own_lo, own_hi)
    int own_lo = INT_MIN, own_hi = INT_MAX;
    if (np > 1) {
        // a box of one or two passes stays whole (smooth flow: the spare workgroups leave here, after the set-up)
        if (t.U <= 2 * UP || t.bh < 2 * np) { if (part > 0) return; }    // (whole workgroup)
        else {
            const int y0 = t.ymin + t.bh * part / np;
            const int y1 = part + 1 < np ? t.ymin + t.bh * (part + 1) / np : t.ymin + t.bh - 1;   // inclusive, shared
            if (part > 0) own_lo = y0;
            if (part + 1 < np) own_hi = y1;
            t.ymin = y0; t.bh = y1 - y0 + 1; t.U = t.bw * t.bh;
        }
    }
    ALT_STAMP();

    const uint4* __restrict__ f1b = p.f1p + (size_t)t.b * P * nkc * 8;
    const uint4* __restrict__ f2b = p.f2p[t.l] + (size_t)t.b * t.HWl * nkc * 8;
    // DMA source of this lane for the query rows: row groups wave and 4 + wave (8 rows each)
    unsigned asrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = (i * 4 + wave) * 8 + (lane >> 3);
        asrc[i] = (unsigned)max(sh.qpix[rr], 0) * (unsigned)(nkc * 8) + (unsigned)((lane & 7) ^ ((rr >> 1) & 7));
    }

    // outputs of this thread: query qi = tid & 63, window columns a = g4 + 4 i (g4 = wave: uniform per wave)
    float o[NCOL][RDM];
#pragma unroll
    for (int i = 0; i < NCOL; ++i)
#pragma unroll
        for (int bb = 0; bb < RDM; ++bb) o[i][bb] = 0.f;

    for (int ubase = 0; ubase < t.U; ubase += UP) {
        __syncthreads();                                     // previous pass: S and toff are no longer read
        {
            const int u = ubase + tid;
            int off = -1;
            if (u < t.U) { const int uy = u / t.bw, ux = u - uy * t.bw; off = (t.ymin + uy) * t.Wl + t.xmin + ux; }
            if (tid < UP) sh.toff[tid] = off;
        }
        __syncthreads();
        const int nlive = min(t.U - ubase, UP);
        ALT_STAMP();
        // DMA sources of this lane for the pass: UP/32 row groups of 8 targets per wave (rows (i*4 + wave)*8 + lane/8);
        // LDS slot lane&7 of row rr receives source slot (lane&7) ^ ((rr >> 1) & 7)
        unsigned src[UP / 32];
#pragma unroll
        for (int i = 0; i < UP / 32; ++i) {
            const int rr = (i * 4 + wave) * 8 + (lane >> 3);
            src[i] = (unsigned)max(sh.toff[rr], 0) * (unsigned)(nkc * 8) + (unsigned)((lane & 7) ^ ((rr >> 1) & 7));
        }
        auto stage = [&](int kc, uint4* buf) {               // buf: [TQ query rows | UP target rows] x 8 slots
#if !defined(RC_LAB_NOSTAGE)
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(f1b + asrc[i] + kc * 8, buf + (i * 4 + wave) * 64);
#pragma unroll
            for (int i = 0; i < UP / 32; ++i)
                if ((i * 4 + wave) * 8 < nlive)              // wave-uniform: row groups beyond the box are not fetched
                    glds16(f2b + src[i] + kc * 8, buf + TQ * 8 + (i * 4 + wave) * 64);
#endif
        };
        v16f acc[NBW];
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        // live column blocks of this wave (blocks nb0 + 2 j), wave-uniform
        const int nlb = nb0 * 32 < nlive ? min((nlive - nb0 * 32 + 63) / 64, NBW) : 0;
        const int key = (l31 >> 1) & 7;                      // (row >> 1) & 7 for rows nb*32 + l31 and mb*32 + l31

        // One K chunk.  The buffers are restrict-qualified parameters: after inlining, the DMA destination and the
        // fragment source carry distinct alias scopes, so hipcc does not put `s_waitcnt vmcnt(0)` in front of the
        // fragment reads while the next chunk's DMA is in flight (build_pyramid.hip has the same construction).
        auto kchunk = [&](int kc, const uint4* __restrict__ cur, uint4* __restrict__ nxt) {
            // A bare s_barrier: __syncthreads() carries a fence.  After it everybody's part of chunk kc has landed
            // (vmcnt counts a DMA once its data is in LDS) and nobody still reads the buffer that is restaged next.
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kc + 1 < nkc) stage(kc + 1, nxt);
#if !defined(RC_LAB_NOMFMA)
            const uint4* arow = cur + (mb * 32 + l31) * 8;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 ah = __builtin_bit_cast(f16x8, arow[(s * 2 + kh) ^ key]);
                const f16x8 al = __builtin_bit_cast(f16x8, arow[(4 + s * 2 + kh) ^ key]);
                f16x8 bh[NBW], bl[NBW];
#pragma unroll
                for (int j = 0; j < NBW; ++j)
                    if (j < nlb) {
                        const uint4* brow = cur + TQ * 8 + ((nb0 + 2 * j) * 32 + l31) * 8;
                        bh[j] = __builtin_bit_cast(f16x8, brow[(s * 2 + kh) ^ key]);
                        bl[j] = __builtin_bit_cast(f16x8, brow[(4 + s * 2 + kh) ^ key]);
                    }
                // small terms first (lo*hi, hi*lo, hi*hi); the live column blocks take turns inside each term, so
                // that no MFMA waits for the one before it
#pragma unroll
                for (int j = 0; j < NBW; ++j)
                    if (j < nlb) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBW; ++j)
                    if (j < nlb) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBW; ++j)
                    if (j < nlb) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[j], 0, 0, 0);
            }
#endif
        };
        stage(0, smem);
        for (int kc = 0; kc < nkc; kc += 2) {
            kchunk(kc, smem, smem + STG);
            if (kc + 1 < nkc) kchunk(kc + 1, smem + STG, smem);
        }
        __syncthreads();                                     // staging buffers are free: S takes their place
        ALT_STAMP();
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int nb = nb0 + 2 * j;
            if (j < nlb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
                    S[row * LDS_S + nb * 32 + l31] = acc[j][i];
                }
            }
        }
        __syncthreads();
        ALT_STAMP();
        // gather: window column a needs tap columns a, a+1 for the T tap rows; u = (wy+ty-ymin)*bw + (wx+tx-xmin) - ubase
        const int wx = sh.wx0[qi] - t.xmin, wy = sh.wy0[qi] - t.ymin;
        const float fx = sh.fx[qi], fy = sh.fy[qi];
        const float* Sq = S + qi * LDS_S;
#pragma unroll
        for (int i = 0; i < NCOL; ++i) {
            const int a = g4 + 4 * i;
            if (a < rd) {                                    // wave-uniform
                const int x0 = wx + a, x1 = x0 + 1;
                const bool vx0 = (unsigned)x0 < (unsigned)t.bw, vx1 = (unsigned)x1 < (unsigned)t.bw;
                float hprev = 0.f;
#pragma unroll
                for (int ty = 0; ty <= RDM; ++ty) {
                    if (ty < T) {
                        const int y = wy + ty;
                        const int u = y * t.bw + x0 - ubase;
                        const bool vy = (unsigned)y < (unsigned)t.bh;
                        const bool ok0 = vy && vx0 && (unsigned)u < (unsigned)UP;
                        const bool ok1 = vy && vx1 && (unsigned)(u + 1) < (unsigned)UP;
                        const float c0 = ok0 ? Sq[max(u, 0)] : 0.f, c1 = ok1 ? Sq[max(u + 1, 0)] : 0.f;
                        const float h = c0 * (1.f - fx) + c1 * fx;
                        if (ty > 0) o[i][ty - 1] += hprev * (1.f - fy) + h * fy;
                        hprev = h;
                    }
                }
            }
        }
    }
    ALT_STAMP();
#if defined(RC_LAB_STAMPS)
    if (threadIdx.x == 0) { stamps[14] = (unsigned long long)t.l; stamps[15] = (unsigned long long)t.U; }
    return;
#endif
    const int myqi = sh.qpix[qi], wy0q = sh.wy0[qi];
    const float unscale = ldexpf(1.0f, -(range_exp16(wave_max128(mp1)) + range_exp16(wave_max128(mp2))));
    if (myqi >= 0) {
        float* __restrict__ out = p.out + (((size_t)t.b * p.L + t.l) * rd2) * P + myqi;
#pragma unroll
        for (int i = 0; i < NCOL; ++i) {
            const int a = g4 + 4 * i;
            if (a < rd) {
#pragma unroll
                for (int bb = 0; bb < RDM; ++bb) {
                    const int r0 = wy0q + bb;                // row of the element's first tap
                    if (bb < rd && r0 >= own_lo && r0 < own_hi) out[(size_t)(a * rd + bb) * P] = (o[i][bb] * unscale) / p.sqrt_c;
                }
            }
        }
    }
}

}  // namespace

size_t alt_packed_bytes(int B, int C, int H, int W)
{
    const size_t Cp = (size_t)(C + 31) / 32 * 32;
    return (size_t)B * H * W * Cp * 4;
}

int launch_alt_pack_set(const float* const* fmaps, const int* hs, const int* ws, const float* const* maxparts,
                        void* const* packed, int n, int B, int C, hipStream_t s)
{
    PackSet ps = {};
    ps.n = n; ps.C = C; ps.Cp = (C + 31) / 32 * 32;
    for (int m = 0; m < n; ++m) {
        ps.src[m] = fmaps[m]; ps.dst[m] = (uint4*)packed[m]; ps.maxpart[m] = maxparts[m]; ps.h[m] = hs[m]; ps.w[m] = ws[m];
        ps.blk0[m + 1] = ps.blk0[m] + (hs[m] * ws[m] + 255) / 256;
    }
    hipLaunchKernelGGL(alt_pack16_kernel, dim3(ps.blk0[n], ps.Cp / 8, B), dim3(256), 0, s, ps);
    return check_launch("alt_pack16_kernel");
}

int launch_alt_pack(const float* fmap, int B, int C, int H, int W, const float* maxpart, void* packed, hipStream_t s)
{
    return launch_alt_pack_set(&fmap, &H, &W, &maxpart, &packed, 1, B, C, s);
}

size_t alt_lookup_fwd_sorted_workspace_bytes(int B, int H, int W, int L) { (void)L; return alt_groups_plan(B, H, W).bytes; }

// workspace != nullptr: window mode.  The workspace is the lookup's group buffer: built here, and afterwards what the
// lookup's backward pass (launch_alt_lookup_bwd16, `groups`) can take instead of building the same lists again
int launch_alt_lookup_fwd16(const void* f1p, const void* const* f2p, const float* maxpart, int B, int H, int W, int C, int L,
                            int r, const float* coords, float* out, void* workspace, size_t workspace_bytes, hipStream_t s)
{
    if (r > 7) return fail(RAFTCORR_EUNSUPPORTED, "alt_lookup_fwd: radius %d > 7", r);
    AltFwdArgs a = {};
    a.f1p = (const uint4*)f1p;
    for (int l = 0; l < L; ++l) { a.f2p[l] = (const uint4*)f2p[l]; a.hl[l] = H >> l; a.wl[l] = W >> l; }
    a.coords = coords; a.out = out; a.maxpart = maxpart;
    a.B = B; a.C = C; a.Cp = (C + 31) / 32 * 32; a.H = H; a.W = W; a.L = L; a.r = r;
    a.sqrt_c = sqrtf((float)C);
    if ((size_t)H * W * (a.Cp / 32) * 8 >= ((size_t)1 << 31))
        return fail(RAFTCORR_EUNSUPPORTED, "alt_lookup_fwd: a packed map of %d x %d x %d exceeds 2^31 slots", H, W, C);
    const int tiles = ((W + QW - 1) / QW) * ((H + QH - 1) / QH);
    a.band = (tiles + 7) / 8;
    if (workspace) {                                         // window mode: the groups in front
        const AltGroupsPlan gp = alt_groups_plan(B, H, W);
        if (workspace_bytes < gp.bytes)
            return fail(RAFTCORR_EINVAL, "alt_lookup_fwd_sorted: workspace of %zu bytes needed, %zu given", gp.bytes, workspace_bytes);
        const int rc = launch_alt_groups(coords, B, H, W, workspace, workspace_bytes, s);
        if (rc) return rc;
        a.ntiles = (const int*)workspace;
        a.qlist = (const int*)((const char*)workspace + gp.off_qlist);
        a.gmax = gp.gmax;
        a.band = (gp.gmax + 7) / 8;
    }
    const size_t sm = std::max((size_t)2 * STG * sizeof(uint4), (size_t)TQ * LDS_S * sizeof(float));
    // Level-0 boxes of three passes and more in two bands of rows (see the kernel) where the tiles do not fill the chip.
    // KITTI, per call: window groups under random walks 51 -> 46 us, pixel tiles 57 / 83 / 102 -> 42 / 50 / 61 us at
    // sigma = 3 / 4.2 / 6; smooth flow (boxes of two passes, nothing is cut) pays the set-up of the spare workgroups.
    // 1080p (2080 workgroups before the cut): 185 -> 205 us, left whole.
    const long long ntile = workspace ? ((long long)H * W + TQ - 1) / TQ : tiles;
    a.np0 = (long long)B * ntile * L < 1024 ? 2 : 1;
    a.np1 = 1;
#if defined(RC_LAB)
    if (const char* e = getenv("RC_NP0")) a.np0 = atoi(e) >= 1 ? atoi(e) : 1;
    if (const char* e = getenv("RC_NP1")) a.np1 = atoi(e) >= 1 && L > 1 ? atoi(e) : 1;
#endif
    const dim3 grid(8u * a.band * (L + a.np0 - 1 + a.np1 - 1) * B);
    static thread_local bool allowed[64][2] = {};            // >64 KiB of dynamic LDS: opt in once per device and kernel
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    const int which = r <= 4 ? 0 : 1;
    if (e == hipSuccess && !allowed[dev & 63][which]) {
        e = which == 0 ? hipFuncSetAttribute((const void*)alt_fwd16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)
                       : hipFuncSetAttribute((const void*)alt_fwd16_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e == hipSuccess) allowed[dev & 63][which] = true;
    }
    if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "alt_lookup_fwd: %s", hipGetErrorString(e));
    if (which == 0) hipLaunchKernelGGL(alt_fwd16_kernel<4>, grid, dim3(256), sm, s, a);
    else            hipLaunchKernelGGL(alt_fwd16_kernel<7>, grid, dim3(256), sm, s, a);
    return check_launch("alt_fwd16_kernel");
}

}  // namespace raftcorr
