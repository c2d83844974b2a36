// All-pairs correlation volume + pooled pyramid in one launch (gfx950 / CDNA4).
//
// Replaces CorrBlock.__init__ / CorrBlock.corr of the reference (core/corr.py:13-27,53-61):
//   torch.matmul(fmap1^T, fmap2) -> / sqrt(D) -> (L-1) x F.avg_pool2d(2, 2)
// i.e. one sgemm, one elementwise pass and three pooling passes over a 200-260 MB volume.
// Here: a small pack pass (14 MB) + ONE kernel, an fp32-MFMA contraction over D whose epilogue
// scales the tile and produces the level-1..3 cells straight from the accumulators.
//
// C[q, p] = sum_d F1[d, q] * F2[d, t(p)]; the column order p is the patch-tile order of
// layout.h.  Both operands are (D, pixels) with the pixel index contiguous, which is the
// operand shape of v_mfma_f32_32x32x2_f32 (lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]), so no transposes anywhere.
//
//   pack    (one small launch) F1 -> [M tile][k][128], F2 -> [patch][k][cell], zero padded:
//           tile panels are contiguous, the K loop has no bounds checks and each wave moves
//           its share with 16-byte direct-to-LDS loads (global_load_lds_dwordx4).
//   block   256 threads = 4 waves, 128 queries x one 8x16 patch (128 targets)
//   wave w  32 queries x 128 targets: MFMA column j (lane & 31), accumulator n <-> cell 4j + n of
//           the patch, so a lane holds one even-aligned 2x2 block of level 0 (16 contiguous bytes)
//   K loop  BK = 8, two LDS buffers (two __shared__ objects, loop unrolled by two: see the kernel), the DMA
//           of tile k+1 in flight while tile k is multiplied, one barrier per K tile; B fragments of the
//           four accumulators are one ds_read_b128.
//   epilog  level 0: one 16-byte store per lane and accumulator register (32 lanes = 512
//           contiguous bytes of a query row); level 1: the lane's own four values; level 2: one
//           DPP quad reduction; level 3: row_half_mirror + one cross-row swizzle.  Levels 2 / 3
//           have one value per 4 / 16 lanes: the lanes of a group store different registers
//           (query rows), so all 64 lanes of every store are at work -- 37 store instructions per
//           wave (the first version: 112, 48 of them mostly masked).  Sums are scaled by exact
//           powers of two, so the values equal avg_pool2d's up to the order of the three additions.
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <atomic>

namespace raftcorr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

constexpr int BM = 128;      // queries per block
constexpr int BK = 8;       // K rows per LDS tile of the shipped instantiation (bench.py, after the K loop was
                            // de-serialised: BK=8 236.7 us, BK=16 238.9 us; BK=32 is slower, tools/build_lab.hip)
constexpr int WPS = 3;      // waves per SIMD the register allocation is bounded for
constexpr int GROUP_M = 8;   // query tiles walked together (L2 reuse of the operand panels)

// ---------------------------------------------------------------------------------------
// pack: feature maps -> tile-major, zero-padded operand panels
// ---------------------------------------------------------------------------------------
//   F1p[b][tile][k][c]        = F1[b][k][tile*128 + c]                  (queries, linear)
//   F2p[b][patch][k][c]       = F2[b][k][target of cell c of the patch] (patch-tile order, layout.h;
//                               a lane's float4 of B = the 4 accumulators' columns 4j .. 4j+3)
// One launch for both operands: grid (tilesM + tilesN, PACK_KSPLIT, B); a thread owns one column
// of one tile and walks its share of the k rows, so source/destination offsets are computed once
// and both sides are coalesced along the pixel index.
constexpr int PACK_KSPLIT = 8;

struct PackArgs {
    const float* f1; const float* f2; float* f1p; float* f2p;
    int D, Dpad, H, W, tilesM, tilesN;
    const float* maxpart;       // fp16 pieces only: MAXPARTS partial maxima of |fmap1|, then of |fmap2| (device memory)
    int pair_from = 0x7fffffff; // F2 panels >= this hold the upper halves (cells 0..63) of TWO patches, see paired_cell()
};

// Paired panels.  With H % 8 in 1..4 the lower 4x16 half of every patch of the bottom patch row is padding only (cells
// 64..127, layout.h); one panel then carries the upper halves of two such patches -- cells 0..63 of patch
// pair_from + 2 (panel - pair_from) in its columns 0..63, of the next patch in columns 64..127 -- and the bottom row
// costs half the tiles (50x90: 39 panels instead of 42).  The main loops do not know; the pack kernels gather the
// columns and the epilogue addresses the stores accordingly.  cell m of panel `tile` -> (patch, cell inside the patch)
__device__ __forceinline__ void paired_cell(int pair_from, int tile, int m, int& patch, int& cell)
{
    patch = tile; cell = m;
    if (tile >= pair_from) { patch = pair_from + 2 * (tile - pair_from) + (m >> 6); cell = m & 63; }
}

__global__ __launch_bounds__(256)
void corr_pack_kernel(PackArgs p)
{
    const int N = p.H * p.W;
    const int c = threadIdx.x & 127, kp = threadIdx.x >> 7;
    const int b = blockIdx.z;
    int tile = blockIdx.x;
    const bool second = tile >= p.tilesM;
    int t = -1;
    if (!second) {
        const int q = tile * 128 + c;
        if (q < N) t = q;
    } else {
        tile -= p.tilesM;
        const int PW = patches_w(p.W);
        int patch, cell;
        paired_cell(p.pair_from, tile, c, patch, cell);   // column c = cell c of the panel
        const int n = cell >> 5, j = cell & 31;           // tile n of the patch, position j
        const int dx = (j & 1) | (((j >> 3) & 1) << 1) | ((j >> 4) << 2);
        const int dy = ((j >> 1) & 1) | (((j >> 2) & 1) << 1);
        const int y = (patch / PW) * PATCH_H + (n >> 1) * 4 + dy;
        const int x = (patch % PW) * PATCH_W + (n & 1) * 8 + dx;
        if (y < p.H && x < p.W) t = y * p.W + x;          // (an odd patch out: y >= H)
    }
    const float* __restrict__ src = (second ? p.f2 : p.f1) + (size_t)b * p.D * N + (t >= 0 ? t : 0);
    float* __restrict__ dst = (second ? p.f2p + ((size_t)b * p.tilesN + tile) * p.Dpad * 128
                                      : p.f1p + ((size_t)b * p.tilesM + tile) * p.Dpad * 128) + c;
    const int per = p.Dpad / PACK_KSPLIT;                 // Dpad is a multiple of 64
    const int k0 = blockIdx.y * per + kp;
    for (int i0 = 0; i0 < per / 2; i0 += 8) {             // 8 loads in flight per thread (latency-bound pass)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 2 * (i0 + u);
            v[u] = (t >= 0 && k < p.D && i0 + u < per / 2) ? src[(size_t)min(k, p.D - 1) * N] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 2 * (i0 + u);
            if (i0 + u < per / 2) dst[(size_t)k * 128] = v[u];
        }
    }
}

int launch_pack_operands(const float* f1, const float* f2, float* f1p, float* f2p, int B, int D, int Dpad,
                         int H, int W, int tilesM, int tilesN, hipStream_t s, int pair_from = 0x7fffffff)
{
    PackArgs a{f1, f2, f1p, f2p, D, Dpad, H, W, tilesM, tilesN, nullptr, pair_from};
    hipLaunchKernelGGL(corr_pack_kernel, dim3(tilesM + tilesN, PACK_KSPLIT, B), dim3(256), 0, s, a);
    return check_launch("corr_pack_kernel");
}

// ---------------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------------
struct BuildArgs {
    const float* f1p;     // [B][tilesM][Dpad][128]
    const float* f2p;     // [B][tilesN][Dpad][128], patch-tile order
    float* lvl[4];
    long long slab[4];    // elements per query slab, levels 0..3
    int B, Dpad, N, L;
    int tilesM, tilesN, nblk_per_b, PW;
    int group_m;          // query tiles walked together (L2 reuse of the operand panels)
    float inv_sqrt_d;     // used when sqrt(D) is a power of two (exact)
    float sqrt_d;         // divisor otherwise (bit-faithful to corr / sqrt(D), corr.py:61)
    const float* maxpart;     // fp16 pieces only (else null): the operands were scaled, see range_exponent()
    int pair_from = 0x7fffffff;   // panels (tn) >= this are paired (paired_cell()); tilesN counts panels
    int npatch = 0;               // patches of the target map (an odd patch out has no partner)
};

constexpr int MAXPARTS = 128;    // partial maxima per feature map (one per block of corr_maxabs_kernel)

// every lane gets the maximum of a map's partial maxima (no LDS, no barrier: each wave for itself)
__device__ __forceinline__ float wave_max_of_partials(const float* __restrict__ part)
{
    const int lane = threadIdx.x & 63;
    float m = fmaxf(part[lane], part[lane + 64]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}

// fp16 pieces: power-of-two factor that maps max|x| into [2^14, 2^15) -- the top of fp16's range, so
// that the low-order pieces stay normal numbers.  1 for an all-zero or non-finite map.
__device__ __forceinline__ int range_exponent(float m)
{
    if (!(m > 0.f) || !(m < 3.0e38f)) return 0;
    int e;
    frexpf(m, &e);                       // m = f * 2^e, f in [0.5, 1)
    return 15 - e;
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // i -> 7 - i inside groups of 8

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_chunk)
{
    // 64 lanes x 16 B: per-lane global source, LDS destination = chunk base + lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_chunk, 16, 0, 0);
}

// ---- block -> (batch, query tile, patch), XCD-aware ------------------------------------
// Blocks are dealt round-robin over the 8 XCDs; give every XCD a contiguous chunk of the
// tile list (bijective form, any grid size) and walk it GROUP_M query tiles at a time, so
// the blocks resident together on one XCD share a few A and B panels in its L2.
__device__ __forceinline__ void block_to_tile(const BuildArgs& p, int& b, int& tm, int& tn)
{
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7;
    const int qd = nwg >> 3, rm = nwg & 7;
    const int lin = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    b = lin / p.nblk_per_b;
    const int t = lin - b * p.nblk_per_b;
    const int per_group = p.group_m * p.tilesN;
    const int grp = t / per_group;
    const int first_m = grp * p.group_m;
    const int gsz = min(p.group_m, p.tilesM - first_m);
    const int within = t - grp * per_group;
    tm = first_m + within % gsz;
    tn = within / gsz;
}

#ifndef RC_EPI_AUX
#define RC_EPI_AUX 0        // cache-policy bits of the epilogue's buffer stores (lab knob)
#endif

// 16-bit volume elements (RAFTCORR_VOL_*): fp16 = RNE, beyond +-65504 -> inf; bf16 = RNE, NaN stays NaN.  Two values
// per hardware pack conversion (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32): a software RNE (and / compare / add / shift per
// value) made the 16-bit-volume epilogue twice as long as the float32 one -- 480 of its ~800 vector instructions per
// wave, on kernels whose matrix work is 64 MFMAs per wave.
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
template <int VOL>
__device__ __forceinline__ unsigned vol16_pair(float lo, float hi)        // lo -> bits 0..15, hi -> bits 16..31
{
    const f32x2v v = {lo, hi};
    if (VOL == 1) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
template <int VOL>
__device__ __forceinline__ unsigned vol16(float x) { return vol16_pair<VOL>(x, 0.f) & 0xffffu; }

typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));

// one pooled cell -> its level, as float32 or as a 16-bit element
template <int VOL>
__device__ __forceinline__ void store_cell(float v, __amdgpu_buffer_rsrc_t r, unsigned off)
{
    if (VOL == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, RC_EPI_AUX);
    else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)vol16<VOL>(v), r, off, 0, RC_EPI_AUX);
}

// Shared by the fp32 and bf16 kernels (both MFMAs have the same 32x32 C/D map): scale, store
// level 0, pool levels 1..3 out of the accumulators.  VOL: element type of the pyramid (0 = float32;
// 1 = fp16, 2 = bf16: same element order, every value rounded once from the fp32 accumulators, the
// pooled levels from their fp32 sums -- half the store traffic of the build and half the line
// traffic of every lookup).
template <bool EXACT_MUL, int VOL>
__device__ __forceinline__ void build_epilogue(const BuildArgs& p, f32x16 (&acc)[4], int b, int tn, int m0,
                                               int wave, int lane)
{
    const int kh = lane >> 5, l31 = lane & 31;
#if defined(RC_LAB_EPI_LOCAL)         // tools/build_lab.hip only: every workgroup stores to tile (0, 0): no HBM write traffic
    b = 0; tn = 0; m0 = 0;
#endif
    float unscale = 1.0f;                                      // exact power of two
    if (p.maxpart)
        unscale = ldexpf(1.0f, -(range_exponent(wave_max_of_partials(p.maxpart)) +
                                 range_exponent(wave_max_of_partials(p.maxpart + MAXPARTS))));
    // ---- epilogue ------------------------------------------------------------------------
    // C/D map of the 32x32 MFMA: column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // Stores go through buffer descriptors that cover exactly this wave's valid query rows:
    // rows past N (last query tile) and lanes that do not own a pooled cell get an offset
    // outside the descriptor and are dropped by the hardware range check -- no branches.
    const int q0w = m0 + wave * 32;
    const int rows = min(32, p.N - q0w);
    if (rows <= 0) return;                                   // wave-uniform
    const int L = p.L;
    constexpr unsigned ES = VOL ? 2 : 4;                     // bytes per volume element
    __amdgpu_buffer_rsrc_t r0, r1, r2, r3;
    {
        const size_t qb = (size_t)b * p.N + q0w;
        auto base = [&](int l) { return (char*)p.lvl[l] + qb * p.slab[l] * ES; };      // levels are byte-addressed here
        r0 = __builtin_amdgcn_make_buffer_rsrc(base(0), 0, (int)(rows * p.slab[0] * ES), 0x00020000);
        r1 = __builtin_amdgcn_make_buffer_rsrc(L > 1 ? base(1) : base(0), 0, L > 1 ? (int)(rows * p.slab[1] * ES) : 0, 0x00020000);
        r2 = __builtin_amdgcn_make_buffer_rsrc(L > 2 ? base(2) : base(0), 0, L > 2 ? (int)(rows * p.slab[2] * ES) : 0, 0x00020000);
        r3 = __builtin_amdgcn_make_buffer_rsrc(L > 3 ? base(3) : base(0), 0, L > 3 ? (int)(rows * p.slab[3] * ES) : 0, 0x00020000);
    }
#if defined(RC_LAB_TILE_MAJOR)       // tools/build_lab.hip only: level 0 stored as [query tile][patch][128 rows][128 cells]
    r0 = __builtin_amdgcn_make_buffer_rsrc((char*)p.lvl[0] + ((((size_t)b * p.tilesM + m0 / BM) * p.tilesN + tn) * 128 + wave * 32) * 128 * ES,
                                           0, (int)(rows * 128 * ES), 0x00020000);
    const unsigned s0b = 128 * ES, s1b = (unsigned)(p.slab[1] * ES);
#else
    const unsigned s0b = (unsigned)(p.slab[0] * ES), s1b = (unsigned)(p.slab[1] * ES);
#endif
    const unsigned s2b = (unsigned)(p.slab[2] * ES), s3b = (unsigned)(p.slab[3] * ES);
    // Lane l31 holds cells 4*l31 .. 4*l31+3 of the patch (accumulator n = cell 4*l31 + n): one
    // even-aligned 2x2 block of level 0.  Bits of l31 (layout.h): a0 = dy1, a1 = dx1, a2 = dx2,
    // a3 = column half, a4 = row half of the patch.
    const int a0 = lane & 1, a1 = (lane >> 1) & 1, a2 = (lane >> 2) & 1, a3 = (lane >> 3) & 1, a4 = (lane >> 4) & 1;
    unsigned c0, c1, c2, c3;
    if (tn < p.pair_from) {                                  // (workgroup-uniform)
#if defined(RC_LAB_TILE_MAJOR)
        c0 = (unsigned)(4 * l31) * ES;
#else
        c0 = (unsigned)(tn * 128 + 4 * l31) * ES;
#endif
        c1 = (unsigned)(tn * 32 + (l31 >> 3) * 8 + (a1 | (a0 << 1) | (a2 << 2))) * ES;
        c2 = (unsigned)(tn * 8 + (a2 | (a4 << 1) | (a3 << 2))) * ES;
        const int ph = tn / p.PW, pwi = tn - ph * p.PW;
        c3 = (unsigned)(((ph >> 1) * p.PW + pwi) * 4 + a3 + (ph & 1) * 2) * ES;
    } else {
        // paired panel: MFMA columns 0..15 are the upper half of one patch, 16..31 of the next (a4 picks the patch; both
        // sit in the upper row half).  The pooled values that reach across a4 -- level 3, and nothing else -- belong to
        // cells below the level's last row (H % 8 <= 4) and land in those cells' own slots; an odd patch out stores nothing.
        const int patch = p.pair_from + 2 * (tn - p.pair_from) + a4, l15 = l31 & 15;
        const int ph = patch / p.PW, pwi = patch - ph * p.PW;
        c0 = (unsigned)(patch * 128 + 4 * l15) * ES;
        c1 = (unsigned)(patch * 32 + a3 * 8 + (a1 | (a0 << 1) | (a2 << 2))) * ES;
        c2 = (unsigned)(patch * 8 + (a2 | (a3 << 2))) * ES;
        c3 = (unsigned)(((ph >> 1) * p.PW + pwi) * 4 + a3 + (ph & 1) * 2) * ES;
        if (patch >= p.npatch) c0 = c1 = c2 = c3 = 0x80000000u;            // beyond every descriptor: dropped
    }
    const int quad = lane & 3;
    const unsigned rowk = 4 * kh;

    f32x16 t2, t3;                                    // vector types: scalar arrays here end up in scratch memory
#pragma unroll
    for (int i = 0; i < 16; ++i) { t2[i] = 0.f; t3[i] = 0.f; }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const unsigned row = (unsigned)((reg & 3) + 8 * (reg >> 2)) + rowk;
        float v[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
            v[n] = EXACT_MUL ? acc[n][reg] * (p.inv_sqrt_d * unscale) : (acc[n][reg] * unscale) / p.sqrt_d;
        if (VOL == 0) {
            u32x4s w;
            w.x = __float_as_uint(v[0]); w.y = __float_as_uint(v[1]); w.z = __float_as_uint(v[2]); w.w = __float_as_uint(v[3]);
            __builtin_amdgcn_raw_buffer_store_b128(w, r0, row * s0b + c0, 0, RC_EPI_AUX);    // 32 lanes = 512 contiguous bytes
        } else {
            u32x2s w;
            w.x = vol16_pair<VOL>(v[0], v[1]); w.y = vol16_pair<VOL>(v[2], v[3]);
            __builtin_amdgcn_raw_buffer_store_b64(w, r0, row * s0b + c0, 0, RC_EPI_AUX);     // 32 lanes = 256 contiguous bytes
        }
        if (L > 1) {
            const float t1 = (v[0] + v[1]) + (v[2] + v[3]);                   // the lane's own 2x2 block
            store_cell<VOL>(t1 * 0.25f, r1, row * s1b + c1);
            if (L > 2) {
                const float s = t1 + dpp<DPP_XOR1>(t1);
                t2[reg] = s + dpp<DPP_XOR2>(s);                               // 4x4 sum, in all 4 lanes of the quad
                if (L > 3) {
                    const float u = t2[reg] + dpp<DPP_HALF_MIRROR>(t2[reg]);  // partner a2 ^ 1 (values are quad-uniform)
                    t3[reg] = u + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(u), 0x401F));   // a4 ^ 1
                }
            }
        }
    }
    // Levels 2 and 3 have one value per 4 / 16 lanes and register: lanes of a group take different
    // registers (= query rows), so every store instruction has all 64 lanes at work.
    if (L > 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                          // registers 4i .. 4i+3 = rows 8i + 4kh + 0..3
            const float lo = (quad & 1) ? t2[4 * i + 1] : t2[4 * i], hi = (quad & 1) ? t2[4 * i + 3] : t2[4 * i + 2];
            const float mine = (quad & 2) ? hi : lo;
            store_cell<VOL>(mine * 0.0625f, r2, (unsigned)(8 * i + quad + rowk) * s2b + c2);
        }
        if (L > 3) {
            const int rho = (lane & 7) | (a4 << 3);                            // this lane's register among the 16
            const bool k0 = rho & 1, k1 = rho & 2, k2 = rho & 4, k3 = rho & 8;
#define RC_SEL2(i) (k0 ? t3[2 * (i) + 1] : t3[2 * (i)])
#define RC_SEL4(i) (k1 ? RC_SEL2(2 * (i) + 1) : RC_SEL2(2 * (i)))
#define RC_SEL8(i) (k2 ? RC_SEL4(2 * (i) + 1) : RC_SEL4(2 * (i)))
            const float mine = k3 ? RC_SEL8(1) : RC_SEL8(0);
#undef RC_SEL8
#undef RC_SEL4
#undef RC_SEL2
            store_cell<VOL>(mine * 0.015625f, r3, (unsigned)((rho & 3) + 8 * (rho >> 2) + rowk) * s3b + c3);
        }
    }
}

template <bool EXACT_MUL, int BK, int WAVES_PER_SIMD, int VOL = 0>
__global__ __launch_bounds__(256, WAVES_PER_SIMD)
void corr_build_f32_kernel(BuildArgs p)
{
    // Two K-tile buffers as two separate __shared__ objects, and the K loop unrolled by two so that
    // each half names its buffers statically: with one array and a run-time buffer index hipcc cannot
    // tell the LDS-DMA destination (tile k+1) from the ds_read source (tile k) and waits vmcnt(0)
    // before the first fragment read -- every wave then sits out the whole fetch latency of the next
    // tile before it multiplies the current one.
    __shared__ __attribute__((aligned(16))) float smem0[2 * BK * 128];   // [A|B][k][128]
    __shared__ __attribute__((aligned(16))) float smem1[2 * BK * 128];

    int b, tm, tn;
    block_to_tile(p, b, tm, tn);
    const int m0 = tm * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Dpad = p.Dpad;

    const float* __restrict__ pa = p.f1p + ((size_t)b * p.tilesM + tm) * Dpad * 128;
    const float* __restrict__ pb = p.f2p + ((size_t)b * p.tilesN + tn) * Dpad * 128;

    // stage K tile kt into an LDS buffer: BK x 512 B of A and of B, BK/8 + BK/8 DMA pieces per wave
    auto stage = [&](int kt, float* la) {
        float* lb = la + BK * 128;
        const float* ga = pa + (size_t)kt * BK * 128;
        const float* gb = pb + (size_t)kt * BK * 128;
#pragma unroll
        for (int i = 0; i < BK / 8; ++i) {
            const int piece = wave * (BK / 8) + i;       // 1 KiB pieces, BK/2 per operand tile
            glds16(ga + piece * 256 + lane * 4, la + piece * 256);
            glds16(gb + piece * 256 + lane * 4, lb + piece * 256);
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

    const int nk = Dpad / BK;       // even: Dpad is a multiple of 64
    const int kh = lane >> 5;       // which of the 2 k rows of an MFMA step this lane feeds
    const int l31 = lane & 31;
    auto multiply = [&](const float* tile) {
        const float* la = tile + kh * 128 + wave * 32 + l31;
        const float* lb = tile + BK * 128 + kh * 128 + l31 * 4;
        float a_cur = la[0];
        float4 b_cur = *(const float4*)lb;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a_nxt = 0.f;
            float4 b_nxt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk + 1 < BK / 2) {
                a_nxt = la[(kk + 1) * 256];
                b_nxt = *(const float4*)(lb + (kk + 1) * 256);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur.w, acc[3], 0, 0, 0);
            a_cur = a_nxt;
            b_cur = b_nxt;
        }
    };
    stage(0, smem0);
    for (int kt = 0; kt < nk; kt += 2) {
        __syncthreads();            // tile kt has landed (vmcnt(0) precedes the barrier) and
                                    // nobody still reads the buffer that is restaged next
        stage(kt + 1, smem1);
        __builtin_amdgcn_sched_barrier(0);      // the DMA is issued here, not sunk below the MFMAs
        multiply(smem0);
        __syncthreads();
        if (kt + 2 < nk) stage(kt + 2, smem0);
        __builtin_amdgcn_sched_barrier(0);
        multiply(smem1);
    }

#if defined(RC_LAB_SKIP_EPILOGUE)      // tools/build_lab.hip only: time the main loop alone
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(acc[n][i]));
    return;
#endif
    build_epilogue<EXACT_MUL, VOL>(p, acc, b, tn, m0, wave, lane);
}

// ---------------------------------------------------------------------------------------
// bf16-operand variants: the same tiling and the same epilogue, products on
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) accumulated in fp32, pyramid in fp32.
//   NP = 1  operands rounded to bf16 (RNE): the "mixed precision" build, <= 1e-2 of the fp32 one.
//   NP = 3  every operand split into three bf16 pieces x = hi + mid + lo (24 significand bits,
//           each piece with its own exponent) and the six products of weight >= 2^-16 summed:
//           hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi.  The dropped terms are <= 2^-24
//           of a product, i.e. fp32 rounding level: the result is as close to the exact dot
//           product as an fp32 FMA chain is (measured 7e-9 vs 7e-7 of max|C| at D = 256) at 6/16
//           of its matrix-core time.
// With either, the contraction stops being the bottleneck: the kernel runs at the speed of its
// 1.33 x 4 N^2 bytes of stores.
//
// Operand panels, per tile:  [k-step (16 k)][piece][h = k half][column 128][8 bf16]  (16 B per lane)
//   lane (r = l & 31, h = l >> 5) of the MFMA holds A[row r][k = 8h + j], B[k = 8h + j][col r], j = 0..7
//   A columns = the tile's queries; B slot n*32 + j (accumulator n, MFMA column j) = cell 4j + n of the patch.
// A K tile (2 k-steps for NP = 1, 1 for NP = 3) is contiguous in the panel and staged like the
// fp32 kernel's, in 1 KiB DMA pieces.
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// partial maxima of |x| of both feature maps: grid (MAXPARTS, 2), out[map * MAXPARTS + block]
__global__ __launch_bounds__(256)
void corr_maxabs_kernel(const float* __restrict__ f1, const float* __restrict__ f2, size_t n, float* __restrict__ out,
                        const float* __restrict__ f3, size_t n3)
{
    __shared__ float red[4];
    const float* __restrict__ x = blockIdx.y == 2 ? f3 : blockIdx.y ? f2 : f1;
    if (blockIdx.y == 2) n = n3;
    float m = 0.f;
    const size_t n4 = (((uintptr_t)x & 15) == 0) ? n >> 2 : 0;           // 16-byte loads when aligned
    constexpr size_t STRIDE = (size_t)MAXPARTS * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * STRIDE < n4; i += 4 * STRIDE) {                       // four loads in flight per thread
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((const float4*)x)[i + j * STRIDE];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
    }
    for (; i < n4; i += STRIDE) {
        const float4 v = ((const float4*)x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)MAXPARTS * 256)
        m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.y * MAXPARTS + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// partial maxima of two equally sized maps (out: [2][MAXPARTS]); also the range pass of the on-the-fly path's pack
int launch_maxabs(const float* f1, const float* f2, size_t n, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(corr_maxabs_kernel, dim3(MAXPARTS, 2), dim3(256), 0, s, f1, f2, n, out, (const float*)nullptr, (size_t)0);
    return check_launch("corr_maxabs_kernel");
}

// two equally sized maps and a third array of its own length in one launch (out: [3][MAXPARTS])
int launch_maxabs3(const float* f1, const float* f2, size_t n, const float* f3, size_t n3, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(corr_maxabs_kernel, dim3(MAXPARTS, 3), dim3(256), 0, s, f1, f2, n, out, f3, n3);
    return check_launch("corr_maxabs_kernel");
}

// partial maxima of one array (out: [MAXPARTS]); the range pass of the on-the-fly backward's d(out)
int launch_maxabs1(const float* x, size_t n, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(corr_maxabs_kernel, dim3(MAXPARTS, 1), dim3(256), 0, s, x, x, n, out, (const float*)nullptr, (size_t)0);
    return check_launch("corr_maxabs_kernel");
}

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float x)
{
    unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// grid (tilesM + tilesN, PACK_KSPLIT, B); thread = one column x (Dpad/8/PACK_KSPLIT/2) groups of 8 k
template <int NP, bool F16>
__global__ __launch_bounds__(256)
void corr_pack_bf16_kernel(PackArgs p)
{
    const int N = p.H * p.W;
    const int c = threadIdx.x & 127, gp = threadIdx.x >> 7;
    const int b = blockIdx.z;
    int tile = blockIdx.x;
    const bool second = tile >= p.tilesM;
    int t = -1;
    if (!second) {
        const int q = tile * 128 + c;
        if (q < N) t = q;
    } else {
        tile -= p.tilesM;
        const int PW = patches_w(p.W);
        const int m = (c & 31) * 4 + (c >> 5);            // slot = accumulator*32 + MFMA column <-> cell 4*column + accumulator
        int patch, cell;
        paired_cell(p.pair_from, tile, m, patch, cell);
        const int n = cell >> 5, j = cell & 31;
        const int dx = (j & 1) | (((j >> 3) & 1) << 1) | ((j >> 4) << 2);
        const int dy = ((j >> 1) & 1) | (((j >> 2) & 1) << 1);
        const int y = (patch / PW) * PATCH_H + (n >> 1) * 4 + dy;
        const int x = (patch % PW) * PATCH_W + (n & 1) * 8 + dx;
        if (y < p.H && x < p.W) t = y * p.W + x;
    }
    const float* __restrict__ src = (second ? p.f2 : p.f1) + (size_t)b * p.D * N + (t >= 0 ? t : 0);
    uint4* __restrict__ dst = (uint4*)(second ? p.f2p : p.f1p) +
                              ((size_t)b * (second ? p.tilesN : p.tilesM) + tile) * (p.Dpad / 8) * NP * 128 + c;
    const int groups = p.Dpad / 8, per = groups / PACK_KSPLIT;                // Dpad is a multiple of 64
    float rscale = 1.0f;
    if (F16) rscale = ldexpf(1.0f, range_exponent(wave_max_of_partials(p.maxpart + (second ? MAXPARTS : 0))));
    for (int g = blockIdx.y * per + gp; g < (blockIdx.y + 1) * per; g += 2) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = src[(size_t)min(g * 8 + j, p.D - 1) * N];   // always a valid address: all 8 in flight
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (t >= 0 && g * 8 + j < p.D) ? x[j] * rscale : 0.f;
#pragma unroll
        for (int piece = 0; piece < NP; ++piece) {
            unsigned short h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (F16) {
                    const _Float16 hv = (_Float16)x[j];                       // RNE
                    h[j] = __builtin_bit_cast(unsigned short, hv);
                    x[j] -= (float)hv;                                        // exact: the next piece's input
                } else {
                    h[j] = f32_to_bf16_rne(x[j]);
                    x[j] -= __uint_as_float((unsigned)h[j] << 16);
                }
            }
            uint4 w;
            w.x = h[0] | ((unsigned)h[1] << 16); w.y = h[2] | ((unsigned)h[3] << 16);
            w.z = h[4] | ((unsigned)h[5] << 16); w.w = h[6] | ((unsigned)h[7] << 16);
            dst[(size_t)(((g >> 1) * NP + piece) * 2 + (g & 1)) * 128] = w;   // [k-step][piece][h][column][8]
        }
    }
}

#ifndef RC_BF16_WAVES
#define RC_BF16_WAVES 3     // waves per SIMD the 16-bit kernels' registers are bounded for (lab knob)
#endif

template <bool EXACT_MUL, int NP, bool F16, int VOL = 0>
__global__ __launch_bounds__(256, RC_BF16_WAVES)
void corr_build_bf16_kernel(BuildArgs p)
{
    // The B tile (one patch of targets) is shared by the four waves and goes through LDS (two
    // buffers, direct-to-LDS DMA).  A wave's A fragments (its own 32 queries) are private, so they
    // skip LDS: one coalesced 16-byte global load per lane and fragment, fetched one K tile ahead.
    constexpr int KS = NP == 1 ? 4 : 2;                 // k-steps per K tile
    static_assert(NP <= 3, "pieces");
    constexpr int OPF = NP * KS * 1024;                 // floats of B per stage (4 KiB per piece and k-step)
    __shared__ __attribute__((aligned(16))) float smem[2 * OPF];

    int b, tm, tn;
    block_to_tile(p, b, tm, tn);
    const int m0 = tm * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, l31 = lane & 31;
    const size_t panel = (size_t)p.Dpad * 64 * NP;                            // floats per panel
    const float* __restrict__ pb = p.f2p + ((size_t)b * p.tilesN + tn) * panel;
    // this lane's slot in every A fragment: ((k-step * NP + piece) * 2 + kh) * 128 + row
    const float4* __restrict__ pa = (const float4*)(p.f1p + ((size_t)b * p.tilesM + tm) * panel) +
                                    kh * 128 + wave * 32 + l31;

    auto stage_b = [&](int kt, float* lb) {
        const float* gb = pb + (size_t)kt * OPF;
#pragma unroll
        for (int i = 0; i < NP * KS; ++i) {
            const int piece = wave * NP * KS + i;                              // 1 KiB DMA pieces
            glds16(gb + piece * 256 + lane * 4, lb + piece * 256);
        }
    };
    float4 a_nxt[KS * NP], a_cur[KS * NP];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < KS * NP; ++i) a_nxt[i] = pa[(size_t)(kt * KS * NP + i) * 256];
    };

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

#if defined(RC_LAB_SKIP_MAINLOOP)      // tools/build_lab.hip only: time the epilogue alone
    const int nk = 1;
#else
    const int nk = p.Dpad / (16 * KS);
#endif
    stage_b(0, smem);
    load_a(0);
    // restrict-qualified parameters: after inlining, the DMA destination (`other`) and the ds_read
    // source (`tile`) carry distinct alias scopes
    auto ktile = [&](int kt, const float* __restrict__ tile, float* __restrict__ other) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KS * NP; ++i) a_cur[i] = a_nxt[i];
#if defined(RC_LAB_NO_BSTAGE) || defined(RC_LAB_NO_ALOAD)   // tools/build_lab.hip only: ablations
        if (kt + 1 < nk) {
#if !defined(RC_LAB_NO_BSTAGE)
            stage_b(kt + 1, other);
#endif
#if !defined(RC_LAB_NO_ALOAD)
            load_a(kt + 1);
#endif
        }
#else
        if (kt + 1 < nk) { stage_b(kt + 1, other); load_a(kt + 1); }
#endif
        // B fragment (k-step s, piece c): 16-byte slot (((s * NP + c) * 2 + kh) * 128 + column)
        const float4* lb = (const float4*)tile + kh * 128 + l31;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float4 a[NP];
#pragma unroll
            for (int c = 0; c < NP; ++c) a[c] = a_cur[s * NP + c];
            // piece of B outermost, small products first: lo*hi | mid*mid, mid*hi | hi*lo, hi*mid, hi*hi
#pragma unroll
            for (int cb = NP - 1; cb >= 0; --cb) {
                float4 bb[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) bb[n] = lb[(s * NP + cb) * 256 + n * 32];
#pragma unroll
                for (int ca = NP - 1 - cb; ca >= 0; --ca)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (F16) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, a[ca]), __builtin_bit_cast(f16x8, bb[n]), acc[n], 0, 0, 0);
                        else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[ca]), __builtin_bit_cast(bf16x8, bb[n]), acc[n], 0, 0, 0);
                    }
            }
        }
    };
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        ktile(kt, smem + buf * OPF, smem + (buf ^ 1) * OPF);
    }
#if defined(RC_LAB_SKIP_EPILOGUE)      // tools/build_lab.hip only: time the main loop alone
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(acc[n][i]));
    return;
#endif
    build_epilogue<EXACT_MUL, VOL>(p, acc, b, tn, m0, wave, lane);
}

// levels >= 4: 2x2 mean of the previous level (rare: RAFT uses 4 levels); output row-major
__global__ void pool2x2_kernel(const float* __restrict__ in, float* __restrict__ out, size_t M,
                               LevelGeom gi, LevelGeom go)
{
    const size_t total = M * (size_t)go.h * go.w;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int wo = (int)(e % go.w);
        const size_t r = e / go.w;
        const int ho = (int)(r % go.h);
        const size_t m = r / go.h;
        const float* s = in + m * gi.slab;
        const float a = s[level_pos(gi, 2 * ho, 2 * wo)], b = s[level_pos(gi, 2 * ho, 2 * wo + 1)];
        const float c = s[level_pos(gi, 2 * ho + 1, 2 * wo)], d = s[level_pos(gi, 2 * ho + 1, 2 * wo + 1)];
        out[m * go.slab + level_pos(go, ho, wo)] = ((a + b) + (c + d)) * 0.25f;
    }
}

size_t build_workspace_bytes(int B, int D, int H, int W)
{
    const int N = H * W;
    const size_t Dpad = (size_t)(D + 63) / 64 * 64;
    const size_t tilesM = (N + BM - 1) / BM, tilesN = (size_t)patches_h(H) * patches_w(W);
    // operand panels (fp32: 4 B, bf16: 2 B, 2 x fp16: 4 B, 3 x bf16: 6 B per element) + partial maxima
    return (size_t)B * Dpad * 128 * (tilesM + tilesN) * 6 + 2 * MAXPARTS * sizeof(float);
}

int launch_build_pyramid(const float* f1, const float* f2, int B, int D, int H, int W, int L,
                         const LevelSet& lv, void* workspace, size_t workspace_bytes, int precision, hipStream_t s, int vol)
{
    if (vol != RAFTCORR_VOL_FP32 && L > 4)
        return fail(RAFTCORR_EUNSUPPORTED, "build_pyramid: a 16-bit volume has at most 4 levels (got %d)", L);
    const bool bf16 = precision != RAFTCORR_PREC_FP32;          // 16-bit pieces on the bf16/fp16 matrix cores
    const bool f16 = precision == RAFTCORR_PREC_FP16X2;
    const int np = precision == RAFTCORR_PREC_BF16X3 ? 3 : f16 ? 2 : 1;
    if (workspace_bytes < build_workspace_bytes(B, D, H, W) || !workspace)
        return fail(RAFTCORR_EINVAL, "build_pyramid: workspace too small (%zu < %zu bytes)",
                    workspace_bytes, build_workspace_bytes(B, D, H, W));
    BuildArgs a;
    a.B = B; a.N = H * W; a.L = L < 4 ? L : 4;
    a.Dpad = (D + 63) / 64 * 64;             // multiple of every K tile and of the pack kernels' k split
    a.tilesM = (a.N + BM - 1) / BM;
    a.npatch = patches_h(H) * patches_w(W);
    a.tilesN = a.npatch;
    a.PW = patches_w(W);
    if (H % PATCH_H >= 1 && H % PATCH_H <= 4) {              // bottom patch row: two upper halves per panel (paired_cell())
        a.pair_from = (patches_h(H) - 1) * a.PW;
        a.tilesN = a.pair_from + (a.PW + 1) / 2;
    }
#if defined(RC_LAB)                    // tools/ only
    if (getenv("RC_NO_PAIRS")) { a.pair_from = 0x7fffffff; a.tilesN = a.npatch; }
#endif
    // bf16 on volumes far beyond the 256 MB infinity cache is bound by HBM write efficiency: one
    // query tile at a time per XCD keeps each row's concurrently written segment long (measured at
    // 1080p: 1.61 ms vs 2.35 ms with groups of 8; at Sintel size the grouping of 8 is the faster one)
    a.group_m = (bf16 && (double)B * a.N * (double)lv.g[0].slab * (vol ? 2.0 : 4.0) > 1.0e9) ? 1 : GROUP_M;
#if defined(RC_LAB)                    // tools/ only (tools/Makefile passes -DRC_LAB): never in libraftcorr.so
    if (const char* e = getenv("RC_GROUP_M")) a.group_m = atoi(e) > 0 ? atoi(e) : GROUP_M;
#endif
    a.nblk_per_b = a.tilesM * a.tilesN;
    float* f1p = (float*)workspace;
    float* f2p = f1p + (size_t)B * a.tilesM * a.Dpad * (bf16 ? 64 * np : 128);
    int rc;
    a.maxpart = nullptr;
    if (bf16) {
        PackArgs pk{f1, f2, f1p, f2p, D, a.Dpad, H, W, a.tilesM, a.tilesN, nullptr, a.pair_from};
        const dim3 pgrid(a.tilesM + a.tilesN, PACK_KSPLIT, B);
        if (f16) {          // range pass: max|x| of both maps -> power-of-two scales (pack) and the unscale (epilogue)
            float* mp = (float*)((char*)workspace + build_workspace_bytes(B, D, H, W) - 2 * MAXPARTS * sizeof(float));
            hipLaunchKernelGGL(corr_maxabs_kernel, dim3(MAXPARTS, 2), dim3(256), 0, s, f1, f2, (size_t)B * D * a.N, mp, (const float*)nullptr, (size_t)0);
            rc = check_launch("corr_maxabs_kernel");
            if (rc) return rc;
            pk.maxpart = mp; a.maxpart = mp;
            hipLaunchKernelGGL((corr_pack_bf16_kernel<2, true>), pgrid, dim3(256), 0, s, pk);
        } else if (np == 3) hipLaunchKernelGGL((corr_pack_bf16_kernel<3, false>), pgrid, dim3(256), 0, s, pk);
        else                hipLaunchKernelGGL((corr_pack_bf16_kernel<1, false>), pgrid, dim3(256), 0, s, pk);
        rc = check_launch("corr_pack_bf16_kernel");
    } else {
        rc = launch_pack_operands(f1, f2, f1p, f2p, B, D, a.Dpad, H, W, a.tilesM, a.tilesN, s, a.pair_from);
    }
    if (rc) return rc;
    a.f1p = f1p; a.f2p = f2p;
    for (int l = 0; l < 4; ++l) {
        a.lvl[l] = l < L ? lv.ptr[l] : nullptr;
        a.slab[l] = l < L ? (long long)lv.g[l].slab : 0;
        if ((long long)32 * a.slab[l] * 4 > 0x7fffffffLL)
            return fail(RAFTCORR_EUNSUPPORTED, "build_pyramid: level %d slab too large", l);
    }
    const float sq = sqrtf((float)D);
    a.sqrt_d = sq;
    a.inv_sqrt_d = 1.0f / sq;
    int e; const float mant = frexpf(sq, &e);
    const bool exact = (mant == 0.5f) && (sq * sq == (float)D);   // sqrt(D) is a power of two
    const long long nblk = (long long)a.nblk_per_b * B;
    if (nblk > 0x7fffffffLL) return fail(RAFTCORR_EUNSUPPORTED, "build_pyramid: grid too large");
    dim3 grid((unsigned)nblk), block(256);
#define RC_LAUNCH_VOL(KERNEL, ...)                                                                           \
    do {                                                                                                     \
        if (vol == RAFTCORR_VOL_FP16)      hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 1>), grid, block, 0, s, a); \
        else if (vol == RAFTCORR_VOL_BF16) hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 2>), grid, block, 0, s, a); \
        else                               hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 0>), grid, block, 0, s, a); \
    } while (0)
    if (bf16) {
        if (f16) {
            if (exact) RC_LAUNCH_VOL(corr_build_bf16_kernel, true, 2, true);
            else       RC_LAUNCH_VOL(corr_build_bf16_kernel, false, 2, true);
        } else if (np == 3) {
            if (exact) RC_LAUNCH_VOL(corr_build_bf16_kernel, true, 3, false);
            else       RC_LAUNCH_VOL(corr_build_bf16_kernel, false, 3, false);
        } else {
            if (exact) RC_LAUNCH_VOL(corr_build_bf16_kernel, true, 1, false);
            else       RC_LAUNCH_VOL(corr_build_bf16_kernel, false, 1, false);
        }
    } else {
        if (exact) RC_LAUNCH_VOL(corr_build_f32_kernel, true, BK, WPS);
        else       RC_LAUNCH_VOL(corr_build_f32_kernel, false, BK, WPS);
    }
#undef RC_LAUNCH_VOL
    rc = check_launch(bf16 ? "corr_build_bf16_kernel" : "corr_build_f32_kernel");
    if (rc) return rc;
    for (int l = 4; l < L; ++l) {
        const size_t M = (size_t)B * a.N;
        const size_t total = M * lv.g[l].h * lv.g[l].w;
        if (total == 0) continue;
        const unsigned g = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(pool2x2_kernel, dim3(g), dim3(256), 0, s, lv.ptr[l - 1], lv.ptr[l], M,
                           lv.g[l - 1], lv.g[l]);
        rc = check_launch("pool2x2_kernel");
        if (rc) return rc;
    }
    return RAFTCORR_OK;
}


// ---------------------------------------------------------------------------------------
// Allocation-free variant (raftcorr_build_pyramid_noalloc, include/raftcorr.h): the operand panels
// and the fp16 range pass' partial maxima -- the scratch that the other build entry points take as a
// caller-allocated workspace -- live in a file-scope device array instead, so callers that cannot
// allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc) need no workspace.
// Every GPU the library runs on holds its own copy of the array (at its own address) until the
// process ends.  In host code the array's NAME is only a stand-in: its device address is looked up
// once per device with hipGetSymbolAddress (that device current) and kept; host code uses that
// pointer everywhere (kernel arguments), never the name.
// Nothing in the scratch has to start at zero: the pack kernels write the panels in full (zero
// padding included) and the range pass writes all of its partial maxima, so a call does not see
// what the previous call left.  Calls that overlap in time on one GPU (two streams) share the
// array and race: serialise them.
// ---------------------------------------------------------------------------------------
constexpr size_t NOALLOC_SCRATCH_BYTES = RAFTCORR_NOALLOC_MAX_WORKSPACE;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

constexpr int NOALLOC_MAX_DEVICES = 64;
static std::atomic<void*> g_scratch_ptr[NOALLOC_MAX_DEVICES];   // per device, null until its first call

static int noalloc_scratch(void** out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "build_pyramid_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES)
        return fail(RAFTCORR_EUNSUPPORTED, "build_pyramid_noalloc: device %d >= %d", dev, NOALLOC_MAX_DEVICES);
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {           // first call on this device (two threads racing here look up the same address)
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(RAFTCORR_ELAUNCH, "build_pyramid_noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev,
                        hipGetErrorString(e));
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return RAFTCORR_OK;
}

int launch_build_pyramid_noalloc(const float* f1, const float* f2, int B, int D, int H, int W, int L,
                                 const LevelSet& lv, int precision, hipStream_t s)
{
    const size_t need = build_workspace_bytes(B, D, H, W);
    if (need > NOALLOC_SCRATCH_BYTES)
        return fail(RAFTCORR_EUNSUPPORTED,
                    "build_pyramid_noalloc: B=%d D=%d H=%d W=%d needs %zu bytes of scratch, the built-in array holds %zu "
                    "(use raftcorr_build_pyramid_prec with a caller-allocated workspace)", B, D, H, W, need,
                    NOALLOC_SCRATCH_BYTES);
    void* ptr = nullptr;
    const int rc = noalloc_scratch(&ptr);
    if (rc) return rc;
    return launch_build_pyramid(f1, f2, B, D, H, W, L, lv, ptr, NOALLOC_SCRATCH_BYTES, precision, s, RAFTCORR_VOL_FP32);
}

}  // namespace raftcorr
