// The two feature-gradient contractions of the deferred backward on PRE-SPLIT operands (training path, BASELINE
// configs[3]; what autograd runs behind core/corr.py:59,61 in the reference: the two GEMMs of bmm's backward).
//
// EXPERIMENT, NOT ADOPTED (tools/experiments/README.md has the measurements): the premise below -- that the shipped
// gemm_bf16x3_kernel is bound by its in-flight conversion -- turned out wrong; this kernel is correct and no faster.
//
// gemm_bf16x3_kernel (build_bwd.hip) reads fp32 operands and splits them into two fp16 pieces while it stages them:
// at Things size (B = 6, 50 x 90) that is ~700 conversion instructions per thread and K tile against 48 MFMAs, and
// every workgroup converts the SMALL operand (the feature map) again.  Here the split happens once, where the values
// are produced:
//   * the folded level-0 gradient dC0 is written by corr_splat_fold_kernel as fp16 pieces (same bytes as fp32),
//     scaled by a power of two from an a-priori bound of |dC0| (corr_bound_kernel below);
//   * the feature maps are packed once per call (pk16_rows_kernel, pk16_gather_cols_kernel).
// The GEMM then does no conversion at all: operands go global -> LDS by DMA (global_load_lds_dwordx4), three stages
// deep, fragments come out of LDS with ds_read_b128 (k-contiguous operands) or ds_read_b64_tr_b16 (the gradient
// volume read along its rows while the contraction runs down its columns), three products per k step
// (lo*hi + hi*lo + hi*hi), exact power-of-two unscale in the epilogue.  No split-K planes, no atomics: one
// workgroup owns a 256 x 128 tile of the result and the whole K range, results are bit-reproducible.
//
// "pk16" row format (PK_BLK = 32 columns per block): block j of a row is 128 bytes = one cache line,
//     [ 32 x fp16 hi of columns 32 j .. 32 j + 31 | 32 x fp16 lo of the same columns ],
// so a K tile of 32 of a k-contiguous operand is exactly one line per row (both pieces), and a row of pk16 has the
// byte size of the fp32 row it replaces.
#include "../../../raft_amd/csrc/common.h"
#include "gemm16.h"
#include <algorithm>
#include <stdlib.h>

namespace raftcorr {

// (declarations that would live in common.h had this kernel gone into the library)
int launch_pk16_rows(const float* src, size_t rows, int cols, int colsp, const float* maxpart, void* dst, hipStream_t s,
                     int tile_rows = 0);
int launch_pk16_gather_cols(const float* src, int BD, int H, int W, const float* maxpart, void* dst, hipStream_t s);
int launch_corr_bound(const float* const* maxima, const int* counts, int K, float factor, float* out, hipStream_t s);
int launch_gemm_pk16(const void* A, const void* Bm, float* C, int M, int N, int K, int b_rows_are_k,
                     int colsA, int colsB, int rowsB, int ldc, size_t sA_bytes, size_t sB_bytes, size_t sC,
                     int batch, const float* maxA, const float* maxB, hipStream_t s, int b_tile_major = 0);

typedef float f32x16g __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
typedef short s16x4p __attribute__((ext_vector_type(4)));

namespace {

// ---------------------------------------------------------------------------------------------
// packing of the small operands

struct PackRowsArgs {
    const float* src;       // [rows][cols]
    uint2* dst;             // pk16 [rows][colsp], colsp a multiple of 32; columns >= cols are zeros
    const float* maxpart;   // [128] partial maxima of |src|
    size_t rows;
    int cols, colsp;
    int tile_rows;          // > 0: tile-major destination (pk16_tile_u4), images of tile_rows rows
};

// one thread = 4 consecutive columns: 8 bytes of hi, 8 bytes of lo
__global__ __launch_bounds__(256)
void pk16_rows_kernel(PackRowsArgs p)
{
    const float sc = ldexpf(1.0f, pk16_range_exp(p.maxpart));
    const int qpr = p.colsp >> 2;                            // quads per row
    const size_t total = p.rows * (size_t)qpr;
    const bool vec = (p.cols & 3) == 0 && ((uintptr_t)p.src & 15) == 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / qpr;
        const int c = (int)(e - row * qpr) * 4;
        const float* s = p.src + row * (size_t)p.cols + c;
        float v[4];
        if (vec && c + 4 <= p.cols) { const float4 t = *(const float4*)s; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = c + j < p.cols ? s[j] : 0.f;
        }
        uint2 hi, lo;
        pk16_split4(v, sc, hi, lo);
        uint2* d = p.dst + row * (size_t)(p.colsp >> 1) + pk16_u2(c);
        if (p.tile_rows > 0) {
            const size_t img = row / p.tile_rows;
            const int q = (int)(row - img * p.tile_rows), qt = (p.tile_rows + 127) >> 7, nblk = p.colsp >> 5;
            d = p.dst + (img * qt * nblk * 128 + pk16_tile_line(q, c >> 5, nblk)) * 16 + ((c & 31) >> 2);
        }
        d[0] = hi;
        d[8] = lo;
    }
}

struct PackColsArgs {
    const float* src;       // [BD][H * W]
    uint2* dst;             // pk16 [BD][S0] in volume column order (layout.h), padding columns zero
    const float* maxpart;
    int BD, H, W, S0;
};

// dst[bd][p] = src[bd][t(p)]: one thread = the 4 consecutive columns p = 4 e .. 4 e + 3 (one 2x2 block of targets)
__global__ __launch_bounds__(256)
void pk16_gather_cols_kernel(PackColsArgs p)
{
    const float sc = ldexpf(1.0f, pk16_range_exp(p.maxpart));
    const int PW = patches_w(p.W), qpr = p.S0 >> 2;
    const size_t total = (size_t)p.BD * qpr;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t bd = e / qpr;
        const int c = (int)(e - bd * qpr) * 4;
        const int patch = c >> 7, n = (c >> 5) & 3, j = c & 31;             // j & 3 = 0: dx bit 0 and dy bit 0 run inside
        const int dxh = (((j >> 3) & 1) << 1) | ((j >> 4) << 2), dyh = ((j >> 2) & 1) << 1;
        const int y0 = (patch / PW) * PATCH_H + (n >> 1) * 4 + dyh, x0 = (patch % PW) * PATCH_W + (n & 1) * 8 + dxh;
        const float* s = p.src + bd * (size_t)(p.H * p.W);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = y0 + (i >> 1), x = x0 + (i & 1);
            v[i] = (y < p.H && x < p.W) ? s[y * p.W + x] : 0.f;
        }
        uint2 hi, lo;
        pk16_split4(v, sc, hi, lo);
        uint2* d = p.dst + bd * (size_t)(p.S0 >> 1) + pk16_u2(c);
        d[0] = hi;
        d[8] = lo;
    }
}

// ---------------------------------------------------------------------------------------------
// a-priori bound of |dC0| from the records' maxima:  every cell of a level's gradient row receives, per lookup, the
// d(out) cells of at most four window taps with bilinear weights that add up to at most 1, so
//     |dC0[q][p]| <= ( sum_k max|d(out)_k| ) * ( sum_l 4^-l ) / sqrt(D).
// out[0 .. 127] = that bound (x 1 + 2^-10 for the rounding of the sums); block 256, grid 1.
struct BoundArgs {
    const float* mx[32];
    int n[32];
    int K;
    float factor;
    float* out;
};

__global__ __launch_bounds__(256)
void corr_bound_kernel(BoundArgs p)
{
    __shared__ float wm[4];
    __shared__ float total;
    if (threadIdx.x == 0) total = 0.f;
    for (int k = 0; k < p.K; ++k) {
        float m = 0.f;
        for (int i = threadIdx.x; i < p.n[k]; i += 256) {
            const float v = p.mx[k][i];
            m = (v == v) ? fmaxf(m, fabsf(v)) : INFINITY;    // a NaN anywhere: no finite bound (the range exponent falls back to 0)
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) total += fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        __syncthreads();
    }
    if (threadIdx.x < 128) p.out[threadIdx.x] = total * p.factor;
}

// ---------------------------------------------------------------------------------------------
// the GEMM:  C[m][n] = unscale * sum_k A[m][k] * Bm(k, n)
//   A: pk16 rows m, k contiguous.   BN = false: B pk16 rows n, k contiguous.   BN = true: B pk16 rows k, n contiguous.
// Tile TM = 64 WM (256 or 128) x 128, K tile 32; WM x 2 waves, each 64 x 64 = 2 x 2 accumulators of
// v_mfma_f32_32x32x16_f16.  LDS stage = A image + B image, three stages.
//   k-contiguous image: [row][8 slots of 16 B] = [row][hi k 0..7 | 8..15 | 16..23 | 24..31 | lo ...]; slot x of row r
//       lives at r * 8 + (x ^ ((r >> 1) & 7)): the 16 rows a quarter-wave reads (fixed x) cover all 64 banks, and the
//       DMA writes a row's 8 slots from one 128-byte line.
//   n-contiguous image: [k row][32 slots] = [k row][column block 0..3][hi | lo][4 slots]; slot x of row k lives at
//       k * 32 + (x ^ (4 * (k & 3))): the four rows a transposed read touches fall into different bank quarters.
struct Gemm16Args {
    const uint4* A; const uint4* B; float* C;
    int M, N, T;                 // T: K tiles of 32
    unsigned pa, pb;             // row pitches in 16-byte units
    int rowsA, rowsB;            // rows that exist (sources are clamped to the last one)
    int ldc;
    size_t sA, sB, sC;           // image strides (16-byte units; floats for C)
    int ntiles, total, per;      // N tiles per image; tiles in all; tiles per XCD
    int nblkB;                   // > 0: B is tile-major (pk16_tile_line) with nblkB column blocks per row; pb/rowsB unused
    const float* maxA; const float* maxB;
};

__device__ __forceinline__ void dma16(const uint4* gsrc, uint4* lds_chunk)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_chunk, 16, 0, 0);
}

template <int WM, bool BN, int NST>
__global__ __launch_bounds__(WM * 128)
void gemm_pk16_kernel(Gemm16Args p)
{
    constexpr int TM = WM * 64, NW = WM * 2;
    constexpr int A_U4 = TM * 8, B_U4 = 1024, STG = A_U4 + B_U4;
    constexpr int NA = (TM / 8) / NW, NB = 16 / NW;          // DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];           // [NST][STG]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // tile of this workgroup: ids go round-robin over the 8 XCDs; each XCD takes a contiguous run of (image, N tile),
    // i.e. at most two images' small operands pass through its L2
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, tile = xcd * p.per + idx;
    if (idx >= p.per || tile >= p.total) return;
    const int bz = tile / p.ntiles, nt = tile - bz * p.ntiles;
    const int m0 = blockIdx.y * TM, n0 = nt * 128;

    // partial maxima behind the unscale factor: fetched now, used in the epilogue
    const float mpa = fmaxf(p.maxA[lane], p.maxA[lane + 64]), mpb = fmaxf(p.maxB[lane], p.maxB[lane + 64]);

    const uint4* __restrict__ Ab = p.A + (size_t)bz * p.sA;
    const uint4* __restrict__ Bb = p.B + (size_t)bz * p.sB;
    // DMA sources of this lane, K tile 0
    const uint4* asrc[NA];
    const uint4* bsrc[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int rl = (wave * NA + j) * 8 + (lane >> 3);
        asrc[j] = Ab + (size_t)min(m0 + rl, p.rowsA - 1) * p.pa + ((lane & 7) ^ ((rl >> 1) & 7));
    }
    int bk[NB];                                              // BN: k row of the lane inside a tile
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (BN) {
            bk[j] = (wave * NB + j) * 2 + (lane >> 5);
            const int x = (lane & 31) ^ (4 * (bk[j] & 3));
            bsrc[j] = p.nblkB ? Bb + (size_t)(4 * nt + (x >> 3)) * 1024 + (x & 7) : Bb + (size_t)nt * 32 + x;
        } else {
            const int rl = (wave * NB + j) * 8 + (lane >> 3);
            bk[j] = 0;
            const int x = (lane & 7) ^ ((rl >> 1) & 7);
            bsrc[j] = p.nblkB ? Bb + ((size_t)nt * p.nblkB * 128 + rl) * 8 + x
                              : Bb + (size_t)min(n0 + rl, p.rowsB - 1) * p.pb + x;
        }
    }
    auto stage = [&](int t, uint4* buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) dma16(asrc[j] + (size_t)t * 8, buf + (wave * NA + j) * 64);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint4* s;
            if (p.nblkB) {                                   // tile-major: 128 rows x 128 B per (row tile, column block)
                const int k = min(t * 32 + bk[j], p.rowsB - 1);
                s = BN ? bsrc[j] + ((size_t)(k >> 7) * p.nblkB * 128 + (k & 127)) * 8 : bsrc[j] + (size_t)t * 1024;
            } else {
                s = BN ? bsrc[j] + (size_t)min(t * 32 + bk[j], p.rowsB - 1) * p.pb : bsrc[j] + (size_t)t * 8;
            }
            dma16(s, buf + A_U4 + (wave * NB + j) * 64);
        }
    };

    f32x16g acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int key = (l31 >> 1) & 7;
    const int qq = (l31 & 15) >> 2, pp = l31 & 3, gh = l31 >> 4;
    auto tr = [](const short* a) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4p*)a);
    };
    struct Frag { f16x8p ah[2], al[2], bh[2], bl[2]; };
    auto load = [&](const uint4* cur, int st, Frag& f) {     // fragments of k step st (16 k) of the stage
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const uint4* row = cur + (wm * 64 + a * 32 + l31) * 8;
            f.ah[a] = __builtin_bit_cast(f16x8p, row[(2 * st + kh) ^ key]);
            f.al[a] = __builtin_bit_cast(f16x8p, row[(4 + 2 * st + kh) ^ key]);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (BN) {
                // k rows 16 st + 8 kh + qq (+ 4); column block wn * 2 + b; slots 2 gh + (pp >> 1) of hi / lo
                const short* base = (const short*)(cur + A_U4) + (16 * st + 8 * kh + qq) * 256 + (pp & 1) * 4;
                const int xh = (wn * 2 + b) * 8 + 2 * gh + (pp >> 1);
                const int sh = (xh ^ (4 * qq)) * 8, sl = ((xh + 4) ^ (4 * qq)) * 8;
                const s16x4p h0 = tr(base + sh), h1 = tr(base + 4 * 256 + sh);
                const s16x4p l0 = tr(base + sl), l1 = tr(base + 4 * 256 + sl);
                f.bh[b] = __builtin_bit_cast(f16x8p, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                f.bl[b] = __builtin_bit_cast(f16x8p, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                const uint4* row = cur + A_U4 + (wn * 64 + b * 32 + l31) * 8;
                f.bh[b] = __builtin_bit_cast(f16x8p, row[(2 * st + kh) ^ key]);
                f.bl[b] = __builtin_bit_cast(f16x8p, row[(4 + 2 * st + kh) ^ key]);
            }
        }
    };
    auto mma = [&](const Frag& f) {                          // small terms first; the four accumulators take turns inside each term
#if !defined(RC_LAB_G16_NOMFMA)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[a], f.bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[a], f.bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[a], f.bh[b], acc[a][b], 0, 0, 0);
#endif
    };
    // Schedule of one K tile t (its first-half fragments f0 are in registers already):
    //     read the second-half fragments f1 | products of f0 | wait + barrier | DMA of tile t + 3 | read f0 of tile t + 1 |
    //     products of f1
    // The LDS reads of a half run under the matrix work of the other half.  At the barrier every wave has its tile-t
    // fragments in registers (lgkmcnt(0)) and its share of tile t + 1 in LDS (vector-memory operations complete in order:
    // all but the NA + NB of tile t + 2), so the stage of tile t is free for tile t + 3: three tiles in flight on three
    // stages.  The stages are restrict-qualified parameters of ONE function so that hipcc sees the DMA destination and
    // the next tile's fragment source as different objects and leaves the DMA in flight.
    const int T = p.T;
    Frag f0, f1;
    constexpr int PW = NA + NB;                              // DMA instructions per wave and tile (6 or 8)
    auto wait_all_but = [](int tiles) {                      // s_waitcnt vmcnt(tiles * PW) lgkmcnt(0); s_barrier
        if (tiles <= 0)      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (tiles == 1) {
            if (PW == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else         asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            if (PW == 6) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else         asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };
    auto step = [&](int t, uint4* cur, const uint4* nxt) {
        load(cur, 1, f1);
        mma(f0);
        // (a use of f1 in front of the hand-written wait: hipcc then knows that these reads have completed and does not
        // wait for the NEXT tile's reads, issued after the barrier, in front of the products of f1)
        __builtin_amdgcn_sched_barrier(0);                   // (the products stay in front of that use)
        asm volatile("" :: "v"(f1.ah[0]), "v"(f1.ah[1]), "v"(f1.al[0]), "v"(f1.al[1]), "v"(f1.bh[0]), "v"(f1.bh[1]),
                     "v"(f1.bl[0]), "v"(f1.bl[1]));
        // tile t + 1 must have landed; the tiles t + 2 .. t + NST - 1 that exist may stay in flight
        wait_all_but(min(max(T - (t + 2), 0), NST - 2));
#if !defined(RC_LAB_G16_NODMA)
        if (t + NST < T) stage(t + NST, cur);
#endif
        if (t + 1 < T) load(nxt, 0, f0);
        mma(f1);
    };
    auto round3 = [&](int t, uint4* __restrict__ b0, uint4* __restrict__ b1, uint4* __restrict__ b2) {
        step(t, b0, b1);
        if (t + 1 < T) step(t + 1, b1, b2);
        if (t + 2 < T) step(t + 2, b2, b0);
    };
    auto round2 = [&](int t, uint4* __restrict__ b0, uint4* __restrict__ b1) {
        step(t, b0, b1);
        if (t + 1 < T) step(t + 1, b1, b0);
    };
#pragma unroll
    for (int i = 0; i < NST; ++i)
        if (i < T) stage(i, smem + i * STG);
    // tile 0 landed (the other min(T, NST) - 1 tiles stay in flight)
    if (min(T, NST) - 1 >= 2)      { if (PW == 6) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
                                     else         asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory"); }
    else if (min(T, NST) - 1 == 1) { if (PW == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
                                     else         asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); }
    else                           asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    load(smem, 0, f0);
    if (NST == 3) for (int t = 0; t < T; t += 3) round3(t, smem, smem + STG, smem + 2 * STG);
    else          for (int t = 0; t < T; t += 2) round2(t, smem, smem + STG);

    float ma = mpa, mb = mpb;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, o)); mb = fmaxf(mb, __shfl_xor(mb, o)); }
    const float unscale = ldexpf(1.0f, -(pk16_exp_of_max(ma) + pk16_exp_of_max(mb)));
    float* __restrict__ C = p.C + (size_t)bz * p.sC;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 64 + b * 32 + l31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wm * 64 + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                if (m < p.M && n < p.N) C[(size_t)m * p.ldc + n] = acc[a][b][reg] * unscale;
            }
        }
}

unsigned grid_of(size_t total, unsigned cap)
{
    const size_t g = (total + 255) / 256;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

}  // namespace

int launch_pk16_rows(const float* src, size_t rows, int cols, int colsp, const float* maxpart, void* dst, hipStream_t s,
                     int tile_rows)
{
    PackRowsArgs a = {src, (uint2*)dst, maxpart, rows, cols, colsp, tile_rows};
    hipLaunchKernelGGL(pk16_rows_kernel, dim3(grid_of(rows * (size_t)(colsp >> 2), 8192)), dim3(256), 0, s, a);
    return check_launch("pk16_rows_kernel");
}

int launch_pk16_gather_cols(const float* src, int BD, int H, int W, const float* maxpart, void* dst, hipStream_t s)
{
    const int S0 = (int)level_geom(H, W, 0).slab;
    PackColsArgs a = {src, (uint2*)dst, maxpart, BD, H, W, S0};
    hipLaunchKernelGGL(pk16_gather_cols_kernel, dim3(grid_of((size_t)BD * (S0 >> 2), 8192)), dim3(256), 0, s, a);
    return check_launch("pk16_gather_cols_kernel");
}

int launch_corr_bound(const float* const* maxima, const int* counts, int K, float factor, float* out, hipStream_t s)
{
    if (K < 1 || K > 32) return fail(RAFTCORR_EUNSUPPORTED, "corr_bound: %d records (1..32)", K);
    BoundArgs a = {};
    for (int k = 0; k < K; ++k) { a.mx[k] = maxima[k]; a.n[k] = counts[k]; }
    a.K = K; a.factor = factor; a.out = out;
    hipLaunchKernelGGL(corr_bound_kernel, dim3(1), dim3(256), 0, s, a);
    return check_launch("corr_bound_kernel");
}

// C[img][m][n] = sum_k A[img][m][k] * Bm[img](k, n): A pk16 [M][Kp]; b_rows_are_k == 0: B pk16 [rowsB >= N][Kp] (row n);
// b_rows_are_k != 0: B pk16 [rowsB >= K][Np] (row k), Np a multiple of 128 covering N.  Kp (the k extent that is read) =
// 32 * ceil(K / 32); A's pad columns must be zeros and B's rows finite.
int launch_gemm_pk16(const void* A, const void* Bm, float* C, int M, int N, int K, int b_rows_are_k,
                     int colsA, int colsB, int rowsB, int ldc, size_t sA_bytes, size_t sB_bytes, size_t sC,
                     int batch, const float* maxA, const float* maxB, hipStream_t s, int b_tile_major)
{
    if ((colsA & 31) || (colsB & 31) || (sA_bytes & 15) || (sB_bytes & 15) || (((uintptr_t)A | (uintptr_t)Bm) & 15))
        return fail(RAFTCORR_EINVAL, "gemm_pk16: operands must be pk16 rows of a multiple of 32 columns, 16-byte aligned");
    const int T = (K + 31) / 32;
    if (T * 32 > colsA || (b_rows_are_k ? (((N + 127) / 128) * 128 > colsB || rowsB < 1) : (T * 32 > colsB || rowsB < 1)))
        return fail(RAFTCORR_EINVAL, "gemm_pk16: operand extents do not cover the tile grid");
    Gemm16Args a = {};
    a.A = (const uint4*)A; a.B = (const uint4*)Bm; a.C = C;
    a.M = M; a.N = N; a.T = T;
    a.pa = (unsigned)colsA / 4; a.pb = (unsigned)colsB / 4;
    a.rowsA = M; a.rowsB = rowsB; a.ldc = ldc;
    a.sA = sA_bytes / 16; a.sB = sB_bytes / 16; a.sC = sC;
    a.ntiles = (N + 127) / 128; a.total = a.ntiles * batch; a.per = (a.total + 7) / 8;
    a.maxA = maxA; a.maxB = maxB;
    a.nblkB = b_tile_major ? colsB / 32 : 0;
    bool wide = M > 128;
    int nst = 3;
#if defined(RC_LAB)                    // tools/ only: 128-row tiles, two stages, two workgroups per CU
    if (getenv("LAB_NARROW")) { wide = false; nst = 2; }
    if (getenv("LAB_NST2")) nst = 2;
#endif
    const int TM = wide ? 256 : 128;
    const size_t lds = (size_t)nst * (TM * 8 + 1024) * 16;
    const dim3 grid(8 * a.per, (M + TM - 1) / TM);
    typedef void (*Kern)(Gemm16Args);
    Kern kern;
    if (nst == 3) kern = wide ? (b_rows_are_k ? gemm_pk16_kernel<4, true, 3> : gemm_pk16_kernel<4, false, 3>)
                              : (b_rows_are_k ? gemm_pk16_kernel<2, true, 3> : gemm_pk16_kernel<2, false, 3>);
    else          kern = wide ? (b_rows_are_k ? gemm_pk16_kernel<4, true, 2> : gemm_pk16_kernel<4, false, 2>)
                              : (b_rows_are_k ? gemm_pk16_kernel<2, true, 2> : gemm_pk16_kernel<2, false, 2>);
    if (lds > 64 * 1024) {
        const hipError_t err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return fail(RAFTCORR_ELAUNCH, "gemm_pk16: %s", hipGetErrorString(err));
    }
    hipLaunchKernelGGL(kern, grid, dim3(wide ? 512 : 256), lds, s, a);
    return check_launch("gemm_pk16_kernel");
}

}  // namespace raftcorr
