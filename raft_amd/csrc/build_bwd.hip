// Backward of the pyramid build: pooled-level gradients are folded down to level 0, then two
// contractions give the feature gradients.  Aligned operands (the normal case) run them on the
// 16-bit matrix cores at fp32-grade accuracy (gemm_bf16x3_kernel<., F16 = true>): both operands are
// range-scaled by a power of two and split into two fp16 pieces while they are staged, three products
// (hi*hi + hi*lo + lo*hi), exact unscale -- the fp16x2 scheme of build_pyramid.hip.  The ranges cost
// almost nothing: max|F1|, max|F2| are one small pass, max|folded gradient| is taken by the fold kernel
// on the values it writes anyway (per-workgroup maxima, reduced by corr_maxabs_kernel: no atomics).
// The three-piece bf16 variant (six products, no ranges needed) remains as the F16 = false
// instantiation; unaligned operands take the exact fp32-MFMA kernel (gemm_f32_kernel).  Split-K partial
// products go to separate planes and are added in a fixed order: the gradients are bit-reproducible.
//
// Replaces what autograd runs behind core/corr.py:26,59,61 in the reference
// (avg_pool2d_backward x (L-1), the scale, and the two GEMMs of bmm's backward):
//   dC0[q,p]  = ( sum_l 4^-l * dLevel_l[q, parent_l(p)] ) / sqrt(D)   (fold, in place in level 0)
//   dF1[d,q]  = sum_p F2[d,t(p)] * dC0[q,p]
//   dF2[d,t]  = sum_q F1[d,q]    * dC0[q,p(t)]
// In the patch-tile order of layout.h the parent of column p at level l is pure bit
// arithmetic, and padding columns carry exact zeros (lookups never write them), so the fold
// is an elementwise pass and the contractions run over the padded column range unchecked.
#include "common.h"
#include <limits.h>
#include <algorithm>
#include <stdlib.h>

namespace raftcorr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FoldArgs {
    float* d0; const float* d1; const float* d2; const float* d3;
    long long s0, s1, s2, s3;
    int L, PW;
    size_t BN;          // B * N query slabs
    float scale;        // 1/sqrt(D)
    float sqrt_d;
    int exact;
    float* wgmax;       // [gridDim.x] max |folded level 0| per workgroup (the range pass of the fp16x2 contractions), or nullptr
};

constexpr unsigned FOLD_MAX_GRID = 32768;

// max |v| of the workgroup's values -> wgmax[blockIdx.x] (plain stores, reduced by corr_maxabs_kernel afterwards)
__device__ __forceinline__ void fold_note_max(float* wgmax, float m)
{
    __shared__ float wm[4];
    if (!wgmax) return;                                      // (uniform)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) wgmax[blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}

__global__ __launch_bounds__(256)
void corr_fold_levels_kernel(FoldArgs p)
{
    // one thread = four consecutive level-0 cells = one 2x2 block of cells, which shares all its
    // parents (layout.h): a 16-byte load, up to three scalar loads, a 16-byte store
    const size_t quads = p.BN * (size_t)(p.s0 >> 2);
    const int qps = (int)(p.s0 >> 2);                       // quads per slab
    float vmax = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < quads; e += (size_t)gridDim.x * 256) {
        const size_t qb = e / qps;
        const int c = (int)(e - qb * qps) << 2;
        const int patch = c >> 7, n = (c >> 5) & 3, j = c & 31;
        float4 v = *(const float4*)(p.d0 + qb * p.s0 + c);
        // parents in the pooled levels (layout.h): j bits 2,3,4 = dy1, dx1, dx2
        const int dy1 = (j >> 2) & 1, dx1 = (j >> 3) & 1, dx2 = (j >> 4) & 1;
        float g = 0.f;
        if (p.L > 1) g += 0.25f * p.d1[qb * p.s1 + patch * 32 + n * 8 + (dx1 | (dy1 << 1) | (dx2 << 2))];
        if (p.L > 2) g += 0.0625f * p.d2[qb * p.s2 + patch * 8 + (dx2 | ((n >> 1) << 1) | ((n & 1) << 2))];
        if (p.L > 3) {
            const int ph = patch / p.PW, pwi = patch - ph * p.PW;
            g += 0.015625f * p.d3[qb * p.s3 + ((ph >> 1) * p.PW + pwi) * 4 + (n & 1) + (ph & 1) * 2];
        }
        v.x += g; v.y += g; v.z += g; v.w += g;
        if (p.exact) { v.x *= p.scale; v.y *= p.scale; v.z *= p.scale; v.w *= p.scale; }
        else { v.x /= p.sqrt_d; v.y /= p.sqrt_d; v.z /= p.sqrt_d; v.w /= p.sqrt_d; }
        *(float4*)(p.d0 + qb * p.s0 + c) = v;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    fold_note_max(p.wgmax, vmax);
}

// the same, one cell per thread: for gradient buffers that are not 16-byte aligned
__global__ __launch_bounds__(256)
void corr_fold_levels_scalar_kernel(FoldArgs p)
{
    const size_t total = p.BN * (size_t)p.s0;
    float vmax = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t qb = e / p.s0;
        const int c = (int)(e - qb * p.s0);
        const int patch = c >> 7, n = (c >> 5) & 3, j = c & 31;
        const int dy1 = (j >> 2) & 1, dx1 = (j >> 3) & 1, dx2 = (j >> 4) & 1;
        float g = 0.f;
        if (p.L > 1) g += 0.25f * p.d1[qb * p.s1 + patch * 32 + n * 8 + (dx1 | (dy1 << 1) | (dx2 << 2))];
        if (p.L > 2) g += 0.0625f * p.d2[qb * p.s2 + patch * 8 + (dx2 | ((n >> 1) << 1) | ((n & 1) << 2))];
        if (p.L > 3) {
            const int ph = patch / p.PW, pwi = patch - ph * p.PW;
            g += 0.015625f * p.d3[qb * p.s3 + ((ph >> 1) * p.PW + pwi) * 4 + (n & 1) + (ph & 1) * 2];
        }
        const float v = p.d0[e] + g, w = p.exact ? v * p.scale : v / p.sqrt_d;
        p.d0[e] = w;
        vmax = fmaxf(vmax, fabsf(w));
    }
    fold_note_max(p.wgmax, vmax);
}

// levels >= 4 (row-major): fine += 0.25 * coarse[parent]
__global__ void corr_unpool_add_kernel(float* __restrict__ fine, const float* __restrict__ coarse,
                                       size_t M, LevelGeom gf, LevelGeom gc)
{
    const size_t total = M * (size_t)gf.h * gf.w;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % gf.w);
        const size_t r = e / gf.w;
        const int y = (int)(r % gf.h);
        const size_t m = r / gf.h;
        if ((y >> 1) < gc.h && (x >> 1) < gc.w)
            fine[m * gf.slab + level_pos(gf, y, x)] += 0.25f * coarse[m * gc.slab + level_pos(gc, y >> 1, x >> 1)];
    }
}

// dst[b][d][p] = src[b][d][t(p)] (0 in the padding): F2 in volume column order
__global__ __launch_bounds__(256)
void corr_cols_gather_kernel(const float* __restrict__ src, float* __restrict__ dst,
                             int BD, int H, int W, long long S0)
{
    const int PW = patches_w(W);
    const size_t total = (size_t)BD * S0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t bd = e / S0;
        const int c = (int)(e - bd * S0);
        const int patch = c >> 7, n = (c >> 5) & 3, j = c & 31;
        const int dx = (j & 1) | (((j >> 3) & 1) << 1) | ((j >> 4) << 2);
        const int dy = ((j >> 1) & 1) | (((j >> 2) & 1) << 1);
        const int y = (patch / PW) * PATCH_H + (n >> 1) * 4 + dy;
        const int x = (patch % PW) * PATCH_W + (n & 1) * 8 + dx;
        dst[e] = (y < H && x < W) ? src[bd * (size_t)(H * W) + y * W + x] : 0.f;
    }
}

// dst[b][d][t] = src[b][d][p(t)]
__global__ __launch_bounds__(256)
void corr_cols_scatter_back_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                   int BD, int H, int W, LevelGeom g0)
{
    const int N = H * W;
    const size_t total = (size_t)BD * N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t bd = e / N;
        const int t = (int)(e - bd * N);
        dst[e] = src[bd * g0.slab + level_pos(g0, t / W, t % W)];
    }
}

// C[m][n] (+)= sum_k A[m][k] * Bm(k, n);  A row-major (lda, k contiguous);
// Bm(k, n) = B[n*ldb + k] (B_K_CONTIG) or B[k*ldb + n] (n contiguous).
// 128x128 block tile, 4 waves as 2x2, each 64x64 = 2x2 accumulators of v_mfma_f32_32x32x2_f32,
// K tile 32 (a row's share of a tile = one 128-B line).  Operands are fetched with branch-free
// buffer loads (rows / k beyond the matrix read as zero), held in registers while the previous
// tile is multiplied, and stored into LDS as [k][row] images whose reads and (transposed) writes
// are both bank-conflict-free.  The K range can be split over gridDim.z (the feature-gradient
// GEMMs have M = D = 256 only, i.e. few tiles and a long K): every split then stores its partial
// product into its own plane of a workspace and gemm_reduce_kernel adds the planes in split order --
// no atomics, no zero-fill, the same bits on every run.
// Block -> (N tile, M tile).  The M tiles of one N tile read the same B tile (the gradient volume, the one big
// operand: M = D is 256, i.e. two M tiles): dealt 8 ids apart they land on the same XCD back to back (ids go
// round-robin over the 8 XCDs), so the second one finds the tile in that XCD's L2 instead of fetching it again
// (PMC: 439 MB per launch at Sintel size for a 202 MB operand before).
__device__ __forceinline__ void gemm_tile_of_block(int& nt, int& mt)
{
    const int gx = gridDim.x, gy = gridDim.y;
    const int lin = blockIdx.x + gx * blockIdx.y;
    const int nfull = gx & ~7;
    if (lin < nfull * gy) { const int r = lin >> 3; mt = r % gy; nt = (r / gy) * 8 + (lin & 7); }
    else { const int rem = lin - nfull * gy; mt = rem % gy; nt = nfull + rem / gy; }
}

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    int lda, ldb, ldc;
    size_t sA, sB, sC;   // batch strides (elements)
    int ksplit, kper;    // gridDim.z = batch * ksplit; kper = K range per split (multiple of 64)
    float* P;            // ksplit > 1: partial products [ksplit][batch][M][ldc]
    int batch;
    int vecA, vecB;      // 16-byte loads legal (alignment + extents)
    const float* maxA;   // fp16x2 kernel: [128] partial maxima of |A| and of |B| (range scaling)
    const float* maxB;
    // Columns of the folded gradient that are padding only (layout.h: the lower 4x16 half of every patch of the bottom
    // patch row when H % 8 is 1..4 -- cells 64..127 of those patches; 50x90: 384 of 5376 columns).  Both are "off" at
    // INT_MAX.  16-bit kernels only; the fp32 fallback multiplies the zeros.
    int kskip_from;      // K = columns (dF1): K indices >= this are skipped where (k >> 6) & 1
    int nremap_from;     // N = columns (dF2): N tiles >= this hold the upper halves of TWO patches (cells 0..63 each)
    // dF2 without split planes: the epilogue stores column p (patch-tile order) at its pixel, C = dF2 [M][H * W] itself
    // (no column-order copy, no corr_cols_scatter_back launch).  Off at scat_h = 0.
    int scat_h, scat_w, scat_pw;
};

typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));
constexpr unsigned GOOB = 0x80000000u;

__device__ __forceinline__ float4 gemm_quad(__amdgpu_buffer_rsrc_t rs, bool row_ok, size_t row_off, int c, int C,
                                            int vec)
{
    // 4 consecutive elements c..c+3 of a row (element offset row_off + c), zero beyond C / bad row
    const unsigned base = (unsigned)((row_off + (size_t)c) * 4);
    float4 v;
    if (vec) {
        const u32x4g w = __builtin_amdgcn_raw_buffer_load_b128(rs, (row_ok && c < C) ? base : GOOB, 0, 0);
        v = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    } else {
        v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (row_ok && c < C) ? base : GOOB, 0, 0));
        v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (row_ok && c + 1 < C) ? base + 4 : GOOB, 0, 0));
        v.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (row_ok && c + 2 < C) ? base + 8 : GOOB, 0, 0));
        v.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (row_ok && c + 3 < C) ? base + 12 : GOOB, 0, 0));
    }
    return v;
}

template <bool B_K_CONTIG>
__global__ __launch_bounds__(256, 2)
void gemm_f32_kernel(GemmArgs p)
{
    constexpr int TM = 128, TN = 128, TK = 32;
    constexpr int LDT = TM + 1;          // transposed images ([k][row], written one scalar at a time)
    constexpr int LDN = TN + 4;          // n-contiguous image (written as float4)
    __shared__ __attribute__((aligned(16))) float As[TK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[TK * (B_K_CONTIG ? LDT : LDN)];
    constexpr int LDB = B_K_CONTIG ? LDT : LDN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int nt, mt;
    gemm_tile_of_block(nt, mt);
    const int n0 = nt * TN, m0 = mt * TM;
    const int bz = blockIdx.z / p.ksplit, ks = blockIdx.z - bz * p.ksplit;
    const int kbeg = ks * p.kper, kend = min(p.K, kbeg + p.kper);
    const int M = p.M, N = p.N;
    const float* __restrict__ Ab = p.A + bz * p.sA;
    const float* __restrict__ Bb = p.B + bz * p.sB;
    float* __restrict__ C = p.ksplit > 1 ? p.P + ((size_t)ks * p.batch + bz) * p.sC : p.C + bz * p.sC;
    // buffer descriptors are re-based on every K tile (tile origin), so offsets stay small
    // whatever the matrix size; validity is decided per lane and spelled as an OOB offset
    auto desc = [](const float* base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    };

    float4 qa[4], qb[4];
    auto gload = [&](int k0) {
        {   // A[m][k]: thread = (row tid>>3 + 32 i, k quad tid&7)
            const __amdgpu_buffer_rsrc_t rs = desc(Ab + (size_t)m0 * p.lda + k0);
            const int kq = 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (tid >> 3) + 32 * i;
                qa[i] = gemm_quad(rs, m0 + r < M, (size_t)r * p.lda, kq, kend - k0, p.vecA);
            }
        }
        if (B_K_CONTIG) {
            const __amdgpu_buffer_rsrc_t rs = desc(Bb + (size_t)n0 * p.ldb + k0);
            const int kq = 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (tid >> 3) + 32 * i;
                qb[i] = gemm_quad(rs, n0 + r < N, (size_t)r * p.ldb, kq, kend - k0, p.vecB);
            }
        } else {        // B[k][n]: thread = (k row tid>>5 + 8 i, n quad tid&31)
            const __amdgpu_buffer_rsrc_t rs = desc(Bb + (size_t)k0 * p.ldb + n0);
            const int nq = 4 * (tid & 31);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (tid >> 5) + 8 * i;
                qb[i] = gemm_quad(rs, k0 + r < kend, (size_t)r * p.ldb, nq, N - n0, p.vecB);
            }
        }
    };
    auto lstore = [&]() {
        {
            const int k = 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* d = As + k * LDT + (tid >> 3) + 32 * i;
                d[0] = qa[i].x; d[LDT] = qa[i].y; d[2 * LDT] = qa[i].z; d[3 * LDT] = qa[i].w;
            }
        }
        if (B_K_CONTIG) {
            const int k = 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* d = Bs + k * LDB + (tid >> 3) + 32 * i;
                d[0] = qb[i].x; d[LDB] = qb[i].y; d[2 * LDB] = qb[i].z; d[3 * LDB] = qb[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(float4*)(Bs + ((tid >> 5) + 8 * i) * LDB + 4 * (tid & 31)) = qb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int kh = lane >> 5, l31 = lane & 31;
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (k0 + TK < kend) gload(k0 + TK);
        const float* la = As + kh * LDT + wm * 64 + l31;
        const float* lb = Bs + kh * LDB + wn * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < TK / 2; ++kk) {
            const float a0 = la[2 * kk * LDT], a1 = la[2 * kk * LDT + 32];
            const float b0 = lb[2 * kk * LDB], b1 = lb[2 * kk * LDB + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 64 + b * 32 + l31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wm * 64 + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                if (m < M && n < N) {
                    C[(size_t)m * p.ldc + n] = acc[a][b][reg];
                }
            }
        }
}

// The same GEMM on the 16-bit matrix cores at fp32-grade accuracy: both operands are split into
// three bf16 pieces while they are staged (x = hi + mid + lo, 24 significand bits) and the six
// products of weight >= 2^-16 are accumulated in fp32 (the bf16x3 scheme of build_pyramid.hip):
// 24 v_mfma_f32_32x32x16_bf16 per 16 k and wave instead of 32 v_mfma_f32_32x32x2_f32 at twice the
// cycles each.  LDS image per operand: [piece][k group of 8][row 128] x 16 bytes = one MFMA fragment
// slot per (k group, row); k-group planes are padded by 32 bytes so that the staging writes of
// lanes that run along k do not collide.  Needs 16-byte loads (vecA && vecB); the fp32 kernel
// above remains for operands that are not aligned.
typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2g __attribute__((ext_vector_type(2)));
typedef float f32x2g __attribute__((ext_vector_type(2)));

// x -> three bf16 pieces, two values per v_cvt_pk_bf16_f32 (round to nearest even in hardware)
__device__ __forceinline__ void split3(const float (&x)[8], uint4 (&out)[3])
{
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = x[j];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2g v = {r[2 * j], r[2 * j + 1]};
            const bf16x2g h = __builtin_convertvector(v, bf16x2g);
            w[j] = __builtin_bit_cast(unsigned, h);
            if (c < 2) {
                r[2 * j] -= __uint_as_float(w[j] << 16);
                r[2 * j + 1] -= __uint_as_float(w[j] & 0xffff0000u);
            }
        }
        out[c] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// x * scale -> two fp16 pieces (hi = RNE, lo = the exact remainder rounded once): the fp16x2 scheme of
// build_pyramid.hip / alt_fwd16.hip, 22 significand bits when the scale puts max|x| into [2^14, 2^15)
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2(const float (&x)[8], float scale, uint4 (&out)[3])
{
    unsigned short h[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = x[j] * scale;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        h[0][j] = __builtin_bit_cast(unsigned short, hi);
        h[1][j] = __builtin_bit_cast(unsigned short, lo);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
        out[c] = make_uint4(h[c][0] | ((unsigned)h[c][1] << 16), h[c][2] | ((unsigned)h[c][3] << 16),
                            h[c][4] | ((unsigned)h[c][5] << 16), h[c][6] | ((unsigned)h[c][7] << 16));
}

// power-of-two factor that maps the maximum of 128 partial maxima into [2^14, 2^15) (as build_pyramid.hip:range_exponent)
__device__ __forceinline__ int gemm_range_exp(const float* part)
{
    const int lane = threadIdx.x & 63;
    float m = fmaxf(part[lane], part[lane + 64]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (!(m > 0.f) || !(m < 3.0e38f)) return 0;
    int e;
    frexpf(m, &e);
    return 15 - e;
}

// F16 = false: three bf16 pieces per operand, six products (any finite input).  F16 = true: two fp16 pieces of the
// range-scaled operands, three products (hi*hi + hi*lo + lo*hi), exact power-of-two unscale in the epilogue: half the
// matrix-core work and two thirds of the conversion work; needs the operands' maxima (p.maxA, p.maxB).
template <bool B_K_CONTIG, bool F16>
__global__ __launch_bounds__(256, 2)
void gemm_bf16x3_kernel(GemmArgs p)
{
    // K tile: the loop keeps ONE tile of global loads in flight per workgroup, and an iteration is about as long as a
    // memory round trip under load, so the bytes per tile set the pace: 64 deep for the two-piece kernel with two
    // K-contiguous operands (365 -> 328 us at Things B=6; the n-contiguous B operand is fetched with dword loads,
    // twice as many of them did not pay: 296 -> 301 us, and 16-byte loads + a register transpose 326 us)
    constexpr int TM = 128, TN = 128, TK = (F16 && B_K_CONTIG) ? 64 : 32;
    constexpr int KG = TK / 8, SL = TK / 16;          // k groups of 8 per tile; 8-k slots per thread and operand
    constexpr int NP = F16 ? 2 : 3;                   // pieces per operand
    constexpr int PLANE = 128 + 2;                    // uint4 slots per k-group plane (2 slots = 32 B of padding)
    constexpr int OPU = NP * KG * PLANE;              // uint4 slots per operand image
    __shared__ uint4 As[OPU];
    __shared__ uint4 Bs[OPU];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int nt, mt;
    gemm_tile_of_block(nt, mt);
    const int n0 = nt * TN, m0 = mt * TM;
    // column bases of the tile's two 64-column halves (n-contiguous B only: the tiles past nremap_from pair the
    // upper halves of two bottom-row patches, an odd one out has no second half)
    int n0a = n0, n0b = n0 + 64;
    if (!B_K_CONTIG && nt >= p.nremap_from) { n0a = (2 * nt - p.nremap_from) * TN; n0b = n0a + TN; }
    const int bz = blockIdx.z / p.ksplit, ks = blockIdx.z - bz * p.ksplit;
    const int kend = min(p.K, (ks + 1) * p.kper);
    // first K tile at or after k that is not all padding (skipped blocks are the odd 64-blocks past kskip_from)
    auto adv = [&](int k) { return (B_K_CONTIG && k >= p.kskip_from && ((k >> 6) & 1)) ? ((k >> 6) + 1) << 6 : k; };
    const int kbeg = adv(ks * p.kper);
    const int M = p.M, N = p.N;
    const float* __restrict__ Ab = p.A + bz * p.sA;
    const float* __restrict__ Bb = p.B + bz * p.sB;
    float* __restrict__ C = p.ksplit > 1 ? p.P + ((size_t)ks * p.batch + bz) * p.sC : p.C + bz * p.sC;
    auto desc = [](const float* base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    };
    const int eA = F16 ? gemm_range_exp(p.maxA) : 0, eB = F16 ? gemm_range_exp(p.maxB) : 0;
    const float scA = ldexpf(1.0f, eA), scB = ldexpf(1.0f, eB), unscale = ldexpf(1.0f, -(eA + eB));
    auto split = [](const float (&x)[8], float scale, uint4 (&out)[3]) {
        if (F16) split2(x, scale, out);
        else     split3(x, out);
    };

    // staging registers: K-contiguous operand: SL slots x 8 k (two 16-byte loads each);
    // n-contiguous B: SL slots (column tid & 127, k groups (tid >> 7) + 2 i) x 8 dword loads
    float xa[SL][8], xb[SL][8];
    auto gload = [&](int k0) {
        {
            const __amdgpu_buffer_rsrc_t rs = desc(Ab + (size_t)m0 * p.lda + k0);
            const int kg = tid & (KG - 1);
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int r = tid / KG + (256 / KG) * i;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 v = gemm_quad(rs, m0 + r < M, (size_t)r * p.lda, 8 * kg + 4 * h, kend - k0, 1);
                    xa[i][4 * h] = v.x; xa[i][4 * h + 1] = v.y; xa[i][4 * h + 2] = v.z; xa[i][4 * h + 3] = v.w;
                }
            }
        }
        if (B_K_CONTIG) {
            const __amdgpu_buffer_rsrc_t rs = desc(Bb + (size_t)n0 * p.ldb + k0);
            const int kg = tid & (KG - 1);
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int r = tid / KG + (256 / KG) * i;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 v = gemm_quad(rs, n0 + r < N, (size_t)r * p.ldb, 8 * kg + 4 * h, kend - k0, 1);
                    xb[i][4 * h] = v.x; xb[i][4 * h + 1] = v.y; xb[i][4 * h + 2] = v.z; xb[i][4 * h + 3] = v.w;
                }
            }
        } else {
            const int col = (tid & 64) ? n0b + (tid & 63) : n0a + (tid & 63);
            const __amdgpu_buffer_rsrc_t rs = desc(Bb + (size_t)k0 * p.ldb);
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int kg = (tid >> 7) + 2 * i;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * kg + j;
                    const bool ok = k0 + k < kend && col < N;
                    xb[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rs, ok ? (unsigned)(((size_t)k * p.ldb + col) * 4) : GOOB, 0, 0));
                }
            }
        }
    };
    auto lstore = [&]() {
        uint4 pc[3];
        {
            const int kg = tid & (KG - 1);
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                split(xa[i], scA, pc);
#pragma unroll
                for (int c = 0; c < NP; ++c) As[(c * KG + kg) * PLANE + tid / KG + (256 / KG) * i] = pc[c];
            }
        }
        if (B_K_CONTIG) {
            const int kg = tid & (KG - 1);
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                split(xb[i], scB, pc);
#pragma unroll
                for (int c = 0; c < NP; ++c) Bs[(c * KG + kg) * PLANE + tid / KG + (256 / KG) * i] = pc[c];
            }
        } else {
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                split(xb[i], scB, pc);
#pragma unroll
                for (int c = 0; c < NP; ++c) Bs[(c * KG + (tid >> 7) + 2 * i) * PLANE + (tid & 127)] = pc[c];
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int kh = lane >> 5, l31 = lane & 31;
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg, kn; k0 < kend; k0 = kn) {
        kn = adv(k0 + TK);
        __syncthreads();
        lstore();
        __syncthreads();
        if (kn < kend) gload(kn);
#pragma unroll
        for (int st = 0; st < TK / 16; ++st) {
            // piece of B outermost, small products first: lo*hi | mid*mid, mid*hi | hi*lo, hi*mid, hi*hi
            uint4 af[NP][2];
#pragma unroll
            for (int c = 0; c < NP; ++c)
#pragma unroll
                for (int a = 0; a < 2; ++a) af[c][a] = As[(c * KG + 2 * st + kh) * PLANE + wm * 64 + a * 32 + l31];
#pragma unroll
            for (int cb = NP - 1; cb >= 0; --cb) {
                uint4 bf[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b] = Bs[(cb * KG + 2 * st + kh) * PLANE + wn * 64 + b * 32 + l31];
#pragma unroll
                for (int ca = NP - 1 - cb; ca >= 0; --ca)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            if (F16)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    __builtin_bit_cast(f16x8g, af[ca][a]), __builtin_bit_cast(f16x8g, bf[b]), acc[a][b], 0, 0, 0);
                            else
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                    __builtin_bit_cast(bf16x8g, af[ca][a]), __builtin_bit_cast(bf16x8g, bf[b]), acc[a][b], 0, 0, 0);
                        }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int n = (wn ? n0b : n0a) + b * 32 + l31;
            bool n_ok = n < N;
            if (!B_K_CONTIG && p.scat_h) {          // column of the volume -> pixel (inverse of layout.h level_pos, level 0)
                const int patch = n >> 7, t = (n >> 5) & 3, j = n & 31;
                const int dx = (j & 1) | (((j >> 3) & 1) << 1) | ((j >> 4) << 2);
                const int dy = ((j >> 1) & 1) | (((j >> 2) & 1) << 1);
                const int py = patch / p.scat_pw;
                const int y = py * PATCH_H + (t >> 1) * 4 + dy, x = (patch - py * p.scat_pw) * PATCH_W + (t & 1) * 8 + dx;
                n_ok = n_ok && y < p.scat_h && x < p.scat_w;
                n = y * p.scat_w + x;
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wm * 64 + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                if (m < M && n_ok) {
                    C[(size_t)m * p.ldc + n] = F16 ? acc[a][b][reg] * unscale : acc[a][b][reg];
                }
            }
        }
}

// C[i] = sum over the split planes, in split order.  grid: ceil(total / 4 / 256)
__global__ __launch_bounds__(256)
void gemm_reduce_kernel(const float* __restrict__ P, float* __restrict__ C, int ksplit, size_t total)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    if (i + 4 <= total && (((uintptr_t)P | (uintptr_t)C | (total * 4)) & 15) == 0) {
        float4 sum = *(const float4*)(P + i);
        for (int k = 1; k < ksplit; ++k) {
            const float4 v = *(const float4*)(P + (size_t)k * total + i);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        *(float4*)(C + i) = sum;
    } else {
        for (size_t e = i; e < total && e < i + 4; ++e) {
            float sum = P[e];
            for (int k = 1; k < ksplit; ++k) sum += P[(size_t)k * total + e];
            C[e] = sum;
        }
    }
}

// K splits of one batched GEMM: enough workgroups for the chip while every split keeps a K range of >= 256
static int gemm_ksplit(int M, int N, int K, int batch)
{
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128) * batch;
    int ksplit = 1;
    // 512 resident workgroups fill the chip; a launch of 400+ runs as one round either way, and not splitting saves the
    // partial planes and their reduction (Things B=6: 432 and 504 tiles, backward 1.233 -> 1.202 ms)
    int target = 400;
#if defined(RC_LAB)                    // tools/ only: workgroups below which K is split once more
    if (const char* e = getenv("RC_KSPLIT_TARGET")) target = atoi(e);
#endif
    while (tiles * ksplit < target && ksplit < 16 && K / (ksplit * 2) >= 256) ksplit *= 2;
    return ksplit;
}

// fills the launch-shape fields and runs one batched GEMM (+ the reduction of its split planes); C must be
// (M x ldc) per image, images back to back
template <bool B_K_CONTIG>
static int run_gemm(GemmArgs a, int batch, float* partial, const char* what, hipStream_t s)
{
    if ((size_t)128 * std::max(a.lda, a.ldb) * 4 >= ((size_t)1 << 31))
        return fail(RAFTCORR_EUNSUPPORTED, "%s: leading dimension too large", what);
    const int ksplit = gemm_ksplit(a.M, a.N, a.K, batch);
    a.ksplit = ksplit;
    a.kper = (((a.K + ksplit - 1) / ksplit) + 63) / 64 * 64;         // a multiple of every kernel's K tile
    a.P = partial; a.batch = batch;
    a.vecA = (a.lda % 4 == 0) && (a.K % 4 == 0) && (((uintptr_t)a.A | (a.sA * 4)) % 16 == 0);
    a.vecB = (a.ldb % 4 == 0) && ((B_K_CONTIG ? a.K : a.N) % 4 == 0) && (((uintptr_t)a.B | (a.sB * 4)) % 16 == 0);
    if (a.sC != (size_t)a.M * a.ldc) return fail(RAFTCORR_EINVAL, "%s: C images must be contiguous", what);
    dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, batch * ksplit);
    bool split16 = a.vecA && a.vecB;                         // unaligned operands: the fp32-MFMA kernel
#if defined(RC_LAB)                    // tools/ only: force the fp32-MFMA kernel
    if (getenv("RC_BWD_FP32")) split16 = false;
#endif
    bool f16 = a.maxA && a.maxB;
#if defined(RC_LAB)                    // tools/ only: the three-piece bf16 kernel
    if (getenv("RC_BWD_BF16X3")) f16 = false;
    if (getenv("RC_BWD_NOSKIP")) { a.kskip_from = INT_MAX; a.nremap_from = INT_MAX; }
#endif
    if (!split16) { a.kskip_from = INT_MAX; a.nremap_from = INT_MAX; }
    if (a.scat_h && (!split16 || ksplit > 1)) return fail(RAFTCORR_EINVAL, "%s: scattered store needs the 16-bit kernel without K split", what);
    if (!B_K_CONTIG && a.nremap_from != INT_MAX) {           // the paired tiles: half as many past nremap_from
        const int tiles = (a.N + 127) / 128;
        grid.x = a.nremap_from + (tiles - a.nremap_from + 1) / 2;
    }
    if (split16 && f16)
        hipLaunchKernelGGL((gemm_bf16x3_kernel<B_K_CONTIG, true>), grid, dim3(256), 0, s, a);
    else if (split16)
        hipLaunchKernelGGL((gemm_bf16x3_kernel<B_K_CONTIG, false>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<B_K_CONTIG>, grid, dim3(256), 0, s, a);
    int rc = check_launch(what);
    if (rc || ksplit == 1) return rc;
    const size_t total = a.sC * batch;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, s, partial, a.C, ksplit, total);
    return check_launch("gemm_reduce_kernel");
}

size_t build_bwd_workspace_bytes(int B, int D, int H, int W)
{
    const LevelGeom g0 = level_geom(H, W, 0);
    const size_t S0 = (size_t)g0.slab, N = (size_t)H * W;
    // F2 in column order + dF2 in column order + the split-K planes of the larger of the two GEMMs
    const size_t p1 = gemm_ksplit(D, (int)N, (int)S0, B) > 1 ? gemm_ksplit(D, (int)N, (int)S0, B) * (size_t)B * D * N : 0;
    const size_t p2 = gemm_ksplit(D, (int)S0, (int)N, B) > 1 ? gemm_ksplit(D, (int)S0, (int)N, B) * (size_t)B * D * S0 : 0;
    // + 3 x 128 partial maxima (|F1|, |F2|, |folded gradient|) and the fold's per-workgroup maxima behind the planes
    return (2 * (size_t)B * D * S0 + std::max(p1, p2) + 512 + FOLD_MAX_GRID) * sizeof(float);
}

static unsigned grid_for(size_t total, unsigned cap)
{
    const size_t g = (total + 255) / 256;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// forward declarations (bwd_deferred.hip)
int launch_splat_fold(int B, int H, int W, int L, int r, int mode, int D, int K, const float* const* coords,
                      const float* const* records, float* d0, int accumulate, float* wgmax, unsigned* n_wg, hipStream_t s);

size_t build_bwd_deferred_workspace_bytes(int B, int D, int H, int W)
{
    // the plain backward's workspace + the folded level-0 gradient [B * N][S0] behind it (256-byte aligned)
    const size_t base = (build_bwd_workspace_bytes(B, D, H, W) + 255) / 256 * 256;
    return base + (size_t)B * H * W * (size_t)level_geom(H, W, 0).slab * sizeof(float);
}

// classic fold of a dense gradient pyramid into its level 0 (in place); per-workgroup maxima -> wgmax[0 .. *n_wg)
static int fold_stage(int B, int D, int H, int W, int L, const LevelSet& dlv, float* wgmax, unsigned* n_wg, hipStream_t s)
{
    const int N = H * W;
    const size_t BN = (size_t)B * N;
    int rc;
    // levels >= 4 fold into level 3 first (coarse to fine)
    for (int l = L - 1; l >= 4; --l) {
        const size_t total = BN * dlv.g[l - 1].h * dlv.g[l - 1].w;
        if (total == 0 || dlv.g[l].h == 0 || dlv.g[l].w == 0) continue;
        hipLaunchKernelGGL(corr_unpool_add_kernel, dim3(grid_for(total, 4096)), dim3(256), 0, s,
                           dlv.ptr[l - 1], dlv.ptr[l], BN, dlv.g[l - 1], dlv.g[l]);
        rc = check_launch("corr_unpool_add_kernel");
        if (rc) return rc;
    }
    FoldArgs f;
    f.wgmax = wgmax;
    const int Lf = L < 4 ? L : 4;
    f.d0 = dlv.ptr[0]; f.d1 = dlv.ptr[1]; f.d2 = dlv.ptr[2]; f.d3 = dlv.ptr[3];
    f.s0 = dlv.g[0].slab; f.s1 = dlv.g[1].slab; f.s2 = dlv.g[2].slab; f.s3 = dlv.g[3].slab;
    f.L = Lf; f.BN = BN; f.PW = patches_w(W);
    const float sq = sqrtf((float)D);
    int e; const float mant = frexpf(sq, &e);
    f.exact = (mant == 0.5f) && (sq * sq == (float)D);
    f.scale = 1.0f / sq; f.sqrt_d = sq;
    const size_t S0 = (size_t)dlv.g[0].slab;
    const bool vec = (uintptr_t)dlv.ptr[0] % 16 == 0;        // slabs are multiples of 128 floats
    const unsigned fold_grid = vec ? grid_for(BN * S0 / 4, FOLD_MAX_GRID) : grid_for(BN * S0, 16384);
    if (vec) hipLaunchKernelGGL(corr_fold_levels_kernel, dim3(fold_grid), dim3(256), 0, s, f);
    else     hipLaunchKernelGGL(corr_fold_levels_scalar_kernel, dim3(fold_grid), dim3(256), 0, s, f);
    *n_wg = fold_grid;
    return check_launch("corr_fold_levels_kernel");
}

// the two contractions on the folded level-0 gradient d0 [B * N][S0]; maxima: [3][128] (|F1|, |F2| filled here,
// |d0| reduced here from wgmax[0 .. n_wg))
static int contract_stage(const float* f1, const float* f2, int B, int D, int H, int W, const float* d0,
                          float* df1, float* df2, void* workspace, float* maxima, const float* wgmax, unsigned n_wg,
                          hipStream_t s)
{
    const int N = H * W;
    const LevelGeom g0 = level_geom(H, W, 0);
    const size_t S0 = (size_t)g0.slab;
    int rc = launch_maxabs3(f1, f2, (size_t)B * D * N, wgmax, n_wg, maxima, s);      // maxima[0..127], [128..255], [256..383]
    if (rc) return rc;

    float* f2v = (float*)workspace;                 // [B][D][S0]
    float* df2v = f2v + (size_t)B * D * S0;         // [B][D][S0]
    float* planes = df2v + (size_t)B * D * S0;      // split-K partial products
    hipLaunchKernelGGL(corr_cols_gather_kernel, dim3(grid_for((size_t)B * D * S0, 16384)), dim3(256), 0, s,
                       f2, f2v, B * D, H, W, (long long)S0);
    rc = check_launch("corr_cols_gather_kernel");
    if (rc) return rc;

    GemmArgs a = {};
    // bottom patch row with an all-padding lower half: dF1 skips those K blocks, dF2 pairs the upper halves
    const bool low_pad = H % PATCH_H >= 1 && H % PATCH_H <= 4;
    const int first_bottom = (patches_h(H) - 1) * patches_w(W);
    a.kskip_from = low_pad ? first_bottom * PATCH_CELLS : INT_MAX;
    a.nremap_from = low_pad ? first_bottom : INT_MAX;
    // dF1[d][q] = sum_p F2v[d][p] * dC0[q][p]      (B operand: k = p contiguous)
    a.A = f2v; a.B = d0; a.C = df1; a.M = D; a.N = N; a.K = (int)S0;
    a.lda = (int)S0; a.ldb = (int)S0; a.ldc = N;
    a.sA = (size_t)D * S0; a.sB = (size_t)N * S0; a.sC = (size_t)D * N;
    a.maxA = maxima + 128; a.maxB = maxima + 256;
    rc = run_gemm<true>(a, B, planes, "gemm_f32_kernel<dF1>", s);
    if (rc) return rc;
    // dF2v[d][p] = sum_q F1[d][q] * dC0[q][p]      (B operand: n = p contiguous)
    a.A = f1; a.B = d0; a.C = df2v; a.M = D; a.N = (int)S0; a.K = N;
    a.lda = N; a.ldb = (int)S0; a.ldc = (int)S0;
    a.sA = (size_t)D * N; a.sB = (size_t)N * S0; a.sC = (size_t)D * S0;
    a.maxA = maxima;
    // one K range and 16-byte-aligned operands (the 16-bit kernel): the GEMM stores dF2 at its pixels itself
    const bool aligned = N % 4 == 0 && ((uintptr_t)f1 | (uintptr_t)d0) % 16 == 0;
    if (aligned && gemm_ksplit(a.M, a.N, a.K, B) == 1) {
        a.C = df2; a.ldc = N; a.sC = (size_t)D * N;
        a.scat_h = H; a.scat_w = W; a.scat_pw = patches_w(W);
        return run_gemm<false>(a, B, planes, "gemm_bf16x3_kernel<dF2>", s);
    }
    rc = run_gemm<false>(a, B, planes, "gemm_f32_kernel<dF2>", s);
    if (rc) return rc;
    hipLaunchKernelGGL(corr_cols_scatter_back_kernel, dim3(grid_for((size_t)B * D * N, 16384)), dim3(256), 0, s,
                       df2v, df2, B * D, H, W, g0);
    return check_launch("corr_cols_scatter_back_kernel");
}

int launch_build_bwd(const float* f1, const float* f2, int B, int D, int H, int W, int L,
                     const LevelSet& dlv, float* df1, float* df2, void* workspace, size_t workspace_bytes,
                     hipStream_t s)
{
    if (!workspace || workspace_bytes < build_bwd_workspace_bytes(B, D, H, W))
        return fail(RAFTCORR_EINVAL, "build_bwd: workspace too small (%zu < %zu bytes)", workspace_bytes,
                    build_bwd_workspace_bytes(B, D, H, W));
    // range pass for the fp16x2 contractions: |F1|, |F2| by a small pass, |folded gradient| by the fold kernel itself
    float* maxima = (float*)((char*)workspace + build_bwd_workspace_bytes(B, D, H, W)) - (512 + FOLD_MAX_GRID);   // [3][128]
    float* wgmax = maxima + 512;                             // [FOLD_MAX_GRID]
    unsigned n_wg = 0;
    int rc = fold_stage(B, D, H, W, L, dlv, wgmax, &n_wg, s);
    if (rc) return rc;
    return contract_stage(f1, f2, B, D, H, W, dlv.ptr[0], df1, df2, workspace, maxima, wgmax, n_wg, s);
}

// Deferred form: the lookups' gradients arrive as K recorded (coords, d(out)) pairs instead of (or, dlv != nullptr, on
// top of) a dense gradient pyramid; see bwd_deferred.hip.
int launch_build_bwd_deferred(const float* f1, const float* f2, int B, int D, int H, int W, int L, int r, int mode,
                              int K, const float* const* coords, const float* const* records, const LevelSet* dlv,
                              float* df1, float* df2, void* workspace, size_t workspace_bytes, hipStream_t s)
{
    const size_t base = (build_bwd_workspace_bytes(B, D, H, W) + 255) / 256 * 256;
    const size_t need = dlv ? build_bwd_workspace_bytes(B, D, H, W) : build_bwd_deferred_workspace_bytes(B, D, H, W);
    if (!workspace || workspace_bytes < need)
        return fail(RAFTCORR_EINVAL, "build_bwd_deferred: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    if (((uintptr_t)workspace & 15) != 0) return fail(RAFTCORR_EINVAL, "build_bwd_deferred: workspace must be 16-byte aligned");
    float* maxima = (float*)((char*)workspace + build_bwd_workspace_bytes(B, D, H, W)) - (512 + FOLD_MAX_GRID);
    float* wgmax = maxima + 512;
    unsigned n_wg = 0;
    float* d0;
    int rc;
    if (dlv) {                                               // a dense pyramid gradient as well: fold it first, add the records
        if ((uintptr_t)dlv->ptr[0] % 16 != 0) return fail(RAFTCORR_EINVAL, "build_bwd_deferred: dlevels[0] must be 16-byte aligned");
        rc = fold_stage(B, D, H, W, L, *dlv, wgmax, &n_wg, s);
        if (rc) return rc;
        d0 = dlv->ptr[0];
    } else {
        d0 = (float*)((char*)workspace + base);
    }
    if (K > 0) {
        rc = launch_splat_fold(B, H, W, L, r, mode, D, K, coords, records, d0, dlv ? 1 : 0, wgmax, &n_wg, s);
        if (rc) return rc;
    } else if (!dlv) {
        return fail(RAFTCORR_EINVAL, "build_bwd_deferred: neither records nor a pyramid gradient given");
    }
    return contract_stage(f1, f2, B, D, H, W, d0, df1, df2, workspace, maxima, wgmax, n_wg, s);
}

}  // namespace raftcorr
