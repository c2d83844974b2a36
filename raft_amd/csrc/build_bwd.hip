// Backward of the pyramid build: pooled-level gradients are folded down to level 0, then
// two fp32-MFMA contractions give the feature gradients.
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
};

__global__ __launch_bounds__(256)
void corr_fold_levels_kernel(FoldArgs p)
{
    const size_t total = p.BN * (size_t)p.s0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t qb = e / p.s0;
        const int c = (int)(e - qb * p.s0);
        const int patch = c >> 7, n = (c >> 5) & 3, j = c & 31;
        float g = p.d0[e];
        // parents in the pooled levels (layout.h): j bits 2,3,4 = dy1, dx1, dx2
        const int dy1 = (j >> 2) & 1, dx1 = (j >> 3) & 1, dx2 = (j >> 4) & 1;
        if (p.L > 1) g += 0.25f * p.d1[qb * p.s1 + patch * 32 + n * 8 + (dx1 | (dy1 << 1) | (dx2 << 2))];
        if (p.L > 2) g += 0.0625f * p.d2[qb * p.s2 + patch * 8 + (dx2 | ((n >> 1) << 1) | ((n & 1) << 2))];
        if (p.L > 3) {
            const int ph = patch / p.PW, pwi = patch - ph * p.PW;
            g += 0.015625f * p.d3[qb * p.s3 + ((ph >> 1) * p.PW + pwi) * 4 + (n & 1) + (ph & 1) * 2];
        }
        p.d0[e] = p.exact ? g * p.scale : g / p.sqrt_d;
    }
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

// C[m][n] = sum_k A[m][k] * Bm(k, n);  A row-major (lda, k contiguous);
// Bm(k, n) = B[k*ldb + n] (B_N_CONTIG) or B[n*ldb + k] (k contiguous).
// 64x64 block tile, 4 waves as 2x2, one 32x32 accumulator each, BK = 32.
struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    size_t lda, ldb, ldc;
    size_t sA, sB, sC;   // batch strides
};

template <bool B_N_CONTIG>
__global__ __launch_bounds__(256)
void gemm_f32_kernel(GemmArgs p)
{
    constexpr int TM = 64, TN = 64, TK = 32, LD = 65;
    __shared__ float As[2][TK][LD];
    __shared__ float Bs[2][TK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * TN, m0 = blockIdx.y * TM, bz = blockIdx.z;
    const float* __restrict__ A = p.A + bz * p.sA;
    const float* __restrict__ B = p.B + bz * p.sB;
    float* __restrict__ C = p.C + bz * p.sC;
    const int M = p.M, N = p.N, K = p.K;

    float ra[8], rb[8];
    auto gload = [&](int k0) {
        {   // A: lanes along k
            const int kk = tid & 31, k = k0 + kk;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + (tid >> 5) + 8 * i;
                ra[i] = (m < M && k < K) ? A[(size_t)m * p.lda + k] : 0.f;
            }
        }
        if (B_N_CONTIG) {
            const int n = n0 + (tid & 63);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = k0 + (tid >> 6) + 4 * i;
                rb[i] = (n < N && k < K) ? B[(size_t)k * p.ldb + n] : 0.f;
            }
        } else {
            const int kk = tid & 31, k = k0 + kk;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = n0 + (tid >> 5) + 8 * i;
                rb[i] = (n < N && k < K) ? B[(size_t)n * p.ldb + k] : 0.f;
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) As[buf][tid & 31][(tid >> 5) + 8 * i] = ra[i];
        if (B_N_CONTIG) {
#pragma unroll
            for (int i = 0; i < 8; ++i) Bs[buf][(tid >> 6) + 4 * i][tid & 63] = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) Bs[buf][tid & 31][(tid >> 5) + 8 * i] = rb[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nk = (K + TK - 1) / TK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int kh = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * TK);
#pragma unroll
        for (int kk = 0; kk < TK / 2; ++kk) {
            const float a = As[buf][2 * kk + kh][wm * 32 + l31];
            const float b = Bs[buf][2 * kk + kh][wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + l31;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        if (m < M && n < N) C[(size_t)m * p.ldc + n] = acc[reg];
    }
}

size_t build_bwd_workspace_bytes(int B, int D, int H, int W)
{
    const LevelGeom g0 = level_geom(H, W, 0);
    return 2 * (size_t)B * D * g0.slab * sizeof(float);       // F2 in column order + dF2 in column order
}

static unsigned grid_for(size_t total, unsigned cap)
{
    const size_t g = (total + 255) / 256;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

int launch_build_bwd(const float* f1, const float* f2, int B, int D, int H, int W, int L,
                     const LevelSet& dlv, float* df1, float* df2, void* workspace, size_t workspace_bytes,
                     hipStream_t s)
{
    if (!workspace || workspace_bytes < build_bwd_workspace_bytes(B, D, H, W))
        return fail(RAFTCORR_EINVAL, "build_bwd: workspace too small (%zu < %zu bytes)", workspace_bytes,
                    build_bwd_workspace_bytes(B, D, H, W));
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
    const int Lf = L < 4 ? L : 4;
    f.d0 = dlv.ptr[0]; f.d1 = dlv.ptr[1]; f.d2 = dlv.ptr[2]; f.d3 = dlv.ptr[3];
    f.s0 = dlv.g[0].slab; f.s1 = dlv.g[1].slab; f.s2 = dlv.g[2].slab; f.s3 = dlv.g[3].slab;
    f.L = Lf; f.BN = BN; f.PW = patches_w(W);
    const float sq = sqrtf((float)D);
    int e; const float mant = frexpf(sq, &e);
    f.exact = (mant == 0.5f) && (sq * sq == (float)D);
    f.scale = 1.0f / sq; f.sqrt_d = sq;
    const size_t S0 = (size_t)dlv.g[0].slab;
    hipLaunchKernelGGL(corr_fold_levels_kernel, dim3(grid_for(BN * S0, 16384)), dim3(256), 0, s, f);
    rc = check_launch("corr_fold_levels_kernel");
    if (rc) return rc;

    float* f2v = (float*)workspace;                 // [B][D][S0]
    float* df2v = f2v + (size_t)B * D * S0;         // [B][D][S0]
    hipLaunchKernelGGL(corr_cols_gather_kernel, dim3(grid_for((size_t)B * D * S0, 16384)), dim3(256), 0, s,
                       f2, f2v, B * D, H, W, (long long)S0);
    rc = check_launch("corr_cols_gather_kernel");
    if (rc) return rc;

    GemmArgs a;
    // dF1[d][q] = sum_p F2v[d][p] * dC0[q][p]      (B operand: k = p contiguous)
    a.A = f2v; a.B = dlv.ptr[0]; a.C = df1; a.M = D; a.N = N; a.K = (int)S0;
    a.lda = S0; a.ldb = S0; a.ldc = N;
    a.sA = (size_t)D * S0; a.sB = (size_t)N * S0; a.sC = (size_t)D * N;
    hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3((N + 63) / 64, (D + 63) / 64, B), dim3(256), 0, s, a);
    rc = check_launch("gemm_f32_kernel<dF1>");
    if (rc) return rc;
    // dF2v[d][p] = sum_q F1[d][q] * dC0[q][p]      (B operand: n = p contiguous)
    a.A = f1; a.B = dlv.ptr[0]; a.C = df2v; a.M = D; a.N = (int)S0; a.K = N;
    a.lda = N; a.ldb = S0; a.ldc = S0;
    a.sA = (size_t)D * N; a.sB = (size_t)N * S0; a.sC = (size_t)D * S0;
    hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3((unsigned)((S0 + 63) / 64), (D + 63) / 64, B), dim3(256), 0, s, a);
    rc = check_launch("gemm_f32_kernel<dF2>");
    if (rc) return rc;
    hipLaunchKernelGGL(corr_cols_scatter_back_kernel, dim3(grid_for((size_t)B * D * N, 16384)), dim3(256), 0, s,
                       df2v, df2, B * D, H, W, dlv.g[0]);
    return check_launch("corr_cols_scatter_back_kernel");
}

}  // namespace raftcorr
