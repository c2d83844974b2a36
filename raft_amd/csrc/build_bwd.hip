// Backward of the pyramid build: pooled-level gradients are folded down to level 0, then
// two fp32-MFMA contractions give the feature gradients.
//
// Replaces what autograd runs behind core/corr.py:26,59,61 in the reference
// (avg_pool2d_backward x (L-1), the scale, and the two GEMMs of bmm's backward):
//   dC0[q,t]  = ( sum_l 4^-l * dLevel_l[q, t>>l] ) / sqrt(D)       (fold, in place in level 0)
//   dF1[d,q]  = sum_t F2[d,t] * dC0[q,t]
//   dF2[d,t]  = sum_q F1[d,q] * dC0[q,t]
#include "common.h"

namespace raftcorr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FoldArgs {
    LevelSet dlv;
    int L, H, W;
    size_t BN;          // B * N query slabs
    float scale;        // 1/sqrt(D)
    float sqrt_d;
    int exact;
};

__global__ __launch_bounds__(256)
void corr_fold_levels_kernel(FoldArgs p)
{
    const int HW = p.H * p.W;
    const size_t total = p.BN * (size_t)HW;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t qb = e / HW;
        const int t = (int)(e - qb * HW);
        const int h2 = t / p.W, w2 = t - h2 * p.W;
        float g = p.dlv.ptr[0][e];
        float wgt = 0.25f;
        for (int l = 1; l < p.L; ++l) {
            const int hc = h2 >> l, wc = w2 >> l;
            const int Hl = p.dlv.h[l], Wl = p.dlv.w[l];
            if (hc >= Hl || wc >= Wl) break;        // not covered at level l => not at l+1 either
            g += wgt * p.dlv.ptr[l][(qb * Hl + hc) * Wl + wc];
            wgt *= 0.25f;
        }
        p.dlv.ptr[0][e] = p.exact ? g * p.scale : g / p.sqrt_d;
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

int launch_build_bwd(const float* f1, const float* f2, int B, int D, int H, int W, int L,
                     const LevelSet& dlv, float* df1, float* df2, hipStream_t s)
{
    const int N = H * W;
    FoldArgs f;
    f.dlv = dlv; f.L = L; f.H = H; f.W = W; f.BN = (size_t)B * N;
    const float sq = sqrtf((float)D);
    int e; const float mant = frexpf(sq, &e);
    f.exact = (mant == 0.5f) && (sq * sq == (float)D);
    f.scale = 1.0f / sq; f.sqrt_d = sq;
    const size_t total = f.BN * (size_t)N;
    const unsigned g = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(corr_fold_levels_kernel, dim3(g), dim3(256), 0, s, f);
    int rc = check_launch("corr_fold_levels_kernel");
    if (rc) return rc;

    GemmArgs a;
    // dF1[d][q] = sum_t F2[d][t] * dC0[q][t]      (B operand: k = t contiguous)
    a.A = f2; a.B = dlv.ptr[0]; a.C = df1; a.M = D; a.N = N; a.K = N;
    a.lda = N; a.ldb = N; a.ldc = N;
    a.sA = (size_t)D * N; a.sB = (size_t)N * N; a.sC = (size_t)D * N;
    dim3 grid((N + 63) / 64, (D + 63) / 64, B);
    hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, s, a);
    rc = check_launch("gemm_f32_kernel<dF1>");
    if (rc) return rc;
    // dF2[d][t] = sum_q F1[d][q] * dC0[q][t]      (B operand: n = t contiguous)
    a.A = f1; a.C = df2;
    hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, s, a);
    return check_launch("gemm_f32_kernel<dF2>");
}

}  // namespace raftcorr
