// All-pairs correlation volume + pooled pyramid in one launch (gfx950 / CDNA4).
//
// Replaces CorrBlock.__init__ / CorrBlock.corr of the reference (core/corr.py:13-27,53-61):
//   torch.matmul(fmap1^T, fmap2) -> / sqrt(D) -> (L-1) x F.avg_pool2d(2, 2)
// i.e. one sgemm, one elementwise pass and three pooling passes over a 200-260 MB volume.
// Here it is one kernel: an fp32-MFMA contraction over D whose epilogue scales the tile and
// produces the level-1..3 cells straight from the accumulator registers.
//
// Tiling.  C[q, t] = sum_d F1[d, q] * F2[d, t].  Both feature maps are stored (D, N) with
// the pixel index contiguous, which is exactly the operand shape of v_mfma_f32_32x32x2_f32
// (lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]): LDS tiles are kept [k][m]
// and every fragment read is one conflict-free ds_read_b32.
//   block  = 256 threads = 4 waves, 128 queries x 128 targets
//   wave w = 32 queries (rows 32w..32w+31) x all 128 targets  -> 4 accumulators of 32x32
//   the 128 targets of a tile are an 8-row x 16-column PATCH of the target image, so every
//   2x2, 4x4 and 8x8 pooling block lies inside one wave's registers:
//     accumulator n  <-> patch rows 2n, 2n+1;  lane&15 <-> patch column, (lane>>4)&1 <-> row
//     level 1 = lanes xor 1, xor 16;  level 2 = accumulators (2m, 2m+1) + lanes xor 2;
//     level 3 = both halves + lanes xor 4.
// K loop: BK = 32, register-staged double buffering (global loads of tile k+1 are in flight
// while tile k is multiplied), one barrier per K tile.
#include "common.h"

namespace raftcorr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;      // queries per block
constexpr int BN = 128;      // targets per block (8 x 16 patch)
constexpr int BK = 32;
constexpr int PATCH_H = 8;
constexpr int PATCH_W = 16;
constexpr int GROUP_M = 8;   // query tiles walked together (L2 reuse of the operand panels)

struct BuildArgs {
    const float* f1;
    const float* f2;
    LevelSet lv;
    int B, D, H, W, N, L;
    int tilesM, PH, PW, tilesN;
    int nblk_per_b;
    float inv_sqrt_d;   // used when exact (sqrt(D) is a power of two)
    float sqrt_d;       // divisor otherwise (bit-faithful to corr / sqrt(D), corr.py:61)
};

template <bool EXACT_MUL>
__global__ __launch_bounds__(256, 2)
void corr_build_f32_kernel(BuildArgs p)
{
    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];

    // ---- block -> (batch, query tile, target patch), XCD-aware ----------------------------
    // Blocks are dealt round-robin over the 8 XCDs; give every XCD a contiguous chunk of the
    // tile list (bijective form, any grid size) and walk that chunk GROUP_M query tiles at a
    // time, so the blocks resident together on one XCD share a few A and B panels in its L2.
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7;
    const int qd = nwg >> 3, rm = nwg & 7;
    const int lin = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    const int b = lin / p.nblk_per_b;
    const int t = lin - b * p.nblk_per_b;
    const int per_group = GROUP_M * p.tilesN;
    const int grp = t / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(GROUP_M, p.tilesM - first_m);
    const int within = t - grp * per_group;
    const int tm = first_m + within % gsz;
    const int tn = within / gsz;
    const int ph = tn / p.PW;
    const int pw = tn - ph * p.PW;
    const int m0 = tm * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int N = p.N, D = p.D, H = p.H, W = p.W;

    // ---- staging addresses (fixed over the K loop) ---------------------------------------
    const int c = tid & 127;          // column of the A / B tile this thread stages
    const int r0 = tid >> 7;          // first k row (0/1), step 2
    const int qa = m0 + c;
    const bool va = qa < N;
    const float* pa = p.f1 + (size_t)b * D * N + (va ? qa : 0);
    const int h2s = ph * PATCH_H + (c >> 4);
    const int w2s = pw * PATCH_W + (c & 15);
    const bool vb = (h2s < H) && (w2s < W);
    const float* pb = p.f2 + (size_t)b * D * N + (vb ? h2s * W + w2s : 0);

    float ra[BK / 2], rb[BK / 2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < BK / 2; ++i) {
            const int k = k0 + r0 + 2 * i;
            const bool vk = k < D;
            ra[i] = (va && vk) ? pa[(size_t)k * N] : 0.f;
            rb[i] = (vb && vk) ? pb[(size_t)k * N] : 0.f;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BK / 2; ++i) {
            As[buf][r0 + 2 * i][c] = ra[i];
            Bs[buf][r0 + 2 * i][c] = rb[i];
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

    const int nk = (D + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();

    const int kh = lane >> 5;       // which of the 2 k rows of an MFMA step this lane feeds
    const int l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a = As[buf][2 * kk + kh][wave * 32 + l31];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float bv = Bs[buf][2 * kk + kh][n * 32 + l31];
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[n], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: scale, store level 0, pool levels 1..3 in registers --------------------
    // C/D map of the 32x32 MFMA: column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    const int dx = lane & 15;
    const int dyl = (lane >> 4) & 1;
    const int w2 = pw * PATCH_W + dx;
    const int L = p.L;
    const int H1 = p.lv.h[1], W1 = p.lv.w[1];
    const int H2 = p.lv.h[2], W2 = p.lv.w[2];
    const int H3 = p.lv.h[3], W3 = p.lv.w[3];
    float* __restrict__ o0 = p.lv.ptr[0];
    float* __restrict__ o1 = p.lv.ptr[1];
    float* __restrict__ o2 = p.lv.ptr[2];
    float* __restrict__ o3 = p.lv.ptr[3];

#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        const int q = m0 + wave * 32 + row;
        const bool vq = q < N;
        const size_t qb = (size_t)b * N + q;
        float v[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            v[n] = EXACT_MUL ? acc[n][reg] * p.inv_sqrt_d : acc[n][reg] / p.sqrt_d;
            const int h2 = ph * PATCH_H + 2 * n + dyl;
            if (vq && h2 < H && w2 < W) o0[(qb * H + h2) * W + w2] = v[n];
        }
        if (L > 1) {
            float s1[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float tt = v[n] + __shfl_xor(v[n], 1);
                tt += __shfl_xor(tt, 16);
                s1[n] = tt * 0.25f;
                const int hc = ph * 4 + n, wc = pw * 8 + (dx >> 1);
                if (vq && (lane & 17) == 0 && hc < H1 && wc < W1) o1[(qb * H1 + hc) * W1 + wc] = s1[n];
            }
            if (L > 2) {
                float s2[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float u = s1[2 * m] + s1[2 * m + 1];
                    u += __shfl_xor(u, 2);
                    s2[m] = u * 0.25f;
                    const int hc = ph * 2 + m, wc = pw * 4 + (dx >> 2);
                    if (vq && (lane & 19) == 0 && hc < H2 && wc < W2) o2[(qb * H2 + hc) * W2 + wc] = s2[m];
                }
                if (L > 3) {
                    float u = s2[0] + s2[1];
                    u += __shfl_xor(u, 4);
                    const float s3 = u * 0.25f;
                    const int hc = ph, wc = pw * 2 + (dx >> 3);
                    if (vq && (lane & 23) == 0 && hc < H3 && wc < W3) o3[(qb * H3 + hc) * W3 + wc] = s3;
                }
            }
        }
    }
}

// levels >= 4: plain 2x2 mean of the previous level (rare: RAFT uses 4 levels)
__global__ void pool2x2_kernel(const float* __restrict__ in, float* __restrict__ out,
                               size_t M, int Hi, int Wi, int Ho, int Wo)
{
    const size_t total = M * (size_t)Ho * Wo;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int wo = (int)(e % Wo);
        const size_t r = e / Wo;
        const int ho = (int)(r % Ho);
        const size_t m = r / Ho;
        const float* s = in + (m * Hi + 2 * ho) * Wi + 2 * wo;
        out[e] = ((s[0] + s[1]) + (s[Wi] + s[Wi + 1])) * 0.25f;
    }
}

int launch_build_pyramid(const float* f1, const float* f2, int B, int D, int H, int W, int L,
                         const LevelSet& lv, hipStream_t s)
{
    BuildArgs a;
    a.f1 = f1; a.f2 = f2; a.lv = lv;
    for (int l = L; l < RAFTCORR_MAX_LEVELS; ++l) { a.lv.ptr[l] = nullptr; a.lv.h[l] = 0; a.lv.w[l] = 0; }
    a.B = B; a.D = D; a.H = H; a.W = W; a.N = H * W; a.L = L < 4 ? L : 4;
    a.tilesM = (a.N + BM - 1) / BM;
    a.PH = (H + PATCH_H - 1) / PATCH_H;
    a.PW = (W + PATCH_W - 1) / PATCH_W;
    a.tilesN = a.PH * a.PW;
    a.nblk_per_b = a.tilesM * a.tilesN;
    const float sq = sqrtf((float)D);
    a.sqrt_d = sq;
    a.inv_sqrt_d = 1.0f / sq;
    int e; const float mant = frexpf(sq, &e);
    const bool exact = (mant == 0.5f) && (sq * sq == (float)D);   // sqrt(D) is a power of two
    const long long nblk = (long long)a.nblk_per_b * B;
    if (nblk > 0x7fffffffLL) return fail(RAFTCORR_EUNSUPPORTED, "build_pyramid: grid too large");
    dim3 grid((unsigned)nblk), block(256);
    if (exact) hipLaunchKernelGGL(corr_build_f32_kernel<true>, grid, block, 0, s, a);
    else       hipLaunchKernelGGL(corr_build_f32_kernel<false>, grid, block, 0, s, a);
    int rc = check_launch("corr_build_f32_kernel");
    if (rc) return rc;
    for (int l = 4; l < L; ++l) {
        const size_t M = (size_t)B * a.N;
        const size_t total = M * lv.h[l] * lv.w[l];
        if (total == 0) continue;
        const unsigned g = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(pool2x2_kernel, dim3(g), dim3(256), 0, s, lv.ptr[l - 1], lv.ptr[l], M,
                           lv.h[l - 1], lv.w[l - 1], lv.h[l], lv.w[l]);
        rc = check_launch("pool2x2_kernel");
        if (rc) return rc;
    }
    return RAFTCORR_OK;
}

}  // namespace raftcorr
