// On-the-fly ("alternate") correlation: the memory-lean path that never materialises the
// all-pairs volume.  Forward and backward, single level (drop-in for the reference's pybind
// functions) and all levels fused (what AlternateCorrBlock.__call__ needs per iteration).
//
// Replaces alt_cuda_corr (alt_cuda_corr/correlation_kernel.cu:19-119, :123-256) and the
// per-level Python glue of core/corr.py:75-92.  The reference kernel runs 32-thread blocks,
// re-reads fmap2 through a 32-channel shared-memory chunk with two barriers per tap and does
// 3200 global read-modify-writes per pixel.  Here:
//   * one WAVE (64 lanes) owns a query; lanes run along the channel dim, so a target pixel's
//     feature vector is one coalesced 1 KiB read (4 channels per lane at C = 256);
//   * the dot product of a tap is finished with a wavefront butterfly reduction (no LDS, no
//     barrier); lane t keeps tap t, the 2x2 bilinear blend is done by cross-lane reads;
//   * a block stages its 32 queries' results in LDS and writes every output channel as one
//     128-byte run; the output is written exactly once (no zero-fill + accumulate);
//   * backward: d(fmap1) is accumulated in registers across taps, levels and coordinate sets
//     and stored once (no atomics); d(fmap2) collides across queries and uses float atomics,
//     one 1 KiB wave-instruction per tap.
#include "common.h"

namespace raftcorr {

constexpr int AQ = 32;          // queries per block
constexpr int MAXCH = 4;        // channel chunks of 256 a lane can hold (C <= 1024)

struct AltArgs {
    const float* f1;            // (B, H1, W1, C)
    LevelSet f2;                // level l: (B, H_l, W_l, C)
    LevelSet df2;               // bwd: gradients, atomically accumulated
    const float* coords;
    const float* dout;          // bwd: (B*Nc, L*rd2, P)
    float* out;                 // fwd: (B*Nc, L*rd2, P)
    float* df1;                 // bwd: (B, H1, W1, C)
    int B, Nc, P, C, L, r;      // P = H1*W1 queries per image
    int planar;                 // 0: coords (B,Nc,H1,W1,2); 1: coords (B,2,H1,W1) (Nc == 1)
    int level_scale;            // divide coords by 2**l
    float out_div;              // 1 (drop-in) or sqrt(C): corr / sqrt(dim), corr.py:92
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ void alt_xy(const AltArgs& p, int b, int n, int q, int l, float& x, float& y)
{
    if (p.planar) {
        const float* c = p.coords + (size_t)b * 2 * p.P + q;
        x = c[0]; y = c[p.P];
    } else {
        const float* c = p.coords + (((size_t)b * p.Nc + n) * p.P + q) * 2;
        x = c[0]; y = c[1];
    }
    if (p.level_scale) { const float s = 1.0f / (float)(1 << l); x *= s; y *= s; }
}

// grid: (ceil(P/AQ), L, B*Nc); block 256 = 4 waves, wave w handles queries w*8 .. w*8+7
__global__ __launch_bounds__(256)
void alt_corr_fwd_kernel(AltArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [rd2][AQ]
    const int r = p.r, T = 2 * r + 2, TT = T * T, rd = 2 * r + 1, rd2 = rd * rd;
    const int l = blockIdx.y;
    const int bn = blockIdx.z, b = bn / p.Nc, n = bn - b * p.Nc;
    const int q0 = blockIdx.x * AQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, P = p.P;
    const int Hl = p.f2.g[l].h, Wl = p.f2.g[l].w;
    const float* __restrict__ f2 = p.f2.ptr[l] + (size_t)b * Hl * Wl * C;
    const bool vec = (C & 3) == 0;
    const int nch = (C + 255) / 256;

    for (int qi = wave * 8; qi < wave * 8 + 8; ++qi) {
        const int q = q0 + qi;
        if (q >= P) break;                                   // wave-uniform
        const float* __restrict__ f1 = p.f1 + ((size_t)b * P + q) * C;
        float4 a[MAXCH];
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
            a[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = ch * 256 + lane * 4;
            if (ch < nch) {
                if (vec) { if (c < C) a[ch] = *(const float4*)(f1 + c); }
                else {
                    if (c < C) a[ch].x = f1[c];
                    if (c + 1 < C) a[ch].y = f1[c + 1];
                    if (c + 2 < C) a[ch].z = f1[c + 2];
                    if (c + 3 < C) a[ch].w = f1[c + 3];
                }
            }
        }
        float x, y;
        alt_xy(p, b, n, q, l, x, y);
        int x0, y0; float fx, fy;
        split_coord(x, x0, fx);
        split_coord(y, y0, fy);
        float tap0 = 0.f, tap1 = 0.f, tap2 = 0.f, tap3 = 0.f;            // lane t keeps taps t, t+64, ...
        for (int t = 0; t < TT; ++t) {
            const int ty = t / T, tx = t - ty * T;
            const int yi = y0 - r + ty, xi = x0 - r + tx;
            if (xi < 0 || xi >= Wl || yi < 0 || yi >= Hl) continue;     // wave-uniform; tap stays 0
            const float* __restrict__ g = f2 + ((size_t)yi * Wl + xi) * C;
            float s = 0.f;
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) {
                const int c = ch * 256 + lane * 4;
                if (ch < nch) {
                    if (vec) {
                        if (c < C) {
                            const float4 v = *(const float4*)(g + c);
                            s += a[ch].x * v.x + a[ch].y * v.y + a[ch].z * v.z + a[ch].w * v.w;
                        }
                    } else {
                        if (c < C) s += a[ch].x * g[c];
                        if (c + 1 < C) s += a[ch].y * g[c + 1];
                        if (c + 2 < C) s += a[ch].z * g[c + 2];
                        if (c + 3 < C) s += a[ch].w * g[c + 3];
                    }
                }
            }
            s = wave_sum(s);
            const int slot = t >> 6;
            if ((t & 63) == lane) {
                if (slot == 0) tap0 = s; else if (slot == 1) tap1 = s; else if (slot == 2) tap2 = s; else tap3 = s;
            }
        }
        // blend: lane handles cells ch = lane, lane + 64, ...
        auto tap_at = [&](int t) -> float {
            const int src = t & 63, slot = t >> 6;
            const float v0 = __shfl(tap0, src), v1 = __shfl(tap1, src);
            const float v2 = __shfl(tap2, src), v3 = __shfl(tap3, src);
            return slot == 0 ? v0 : slot == 1 ? v1 : slot == 2 ? v2 : v3;
        };
        for (int base = 0; base < rd2; base += 64) {
            const int ch = base + lane;
            const int chc = ch < rd2 ? ch : 0;
            const int ai = chc / rd, bi = chc - ai * rd;               // ai <-> x offset, bi <-> y offset
            const int t00 = bi * T + ai;
            const float v00 = tap_at(t00), v01 = tap_at(t00 + 1);
            const float v10 = tap_at(t00 + T), v11 = tap_at(t00 + T + 1);
            const float o = v00 * ((1.f - fx) * (1.f - fy)) + v01 * (fx * (1.f - fy)) +
                            v10 * ((1.f - fx) * fy) + v11 * (fx * fy);
            if (ch < rd2) smem[ch * AQ + qi] = o / p.out_div;
        }
    }
    __syncthreads();
    float* __restrict__ out = p.out + ((size_t)bn * p.L * rd2 + (size_t)l * rd2) * P + q0;
    const int qi = tid & (AQ - 1);
    if (q0 + qi < P)
        for (int ch = tid / AQ; ch < rd2; ch += 256 / AQ) out[(size_t)ch * P + qi] = smem[ch * AQ + qi];
}

// grid: (ceil(P/AQ), 1, B); a wave walks its 8 queries x Nc coordinate sets x L levels and keeps
// d(fmap1) of the query in registers.
__global__ __launch_bounds__(256)
void alt_corr_bwd_kernel(AltArgs p)
{
    const int r = p.r, T = 2 * r + 2, TT = T * T, rd = 2 * r + 1, rd2 = rd * rd;
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * AQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, P = p.P;
    const bool vec = (C & 3) == 0;
    const int nch = (C + 255) / 256;

    for (int qi = wave * 8; qi < wave * 8 + 8; ++qi) {
        const int q = q0 + qi;
        if (q >= P) break;
        const float* __restrict__ f1 = p.f1 + ((size_t)b * P + q) * C;
        float4 a[MAXCH], da[MAXCH];
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
            a[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
            da[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = ch * 256 + lane * 4;
            if (ch < nch) {
                if (vec) { if (c < C) a[ch] = *(const float4*)(f1 + c); }
                else {
                    if (c < C) a[ch].x = f1[c];
                    if (c + 1 < C) a[ch].y = f1[c + 1];
                    if (c + 2 < C) a[ch].z = f1[c + 2];
                    if (c + 3 < C) a[ch].w = f1[c + 3];
                }
            }
        }
        for (int n = 0; n < p.Nc; ++n)
        for (int l = 0; l < p.L; ++l) {
            const int Hl = p.f2.g[l].h, Wl = p.f2.g[l].w;
            const float* __restrict__ f2 = p.f2.ptr[l] + (size_t)b * Hl * Wl * C;
            float* __restrict__ d2 = p.df2.ptr[l] + (size_t)b * Hl * Wl * C;
            float x, y;
            alt_xy(p, b, n, q, l, x, y);
            int x0, y0; float fx, fy;
            split_coord(x, x0, fx);
            split_coord(y, y0, fy);
            const float* __restrict__ dout =
                p.dout + (((size_t)b * p.Nc + n) * p.L * rd2 + (size_t)l * rd2) * P + q;
            // lane t computes the gradient of tap t (+64, ...): 2x2 box filter of d(out)
            float gt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int slot = 0; slot < 4; ++slot) {
                const int t = slot * 64 + lane;
                if (t < TT) {
                    const int ty = t / T, tx = t - ty * T;
                    float g = 0.f;
                    if (tx < rd && ty < rd) g += dout[(size_t)(tx * rd + ty) * P] * ((1.f - fx) * (1.f - fy));
                    if (tx > 0 && ty < rd)  g += dout[(size_t)((tx - 1) * rd + ty) * P] * (fx * (1.f - fy));
                    if (tx < rd && ty > 0)  g += dout[(size_t)(tx * rd + ty - 1) * P] * ((1.f - fx) * fy);
                    if (tx > 0 && ty > 0)   g += dout[(size_t)((tx - 1) * rd + ty - 1) * P] * (fx * fy);
                    gt[slot] = g / p.out_div;
                }
            }
            for (int t = 0; t < TT; ++t) {
                const int ty = t / T, tx = t - ty * T;
                const int yi = y0 - r + ty, xi = x0 - r + tx;
                if (xi < 0 || xi >= Wl || yi < 0 || yi >= Hl) continue;  // wave-uniform
                const int slot = t >> 6;
                const float gsel = slot == 0 ? gt[0] : slot == 1 ? gt[1] : slot == 2 ? gt[2] : gt[3];
                const float g = __shfl(gsel, t & 63);
                const size_t off = ((size_t)yi * Wl + xi) * C;
#pragma unroll
                for (int ch = 0; ch < MAXCH; ++ch) {
                    const int c = ch * 256 + lane * 4;
                    if (ch < nch) {
                        if (vec) {
                            if (c < C) {
                                const float4 v = *(const float4*)(f2 + off + c);
                                da[ch].x += g * v.x; da[ch].y += g * v.y;
                                da[ch].z += g * v.z; da[ch].w += g * v.w;
                                atomicAdd(d2 + off + c, g * a[ch].x);
                                atomicAdd(d2 + off + c + 1, g * a[ch].y);
                                atomicAdd(d2 + off + c + 2, g * a[ch].z);
                                atomicAdd(d2 + off + c + 3, g * a[ch].w);
                            }
                        } else {
                            if (c < C)     { da[ch].x += g * f2[off + c];     atomicAdd(d2 + off + c, g * a[ch].x); }
                            if (c + 1 < C) { da[ch].y += g * f2[off + c + 1]; atomicAdd(d2 + off + c + 1, g * a[ch].y); }
                            if (c + 2 < C) { da[ch].z += g * f2[off + c + 2]; atomicAdd(d2 + off + c + 2, g * a[ch].z); }
                            if (c + 3 < C) { da[ch].w += g * f2[off + c + 3]; atomicAdd(d2 + off + c + 3, g * a[ch].w); }
                        }
                    }
                }
            }
        }
        float* __restrict__ d1 = p.df1 + ((size_t)b * P + q) * C;
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
            const int c = ch * 256 + lane * 4;
            if (ch < nch) {
                if (vec) { if (c < C) *(float4*)(d1 + c) = da[ch]; }
                else {
                    if (c < C) d1[c] = da[ch].x;
                    if (c + 1 < C) d1[c + 1] = da[ch].y;
                    if (c + 2 < C) d1[c + 2] = da[ch].z;
                    if (c + 3 < C) d1[c + 3] = da[ch].w;
                }
            }
        }
    }
}

static int alt_check(int C, int r, const char* who)
{
    if (C > 256 * MAXCH) return fail(RAFTCORR_EUNSUPPORTED, "%s: C = %d > %d", who, C, 256 * MAXCH);
    if ((2 * r + 2) * (2 * r + 2) > 256) return fail(RAFTCORR_EUNSUPPORTED, "%s: radius %d > 7", who, r);
    return RAFTCORR_OK;
}

static int alt_fwd_common(AltArgs& a, hipStream_t s)
{
    const int rd = 2 * a.r + 1;
    const size_t sm = sizeof(float) * rd * rd * AQ;
    dim3 grid((a.P + AQ - 1) / AQ, a.L, a.B * a.Nc), block(256);
    hipLaunchKernelGGL(alt_corr_fwd_kernel, grid, block, sm, s, a);
    return check_launch("alt_corr_fwd_kernel");
}

static int alt_bwd_common(AltArgs& a, hipStream_t s)
{
    dim3 grid((a.P + AQ - 1) / AQ, 1, a.B), block(256);
    hipLaunchKernelGGL(alt_corr_bwd_kernel, grid, block, 0, s, a);
    return check_launch("alt_corr_bwd_kernel");
}

int launch_alt_level_fwd(const float* f1, const float* f2, const float* coords, float* corr,
                         int B, int N, int H1, int W1, int H2, int W2, int C, int r, hipStream_t s)
{
    int rc = alt_check(C, r, "alt_forward");
    if (rc) return rc;
    AltArgs a = {};
    a.f1 = f1; a.f2.ptr[0] = const_cast<float*>(f2); a.f2.g[0].h = H2; a.f2.g[0].w = W2;
    a.coords = coords; a.out = corr;
    a.B = B; a.Nc = N; a.P = H1 * W1; a.C = C; a.L = 1; a.r = r;
    a.planar = 0; a.level_scale = 0; a.out_div = 1.0f;
    return alt_fwd_common(a, s);
}

int launch_alt_level_bwd(const float* f1, const float* f2, const float* coords, const float* cg,
                         float* g1, float* g2, float* gc,
                         int B, int N, int H1, int W1, int H2, int W2, int C, int r, hipStream_t s)
{
    int rc = alt_check(C, r, "alt_backward");
    if (rc) return rc;
    hipError_t e = hipMemsetAsync(g2, 0, sizeof(float) * (size_t)B * H2 * W2 * C, s);
    if (e == hipSuccess && gc) e = hipMemsetAsync(gc, 0, sizeof(float) * (size_t)B * N * H1 * W1 * 2, s);
    if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "alt_backward: memset: %s", hipGetErrorString(e));
    AltArgs a = {};
    a.f1 = f1; a.f2.ptr[0] = const_cast<float*>(f2); a.f2.g[0].h = H2; a.f2.g[0].w = W2;
    a.df2.ptr[0] = g2; a.df2.g[0].h = H2; a.df2.g[0].w = W2;
    a.coords = coords; a.dout = cg; a.df1 = g1;
    a.B = B; a.Nc = N; a.P = H1 * W1; a.C = C; a.L = 1; a.r = r;
    a.planar = 0; a.level_scale = 0; a.out_div = 1.0f;
    return alt_bwd_common(a, s);
}

}  // namespace raftcorr
