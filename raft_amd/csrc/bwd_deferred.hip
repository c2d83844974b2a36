// Deferred backward of the window lookups (training path, BASELINE configs[3]).
//
// Replaces what autograd runs behind core/corr.py:26,46 in the reference for EVERY lookup of a frame pair
// (grid_sampler_2d_backward: a dense zero-filled gradient of each pyramid level + a scatter-add, then
// avg_pool2d_backward down the levels).  Round 1-2 of this library accumulated the lookups into ONE gradient pyramid
// (raftcorr_lookup_bwd: read-modify-write per lookup) and folded it once (corr_fold_levels_kernel): still three
// passes over a dense B*N*(S0+S1+S2+S3) buffer -- zero fill, 12 sparse read-modify-write passes, fold.
//
// Here a lookup's backward pass only RECORDS (coords_k, d(out)_k); when autograd reaches the build node, one kernel
// makes the folded level-0 gradient row of every query in LDS and writes it once:
//     row_l[q] = sum_k splat(d(out)_k[q, level l], coords_k[q] / 2^l)          (all L levels, K lookups, in LDS)
//     dC0[q][p] = ( row_0[p] + row_1[parent_1(p)]/4 + row_2[parent_2(p)]/16 + row_3[parent_3(p)]/64 ) / sqrt(D)
// No zero fill, no read-modify-write of global memory, no separate fold pass; the only global traffic is the
// records (read once) and dC0 (written once, padding columns as exact zeros).  The two feature-gradient
// contractions of build_bwd.hip then read dC0 as before.
//
//  * corr_record_kernel: d(out) arrives as (B, C, H, W) in canonical mode, i.e. a query's C channels are N floats
//    apart; the splat kernel works query by query, so the record is the transposed copy (B, N, C) (LDS tile
//    transposition, 128-byte row segments in, whole contiguous rows out).  Fork-mode d(out) is (B, N, C) already.
//  * corr_splat_fold_kernel: persistent workgroups, one query at a time.  The query's records of all K lookups
//    (K * C floats, contiguous per lookup) are staged in LDS, fetched into registers one query AHEAD (the loads fly
//    under the previous query's splat and fold).  Wave w owns pyramid level w: for every lookup it turns the level's
//    (2r+1)^2 d(out) cells into the (2r+2)^2 tap gradients (2x2 box filter with the query's fractional offsets, the
//    arithmetic of corr_lookup_bwd_kernel) and adds them into the level's LDS row -- the taps of one lookup are
//    distinct cells, lookups follow each other in program order inside the wave, the levels' rows are disjoint:
//    no atomics, no barriers inside the splat, a fixed summation order (bit-reproducible).  Then all threads fold
//    (parents by bit arithmetic in patch-tile order, layout.h) and store the level-0 row with 16-byte stores,
//    tracking max|dC0| for the range scaling of the fp16x2 contractions.
#include "common.h"
#include <algorithm>

namespace raftcorr {

namespace {

constexpr int DEF_MAXK = 16;                 // lookups per launch (more: further launches that add to dC0)
constexpr int DEF_STAGE_FLOATS = 6144;       // staging capacity (K * C floats per query)
constexpr int DEF_MAX_LDS = 150 * 1024;
constexpr unsigned DEF_MAX_GRID = 32768;     // = FOLD_MAX_GRID of build_bwd.hip (per-workgroup maxima)

constexpr int REC_MAXK = 16;  // lookups per launch of corr_record_rows_kernel (blockIdx.z)
struct RecordArgs {
    const float* dout;   // (B, C, N)
    float* rec;          // (B, N, C)
    int N, C, CH;        // CH: channels per tile
    const float* douts[REC_MAXK];    // corr_record_rows_kernel: lookup blockIdx.z
    float* recs[REC_MAXK];
};

// grid (ceil(N / 32), ceil(C / CH), B), block 256, LDS CH * 33 floats; CH <= 128: every load of a thread is in flight
// before the first is used, and a launch has several times more workgroups than the chip has slots
__global__ __launch_bounds__(256)
void corr_record_kernel(RecordArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [CH][33]
    const int tid = threadIdx.x, q0 = blockIdx.x * 32, c0 = blockIdx.y * p.CH, b = blockIdx.z;
    const int nc = min(p.CH, p.C - c0), nq = min(32, p.N - q0);
    const float* __restrict__ src = p.dout + ((size_t)b * p.C + c0) * p.N + q0;
    const int qi = tid & 31, cr = tid >> 5;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = min(cr + 8 * i, nc - 1);
        v[i] = src[(size_t)c * p.N + min(qi, nq - 1)];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = cr + 8 * i;
        if (c < nc) tile[c * 33 + qi] = v[i];
    }
    __syncthreads();
    float* __restrict__ dst = p.rec + ((size_t)b * p.N + q0) * p.C + c0;
    if (((nc | p.C | c0) & 3) == 0 && ((uintptr_t)p.rec & 15) == 0) {  // 16-byte stores: 4 channels of one query
        const int n4 = nc >> 2;
        for (int e = tid; e < nq * n4; e += 256) {
            const int q = e / n4, c = (e - q * n4) * 4;
            const float4 o = make_float4(tile[c * 33 + q], tile[(c + 1) * 33 + q], tile[(c + 2) * 33 + q], tile[(c + 3) * 33 + q]);
            *(float4*)(dst + (size_t)q * p.C + c) = o;
        }
    } else {
        for (int e = tid; e < nq * nc; e += 256) {
            const int q = e / nc, c = e - q * nc;
            dst[(size_t)q * p.C + c] = tile[c * 33 + q];
        }
    }
}

// The same transposition with ALL channels of QB queries in one workgroup (C % 4 == 0, C <= 512): a thread loads four
// consecutive channels of its query and writes them to LDS as one 16-byte slot of the query's row (row pitch an odd
// number of slots: the 16 rows a quarter-wave writes fall on different banks), and the workgroup's part of the record
// -- QB complete rows, QB * C floats -- goes out as ONE contiguous run (21 KB at C = 324, QB = 16) instead of 432-byte
// pieces.  grid (ceil(N / QB), B), block 256, dynamic LDS QB * (C / 4 | 1) * 16 bytes; IT: channel quads per thread
// (IT * 256 / QB >= C / 4)
template <int IT, int QB>
__global__ __launch_bounds__(256)
void corr_record_rows_kernel(RecordArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float4 rows4[];   // [QB][S4]
    // Workgroup ids go round-robin over the 8 XCDs; every XCD takes a contiguous run of (image, query block) instead: a
    // channel's QB floats of a query block are part of a 128-byte line that the neighbouring block needs too.
    int b, q0;
    {
        const int total = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = lin & 7, qd = total >> 3, rm = total & 7;
        const int at = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (lin >> 3);
        b = at / gridDim.x;
        q0 = (at - b * gridDim.x) * QB;
    }
    constexpr int CGS = 256 / QB;                                    // channel quads loaded side by side
    const int tid = threadIdx.x;
    const int C4 = p.C >> 2, S4 = C4 | 1, nq = min(QB, p.N - q0);
    const int qi = tid & (QB - 1), cg = tid / QB;
    const float* __restrict__ src = p.douts[blockIdx.z] + (size_t)b * p.C * p.N + q0 + min(qi, nq - 1);
    float4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {                                   // every load in flight before the first use
        const int c = 4 * min(cg + CGS * i, C4 - 1);
        v[i].x = src[(size_t)c * p.N]; v[i].y = src[(size_t)(c + 1) * p.N];
        v[i].z = src[(size_t)(c + 2) * p.N]; v[i].w = src[(size_t)(c + 3) * p.N];
    }
#pragma unroll
    for (int i = 0; i < IT; ++i)
        if (cg + CGS * i < C4) rows4[qi * S4 + cg + CGS * i] = v[i];
    __syncthreads();
    float4* __restrict__ dst = (float4*)(p.recs[blockIdx.z] + ((size_t)b * p.N + q0) * p.C);
    for (int e = tid; e < nq * C4; e += 256) {
        const int q = e / C4, c4 = e - q * C4;
        dst[e] = rows4[q * S4 + c4];
    }
}

struct DeferredArgs {
    const float* rec[DEF_MAXK];      // (B, N, C) each
    const float* coords[DEF_MAXK];   // canonical: (B, 2, H, W); fork: (B, N, 2)
    float* d0;                       // [B * N][S0]
    float* wgmax;                    // [gridDim.x] or nullptr
    int h[4], w[4], off[4];          // logical size and LDS row offset (floats) of each level
    int S0, PW;                      // level-0 slab (floats), patches per patch row
    int rows;                        // LDS floats of all level rows (multiple of 4)
    int K, B, N, L, r, mode, accumulate;
    float scale, sqrt_d;
    int exact;
};

// position of cell (y, x) inside a level's row: layout.h:level_pos with the level kind at compile time
template <int KIND>
__device__ __forceinline__ int cell_pos(int pw, int y, int x)
{
    if (KIND == 0) {
        const int patch = (y >> 3) * pw + (x >> 4);
        const int n = (((y >> 2) & 1) << 1) | ((x >> 3) & 1);
        const int dy = y & 3, dx = x & 7;
        return patch * 128 + n * 32 + ((dx & 1) | ((dy & 1) << 1) | ((dy >> 1) << 2) | (((dx >> 1) & 1) << 3) | ((dx >> 2) << 4));
    } else if (KIND == 1) {
        const int patch = (y >> 2) * pw + (x >> 3);
        const int n = (((y >> 1) & 1) << 1) | ((x >> 2) & 1);
        return patch * 32 + n * 8 + ((x & 1) | ((y & 1) << 1) | (((x >> 1) & 1) << 2));
    } else if (KIND == 2) {
        return ((y >> 1) * pw + (x >> 2)) * 8 + ((x & 1) | ((y & 1) << 1) | (((x >> 1) & 1) << 2));
    }
    return ((y >> 1) * pw + (x >> 1)) * 4 + (x & 1) + (y & 1) * 2;
}

// row[cell] += v.  A plain read-modify-write: the lanes of one instruction hit distinct cells (the taps of one
// lookup), a wave's LDS operations execute in program order and the levels' rows belong to different waves, so a cell
// receives its lookups' contributions in lookup order on every run.  (The return-less LDS float atomic, ds_add_f32,
// would do the same without the read's round trip, but runs at a fraction of the plain operations' rate here: the
// whole kernel 784 instead of 430 us, tools/deferred_lab.hip.)
__device__ __forceinline__ void lds_add(float* cell, float v)
{
#if defined(RC_LAB_DEF_ATOMIC)
    __hip_atomic_fetch_add(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#elif defined(RC_LAB_DEF_NOADD)
    asm volatile("" :: "v"(cell), "v"(v));
#else
    *cell += v;
#endif
}

// Staged d(out) of one (lookup, level): the (2r+1)^2 cells with a ZERO BORDER of one cell, P[a + 1][b + 1] = channel
// a * rd + b (a <-> x, b <-> y), pitch rd + 2.  Tap (tx, ty) of the window blends cells (tx, ty), (tx - 1, ty),
// (tx, ty - 1), (tx - 1, ty - 1) = P[tx + 1][ty + 1], P[tx][ty + 1], P[tx + 1][ty], P[tx][ty]: four reads at fixed
// offsets from one per-lane base, no validity tests (absent cells are border zeros).
__host__ __device__ constexpr int pad_pitch(int r) { return 2 * r + 3; }
__host__ __device__ constexpr int pad_size(int r) { return (pad_pitch(r) * pad_pitch(r) + 3) & ~3; }

// one wave, one level: the tap gradients of lookups 0 .. K-1 of the staged query, added into the level's LDS row.
// A lane keeps the same taps for every lookup (tap t = lane + 64 round).  What changes per lookup -- the window origin
// and the four blend weights -- is wave-uniform: lane k computes it for lookup k once per query, the loop broadcasts it
// with v_readlane (no LDS round trip in the chain).  Lookups go in pairs: the reads of both are in flight before the
// first read-modify-write.
template <int R, int KIND>
__device__ __forceinline__ void splat_level(const DeferredArgs& p, float* __restrict__ row, const float* __restrict__ stage,
                                            const float* __restrict__ cxy, int lane)
{
    const int r = R >= 0 ? R : p.r, T = 2 * r + 2, TT = T * T, PP = pad_pitch(r), PS = pad_size(r);
    const int hl = p.h[KIND], wl = p.w[KIND];
    constexpr int NR = R >= 0 ? ((2 * R + 2) * (2 * R + 2) + 63) / 64 : 4;               // rounds of 64 taps (r <= 7: 256 taps)
    int tx[NR], ty[NR], base[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int t = lane + 64 * i;
        ty[i] = t / T; tx[i] = t - ty[i] * T;
        base[i] = tx[i] * PP + ty[i];
        if (t >= TT) { tx[i] = -(1 << 20); base[i] = 0; }              // never inside the level
    }
    // lane k: window of lookup k at this level
    int pxs, pys; float pw00, pw10, pw01, pw11;
    {
        const int k = min(lane, DEF_MAXK - 1);
        const float cs = p.mode == RAFTCORR_MODE_FORK ? 1.0f : 1.0f / (float)(1 << KIND);    // coords / 2**l, exact
        int x0, y0; float fx, fy;
        split_coord(cxy[k] * cs, x0, fx);
        split_coord(cxy[DEF_MAXK + k] * cs, y0, fy);
        pxs = x0 - r; pys = y0 - r;
        pw00 = (1.f - fx) * (1.f - fy); pw10 = fx * (1.f - fy); pw01 = (1.f - fx) * fy; pw11 = fx * fy;
    }
    auto bc = [](float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); };
    const float* __restrict__ g = stage + KIND * PS;                   // lookup k: + k * L * PS
    const int kstride = p.L * PS;
    for (int k = 0; k < p.K; k += 2) {
        const bool two = k + 1 < p.K;                                  // (uniform)
        const int k1 = two ? k + 1 : k;
        const int xs0 = __builtin_amdgcn_readlane(pxs, k), ys0 = __builtin_amdgcn_readlane(pys, k);
        const int xs1 = __builtin_amdgcn_readlane(pxs, k1), ys1 = __builtin_amdgcn_readlane(pys, k1);
        const float a00 = bc(pw00, k), a10 = bc(pw10, k), a01 = bc(pw01, k), a11 = bc(pw11, k);
        const float b00 = bc(pw00, k1), b10 = bc(pw10, k1), b01 = bc(pw01, k1), b11 = bc(pw11, k1);
        const float* __restrict__ g1 = g + (two ? kstride : 0);
        float v0[NR], v1[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (R < 0 && 64 * i >= TT) { v0[i] = v1[i] = 0.f; continue; }      // (uniform)
            const float* c = g + base[i];
            const float* d = g1 + base[i];
            v0[i] = c[PP + 1] * a00 + c[1] * a10 + c[PP] * a01 + c[0] * a11;
            v1[i] = d[PP + 1] * b00 + d[1] * b10 + d[PP] * b01 + d[0] * b11;
        }
        // two lookups whose windows start at the same cell (at the coarse levels the rule once the flow has settled:
        // coords / 2^l moves by a fraction of a cell per iteration) hit the same cells: one read-modify-write for both
        const bool same = two && xs1 == xs0 && ys1 == ys0;             // (uniform)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (R < 0 && 64 * i >= TT) continue;
            const int X = xs0 + tx[i], Y = ys0 + ty[i];
            if ((unsigned)X < (unsigned)wl && (unsigned)Y < (unsigned)hl)
                lds_add(row + cell_pos<KIND>(p.PW, Y, X), same ? v0[i] + v1[i] : v0[i]);
        }
        if (two && !same) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                if (R < 0 && 64 * i >= TT) continue;
                const int X = xs1 + tx[i], Y = ys1 + ty[i];
                if ((unsigned)X < (unsigned)wl && (unsigned)Y < (unsigned)hl) lds_add(row + cell_pos<KIND>(p.PW, Y, X), v1[i]);
            }
        }
        g += 2 * kstride;
    }
}

// grid: min(B * N, 256 CUs x resident workgroups) persistent workgroups, block 256 = 4 waves
template <int R, int NST>
__global__ __launch_bounds__(256)
void corr_splat_fold_kernel(DeferredArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ const float* rptr[DEF_MAXK];
    __shared__ const float* cptr[DEF_MAXK];
    __shared__ float wm[4];
    const int r = R >= 0 ? R : p.r, rd = 2 * r + 1, rd2 = rd * rd, PP = pad_pitch(r), PS = pad_size(r);
    const int C = p.L * rd2, KC = p.K * C;
    const int nstage = p.K * p.L * PS;
    float* rows = sm;                                   // [p.rows]
    float* stage = sm + p.rows;                         // [K][L][PS] zero-bordered d(out) tiles, + 4 floats (dummy slot)
    float* cxy = stage + nstage + 4;                    // [2][DEF_MAXK]: x, y of the query in every lookup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned BN = (unsigned)p.B * (unsigned)p.N;

    for (int k = 0; k < p.K; ++k)                       // pointer tables: kernarg -> LDS (per-thread k below)
        if (tid == 0) { rptr[k] = p.rec[k]; cptr[k] = p.coords[k]; }
    for (int i = tid; i < nstage + 4; i += 256) stage[i] = 0.f;        // the borders stay zero for good
    __syncthreads();

    // The record pointers come out of LDS, i.e. as generic pointers: a load through one is a FLAT load, which counts
    // on the LDS counter as well -- every wait for an LDS read in the splat would then wait for global loads.
    typedef const __attribute__((address_space(1))) float* gfp;
    // staging plan of this thread: element e = tid + 256 i of the query's K * C record floats is read from
    // rec[k_i] + c_i + q * C and goes to the interior of tile (k_i, l_i) -- neither depends on the query
    gfp sbase[NST];
    int sdst[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + 256 * i, ec = min(e, KC - 1);
        const int k = ec / C, ch = ec - k * C, l = ch / rd2, cc = ch - l * rd2, ca = cc / rd, cb = cc - ca * rd;
        sbase[i] = (gfp)rptr[k] + ch;
        sdst[i] = e < KC ? (k * p.L + l) * PS + (ca + 1) * PP + (cb + 1) : nstage;      // beyond the records: dummy slot
    }
    // coordinates: thread t < 2 K fetches axis t / K of lookup t % K
    const int ck = tid % p.K, cax = min(tid / p.K, 1);
    const gfp cbase = (gfp)cptr[ck];
    const bool fork = p.mode == RAFTCORR_MODE_FORK;

    float sv[NST], cv = 0.f;
    auto fetch = [&](unsigned qq) {                     // global -> registers; unconditional loads (always valid addresses):
        const unsigned b = qq / (unsigned)p.N, q = qq - b * (unsigned)p.N;             // a load behind a run-time condition is
        const size_t o = (size_t)qq * C;                                               // branched around and waited for
#pragma unroll
        for (int i = 0; i < NST; ++i) sv[i] = sbase[i][o];
        cv = fork ? cbase[(size_t)qq * 2 + cax] : cbase[((size_t)b * 2 + cax) * p.N + q];
    };
    auto commit = [&]() {                               // registers -> LDS
#pragma unroll
        for (int i = 0; i < NST; ++i) stage[sdst[i]] = sv[i];
        if (tid < 2 * p.K) cxy[cax * DEF_MAXK + ck] = cv;
    };
    // fold plan: the level-0 quads of this thread are c = 4 tid + 1024 it; their level-3 parent needs patch / PW,
    // kept incrementally (patch grows by 8 per iteration)
    const int patch0 = tid >> 5;
    const int ph0 = patch0 / p.PW, pw0 = patch0 - ph0 * p.PW;

    float vmax = 0.f;
    unsigned qq = blockIdx.x;                           // B * N < 2^31 (the launcher checks)
    if (qq < BN) { fetch(qq); commit(); }
    __syncthreads();
    for (; qq < BN; qq += gridDim.x) {
        const unsigned next = qq + gridDim.x;
#if !defined(RC_LAB_DEF_NOFETCH)
        if (next < BN) fetch(next);                     // flies under this query's splat and fold
#endif
#if !defined(RC_LAB_DEF_NOZERO)
        for (int i = tid; i < p.rows / 4; i += 256) ((float4*)rows)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        __syncthreads();                                // rows zeroed

#if !defined(RC_LAB_DEF_NOSPLAT)
        for (int l = wave; l < p.L; l += 4) {           // wave-uniform
            switch (l) {
            case 0: splat_level<R, 0>(p, rows + p.off[0], stage, cxy, lane); break;
            case 1: splat_level<R, 1>(p, rows + p.off[1], stage, cxy, lane); break;
            case 2: splat_level<R, 2>(p, rows + p.off[2], stage, cxy, lane); break;
            default: splat_level<R, 3>(p, rows + p.off[3], stage, cxy, lane); break;
            }
        }
#endif
        __syncthreads();                                // rows complete; staging and coordinates no longer read

        if (next < BN) commit();                        // the next query's records (not read by the fold)
        // fold: one thread = four consecutive level-0 cells = one 2x2 block, which shares all its parents (layout.h)
        const float* r1 = rows + p.off[1]; const float* r2 = rows + p.off[2]; const float* r3 = rows + p.off[3];
        float* __restrict__ out = p.d0 + (size_t)qq * (size_t)p.S0;
        const int j = (tid * 4) & 31, n = (tid >> 3) & 3;
        const int dy1 = (j >> 2) & 1, dx1 = (j >> 3) & 1, dx2 = (j >> 4) & 1;
        const int q1 = n * 8 + (dx1 | (dy1 << 1) | (dx2 << 2)), q2 = dx2 | ((n >> 1) << 1) | ((n & 1) << 2);
        int ph = ph0, pwi = pw0;
#if defined(RC_LAB_DEF_NOFOLD)
        for (int c = tid * 4; c < 0; c += 1024) {
#else
        for (int c = tid * 4; c < p.S0; c += 1024) {
#endif
            const int patch = c >> 7;
            float4 v = *(const float4*)(rows + c);
            float g = 0.f;
            if (p.L > 1) g += 0.25f * r1[patch * 32 + q1];
            if (p.L > 2) g += 0.0625f * r2[patch * 8 + q2];
            if (p.L > 3) g += 0.015625f * r3[((ph >> 1) * p.PW + pwi) * 4 + (n & 1) + (ph & 1) * 2];
            pwi += 8;
            while (pwi >= p.PW) { pwi -= p.PW; ++ph; }
            v.x += g; v.y += g; v.z += g; v.w += g;
            if (p.exact) { v.x *= p.scale; v.y *= p.scale; v.z *= p.scale; v.w *= p.scale; }
            else { v.x /= p.sqrt_d; v.y /= p.sqrt_d; v.z /= p.sqrt_d; v.w /= p.sqrt_d; }
            if (p.accumulate) {
                const float4 o = *(const float4*)(out + c);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
#if !defined(RC_LAB_DEF_NOSTORE)
            *(float4*)(out + c) = v;
#endif
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        __syncthreads();                                // rows free again, next staging and coordinates visible
    }
    if (p.wgmax) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0) wm[wave] = vmax;
        __syncthreads();
        if (tid == 0) p.wgmax[blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    }
}

struct DeferredPlan {
    int rows, off[4], C, kc;
    size_t lds;
    bool ok;
};

// dynamic LDS of a launch that stages K lookups: level rows, zero-bordered d(out) tiles (+ dummy slot), coordinates
size_t deferred_lds(int rows, int K, int L, int r)
{
    return ((size_t)rows + (size_t)K * L * pad_size(r) + 4 + 2 * DEF_MAXK) * sizeof(float);
}

DeferredPlan deferred_plan(int H, int W, int L, int r)
{
    DeferredPlan pl = {};
    if (L < 1 || L > 4 || r < 0 || r > 7 || H < 1 || W < 1 || (H >> (L - 1)) < 1 || (W >> (L - 1)) < 1) return pl;
    long long rows = 0;
    for (int l = 0; l < L; ++l) { pl.off[l] = (int)rows; rows += level_geom(H, W, l).slab; rows = (rows + 3) & ~3LL; }
    for (int l = L; l < 4; ++l) pl.off[l] = 0;
    pl.C = L * (2 * r + 1) * (2 * r + 1);
    pl.kc = std::min(DEF_MAXK, DEF_STAGE_FLOATS / pl.C);     // K * C <= 24 staging registers per thread
    if (pl.kc < 1 || rows > (1 << 20)) return pl;
    pl.rows = (int)rows;
    pl.lds = deferred_lds(pl.rows, pl.kc, L, r);             // at full staging capacity
    pl.ok = pl.lds <= (size_t)DEF_MAX_LDS;
    return pl;
}

}  // namespace

bool build_bwd_deferred_supported(int H, int W, int L, int r) { return deferred_plan(H, W, L, r).ok; }

// K transpositions (B, C, N) -> (B, N, C).  With whole rows per workgroup they are ONE launch per 16 lookups: a single
// lookup's launch is one round of resident workgroups that first all read and then all write (18 us for 70 MB at
// Things B=6, launch latency included); twelve lookups in one launch are twelve rounds that overlap.
int launch_lookup_bwd_record_many(const float* const* douts, int K, int B, int N, int C, float* const* recs, hipStream_t s)
{
    bool rows = (C & 3) == 0 && C <= 512;
    for (int k = 0; k < K; ++k) rows = rows && ((uintptr_t)recs[k] & 15) == 0;
    if (rows) {                                                      // whole rows per workgroup
        // 16 queries per workgroup (21 KB of LDS at C = 324, 7 workgroups per CU): at Things B=6 the 1692 workgroups of
        // a lookup are resident together; with 32 queries (42 KB, 3 per CU) a launch was one round of 768 and one of 78
        constexpr int QB = 16;
        const int it = ((C >> 2) + 15) / 16;
        const size_t lds = (size_t)QB * ((C >> 2) | 1) * 16;
        for (int k0 = 0; k0 < K; k0 += REC_MAXK) {
            const int kn = std::min(REC_MAXK, K - k0);
            RecordArgs a = {};
            a.N = N; a.C = C; a.CH = C;
            for (int k = 0; k < kn; ++k) { a.douts[k] = douts[k0 + k]; a.recs[k] = recs[k0 + k]; }
            const dim3 grid((N + QB - 1) / QB, B, kn);
            if (it <= 2)      hipLaunchKernelGGL((corr_record_rows_kernel<2, QB>), grid, dim3(256), lds, s, a);
            else if (it <= 4) hipLaunchKernelGGL((corr_record_rows_kernel<4, QB>), grid, dim3(256), lds, s, a);     // r = 3, 4 levels
            else if (it <= 6) hipLaunchKernelGGL((corr_record_rows_kernel<6, QB>), grid, dim3(256), lds, s, a);     // r = 4, 4 levels
            else              hipLaunchKernelGGL((corr_record_rows_kernel<8, QB>), grid, dim3(256), lds, s, a);
            const int rc = check_launch("corr_record_rows_kernel");
            if (rc) return rc;
        }
        return RAFTCORR_OK;
    }
    for (int k = 0; k < K; ++k) {
        RecordArgs a = {};
        a.dout = douts[k]; a.rec = recs[k]; a.N = N; a.C = C;
        const int chunks = (C + 127) / 128;
        a.CH = std::min(128, ((C + chunks - 1) / chunks + 3) & ~3);
        const dim3 grid((N + 31) / 32, (C + a.CH - 1) / a.CH, B);
        hipLaunchKernelGGL(corr_record_kernel, grid, dim3(256), (size_t)a.CH * 33 * sizeof(float), s, a);
        const int rc = check_launch("corr_record_kernel");
        if (rc) return rc;
    }
    return RAFTCORR_OK;
}

int launch_lookup_bwd_record(const float* dout, int B, int N, int C, float* rec, hipStream_t s)
{
    return launch_lookup_bwd_record_many(&dout, 1, B, N, C, &rec, s);
}

// d0[B * N][S0] = folded, scaled level-0 gradient of the recorded lookups (+= when accumulate); wgmax: per-workgroup
// maxima of |d0| (n_wg entries are written, returned through *n_wg), or nullptr
int launch_splat_fold(int B, int H, int W, int L, int r, int mode, int D, int K, const float* const* coords,
                      const float* const* records, float* d0, int accumulate, float* wgmax, unsigned* n_wg, hipStream_t s)
{
    const DeferredPlan pl = deferred_plan(H, W, L, r);
    if (!pl.ok) return fail(RAFTCORR_EUNSUPPORTED, "build_bwd_deferred: %d x %d, %d levels, radius %d needs %zu bytes of LDS",
                            H, W, L, r, pl.lds);
    DeferredArgs a = {};
    const LevelGeom g0 = level_geom(H, W, 0);
    for (int l = 0; l < 4; ++l) {
        const LevelGeom g = level_geom(H, W, l);
        a.h[l] = l < L ? g.h : 0; a.w[l] = l < L ? g.w : 0; a.off[l] = pl.off[l];
    }
    a.S0 = (int)g0.slab; a.PW = g0.pw; a.rows = pl.rows;
    a.B = B; a.N = H * W; a.L = L; a.r = r; a.mode = mode; a.d0 = d0;
    const float sq = sqrtf((float)D);
    int e; const float mant = frexpf(sq, &e);
    a.exact = (mant == 0.5f) && (sq * sq == (float)D);
    a.scale = 1.0f / sq; a.sqrt_d = sq;

    const size_t BN = (size_t)B * a.N;
    if (BN >= ((size_t)1 << 31)) return fail(RAFTCORR_EUNSUPPORTED, "build_bwd_deferred: B * H * W = %zu too large", BN);
    // LDS of a launch: the level rows + the K * C record floats it stages (the plan's figure is the capacity)
    const int kmax = std::min(pl.kc, K);
    const size_t lds = deferred_lds(pl.rows, kmax, L, r);
    const int nst = (kmax * pl.C + 255) / 256;               // staging registers per thread: 8, 16 or 24
    typedef void (*Kern)(DeferredArgs);
    Kern kern;
    if (r == 4)      kern = nst <= 8 ? corr_splat_fold_kernel<4, 8> : nst <= 16 ? corr_splat_fold_kernel<4, 16> : corr_splat_fold_kernel<4, 24>;
    else if (r == 3) kern = nst <= 8 ? corr_splat_fold_kernel<3, 8> : nst <= 16 ? corr_splat_fold_kernel<3, 16> : corr_splat_fold_kernel<3, 24>;
    else             kern = corr_splat_fold_kernel<-1, 24>;
    if (lds > 64 * 1024) {                                   // > 64 KiB of dynamic LDS: opt in (cheap, once per call)
        const hipError_t err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, DEF_MAX_LDS);
        if (err != hipSuccess) return fail(RAFTCORR_ELAUNCH, "build_bwd_deferred: %s", hipGetErrorString(err));
    }
    unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / (lds + 512)));
#if defined(RC_LAB)                    // tools/ only
    if (const char* e = getenv("LAB_PER_CU")) per_cu = (unsigned)atoi(e);
#endif
    const unsigned grid = (unsigned)std::min<size_t>(std::min<size_t>(BN, 256u * per_cu), DEF_MAX_GRID);
    if (n_wg) *n_wg = grid;
    for (int k0 = 0; k0 < K; k0 += pl.kc) {
        a.K = std::min(pl.kc, K - k0);
        for (int k = 0; k < a.K; ++k) { a.rec[k] = records[k0 + k]; a.coords[k] = coords[k0 + k]; }
        a.accumulate = (accumulate || k0 > 0) ? 1 : 0;
        a.wgmax = (k0 + pl.kc >= K) ? wgmax : nullptr;       // the maxima of the final values
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
        const int rc = check_launch("corr_splat_fold_kernel");
        if (rc) return rc;
    }
    return RAFTCORR_OK;
}

}  // namespace raftcorr
