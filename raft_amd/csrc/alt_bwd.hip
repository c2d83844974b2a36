// On-the-fly ("alternate") correlation, backward w.r.t. the features, all levels: no atomics, no memsets,
// bit-reproducible.  Gradient of AlternateCorrBlock.__call__ (core/corr.py:75-92); per level the arithmetic of
// alt_cuda_corr's corr_backward_kernel (correlation_kernel.cu:123-256: g = bilinear-weighted sum of <= 4 corr_grad
// cells per tap, fmap1_grad += g * fmap2, fmap2_grad += g * fmap1), which the reference writes with one atomicAdd per
// (query, tap, channel) (:229-238) and the round-1 kernel with one per (tile, box target, channel).
//
// Both gradients are sums over (query, tap) pairs of g[q][u] times the OTHER map's feature vector:
//     d(fmap1)[q][c]   = sum_l sum_u g_l[q][u] fmap2_l[u][c]          d(fmap2_l)[u][c] = sum_q g_l[q][u] fmap1[q][c]
// and each of them has a natural OWNER: a workgroup that owns an 8x8 tile of queries walks the 8x8 target tiles its
// windows touch, one that owns an 8x8 tile of level-l targets walks the query tiles whose windows touch it.  Either
// way one STEP is the same dense 64 x 64 block: build G[q][u] from the 2x2-filtered d(out) window of every query
// (zero where u is not a tap of q), multiply it with the partner tile's features on the 16-bit matrix cores and keep
// the owner's 64 x 256 slice of the gradient in registers until it is stored -- once, by its only writer.
//
//  * Operands: the features are packed once per frame pair TILE-MAJOR (raftcorr_alt_pack_tiles): per 8x8 pixel tile
//    and channel 256 bytes = [hi | lo] fp16 pieces of the 64 pixels (the fp16x2 scheme of alt_fwd16.hip: maps scaled
//    by a power of two into the top of fp16's range, x = hi + lo, 22 significand bits).  The contraction index of both
//    products is a PIXEL index, so an MFMA operand fragment (8 consecutive k) is one row of 8 pixels of one channel =
//    one 16-byte slot, and a partner tile's 128-channel slice is 32 KiB of contiguous memory that goes to LDS by
//    direct-to-LDS DMA.  G is scaled by a power of two taken from max|d(out)| (one small range pass) and split the
//    same way; the product is hi*hi + hi*lo + lo*hi in fp32 accumulators, unscaled exactly at the end.
//  * LDS images have 256-byte rows (row = query / target / channel, 16 slots); slot j of row r lives at j ^ (r & 15),
//    which makes the ds_read_b128 fragment reads conflict-free (the DMA applies the XOR to its source address).
//  * d(fmap1): workgroup = (query tile, 256-channel group), all levels -> stored directly.  d(fmap2_l): workgroup =
//    (target tile, channel group, split s of the list of overlapping query tiles); level 0 stores directly, the
//    coarser levels (few target tiles, each touched by hundreds of query tiles) write per-split partial tiles that a
//    small second kernel adds in a fixed order.  The list comes from a table of per-(level, query tile) window
//    bounding boxes (alt_bwd_pre_kernel) and is walked in increasing tile order: the summation order never changes.
#include "common.h"
#include "alt_common.h"
#include <algorithm>

namespace raftcorr {

namespace {

constexpr int NCB = 4;                            // MFMA column blocks (32 channels each) per staged partner slice
constexpr int CG = 32 * NCB;                      // channels per slice
constexpr int NH = 2;                             // slices per workgroup: G is built once per step and multiplied with each
constexpr int GW = NH * CG;                       // channels per workgroup
constexpr int IMG_A = TQ * 16;                    // uint4 slots of the G image  [64 rows][16 slots]
constexpr int IMG_B = CG * 16;                    // uint4 slots of the partner image [CG channel rows][16 slots]
constexpr int MAXSPLIT = 32;
constexpr int DMAXP = 512;                        // partial maxima of |d(out)| (alt_bwd_pre_kernel)
constexpr int ALT_KMAX = 16;                      // lookups per call (kernel-argument arrays)

struct AltBwdArgs {
    const uint4* f1t;                             // tile-major fmap1: [B][qtiles][Cp][16 slots]
    const uint4* f2t[RAFTCORR_MAX_LEVELS];        // tile-major pooled fmap2, level l: [B][ttiles_l][Cp][16]
    int hl[RAFTCORR_MAX_LEVELS], wl[RAFTCORR_MAX_LEVELS];
    // One call serves nk lookups of the same frame pair(s) at once (the deferred backward pass): the kernels run over a
    // VIRTUAL batch B = nk * Bf, item b = lookup b / Bf of real batch item b % Bf.  Per-lookup inputs come as arrays,
    // the features are indexed by the real item, everything in the workspace and the outputs by the virtual one.
    const float* coords_k[ALT_KMAX];              // per lookup: (Bf, 2, H, W)
    const float* dout_k[ALT_KMAX];                // per lookup: (Bf, L*rd*rd, H, W)
    int nk, Bf;
    const float* maxpart;                         // [2][128] partial maxima of |fmap1|, |fmap2| (pack scales)
    const float* dmax;                            // [nk][ndmax] partial maxima of |dout| per lookup
    int ndmax;                                    // a multiple of 4; pixel tiles: DMAXP (a range pass of its own), window mode: one
                                                  // per (item, level, group slot), written by the workgroup that gathers the d(out) tile
    float* df1;                                   // (B, H, W, C)
    float* df2[RAFTCORR_MAX_LEVELS];              // (B, H_l, W_l, C)
    int4* boxes;                                  // [B][L][qtiles]: (xmin, xmax, ymin, ymax) of the tile's windows at level l
    int4* wins;                                   // [B][L][qtiles][64]: (x0 - r, y0 - r, bits of fx, bits of fy) per query
    float* partial;                               // d(fmap2) partial tiles of the split levels
    long long poff[RAFTCORR_MAX_LEVELS];          // level l's offset in `partial` (floats)
    int nsplit[RAFTCORR_MAX_LEVELS];
    int seg[RAFTCORR_MAX_LEVELS + 1];             // workgroups [0, seg[0]): d(fmap1); [seg[i], seg[i+1]): d(fmap2_l), l = lorder[i];
                                                  // every segment is padded to a multiple of 8 (one share per XCD)
    int lorder[RAFTCORR_MAX_LEVELS];              // the levels in launch order (long workgroups first, the shortest last)
    int B, C, Cp, H, W, L, r;
    int qtx, qtiles, ngroups;
    int gw;                                       // channels per workgroup: CG or GW (see make_plan)
    int accumulate;                               // != 0: df1 / df2 += (same single writer per element), else overwritten
    float sqrt_c;
    // window mode (qlist != nullptr): a "query tile" is a GROUP of 64 queries whose level-0 windows start in the same
    // 8x8 cell (alt_wfill_kernel: groups in bucket order, members in query order -- a function of the coordinates
    // alone); qtiles = group slots per batch item, ngr[b] of them in use.  f1t then is the tile-major copy of fmap1
    // gathered for the groups, dg the groups' d(out) tiles [B][L][qtiles][rd*rd][64].
    bool window;
    const int* qlist_k[ALT_KMAX];                 // per lookup: [Bf][qtiles][64] query pixel indices, -1 = empty slot
    const int* ngr_k[ALT_KMAX];                   // per lookup: [Bf]
    float* dg;
};

struct BwdSh {
    int wx0[TQ], wy0[TQ];                         // window origins of the step's 64 queries
    float fx[TQ], fy[TQ];
    int list[256];
    int wc[4];
};

// exponent that maps max|d(out)| into [2^14, 2^15): every wave reduces the DMAXP partial maxima itself
__device__ __forceinline__ int dout_exp(const AltBwdArgs& p, int lane, int look)
{
    float m = 0.f;
    const float4* __restrict__ d = (const float4*)(p.dmax + (size_t)look * p.ndmax);
    for (int i = lane; i < (p.ndmax >> 2); i += 64) {
        const float4 v = d[i];
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    return range_exp16(wave_max128(m));
}

// virtual batch item b -> its lookup's inputs
__device__ __forceinline__ const float* coords_of(const AltBwdArgs& p, int b)
{
    const int look = b / p.Bf;
    return p.coords_k[look] + (size_t)(b - look * p.Bf) * 2 * p.H * p.W;
}
__device__ __forceinline__ const float* dout_of(const AltBwdArgs& p, int b)       // (L*rd*rd, H, W) of the item
{
    const int look = b / p.Bf, rd = 2 * p.r + 1;
    return p.dout_k[look] + (size_t)(b - look * p.Bf) * p.L * rd * rd * p.H * p.W;
}
__device__ __forceinline__ const int* qlist_of(const AltBwdArgs& p, int b)        // [qtiles][64] of the item
{
    const int look = b / p.Bf;
    return p.qlist_k[look] + (size_t)(b - look * p.Bf) * p.qtiles * TQ;
}
__device__ __forceinline__ int ngr_of(const AltBwdArgs& p, int b)
{
    const int look = b / p.Bf;
    return p.ngr_k[look][b - look * p.Bf];
}

__device__ __forceinline__ int ttx_of(const AltBwdArgs& p, int l) { return (p.wl[l] + 7) >> 3; }
__device__ __forceinline__ int ttiles_of(const AltBwdArgs& p, int l) { return ttx_of(p, l) * ((p.hl[l] + 7) >> 3); }

// ---------------------------------------------------------------------------------------------
// tile-major pack: (B, C, Hm, Wm) fp32 NCHW -> [B][tile][Cp][hi 8 rows | lo 8 rows][8 px] fp16, scaled, zero padded.
// grid (sum over the maps of their tiles, Cp / 32, B), block 256: thread = (row = tid & 7, channel = tid >> 3)
__global__ __launch_bounds__(256)
void alt_pack_tiles_kernel(PackSet ps)
{
    const int m = packset_map(ps, blockIdx.x);
    const float* __restrict__ f = ps.src[m];
    uint4* __restrict__ packed = ps.dst[m];
    const float* __restrict__ maxpart = ps.maxpart[m];
    const int C = ps.C, Cp = ps.Cp, Hm = ps.h[m], Wm = ps.w[m];
    const float scale = ldexpf(1.0f, range_exp16(wave_max128(fmaxf(maxpart[threadIdx.x & 63], maxpart[(threadIdx.x & 63) + 64]))));
    const int tx_n = (Wm + 7) >> 3;
    const int tile = blockIdx.x - ps.blk0[m], ntiles = ps.blk0[m + 1] - ps.blk0[m];
    const int row = threadIdx.x & 7, c = blockIdx.y * 32 + (threadIdx.x >> 3), b = blockIdx.z;
    const int y = (tile / tx_n) * 8 + row, x0 = (tile % tx_n) * 8;
    const bool live = c < C && y < Hm;
    const float* src = f + ((size_t)b * C + min(c, C - 1)) * Hm * Wm + (size_t)min(y, Hm - 1) * Wm;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = src[min(x0 + j, Wm - 1)];                 // all 8 loads in flight
    unsigned short h[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = (live && x0 + j < Wm) ? x[j] * scale : 0.f;
        const _Float16 hi = (_Float16)v;                     // RNE
        const _Float16 lo = (_Float16)(v - (float)hi);       // exact remainder, rounded once
        h[0][j] = __builtin_bit_cast(unsigned short, hi);
        h[1][j] = __builtin_bit_cast(unsigned short, lo);
    }
    uint4* dst = packed + (((size_t)b * ntiles + tile) * Cp + c) * 16 + row;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        uint4 w;
        w.x = h[pc][0] | ((unsigned)h[pc][1] << 16); w.y = h[pc][2] | ((unsigned)h[pc][3] << 16);
        w.z = h[pc][4] | ((unsigned)h[pc][5] << 16); w.w = h[pc][6] | ((unsigned)h[pc][7] << 16);
        dst[pc * 8] = w;
    }
}

// ---------------------------------------------------------------------------------------------
// Window mode: query groups by window position, deterministic.  The queries are dealt into buckets by the 8x8 cell of
// level-0 targets that holds their window centre (one ring of cells around the map takes the centres outside it);
// the buckets of a row of cells are laid end to end, queries IN QUERY ORDER inside a bucket, and every 64 slots form a
// group (a row starts a new group).  Three small
// launches: count (one wave per bucket scans the coordinates), scan (groups per bucket -> first group), fill (the same
// scan as the count, ordered compaction by ballot + popcount).  No atomics: the lists depend on the coordinates only,
// so the backward pass stays bit-reproducible.  Under smooth flow a bucket is a pixel tile; under per-pixel noise the
// box of a group stays (8 + 2r + 2)^2 at level 0 where a pixel tile's box grows with the noise.
struct WSortArgs {
    const float* coords;                          // (B, 2, H, W)
    int* cnt;                                     // [B][nb][nch] queries of chunk ch in the bucket; after the scan: the
                                                  // bucket's queries in the chunks before ch
    int* toff;                                    // [B][nb] first slot of the bucket in qlist (rows of cells end to end)
    int* ngr;                                     // [B] groups in use
    int* qlist;                                   // [B][gmax][64]
    int N, nbx, nby, nb, nch, gmax;
};

constexpr int WCHUNK = 1024;                      // queries per (bucket, chunk) wave

__device__ __forceinline__ int wbucket_key(const WSortArgs& p, float x, float y)
{
    int x0, y0;
    float f;
    split_coord(x, x0, f);
    split_coord(y, y0, f);
    return (min(max(y0 >> 3, -1), p.nby - 2) + 1) * p.nbx + min(max(x0 >> 3, -1), p.nbx - 2) + 1;
}

// One wave per (bucket, chunk of WCHUNK consecutive queries).  FILL = false: cnt[b][bucket][chunk]; FILL = true: the
// chunk's queries of the bucket -> qlist behind those of the earlier chunks (ordered compaction by ballot + popcount).
// The buckets of one ROW of cells are laid end to end (toff = first SLOT of the bucket), so a group of 64 may hold the
// tail of one cell and the head of its right neighbour: under per-pixel noise a cell holds 30..40 queries, and groups
// that stop at cell borders would be half empty (KITTI random walks: 230..250 groups for 115 full ones).  A row's
// last group is padded with -1 by the row's last bucket.  grid (nb, nch, B), block 64
template <bool FILL>
__global__ __launch_bounds__(64)
void alt_wbucket_kernel(WSortArgs p)
{
    const int bucket = blockIdx.x, ch = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const float* __restrict__ cx = p.coords + (size_t)b * 2 * p.N;
    const int slot = (b * p.nb + bucket) * p.nch + ch;
    int n = FILL ? p.cnt[slot] : 0;               // wave-uniform: position inside the bucket / count
    int* __restrict__ list = FILL ? p.qlist + (size_t)b * p.gmax * 64 + p.toff[b * p.nb + bucket] : nullptr;
    const int qa = ch * WCHUNK;
    float x[WCHUNK / 64], y[WCHUNK / 64];
#pragma unroll
    for (int u = 0; u < WCHUNK / 64; ++u) {       // all loads of the chunk in flight
        const int q = min(qa + u * 64 + lane, p.N - 1);
        x[u] = cx[q]; y[u] = cx[p.N + q];
    }
#pragma unroll
    for (int u = 0; u < WCHUNK / 64; ++u) {
        const int q = qa + u * 64 + lane;
        const bool mine = q < p.N && wbucket_key(p, x[u], y[u]) == bucket;
        const unsigned long long m = __ballot(mine);
        if (FILL && mine) list[n + __popcll(m & ((1ull << lane) - 1))] = q;
        n += __popcll(m);
    }
    if (!FILL) { if (lane == 0) p.cnt[slot] = n; return; }
    if (ch == p.nch - 1 && bucket % p.nbx == p.nbx - 1) {            // end of the row: pad its last group
        const int end = p.toff[b * p.nb + bucket] + n;               // (n = the whole bucket now)
        if ((end & 63) && lane >= (end & 63)) p.qlist[(size_t)b * p.gmax * 64 + (end & ~63) + lane] = -1;
    }
}

// queries per bucket -> first slot of every bucket (rows of cells end to end, every row starting a new group), groups
// in use; cnt becomes the exclusive prefix over a bucket's chunks.  grid (B), block 1024
__global__ __launch_bounds__(1024)
void alt_wscan_kernel(WSortArgs p)
{
    __shared__ int rowbase[1024];                 // first group of a row of cells (nby <= 1024: checked by the launcher)
    const int b = blockIdx.x, tid = threadIdx.x;
    int* __restrict__ cnt = p.cnt + (size_t)b * p.nb * p.nch;
    int* __restrict__ off = p.toff + b * p.nb;
    for (int i = tid; i < p.nb; i += 1024) {
        int tot = 0;
        for (int c = 0; c < p.nch; ++c) { const int v = cnt[i * p.nch + c]; cnt[i * p.nch + c] = tot; tot += v; }
        off[i] = tot;                                        // queries of the bucket, replaced by its first slot below
    }
    __syncthreads();
    if (tid < p.nby) {                                       // one thread per row of cells: prefix inside the row
        int run = 0;
        for (int i = tid * p.nbx; i < (tid + 1) * p.nbx; ++i) { const int v = off[i]; off[i] = run; run += v; }
        rowbase[tid] = (run + 63) >> 6;                      // groups of the row
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int r = 0; r < p.nby; ++r) { const int g = rowbase[r]; rowbase[r] = run; run += g; }
        p.ngr[b] = run;
    }
    __syncthreads();
    for (int i = tid; i < p.nb; i += 1024) off[i] += rowbase[i / p.nbx] * 64;
}

// The same groups in ONE launch for maps of up to WS_MAXQ queries (every BASELINE shape but 1080p): one workgroup per
// batch item keeps the whole count table in LDS -- cnt[chunk of 64 queries][bucket], 16 bits each -- so the three
// passes above become three phases between barriers and two of the three launch latencies (5 us each at this size)
// go away.  A wave owns WS_MAXC consecutive chunks; inside a chunk the queries of one bucket find each other through a
// lane mask per bucket, which gives the per-chunk counts and each query's rank among its chunk mates of that bucket.
// The wave walks its chunks in order and keeps a running count per bucket (its own row of `run`), so what it leaves in
// cnt is already the exclusive prefix over ITS chunks; the prefix over all chunks of a bucket is then a scan over the 16
// waves' totals (16 table reads per bucket; the first version scanned the 115 chunks of a bucket with one thread:
// 3.7 of the launch's 11.5 us at KITTI size, and had to clear the whole count table first).  The rows of cells laid
// end to end and the row-end padding are those of alt_wscan_kernel / alt_wbucket_kernel<true>: the lists are the same,
// bit for bit.
// grid (B), block 1024, dynamic LDS = ws_small_lds(nch64, nb, nby)
constexpr int WS_WAVES = 16, WS_MAXC = 8, WS_MAXQ = WS_WAVES * WS_MAXC * 64;

__host__ __device__ inline size_t ws_small_lds(int nch64, int nb, int nby)
{
    return (((size_t)nch64 * nb * 2 + 15) & ~(size_t)15) + (8 + sizeof(int)) * (size_t)WS_WAVES * nb +
           sizeof(int) * (2 * (size_t)nb + 2 * (size_t)nby);
}

// exclusive prefix of v over the lanes of a wave; total = the sum over all 64 lanes
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total)
{
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    total = __shfl(inc, 63);
    return inc - v;
}

__global__ __launch_bounds__(1024)
void alt_wgroups_small_kernel(WSortArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch64 = (p.N + 63) >> 6, nb = p.nb;
    unsigned short* __restrict__ cnt = (unsigned short*)wsm;                               // [nch64][nb], written where read
    unsigned long long* masks = (unsigned long long*)(wsm + (((size_t)nch64 * nb * 2 + 15) & ~(size_t)15));   // [16 waves][nb]
    int* __restrict__ run = (int*)(masks + (size_t)WS_WAVES * nb);                         // [16 waves][nb] running counts
    int* __restrict__ tot = run + (size_t)WS_WAVES * nb;                                   // [nb] queries of the bucket
    int* __restrict__ toffs = tot + nb;                                                    // [nb] first slot of the bucket
    int* __restrict__ rowg = toffs + nb;                                                   // [nby] first group of the row
    int* __restrict__ rowlen = rowg + p.nby;                                               // [nby] queries of the row
    const float* __restrict__ cx = p.coords + (size_t)b * 2 * p.N;

    float x[WS_MAXC], y[WS_MAXC];
#pragma unroll
    for (int u = 0; u < WS_MAXC; ++u) {                       // all loads of the wave in flight
        const int q = min((wave * WS_MAXC + u) * 64 + lane, p.N - 1);
        x[u] = cx[q]; y[u] = cx[p.N + q];
    }
    for (int i = tid; i < WS_WAVES * nb; i += 1024) { masks[i] = 0ull; run[i] = 0; }
    __syncthreads();

    // phase 1: per-chunk counts and ranks.  The lanes of a chunk that share a bucket find each other through a 64-bit
    // lane mask per bucket (this wave's own table, LDS atomic OR: the result does not depend on the order of the lanes),
    // rank = set bits below the lane, count = all set bits; every lane clears the entry it used.
    int key[WS_MAXC], rank[WS_MAXC];
    unsigned long long* __restrict__ mask = masks + (size_t)wave * nb;
    int* __restrict__ myrun = run + (size_t)wave * nb;
#pragma unroll
    for (int u = 0; u < WS_MAXC; ++u) {
        const int chunk = wave * WS_MAXC + u, q = chunk * 64 + lane;
        const bool valid = q < p.N;
        key[u] = valid ? wbucket_key(p, x[u], y[u]) : 0;
        rank[u] = 0;
        if (valid) atomicOr(&mask[key[u]], 1ull << lane);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const unsigned long long m = mask[key[u]];
            rank[u] = __popcll(m & ((1ull << lane) - 1));
            if (rank[u] == 0) {                                  // one lane per bucket of the chunk
                const int before = myrun[key[u]];                // the bucket's queries in this wave's earlier chunks
                cnt[chunk * nb + key[u]] = (unsigned short)before;
                myrun[key[u]] = before + __popcll(m);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (valid) mask[key[u]] = 0ull;
    }
    __syncthreads();

    // phase 2: exclusive prefix over the waves' totals of a bucket (all 16 reads in flight) ...
    for (int k = tid; k < nb; k += 1024) {
        int v[WS_WAVES];
#pragma unroll
        for (int w = 0; w < WS_WAVES; ++w) v[w] = run[w * nb + k];
        int acc = 0;
#pragma unroll
        for (int w = 0; w < WS_WAVES; ++w) { run[w * nb + k] = acc; acc += v[w]; }
        tot[k] = acc;
    }
    __syncthreads();
    // ... the buckets of a row of cells end to end (one wave per row) ...
    for (int r = wave; r < p.nby; r += WS_WAVES) {
        int carry = 0;
        for (int i0 = 0; i0 < p.nbx; i0 += 64) {
            const int i = i0 + lane, v = i < p.nbx ? tot[r * p.nbx + i] : 0;
            int sum;
            const int ex = wave_excl_scan(v, lane, sum);
            if (i < p.nbx) toffs[r * p.nbx + i] = carry + ex;
            carry += sum;
        }
        if (lane == 0) rowlen[r] = carry;
    }
    __syncthreads();
    // ... and every row starting a new group
    if (wave == 0) {
        int carry = 0;
        for (int r0 = 0; r0 < p.nby; r0 += 64) {
            const int r = r0 + lane, g = r < p.nby ? (rowlen[r] + 63) >> 6 : 0;
            int sum;
            const int ex = wave_excl_scan(g, lane, sum);
            if (r < p.nby) rowg[r] = carry + ex;
            carry += sum;
        }
        if (lane == 0) p.ngr[b] = carry;
    }
    __syncthreads();

    // phase 3: every query to its slot; the last group of a row padded with -1
    int* __restrict__ ql = p.qlist + (size_t)b * p.gmax * 64;
#pragma unroll
    for (int u = 0; u < WS_MAXC; ++u) {
        const int chunk = wave * WS_MAXC + u, q = chunk * 64 + lane;
        if (q < p.N) {
            const int k = key[u];
            ql[rowg[k / p.nbx] * 64 + toffs[k] + myrun[k] + cnt[chunk * nb + k] + rank[u]] = q;
        }
    }
    for (int r = wave; r < p.nby; r += WS_WAVES) {
        const int end = rowg[r] * 64 + rowlen[r];
        if ((end & 63) && lane >= (end & 63)) ql[(end & ~63) + lane] = -1;
    }
}

// tile-major copy of fmap1 for the groups (the partner images of the d(fmap2) role), gathered from the forward
// pass' packed copy [pixel][Cp/32][hi: 4 slots of 8 channels | lo: 4 slots] (alt_fwd16.hip; same scaling).
// One workgroup of 256 threads per (group g, 32-channel chunk, batch item); lines: TQ * 8 slots of LDS
// f1p: the packed map of the REAL batch item; ql: the group's 64 query indices; f1g: the group's image (Cp rows)
__device__ __forceinline__ void wgather_f1(const uint4* __restrict__ f1p, uint4* __restrict__ f1g, const int* __restrict__ ql,
                                           int Cp, int chunk, uint4* lines)
{
    const int tid = threadIdx.x, nkc = Cp / 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i, pix = ql[e >> 3];
        const uint4 v = f1p[((size_t)max(pix, 0) * nkc + chunk) * 8 + (e & 7)];
        lines[e] = pix >= 0 ? v : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const unsigned short* h = (const unsigned short*)lines;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int o = tid + 256 * i, c = o >> 4, sl = o & 15, piece = sl >> 3, q0 = (sl & 7) * 8;
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = h[((q0 + j) * 8 + piece * 4 + (c >> 3)) * 8 + (c & 7)];
        uint4 w;
        w.x = v[0] | ((unsigned)v[1] << 16); w.y = v[2] | ((unsigned)v[3] << 16);
        w.z = v[4] | ((unsigned)v[5] << 16); w.w = v[6] | ((unsigned)v[7] << 16);
        f1g[(size_t)(chunk * 32 + c) * 16 + sl] = w;
    }
}

// ---------------------------------------------------------------------------------------------
// The small passes in front of the main kernel, in one launch (they do not depend on each other):
//   workgroups [0, nwin = B * L * qtiles): per (batch, level, query tile) the 64 windows (origin + fractions), their
//                        bounding box and, in window mode, the group's d(out) tile;
//   the nk * DMAXP workgroups after them: partial maxima of |d(out)| per lookup (n floats each), the range of the G images;
//   window mode, ngather more: the groups' tile-major copy of fmap1 (wgather_f1), one workgroup per (group, 32-channel
//                        chunk, batch item).
// block 64 (pixel tiles) or 256 (window mode)
__global__ __launch_bounds__(256)
void alt_bwd_pre_kernel(AltBwdArgs p, int nwin, size_t n, const uint4* __restrict__ f1p)
{
    __shared__ TileQ sh;
    __shared__ uint4 lines[TQ * 8];               // wgather_f1: the queries' 128-byte lines of one channel chunk
    __shared__ float red[4];
    const int nwinp = (nwin + 7) & ~7;                       // the window section of the launch, padded
    const int nmaxwg = p.window ? 0 : p.nk * DMAXP;          // (window mode: the gathering workgroups take the maxima)
    if ((int)blockIdx.x >= nwinp + nmaxwg) {
        const int idx = blockIdx.x - nwinp - nmaxwg, nkc = p.Cp / 32;
        const int g = idx % p.qtiles, chunk = (idx / p.qtiles) % nkc, b = idx / (p.qtiles * nkc);
        if (g >= ngr_of(p, b)) return;                       // (whole workgroup)
        wgather_f1(f1p + (size_t)(b % p.Bf) * p.H * p.W * nkc * 8, const_cast<uint4*>(p.f1t) + ((size_t)b * p.qtiles + g) * p.Cp * 16,
                   qlist_of(p, b) + (size_t)g * TQ, p.Cp, chunk, lines);
        return;
    }
    if ((int)blockIdx.x >= nwinp) {
        const int j = (blockIdx.x - nwinp) % DMAXP, look = (blockIdx.x - nwinp) / DMAXP;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        const float* __restrict__ x = p.dout_k[look];
        float m = 0.f;
        const size_t n4 = (((uintptr_t)x & 15) == 0) ? n >> 2 : 0;       // 16-byte loads when aligned
        const size_t STRIDE = (size_t)DMAXP * nw * 64;
        size_t i = ((size_t)j * nw + w) * 64 + lane;
        for (; i + 3 * STRIDE < n4; i += 4 * STRIDE) {                   // four loads in flight per thread
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ((const float4*)x)[i + u * STRIDE];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
        }
        for (; i < n4; i += STRIDE) {
            const float4 v = ((const float4*)x)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (size_t t = n4 * 4 + ((size_t)j * nw + w) * 64 + lane; t < n; t += STRIDE) m = fmaxf(m, fabsf(x[t]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[w] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int u = 1; u < nw; ++u) m = fmaxf(m, red[u]);
            const_cast<float*>(p.dmax)[look * p.ndmax + j] = m;
        }
        return;
    }
    // XCD-aware: the hardware deals workgroups round-robin over the 8 XCDs; XCD x takes the x-th eighth of the item list,
    // so that neighbouring groups -- whose d(out) rows share their 128-byte lines -- go through the same L2 (measured on
    // the 12-lookup call at KITTI size: 1.66 GB of L2 fills for 114 MB of d(out) with the plain order)
    const int nwin8 = (nwin + 7) >> 3, item = (blockIdx.x & 7) * nwin8 + (blockIdx.x >> 3);
    if (item >= nwin) return;                                // (whole workgroup; the launch pads nwin to a multiple of 8)
    const int tile = item % p.qtiles, bl = item / p.qtiles;
    const int q = threadIdx.x;
    Tile16 t;
    t.l = bl % p.L; t.b = bl / p.L;
    if (p.window) {                                          // window mode: the tile is a group of the list
        // the range pass rides along: this workgroup reads its part of d(out) anyway, its maximum is one of the lookup's
        // partial maxima (slot = the workgroup's place among the lookup's; the last one zeroes the padding up to ndmax)
        const int nper = p.Bf * p.L * p.qtiles, look = t.b / p.Bf, slot = item - look * nper;
        float* __restrict__ dmx = const_cast<float*>(p.dmax) + (size_t)look * p.ndmax;
        if (slot == nper - 1 && q >= 1 && q <= p.ndmax - nper) dmx[nper - 1 + q] = 0.f;
        if (tile >= ngr_of(p, t.b)) {                        // slot not in use: an empty box keeps it out of every walk
            if (q == 0) { p.boxes[item] = make_int4(INT_MAX, INT_MIN, INT_MAX, INT_MIN); dmx[slot] = 0.f; }
            return;
        }
        t.ty0 = t.tx0 = 0;
        tile_level_setup(p, sh, t, coords_of(p, t.b), qlist_of(p, t.b) + (size_t)tile * TQ);
        // the group's d(out) tile [rd*rd][64], gathered once: the main kernel then fetches it with contiguous loads.
        // Four waves, each every fourth batch of 16 channels.
        const int rd2 = (2 * p.r + 1) * (2 * p.r + 1), qq = q & 63, pix = sh.qpix[qq];
        const size_t P = (size_t)p.H * p.W;
        const float* __restrict__ src = dout_of(p, t.b) + (size_t)t.l * rd2 * P + max(pix, 0);
        float* __restrict__ dst = p.dg + (size_t)item * rd2 * TQ + qq;
        float m = 0.f;
        for (int c0 = 16 * (q >> 6); c0 < rd2; c0 += 16 * (blockDim.x >> 6)) {     // 16 loads in flight
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = src[(size_t)min(c0 + j, rd2 - 1) * P];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (c0 + j < rd2) dst[(c0 + j) * TQ] = pix >= 0 ? v[j] : 0.f;
                if (pix >= 0) m = fmaxf(m, fabsf(v[j]));     // (rows beyond rd2 repeat the last one)
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((q & 63) == 0) red[q >> 6] = m;
        __syncthreads();
        if (q == 0) dmx[slot] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (q >= 64) return;
    } else {
        t.ty0 = (tile / p.qtx) * QH; t.tx0 = (tile % p.qtx) * QW;
        tile_level_setup(p, sh, t, coords_of(p, t.b));
    }
    p.wins[(size_t)item * TQ + q] = make_int4(sh.wx0[q], sh.wy0[q], __float_as_int(sh.fx[q]), __float_as_int(sh.fy[q]));
    if (q == 0) p.boxes[item] = make_int4(sh.box[0], sh.box[1], sh.box[2], sh.box[3]);
}

// ---------------------------------------------------------------------------------------------
// pieces of one step

// 8 scaled gradients -> two fp16 pieces -> slots `slot` (hi) and 8 + slot (lo) of row `row` of the G image, XOR key
__device__ __forceinline__ void store_g(uint4* img, int row, int slot, int key, const float (&g)[8])
{
    unsigned short h[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 hi = (_Float16)g[j];
        const _Float16 lo = (_Float16)(g[j] - (float)hi);
        h[0][j] = __builtin_bit_cast(unsigned short, hi);
        h[1][j] = __builtin_bit_cast(unsigned short, lo);
    }
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        uint4 w;
        w.x = h[pc][0] | ((unsigned)h[pc][1] << 16); w.y = h[pc][2] | ((unsigned)h[pc][3] << 16);
        w.z = h[pc][4] | ((unsigned)h[pc][5] << 16); w.w = h[pc][6] | ((unsigned)h[pc][7] << 16);
        img[row * 16 + ((pc * 8 + slot) ^ key)] = w;
    }
}

// partner image: `rows` channel rows (a multiple of 32) of 256 B, global -> LDS by DMA, 4 rows per wave-instruction;
// LDS slot j of row c receives source slot j ^ (c & 15)
__device__ __forceinline__ void stage_partner(const uint4* __restrict__ src, uint4* img, int rows, int wave, int lane)
{
    for (int r4 = wave; r4 * 4 < rows; r4 += 4) {
        const int row = r4 * 4 + (lane >> 4);
        glds16(src + row * 16 + ((lane & 15) ^ (row & 15)), img + r4 * 64);
    }
}

// d(out) window of the 8x8 query tile qt, level l, in two halves so that the loads can fly under other work:
// dtile_issue starts the loads of channels wave, wave + 4, .. (< 96) into v, dtile_commit writes them to LDS and
// fetches what a radius > 4 has beyond 96 channels.
struct DTile {
    float v[24];
    int4 win;
};

// where the d(out) window of query q of tile qt lives: channel ch at (base + ch * stride)[lane_off].  Pixel tiles read
// d(out) itself (B, L*rd*rd, H, W); window mode reads the group's gathered tile [rd*rd][64] (alt_bwd_pre_kernel).
__device__ __forceinline__ const float* dtile_source(const AltBwdArgs& p, int b, int l, int qt, int rd2, int q,
                                                     size_t& stride, unsigned& lane_off)
{
    if (p.window) {
        stride = TQ; lane_off = (unsigned)q;
        return p.dg + (((size_t)b * p.L + l) * p.qtiles + qt) * rd2 * TQ;
    }
    const size_t P = (size_t)p.H * p.W;
    const int gy = (qt / p.qtx) * QH + (q >> 3), gx = (qt % p.qtx) * QW + (q & 7);
    stride = P;
    lane_off = (gy < p.H && gx < p.W) ? (unsigned)(gy * p.W + gx) : 0u;               // out-of-map queries: any address
    return dout_of(p, b) + (size_t)l * rd2 * P;
}

__device__ __forceinline__ void dtile_issue(const AltBwdArgs& p, DTile& d, int b, int l, int qt, int rd2, int wave)
{
    const int q = threadIdx.x & 63;
    size_t stride;
    unsigned lane_off;
    const float* __restrict__ src = dtile_source(p, b, l, qt, rd2, q, stride, lane_off);   // wave-uniform base + lane offset
#pragma unroll
    for (int j = 0; j < 24; ++j) d.v[j] = (src + (size_t)min(wave + 4 * j, rd2 - 1) * stride)[lane_off];
    d.win = p.wins[(((size_t)b * p.L + l) * p.qtiles + qt) * TQ + q];
}

// D[ch][64]; room for max(rd2, 96) channels, so the 24 values go out unconditionally.  Queries outside the map keep
// whatever was loaded: their windows sit at -(1 << 24), every tap test fails and G is exactly zero.
__device__ __forceinline__ void dtile_commit(const AltBwdArgs& p, const DTile& d, float* D, BwdSh& sh, int b, int l, int qt,
                                             int rd2, int wave)
{
    const int q = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 24; ++j) D[(wave + 4 * j) * 64 + q] = d.v[j];
    // every wave writes the (same) windows: a wave that never read its d.win would carry the pending load into the next
    // step, where hipcc has to wait for ALL vector-memory traffic before it can reuse the registers
    sh.wx0[q] = d.win.x; sh.wy0[q] = d.win.y; sh.fx[q] = __int_as_float(d.win.z); sh.fy[q] = __int_as_float(d.win.w);
    if (rd2 > 96) {
        size_t stride;
        unsigned lane_off;
        const float* __restrict__ src = dtile_source(p, b, l, qt, rd2, q, stride, lane_off);
        for (int ch = 96 + wave; ch < rd2; ch += 4) D[ch * 64 + q] = (src + (size_t)ch * stride)[lane_off];
    }
}

// G image for an OWNED QUERY tile: row = query, k = target of the partner tile at (ty0, tx0) (level coordinates).
// thread = (query tid & 63, target rows 2 * wave, 2 * wave + 1); D[ch][64].
// TR: the image is read back TRANSPOSED (ds_read_b64_tr_b16, rows q .. q + 3 at a time): slot key 4 * (q & 3) keeps those
// four rows on different banks; row reads (ds_read_b128, 16 different rows at a time) want the key q & 15.
template <bool TR>
__device__ __forceinline__ void make_g_by_query(const BwdSh& sh, const float* D, uint4* Aimg, int ty0, int tx0, int rd, float sg)
{
    const int q = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wx0 = sh.wx0[q], wy0 = sh.wy0[q];
    const float fx = sh.fx[q], fy = sh.fy[q];
    const int txs = tx0 - wx0;                               // tap column of the tile's first target column
    const int uy = 2 * wave, ty = ty0 + uy - wy0;            // tap rows ty, ty + 1 <- d(out) window rows ty - 1, ty, ty + 1
    const bool okm = (unsigned)(ty - 1) < (unsigned)rd, ok0 = (unsigned)ty < (unsigned)rd, okp = (unsigned)(ty + 1) < (unsigned)rd;
    const int key = TR ? 4 * (q & 3) : (q & 15);
    const bool live = (okm || ok0 || okp) && txs + 7 >= 0 && txs - 1 < rd;
    float g0[8], g1[8];
    if (__ballot(live) == 0) {                               // wave-uniform: no query of the tile has a tap in these rows
#pragma unroll
        for (int i = 0; i < 8; ++i) g0[i] = g1[i] = 0.f;
    } else {
        // vertical blend weights with the row validity folded in (an invalid row is read at a clamped position, times 0)
        const float a0m = okm ? fy : 0.f, a00 = ok0 ? 1.f - fy : 0.f;            // tap row ty     = rows ty - 1, ty
        const float a10 = ok0 ? fy : 0.f, a1p = okp ? 1.f - fy : 0.f;            // tap row ty + 1 = rows ty, ty + 1
        const float* Dm = D + min(max(ty - 1, 0), rd - 1) * 64 + q;
        const float* D0 = D + min(max(ty, 0), rd - 1) * 64 + q;
        const float* Dp = D + min(max(ty + 1, 0), rd - 1) * 64 + q;
        const float cx1 = (1.f - fx) * sg, cx0 = fx * sg;
        float E0[9], E1[9];                                  // window column txs - 1 + i, blended vertically
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int a = txs - 1 + i;
            const bool okx = (unsigned)a < (unsigned)rd;
            const int base = min(max(a, 0), rd - 1) * rd * 64;
            const float dm = Dm[base], d0 = D0[base], dp = Dp[base];
            const float e0 = dm * a0m + d0 * a00, e1 = d0 * a10 + dp * a1p;
            E0[i] = okx ? e0 : 0.f;
            E1[i] = okx ? e1 : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { g0[i] = E0[i + 1] * cx1 + E0[i] * cx0; g1[i] = E1[i + 1] * cx1 + E1[i] * cx0; }
    }
    store_g(Aimg, q, uy, key, g0);
    store_g(Aimg, q, uy + 1, key, g1);
}

// acc[j] (rows 32 mb .. + 31 of the G image x channels 32 (nb0 + 2 j) .. + 31 of the partner image) += G * partner, K = 64.
// Unconditional on purpose: column blocks beyond the map's channels multiply whatever the image rows hold and are
// never stored (a column of the product depends on its own column of the partner only); a wave-uniform `if` around
// an MFMA makes hipcc copy whole accumulators between the two paths after every instruction.
// TR: the G image is [k][row] (stored by make_g_by_query<true>) and its fragments come through transposed reads: a
// 16-lane group fetches 4 image rows (k) x 16 columns and every lane receives one column; lane 4 qq + pp of a group
// supplies the address of row qq, columns 4 pp .. 4 pp + 3.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <bool TR>
__device__ __forceinline__ void mfma_step(const uint4* Aimg, const uint4* Bimg, v16f (&acc)[NCB / 2], int mb, int nb0,
                                          int l31, int kh)
{
    const int key = l31 & 15;
    const uint4* arow = Aimg + (mb * 32 + l31) * 16;
    // transposed reads: image row of this lane's address = 16 s + 8 kh + 4 r + qq; 16-byte slot = 4 mb + 2 gh + (pp >> 1)
    const int qq = (l31 & 15) >> 2, pp = l31 & 3, gh = l31 >> 4;
    const short* abase = (const short*)Aimg + (8 * kh + qq) * 128 + (pp & 1) * 4;
    const int slot_hi = ((4 * mb + 2 * gh + (pp >> 1)) ^ (4 * qq)) * 8, slot_lo = ((8 + 4 * mb + 2 * gh + (pp >> 1)) ^ (4 * qq)) * 8;
    auto tr = [](const short* a) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f16x8 ah, al;
        if (TR) {
            const short* a = abase + s * 16 * 128;
            const s16x4 h0 = tr(a + slot_hi), h1 = tr(a + 4 * 128 + slot_hi), l0 = tr(a + slot_lo), l1 = tr(a + 4 * 128 + slot_lo);
            ah = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
            al = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
        } else {
            ah = __builtin_bit_cast(f16x8, arow[(2 * s + kh) ^ key]);
            al = __builtin_bit_cast(f16x8, arow[(8 + 2 * s + kh) ^ key]);
        }
        f16x8 bh[NCB / 2], bl[NCB / 2];
#pragma unroll
        for (int j = 0; j < NCB / 2; ++j) {
            const uint4* brow = Bimg + ((nb0 + 2 * j) * 32 + l31) * 16;
            bh[j] = __builtin_bit_cast(f16x8, brow[(2 * s + kh) ^ key]);
            bl[j] = __builtin_bit_cast(f16x8, brow[(8 + 2 * s + kh) ^ key]);
        }
        // small terms first; the column blocks take turns inside each term, so that no MFMA waits for the one before it
#pragma unroll
        for (int j = 0; j < NCB / 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NCB / 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NCB / 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[j], 0, 0, 0);
    }
}

// s_barrier without the fence of __syncthreads() (which waits for every outstanding vector-memory operation once an
// LDS DMA is among them): LDS traffic of this wave done, then the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One step after the d(out) tile and the windows are in LDS: partner image by DMA, G image, product.  The LDS regions
// are restrict-qualified parameters: after inlining, the DMA destination and the reads / writes of the other regions
// carry distinct alias scopes, so hipcc does not wait for the DMA in front of the first LDS access of make_g (the
// construction of alt_fwd16.hip and build_pyramid.hip).  `prefetch()` may start global loads (no LDS access) that
// stay in flight over the step; keep25 = the 25 loads of dtile_issue (vector-memory loads complete in order: waiting
// until only those are outstanding means the older DMA has landed).  `rows` = channels of this workgroup (<= GW): the
// G image is built once and multiplied with one CG-channel slice of the partner after the other.
template <bool TR, class F>
__device__ __forceinline__ void pair_step(const uint4* __restrict__ src, int rows, uint4* __restrict__ Bimg,
                                          uint4* __restrict__ Aimg, const float* __restrict__ D, const BwdSh& sh,
                                          int ty0, int tx0, int rd, float sg, v16f (&acc)[NH][NCB / 2],
                                          int mb, int nb0, int l31, int kh, int wave, int lane, bool keep25, F prefetch)
{
    stage_partner(src, Bimg, min(rows, CG), wave, lane);
    asm volatile("" ::: "memory");                           // the DMA stays older than the prefetch
    prefetch();
#if !defined(RC_LAB_BWD_NOG)
    make_g_by_query<TR>(sh, D, Aimg, ty0, tx0, rd, sg);
#endif
    if (keep25) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
    else        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
#if !defined(RC_LAB_BWD_NOMFMA)
    mfma_step<TR>(Aimg, Bimg, acc[0], mb, nb0, l31, kh);
#endif
#pragma unroll
    for (int h = 1; h < NH; ++h) {
        if (rows <= h * CG) break;                           // workgroup-uniform
        lds_barrier();                                       // the slice in LDS has been multiplied
        stage_partner(src + h * CG * 16, Bimg, min(rows - h * CG, CG), wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the prefetch is older than this DMA and has had the step to land)
        lds_barrier();
#if !defined(RC_LAB_BWD_NOMFMA)
        mfma_step<TR>(Aimg, Bimg, acc[h], mb, nb0, l31, kh);
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// grid: seg[L] workgroups (see AltBwdArgs::seg), block 256 = 4 waves; wave w: row block w & 1, column blocks (w >> 1) + 2 j
__global__ __launch_bounds__(256, 2)
void alt_bwd_kernel(AltBwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    __shared__ BwdSh sh;
    uint4* Aimg = smem;
    uint4* Bimg = smem + IMG_A;
    float* D = (float*)(smem + IMG_A + IMG_B);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wave & 1, nb0 = wave >> 1;
    const int rd = 2 * p.r + 1, rd2 = rd * rd;
    const int e1 = range_exp16(wave_max128(fmaxf(p.maxpart[lane], p.maxpart[lane + 64])));
    const int e2 = range_exp16(wave_max128(fmaxf(p.maxpart[MAXPARTS16 + lane], p.maxpart[MAXPARTS16 + lane + 64])));
    const size_t P = (size_t)p.H * p.W;
    const int wg = blockIdx.x;

    v16f acc[NH][NCB / 2];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < NCB / 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[h][j][i] = 0.f;

    // Workgroup id -> work item, XCD-aware: the hardware deals workgroups round-robin over the 8 XCDs, each with its
    // own L2.  Inside a segment XCD x gets the x-th eighth of the item list, and the list is ordered so that an eighth
    // is a band of image rows (tiles row-major; for the split levels the split index, i.e. a band of query tiles,
    // comes first): the tiles a band shares -- partner images, d(out) tiles -- stay in that XCD's L2.
    auto banded = [](int idx, int len) { return (idx & 7) * (len >> 3) + (idx >> 3); };
    if (wg < p.seg[0]) {
        // ---------------- d(fmap1): owner = query tile, partners = target tiles of every level ----------------
        int g, tile, b;
        if (p.window) {
            // window mode: the segment holds pad8(qtiles * ngroups) ids per batch item, but only ngr[b] group slots are
            // in use: the bands are eighths of THOSE (groups are in bucket order = bands of image rows)
            const int perb = (p.qtiles * p.ngroups + 7) & ~7;
            b = wg / perb;
            const int wb = wg - b * perb, n_items = ngr_of(p, b) * p.ngroups, band = (n_items + 7) >> 3;
            const int item = (wb & 7) * band + (wb >> 3);
            if ((wb >> 3) >= band || item >= n_items) return;    // (whole workgroup)
            g = item % p.ngroups; tile = item / p.ngroups;
        } else {
            const int item = banded(wg, p.seg[0]);
            if (item >= p.B * p.qtiles * p.ngroups) return;  // padding of the segment (whole workgroup)
            g = item % p.ngroups;
            const int rest = item / p.ngroups;
            tile = rest % p.qtiles; b = rest / p.qtiles;
        }
        const int rows = min(p.gw, p.Cp - g * p.gw);             // channels of this workgroup (a multiple of 32)
        const int fb = b % p.Bf;                                 // the real batch item behind the virtual one
        const int eg = dout_exp(p, lane, b / p.Bf);
        const float sg = ldexpf(1.0f, eg);
        DTile dt;
        dtile_issue(p, dt, b, 0, tile, rd2, wave);
        for (int l = 0; l < p.L; ++l) {
            const int4 box = p.boxes[((size_t)b * p.L + l) * p.qtiles + tile];
            __syncthreads();                                 // the previous level's D and window table are no longer read
            dtile_commit(p, dt, D, sh, b, l, tile, rd2, wave);
            if (l + 1 < p.L) dtile_issue(p, dt, b, l + 1, tile, rd2, wave);        // flies under this level's steps
            __syncthreads();
            if (box.x > box.y || box.z > box.w) continue;
            const int ttx = ttx_of(p, l), ttiles = ttiles_of(p, l);
            const uint4* __restrict__ f2b = p.f2t[l] + ((size_t)fb * ttiles * p.Cp + g * p.gw) * 16;
            for (int ty = box.z >> 3; ty <= box.w >> 3; ++ty)
                for (int tx = box.x >> 3; tx <= box.y >> 3; ++tx) {
                    pair_step<false>(f2b + (size_t)(ty * ttx + tx) * p.Cp * 16, rows, Bimg, Aimg, D, sh, ty * 8, tx * 8, rd, sg,
                                     acc, mb, nb0, l31, kh, wave, lane, false, [] {});
                    lds_barrier();                           // images are free again
                }
        }
        const float unscale = ldexpf(1.0f, -(eg + e2));
        float* __restrict__ out = p.df1 + (size_t)b * P * p.C;
        int opix[16];                                        // pixel of the query behind accumulator register i, -1 = none
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            if (p.window) opix[i] = qlist_of(p, b)[(size_t)tile * TQ + q];
            else {
                const int gy = (tile / p.qtx) * QH + (q >> 3), gx = (tile % p.qtx) * QW + (q & 7);
                opix[i] = (gy < p.H && gx < p.W) ? gy * p.W + gx : -1;
            }
        }
#pragma unroll
        for (int hj = 0; hj < NH * (NCB / 2); ++hj) {
            const int h = hj / (NCB / 2), j = hj % (NCB / 2);
            const int c = g * p.gw + h * CG + (nb0 + 2 * j) * 32 + l31;
            if (h * CG < rows && c < p.C) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (opix[i] >= 0) {
                        float* o = out + (size_t)opix[i] * p.C + c;
                        float v = (acc[h][j][i] * unscale) / p.sqrt_c;
                        if (p.accumulate) v += *o;
                        *o = v;
                    }
                }
            }
        }
        return;
    }

    // ---------------- d(fmap2_l): owner = target tile, partners = query tiles whose windows touch it ----------------
    int si = 0;
    while (si + 1 < p.L && wg >= p.seg[si + 1]) ++si;
    const int l = p.lorder[si];
    const int nsplit = p.nsplit[l];
    const int ttx = ttx_of(p, l), ttiles = ttiles_of(p, l);
    int idx = banded(wg - p.seg[si], p.seg[si + 1] - p.seg[si]);
    if (idx >= p.B * ttiles * nsplit * p.ngroups) return;    // padding of the segment (whole workgroup)
    const int g = idx % p.ngroups; idx /= p.ngroups;
    int s, tt;
    if (ttiles >= 8 * nsplit) { s = idx % nsplit; idx /= nsplit; tt = idx % ttiles; idx /= ttiles; }    // bands of target tiles
    else                      { tt = idx % ttiles; idx /= ttiles; s = idx % nsplit; idx /= nsplit; }    // bands of query tiles
    const int b = idx;
    const int rows = min(p.gw, p.Cp - g * p.gw);                 // channels of this workgroup (a multiple of 32)
    const int ty0 = (tt / ttx) * 8, tx0 = (tt % ttx) * 8;
    const int eg = dout_exp(p, lane, b / p.Bf);
    const float sg = ldexpf(1.0f, eg);
    const int4* __restrict__ boxes = p.boxes + ((size_t)b * p.L + l) * p.qtiles;
    // partner images: window mode = the groups' gathered copies (one set per virtual item); pixel tiles = the packed map
    const uint4* __restrict__ f1b = p.f1t + ((size_t)(p.window ? b : b % p.Bf) * p.qtiles * p.Cp + g * p.gw) * 16;

    auto touches = [&](int e) {
        const int4 bx = boxes[min(e, p.qtiles - 1)];
        return e < p.qtiles && bx.x <= tx0 + 7 && bx.y >= tx0 && bx.z <= ty0 + 7 && bx.w >= ty0;
    };
    int n_touch = 0;
    for (int base = 0; base < p.qtiles; base += 256) n_touch += __syncthreads_count(touches(base + tid));
    const int lo = (int)((long long)s * n_touch / nsplit), hi = (int)((long long)(s + 1) * n_touch / nsplit);

    // The d(out) tile and the windows of partner i + 1 are fetched into registers while step i runs; the partners are
    // known one chunk of 256 table entries at a time, so the first step of a chunk waits for its own loads.
    DTile dt;
    int pos = 0;
    for (int base = 0; base < p.qtiles && pos < hi; base += 256) {
        const bool flag = touches(base + tid);
        const unsigned long long mask = __ballot(flag);
        if (lane == 0) sh.wc[wave] = __popcll(mask);
        __syncthreads();
        int before = 0, chunk_n = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int n = sh.wc[w]; before += w < wave ? n : 0; chunk_n += n; }
        if (flag) sh.list[before + __popcll(mask & ((1ull << lane) - 1))] = base + tid;
        __syncthreads();
        const int i0 = max(lo - pos, 0), i1 = min(hi - pos, chunk_n);
        if (i0 < i1) dtile_issue(p, dt, b, l, sh.list[i0], rd2, wave);
        for (int i = i0; i < i1; ++i) {
            const int qt = sh.list[i];
            const bool more = i + 1 < i1;
            const int qt_next = sh.list[more ? i + 1 : i];
            lds_barrier();                                   // the previous step's images, D and window table are free
            dtile_commit(p, dt, D, sh, b, l, qt, rd2, wave);
            lds_barrier();
            // the 25 loads of dtile_issue (24 channels + the windows) fly under this step and the start of the next
            pair_step<true>(f1b + (size_t)qt * p.Cp * 16, rows, Bimg, Aimg, D, sh, ty0, tx0, rd, sg,
                            acc, mb, nb0, l31, kh, wave, lane, more, [&] { if (more) dtile_issue(p, dt, b, l, qt_next, rd2, wave); });
        }
        pos += chunk_n;
        __syncthreads();                                     // list / counts are rewritten by the next round
    }

    const int Hl = p.hl[l], Wl = p.wl[l];
    if (nsplit == 1) {
        const float unscale = ldexpf(1.0f, -(eg + e1));
        float* __restrict__ out = p.df2[l] + (size_t)b * Hl * Wl * p.C;
#pragma unroll
        for (int hj = 0; hj < NH * (NCB / 2); ++hj) {
            const int h = hj / (NCB / 2), j = hj % (NCB / 2);
            const int c = g * p.gw + h * CG + (nb0 + 2 * j) * 32 + l31;
            if (h * CG < rows && c < p.C) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int u = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
                    const int Y = ty0 + (u >> 3), X = tx0 + (u & 7);
                    if (Y < Hl && X < Wl) {
                        float* o = out + ((size_t)Y * Wl + X) * p.C + c;
                        float v = (acc[h][j][i] * unscale) / p.sqrt_c;
                        if (p.accumulate) v += *o;
                        *o = v;
                    }
                }
            }
        }
    } else {
        float* __restrict__ out = p.partial + p.poff[l] + ((((size_t)b * ttiles + tt) * nsplit + s) * TQ) * p.Cp;
#pragma unroll
        for (int hj = 0; hj < NH * (NCB / 2); ++hj) {
            const int h = hj / (NCB / 2), j = hj % (NCB / 2);
            const int c = g * p.gw + h * CG + (nb0 + 2 * j) * 32 + l31;
            if (h * CG < rows && c < p.Cp) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int u = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
                    out[(size_t)u * p.Cp + c] = acc[h][j][i];
                }
            }
        }
    }
}

// d(fmap2_l) of the split levels: partial tiles added in split order.  grid (ceil(max_l B*H_l*W_l*Cp/4 / 256), L)
__global__ __launch_bounds__(256)
void alt_bwd_reduce_kernel(AltBwdArgs p)
{
    const int lane = threadIdx.x & 63, l = blockIdx.y;
    const int e1 = range_exp16(wave_max128(fmaxf(p.maxpart[lane], p.maxpart[lane + 64])));
    const int nsplit = p.nsplit[l];
    if (nsplit <= 1) return;
    const int Hl = p.hl[l], Wl = p.wl[l], HWl = Hl * Wl, c4n = p.Cp >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    // an item's elements are padded to a multiple of 64 in the index space: a wave stays inside one virtual item, i.e. one
    // lookup with its own range exponent (reduced with wave shuffles by all 64 lanes)
    const long long per = ((long long)HWl * c4n + 63) & ~63ll;
    const int b = (int)min(idx / per, (long long)p.B - 1);
    const long long rem = idx - (long long)b * per;
    const int eg = dout_exp(p, lane, b / p.Bf);
    if (idx >= per * p.B || rem >= (long long)HWl * c4n) return;
    const int c4 = (int)(rem % c4n), pix = (int)(rem / c4n);
    const int Y = pix / Wl, X = pix - Y * Wl, ttx = ttx_of(p, l), ttiles = ttiles_of(p, l);
    const int tt = (Y >> 3) * ttx + (X >> 3), u = (Y & 7) * 8 + (X & 7);
    const float4* __restrict__ src = (const float4*)(p.partial + p.poff[l]) + ((((size_t)b * ttiles + tt) * nsplit) * TQ + u) * c4n + c4;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nsplit; ++s) {
        const float4 v = src[(size_t)s * TQ * c4n];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const float unscale = ldexpf(1.0f, -(eg + e1));
    float* __restrict__ out = p.df2[l] + ((size_t)b * HWl + pix) * p.C + 4 * c4;
    const float r[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * c4 + j < p.C) {
            float v = (r[j] * unscale) / p.sqrt_c;
            if (p.accumulate) v += out[j];
            out[j] = v;
        }
}

// Several lookups per call: every lookup's gradients land in staging arrays of the workspace (virtual batch item
// k * Bf + b), this kernel adds them in lookup order -- the same float additions, in the same order, as nk single calls
// with accumulate = 1 make in place.  V floats per thread; grid (ceil(max n / V / 256), arrays)
struct KSumArgs {
    const float* stage[RAFTCORR_MAX_LEVELS + 1];
    float* out[RAFTCORR_MAX_LEVELS + 1];
    long long n[RAFTCORR_MAX_LEVELS + 1];         // floats per lookup
    int nk, accumulate;
};

template <int V>
__global__ __launch_bounds__(256)
void alt_bwd_ksum_kernel(KSumArgs a)
{
    const int arr = blockIdx.y;
    const long long n = a.n[arr], i = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
    if (i >= n) return;
    const float* __restrict__ st = a.stage[arr];
    float* __restrict__ out = a.out[arr] + i;
    float v[V];
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = st[i + j];
    if (a.accumulate) {
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] += out[j];
    }
    for (int k = 1; k < a.nk; ++k) {
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] += st[(long long)k * n + i + j];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) out[j] = v[j];
}

struct BwdPlan {
    int Cp, qtx, qtiles, ngroups, gw;
    int nsplit[RAFTCORR_MAX_LEVELS], ttiles[RAFTCORR_MAX_LEVELS], seg[RAFTCORR_MAX_LEVELS + 1];
    size_t poff[RAFTCORR_MAX_LEVELS];                        // floats
    size_t off_boxes, off_wins, off_partial, bytes;
    size_t off_stage1, off_stage2[RAFTCORR_MAX_LEVELS];       // several lookups per call: per-lookup gradients
    int ndmax;
    int lorder[RAFTCORR_MAX_LEVELS];
    bool any_split;
    // window mode
    AltGroupsPlan gp;
    size_t off_groups, off_dg, off_f1g;
};

// B: the virtual batch nk * Bf (nk lookups of Bf frame pairs per call)
BwdPlan make_plan(int B, int H, int W, int C, int L, int r, bool window, int nk = 1)
{
    BwdPlan pl = {};
    const int Bf = B / nk;
    pl.Cp = (C + 31) / 32 * 32;
    pl.qtx = (W + QW - 1) / QW;
    const int ptiles = pl.qtx * ((H + QH - 1) / QH);         // 8x8 pixel tiles
    pl.qtiles = ptiles;
    if (window) {
        pl.gp = alt_groups_plan(Bf, H, W);
        pl.qtiles = pl.gp.gmax;
    }
    // Channels per workgroup.  GW: the G image of a step is built once and multiplied with two partner slices (half the
    // construction work, the kernel's bottleneck) -- when the query tiles alone fill the chip; with few tiles (KITTI at
    // B = 1: 120) twice as many workgroups with one slice each balance better (measured both ways at both sizes).
    pl.gw = (long long)B * ptiles >= 256 ? GW : CG;
    pl.ngroups = (pl.Cp + pl.gw - 1) / pl.gw;
    auto pad8 = [](long long n) { return (int)((n + 7) / 8 * 8); };
    pl.seg[0] = window ? B * pad8((long long)pl.qtiles * pl.ngroups) : pad8((long long)B * pl.qtiles * pl.ngroups);
    // partners per split workgroup: finer where the query tiles alone do not fill the chip twice over and the launch's tail
    // counts (KITTI: 8 against 12: -1 us on top of the order below; 1080p: 12 against 8: -1 .. 3 % less reduce traffic)
    // (a function of ONE lookup's geometry, not of the virtual batch: the split sizes decide the order of the partial sums,
    // and a lookup's gradients must not depend on how many lookups share its call)
    int splitf = (long long)Bf * ptiles * ((pl.Cp + CG - 1) / CG) < 512 ? 8 : 12;
#if defined(RC_LAB)
    if (const char* e = getenv("RC_SPLITF")) splitf = atoi(e);
#endif
    long long wsteps[RAFTCORR_MAX_LEVELS] = {};              // expected steps of one workgroup of the level, x 256
    size_t pf = 0;
    for (int l = 0; l < L; ++l) {
        const int hl = H >> l, wl = W >> l;
        pl.ttiles[l] = ((wl + 7) / 8) * ((hl + 7) / 8);
        // query tiles that touch one target tile when the flow is smooth: ((8 + 2r + 2) * 2^l / 8 + 1)^2
        const long long side = (((long long)(8 + 2 * r + 2) << l) >> 3) + 1;
        const long long est = std::min<long long>(ptiles, side * side);
        pl.nsplit[l] = (int)std::min<long long>(MAXSPLIT, std::max<long long>(1, (est + splitf - 1) / splitf));
        if (l == 0) pl.nsplit[l] = 1;
        // (the estimate above ignores that the map clips the square: deliberately, the levels it overestimates get the
        // finest splits; for the ORDER the clipped count is the better guess of a workgroup's length)
        const long long ty_n = (H + QH - 1) / QH, tx_n = (W + QW - 1) / QW;
        wsteps[l] = 256 * std::min(side, ty_n) * std::min(side, tx_n) / pl.nsplit[l];
        pl.poff[l] = pf;
        if (pl.nsplit[l] > 1) { pf += (size_t)B * pl.ttiles[l] * pl.nsplit[l] * TQ * pl.Cp; pl.any_split = true; }
    }
    // Launch order of the d(fmap2) levels: behind the d(fmap1) workgroups (the longest: they walk the boxes of all levels)
    // level 0, then the split levels by expected workgroup length, the shortest last -- what is dispatched last decides how
    // long the launch's tail is (KITTI, window groups: 122 -> 114 us per call against the plain order 0 1 2 3).
    for (int i = 0; i < L; ++i) pl.lorder[i] = i;
    std::stable_sort(pl.lorder + (L > 1 ? 1 : 0), pl.lorder + L, [&](int a, int b) { return wsteps[a] > wsteps[b]; });
#if defined(RC_LAB)
    if (const char* e = getenv("RC_LORDER")) for (int i = 0; i < L && e[i]; ++i) pl.lorder[i] = e[i] - '0';
#endif
    for (int i = 0; i < L; ++i) {
        const int l = pl.lorder[i];
        pl.seg[i + 1] = pl.seg[i] + pad8((long long)B * pl.ttiles[l] * pl.nsplit[l] * pl.ngroups);
    }
    auto up = [](size_t n) { return (n + 255) / 256 * 256; };
    pl.ndmax = window ? (Bf * L * pl.qtiles + 3) & ~3 : DMAXP;
    pl.off_boxes = up((size_t)nk * pl.ndmax * sizeof(float)); // in front: partial maxima of |dout| per lookup
    pl.off_wins = up(pl.off_boxes + sizeof(int4) * (size_t)B * L * pl.qtiles);
    pl.off_partial = up(pl.off_wins + sizeof(int4) * (size_t)B * L * pl.qtiles * TQ);
    size_t end = up(pl.off_partial + pf * sizeof(float));
    if (window) {
        const int rd2 = (2 * r + 1) * (2 * r + 1);
        pl.off_groups = end; end = up(end + (size_t)nk * pl.gp.bytes);     // (for lookups whose groups the caller does not bring)
        pl.off_dg = end;     end = up(end + sizeof(float) * (size_t)B * L * pl.qtiles * rd2 * TQ);
        pl.off_f1g = end;    end = up(end + (size_t)B * pl.qtiles * pl.Cp * 256);
    }
    if (nk > 1) {
        pl.off_stage1 = end; end = up(end + sizeof(float) * (size_t)B * H * W * C);
        for (int l = 0; l < L; ++l) { pl.off_stage2[l] = end; end = up(end + sizeof(float) * (size_t)B * (H >> l) * (W >> l) * C); }
    }
    pl.bytes = end;
    return pl;
}

}  // namespace

// buckets = 8x8 cells of level-0 targets + one ring; group slots: every row of cells may end in a partly filled group
AltGroupsPlan alt_groups_plan(int B, int H, int W)
{
    AltGroupsPlan g = {};
    g.nbx = ((W + 7) >> 3) + 2; g.nby = ((H + 7) >> 3) + 2; g.nb = g.nbx * g.nby;
    g.nch = (H * W + WCHUNK - 1) / WCHUNK;
    g.gmax = (H * W + TQ - 1) / TQ + g.nb;
    auto up = [](size_t n) { return (n + 255) / 256 * 256; };
    g.off_qlist = up(sizeof(int) * (size_t)B);
    g.off_toff = up(g.off_qlist + sizeof(int) * (size_t)B * g.gmax * TQ);
    g.off_cnt = up(g.off_toff + sizeof(int) * (size_t)B * g.nb);
    g.bytes = up(g.off_cnt + sizeof(int) * (size_t)B * g.nb * g.nch);
    return g;
}

// count / scan / fill: the groups of `coords` -> groups[0 .. B) = groups in use, qlist behind them
int launch_alt_groups(const float* coords, int B, int H, int W, void* groups, size_t groups_bytes, hipStream_t s)
{
    const AltGroupsPlan g = alt_groups_plan(B, H, W);
    if (!groups || groups_bytes < g.bytes)
        return fail(RAFTCORR_EINVAL, "alt groups: buffer of %zu bytes needed, %zu given", g.bytes, groups_bytes);
    if (((uintptr_t)groups & 255) != 0) return fail(RAFTCORR_EINVAL, "alt groups: the buffer must be 256-byte aligned");
    if (g.nby > 1024 || g.nb > 65535 || g.nch > 65535 || B > 65535 || (size_t)H * W >= ((size_t)1 << 30))
        return fail(RAFTCORR_EUNSUPPORTED, "alt groups: %d x %d x %d is too large", B, H, W);
    WSortArgs w = {};
    w.coords = coords; w.N = H * W; w.nbx = g.nbx; w.nby = g.nby; w.nb = g.nb; w.nch = g.nch; w.gmax = g.gmax;
    w.ngr = (int*)groups;
    w.qlist = (int*)((char*)groups + g.off_qlist);
    w.toff = (int*)((char*)groups + g.off_toff);
    w.cnt = (int*)((char*)groups + g.off_cnt);
    int rc;
    const int nch64 = (H * W + 63) / 64;
    bool small = H * W <= WS_MAXQ && ws_small_lds(nch64, g.nb, g.nby) <= 128 * 1024;
#if defined(RC_LAB)
    if (getenv("RC_GROUPS_LEGACY")) small = false;
#endif
    if (small) {                                             // one launch instead of three
        static thread_local bool allowed[64] = {};           // > 64 KiB of dynamic LDS: opt in once per device
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess && !allowed[dev & 63]) {
            e = hipFuncSetAttribute((const void*)alt_wgroups_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e == hipSuccess) allowed[dev & 63] = true;
        }
        if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "alt groups: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(alt_wgroups_small_kernel, dim3(B), dim3(1024), ws_small_lds(nch64, g.nb, g.nby), s, w);
        return check_launch("alt_wgroups_small_kernel");
    }
    hipLaunchKernelGGL(alt_wbucket_kernel<false>, dim3(g.nb, g.nch, B), dim3(64), 0, s, w);
    if ((rc = check_launch("alt_wbucket_kernel<count>"))) return rc;
    hipLaunchKernelGGL(alt_wscan_kernel, dim3(B), dim3(1024), 0, s, w);
    if ((rc = check_launch("alt_wscan_kernel"))) return rc;
    hipLaunchKernelGGL(alt_wbucket_kernel<true>, dim3(g.nb, g.nch, B), dim3(64), 0, s, w);
    return check_launch("alt_wbucket_kernel<fill>");
}

size_t alt_tiles_bytes(int B, int C, int H, int W)
{
    const size_t Cp = (size_t)(C + 31) / 32 * 32;
    return (size_t)B * ((W + 7) / 8) * ((H + 7) / 8) * Cp * 256;
}

int launch_alt_pack_tiles_set(const float* const* fmaps, const int* hs, const int* ws, const float* const* maxparts,
                              void* const* packed, int n, int B, int C, hipStream_t s)
{
    PackSet ps = {};
    ps.n = n; ps.C = C; ps.Cp = (C + 31) / 32 * 32;
    for (int m = 0; m < n; ++m) {
        ps.src[m] = fmaps[m]; ps.dst[m] = (uint4*)packed[m]; ps.maxpart[m] = maxparts[m]; ps.h[m] = hs[m]; ps.w[m] = ws[m];
        ps.blk0[m + 1] = ps.blk0[m] + ((ws[m] + 7) / 8) * ((hs[m] + 7) / 8);
    }
    hipLaunchKernelGGL(alt_pack_tiles_kernel, dim3(ps.blk0[n], ps.Cp / 32, B), dim3(256), 0, s, ps);
    return check_launch("alt_pack_tiles_kernel");
}

int launch_alt_pack_tiles(const float* fmap, int B, int C, int H, int W, const float* maxpart, void* packed, hipStream_t s)
{
    return launch_alt_pack_tiles_set(&fmap, &H, &W, &maxpart, &packed, 1, B, C, s);
}

size_t alt_lookup_bwd_workspace_bytes(int B, int H, int W, int C, int L, int r, bool window, int nk)
{
    return make_plan(B * std::max(nk, 1), H, W, C, L, r, window, std::max(nk, 1)).bytes;
}

// nk lookups of the same B frame pairs in one call (1 <= nk <= ALT_KMAX): coords / douts (/ groups) are HOST arrays of nk
// device pointers.  window: f1 is the forward pass' packed copy of fmap1 (the tile-major partner images are gathered per
// call for the query groups) and groups[k] the group buffer of lookup k as launch_alt_groups left it, or nullptr: built
// here; else f1 is the tile-major copy.  df1 / df2 receive (accumulate: are added) the SUM over the lookups, added in
// array order.
int launch_alt_lookup_bwd16_many(const void* f1, const void* const* f2t, const float* maxpart, int B, int H, int W, int C, int L,
                                 int r, int nk, const float* const* coords, const float* const* douts, const void* const* groups,
                                 bool window, float* df1, float* const* df2, int accumulate,
                                 void* workspace, size_t workspace_bytes, hipStream_t s)
{
    if (r > 7) return fail(RAFTCORR_EUNSUPPORTED, "alt_lookup_bwd: radius %d > 7", r);
    if (nk < 1 || nk > ALT_KMAX) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd: %d lookups per call (1 .. %d)", nk, ALT_KMAX);
    if ((long long)B * nk > 65535) return fail(RAFTCORR_EUNSUPPORTED, "alt_lookup_bwd: %d lookups x batch %d is too large", nk, B);
    const int VB = B * nk;
    const BwdPlan pl = make_plan(VB, H, W, C, L, r, window, nk);
    if (!workspace || workspace_bytes < pl.bytes)
        return fail(RAFTCORR_EINVAL, "alt_lookup_bwd: workspace of %zu bytes needed, %zu given", pl.bytes, workspace_bytes);
    if (((uintptr_t)workspace & 255) != 0) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd: workspace must be 256-byte aligned");
    if ((long long)pl.seg[L] <= 0 || (size_t)VB * pl.qtiles * pl.Cp * 16 >= ((size_t)1 << 31) ||
        (window && ((size_t)H * W * (pl.Cp / 32) * 8 >= ((size_t)1 << 31) ||
                    (size_t)VB * L * pl.qtiles * (2 * r + 1) * (2 * r + 1) * TQ >= ((size_t)1 << 31))))
        return fail(RAFTCORR_EUNSUPPORTED, "alt_lookup_bwd: %d x %d x %d x %d x %d lookups is too large", B, H, W, C, nk);
    AltBwdArgs a = {};
    a.f1t = (const uint4*)f1;
    a.nk = nk; a.Bf = B; a.window = window;
    float* stage1 = nk > 1 ? (float*)((char*)workspace + pl.off_stage1) : df1;
    for (int l = 0; l < L; ++l) {
        a.f2t[l] = (const uint4*)f2t[l]; a.hl[l] = H >> l; a.wl[l] = W >> l;
        a.df2[l] = nk > 1 ? (float*)((char*)workspace + pl.off_stage2[l]) : df2[l];
        a.poff[l] = (long long)pl.poff[l]; a.nsplit[l] = pl.nsplit[l];
    }
    for (int l = 0; l <= L; ++l) a.seg[l] = pl.seg[l];
    for (int l = 0; l < L; ++l) a.lorder[l] = pl.lorder[l];
    for (int k = 0; k < nk; ++k) {
        if (!coords[k] || !douts[k]) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd: null coords / dout of lookup %d", k);
        a.coords_k[k] = coords[k]; a.dout_k[k] = douts[k];
    }
    a.maxpart = maxpart; a.df1 = stage1;
    a.dmax = (const float*)workspace;
    a.ndmax = pl.ndmax;
    a.boxes = (int4*)((char*)workspace + pl.off_boxes);
    a.wins = (int4*)((char*)workspace + pl.off_wins);
    a.partial = (float*)((char*)workspace + pl.off_partial);
    a.B = VB; a.C = C; a.Cp = pl.Cp; a.H = H; a.W = W; a.L = L; a.r = r;
    a.qtx = pl.qtx; a.qtiles = pl.qtiles; a.ngroups = pl.ngroups; a.gw = pl.gw;
    a.sqrt_c = sqrtf((float)C);
    a.accumulate = nk > 1 ? 0 : accumulate;                   // (several lookups: the staging arrays are overwritten)
    const int rd2 = (2 * r + 1) * (2 * r + 1);
    int rc;
    const void* f1p_window = nullptr;
    if (window) {
        f1p_window = f1;
        for (int k = 0; k < nk; ++k) {
            const void* g = groups ? groups[k] : nullptr;
            if (!g) {
                void* own = (char*)workspace + pl.off_groups + (size_t)k * pl.gp.bytes;
                if ((rc = launch_alt_groups(coords[k], B, H, W, own, pl.gp.bytes, s))) return rc;
                g = own;
            } else if (((uintptr_t)g & 255) != 0) {
                return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_sorted: groups must be 256-byte aligned");
            }
            a.ngr_k[k] = (const int*)g;
            a.qlist_k[k] = (const int*)((const char*)g + pl.gp.off_qlist);
        }
        a.dg = (float*)((char*)workspace + pl.off_dg);
        a.f1t = (const uint4*)((char*)workspace + pl.off_f1g);
    }
    const long long ngather = window ? (long long)VB * pl.qtiles * (pl.Cp / 32) : 0;
    const long long nmaxwg = window ? 0 : (long long)nk * DMAXP;
    if ((long long)VB * L * pl.qtiles + 8 + nmaxwg + ngather >= ((long long)1 << 31))
        return fail(RAFTCORR_EUNSUPPORTED, "alt_lookup_bwd: %d x %d x %d x %d x %d lookups is too large", B, H, W, C, nk);
    hipLaunchKernelGGL(alt_bwd_pre_kernel, dim3((unsigned)((((long long)VB * L * pl.qtiles + 7) & ~7ll) + nmaxwg + ngather)),
                       dim3(window ? 256 : 64), 0, s,
                       a, VB * L * pl.qtiles, (size_t)B * L * rd2 * H * W, (const uint4*)f1p_window);
    rc = check_launch("alt_bwd_pre_kernel");
    if (rc) return rc;
    const size_t sm = (size_t)(IMG_A + IMG_B) * sizeof(uint4) + (size_t)TQ * std::max(rd2, 96) * sizeof(float);
    static thread_local bool allowed[64] = {};               // > 64 KiB of dynamic LDS: opt in once per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && !allowed[dev & 63]) {
        e = hipFuncSetAttribute((const void*)alt_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (e == hipSuccess) allowed[dev & 63] = true;
    }
    if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "alt_lookup_bwd: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(alt_bwd_kernel, dim3(pl.seg[L]), dim3(256), sm, s, a);
    rc = check_launch("alt_bwd_kernel");
    if (rc) return rc;
    if (pl.any_split) {
        size_t most = 0;
        for (int l = 0; l < L; ++l)
            if (pl.nsplit[l] > 1) most = std::max(most, (size_t)VB * ((((size_t)(H >> l) * (W >> l) * (pl.Cp / 4)) + 63) / 64 * 64));
        hipLaunchKernelGGL(alt_bwd_reduce_kernel, dim3((unsigned)((most + 255) / 256), L), dim3(256), 0, s, a);
        if ((rc = check_launch("alt_bwd_reduce_kernel"))) return rc;
    }
    if (nk == 1) return RAFTCORR_OK;
    KSumArgs ks = {};
    ks.nk = nk; ks.accumulate = accumulate;
    ks.stage[0] = stage1; ks.out[0] = df1; ks.n[0] = (long long)B * H * W * C;
    long long most = ks.n[0];
    bool vec = ((uintptr_t)df1 & 15) == 0 && ks.n[0] % 4 == 0;
    for (int l = 0; l < L; ++l) {
        ks.stage[l + 1] = a.df2[l]; ks.out[l + 1] = df2[l]; ks.n[l + 1] = (long long)B * (H >> l) * (W >> l) * C;
        most = std::max(most, ks.n[l + 1]);
        vec = vec && ((uintptr_t)df2[l] & 15) == 0 && ks.n[l + 1] % 4 == 0;
    }
    if (vec) hipLaunchKernelGGL(alt_bwd_ksum_kernel<4>, dim3((unsigned)((most / 4 + 255) / 256), L + 1), dim3(256), 0, s, ks);
    else     hipLaunchKernelGGL(alt_bwd_ksum_kernel<1>, dim3((unsigned)((most + 255) / 256), L + 1), dim3(256), 0, s, ks);
    return check_launch("alt_bwd_ksum_kernel");
}

// one lookup: f1p_window != nullptr selects window mode (f1t is ignored); groups: see above
int launch_alt_lookup_bwd16(const void* f1t, const void* const* f2t, const float* maxpart, int B, int H, int W, int C, int L,
                            int r, const float* coords, const float* dout, float* df1, float* const* df2, int accumulate,
                            void* workspace, size_t workspace_bytes, hipStream_t s, const void* f1p_window, const void* groups)
{
    const bool window = f1p_window != nullptr;
    return launch_alt_lookup_bwd16_many(window ? f1p_window : f1t, f2t, maxpart, B, H, W, C, L, r, 1, &coords, &dout, &groups, window,
                                        df1, df2, accumulate, workspace, workspace_bytes, s);
}

}  // namespace raftcorr
