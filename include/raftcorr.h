/* raftcorr.h -- C ABI of libraftcorr.so, the MI355X (gfx950) implementation of RAFT's
 * correlation hot path.
 *
 * Every entry point takes raw device pointers, explicit sizes and a hipStream_t (passed as
 * void*); nothing here depends on torch.  The library never allocates or frees device
 * memory and keeps no reference to any buffer after a call returns: the caller owns all
 * buffers (the Python host code allocates them through PyTorch's caching allocator).
 * All kernels are launched asynchronously on the stream that is passed in.
 *
 * Return value: 0 on success, a negative RAFTCORR_E* code otherwise; the message for the
 * calling thread's last failure is returned by raftcorr_last_error().
 *
 * The comments name the reference interface (damien911224/RAFT, paths relative to the
 * repository root) that each entry point replaces.
 */
#ifndef RAFTCORR_H
#define RAFTCORR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAFTCORR_VERSION 1
#define RAFTCORR_MAX_LEVELS 8

#define RAFTCORR_OK            0
#define RAFTCORR_EINVAL       (-1)   /* bad argument (null pointer, non-positive size, ...) */
#define RAFTCORR_EUNSUPPORTED (-2)   /* valid request this build does not implement */
#define RAFTCORR_ELAUNCH      (-3)   /* hipGetLastError() reported a launch failure */

/* lookup conventions */
#define RAFTCORR_MODE_FORK      0    /* core/corr.py:29-51 as shipped: coords (B,N,2) (x,y) level-0
                                        pixels, SAME coords at every level, out (B,N,L*rd*rd) */
#define RAFTCORR_MODE_CANONICAL 1    /* what core/raft.py:125 + core/update.py:82-91 consume and
                                        core/corr.py:75-92 keeps: coords (B,2,H,W), coords/2**l per
                                        level, out (B,L*rd*rd,H,W) */

int raftcorr_version(void);
const char* raftcorr_last_error(void);

/* ---- pyramid memory layout -------------------------------------------------------------
 * Level l of the pyramid is a (B*H*W) x slab[l] float32 matrix: one SLAB of slab[l] elements per
 * query pixel (row q of the all-pairs matrix), holding that query's H_l x W_l correlation map
 * with the target pixels in "patch-tile order" (8x16 patches of four 4x8 tiles; one tile = one
 * 128-byte line; raft_amd/csrc/layout.h), padded at the far edges.  The order is what lets a
 * lookup window touch few cache lines and the build kernel pool in registers; callers that
 * want the reference's (B*H*W, 1, H_l, W_l) tensors (core/corr.py:24-27) gather them with
 * raftcorr_level_positions (the Python host does, lazily, for CorrBlock.corr_pyramid). */

/* Host-only.  Logical level sizes H_l = floor(H_{l-1}/2) (F.avg_pool2d(corr, 2, stride=2),
 * core/corr.py:25-27) and slab sizes in elements.  hl, wl: int32[num_levels]; slab: int64[num_levels]. */
int raftcorr_level_layout(int H, int W, int num_levels, int32_t* hl, int32_t* wl, int64_t* slab);

/* Host-only.  pos[y*W_l + x] = offset of logical cell (y, x) of level `level` inside a slab.
 * pos: int32[H_l * W_l]. */
int raftcorr_level_positions(int H, int W, int level, int32_t* pos);

/* Host-only.  Scratch bytes raftcorr_build_pyramid / raftcorr_build_bwd need (caller-allocated
 * device memory; contents are don't-care before and after the call). */
int64_t raftcorr_build_workspace_bytes(int B, int D, int H, int W);
int64_t raftcorr_build_bwd_workspace_bytes(int B, int D, int H, int W);

/* Replaces CorrBlock.__init__ + CorrBlock.corr (core/corr.py:13-27, :53-61):
 *   level0[b,q,t] = <fmap1[b,:,q], fmap2[b,:,t]> / sqrt(D),  q,t in [0, H*W)
 *   level l       = 2x2 mean of level l-1 over the target (h2,w2) dims, floor truncation.
 * fmap1, fmap2: (B,D,H,W) float32 contiguous.  levels: HOST array of num_levels DEVICE pointers;
 * levels[l] receives B*H*W slabs of slab[l] floats (fully overwritten, padding included).
 * A small pack pass + one fused launch produce levels 0..3 (fp32 MFMA contraction, pooling in
 * the epilogue); levels >= 4 are pooled from their predecessor by a further small kernel. */
int raftcorr_build_pyramid(const float* fmap1, const float* fmap2,
                           int B, int D, int H, int W, int num_levels,
                           float* const* levels, void* workspace, int64_t workspace_bytes,
                           void* stream);

/* Same call with a choice of arithmetic for the contraction (the pyramid is fp32 in every case):
 *   RAFTCORR_PREC_FP32    v_mfma_f32_32x32x2_f32, an exact fp32 FMA chain (= raftcorr_build_pyramid);
 *   RAFTCORR_PREC_BF16    operands rounded to bfloat16 (RNE), fp32 accumulation on the bf16 matrix
 *                         cores: the "mixed precision" build (the reference runs its encoders under
 *                         autocast, core/raft.py:99-103; the north star's bf16 option); differs from
 *                         fp32 by the input rounding, <= 1e-2 relative (2e-3 of max|level0| at D = 256);
 *   RAFTCORR_PREC_BF16X3  each operand split into three bfloat16 pieces (24 significand bits) and the
 *                         six products of weight >= 2^-16 accumulated in fp32: fp32-grade results
 *                         (error of the order of an fp32 FMA chain's own rounding, far inside the
 *                         1e-5 parity bar) at bf16 matrix-core speed; any finite fp32 inputs;
 *   RAFTCORR_PREC_FP16X2  each map scaled by a power of two into the top of fp16's range (one extra
 *                         max|x| pass), each operand split into two fp16 pieces (22 significand
 *                         bits), three products (hi*hi + hi*lo + lo*hi) accumulated in fp32 and
 *                         unscaled exactly: also fp32-grade (2e-7 of max|level0| at D = 256, the
 *                         fp32 FMA chain itself is at 7e-7) at half the matrix-core work of BF16X3.
 *                         Elements more than ~2^22 below their map's max lose low-order bits.
 * Same workspace size for all. */
#define RAFTCORR_PREC_FP32   0
#define RAFTCORR_PREC_BF16   1
#define RAFTCORR_PREC_BF16X3 2
#define RAFTCORR_PREC_FP16X2 3
int raftcorr_build_pyramid_prec(const float* fmap1, const float* fmap2,
                                int B, int D, int H, int W, int num_levels,
                                float* const* levels, void* workspace, int64_t workspace_bytes,
                                int precision, void* stream);

/* 16-bit volume (no reference line: the reference's volume is always float32; north star "bf16" option, SURVEY 8
 * shape table "4.26 GB (bf16 2.13)").  raftcorr_build_pyramid_vol = raftcorr_build_pyramid_prec with the pyramid
 * stored as fp16 or bf16 ELEMENTS in the same slab / patch-tile element order (levels[l] then points at
 * B*H*W*slab[l] 16-bit elements): every value is rounded once from the fp32 accumulator, pooled levels from
 * their fp32 sums.  Halves the build's store traffic and the lookup's line traffic; results differ from the
 * float32 volume by the element rounding (bf16: 2^-9 relative, the <= 1e-2 class; fp16: 2^-12, values beyond
 * +-65504 become inf).  num_levels <= 4.  raftcorr_lookup_fwd_vol reads such a pyramid (volume_dtype must be
 * the one it was built with); output float32 as raftcorr_lookup_fwd.  The backward entry points never read the
 * volume values (raftcorr_lookup_bwd with dcoords == NULL, raftcorr_build_bwd) and serve either kind. */
#define RAFTCORR_VOL_FP32 0
#define RAFTCORR_VOL_FP16 1
#define RAFTCORR_VOL_BF16 2
int raftcorr_build_pyramid_vol(const float* fmap1, const float* fmap2,
                               int B, int D, int H, int W, int num_levels,
                               void* const* levels, void* workspace, int64_t workspace_bytes,
                               int precision, int volume_dtype, void* stream);
int raftcorr_lookup_fwd_vol(const void* const* levels, int volume_dtype, int B, int H, int W, int num_levels,
                            int radius, const float* coords, int mode, float* out, void* stream);

/* Allocation-free variant of raftcorr_build_pyramid_prec.  NO COUNTERPART IN THE REFERENCE: it exists
 * for callers that cannot allocate device memory -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc -- and therefore cannot bring the workspace the other build
 * entry points ask for.  Same arguments minus the workspace, same kernels, bit-identical results;
 * the scratch (operand panels + range-pass partial maxima) is a device array built into the library,
 * RAFTCORR_NOALLOC_MAX_WORKSPACE bytes per GPU, held until the process ends.  The call allocates and
 * frees nothing (no hipMalloc/hipFree, no allocating library routine); its first call on a device
 * looks the array's address up once (hipGetSymbolAddress) -- make that call outside graph capture.
 * Shapes with raftcorr_build_workspace_bytes(B, D, H, W) > RAFTCORR_NOALLOC_MAX_WORKSPACE (e.g. 1080p,
 * or B = 6 Things crops) are rejected with RAFTCORR_EUNSUPPORTED; Sintel and KITTI at B = 1 fit.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (different
 * streams) race; the array keeps the last call's contents and no call depends on them. */
#define RAFTCORR_NOALLOC_MAX_WORKSPACE (32u << 20)
int raftcorr_build_pyramid_noalloc(const float* fmap1, const float* fmap2,
                                   int B, int D, int H, int W, int num_levels,
                                   float* const* levels, int precision, void* stream);

/* Replaces CorrBlock.__call__ + utils.bilinear_sampler (core/corr.py:29-51,
 * core/utils/utils.py:57-65): for every query q and level l the (2r+1)^2 window of 4-tap
 * bilinear samples (zero padding) of levels[l][b*N+q] at (x + a - r, y + b - r); output channel
 * l*rd*rd + a*rd + b (a = x offset = slow index).  mode selects the coordinate/ output
 * convention (see RAFTCORR_MODE_*).  levels: HOST array of DEVICE pointers laid out as
 * raftcorr_build_pyramid writes them (slabs in patch-tile order).  out is fully overwritten. */
int raftcorr_lookup_fwd(const float* const* levels, int B, int H, int W, int num_levels,
                        int radius, const float* coords, int mode, float* out, void* stream);

/* Lookup fused with its consumer in RAFT's update block, the motion encoder's 1x1 convolution
 * `convc1` + bias + ReLU (core/update.py:83,90: cor = F.relu(self.convc1(corr))):
 *   out[b, c, q] = act( sum_k weight[c, k] * lookup(coords)[b, k, q] + bias[c] ),  k = l*rd*rd + a*rd + b
 * canonical coordinates (RAFTCORR_MODE_CANONICAL), num_levels <= 4, radius 3 or 4.  The
 * (B, L*rd*rd, H, W) lookup result is never written to memory; the product runs on the 16-bit
 * matrix cores in the bf16x3 scheme of raftcorr_build_pyramid_prec (lookup tile and weight split into
 * three bfloat16 pieces, six v_mfma_f32_32x32x16_bf16 products, fp32 accumulation, partial sums added
 * in a fixed order): fp32-grade, bit-reproducible.  This entry point is the forward pass; the Python
 * host differentiates it (raft_amd/corr.py:_LookupConv: the lookup is re-created with
 * raftcorr_lookup_fwd, W^T d(pre) goes through raftcorr_lookup_bwd / the deferred backward below).
 * The weight is packed once per model with raftcorr_conv1x1_pack_weight (weight: (Cout, Cin) row-major
 * = the conv's (Cout, Cin, 1, 1) tensor; packed: raftcorr_conv1x1_packed_bytes(Cout, Cin) bytes of
 * device memory).  bias: (Cout) or NULL.  relu: 0/1.  out: (B, Cout, H, W), overwritten. */
int64_t raftcorr_conv1x1_packed_bytes(int Cout, int Cin);
int raftcorr_conv1x1_pack_weight(const float* weight, int Cout, int Cin, float* packed,
                                 int64_t packed_bytes, void* stream);
int raftcorr_lookup_conv1x1_fwd(const float* const* levels, int B, int H, int W, int num_levels,
                                int radius, const float* coords, const float* packed_weight,
                                const float* bias, int Cout, int relu, float* out, void* stream);

/* Backward of raftcorr_lookup_fwd (replaces grid_sampler_2d_backward reached through
 * core/utils/utils.py:65).  dlevels[l] (same layout as levels, float32) is ACCUMULATED into
 * (+=): zero it once, then call this for every lookup of the frame pair; no dense per-call
 * gradient volume is created.  A (query, tap) cell is owned by exactly one thread, so no
 * atomics are used and the result is deterministic.
 * dcoords (nullable): gradient w.r.t. coords, same shape as coords, overwritten; needs levels.
 * dlevels may be NULL when only dcoords is wanted (levels may be NULL when it is not). */
int raftcorr_lookup_bwd(const float* const* levels, float* const* dlevels,
                        int B, int H, int W, int num_levels, int radius,
                        const float* coords, int mode, const float* dout,
                        float* dcoords, void* stream);

/* Backward of raftcorr_build_pyramid (replaces avg_pool2d_backward + the scale + bmm backward
 * behind core/corr.py:26,59,61).  dlevels: accumulated pyramid gradient (padding cells must be
 * zero: zero the buffers once, lookups never touch padding); level 0 is used as scratch and is
 * CLOBBERED (it receives the folded level-0 gradient); levels 3.. are clobbered when L > 4.
 * dfmap1, dfmap2: (B,D,H,W) float32, overwritten.  No atomics: split-K partial products are kept in
 * the workspace and added in a fixed order, so two calls on the same inputs give the same bits. */
int raftcorr_build_bwd(const float* fmap1, const float* fmap2, int B, int D, int H, int W,
                       int num_levels, float* const* dlevels,
                       float* dfmap1, float* dfmap2, void* workspace, int64_t workspace_bytes,
                       void* stream);

/* Deferred form of the two calls above (same reference lines: grid_sampler_2d_backward per lookup behind
 * core/utils/utils.py:65, then avg_pool2d_backward / scale / bmm backward behind core/corr.py:26,59,61), for the
 * training path where ONE frame pair is looked up K times (core/raft.py:123-125) before autograd reaches the build:
 * a lookup's backward pass only records its (coords, d(out)); raftcorr_build_bwd_deferred then makes every query's
 * folded level-0 gradient row in LDS from all K records and all levels and writes it once -- no dense gradient
 * pyramid is zero-filled, read-modified-written per lookup or folded in a separate pass.
 *   raftcorr_lookup_bwd_record   canonical-mode d(out) (B, channels, H, W) -> record (B, H*W, channels), the layout the
 *                         fold reads (a query's channels contiguous); channels = L*rd*rd.  Fork-mode d(out)
 *                         (B, N, channels) IS a record already.
 *   raftcorr_lookup_bwd_record_many  the same for n_lookups gradients in one call (douts / records: HOST arrays of
 *                         DEVICE pointers): one launch per 16 lookups, whose rounds of workgroups overlap -- for a
 *                         caller that keeps the d(out) tensors until the build node runs (12 lookups at Things B=6:
 *                         one launch instead of 12 x 18 us).  Same bits as n_lookups single calls.
 *   raftcorr_build_bwd_deferred  coords / records: HOST arrays of n_lookups DEVICE pointers (coords as given to the
 *                         lookup: (B,N,2) fork, (B,2,H,W) canonical; any order: the sum runs in array order, the same
 *                         bits on every run).  dlevels: NULL, or a dense pyramid gradient as raftcorr_build_bwd takes it
 *                         (someone differentiated the pyramid tensors directly): folded first, the records are added;
 *                         its level 0 is clobbered.  dfmap1, dfmap2 (B,D,H,W) overwritten.  workspace:
 *                         raftcorr_build_bwd_deferred_workspace_bytes(...) bytes, 256-byte aligned, contents don't-care.
 *   raftcorr_build_bwd_deferred_supported  1 when the shape fits (num_levels <= 4, radius <= 7, the query's level rows
 *                         + staged records <= 150 KiB of LDS: up to ~190 x 190 maps at 4 levels), else 0: use
 *                         raftcorr_lookup_bwd + raftcorr_build_bwd. */
int64_t raftcorr_build_bwd_deferred_workspace_bytes(int B, int D, int H, int W);
int raftcorr_build_bwd_deferred_supported(int H, int W, int num_levels, int radius);
int raftcorr_lookup_bwd_record(const float* dout, int B, int H, int W, int channels, float* record, void* stream);
int raftcorr_lookup_bwd_record_many(const float* const* douts, int n_lookups, int B, int H, int W, int channels,
                                    float* const* records, void* stream);
int raftcorr_build_bwd_deferred(const float* fmap1, const float* fmap2, int B, int D, int H, int W, int num_levels,
                                int radius, int mode, int n_lookups, const float* const* coords,
                                const float* const* records, float* const* dlevels, float* dfmap1, float* dfmap2,
                                void* workspace, int64_t workspace_bytes, void* stream);

/* Drop-in for alt_cuda_corr.forward (alt_cuda_corr/correlation.cpp:23-33 ->
 * correlation_kernel.cu:260-286): one level of on-the-fly correlation.
 * fmap1 (B,H1,W1,C), fmap2 (B,H2,W2,C), coords (B,N,H1,W1,2) -> corr (B,N,rd*rd,H1,W1),
 * all float32 contiguous; corr is fully overwritten and is NOT scaled by 1/sqrt(C). */
int raftcorr_alt_forward(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                         int B, int N, int H1, int W1, int H2, int W2, int C, int radius,
                         void* stream);

/* Drop-in for alt_cuda_corr.backward (alt_cuda_corr/correlation.cpp:36-48 ->
 * correlation_kernel.cu:288-324).  fmap1_grad (B,H1,W1,C), fmap2_grad (B,H2,W2,C) and
 * coords_grad (B,N,H1,W1,2) are fully overwritten; coords_grad is all zeros, as in the
 * reference (allocated at :307 and never written). coords_grad may be NULL.  fmap1_grad is summed in LDS by the
 * workgroup that owns the query tile (over the N coordinate sets and the window boxes, in program order) and
 * written once: no zero fill, no atomics, the same bits on every run (C <= 512; beyond that it falls back to float
 * atomics).  fmap2_grad is a scatter across query tiles: like the reference kernel (atomicAdd at :229-238) it is
 * zero-filled and added into with float atomics, so its summation order -- not the value beyond rounding -- varies
 * from run to run; the all-level call below (raftcorr_alt_lookup_bwd) is the atomic-free, reproducible path. */
int raftcorr_alt_backward(const float* fmap1, const float* fmap2, const float* coords,
                          const float* corr_grad, float* fmap1_grad, float* fmap2_grad,
                          float* coords_grad,
                          int B, int N, int H1, int W1, int H2, int W2, int C, int radius,
                          void* stream);

/* Replaces AlternateCorrBlock (core/corr.py:64-92) for the per-iteration call: all levels, the per-level
 * coords/2**l, the stack/reshape and the 1/sqrt(C) scale in ONE launch, on the 16-bit matrix cores at
 * fp32-grade accuracy.  The features are packed ONCE per frame pair (the constructor's job, as the
 * reference builds its feature pyramid once, corr.py:69-73; it re-permutes the maps on every call, :83-84):
 *   raftcorr_alt_scales   partial maxima of |fmap1| and |fmap2| -> scales (2 x 128 floats, device memory);
 *                         every map is scaled by a power of two into the top of fp16's range
 *   raftcorr_alt_pack     one map (B,C,H,W) float32 NCHW -> packed: per pixel and 32-channel chunk one
 *                         128-byte line of two fp16 pieces per value (x = hi + lo, 22 significand bits);
 *                         scale: the 128 partial maxima of THAT map's family (scales for fmap1,
 *                         scales + 128 for fmap2 and all of its avg-pooled levels, whose maxima are no larger);
 *                         packed: raftcorr_alt_packed_bytes(B, C, H, W) bytes, caller-allocated
 *   raftcorr_alt_lookup_fwd   packed1: fmap1; packed2_levels: HOST array of DEVICE pointers, level l = packed
 *                         F.avg_pool2d chain of fmap2 ((B,C,H>>l,W>>l)); coords (B,2,H,W); out (B, L*rd*rd, H, W),
 *                         overwritten.  Values: the reference's sum of products in fp32-grade arithmetic
 *                         (three fp16 MFMA terms hi*hi + hi*lo + lo*hi, fp32 accumulation, exact unscale),
 *                         then / sqrt(C) as corr.py:92. */
int64_t raftcorr_alt_packed_bytes(int B, int C, int H, int W);
int raftcorr_alt_scales(const float* fmap1, const float* fmap2, int B, int C, int H, int W, float* scales, void* stream);
int raftcorr_alt_pack(const float* fmap, int B, int C, int H, int W, const float* scale, void* packed,
                      int64_t packed_bytes, void* stream);
int raftcorr_alt_lookup_fwd(const void* packed1, const void* const* packed2_levels, const float* scales,
                            int B, int H, int W, int C, int num_levels, int radius,
                            const float* coords, float* out, void* stream);

/* raftcorr_alt_lookup_fwd with the query tiles formed by WINDOW position instead of pixel position: same arguments,
 * same results up to the last bit of a sum (a query's taps do not depend on which queries share its tile, but the
 * tile's box does, and a window split over two passes of a box adds its halves in another order).
 * A pixel tile's cost grows with the disagreement of its 64 flow vectors (the product covers the bounding box of their
 * windows); tiles of queries whose windows start in the same row of 8x8 target cells, neighbours in that row, cost
 * the same for any flow field.  The tiles ("groups") are built by three small launches without atomics -- they are a
 * function of the coordinates alone, so the call is bit-reproducible -- and are left in the workspace:
 * raftcorr_alt_lookup_bwd_sorted takes them as `groups` for the same coordinates instead of building them again.
 * workspace: raftcorr_alt_lookup_fwd_sorted_workspace_bytes(...) bytes, 256-byte aligned, contents don't-care before. */
int64_t raftcorr_alt_lookup_fwd_sorted_workspace_bytes(int B, int H, int W, int num_levels);
int raftcorr_alt_lookup_fwd_sorted(const void* packed1, const void* const* packed2_levels, const float* scales,
                                   int B, int H, int W, int C, int num_levels, int radius,
                                   const float* coords, float* out,
                                   void* workspace, int64_t workspace_bytes, void* stream);

/* Backward of raftcorr_alt_lookup_fwd w.r.t. the features, all levels (the reference exports corr_backward,
 * correlation.cpp:36-48 -> correlation_kernel.cu:123-256, but never wires it into autograd; per level this is that
 * kernel's arithmetic: g = bilinear-weighted sum of <= 4 d(out) cells per tap, d(fmap1) += g * fmap2,
 * d(fmap2) += g * fmap1, then the 1/sqrt(C) of corr.py:92).  No atomics and no zero-filled outputs: every element of
 * dfmap1 and of each dfmap2 level has one writer, sums run in a fixed order, results are bit-reproducible.
 *   raftcorr_alt_pack_tiles   one map (B,C,H,W) float32 NCHW -> TILE-MAJOR packed copy for the backward pass: per 8x8
 *                         pixel tile and channel 256 bytes of two fp16 pieces per value (same power-of-two scaling and
 *                         `scale` argument as raftcorr_alt_pack); packed: raftcorr_alt_tiles_bytes(B, C, H, W) bytes
 *   raftcorr_alt_lookup_bwd   tiles1: fmap1; tiles2_levels: HOST array of DEVICE pointers, level l = packed avg-pooled
 *                         fmap2 ((B,C,H>>l,W>>l)); scales: as for the forward call; coords (B,2,H,W); dout
 *                         (B, L*rd*rd, H, W); dfmap1 (B,H,W,C) and dfmap2_levels[l] (B,H_l,W_l,C) float32 NHWC (the
 *                         layout of the reference kernel's gradients, correlation_kernel.cu:304-306), fully
 *                         overwritten when accumulate == 0 and added to (every element read and written by its one
 *                         writer, so still reproducible) when accumulate != 0: the gradients of the 12 lookups of one
 *                         frame pair sum up in place, with no separate additions; workspace: raftcorr_alt_lookup_bwd_workspace_bytes(...) bytes of caller-allocated
 *                         device memory, 256-byte aligned, contents don't-care before and after.
 *                         Arithmetic: fp16x2 split operands (features as packed; g scaled by a power of two taken from
 *                         max|dout|), three fp16 MFMA terms, fp32 accumulation: fp32-grade (gradient parity <= 2e-5). */
int64_t raftcorr_alt_tiles_bytes(int B, int C, int H, int W);
int raftcorr_alt_pack_tiles(const float* fmap, int B, int C, int H, int W, const float* scale, void* packed,
                            int64_t packed_bytes, void* stream);
int64_t raftcorr_alt_lookup_bwd_workspace_bytes(int B, int H, int W, int C, int num_levels, int radius);
int raftcorr_alt_lookup_bwd(const void* tiles1, const void* const* tiles2_levels, const float* scales,
                            int B, int H, int W, int C, int num_levels, int radius,
                            const float* coords, const float* dout,
                            float* dfmap1, float* const* dfmap2_levels, int accumulate,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* raftcorr_alt_lookup_bwd with the query tiles formed by WINDOW position, the backward counterpart of
 * raftcorr_alt_lookup_fwd_sorted (no reference line: the reference's kernel works pixel by pixel).  Same arguments and
 * results as raftcorr_alt_lookup_bwd (to the order of the fp32 sums), except packed1: the FORWARD pass' packed copy of
 * fmap1 (raftcorr_alt_pack) instead of the tile-major one -- the tile-major partner images are gathered per call for
 * the query groups -- and groups: the workspace of raftcorr_alt_lookup_fwd_sorted as that call left it for the SAME
 * coords (B, H, W), or NULL (the groups are then built here: three more small launches).  The groups are a function
 * of the coordinates alone (cells in raster order, members in query order, no atomics), so the results are
 * bit-reproducible like the pixel-tile call's.  A pixel tile's cost grows with the disagreement of its 64 flow
 * vectors; a group's does not: choose it when the flow is rough.
 * workspace: raftcorr_alt_lookup_bwd_sorted_workspace_bytes(...). */
int64_t raftcorr_alt_lookup_bwd_sorted_workspace_bytes(int B, int H, int W, int C, int num_levels, int radius);
int raftcorr_alt_lookup_bwd_sorted(const void* packed1, const void* const* tiles2_levels, const float* scales,
                                   int B, int H, int W, int C, int num_levels, int radius,
                                   const float* coords, const float* dout,
                                   float* dfmap1, float* const* dfmap2_levels, int accumulate,
                                   const void* groups, void* workspace, int64_t workspace_bytes, void* stream);

/* Deferred form of the two backward calls above for the training path, where ONE frame pair is looked up K times
 * (core/raft.py:123-125) before autograd reaches the features: a lookup's backward pass only keeps its (coords, d(out),
 * groups), and this call does the work of n_lookups calls at once -- the kernels run over n_lookups x B "virtual" batch
 * items, which fills the chip where one lookup of one pair does not (KITTI size: 77 instead of 116 us per lookup).
 *   coords, douts    HOST arrays of n_lookups DEVICE pointers, as the single calls take them ((B,2,H,W), (B,L*rd*rd,H,W))
 *   window           0: pixel tiles, fmap1_packed = the tile-major copy (raftcorr_alt_pack_tiles), groups ignored;
 *                    1: window groups, fmap1_packed = the forward pass' copy (raftcorr_alt_pack), groups = HOST array of
 *                       n_lookups DEVICE pointers (the workspaces of raftcorr_alt_lookup_fwd_sorted for the same coords;
 *                       an entry or the whole array may be NULL: those groups are built here)
 *   dfmap1, dfmap2_levels   receive (accumulate != 0: are added to) the SUM over the lookups, added in array order: the
 *                    same float additions in the same order as n_lookups single calls with accumulate = 1, i.e. the
 *                    same bits; no atomics, bit-reproducible
 *   workspace        raftcorr_alt_lookup_bwd_many_workspace_bytes(...) bytes, 256-byte aligned, contents don't-care (it
 *                    holds every lookup's tables, d(out) tiles and gradients: ~80 MB per lookup at KITTI size)
 * n_lookups <= RAFTCORR_ALT_MAX_LOOKUPS per call; longer lists are cut by the caller (accumulate = 1 from the second). */
#define RAFTCORR_ALT_MAX_LOOKUPS 16
int64_t raftcorr_alt_lookup_bwd_many_workspace_bytes(int B, int H, int W, int C, int num_levels, int radius, int n_lookups,
                                                     int window);
int raftcorr_alt_lookup_bwd_many(const void* fmap1_packed, const void* const* tiles2_levels, const float* scales,
                                 int B, int H, int W, int C, int num_levels, int radius, int n_lookups,
                                 const float* const* coords, const float* const* douts, const void* const* groups, int window,
                                 float* dfmap1, float* const* dfmap2_levels, int accumulate,
                                 void* workspace, int64_t workspace_bytes, void* stream);

/* Set-up and tear-down of one frame pair of the on-the-fly block, one host call each (what core/corr.py:64-73 does in
 * its constructor -- the avg_pool2d chain -- plus the packed copies the kernels above read; and its adjoint).
 *   raftcorr_alt_prepare       = raftcorr_alt_scales + the fmap2 pyramid (avg_pool2d(., 2, stride=2) chain, levels
 *                         1 .. L-1; same summation order as torch's kernel) + raftcorr_alt_pack of fmap1 and of every
 *                         level (+ raftcorr_alt_pack_tiles of the fmap2 levels when tiles2_levels is not NULL, and of
 *                         fmap1 when tiles1 is not NULL: the backward calls read them; the window-group calls take
 *                         packed1 instead of tiles1, so tiles1 may be NULL with tiles2_levels given).  packed2_levels / tiles2_levels: HOST arrays of L
 *                         DEVICE buffers of raftcorr_alt_packed_bytes / raftcorr_alt_tiles_bytes(B, C, H>>l, W>>l) bytes;
 *                         workspace: raftcorr_alt_prepare_workspace_bytes(...) bytes, 256-byte aligned (receives the
 *                         pooled float32 levels; don't-care afterwards).
 *   raftcorr_alt_features_bwd  d1 (B,H,W,C), d2_levels[l] (B,H>>l,W>>l,C): the NHWC gradients raftcorr_alt_lookup_bwd
 *                         produces (summed over the lookups of the pair) -> dfmap1, dfmap2 (B,C,H,W) NCHW, fully
 *                         overwritten: dfmap2 = level 0 + the pooled levels passed down the chain as avg_pool2d's
 *                         backward does (t_l = d2_l + t_(l+1) / 4; cells the pooling dropped get nothing). */
int64_t raftcorr_alt_prepare_workspace_bytes(int B, int C, int H, int W, int num_levels);
int raftcorr_alt_prepare(const float* fmap1, const float* fmap2, int B, int C, int H, int W, int num_levels,
                         float* scales, void* packed1, void* const* packed2_levels,
                         void* tiles1, void* const* tiles2_levels,
                         void* workspace, int64_t workspace_bytes, void* stream);
int raftcorr_alt_features_bwd(const float* d1, const float* const* d2_levels, int B, int C, int H, int W, int num_levels,
                              float* dfmap1, float* dfmap2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAFTCORR_H */
