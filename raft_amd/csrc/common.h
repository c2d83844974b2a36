// Internal declarations shared by the kernel files of libraftcorr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/raftcorr.h"
#include "layout.h"

namespace raftcorr {

// thread-local error text behind raftcorr_last_error()
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

struct LevelSet {                 // by-value kernel argument: per-level pointers and geometry
    float* ptr[RAFTCORR_MAX_LEVELS];
    LevelGeom g[RAFTCORR_MAX_LEVELS];
};

// launchers (each validates nothing: the extern "C" layer in api.hip does)
size_t build_workspace_bytes(int B, int D, int H, int W);
size_t build_bwd_workspace_bytes(int B, int D, int H, int W);
int launch_build_pyramid(const float* f1, const float* f2, int B, int D, int H, int W, int L,
                         const LevelSet& lv, void* workspace, size_t workspace_bytes, int precision, hipStream_t s,
                         int vol = RAFTCORR_VOL_FP32);
int launch_build_pyramid_noalloc(const float* f1, const float* f2, int B, int D, int H, int W, int L,
                                 const LevelSet& lv, int precision, hipStream_t s);
int launch_lookup_fwd(const LevelSet& lv, int B, int H, int W, int L, int r, const float* coords,
                      int mode, float* out, hipStream_t s, int vol = RAFTCORR_VOL_FP32);
int launch_lookup_bwd(const LevelSet& lv, const LevelSet& dlv, bool have_dlevels, int B, int H, int W,
                      int L, int r, const float* coords, int mode, const float* dout,
                      float* dcoords, hipStream_t s);
size_t conv1x1_packed_bytes(int Cout, int Cin);
int launch_conv1x1_pack_weight(const float* w, int Cout, int Cin, float* packed, hipStream_t s);
int launch_lookup_conv1x1(const LevelSet& lv, int B, int H, int W, int L, int r, const float* coords,
                          const float* wpacked, const float* bias, int Cout, int relu, float* out, hipStream_t s);
int launch_build_bwd(const float* f1, const float* f2, int B, int D, int H, int W, int L,
                     const LevelSet& dlv, float* df1, float* df2, void* workspace, size_t workspace_bytes,
                     hipStream_t s);
size_t build_bwd_deferred_workspace_bytes(int B, int D, int H, int W);
bool build_bwd_deferred_supported(int H, int W, int L, int r);
int launch_lookup_bwd_record(const float* dout, int B, int N, int C, float* rec, hipStream_t s);
int launch_lookup_bwd_record_many(const float* const* douts, int K, int B, int N, int C, float* const* recs, hipStream_t s);
int launch_build_bwd_deferred(const float* f1, const float* f2, int B, int D, int H, int W, int L, int r, int mode,
                              int K, const float* const* coords, const float* const* records, const LevelSet* dlv,
                              float* df1, float* df2, void* workspace, size_t workspace_bytes, hipStream_t s);
// Query groups by window position ("window tiles"), built without atomics by launch_alt_groups (alt_bwd.hip) and used by
// the forward and the backward pass alike.  Buffer layout: [ngr: B ints][qlist: B x gmax x 64 ints][scratch].
struct AltGroupsPlan {
    int nbx, nby, nb, nch, gmax;
    size_t off_qlist, off_toff, off_cnt, bytes;             // ngr sits at offset 0
};
AltGroupsPlan alt_groups_plan(int B, int H, int W);
int launch_alt_groups(const float* coords, int B, int H, int W, void* groups, size_t groups_bytes, hipStream_t s);

int launch_alt_level_fwd(const float* f1, const float* f2, const float* coords, float* corr,
                         int B, int N, int H1, int W1, int H2, int W2, int C, int r, hipStream_t s);
int launch_alt_level_bwd(const float* f1, const float* f2, const float* coords, const float* cg,
                         float* g1, float* g2, float* gc,
                         int B, int N, int H1, int W1, int H2, int W2, int C, int r, hipStream_t s);
int launch_maxabs(const float* f1, const float* f2, size_t n, float* out, hipStream_t s);
size_t alt_packed_bytes(int B, int C, int H, int W);
int launch_alt_pack_set(const float* const* fmaps, const int* hs, const int* ws, const float* const* maxparts,
                        void* const* packed, int n, int B, int C, hipStream_t s);
int launch_alt_pack_tiles_set(const float* const* fmaps, const int* hs, const int* ws, const float* const* maxparts,
                              void* const* packed, int n, int B, int C, hipStream_t s);
int launch_alt_pack(const float* fmap, int B, int C, int H, int W, const float* maxpart, void* packed, hipStream_t s);
int launch_alt_lookup_fwd16(const void* f1p, const void* const* f2p, const float* maxpart, int B, int H, int W, int C, int L,
                            int r, const float* coords, float* out, void* workspace, size_t workspace_bytes, hipStream_t s);
size_t alt_lookup_fwd_sorted_workspace_bytes(int B, int H, int W, int L);
int launch_maxabs1(const float* x, size_t n, float* out, hipStream_t s);
int launch_maxabs3(const float* f1, const float* f2, size_t n, const float* f3, size_t n3, float* out, hipStream_t s);
size_t alt_tiles_bytes(int B, int C, int H, int W);
int launch_alt_pack_tiles(const float* fmap, int B, int C, int H, int W, const float* maxpart, void* packed, hipStream_t s);
size_t alt_lookup_bwd_workspace_bytes(int B, int H, int W, int C, int L, int r, bool window = false, int nk = 1);
int launch_alt_lookup_bwd16_many(const void* f1, const void* const* f2t, const float* maxpart, int B, int H, int W, int C, int L,
                                 int r, int nk, const float* const* coords, const float* const* douts, const void* const* groups,
                                 bool window, float* df1, float* const* df2, int accumulate,
                                 void* workspace, size_t workspace_bytes, hipStream_t s);
size_t alt_prepare_workspace_bytes(int B, int C, int H, int W, int L);
int launch_alt_prepare(const float* fmap1, const float* fmap2, int B, int C, int H, int W, int L, float* scales,
                       void* packed1, void* const* packed2, void* tiles1, void* const* tiles2,
                       void* workspace, size_t workspace_bytes, hipStream_t s);
int launch_alt_features_bwd(const float* d1, const float* const* d2, int B, int C, int H, int W, int L,
                            float* df1, float* df2, hipStream_t s);
int launch_alt_lookup_bwd16(const void* f1t, const void* const* f2t, const float* maxpart, int B, int H, int W, int C, int L,
                            int r, const float* coords, const float* dout, float* df1, float* const* df2, int accumulate,
                            void* workspace, size_t workspace_bytes, hipStream_t s, const void* f1p_window = nullptr,
                            const void* groups = nullptr);

// floor + fraction of a sampling coordinate, memory-safe for NaN / huge values: anything
// that is not a sane pixel coordinate is pushed far outside the map (all taps read as zero).
__device__ __forceinline__ void split_coord(float x, int& x0, float& fx) {
    if (!(x > -1.0e6f && x < 1.0e6f)) { x0 = -(1 << 24); fx = 0.f; return; }
    float f = floorf(x);
    x0 = (int)f;
    fx = x - f;
}

}  // namespace raftcorr
