// extern "C" surface of libraftcorr.so: argument validation, error text, dispatch.
// Declarations and the reference interface each entry point replaces: include/raftcorr.h.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

namespace raftcorr {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RAFTCORR_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return RAFTCORR_OK;
}

// levels: host array of device pointers -> by-value LevelSet.  volume = true: pyramid levels in the
// patch-tile order of layout.h; false: plain (h, w) feature-map levels (on-the-fly path).
static int make_levels(const char* who, const float* const* ptrs, int H, int W, int L, LevelSet& lv,
                       bool volume = true)
{
    if (L < 1 || L > RAFTCORR_MAX_LEVELS)
        return fail(RAFTCORR_EINVAL, "%s: num_levels = %d not in [1, %d]", who, L, RAFTCORR_MAX_LEVELS);
    if (!ptrs) return fail(RAFTCORR_EINVAL, "%s: null level pointer array", who);
    for (int l = 0; l < RAFTCORR_MAX_LEVELS; ++l) { lv.ptr[l] = nullptr; lv.g[l] = LevelGeom{0, 0, 0, 0, 0, 4, 0}; }
    for (int l = 0; l < L; ++l) {
        if (volume) lv.g[l] = level_geom(H, W, l);
        else { lv.g[l].h = lv.g[l].hp = H >> l; lv.g[l].w = lv.g[l].wp = W >> l; lv.g[l].pw = 0; lv.g[l].kind = 4;
               lv.g[l].slab = (int64_t)(H >> l) * (W >> l); }
        lv.ptr[l] = const_cast<float*>(ptrs[l]);
        if (lv.g[l].h < 1 || lv.g[l].w < 1)      // the reference's avg_pool2d raises "Output size is too small" here
            return fail(RAFTCORR_EINVAL, "%s: level %d of a %d x %d map is empty", who, l, H, W);
        if (!ptrs[l]) return fail(RAFTCORR_EINVAL, "%s: level %d pointer is null", who, l);
    }
    return RAFTCORR_OK;
}

static int check_dims(const char* who, int B, int H, int W)
{
    if (B < 1 || H < 1 || W < 1) return fail(RAFTCORR_EINVAL, "%s: B, H, W must be positive (got %d, %d, %d)", who, B, H, W);
    if (B > 65535) return fail(RAFTCORR_EUNSUPPORTED, "%s: B = %d > 65535", who, B);
    if ((long long)H * W > (1LL << 24)) return fail(RAFTCORR_EUNSUPPORTED, "%s: H*W = %lld too large", who, (long long)H * W);
    return RAFTCORR_OK;
}

}  // namespace raftcorr

using namespace raftcorr;

extern "C" {

int raftcorr_version(void) { return RAFTCORR_VERSION; }

const char* raftcorr_last_error(void) { return g_err; }

int raftcorr_level_layout(int H, int W, int num_levels, int32_t* hl, int32_t* wl, int64_t* slab)
{
    if (H < 1 || W < 1 || num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS || !hl || !wl || !slab)
        return fail(RAFTCORR_EINVAL, "level_layout: bad argument");
    for (int l = 0; l < num_levels; ++l) {
        const LevelGeom g = level_geom(H, W, l);
        hl[l] = g.h; wl[l] = g.w; slab[l] = g.slab;
    }
    return RAFTCORR_OK;
}

int raftcorr_level_positions(int H, int W, int level, int32_t* pos)
{
    if (H < 1 || W < 1 || level < 0 || level >= RAFTCORR_MAX_LEVELS || !pos)
        return fail(RAFTCORR_EINVAL, "level_positions: bad argument");
    const LevelGeom g = level_geom(H, W, level);
    for (int y = 0; y < g.h; ++y)
        for (int x = 0; x < g.w; ++x) pos[y * g.w + x] = level_pos(g, y, x);
    return RAFTCORR_OK;
}

int64_t raftcorr_build_workspace_bytes(int B, int D, int H, int W)
{
    if (B < 1 || D < 1 || H < 1 || W < 1) return 0;
    return (int64_t)build_workspace_bytes(B, D, H, W);
}

int64_t raftcorr_build_bwd_workspace_bytes(int B, int D, int H, int W)
{
    if (B < 1 || D < 1 || H < 1 || W < 1) return 0;
    return (int64_t)build_bwd_workspace_bytes(B, D, H, W);
}

static int build_pyramid_any(const char* who, const float* fmap1, const float* fmap2, int B, int D, int H, int W,
                             int num_levels, float* const* levels, void* workspace, int64_t workspace_bytes,
                             int precision, void* stream, int vol = RAFTCORR_VOL_FP32)
{
    if (vol < RAFTCORR_VOL_FP32 || vol > RAFTCORR_VOL_BF16) return fail(RAFTCORR_EINVAL, "%s: volume_dtype = %d", who, vol);
    int rc = check_dims(who, B, H, W);
    if (rc) return rc;
    if (!fmap1 || !fmap2) return fail(RAFTCORR_EINVAL, "%s: null feature map", who);
    if (D < 1) return fail(RAFTCORR_EINVAL, "%s: D = %d", who, D);
    if (precision < RAFTCORR_PREC_FP32 || precision > RAFTCORR_PREC_FP16X2)
        return fail(RAFTCORR_EINVAL, "%s: precision = %d", who, precision);
    LevelSet lv;
    rc = make_levels(who, levels, H, W, num_levels, lv);
    if (rc) return rc;
    return launch_build_pyramid(fmap1, fmap2, B, D, H, W, num_levels, lv, workspace,
                                workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, precision, (hipStream_t)stream, vol);
}

int raftcorr_build_pyramid(const float* fmap1, const float* fmap2, int B, int D, int H, int W,
                           int num_levels, float* const* levels, void* workspace, int64_t workspace_bytes,
                           void* stream)
{
    return build_pyramid_any("build_pyramid", fmap1, fmap2, B, D, H, W, num_levels, levels, workspace,
                             workspace_bytes, RAFTCORR_PREC_FP32, stream);
}

int raftcorr_build_pyramid_prec(const float* fmap1, const float* fmap2, int B, int D, int H, int W,
                                int num_levels, float* const* levels, void* workspace, int64_t workspace_bytes,
                                int precision, void* stream)
{
    return build_pyramid_any("build_pyramid_prec", fmap1, fmap2, B, D, H, W, num_levels, levels, workspace,
                             workspace_bytes, precision, stream);
}

int raftcorr_build_pyramid_vol(const float* fmap1, const float* fmap2, int B, int D, int H, int W,
                               int num_levels, void* const* levels, void* workspace, int64_t workspace_bytes,
                               int precision, int volume_dtype, void* stream)
{
    return build_pyramid_any("build_pyramid_vol", fmap1, fmap2, B, D, H, W, num_levels, (float* const*)levels, workspace,
                             workspace_bytes, precision, stream, volume_dtype);
}

int raftcorr_lookup_fwd_vol(const void* const* levels, int volume_dtype, int B, int H, int W, int num_levels, int radius,
                            const float* coords, int mode, float* out, void* stream)
{
    int rc = check_dims("lookup_fwd_vol", B, H, W);
    if (rc) return rc;
    if (!coords || !out) return fail(RAFTCORR_EINVAL, "lookup_fwd_vol: null coords/out");
    if (radius < 0) return fail(RAFTCORR_EINVAL, "lookup_fwd_vol: radius = %d", radius);
    if (mode != RAFTCORR_MODE_FORK && mode != RAFTCORR_MODE_CANONICAL)
        return fail(RAFTCORR_EINVAL, "lookup_fwd_vol: mode = %d", mode);
    if (volume_dtype < RAFTCORR_VOL_FP32 || volume_dtype > RAFTCORR_VOL_BF16)
        return fail(RAFTCORR_EINVAL, "lookup_fwd_vol: volume_dtype = %d", volume_dtype);
    if (volume_dtype != RAFTCORR_VOL_FP32 && num_levels > 4)
        return fail(RAFTCORR_EUNSUPPORTED, "lookup_fwd_vol: a 16-bit volume has at most 4 levels (got %d)", num_levels);
    LevelSet lv;
    rc = make_levels("lookup_fwd_vol", (const float* const*)levels, H, W, num_levels, lv);
    if (rc) return rc;
    return launch_lookup_fwd(lv, B, H, W, num_levels, radius, coords, mode, out, (hipStream_t)stream, volume_dtype);
}

int raftcorr_build_pyramid_noalloc(const float* fmap1, const float* fmap2, int B, int D, int H, int W,
                                   int num_levels, float* const* levels, int precision, void* stream)
{
    const char* who = "build_pyramid_noalloc";
    int rc = check_dims(who, B, H, W);
    if (rc) return rc;
    if (!fmap1 || !fmap2) return fail(RAFTCORR_EINVAL, "%s: null feature map", who);
    if (D < 1) return fail(RAFTCORR_EINVAL, "%s: D = %d", who, D);
    if (precision < RAFTCORR_PREC_FP32 || precision > RAFTCORR_PREC_FP16X2)
        return fail(RAFTCORR_EINVAL, "%s: precision = %d", who, precision);
    LevelSet lv;
    rc = make_levels(who, levels, H, W, num_levels, lv);
    if (rc) return rc;
    return launch_build_pyramid_noalloc(fmap1, fmap2, B, D, H, W, num_levels, lv, precision, (hipStream_t)stream);
}

int raftcorr_lookup_fwd(const float* const* levels, int B, int H, int W, int num_levels, int radius,
                        const float* coords, int mode, float* out, void* stream)
{
    int rc = check_dims("lookup_fwd", B, H, W);
    if (rc) return rc;
    if (!coords || !out) return fail(RAFTCORR_EINVAL, "lookup_fwd: null coords/out");
    if (radius < 0) return fail(RAFTCORR_EINVAL, "lookup_fwd: radius = %d", radius);
    if (mode != RAFTCORR_MODE_FORK && mode != RAFTCORR_MODE_CANONICAL)
        return fail(RAFTCORR_EINVAL, "lookup_fwd: mode = %d", mode);
    LevelSet lv;
    rc = make_levels("lookup_fwd", levels, H, W, num_levels, lv);
    if (rc) return rc;
    return launch_lookup_fwd(lv, B, H, W, num_levels, radius, coords, mode, out, (hipStream_t)stream);
}

int64_t raftcorr_conv1x1_packed_bytes(int Cout, int Cin)
{
    if (Cout < 1 || Cin < 1) return 0;
    return (int64_t)conv1x1_packed_bytes(Cout, Cin);
}

int raftcorr_conv1x1_pack_weight(const float* weight, int Cout, int Cin, float* packed, int64_t packed_bytes,
                                 void* stream)
{
    if (!weight || !packed) return fail(RAFTCORR_EINVAL, "conv1x1_pack_weight: null pointer");
    if (Cout < 1 || Cin < 1) return fail(RAFTCORR_EINVAL, "conv1x1_pack_weight: Cout=%d Cin=%d", Cout, Cin);
    if (packed_bytes < (int64_t)conv1x1_packed_bytes(Cout, Cin))
        return fail(RAFTCORR_EINVAL, "conv1x1_pack_weight: packed buffer too small");
    return launch_conv1x1_pack_weight(weight, Cout, Cin, packed, (hipStream_t)stream);
}

int raftcorr_lookup_conv1x1_fwd(const float* const* levels, int B, int H, int W, int num_levels, int radius,
                                const float* coords, const float* packed_weight, const float* bias, int Cout,
                                int relu, float* out, void* stream)
{
    int rc = check_dims("lookup_conv1x1_fwd", B, H, W);
    if (rc) return rc;
    if (!coords || !out || !packed_weight) return fail(RAFTCORR_EINVAL, "lookup_conv1x1_fwd: null pointer");
    if (radius < 0 || Cout < 1) return fail(RAFTCORR_EINVAL, "lookup_conv1x1_fwd: radius=%d Cout=%d", radius, Cout);
    LevelSet lv;
    rc = make_levels("lookup_conv1x1_fwd", levels, H, W, num_levels, lv);
    if (rc) return rc;
    return launch_lookup_conv1x1(lv, B, H, W, num_levels, radius, coords, packed_weight, bias, Cout, relu, out,
                                 (hipStream_t)stream);
}

int raftcorr_lookup_bwd(const float* const* levels, float* const* dlevels, int B, int H, int W,
                        int num_levels, int radius, const float* coords, int mode, const float* dout,
                        float* dcoords, void* stream)
{
    int rc = check_dims("lookup_bwd", B, H, W);
    if (rc) return rc;
    if (!coords || !dout) return fail(RAFTCORR_EINVAL, "lookup_bwd: null coords/dout");
    if (radius < 0) return fail(RAFTCORR_EINVAL, "lookup_bwd: radius = %d", radius);
    if (mode != RAFTCORR_MODE_FORK && mode != RAFTCORR_MODE_CANONICAL)
        return fail(RAFTCORR_EINVAL, "lookup_bwd: mode = %d", mode);
    LevelSet lv, dlv;
    const bool have = levels != nullptr, have_d = dlevels != nullptr;
    if (!have_d && !dcoords) return fail(RAFTCORR_EINVAL, "lookup_bwd: neither dlevels nor dcoords given");
    if (dcoords && !have) return fail(RAFTCORR_EINVAL, "lookup_bwd: dcoords needs levels");
    if (have_d) {
        rc = make_levels("lookup_bwd(dlevels)", dlevels, H, W, num_levels, dlv);
        if (rc) return rc;
    }
    if (have) {
        rc = make_levels("lookup_bwd(levels)", levels, H, W, num_levels, lv);
        if (rc) return rc;
    }
    if (!have) lv = dlv;
    if (!have_d) { dlv = lv; for (int l = 0; l < RAFTCORR_MAX_LEVELS; ++l) dlv.ptr[l] = nullptr; }
    return launch_lookup_bwd(lv, dlv, have_d, B, H, W, num_levels, radius, coords, mode, dout, dcoords,
                             (hipStream_t)stream);
}

int raftcorr_build_bwd(const float* fmap1, const float* fmap2, int B, int D, int H, int W,
                       int num_levels, float* const* dlevels, float* dfmap1, float* dfmap2,
                       void* workspace, int64_t workspace_bytes, void* stream)
{
    int rc = check_dims("build_bwd", B, H, W);
    if (rc) return rc;
    if (!fmap1 || !fmap2 || !dfmap1 || !dfmap2) return fail(RAFTCORR_EINVAL, "build_bwd: null pointer");
    if (D < 1) return fail(RAFTCORR_EINVAL, "build_bwd: D = %d", D);
    LevelSet dlv;
    rc = make_levels("build_bwd", dlevels, H, W, num_levels, dlv);
    if (rc) return rc;
    return launch_build_bwd(fmap1, fmap2, B, D, H, W, num_levels, dlv, dfmap1, dfmap2, workspace,
                            workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, (hipStream_t)stream);
}

int64_t raftcorr_build_bwd_deferred_workspace_bytes(int B, int D, int H, int W)
{
    if (B < 1 || D < 1 || H < 1 || W < 1) return 0;
    return (int64_t)build_bwd_deferred_workspace_bytes(B, D, H, W);
}

int raftcorr_build_bwd_deferred_supported(int H, int W, int num_levels, int radius)
{
    return build_bwd_deferred_supported(H, W, num_levels, radius) ? 1 : 0;
}

int raftcorr_lookup_bwd_record(const float* dout, int B, int H, int W, int channels, float* record, void* stream)
{
    int rc = check_dims("lookup_bwd_record", B, H, W);
    if (rc) return rc;
    if (!dout || !record) return fail(RAFTCORR_EINVAL, "lookup_bwd_record: null pointer");
    if (channels < 1 || channels > 65535 * 256) return fail(RAFTCORR_EINVAL, "lookup_bwd_record: channels = %d", channels);
    return launch_lookup_bwd_record(dout, B, H * W, channels, record, (hipStream_t)stream);
}

int raftcorr_lookup_bwd_record_many(const float* const* douts, int n_lookups, int B, int H, int W, int channels,
                                    float* const* records, void* stream)
{
    int rc = check_dims("lookup_bwd_record_many", B, H, W);
    if (rc) return rc;
    if (n_lookups < 1 || n_lookups > 4096) return fail(RAFTCORR_EINVAL, "lookup_bwd_record_many: %d lookups", n_lookups);
    if (!douts || !records) return fail(RAFTCORR_EINVAL, "lookup_bwd_record_many: null pointer");
    if (channels < 1 || channels > 65535 * 256) return fail(RAFTCORR_EINVAL, "lookup_bwd_record_many: channels = %d", channels);
    for (int k = 0; k < n_lookups; ++k)
        if (!douts[k] || !records[k]) return fail(RAFTCORR_EINVAL, "lookup_bwd_record_many: lookup %d: null pointer", k);
    return launch_lookup_bwd_record_many(douts, n_lookups, B, H * W, channels, records, (hipStream_t)stream);
}

int raftcorr_build_bwd_deferred(const float* fmap1, const float* fmap2, int B, int D, int H, int W, int num_levels,
                                int radius, int mode, int n_lookups, const float* const* coords,
                                const float* const* records, float* const* dlevels, float* dfmap1, float* dfmap2,
                                void* workspace, int64_t workspace_bytes, void* stream)
{
    int rc = check_dims("build_bwd_deferred", B, H, W);
    if (rc) return rc;
    if (!fmap1 || !fmap2 || !dfmap1 || !dfmap2) return fail(RAFTCORR_EINVAL, "build_bwd_deferred: null pointer");
    if (D < 1) return fail(RAFTCORR_EINVAL, "build_bwd_deferred: D = %d", D);
    if (mode != RAFTCORR_MODE_FORK && mode != RAFTCORR_MODE_CANONICAL)
        return fail(RAFTCORR_EINVAL, "build_bwd_deferred: mode = %d", mode);
    if (n_lookups < 0 || (n_lookups > 0 && (!coords || !records)))
        return fail(RAFTCORR_EINVAL, "build_bwd_deferred: %d lookups, null coords/records", n_lookups);
    for (int k = 0; k < n_lookups; ++k)
        if (!coords[k] || !records[k] || ((uintptr_t)records[k] & 3)) return fail(RAFTCORR_EINVAL, "build_bwd_deferred: lookup %d: null pointer", k);
    if (!build_bwd_deferred_supported(H, W, num_levels, radius))
        return fail(RAFTCORR_EUNSUPPORTED, "build_bwd_deferred: %d x %d, %d levels, radius %d is not supported "
                    "(raftcorr_build_bwd_deferred_supported); use raftcorr_lookup_bwd + raftcorr_build_bwd", H, W, num_levels, radius);
    LevelSet dlv;
    if (dlevels) {
        rc = make_levels("build_bwd_deferred", dlevels, H, W, num_levels, dlv);
        if (rc) return rc;
    }
    return launch_build_bwd_deferred(fmap1, fmap2, B, D, H, W, num_levels, radius, mode, n_lookups, coords, records,
                                     dlevels ? &dlv : nullptr, dfmap1, dfmap2, workspace,
                                     workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, (hipStream_t)stream);
}

static int alt_level_check(const char* who, int B, int N, int H1, int W1, int H2, int W2, int C, int r)
{
    int rc = check_dims(who, B, H1, W1);
    if (rc) return rc;
    if (N < 1 || H2 < 1 || W2 < 1 || C < 1 || r < 0)
        return fail(RAFTCORR_EINVAL, "%s: bad size (N=%d H2=%d W2=%d C=%d r=%d)", who, N, H2, W2, C, r);
    if ((long long)B * N > 65535) return fail(RAFTCORR_EUNSUPPORTED, "%s: B*N > 65535", who);
    return RAFTCORR_OK;
}

int raftcorr_alt_forward(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                         int B, int N, int H1, int W1, int H2, int W2, int C, int radius, void* stream)
{
    int rc = alt_level_check("alt_forward", B, N, H1, W1, H2, W2, C, radius);
    if (rc) return rc;
    if (!fmap1 || !fmap2 || !coords || !corr) return fail(RAFTCORR_EINVAL, "alt_forward: null pointer");
    return launch_alt_level_fwd(fmap1, fmap2, coords, corr, B, N, H1, W1, H2, W2, C, radius, (hipStream_t)stream);
}

int raftcorr_alt_backward(const float* fmap1, const float* fmap2, const float* coords,
                          const float* corr_grad, float* fmap1_grad, float* fmap2_grad, float* coords_grad,
                          int B, int N, int H1, int W1, int H2, int W2, int C, int radius, void* stream)
{
    int rc = alt_level_check("alt_backward", B, N, H1, W1, H2, W2, C, radius);
    if (rc) return rc;
    if (!fmap1 || !fmap2 || !coords || !corr_grad || !fmap1_grad || !fmap2_grad)
        return fail(RAFTCORR_EINVAL, "alt_backward: null pointer");
    return launch_alt_level_bwd(fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, coords_grad,
                                B, N, H1, W1, H2, W2, C, radius, (hipStream_t)stream);
}

int64_t raftcorr_alt_packed_bytes(int B, int C, int H, int W)
{
    if (B < 1 || C < 1 || H < 1 || W < 1) return 0;
    return (int64_t)alt_packed_bytes(B, C, H, W);
}

int raftcorr_alt_scales(const float* fmap1, const float* fmap2, int B, int C, int H, int W, float* scales, void* stream)
{
    int rc = check_dims("alt_scales", B, H, W);
    if (rc) return rc;
    if (!fmap1 || !fmap2 || !scales || C < 1) return fail(RAFTCORR_EINVAL, "alt_scales: bad argument");
    return launch_maxabs(fmap1, fmap2, (size_t)B * C * H * W, scales, (hipStream_t)stream);
}

int raftcorr_alt_pack(const float* fmap, int B, int C, int H, int W, const float* scale, void* packed, int64_t packed_bytes,
                      void* stream)
{
    int rc = check_dims("alt_pack", B, H, W);
    if (rc) return rc;
    if (!fmap || !scale || !packed || C < 1) return fail(RAFTCORR_EINVAL, "alt_pack: bad argument");
    if (packed_bytes < (int64_t)alt_packed_bytes(B, C, H, W))
        return fail(RAFTCORR_EINVAL, "alt_pack: packed buffer too small (%lld < %zu bytes)", (long long)packed_bytes,
                    alt_packed_bytes(B, C, H, W));
    return launch_alt_pack(fmap, B, C, H, W, scale, packed, (hipStream_t)stream);
}

int raftcorr_alt_lookup_fwd(const void* packed1, const void* const* packed2_levels, const float* scales,
                            int B, int H, int W, int C, int num_levels, int radius, const float* coords, float* out,
                            void* stream)
{
    int rc = check_dims("alt_lookup_fwd", B, H, W);
    if (rc) return rc;
    if (!packed1 || !packed2_levels || !scales || !coords || !out) return fail(RAFTCORR_EINVAL, "alt_lookup_fwd: null pointer");
    if (C < 1 || radius < 0) return fail(RAFTCORR_EINVAL, "alt_lookup_fwd: C=%d radius=%d", C, radius);
    if (num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS)
        return fail(RAFTCORR_EINVAL, "alt_lookup_fwd: num_levels = %d not in [1, %d]", num_levels, RAFTCORR_MAX_LEVELS);
    for (int l = 0; l < num_levels; ++l) {
        if ((H >> l) < 1 || (W >> l) < 1) return fail(RAFTCORR_EINVAL, "alt_lookup_fwd: level %d of a %d x %d map is empty", l, H, W);
        if (!packed2_levels[l]) return fail(RAFTCORR_EINVAL, "alt_lookup_fwd: level %d pointer is null", l);
    }
    return launch_alt_lookup_fwd16(packed1, packed2_levels, scales, B, H, W, C, num_levels, radius, coords, out,
                                   nullptr, 0, (hipStream_t)stream);
}

int64_t raftcorr_alt_lookup_fwd_sorted_workspace_bytes(int B, int H, int W, int num_levels)
{
    if (B < 1 || H < 1 || W < 1 || num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS) return 0;
    return (int64_t)alt_lookup_fwd_sorted_workspace_bytes(B, H, W, num_levels);
}

int raftcorr_alt_lookup_fwd_sorted(const void* packed1, const void* const* packed2_levels, const float* scales,
                                   int B, int H, int W, int C, int num_levels, int radius, const float* coords, float* out,
                                   void* workspace, int64_t workspace_bytes, void* stream)
{
    int rc = check_dims("alt_lookup_fwd_sorted", B, H, W);
    if (rc) return rc;
    if (!packed1 || !packed2_levels || !scales || !coords || !out || !workspace)
        return fail(RAFTCORR_EINVAL, "alt_lookup_fwd_sorted: null pointer");
    if (C < 1 || radius < 0) return fail(RAFTCORR_EINVAL, "alt_lookup_fwd_sorted: C=%d radius=%d", C, radius);
    if (num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS)
        return fail(RAFTCORR_EINVAL, "alt_lookup_fwd_sorted: num_levels = %d not in [1, %d]", num_levels, RAFTCORR_MAX_LEVELS);
    for (int l = 0; l < num_levels; ++l) {
        if ((H >> l) < 1 || (W >> l) < 1)
            return fail(RAFTCORR_EINVAL, "alt_lookup_fwd_sorted: level %d of a %d x %d map is empty", l, H, W);
        if (!packed2_levels[l]) return fail(RAFTCORR_EINVAL, "alt_lookup_fwd_sorted: level %d pointer is null", l);
    }
    return launch_alt_lookup_fwd16(packed1, packed2_levels, scales, B, H, W, C, num_levels, radius, coords, out,
                                   workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, (hipStream_t)stream);
}

int64_t raftcorr_alt_tiles_bytes(int B, int C, int H, int W)
{
    if (B < 1 || C < 1 || H < 1 || W < 1) return 0;
    return (int64_t)alt_tiles_bytes(B, C, H, W);
}

int raftcorr_alt_pack_tiles(const float* fmap, int B, int C, int H, int W, const float* scale, void* packed,
                            int64_t packed_bytes, void* stream)
{
    int rc = check_dims("alt_pack_tiles", B, H, W);
    if (rc) return rc;
    if (!fmap || !scale || !packed || C < 1) return fail(RAFTCORR_EINVAL, "alt_pack_tiles: bad argument");
    if (packed_bytes < (int64_t)alt_tiles_bytes(B, C, H, W))
        return fail(RAFTCORR_EINVAL, "alt_pack_tiles: packed buffer too small (%lld < %zu bytes)", (long long)packed_bytes,
                    alt_tiles_bytes(B, C, H, W));
    return launch_alt_pack_tiles(fmap, B, C, H, W, scale, packed, (hipStream_t)stream);
}

int64_t raftcorr_alt_lookup_bwd_workspace_bytes(int B, int H, int W, int C, int num_levels, int radius)
{
    if (B < 1 || C < 1 || H < 1 || W < 1 || num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS || radius < 0) return 0;
    return (int64_t)alt_lookup_bwd_workspace_bytes(B, H, W, C, num_levels, radius);
}

int raftcorr_alt_lookup_bwd(const void* tiles1, const void* const* tiles2_levels, const float* scales,
                            int B, int H, int W, int C, int num_levels, int radius,
                            const float* coords, const float* dout, float* dfmap1, float* const* dfmap2_levels,
                            int accumulate, void* workspace, int64_t workspace_bytes, void* stream)
{
    int rc = check_dims("alt_lookup_bwd", B, H, W);
    if (rc) return rc;
    if (!tiles1 || !tiles2_levels || !scales || !coords || !dout || !dfmap1 || !dfmap2_levels)
        return fail(RAFTCORR_EINVAL, "alt_lookup_bwd: null pointer");
    if (C < 1 || radius < 0) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd: C=%d radius=%d", C, radius);
    if (num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS)
        return fail(RAFTCORR_EINVAL, "alt_lookup_bwd: num_levels = %d not in [1, %d]", num_levels, RAFTCORR_MAX_LEVELS);
    for (int l = 0; l < num_levels; ++l) {
        if ((H >> l) < 1 || (W >> l) < 1) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd: level %d of a %d x %d map is empty", l, H, W);
        if (!tiles2_levels[l] || !dfmap2_levels[l]) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd: level %d pointer is null", l);
    }
    return launch_alt_lookup_bwd16(tiles1, tiles2_levels, scales, B, H, W, C, num_levels, radius, coords, dout, dfmap1,
                                   dfmap2_levels, accumulate, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes,
                                   (hipStream_t)stream);
}

int64_t raftcorr_alt_lookup_bwd_sorted_workspace_bytes(int B, int H, int W, int C, int num_levels, int radius)
{
    if (B < 1 || H < 1 || W < 1 || C < 1 || num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS || radius < 0) return 0;
    return (int64_t)alt_lookup_bwd_workspace_bytes(B, H, W, C, num_levels, radius, true);
}

int raftcorr_alt_lookup_bwd_sorted(const void* packed1, const void* const* tiles2_levels, const float* scales,
                                   int B, int H, int W, int C, int num_levels, int radius,
                                   const float* coords, const float* dout, float* dfmap1, float* const* dfmap2_levels,
                                   int accumulate, const void* groups, void* workspace, int64_t workspace_bytes, void* stream)
{
    int rc = check_dims("alt_lookup_bwd_sorted", B, H, W);
    if (rc) return rc;
    if (!packed1 || !tiles2_levels || !scales || !coords || !dout || !dfmap1 || !dfmap2_levels)
        return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_sorted: null pointer");
    if (C < 1 || radius < 0) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_sorted: C=%d radius=%d", C, radius);
    if (num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS)
        return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_sorted: num_levels = %d not in [1, %d]", num_levels, RAFTCORR_MAX_LEVELS);
    for (int l = 0; l < num_levels; ++l) {
        if ((H >> l) < 1 || (W >> l) < 1) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_sorted: level %d of a %d x %d map is empty", l, H, W);
        if (!tiles2_levels[l] || !dfmap2_levels[l]) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_sorted: level %d pointer is null", l);
    }
    return launch_alt_lookup_bwd16(nullptr, tiles2_levels, scales, B, H, W, C, num_levels, radius, coords, dout, dfmap1,
                                   dfmap2_levels, accumulate, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes,
                                   (hipStream_t)stream, packed1, groups);
}

int64_t raftcorr_alt_lookup_bwd_many_workspace_bytes(int B, int H, int W, int C, int num_levels, int radius, int n_lookups,
                                                     int window)
{
    if (B < 1 || H < 1 || W < 1 || C < 1 || num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS || radius < 0 ||
        n_lookups < 1 || n_lookups > RAFTCORR_ALT_MAX_LOOKUPS) return 0;
    return (int64_t)alt_lookup_bwd_workspace_bytes(B, H, W, C, num_levels, radius, window != 0, n_lookups);
}

int raftcorr_alt_lookup_bwd_many(const void* fmap1_packed, const void* const* tiles2_levels, const float* scales,
                                 int B, int H, int W, int C, int num_levels, int radius, int n_lookups,
                                 const float* const* coords, const float* const* douts, const void* const* groups, int window,
                                 float* dfmap1, float* const* dfmap2_levels, int accumulate,
                                 void* workspace, int64_t workspace_bytes, void* stream)
{
    int rc = check_dims("alt_lookup_bwd_many", B, H, W);
    if (rc) return rc;
    if (!fmap1_packed || !tiles2_levels || !scales || !coords || !douts || !dfmap1 || !dfmap2_levels)
        return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_many: null pointer");
    if (C < 1 || radius < 0) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_many: C=%d radius=%d", C, radius);
    if (num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS)
        return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_many: num_levels = %d not in [1, %d]", num_levels, RAFTCORR_MAX_LEVELS);
    if (n_lookups < 1 || n_lookups > RAFTCORR_ALT_MAX_LOOKUPS)
        return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_many: n_lookups = %d not in [1, %d]", n_lookups, RAFTCORR_ALT_MAX_LOOKUPS);
    for (int l = 0; l < num_levels; ++l) {
        if ((H >> l) < 1 || (W >> l) < 1) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_many: level %d of a %d x %d map is empty", l, H, W);
        if (!tiles2_levels[l] || !dfmap2_levels[l]) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_many: level %d pointer is null", l);
    }
    for (int k = 0; k < n_lookups; ++k)
        if (!coords[k] || !douts[k]) return fail(RAFTCORR_EINVAL, "alt_lookup_bwd_many: coords / dout of lookup %d is null", k);
    return launch_alt_lookup_bwd16_many(fmap1_packed, tiles2_levels, scales, B, H, W, C, num_levels, radius, n_lookups, coords, douts,
                                        window ? groups : nullptr, window != 0, dfmap1, dfmap2_levels, accumulate, workspace,
                                        workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, (hipStream_t)stream);
}

int64_t raftcorr_alt_prepare_workspace_bytes(int B, int C, int H, int W, int num_levels)
{
    if (B < 1 || C < 1 || H < 1 || W < 1 || num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS) return 0;
    return (int64_t)alt_prepare_workspace_bytes(B, C, H, W, num_levels);
}

int raftcorr_alt_prepare(const float* fmap1, const float* fmap2, int B, int C, int H, int W, int num_levels,
                         float* scales, void* packed1, void* const* packed2_levels,
                         void* tiles1, void* const* tiles2_levels,
                         void* workspace, int64_t workspace_bytes, void* stream)
{
    int rc = check_dims("alt_prepare", B, H, W);
    if (rc) return rc;
    if (!fmap1 || !fmap2 || !scales || !packed1 || !packed2_levels || C < 1)
        return fail(RAFTCORR_EINVAL, "alt_prepare: bad argument");
    if (tiles1 && !tiles2_levels)
        return fail(RAFTCORR_EINVAL, "alt_prepare: tiles1 without tiles2_levels");
    if (num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS)
        return fail(RAFTCORR_EINVAL, "alt_prepare: num_levels = %d not in [1, %d]", num_levels, RAFTCORR_MAX_LEVELS);
    for (int l = 0; l < num_levels; ++l) {
        if ((H >> l) < 1 || (W >> l) < 1) return fail(RAFTCORR_EINVAL, "alt_prepare: level %d of a %d x %d map is empty", l, H, W);
        if (!packed2_levels[l] || (tiles2_levels && !tiles2_levels[l]))
            return fail(RAFTCORR_EINVAL, "alt_prepare: level %d pointer is null", l);
    }
    return launch_alt_prepare(fmap1, fmap2, B, C, H, W, num_levels, scales, packed1, packed2_levels, tiles1, tiles2_levels,
                              workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, (hipStream_t)stream);
}

int raftcorr_alt_features_bwd(const float* d1, const float* const* d2_levels, int B, int C, int H, int W, int num_levels,
                              float* dfmap1, float* dfmap2, void* stream)
{
    int rc = check_dims("alt_features_bwd", B, H, W);
    if (rc) return rc;
    if (!d1 || !d2_levels || !dfmap1 || !dfmap2 || C < 1) return fail(RAFTCORR_EINVAL, "alt_features_bwd: bad argument");
    if (num_levels < 1 || num_levels > RAFTCORR_MAX_LEVELS)
        return fail(RAFTCORR_EINVAL, "alt_features_bwd: num_levels = %d not in [1, %d]", num_levels, RAFTCORR_MAX_LEVELS);
    for (int l = 0; l < num_levels; ++l) {
        if ((H >> l) < 1 || (W >> l) < 1)
            return fail(RAFTCORR_EINVAL, "alt_features_bwd: level %d of a %d x %d map is empty", l, H, W);
        if (!d2_levels[l]) return fail(RAFTCORR_EINVAL, "alt_features_bwd: level %d pointer is null", l);
    }
    return launch_alt_features_bwd(d1, d2_levels, B, C, H, W, num_levels, dfmap1, dfmap2, (hipStream_t)stream);
}

}  // extern "C"
