/* TEST INFRASTRUCTURE: drives the HOST side of libraftcorr (built by build_host_asan.sh with AddressSanitizer + UBSan on
 * the host code) without a GPU: the host-only entry points over a sweep of shapes with output arrays of exactly the
 * documented sizes, the size queries up to shapes far beyond the configs (signed overflow would be reported), and the
 * argument checks of every compute entry point (bad arguments must come back as a negative status before anything is
 * launched).  The bad-argument section passes made-up device addresses, so it only runs where there is no GPU. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "raftcorr.h"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); ++fails; } } while (0)

static void layout_sweep(void)
{
    static const int shapes[][2] = {{1, 1}, {1, 7}, {2, 2}, {3, 5}, {8, 16}, {9, 17}, {13, 17}, {16, 24}, {47, 156}, {50, 90},
                                    {55, 128}, {136, 240}, {190, 190}, {272, 480}, {5, 1000}, {1000, 3}};
    int cases = 0;
    for (unsigned k = 0; k < sizeof(shapes) / sizeof(shapes[0]); k++)
        for (int L = 1; L <= 6; L++) {
            const int H = shapes[k][0], W = shapes[k][1];
            int32_t* hl = malloc(sizeof(int32_t) * L); int32_t* wl = malloc(sizeof(int32_t) * L);
            int64_t* slab = malloc(sizeof(int64_t) * L);
            const int rc = raftcorr_level_layout(H, W, L, hl, wl, slab);
            if ((H >> (L - 1)) < 1 || (W >> (L - 1)) < 1) {          /* a level of size 0: must be refused or give empty levels */
                free(hl); free(wl); free(slab);
                continue;
            }
            CHECK(rc == RAFTCORR_OK, "level_layout(%d,%d,%d) = %d (%s)", H, W, L, rc, raftcorr_last_error());
            for (int l = 0; rc == RAFTCORR_OK && l < L; l++, cases++) {
                CHECK(hl[l] == (H >> l) && wl[l] == (W >> l), "level %d of %dx%d: %dx%d", l, H, W, hl[l], wl[l]);
                CHECK(slab[l] >= (int64_t)hl[l] * wl[l], "slab %lld < cells", (long long)slab[l]);
                const size_t n = (size_t)hl[l] * wl[l];
                int32_t* pos = malloc(sizeof(int32_t) * n);
                unsigned char* seen = calloc((size_t)slab[l], 1);
                const int rp = raftcorr_level_positions(H, W, l, pos);
                CHECK(rp == RAFTCORR_OK, "level_positions(%d,%d,%d) = %d", H, W, l, rp);
                for (size_t i = 0; rp == RAFTCORR_OK && i < n; i++) {
                    if (pos[i] < 0 || pos[i] >= slab[l] || seen[pos[i]]) { CHECK(0, "position %zu of level %d of %dx%d: %d", i, l, H, W, pos[i]); break; }
                    seen[pos[i]] = 1;
                }
                free(pos); free(seen);
            }
            free(hl); free(wl); free(slab);
        }
    printf("layout sweep: %d levels checked\n", cases);
}

static void size_queries(void)
{
    static const int shapes[][4] = {   /* B D H W */
        {1, 256, 55, 128}, {1, 256, 47, 156}, {6, 256, 50, 90}, {1, 256, 136, 240}, {1, 128, 1, 1}, {1, 1, 3, 5}, {2, 24, 13, 17},
        {8, 256, 272, 480}, {64, 512, 190, 190}, {1, 1024, 544, 960}, {16, 256, 1088, 1920},
    };
    int n = 0;
    for (unsigned k = 0; k < sizeof(shapes) / sizeof(shapes[0]); k++) {
        const int B = shapes[k][0], D = shapes[k][1], H = shapes[k][2], W = shapes[k][3];
        int64_t v[16]; int m = 0;
        v[m++] = raftcorr_build_workspace_bytes(B, D, H, W);
        v[m++] = raftcorr_build_bwd_workspace_bytes(B, D, H, W);
        v[m++] = raftcorr_build_bwd_deferred_workspace_bytes(B, D, H, W);
        v[m++] = raftcorr_alt_packed_bytes(B, D, H, W);
        v[m++] = raftcorr_alt_tiles_bytes(B, D, H, W);
        for (int L = 1; L <= 4; L++) {
            if ((H >> (L - 1)) < 1 || (W >> (L - 1)) < 1) break;
            v[m++] = raftcorr_alt_lookup_fwd_sorted_workspace_bytes(B, H, W, L);
            v[m++] = raftcorr_alt_prepare_workspace_bytes(B, D, H, W, L);
            for (int r = 0; r <= 7; r++) {
                const int64_t a = raftcorr_alt_lookup_bwd_workspace_bytes(B, H, W, D, L, r);
                const int64_t s = raftcorr_alt_lookup_bwd_sorted_workspace_bytes(B, H, W, D, L, r);
                CHECK(a >= -2 && s >= -2, "bwd workspace sizes %lld %lld", (long long)a, (long long)s);   /* negative = status code */
                (void)raftcorr_build_bwd_deferred_supported(H, W, L, r);
                n += 3;
            }
            if (m > 12) m = 5;
        }
        for (int i = 0; i < 5; i++) { CHECK(v[i] > 0 || v[i] == RAFTCORR_EINVAL || v[i] == RAFTCORR_EUNSUPPORTED, "size query %d of shape %u: %lld", i, k, (long long)v[i]); n++; }
    }
    CHECK(raftcorr_conv1x1_packed_bytes(256, 324) > 0 && raftcorr_conv1x1_packed_bytes(96, 196) > 0, "conv1x1_packed_bytes");
    /* non-positive arguments */
    CHECK(raftcorr_build_workspace_bytes(0, 256, 55, 128) <= 0 && raftcorr_build_workspace_bytes(1, -1, 55, 128) <= 0, "bad sizes accepted");
    CHECK(raftcorr_alt_packed_bytes(1, 256, 0, 5) <= 0, "bad sizes accepted");
    printf("size queries: %d values\n", n);
}

static void bad_arguments(void)
{
    /* made-up, never dereferenced on the host: every call below has at least one bad argument and must return < 0 */
    float* dev = (float*)(uintptr_t)0x7f0000001000ull;
    float* lv[8] = {dev, dev, dev, dev, dev, dev, dev, dev};
    const float* clv[8] = {dev, dev, dev, dev, dev, dev, dev, dev};
    void* vlv[8] = {dev, dev, dev, dev, dev, dev, dev, dev};
    const void* cvlv[8] = {dev, dev, dev, dev, dev, dev, dev, dev};
    int n = 0;
#define BAD(call) do { const int rc_ = (call); CHECK(rc_ < 0, "%s returned %d", #call, rc_); CHECK(rc_ >= 0 || strlen(raftcorr_last_error()) > 0, "%s: empty error text", #call); ++n; } while (0)
    BAD(raftcorr_build_pyramid(NULL, dev, 1, 256, 55, 128, 4, lv, dev, 1 << 30, NULL));
    BAD(raftcorr_build_pyramid(dev, dev, 0, 256, 55, 128, 4, lv, dev, 1 << 30, NULL));
    BAD(raftcorr_build_pyramid(dev, dev, 1, 256, 55, 128, 4, NULL, dev, 1 << 30, NULL));
    BAD(raftcorr_build_pyramid(dev, dev, 1, 256, 55, 128, 4, lv, dev, 16, NULL));                       /* workspace too small */
    BAD(raftcorr_build_pyramid(dev, dev, 1, 256, 55, 128, 0, lv, dev, 1 << 30, NULL));
    BAD(raftcorr_build_pyramid_prec(dev, dev, 1, 256, 55, 128, 4, lv, dev, 1 << 30, 17, NULL));          /* unknown precision */
    BAD(raftcorr_build_pyramid_vol(dev, dev, 1, 256, 55, 128, 4, vlv, dev, 1 << 30, RAFTCORR_PREC_BF16, 9, NULL));
    BAD(raftcorr_build_pyramid_vol(dev, dev, 1, 256, 55, 128, 5, vlv, dev, 1 << 30, RAFTCORR_PREC_BF16, RAFTCORR_VOL_BF16, NULL));
    BAD(raftcorr_build_pyramid_noalloc(dev, dev, 1, 256, 136, 240, 4, lv, RAFTCORR_PREC_FP32, NULL));    /* beyond the built-in scratch */
    BAD(raftcorr_build_pyramid_noalloc(NULL, dev, 1, 256, 55, 128, 4, lv, RAFTCORR_PREC_FP32, NULL));
    BAD(raftcorr_lookup_fwd(clv, 1, 55, 128, 4, -1, dev, RAFTCORR_MODE_FORK, dev, NULL));
    BAD(raftcorr_lookup_fwd(NULL, 1, 55, 128, 4, 4, dev, RAFTCORR_MODE_FORK, dev, NULL));
    BAD(raftcorr_lookup_fwd(clv, 1, 55, 128, 4, 4, dev, 7, dev, NULL));                                  /* unknown mode */
    BAD(raftcorr_lookup_fwd(clv, 1, 55, 128, 4, 4, NULL, RAFTCORR_MODE_FORK, dev, NULL));
    BAD(raftcorr_lookup_fwd_vol(cvlv, 5, 1, 55, 128, 4, 4, dev, RAFTCORR_MODE_FORK, dev, NULL));
    BAD(raftcorr_conv1x1_pack_weight(dev, 256, 324, dev, 8, NULL));
    BAD(raftcorr_lookup_conv1x1_fwd(clv, 1, 55, 128, 4, 9, dev, dev, NULL, 256, 1, dev, NULL));          /* radius 9 */
    BAD(raftcorr_lookup_bwd(clv, NULL, 1, 55, 128, 4, 4, dev, RAFTCORR_MODE_FORK, dev, NULL, NULL));     /* nothing to compute */
    BAD(raftcorr_lookup_bwd(NULL, lv, 1, 55, 128, 4, 4, dev, RAFTCORR_MODE_FORK, dev, dev, NULL));       /* dcoords needs levels */
    BAD(raftcorr_build_bwd(dev, dev, 1, 256, 55, 128, 4, lv, dev, dev, dev, 16, NULL));
    BAD(raftcorr_build_bwd(dev, dev, 1, 256, 55, 128, 4, lv, NULL, dev, dev, 1 << 30, NULL));
    BAD(raftcorr_lookup_bwd_record(NULL, 1, 55, 128, 324, dev, NULL));
    BAD(raftcorr_lookup_bwd_record(dev, 1, 55, 128, 0, dev, NULL));
    BAD(raftcorr_build_bwd_deferred(dev, dev, 1, 256, 55, 128, 4, 4, RAFTCORR_MODE_CANONICAL, 0, clv, clv, NULL, dev, dev, dev, 1 << 30, NULL));
    BAD(raftcorr_build_bwd_deferred(dev, dev, 1, 256, 55, 128, 4, 4, RAFTCORR_MODE_CANONICAL, 2, NULL, clv, NULL, dev, dev, dev, 1 << 30, NULL));
    BAD(raftcorr_build_bwd_deferred(dev, dev, 1, 256, 55, 128, 4, 4, RAFTCORR_MODE_CANONICAL, 2, clv, clv, NULL, dev, dev, dev, 16, NULL));
    BAD(raftcorr_alt_forward(dev, dev, dev, NULL, 1, 1, 55, 128, 55, 128, 256, 4, NULL));
    BAD(raftcorr_alt_forward(dev, dev, dev, dev, 1, 1, 55, 128, 55, 128, 256, -2, NULL));
    BAD(raftcorr_alt_backward(dev, dev, dev, dev, NULL, dev, NULL, 1, 1, 55, 128, 55, 128, 256, 4, NULL));
    BAD(raftcorr_alt_backward(dev, dev, dev, dev, dev, dev, NULL, 1, 0, 55, 128, 55, 128, 256, 4, NULL));
    BAD(raftcorr_alt_scales(dev, NULL, 1, 256, 55, 128, dev, NULL));
    BAD(raftcorr_alt_pack(dev, 1, 256, 55, 128, dev, dev, 8, NULL));
    BAD(raftcorr_alt_pack_tiles(dev, 1, 256, 55, 128, dev, dev, 8, NULL));
    BAD(raftcorr_alt_lookup_fwd(dev, NULL, dev, 1, 55, 128, 256, 4, 4, dev, dev, NULL));
    BAD(raftcorr_alt_lookup_fwd(dev, cvlv, dev, 1, 55, 128, 250, 4, 4, dev, dev, NULL));                 /* C not a multiple of 32 */
    BAD(raftcorr_alt_lookup_fwd_sorted(dev, cvlv, dev, 1, 55, 128, 256, 4, 4, dev, dev, dev, 16, NULL));
    BAD(raftcorr_alt_lookup_fwd_sorted(dev, cvlv, dev, 1, 55, 128, 256, 4, 4, dev, dev, NULL, 1 << 30, NULL));
    BAD(raftcorr_alt_lookup_bwd(dev, cvlv, dev, 1, 55, 128, 256, 4, 4, dev, dev, dev, lv, 0, dev, 16, NULL));
    BAD(raftcorr_alt_lookup_bwd(dev, cvlv, dev, 1, 55, 128, 256, 4, 4, dev, dev, dev, NULL, 0, dev, 1 << 30, NULL));
    BAD(raftcorr_alt_lookup_bwd_sorted(dev, cvlv, dev, 1, 55, 128, 256, 4, 4, dev, dev, dev, lv, 0, NULL, dev, 16, NULL));
    BAD(raftcorr_alt_lookup_bwd_sorted(dev, cvlv, dev, 1, 55, 128, 256, 4, 4, NULL, dev, dev, lv, 0, NULL, dev, 1 << 30, NULL));
    BAD(raftcorr_alt_prepare(dev, dev, 1, 256, 55, 128, 4, dev, dev, vlv, NULL, NULL, dev, 16, NULL));
    BAD(raftcorr_alt_prepare(dev, dev, 1, 256, 55, 128, 4, NULL, dev, vlv, NULL, NULL, dev, 1 << 30, NULL));
    BAD(raftcorr_alt_features_bwd(dev, NULL, 1, 256, 55, 128, 4, dev, dev, NULL));
    BAD(raftcorr_alt_features_bwd(dev, clv, 1, 256, 55, 128, 4, NULL, dev, NULL));
    printf("bad-argument calls: %d, all refused\n", n);
}

int main(void)
{
    printf("raftcorr_version %d\n", raftcorr_version());
    layout_sweep();
    size_queries();
    if (access("/dev/kfd", F_OK) == 0) printf("a GPU driver node is present: bad-argument section skipped (it passes made-up device addresses)\n");
    else bad_arguments();
    printf("host layer under asan+ubsan: %d failures\n", fails);
    return fails ? 1 : 0;
}
