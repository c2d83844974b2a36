// Memory layout of the correlation pyramid ("patch-tile order"), shared by host and device.
//
// The level-0 volume is literally the output matrix of the all-pairs contraction,
// C[q][p] (row = query pixel, S0 columns per row), with the target pixels (columns) permuted
// so that (a) a 10x10 lookup window touches few 128-byte lines and (b) every 2x2 / 4x4 / 8x8
// pooling block of targets sits in one wavefront's MFMA accumulators:
//
//   target image -> 8x16 PATCHES (row-major over the image, zero padded at the far edges)
//   patch        -> four 4x8 TILES, n = 2*(row half) + (column half); a tile is one 128-B line
//   tile         -> 32 cells, position j = dx0 | dy0<<1 | dy1<<2 | dx1<<3 | dx2<<4
//                   (dx = column 0..7, dy = row 0..3 inside the tile)
// so   pos0(y, x) = patch*128 + n*32 + j.
// j is also the MFMA column (lane & 31) that produces the cell: the 2x2 pooling partners are
// lanes xor 1 / xor 2 (one DPP quad), the 4x4 partners are the rest of a 16-lane DPP row.
//
// Pooled levels keep the patch order and shrink with it; inside a patch the cells are ordered so
// that EVERY even-aligned 2x2 block of cells of EVERY level is 16 contiguous, 16-byte aligned
// bytes with inner order (x&1) | (y&1)<<1 -- a lookup fetches four taps with one load:
//   level 1: 32 cells per patch (4x8, one line):  pos1 = patch*32 + n*8 + (x&1 | (y&1)<<1 | ((x>>1)&1)<<2)
//   level 2:  8 cells per patch (2x4):            pos2 = patch*8  + (x&1 | (y&1)<<1 | ((x>>1)&1)<<2)
//   level 3:  2 cells per patch (1x2), patch rows paired: pos3 = ((y>>1)*PW + (x>>1))*4 + (x&1) + (y&1)*2
// (x, y = the level's own cell coordinates; n as at level 0 from the block's top-left cell).
// Levels >= 4 (not used by RAFT) are plain row-major H_l x W_l.
//
// Cells of the padding (beyond H_l x W_l) hold finite don't-care values and are never read.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RC_HD __host__ __device__ __forceinline__
#else
#define RC_HD inline
#endif

namespace raftcorr {

constexpr int PATCH_H = 8;
constexpr int PATCH_W = 16;
constexpr int PATCH_CELLS = PATCH_H * PATCH_W;   // 128
constexpr int FUSED_LEVELS = 4;                  // levels produced in patch-tile order

struct LevelGeom {
    int h, w;        // logical size of the level
    int hp, wp;      // allocated (padded) size of the level, both even for kinds 0..3
    int pw;          // patches per patch-row (level 0 geometry), 0 for row-major levels
    int kind;        // 0..3: patch-tile order of that level; 4: row-major
    int64_t slab;    // elements per query slab
};

RC_HD int patches_h(int H) { return (H + PATCH_H - 1) / PATCH_H; }
RC_HD int patches_w(int W) { return (W + PATCH_W - 1) / PATCH_W; }

RC_HD LevelGeom level_geom(int H, int W, int l)
{
    LevelGeom g;
    g.h = H >> l; g.w = W >> l;
    if (l < FUSED_LEVELS) {
        const int np = patches_h(H) * patches_w(W);
        g.pw = patches_w(W);
        g.kind = l;
        g.hp = l < 3 ? (patches_h(H) * PATCH_H) >> l : ((patches_h(H) + 1) >> 1) * 2;
        g.wp = (patches_w(W) * PATCH_W) >> l;
        g.slab = l < 3 ? (int64_t)np * (PATCH_CELLS >> (2 * l))
                       : (int64_t)((patches_h(H) + 1) >> 1) * patches_w(W) * 4;
    } else {
        g.pw = 0; g.kind = 4; g.hp = g.h; g.wp = g.w;
        g.slab = (int64_t)g.h * g.w;
    }
    return g;
}

// position of logical cell (y, x) of a level inside its query slab
RC_HD int level_pos(const LevelGeom& g, int y, int x)
{
    switch (g.kind) {
    case 0: {
        const int patch = (y >> 3) * g.pw + (x >> 4);
        const int n = (((y >> 2) & 1) << 1) | ((x >> 3) & 1);
        const int dy = y & 3, dx = x & 7;
        const int j = (dx & 1) | ((dy & 1) << 1) | ((dy >> 1) << 2) | (((dx >> 1) & 1) << 3) | ((dx >> 2) << 4);
        return patch * 128 + n * 32 + j;
    }
    case 1: {
        const int patch = (y >> 2) * g.pw + (x >> 3);
        const int n = (((y >> 1) & 1) << 1) | ((x >> 2) & 1);
        return patch * 32 + n * 8 + ((x & 1) | ((y & 1) << 1) | (((x >> 1) & 1) << 2));
    }
    case 2: {
        const int patch = (y >> 1) * g.pw + (x >> 2);
        return patch * 8 + ((x & 1) | ((y & 1) << 1) | (((x >> 1) & 1) << 2));
    }
    case 3:
        return ((y >> 1) * g.pw + (x >> 1)) * 4 + (x & 1) + (y & 1) * 2;
    default:
        return y * g.w + x;
    }
}

}  // namespace raftcorr
