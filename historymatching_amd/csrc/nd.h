// nd.h -- nested-dissection (multifrontal) pressure solve for the 128 x 128, 256 x 256 and 512 x 512 grids: symbolic tables shared
// by the host builder and the kernels of press_nd.hip.
//
// The elimination tree is the geometric dissection of the grid (SURVEY.md A.3's five-point system; tests/tools/nd_prototype.py):
// a region is cut across its longer side by a one-cell separator until both sides are <= 4 cells.  For 128 x 128 the
// tree is a complete binary tree: levels 0..9 are separators (128, 64, 64, 32, 32, 16, 16, 8, 8, 4 cells), level 10 the
// 1024 leaves (3x3 .. 4x4 cells).  Front (level l, index i) has id 2^l - 1 + i; its children are (l+1, 2i) and (l+1, 2i+1).
// A grid of 128 * 2^h cells a side has 2 h more levels on top (LO = 2 h: 13 levels at 256 x 256, 15 at 512 x 512); its levels
// LO + 4 .. LO + 10 have the shapes of the 128 x 128 tree's levels 4..10 (same separators, same largest boundaries) and are
// eliminated by the same kernels (level LO + 3 by the workgroup kernel as well); its levels 0 .. LO + 2 are the BIG fronts (up to 25 / 49 tile rows), eliminated tile column by
// tile column out of global memory (press_nd.hip: k_big_*), their update matrices stored as whole 16 x 16 tiles.
//
// A front = s pivots (its separator / leaf cells) + b boundary cells (the part of the region's perimeter that is an
// ancestor's separator) + ONE extra boundary row that carries the right-hand side.  The boundary cells of a front with children are
// ORDERED BY CHILD (round 5): those on child 0's perimeter, those on neither's (next to the separator's ends only), those on child
// 1's -- the assembled update matrix is then block diagonal over the children, and a tile that lies in one child's block (or in
// neither's) skips the other child's gather altogether (NDF_KIDM; half of a front's tile gathers and more).  In tiles of 16: st pivot tiles,
// bt = ceil((b + 1) / 16) boundary tiles.  Front position p: [0, 16 st) pivots (padded with identity rows),
// [16 st, 16 st + b) boundary cells, 16 st + b the right-hand-side row, then padding.
#pragma once

#define ND_FRONT_INTS 24
// ints of one front record
#define NDF_LEVEL 0
#define NDF_S 1
#define NDF_B 2
#define NDF_ST 3
#define NDF_BT 4
#define NDF_C0 5       // child front ids (-1: leaf)
#define NDF_C1 6
#define NDF_CELLS 7    // offset of the front's 16 (st + bt) position -> cell entries in `cells` (-1 padding, -2 right-hand side)
#define NDF_FACT 8     // offset (doubles) of the front's factor in a member's factor block
#define NDF_UPD 9      // offset (doubles) of the front's update matrix in a member's arena (levels <= ND_ARENA_MAX_LEVEL), else -1
#define NDF_KREG 10    // 4-row groups of the LAST pivot tile that hold real pivots (the other pivot tiles are full)
#define NDF_BC0 11     // b of child 0 / child 1 (position of the child's right-hand-side row)
#define NDF_BC1 12
#define NDF_PBOX 13    // bounding box of the front's pivot cells: x0 | x1 << 16 (x1 exclusive); its y range in NDF_PBOY
#define NDF_RBOX 14    // the front's whole region (pivots of the front and of all its descendants), same packing, y range in NDF_RBOY
#define NDF_REC 15     // offset of the front's assembly recipes in `rec`, in blocks of 256 int16 (nd.h: recipes)
#define NDF_UC0 16     // NDF_UPD of child 0 / child 1 (a parent finds its children's updates without reading their records)
#define NDF_UC1 17
#define NDF_PBOY 18    // y0 | y1 << 16 of the pivot box
#define NDF_RBOY 19    // y0 | y1 << 16 of the region
#define NDF_PIMG 20    // big fronts: offset (doubles) of the front's st inverse pivot tiles in a member's pivot-image scratch, else -1
#define NDF_KIDM 21    // fronts of at most 16 tile rows: bit R = tile row R holds a position of child 0's update, bit 16 + R = of child 1's
                       // (the pivot tiles and the right-hand-side row's tile: both); -1: no information, gather everything
#define NDF_COFM 22    // same fronts: bit R = tile row R has a matrix coefficient against some pivot of the front (a pivot tile, the
                       // right-hand-side row's tile, a tile with a cell next to the separator's ends); -1: no information

#define ND_MAX_LEVELS 15       // 512 x 512
// Level numbers below are those of the 128 x 128 tree; a larger grid's level is LO higher (NdInfo::lo).
#define ND_ARENA_MAX_LEVEL 8   // updates of levels 1..8 and of the leaves (level 10: written by k_nd_leaf) live in the per-member arena
                               // (global memory); level 9's in per-wave LDS slots
#define ND_WAVE_TOP_LEVEL 5    // levels 10..5: one wave per front;  levels 4..0: one workgroup per front (per member) at 128 x 128,
                               // the big-front kernels on the larger grids
// Update matrices of the BIG fronts (levels 0 .. LO + 2 of a grid with LO > 0) are stored as whole tiles: tile (R, C), 0 <= C <= R < bt,
// at NDF_UPD + (R (R + 1) / 2 + C) * 256 doubles, entry (i, j) = (16 R + 4 r + lq, 16 C + lc) at [r * 64 + lq * 16 + lc] -- the accumulator
// layout as it stands in the registers.  All other update matrices are packed lower triangles, entry (i, j) at i (i + 1) / 2 + j.

// Assembly RECIPES: what each lane of a wave reads, adds and writes when it assembles a front, precomputed per front (they depend on
// the grid only), so that the kernels spend no instructions on index arithmetic.  One block = 256 int16 in the accumulator layout's
// lane order: entry [lane][r], lane = lq * 16 + lc, r = 0..3  (a lane loads its four values of a block as one 8-byte word).
// Tile entry (lane, r) of a PANEL tile R (transposed panel: pivot k = 4 r + lq, front row m = 16 R + lc) and of a TRAILING tile
// (R, C) (front rows i = 16 R + 4 r + lq, j = 16 C + lc, both counted from the first boundary row).  Kinds:
//   COEF  where A[cell m, cell k] (or q[cell k] on the right-hand-side row) lies: levels >= 5: byte offset into the wave's LDS block
//         (below); levels <= 4: index into the member's coefficient block, as int32 over two blocks, -1: zero, -2: padded pivot (identity)
//   G0,G1 where the entry lies in child 0 / child 1's packed lower-triangular update matrix: levels >= 5: byte offset into the wave's LDS
//         block; levels <= 4: offset in the child's packed array, -1: none
//   OUT   offset of the entry in this front's packed update matrix, -1: not stored (upper triangle, padding)
// Block order per front:  leaves: COEF(R) for R = 0..bt, then OUT(R, C) for R = 1..bt, C = 1..R;
//   levels 5..9: [COEF, G0, G1](R) for R = 0..bt, then [G0, G1, OUT](R, C);
//   levels 0..4 have no recipes (k_nd_top and the big-front kernels work from the position tables).
// Levels >= 5 (one wave per front): COEF, G0 and G1 are BYTE offsets into the wave's LDS block, so that an entry is one ds_read
// and one add -- "none" points at a cell that holds 0.0, "identity" at one that holds 1.0:
//   block (doubles): [0] = 0.0, [1] = 1.0, data from ND_LDS_DATA on:
//     levels 8..10 (k_nd_sub): level-9 slots A, B | level-10 slots A, B | 4 coefficient planes of ND_CF_PLANE_SUB
//     levels 5..7 (k_nd_wave): child 0's update | child 1's update (upd_doubles[level + 1] each) | 4 planes of ND_CF_PLANE_WAVE
#define ND_LDS_ZERO 0
#define ND_LDS_ONE 1
#define ND_LDS_DATA 2
#define ND_CF_PLANE_SUB 100   // LDS coefficient planes of a level-8 subtree: (8 + 2)^2 cells, box = the level-8 front's region
#define ND_CF_PLANE_WAVE 56   // of a level 5..7 front: (1 + 2) x (16 + 2) cells, box = the front's pivot line

struct NdInfo {
    int n_fronts;
    int levels;               // 11, 13, 15
    int lo;                   // levels - 11
    int n_cells;              // entries of `cells` (and, twice, of `cpos`)
    int n_rec_blocks;         // blocks of 256 int16 in `rec`
    long long fact_doubles;   // per member
    long long big_fact_doubles;  // of that, the factor of the big fronts (levels 0 .. lo + 2; they come first): the size of the image scratch
    long long arena_doubles;  // per member
    long long pimg_doubles;   // per member: inverse pivot tiles of the big fronts
    int upd_doubles[ND_MAX_LEVELS];  // largest update matrix per level (packed: (b + 1)(b + 2) / 2 rounded up to even; tiles: bt (bt + 1) / 2 * 256)
    int max_bt[ND_MAX_LEVELS];    // boundary tiles per level (for the kernels' static register arrays)
    int max_st[ND_MAX_LEVELS];
};
