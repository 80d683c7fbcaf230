// nd_plan.h -- the device-side handle and the host-side plan of the nested-dissection pressure solve (press_nd.hip).
// press_nd.hip is compiled three times (ND_LG = 7, 8, 9: press_nd.o, press_nd256.o, press_nd512.o); all three objects allocate, use and
// free `hm_nd`, so both types are defined HERE, once, outside any anonymous namespace: one type under one name in every object (round 4
// defined them per object, NdDev inside an anonymous namespace -- the same name for three formally different types).  Nothing in either
// struct depends on the grid size; the per-grid constants (ND_WORK_INTS, NCACHE, ...) stay in press_nd.hip.
#pragma once
#include "common.h"
#include "nd.h"

struct NdDev {
    const int* fronts;
    const int* cells;
    const short* cpos;
    const short* rec;  // assembly recipes (nd.h), blocks of 256 int16
    // Flat per-leaf and per-subtree tables (round 6, press_nd.hip: nd_build_flat_tables): everything a lane of the leaf kernels / a wave of
    // k_nd_solve_sub derives today from front record -> position table -> cell through two or three DEPENDENT trips to memory, as one
    // coalesced read.  leaft: [ND_LEAF_INTS][leaves] ints, entry-major (a lane per leaf reads consecutive ints);  ssub: per level-8 subtree
    // [ND_SSUB_HDR] wave-uniform ints, then [ND_SSUB_LANE][64] per-lane ints.
    const int* leaft;
    const int* ssub;
    // the leaves' update matrices (k_nd_leaf -> k_nd_sub), per member and level-8 subtree one block of 4 * slot10 doubles, the four leaves
    // INTERLEAVED: entry e of leaf l at [4 e + l].  k_nd_leaf's lanes (one leaf each, four consecutive lanes = the leaves of one subtree)
    // then write 32 contiguous bytes per subtree and entry instead of 8 bytes into four packed arrays 700 bytes apart, and k_nd_sub reads
    // its subtree's block as one contiguous piece (round 6; the arena's per-leaf slots are no longer used)
    double* leafu;
    long long leafu_stride;
    double* fact;
    double* arena;
    double* cf;  // per member: [dg | -TX | -TY | q], CF_STRIDE doubles
    long long fact_stride, arena_stride;
    int slot9, slot10;      // doubles per LDS update slot of levels 9, 10 (k_nd_sub)
    int child_doubles[3];   // largest child update of a level-7 / 6 / 5 front (k_nd_wave's LDS staging)
    int top_child_doubles;  // largest child update of a level <= 4 front
    // Reuse across time steps (k_nd_plan): per member the fronts of levels 8..5 that have to be eliminated this step, compacted
    int* work;                   // N x ND_WORK_INTS: [n8, n7, n6, n5 | list8[256] | list7[128] | list6[64] | list5[32] | nt, fronts of levels 4..0]
                                 // (larger grids: the lists are 4 / 16 times as long; the fronts of levels <= LO + 4 have `todo` bytes instead)
    unsigned char* cached;       // N x NCACHE (512 at 128 x 128): front f (levels 0..8: f = 0..510) holds the results of its all-dry state
    const unsigned char* wells;  // NCACHE: a well somewhere in the front's subtree (its right-hand side rows carry the rates)
    int wells_ok;                // the rates of this time step are those the cached results of such fronts were computed with
    int reuse;                   // 0: every front is eliminated every step
    int top_deal;                // k_nd_top: 1 = trailing tiles in row-major runs on the waves that own no pivot tile (round 6), 0 = round-robin as before
    // larger grids: the plan's inputs / outputs, and the big fronts' (levels 0 .. LO + 2) panel images
    unsigned long long* wet;     // per member: wet-cell bitmap, NB rows x NB / 64 words (k_ndl_assemble -> k_ndl_plan)
    unsigned char* todo;         // per member: NTODO bytes, front f of levels 0 .. LO + 4 is eliminated this step (k_ndl_plan)
    double* vfac;                // per member: the negated pivot-panel tiles V(p, R) at the factor's offsets (operands of the trailing products)
    double* pimg;                // per member: the negated inverse pivot tiles, NDF_PIMG + 256 p
    long long vfac_stride, pimg_stride;
};

struct hm_nd {
    NdInfo info{};
    DevBuf fronts, cells, cpos, rec, fact, arena, dg, work, cached, wells, vfac, pimg, wet, todo, leaft, ssub, leafu;
    NdDev dev{};
    int cap = 0;                // members the per-member buffers hold: larger ensembles are solved in blocks of `cap` members (larger grids)
    long long cached_gen = -1;  // hm_fwd::inputs_gen the cached results belong to
    int cached_q_epoch = -1;    // hm_fwd::q_epoch of the time step the cached results of fronts with wells belong to
};
