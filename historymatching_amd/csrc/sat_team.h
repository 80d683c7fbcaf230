// sat_team.h -- hand-off primitives shared by the workgroup-team saturation sweeps (sat128t.hip: one workgroup per 128 x 128 tile; sat256s.hip,
// sat32s.hip: one per slab of rows): the workgroups of a member exchange tile edges / border rows once per explicit sub-step.
//
// Everything exchanged is a GRANULE: a naturally aligned 8-byte word {tag = event number + 1 (high half), 32 payload bits
// (low half)} written by one write-through (sc1) store and polled with sc1 loads until the tag matches -- the data is its
// own flag, there is no separate counter, drain or cache-wide release/acquire (MI355X_MICROARCH.md, inter-workgroup
// visibility, form R2).  A double travels as two granules, a float as one.  Tags restart at 0 every launch: the host zeroes
// the team blocks before each launch.
#pragma once
#include "fwd.h"

namespace sat_team {

constexpr int TS = 128;          // tile size (cells per side)
constexpr int MAX_TILES = 32;
constexpr int SPIN_LIMIT = 1 << 22;
typedef unsigned long long u64;

// Per-team block in global memory; GPV = granules per value (2 for fp64, 1 for fp32).
template <int GPV>
struct TeamLayout {
    int T;
    __host__ __device__ size_t cfl_off() const { return 0; }                                              // [T][2] granules (a double)
    __host__ __device__ size_t pub_off() const { return ((size_t)T * 2 * 8 + 127) & ~(size_t)127; }       // [T][2 parities][4 edges][GPV][128]
    __host__ __device__ size_t bytes() const { return pub_off() + (size_t)T * 2 * 4 * GPV * TS * 8; }
};

__device__ __forceinline__ void put_granule(u64* g, unsigned payload, unsigned tag) {
    __hip_atomic_store(g, ((u64)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool spin_failed(int& spins, int* dead) {
    if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) || ++spins > SPIN_LIMIT) {
        __hip_atomic_store(dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return true;
    }
    __builtin_amdgcn_s_sleep(2);
    return false;
}
__device__ __forceinline__ void put_double(u64* lo, u64* hi, double v, unsigned tag) {
    put_granule(lo, (unsigned)__double2loint(v), tag);
    put_granule(hi, (unsigned)__double2hiint(v), tag);
}
// poll the granule(s) of one value until they carry `tag`; every active lane polls its own, the wave leaves together
__device__ __forceinline__ bool get_double(const u64* lo, const u64* hi, unsigned tag, double& v, int* dead) {
    for (int spins = 0;;) {
        const u64 x = __hip_atomic_load(lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 y = __hip_atomic_load(hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((unsigned)(x >> 32) == tag && (unsigned)(y >> 32) == tag)) {
            v = __hiloint2double((int)(unsigned)y, (int)(unsigned)x);
            return true;
        }
        if (spin_failed(spins, dead)) return false;
    }
}
__device__ __forceinline__ bool get_float(const u64* g, unsigned tag, float& v, int* dead) {
    for (int spins = 0;;) {
        const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((unsigned)(x >> 32) == tag)) {
            v = __uint_as_float((unsigned)x);
            return true;
        }
        if (spin_failed(spins, dead)) return false;
    }
}

// Workgroup -> (team, tile): the tiles of a team have workgroup ids that differ by multiples of 8 (workgroups are dealt
// round-robin to the 8 XCDs), so a team shares one L2 -- a speed bonus only.
__device__ __forceinline__ void team_of_block(int T, int& team, int& tile) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    team = (slot / T) * 8 + xcd;
    tile = slot % T;
}

// Host: does the multi-tile form apply to this grid, and how many teams fit one launch (one workgroup per CU)?
inline bool tiles_of(const hm_fwd* f, int& TXn, int& TYn, int& max_teams) {
    const FwdParams& p = f->p;
    if (p.Nx % TS || p.Ny % TS) return false;
    TXn = p.Nx / TS;
    TYn = p.Ny / TS;
    const int T = TXn * TYn, slots = f->ctx->num_cu / 8;
    if (T < 2 || T > MAX_TILES || T > slots) return false;
    max_teams = 8 * (slots / T);
    return true;
}
// at most one well per 8 x 4 patch, at most max_wells wells
inline bool wells_fit_patches(const hm_fwd* f, int max_wells) {
    if ((int)f->well_cells_host.size() > max_wells) return false;
    std::vector<long long> seen;
    for (int cell : f->well_cells_host) {
        const long long id = (long long)((cell / f->p.Ny) >> 3) * 100000 + ((cell % f->p.Ny) >> 2);
        for (long long s : seen)
            if (s == id) return false;
        seen.push_back(id);
    }
    return true;
}

}  // namespace sat_team
