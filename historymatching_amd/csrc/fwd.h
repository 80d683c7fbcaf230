// fwd.h -- shared definitions of the ensemble forward-model path (plan struct, kernel parameter block).
#pragma once
#include "common.h"

// Parameter block passed by value to every forward kernel.  All per-member arrays are laid out
// member-major: array[m * stride + cell].  Cell index = ix*Ny + iy (C order of shape (Nx,Ny),
// reference: p.reshape(model.shape) HistoryMatch.py:163).
struct FwdParams {
    int N, Nx, Ny, Nxy;
    int nInj, nPrd, nTime;
    double hx, hy, h2;        // cell sizes, h2 = hx*hy
    double cx, cy;            // cx = (2*hy)/hx, cy = (2*hx)/hy   (TPFA harmonic-mean prefactors)
    double vw, vo, swc, sor;  // fluid
    int fluid_default;        // 1 when vw=vo=1, swc=sor=0: S*=S, Mw=S^2, Mo=(1-S)^2 exactly
    double dt;
    const double* K;          // N*Nxy x-permeability (= y-permeability when Ky is null: set_perm HistoryMatch.py:164 sets Kx = Ky)
    const double* Ky;         // N*Nxy y-permeability or nullptr
    const double* por;        // Nxy porosity or nullptr (=1)
    const double* q;          // q_cols*Nxy source field per time column
    int q_cols;
    long long q_mstride;      // 0: all members share q; else member m's source field starts at q + m*q_mstride (hm_fwd_set_member_wells)
    int prd_mstride;          // 0: shared producer cells; else nPrd (member m's cells at prd_ind + m*nPrd)
    const int* prd_ind;       // nPrd flat cell indices
    const int* well_cells;    // nInj+nPrd flat cell indices of all wells (injectors first)
    // pressure scratch (fp64 always)
    double* TX;               // N*(Nx+1)*Ny   x-face transmissibilities
    double* TY;               // N*Nx*(Ny+1)   y-face transmissibilities
    double* G;                // N*Nx*Ny*Ny    inverse Schur complements of the block elimination
    double* yv;               // N*Nxy         forward-eliminated right-hand side
    double* P;                // N*Nxy         pressure
    double* Vx;               // N*(Nx+1)*Ny   x-face fluxes
    double* Vy;               // N*Nx*(Ny+1)   y-face fluxes
    // generic saturation scratch (dtype of the saturation arithmetic)
    void* coef;               // 6*N*Nxy : cE,cN,cC,cS,cW,fid
    void* fw;                 // N*Nxy
    float* comp;              // 2*N*Nxy (dtype = 32 plans): base and dS of the compensated float32 state (sat32.h)
    int* status;              // N
    int* nts;                 // N*nTime
    // conjugate-gradient pressure solver (grids with Ny > 128, or press_variant 9)
    double* cg_r;             // N*Nxy residual
    double* cg_p;             // N*Nxy search direction   (A p lives in yv, the iterate in P)
    int* n_cg;                // N*nTime iterations used
    double cg_rtol;           // stop at ||r|| <= rtol ||q||
    int cg_max_iter;
    const double* pin;        // N: SPD pin per member (coarse level of the two-level preconditioner only)
};

struct hm_nd;  // nested-dissection pressure solve (press_nd.hip): tables + factor + arena, built on first use
void hm_nd_free(hm_nd* n);

struct hm_fwd {
    hm_ctx* ctx = nullptr;
    hm_nd* nd = nullptr;
    FwdParams p{};
    int dtype = 64;
    int keep_history = 0;
    int press_variant = 0, sat_variant = 0;
    bool raw_state_exposed = false;  // a device pointer to the saturation was handed out: nothing about it is remembered from step to step
    bool raw_q_exposed = false;      // a device pointer to the source field was handed out: the plan no longer runs embedded (the inner plan has a source field of its own)
    bool raw_field_exposed = false;  // a device pointer to K / TX / ... was handed out (hm_fwd_device_ptr): no caching across time steps
    long long inputs_gen = 0;  // bumped by every call that can change K, wells, rates or kernel selection: results cached across time steps (press_nd.hip) die with it
    bool cg_lazy = true;  // CG work vectors not allocated yet
    size_t esz = 8;  // bytes per saturation element
    DevBuf Ky;     // y-permeability of an anisotropic run (hm_fwd_set_perm_y), else unallocated
    DevBuf K, por, q, prd_ind, TX, TY, G, yv, P, Vx, Vy, coef, fw, status, nts, perm_in, cg_r, cg_p, n_cg;
    // two-level CG preconditioner (allocated on first use): coarse transmissibilities, pin, restricted residual, coarse
    // correction, coarse scratch, coarse factor, per-member CG scalars and convergence flags
    DevBuf tl_TXc, tl_TYc, tl_pin, tl_rc, tl_yc, tl_yv, tl_G, tl_cgs, tl_done, tl_ndone, tl_z1, tl_dinv, tl_parts;
    int tl_n = 0;        // members the two-level buffers were sized for (they grow when a larger member block asks)
    int cg_precond = 0;  // 0 = two-level where it applies, 1 = Jacobi
    DevBuf S;      // keep_history ? N*(nTime+1)*Nxy : 2*N*Nxy (ping-pong)
    DevBuf prods;  // N*nTime*nPrd
    int cur = 0;   // time index whose saturation is "current" (row in history / ping-pong parity)
    EvTimer t_total, t_press, t_sat;
    long long n_press = 0, n_sat = 0;
    std::vector<double> q_host;
    std::vector<int> well_cells_host;
    std::vector<int> q_epoch;  // per column of q_host: first time step of the run of equal columns it belongs to (build_q)
    DevBuf well_cells;
    DevBuf comp;      // base / dS images of the generic fp32 sweeps (sat32.h), allocated on first use
    DevBuf retried;   // two ints: member-steps redone by the gated tiled sweep (hm_fwd_team_retries), by the slab sweep's redo launch (hm_fwd_slab_redos)
    DevBuf slab_wet;  // sat32s.hip: which slabs of which member hold water, two images in turn (written by the launch of step k, read by that of k + 1)
    int slab_wet_step = -1;       // time index whose launch may read the record
    long long slab_wet_gen = -1;  // inputs_gen the record belongs to
    int dbg_top_per_level = 1;    // hm_fwd_set_debug "top_per_level": 0 = levels 3 .. 0 of the 128 x 128 nested dissection always as one workgroup per member (default: a launch per level, a front per workgroup, for shards of fewer members than CUs)
    int dbg_top_deal = 1;         // hm_fwd_set_debug "top_deal": 0 = k_nd_top deals its trailing tiles round-robin over all waves (rounds 3-5)
    int dbg_small_wv = 0;         // hm_fwd_set_debug "small_wv": threads per member of the one-launch kernel of small grids (small.hip): 64 / 128 / 256, 0 = its default
    int dbg_sat_teams = -1;       // hm_fwd_set_debug "sat_teams": workgroups per member of the 128 x 128 fp64 sweep -- -1 automatic (2 / 4 where members x slabs <= CUs), 0 never, 2, 4
    int dbg_slab_margin = 1;      // hm_fwd_set_debug "slab_margin": 0 = the float32 slab sweep lets the neighbours of wet slabs sit out too (exercises its REDO launch)
    int dbg_team_rounds = 0;      // hm_fwd_set_debug "team_rounds": 1 = the slab teams in rounds of co-resident teams (round 4's form)
    DevBuf team_mem;  // synchronisation blocks of the multi-tile saturation sweep (sat128t.hip), allocated on first use
    int dbg_nd_force_fallback = -1, dbg_nd_cap = 0;  // hm_fwd_set_debug: test / experiment knobs of the larger grids' direct solver (press_nd.hip)
    long long nd_fallbacks = 0;  // member-steps the direct solver of the larger grids handed to the two-level CG (press_nd.hip: nd_check_and_fall_back)
    long long team_retries = 0, team_retries_seen = 0;  // time steps redone by the tiled sweep after a team gave up waiting
    // EMBEDDED GRIDS (forward.hip: embedded_inner).  A grid of at most 512 x 512 cells that no specialised kernel takes runs INSIDE a
    // 128 x 128, 256 x 256 or 512 x 512 plan (`inner`): its cells in the corner at the origin, the rest padded with cells of zero permeability.  This plan keeps
    // every buffer in the caller's layout; the inner plan shares its outputs per member (status, sub-step counts, producer series).
    hm_fwd* inner = nullptr;
    int emb = 0;                      // side of the inner plan's square grid (128, 256 or 512)
    bool is_inner = false;            // this plan IS the inner plan of another: status / nts / n_cg / prods belong to the outer plan
    int dbg_embed = 1;                // hm_fwd_set_debug "embed": 0 = never (the generic kernels on the grid as given)
    long long inner_K_gen = -1;       // inputs_gen the inner plan's permeability was embedded at
    int inner_S_step = -1;            // time index whose saturation the inner plan holds (-1: none)
    bool inner_V_dirty = false;       // face fluxes were set from the host (hm_fwd_set_field): the inner plan's are stale
    // LAZY FACE FLUXES (round 6, 128 x 128 nested dissection): the pressure step leaves P, TX, TY and does NOT launch k_nd_flux; the default
    // sweep (sat128r.hip) forms the scaled fluxes of its patch from them itself -- the same expression, the same bits -- and Vx / Vy are
    // materialised (nd128_materialize_fluxes: the k_nd_flux launch) only for whoever else reads them: another sweep kernel, hm_fwd_get_field
    // / set_field / device_ptr, the copy-out of an embedded grid.  Saves the write and the read of 0.53 MB per member and step.
    bool flux_pending = false;        // Vx, Vy of this plan are older than its P, TX, TY
    int dbg_lazy_flux = 1;            // hm_fwd_set_debug "lazy_flux": 0 = k_nd_flux behind every pressure step (round 5's form)
    bool fields_stale = false;        // P, Vx, Vy, TX, TY of this plan are older than the inner plan's (copied out when somebody asks)
    double Lx = 0, Ly = 0;            // as given to hm_fwd_create, with the well lists and rates (what the inner plan is created from)
    std::vector<int> inj_ind_host, prd_ind_host;
    std::vector<double> inj_rates_host, prd_rates_host;
    int inj_cols = 1, prd_cols = 1;
};

// Pointer to the saturation of (member 0, time index k) and the member stride in elements.
static inline void* fwd_S_ptr(hm_fwd* f, int k, long long* stride) {
    char* base = (char*)f->S.p;
    long long nxy = f->p.Nxy;
    if (f->keep_history) {
        *stride = (long long)(f->p.nTime + 1) * nxy;
        return base + (size_t)k * nxy * f->esz;
    }
    *stride = nxy;
    return base + (size_t)(k & 1) * f->p.N * nxy * f->esz;
}

// ---- kernels implemented in other translation units ------------------------------------------
// 128x128 fp64 specialisations (press128s.hip / sat128.hip).  Return 0 if launched, -1 if not applicable.
int launch_pressure_128s(hm_fwd* f, const void* S, long long S_stride, int k);  // MFMA panels, symmetric tile storage (default)
bool pressure_nd_applies(const FwdParams& p);
int launch_pressure_nd(hm_fwd* f, const void* S, long long S_stride, int k);     // nested dissection, 128 x 128 (press_nd.hip)
int prepare_pressure_nd(hm_fwd* f);   // tables + buffers before the first launch (outside a run's timed region)
int prepare_pressure_nd256(hm_fwd* f);
int prepare_pressure_nd512(hm_fwd* f);
bool pressure_nd_applies256(const FwdParams& p);
int launch_pressure_nd256(hm_fwd* f, const void* S, long long S_stride, int k);  // 256 x 256 (press_nd.hip compiled with -DND_LG=8)
bool pressure_nd_applies512(const FwdParams& p);
int launch_pressure_nd512(hm_fwd* f, const void* S, long long S_stride, int k);  // 512 x 512 (-DND_LG=9)
int launch_pressure_pcg(hm_fwd* f, const void* S, long long S_stride, int k);   // Jacobi-CG, any grid (press_pcg.hip)
bool pressure_two_level_applies(const FwdParams& p);
int launch_pressure_two_level(hm_fwd* f, const void* S, long long S_stride, int k);  // two-level CG, Ny = 128 c
int launch_saturation_128(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k);
bool sat128s_applies(const hm_fwd* f, int k);  // ... it takes step k of this plan
int launch_saturation_128s(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k);  // 128 x 128 fp64, small member shards: teams of 2 / 4 slab workgroups (sat128s.hip)
int nd128_materialize_fluxes(hm_fwd* f);  // Vx, Vy from P, TX, TY if they are pending (press_nd.hip)
int launch_saturation_128r(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k);  // fw in registers, scaled fluxes (sat128r.hip)
int launch_saturation_32s(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k);   // dtype = 32 plans, grids 128 / 256 / 512 wide (sat32s.hip)
int launch_saturation_128t(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k);   // grids of 128 x 128 tiles, fp64
bool small_forward_applies(const hm_fwd* f);                        // ... takes this plan (grid, LDS image, kernel variants)
int launch_small_forward(hm_fwd* f, int first_step, int n_steps);  // small grids: the whole run as one launch, a wave per member (small.hip)
int launch_saturation_256s(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k);   // grids 256 cells wide, fp64: slabs of 64 rows, fw in registers (sat256s.hip)
