/* hm_abi.h -- C ABI of libhm_amd.so: the MI355X (gfx950) drop-in for the data-parallel hot path of
 * patnr/HistoryMatching (reference @ 2024-11-08).
 *
 * The reference has no FFI of its own: its boundary is the Python call surface of
 * notebooks/HistoryMatch.py.  Every entry point below names the reference interface it replaces
 * (file:line into /root/reference).  Conventions:
 *   - plain pointers and sizes only; all host arrays are C-contiguous, ensemble axis first;
 *   - every function returns 0 on success, nonzero on failure; hm_last_error() then describes it;
 *   - `dtype` is 64 (double) or 32 (float) and is the type of the *saturation* arithmetic and of the
 *     update arithmetic; the pressure solve is always fp64 (in fp32 the TPFA flux
 *     T*(p_c-p_nb) has no correct digits at the permeability contrasts of 0.1+exp(5x));
 *   - host buffers handed in stay owned by the caller; outputs are written into caller buffers;
 *   - one host thread per context; work inside a context is stream-ordered.
 */
#ifndef HM_ABI_H
#define HM_ABI_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hm_ctx hm_ctx;   /* one per process per GPU */
typedef struct hm_fwd hm_fwd;   /* device-resident ensemble forward-model plan */
typedef struct hm_upd hm_upd;   /* device-resident ensemble-smoother update plan */
typedef struct hm_comm hm_comm; /* RCCL communicator of one rank (one process per GPU) */
typedef struct hm_iles hm_iles; /* localised iterative smoother: one weight matrix per local domain, on the device */

/* Timing/accounting of the last run (all times from HIP events on the context's stream). */
typedef struct hm_stats {
    double ms_total;        /* whole call, device side                                         */
    double ms_pressure;     /* sum over launches of the pressure (assemble+solve+flux) kernel   */
    double ms_saturation;   /* sum over launches of the saturation sweep kernel                 */
    double ms_update;       /* ensemble-smoother update kernels                                 */
    double mean_nts;        /* mean explicit sub-steps per member-step           */
    long long n_pressure_launches;
    long long n_saturation_launches;
    long long member_steps; /* N * nTime processed                                              */
    double mean_n_cg;       /* mean CG iterations per member-step (0 when the direct solver ran)   */
    double ms_comm;         /* update plans: collectives issued through hm_upd_all_reduce / hm_upd_run_comm */
} hm_stats;

/* ---- context ------------------------------------------------------------------------------ */
int         hm_create(int device_id, hm_ctx** out);
void        hm_destroy(hm_ctx* ctx);
/* HIP devices visible to this process (-1: the runtime could not be asked).  A launcher's ranks report it so that the
 * reader of a multi-GPU benchmark line can see every rank had the node's GPUs in view (notebooks/tools/utils.py:201-224 is the
 * reference's process pool this library's ranks replace). */
int   hm_device_count(void);
const char* hm_last_error(void);
int         hm_device_name(hm_ctx* ctx, char* buf, int buflen);   /* e.g. "gfx950:..."           */
int         hm_abi_version(void);
/* Raw copies between caller host buffers and device pointers handed out by hm_*_device_ptr /
 * hm_upd_reduce_buffer (host-staged form of the multi-rank update's reductions: CPU-side tests and ranks that share
 * one GPU; over distinct GPUs the reductions run in place on those device buffers through hm_comm_*). Synchronous. */
int         hm_copy_to_host(hm_ctx* ctx, void* dst_host, const void* src_device, long long bytes);
int         hm_copy_to_device(hm_ctx* ctx, void* dst_device, const void* src_host, long long bytes);

/* ---- ranks: replaces the process pool of utils.apply (notebooks/tools/utils.py:201-224) -------------------------
 * The reference's only parallel layer is an ordered process-pool map over independent members; its workers "don't
 * communicate back from child processes" (utils.py:226-228).  Here one process drives one GPU, the forward model needs
 * no exchange at all, and the update's cross-member sums are RCCL collectives issued by the library on
 * the context's stream, in place on the plan's device buffers (librccl.so.1 is dlopen'ed on first use; no PyTorch).
 * Rendezvous: rank 0 calls hm_comm_unique_id and hands the HM_COMM_ID_BYTES bytes to the other ranks by any host
 * channel (historymatching_amd/dist.py: a localhost socket); then every rank calls hm_comm_create.  RCCL does not
 * accept two ranks of one communicator on the same GPU ("Duplicate GPU detected"). */
#define HM_COMM_ID_BYTES 128
#define HM_COMM_SUM 0
#define HM_COMM_MAX 1
int  hm_comm_unique_id(char* id_out /* HM_COMM_ID_BYTES */);
int  hm_comm_probe(void);  /* 0 when librccl can be opened and bound on this rank; starts nothing (ranks other than 0 call this, not hm_comm_unique_id) */
int  hm_comm_create(hm_ctx* ctx, int rank, int world_size, const char* unique_id, hm_comm** out);
void hm_comm_destroy(hm_comm* c);
int  hm_comm_rank(hm_comm* c);
int  hm_comm_world_size(hm_comm* c);
/* Collectives on DEVICE buffers of the communicator's context, in place, asynchronous on the context's stream.
 * dtype: 64 double | 32 float | 1 int32 | 8 bytes.  all_gather: rank r's block of n_per_rank elements lies at
 * buf + r * n_per_rank on entry. */
int  hm_comm_all_reduce(hm_comm* c, void* buf, long long n, int dtype, int op /* HM_COMM_SUM | HM_COMM_MAX */);
int  hm_comm_all_gather(hm_comm* c, void* buf, long long n_per_rank, int dtype);
int  hm_comm_broadcast(hm_comm* c, void* buf, long long n, int dtype, int root);
int  hm_comm_group_start(hm_comm* c);   /* ncclGroupStart / ncclGroupEnd: one launch for the calls in between */
int  hm_comm_group_end(hm_comm* c);
int  hm_comm_sync(hm_comm* c);          /* stream synchronisation + RCCL asynchronous-error check */

/* ---- forward model: replaces utils.apply(comp1, ...) = forward_model --------------------------
 * Reference: forward_model  notebooks/HistoryMatch.py:383-387  (-> utils.apply tools/utils.py:155-242
 *            -> comp1 HistoryMatch.py:358-364 -> set_perm :160-164 -> model.sim :362 -> obs_model :212-213).
 * One call runs all N members for nTime steps.
 *   perm        N*Nxy   pre-permeability x (K = 0.1+exp(5x) applied on device, perm_transf :137-138)
 *                       when perm_is_transformed==0, else the permeability K itself (Kx=Ky, set_perm :164)
 *   wsat0       N*Nxy   initial saturation per member, or NULL = zeros (HistoryMatch.py:223, 1224-1227)
 *   inj_ind/prd_ind     flat cell indices ix*Ny+iy of the wells (model.xy2ind, HistoryMatch.py:209)
 *   inj_rates   nInj*inj_rate_cols (cols = 1: constant in time; = nTime: per step; HistoryMatch.py:189-193)
 *   porosity    Nxy or NULL (=1);  vw,vo,swc,sor fluid parameters (upstream defaults 1,1,0,0)
 *   wsats_out   return_history ? N*(nTime+1)*Nxy (row 0 = wsat0, HistoryMatch.py:224-225) : N*Nxy (final)
 *   prods_out   N*nTime*nPrd  saturation at producer cells after each step (HistoryMatch.py:363)
 *   status      N ints, 0 = ok, else HM_MEMBER_* below
 */
int hm_forward_batched(hm_ctx* ctx, int N, int Nx, int Ny, double Lx, double Ly,
                       const void* perm, int perm_is_transformed, const void* wsat0,
                       int nInj, const int* inj_ind, const double* inj_rates, int inj_rate_cols,
                       int nPrd, const int* prd_ind, const double* prd_rates, int prd_rate_cols,
                       double dt, int nTime, double vw, double vo, double swc, double sor,
                       const double* porosity, int dtype, int return_history,
                       void* wsats_out, void* prods_out, int* status_per_member, hm_stats* stats);

#define HM_MEMBER_OK            0
#define HM_MEMBER_BAD_PIVOT     1   /* non-positive pivot in the pressure factorisation (K<=0, NaN) */
#define HM_MEMBER_BAD_CFL       2   /* CFL sub-step count not finite / out of range               */
#define HM_MEMBER_NONFINITE     4   /* NaN/Inf in the saturation                                    */
#define HM_MEMBER_NO_CONVERGENCE 8  /* CG pressure solver hit max_iter before ||r|| <= rtol ||q||   */
#define HM_MEMBER_SYNC_TIMEOUT   16  /* a tile workgroup of the multi-tile saturation sweep gave up waiting for a neighbour */
#define HM_MEMBER_REDO_STEP      32  /* (internal, never left set) the slab sweep let dry slabs sit a step out and water reached one: the step is redone with every slab */

/* Device-resident form of the same path (what bench.py times; what ES-MDA/IES drivers chain).
 * Any grid up to 4096 cells a side.  The fast kernels exist for 128 x 128, 256 x 256 and 512 x 512 (and grids of 128-wide blocks); every
 * other grid up to 512 x 512 -- unless it is small enough for the one-launch kernel (about 21 x 21), has per-member wells, a porosity
 * field or anisotropic permeability -- runs EMBEDDED in the next of those squares, padded with cells of zero permeability (csrc/forward.hip):
 * same results (sweeps bit-identical, pressure a direct solve of the same system), every buffer of this API in the caller's Nx x Ny layout. */
int  hm_fwd_create(hm_ctx* ctx, int N, int Nx, int Ny, double Lx, double Ly,
                   int nInj, const int* inj_ind, const double* inj_rates, int inj_rate_cols,
                   int nPrd, const int* prd_ind, const double* prd_rates, int prd_rate_cols,
                   double dt, int nTime, double vw, double vo, double swc, double sor,
                   const double* porosity, int dtype, int keep_history, hm_fwd** out);
void hm_fwd_destroy(hm_fwd* f);
int  hm_fwd_set_inputs(hm_fwd* f, const void* perm, int perm_is_transformed, const void* wsat0); /* H2D */
/* Anisotropic K: the y-permeability per member (N*Nxy; `perm` above is then the x-permeability).  The simulator's K is
 * (2, Nx, Ny) (set via model.K, HistoryMatch.py:164, Optimise.py:69,888; the reference itself always stacks Kx = Ky).
 * Call after hm_fwd_set_inputs (perm_in scratch is shared); NULL = back to Kx = Ky. */
int  hm_fwd_set_perm_y(hm_fwd* f, const void* perm_y, int perm_is_transformed);
/* Device-resident chaining (forward -> update -> forward without PCIe): permeability input from a device buffer
 * (fp64 or fp32), initial saturation zero.  Asynchronous on the context's stream. */
int  hm_fwd_set_inputs_device(hm_fwd* f, const void* perm_dev, int perm_dtype, int perm_is_transformed);
/* Steps [first, first + n) of every member (ResSim.sim's time loop, HistoryMatch.py:362, for the whole ensemble).  Asynchronous on the
 * context's stream for grids up to 128 x 128.  256 x 256 / 512 x 512: the default pressure variant reads the members' status words back once
 * per time step (its hand-over of a failed direct solve to the two-level CG is a host decision, hm_fwd_nd_fallbacks); pressure variant 12 does
 * not and is asynchronous as well; other grids beyond 128 poll the CG's convergence on the host.  Small grids (the reference's default
 * 20 x 20; Nx Ny^2 up to about 9 000) run the whole call as ONE launch (small.hip): hm_stats then reports the step counts with zero
 * per-kernel times, the total in ms_total. */
int  hm_fwd_run(hm_fwd* f, int first_step, int n_steps);
int  hm_fwd_sync(hm_fwd* f, hm_stats* stats);
/* Member-steps the direct pressure solver of the 256 x 256 / 512 x 512 grids handed to the two-level CG since the plan was created: a
 * member whose elimination met a non-positive pivot, or whose fluxes missed the wells by more than 1e-4 of the largest rate (the
 * reference's sparse direct solve with partial pivoting, HistoryMatch.py:362, does not fail on such members either). */
long long hm_fwd_nd_fallbacks(hm_fwd* f);
/* Member-steps of the workgroup-team saturation sweeps that were redone by the single-workgroup tiled sweep since the plan was created: a team
 * gave up waiting for a neighbouring workgroup (CUs held by someone else).  Results are the same either way; a number other than 0 says the
 * run was slower than it should be.  (Slabs that sat a step out and were reached by water are counted by hm_fwd_slab_redos, not here.)
 * Synchronises the plan's stream. */
long long hm_fwd_team_retries(hm_fwd* f);
/* Member-steps the float32 slab sweep (sat32s) did twice: it lets slabs that are dry, with dry neighbours, sit a time step out; where the
 * frontier of denormal saturations ahead of the front reached such a slab within the step, the member's step is redone with every slab
 * (same launch form, bit-identical, ~the cost of one more member).  Synchronises the plan's stream. */
long long hm_fwd_slab_redos(hm_fwd* f);
/* Test and experiment knobs of a plan (not part of the reference's surface; nothing reads the environment).  Keys:
 *   "nd_force_fallback"  value = member index: on the larger grids that member is handed to the two-level CG at EVERY time step, whatever
 *                        its direct solve was like (exercises the hand-over deterministically); -1 (default) = off
 *   "nd_cap"             value = members per block of the larger grids' direct solver (0, the default: automatic -- the whole ensemble
 *                        where its buffers fit the device, else blocks within 64 GB); must be set before the plan's first run
 *   "team_rounds"        value = 1: the slab teams of the float32 sweep (sat32s) are launched in rounds of as many teams as are resident at
 *                        once (round 4's form) instead of one launch for the whole ensemble; 0 (default) = one launch
 *   "sat_teams"          value = workgroups per member of the 128 x 128 fp64 saturation sweep: -1 (default) automatic -- a member is a team of
 *                        two or four slab workgroups (sat128s) where members x slabs <= CUs (small shards of a strong-scaled ensemble), else
 *                        one workgroup (sat128r); 0 = never teams; 2, 4 = that many, provided the teams fit the CUs
 *   "top_per_level"      value = 0: levels 3 .. 0 of the 128 x 128 nested dissection as one workgroup per member whatever the shard size; 1
 *                        (default): for shards of fewer members than CUs a launch per level, one front per workgroup (bit-identical)
 *   "top_deal"           value = 0: k_nd_top (levels <= LO + 4 of the nested dissection) as in rounds 3-5 -- trailing tiles round-robin over
 *                        all waves, the next front's children fetched in one go, a full wait for memory at the end of a front; 1 (default):
 *                        trailing tiles in row-major runs on the waves that own no pivot tile, the fetch trickled behind the tile
 *                        updates, the update's stores draining beside the next front's gathers (bit-identical)
 *   "lazy_flux"          value = 0: the 128 x 128 nested dissection writes the face fluxes Vx, Vy behind every pressure step (k_nd_flux); 1
 *                        (default): it leaves P, TX, TY, the default sweep forms its fluxes from them, and Vx / Vy are materialised when
 *                        somebody else reads them (hm_fwd_get_field, another sweep kernel ...): bit-identical
 *   "slab_margin"        value = 0: the float32 slab sweep (sat32s) lets a slab sit a time step out as soon as IT is dry -- by default its
 *                        neighbours must be dry as well -- so the front reaches a sitting-out slab within a few steps; the border check flags
 *                        the member and the gated REDO launch repeats its step with every slab (hm_fwd_slab_redos counts): results unchanged
 *   "embed"              value = 0: never run the grid embedded in the next square (hm_fwd_create): the generic kernels on the grid as
 *                        given -- block elimination / conjugate gradients and the tiled sweep, the in-library cross-check; 1 (default)
 * Returns nonzero for an unknown key. */
int  hm_fwd_set_debug(hm_fwd* f, const char* key, long long value);
int  hm_fwd_get_outputs(hm_fwd* f, void* wsats_out, void* prods_out, int* status_per_member);     /* D2H */
/* All nTime steps from the inputs set, then sync + outputs in one call -- what forward_model(perms) -> [wsats, prods]
 * (HistoryMatch.py:383-387) needs.  From 256 MB of saturation history on, time index k of every member is copied to
 * `wsats_out` on a copy stream while step k runs, so the PCIe transfer of the history hides under the run. */
int  hm_fwd_run_to_host(hm_fwd* f, void* wsats_out, void* prods_out, int* status_per_member, hm_stats* stats);
/* Kernel selection (tests and diagnostics; 0/0 = the fastest applicable kernels).
 *   pressure  : 1 generic block elimination in LDS (any Ny <= 128) | 0: at 128 x 128 nested dissection (press_nd; 12 names it),
 *               at other grids with Ny = 128 the symmetric-tile MFMA block elimination (press128s; 13 names it at 128 x 128),
 *               7 its 16-wave form | 9 Jacobi-CG (any grid; beyond 128 the default is two-level CG where
 *               Ny = 128 c, Nx = c Nx_c, else Jacobi-CG)
 *   saturation: 1 generic (coefficient + fw images) | 2 streaming | 3 LDS-tiled | 0: at 128 x 128 the register/LDS-resident
 *               sweep (fp64: sat128r, fractional flow in registers and scaled fluxes -- 5 names its predecessor sat128 with the
 *               fw image in LDS, which is also what runs when two injectors share a band of 16 rows; all need
 *               uniform porosity and at most one well per 8 x 4 cell patch), else the
 *               tiled sweep from 64 x 64 cells up, the generic one below; fp64 grids 256 cells wide (Nx a multiple of 64:
 *               256^2): teams of workgroups, one per SLAB of 64 rows, the sweep of sat128r per slab (sat256s; 5 names the tile
 *               teams below instead); other fp64 grids of 128 x 128 tiles (512^2, 256 x 128 ...): teams of
 *               workgroups, one per tile (sat128t); dtype = 32 plans on grids 128 / 256 / 512 cells wide: the float32 register
 *               sweep on slabs of 16 384 cells (sat32s: teams of Nx Ny / 16 384 workgroups; at most one well per 4 x 8 patch);
 *               every dtype = 32 sweep carries the saturation as a compensated float32 pair (csrc/sat32.h: <= 1e-3 of the
 *               fp64 mode over a whole run at every grid); a team that gives up waiting for a neighbour -- CUs held by
 *               someone else -- has its time step redone by the tiled sweep; 4 = take that retry path every step, a test
 *               hook | pressure 11: two-level CG with the additive preconditioner
 *               instead of the two-grid cycle */
int  hm_fwd_set_variant(hm_fwd* f, int pressure_variant, int saturation_variant);
/* Conjugate-gradient pressure solver (always used when Ny > 128): relative residual target and iteration cap
 * (defaults 1e-12 and 40*max(Nx,Ny)+1000). */
int  hm_fwd_set_solver(hm_fwd* f, double rtol, int max_iter);
/* A different well configuration per member -- the batch of `npv(model, **params)` evaluations of the optimisation tutorial
 * (Optimise.py:112-125 run through utils.apply at Optimise.py:259,441,514,655): member m uses the source field
 * q_all[m] (q_cols x Nxy: injection rate > 0, production rate < 0 at the well cells, column k for time step k when
 * q_cols = nTime, else column 0) and gathers its producer series at prd_ind_all[m] (nPrd cells, nPrd as at creation).
 * The caller builds q_all from each member's well positions and rates exactly as hm_fwd_create does for the shared ones
 *  and has checked each member's rate balance; a member with q = 0 everywhere takes no step and is
 * flagged HM_MEMBER_BAD_CFL.  The 128 x 128 register-resident saturation kernels keep one shared well list: with per-member
 * wells the tiled/generic sweep runs instead. */
int  hm_fwd_set_member_wells(hm_fwd* f, const double* q_all /* N*q_cols*Nxy */, int q_cols, const int* prd_ind_all /* N*nPrd */);
/* Component hooks used by the parity tests (each maps to one listing of the cited paper):
 * run ONLY the pressure step / ONLY the saturation step of time index k on device state, and read
 * intermediate fields back. */
int  hm_fwd_pressure_only(hm_fwd* f, int k);
int  hm_fwd_saturation_only(hm_fwd* f, int k);
int  hm_fwd_get_field(hm_fwd* f, const char* name /* "S","P","Vx","Vy","TX","TY","K","nts" */, void* out);
int  hm_fwd_set_field(hm_fwd* f, const char* name /* "S","Vx","Vy" */, const void* in);
/* Raw device pointer of a named buffer ("S", "S_all", "prods": outputs; any name of hm_fwd_get_field).  Asking for an INPUT field
 * ("K", "TX", ...) tells the plan that it may be written behind its back from now on: the pressure step then keeps no results
 * across time steps any more (press_nd.hip, reuse of dry fronts) -- prefer hm_fwd_set_inputs_device for device-to-device chaining. */
void* hm_fwd_device_ptr(hm_fwd* f, const char* name);

/* Hardware self-test: D(16x16) = A(16x4) B(4x16) through one v_mfma_f64_16x16x4_f64 with the lane maps the
 * pressure kernel assumes (host buffers, row-major). */
int hm_debug_mfma_f64(hm_ctx* ctx, const double* A, const double* B, double* D);
/* Hardware self-test of the float32 fractional flow fw(s) = s^2 / (s^2 + (1 - s)^2) of dtype = 32 plans (csrc/fracflow.h: reciprocal seed,
 * quotient, ONE residual correction) against the IEEE division on all 2^32 operands s (oracle/ressim.py: frac_flow in float32; the
 * reference's own is fp64, HistoryMatch.py:362 -> ResSim).  out[0] = operands with |s| < 2^62 whose result differs in any bit (claimed: 0),
 * out[1] = operands that differ at all (|s| >= 6.5e18 only). */
int hm_debug_fracflow32_check(hm_ctx* ctx, unsigned long long* out /* 2 */);
/* The same for the fp64 fractional flow of the saturation sweeps (csrc/fracflow.h: seven instructions -- reciprocal seed, one cubic refinement,
 * quotient, one residual correction; what ResSim.sim's fw = mw / (mw + mo) is in NumPy, HistoryMatch.py:362), against the compiler's IEEE division
 * on 2^34 + 2.8e8 operands: dense over [0, 1 + 2^-9), clustered next to 0, 1/2 and 1, and every binade down to the denormals.  out[0] = operands
 * whose result differs in any bit (claimed: 0), out[1] = operands tried. */
int hm_debug_fracflow64_check(hm_ctx* ctx, unsigned long long* out /* 2 */);
/* Symbolic phase of the nested-dissection pressure solve (press_nd.hip; replaces the sparse direct solve inside
 * ResSim.sim, notebooks/HistoryMatch.py:362): the elimination tree of the Nx x Ny grid as the kernels use it.
 * Runs on the host, no device needed.  info[0..3] = fronts, cell entries, factor doubles and arena doubles per member;
 * info[4..6] = largest packed update of levels 8..10 (doubles); info[7] = assembly-recipe blocks of 256 int16 (csrc/nd.h);
 * info[8..18] = 16 * boundary tiles + pivot tiles per level; info[19] = ints per front record.
 * fronts (info[0] * info[19] ints), cells (info[1] ints), cpos (2 * info[1] shorts), rec (256 * info[7] shorts) may be NULL (size query). */
int hm_debug_nd_tables(int Nx, int Ny, long long* info /* 64 */, int* fronts, int* cells, short* cpos, short* rec);

/* ---- ensemble-smoother update: replaces ens_update0 -------------------------------------------
 * Reference: ens_update0  notebooks/HistoryMatch.py:578-586  (center: tools/utils.py:10-28).
 *   E N*M, obs_ens N*n_obs, obs n_obs, perturbs N*n_obs, decorr n_obs*n_obs  ->  E_out N*M
 * Evaluated in the minimum-flop association  E + (D C^-1)(S^T X)  (SURVEY.md 8a row a8). */
int hm_es_update(hm_ctx* ctx, int N, int M, int n_obs, const void* E, const void* obs_ens,
                 const void* obs, const void* perturbs, const void* decorr, int dtype,
                 void* E_out, hm_stats* stats);

/* Reference: center(E, axis=0, rescale)  notebooks/tools/utils.py:10-28:  X = E - mean (optionally times
 * sqrt(N/(N-1))), mean (M).  Standalone form; ens_update0* fuse the centring into their contractions. */
int hm_center(hm_ctx* ctx, int N, int M, const void* E, int dtype, int rescale, void* X_out, void* mean_out);

/* Reference: ens_update0_loc  notebooks/HistoryMatch.py:774-797;  taper is M*n_obs (HistoryMatch.py:863),
 * cutoff is the 1e-2 of HistoryMatch.py:786. */
int hm_es_update_loc(hm_ctx* ctx, int N, int M, int n_obs, const void* E, const void* obs_ens,
                     const void* obs, const void* perturbs, const void* decorr, const void* taper,
                     double cutoff, int dtype, void* E_out, hm_stats* stats);

/* Device-resident / sharded form: rows [row0, row0+N_local) of an N-member ensemble live on this GPU.
 * The cross-rank reductions are exposed as four sum-reduce buffers, all-reduced between the phases
 * (hm_upd_all_reduce: RCCL from the library; or host-staged through hm_upd_reduce_buffer + hm_copy_*):
 *   phase 0: local column sums                      -> buffers 0 (E: M values, dtype) and 1 (obs_ens: n_obs, fp64)
 *   phase 1: (after all-reduce of 0,1) S, D, and the local Gram pair
 *                                                   -> buffers 2 (X^T S: M*n_obs, dtype) and 3 (S^T S: n_obs^2, fp64)
 *   phase 2: (after all-reduce of 2,3) C^-1 (or the per-element local analyses), E_out = E + (D C^-1) Gx  (row-local)
 * Everything of size <= N x n_obs is held in fp64 whatever `dtype` is; only the two contractions over M use `dtype`. */
int   hm_upd_create(hm_ctx* ctx, int N_total, int N_local, int M, int n_obs, int dtype, int localized, hm_upd** out);
void  hm_upd_destroy(hm_upd* u);
int   hm_upd_set_inputs(hm_upd* u, const void* E_local, const void* obs_ens_local, const void* obs,
                        const void* perturbs_local, const void* decorr, const void* taper /* or NULL */, double cutoff);
/* E (N x M) = x0 (M) + W (N x N) X0 (N x M): re-composition of the ensemble from subspace weights, the 2 N^2 M flop
 * step of the iterative ensemble smoother (IES, notebooks/HistoryMatch.py:921, 944).  Host buffers. */
int   hm_recompose(hm_ctx* ctx, int N, int M, const void* W, const void* X0, const void* x0, int dtype, void* E_out);
/* Prior sampler for grids the reference's dense sampler (geostat.gaussian_fields, notebooks/tools/geostat.py:86-99: Cholesky
 * of the Nxy x Nxy covariance) cannot reach: the Gaussian-variogram covariance is separable, Cov = Cx (x) Cy, so
 * X_n = Ux^T Z_n Uy with Ux, Uy the upper Cholesky factors per axis and Z_n (Nx x Ny) the caller's standard normals
 * (host RNG: seeds replay).  N fields (N x Nx x Ny) to a host buffer and/or a device buffer of this context (e.g. the
 * permeability input of a forward plan: hm_fwd_set_inputs_device); either may be NULL, not both. */
int   hm_sample_kron(hm_ctx* ctx, int N, int Nx, int Ny, const double* Ux /* Nx*Nx */, const double* Uy /* Ny*Ny */,
                     const double* Z /* N*Nx*Ny */, double* X_out, void* X_device);
/* Self-test hook: W = inv(G + ridge I) for one SPD matrix of order n (multiple of 16, <= 256) through the matrix-core
 * inverse used for C = S^T S + (N-1) I (HistoryMatch.py:585-586).  Host buffers. */
int   hm_debug_spd_inverse(hm_ctx* ctx, int n, const double* G, double ridge, double* W);
/* Self-test hook: A_T (n x N, fp32) = (X inv(G + ridge I))^T through the block L D L^T factorisation and the gain kernel the
 * fused analysis step uses for its gain D0 B^-1 (n a multiple of 16, <= 176; X: N x n).  Host buffers. */
int   hm_debug_ldl_gain(hm_ctx* ctx, int n, int N, const double* G, double ridge, const double* X, float* A_T);
/* Device-resident chaining: ensemble and/or simulated observations from device buffers (NULL = keep); hm_upd_swap makes
 * the last posterior the next prior (ES-MDA / iterative smoothers, HistoryMatch.py:906-959). */
int   hm_upd_set_inputs_device(hm_upd* u, const void* E_dev, int E_dtype, const void* obs_ens_dev, int obs_dtype);
int   hm_upd_swap(hm_upd* u);
int   hm_upd_phase(hm_upd* u, int phase);
/* Localised plans over several ranks: the per-element solves of phase 2 (HistoryMatch.py:783-793; "<-- can multiprocess
 * this map" :795) are column-sharded -- rank r solves the state elements [r*chunk, (r+1)*chunk), chunk = ceil(M/world) --
 * and phase 2 stops after them; the weights W^T (reduce buffer 4: world*chunk rows of n_obs) are all-gathered, then
 * phase 3 applies them to this rank's members. */
int   hm_upd_set_column_shard(hm_upd* u, int rank, int world_size);
/* The collective that follows `after_phase` (0: buffers 0,1; 1: buffers 2,3; 2: all-gather of buffer 4), queued on the
 * context's stream behind the phase, no host synchronisation. */
int   hm_upd_all_reduce(hm_upd* u, hm_comm* c, int after_phase);
/* The whole analysis step of a row-sharded plan over the ranks of `c`: phases and collectives in stream order. */
int   hm_upd_run_comm(hm_upd* u, hm_comm* c);
/* All three phases of a plan that holds every member (N_local == N_total), no reduction points.  fp32 plans run the
 * second-generation matrix-core kernels here: the Kalman form D0 (Yc^T Yc + (N-1) R)^-1 Yc^T X of the same update
 * (HistoryMatch.py:578-586 with R = (decorr decorr^T)^-1), LDS-DMA-staged contractions over the state, and for the N x n_obs
 * quantities one launch for centring + Gram matrix and one for the block L D L^T factorisation of the n_obs x n_obs matrix
 * with the gain D0 B^-1 computed beside it (n_obs a multiple of 16 up to 176; otherwise matrix-core inverse + product). */
int   hm_upd_run(hm_upd* u);
/* Tuning / test switches (defaults in brackets): "use_mfma" [1] | 0 = generic fp32 GEMMs; "kalman_form" [1] | 0 = decorrelated
 * form; "ldl_gain" [1] = factorisation + gain in one launch | 2 = as two kernels | 0 = explicit inverse + product;
 * "fused_front" [1] | 0 = centring and Gram matrix as two kernels; "mfma_inverse" [1] | 0; "overlap" [0] | 1 = the fp64 chain
 * on a second stream (measured slower); kernel-variant selectors "gxt_dma", "gxt_chunk", "apply_variant", "small_inverse", ... */
int   hm_upd_set_option(hm_upd* u, const char* name, int value);
void* hm_upd_reduce_buffer(hm_upd* u, int which /*0..4*/, long long* n_elems, int* elem_bytes); /* device pointer */
/* Waits for the plan's stream.  THE OUTPUT OF A RUN IS VALID ONLY AFTER hm_upd_sync HAS RETURNED 0: a fused run whose one-launch
 * factorisation stalled (another stream or process held the CUs) is redone here through the two-kernel form, and only the LAST run is
 * redone -- work that reads E_out (hm_fwd_set_inputs_device, a further hm_upd_run after hm_upd_swap) must be queued after this call,
 * as update.es_mda_device and dist.es_mda_sharded do.  hm_upd_chain_fallbacks counts such redone steps of the plan. */
int   hm_upd_sync(hm_upd* u, hm_stats* stats);
int   hm_upd_chain_fallbacks(hm_upd* u);
int   hm_upd_get_output(hm_upd* u, void* E_out_local);
void* hm_upd_device_ptr(hm_upd* u, const char* name);   /* "E","E_out","obs_ens","perturbs" */

/* ---- localised iterative ensemble smoother, partitioned: replaces ILES ---------------------------------------------
 * Reference: ILES  notebooks/HistoryMatch.py:1007-1064 (one N x N weight matrix per state element), in the batched form
 * the reference points at (HistoryMatch.py:802-804; notebooks/tools/localization.py:95-145 rectangular_partitioning): the
 * state elements batch_index[batch_offsets[b] .. batch_offsets[b+1]) share one weight matrix W_b and one taper row
 * taper_b[b] (n_obs).  One element per batch = the reference's algorithm.  fp64.
 *   hm_iles_create   uploads the prior (N x M), centres it (X0, x0: HistoryMatch.py:1017), sets every W_b = I
 *   hm_iles_compose  E = x0 + W_b X0 per batch (HistoryMatch.py:1021-1025) -> host buffer (or NULL: stays on the device, "E")
 *   hm_iles_step     one Gauss-Newton step of every W_b (HistoryMatch.py:1035-1058) from S = center(Eo decorr) and
 *                    D = (obs - Eo - perturbs) decorr (N x n_obs, formed by the caller: HistoryMatch.py:1031-1032)            */
int   hm_iles_create(hm_ctx* ctx, int N, int M, int n_obs, int B, const int* batch_offsets /* B+1 */, const int* batch_index /* M */,
                     const double* taper_b /* B*n_obs */, double cutoff, const double* prior_ens /* N*M */, hm_iles** out);
void  hm_iles_destroy(hm_iles* p);
int   hm_iles_compose(hm_iles* p, double* E_out /* N*M or NULL */);
int   hm_iles_step(hm_iles* p, const double* S, const double* D, double xstep);
/* "blocked" = 1: the step as a blocked elimination over many workgroups per batch (the default from N = 256 members on; needs
 * N <= 1024), 0: one workgroup per batch (the default below that). */
int   hm_iles_set_option(hm_iles* p, const char* name, int value);
int   hm_iles_get_weights(hm_iles* p, int batch, double* W_out /* N*N */);
void* hm_iles_device_ptr(hm_iles* p, const char* name /* "E","W","X0" */);

#ifdef __cplusplus
}
#endif
#endif /* HM_ABI_H */
