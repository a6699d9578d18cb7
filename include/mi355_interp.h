/*
 * mi355_interp.h -- C ABI of libmi355interp.so, the MI355X (gfx950) native
 * batched linear-interpolation path.
 *
 * This is the drop-in boundary for the hot path of
 * kyle-wedgwood/ArmadilloCUDALinearInterpolation.  The reference has no FFI of
 * its own (it is one nvcc-built C++ program); what it has is
 *   - the operator interface  AbstractNonlinearProblem::ComputeF(const
 *     arma::vec&, arma::vec&)           (AbstractNonlinearProblem.hpp:11), and
 *   - the device stages that EventDrivenMap::ComputeF launches
 *     (EventDrivenMap.cu:154-240).
 * Every entry point below names the reference code it replaces.  The
 * Armadillo-facing C++ signatures that sit on top of this ABI are in
 * include/mi355_arma.hpp; the binding a reference maintainer would add is
 * shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every function returns an mi_status (0 = MI_OK) and
 *     never calls exit() (the reference's CUDA_CALL macros do,
 *     EventDrivenMap.cu:18-54; the C++ wrapper restores that behaviour).
 *   - mi_last_error(ctx) returns the message of the last failure on that
 *     context (or on the calling thread when ctx is NULL).
 *   - a context is not thread-safe (one host thread per context, like the
 *     reference's single-threaded driver); use one context per thread/stream.
 *   - "dev" pointers are HBM addresses valid on the context's device; "host"
 *     pointers are ordinary host memory.  Device entry points are
 *     asynchronous on the context's stream and perform no allocation, copy or
 *     synchronisation (safe to capture into a hipGraph).
 *   - fp64 blend shared by every interp entry point (Armadillo
 *     interp1_helper_linear semantics, see oracle/interp_oracle.c):
 *        l = largest node with X[l] <= q,  r = min(l+1, n-1)
 *        a = q - X[l];  b = X[r] - q;  w = a > 0 ? a/(a+b) : 0
 *        out = (1-w)*Y[l] + w*Y[r]          (every op rounded, no FMA)
 *     q < X[0] or q > X[n-1] -> extrap_val;  q = NaN -> NaN.
 */
#ifndef MI355_INTERP_H
#define MI355_INTERP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 3): + mi_group_rccl_ranks, mi_ctx_set_interp2_path, mi_grid2_reserve, mi_grid2_info;
 *              - mi_debug_sweep_timing (the sweep's phase stamps live in scripts/ harnesses now)
 * 4 (round 4): - mi_ctx_set_interp2_path, mi_grid2_reserve, MI_INTERP2_* (the call-wide cell ordering of the bilinear path
 *                was bit-identical and 24 % slower than the direct kernel; it is a recorded experiment now,
 *                scripts/exp_interp2_ordered.hpp, profiles/r03_config3_ordered_*); mi_grid2_info reports the table size only;
 *              + mi_edm_set_kernel_choice (replaces two environment hooks), mi_edm_debug_counters,
 *                mi_group_set_gather_chunks */
#define MI355_INTERP_ABI_VERSION 4

typedef int mi_status;
enum {
    MI_OK = 0,
    MI_ERR_INVALID_ARG = 1,   /* NULL pointer, bad size, bad flag               */
    MI_ERR_GRID = 2,          /* grid not sorted/unique, NaN node, < 2 nodes    */
    MI_ERR_HIP = 3,           /* a HIP runtime call failed (message has detail) */
    MI_ERR_NOMEM = 4,
    MI_ERR_NO_DEVICE = 5
};

typedef struct mi_ctx mi_ctx;       /* one per device (+ stream)            */
typedef struct mi_grid1 mi_grid1;   /* HBM-resident 1-D table               */
typedef struct mi_grid2 mi_grid2;   /* HBM-resident 2-D table               */
typedef struct mi_edm mi_edm;       /* EventDrivenMap device state          */
typedef struct mi_timer mi_timer;   /* pair of HIP events on the ctx stream */

/* ---- context ---------------------------------------------------------- */
/* Replaces the implicit "device 0, NULL stream" of the reference
 * (EventDrivenMap.cu:80-94 allocates on the current device). */
mi_status mi_ctx_create(int device, mi_ctx** out);
mi_status mi_ctx_destroy(mi_ctx* ctx);
/* stream: a hipStream_t (NULL = the device's default stream). */
mi_status mi_ctx_set_stream(mi_ctx* ctx, void* stream);
/* give the context a non-blocking stream of its own (created here, destroyed with the context): lets work of
 * several contexts on one device overlap without the caller touching the HIP API */
mi_status mi_ctx_own_stream(mi_ctx* ctx);
mi_status mi_ctx_synchronize(mi_ctx* ctx);
const char* mi_last_error(const mi_ctx* ctx);
int mi_abi_version(void);
/* Test hook: number of caller host ranges the library currently keeps page-locked on behalf of host-convenience calls
 * in flight (mi_interp1_f64_host, mi_interp2_f64_host); 0 once every such call has returned, on success or error. */
size_t mi_debug_pinned_ranges(void);
/* Hint about the order of the query vectors handed to mi_interp1_f64_dev on this context.  Unordered queries
 * over a table larger than L2 are processed by a "region sweep" kernel (workgroup-local ordering by table region
 * in LDS; results keep the caller's order), ordered/clustered ones by the plain streaming kernel.  AUTO decides on
 * the device with a 1024-sample probe (no host synchronisation): the first call launches both kernels gated on the
 * probe's flag, later calls launch only the kernel the previous probe's verdict predicts (read from a pinned host
 * int, never waited for; both kernels are correct on any input, so a stale verdict only costs time).  The other
 * values skip the probe; changing the value forgets the verdict.  Results never depend on any of this. */
#define MI_QUERIES_AUTO    0
#define MI_QUERIES_RANDOM  1
#define MI_QUERIES_ORDERED 2
mi_status mi_ctx_set_query_order(mi_ctx* ctx, int order);
/* name / CU count of the context's device (for bench reports) */
mi_status mi_ctx_device_info(mi_ctx* ctx, char* name, size_t name_len, int* compute_units,
                             size_t* hbm_bytes);

/* ---- timing (HIP events recorded on the context's stream) -------------- */
mi_status mi_timer_create(mi_ctx* ctx, mi_timer** out);
mi_status mi_timer_destroy(mi_timer* t);
mi_status mi_timer_start(mi_timer* t);
mi_status mi_timer_stop(mi_timer* t);
/* blocks until the stop event has completed */
mi_status mi_timer_elapsed_ms(mi_timer* t, float* ms);

/* ---- 1-D tables ---------------------------------------------------------
 * General grid: explicit abscissae, the arma::interp1(X, Y, XI, YI) shape.
 * x/y are HOST pointers (n doubles each); the table is uploaded once and
 * stays resident in HBM: as Y only when a closed form (linspace-like grids)
 * reproduces every abscissa bit for bit, else as interleaved {x,y} nodes.
 * flags: MI_GRID_SANITISE    sort + de-duplicate X first (what arma::interp1
 *                            does unless the method is "*linear"); without it
 *                            X must be strictly increasing or MI_ERR_GRID.
 *        MI_GRID_DEVICE_PTRS x/y are device pointers (copied to the host once
 *                            for validation and index construction).
 */
#define MI_GRID_SANITISE     0x1u
#define MI_GRID_DEVICE_PTRS  0x2u
mi_status mi_grid1_create(mi_ctx* ctx, const double* x, const double* y, size_t n,
                          unsigned flags, mi_grid1** out);
/* Implicit uniform grid: X_i := fma(i, dx, x0), i = 0..n-1, dx > 0.
 * Only Y is stored (half the table bytes of the general form). */
mi_status mi_grid1_create_uniform(mi_ctx* ctx, double x0, double dx, const double* y, size_t n,
                                  unsigned flags, mi_grid1** out);
mi_status mi_grid1_destroy(mi_grid1* g);
/* number of nodes after sanitising; table mode chosen at build time
 * (0 = Y only, abscissae from a closed form -- declared by
 *      mi_grid1_create_uniform or detected on an explicit grid,
 *  1 = explicit {x,y} nodes + analytic guess and a bounded walk,
 *  2 = explicit {x,y} nodes + bucket index,
 *  3 = explicit {x,y} nodes + centred analytic guess: the grid stays within
 *      one cell of a straight line, the bracket is one comparison away) */
mi_status mi_grid1_info(const mi_grid1* g, size_t* n_nodes, int* mode, size_t* table_bytes);

/* ---- 1-D interpolation: the hot path -----------------------------------
 * yq[i] = interp(grid, xq[i]), i < nq.  Device pointers, asynchronous.
 * Algorithmic HBM bytes per query: 8 (xq) + 8 (yq); the table is read through
 * L2 / Infinity Cache.
 * Generalises RestrictKernel's two-point blend (EventDrivenMap.cu:769-785) to
 * a tabulated grid with a gather index, as BASELINE.json's north_star asks.
 */
mi_status mi_interp1_f64_dev(mi_ctx* ctx, const mi_grid1* g, const double* xq_dev, double* yq_dev,
                             size_t nq, double extrap_val);
/* Host convenience used by the arma::vec wrapper: uploads xq, runs, downloads
 * (synchronous). */
mi_status mi_interp1_f64_host(mi_ctx* ctx, const mi_grid1* g, const double* xq, double* yq,
                              size_t nq, double extrap_val);
/* One-shot arma::interp1(X,Y,XI,YI,"linear",extrap) equivalent on host
 * pointers (sanitises X, builds a temporary table, synchronous). */
mi_status mi_interp1_f64(mi_ctx* ctx, const double* x, const double* y, size_t n, const double* xq,
                         double* yq, size_t nq, double extrap_val);

/* ---- 2-D tables and scattered bilinear interpolation ------------------
 * z is column-major ny x nx, i.e. arma::mat(ny, nx).memptr(): z[i + j*ny] =
 * Z(y_i, x_j).  Blend along y inside the two bracketing columns, then along
 * x (oracle/interp_oracle.c orc_interp2_bilinear).  No counterpart in the
 * reference (BASELINE.json config 3). */
/* flags: MI_GRID_DEVICE_PTRS (z, x, y are device pointers); MI_GRID2_COMPACT keeps the resident table at 2x the
 * input bytes (column pairs) instead of the default 4x (quad cells: one 64-B sector per query, ~8 % faster). */
#define MI_GRID2_COMPACT 0x4u
mi_status mi_grid2_create(mi_ctx* ctx, const double* x, size_t nx, const double* y, size_t ny,
                          const double* z, unsigned flags, mi_grid2** out);
mi_status mi_grid2_create_uniform(mi_ctx* ctx, double x0, double dx, size_t nx, double y0, double dy,
                                  size_t ny, const double* z, unsigned flags, mi_grid2** out);
mi_status mi_grid2_destroy(mi_grid2* g);
/* resident bytes of the table */
mi_status mi_grid2_info(const mi_grid2* g, size_t* table_bytes);
mi_status mi_interp2_f64_dev(mi_ctx* ctx, const mi_grid2* g, const double* xq_dev,
                             const double* yq_dev, double* zq_dev, size_t nq, double extrap_val);
mi_status mi_interp2_f64_host(mi_ctx* ctx, const mi_grid2* g, const double* xq, const double* yq,
                              double* zq, size_t nq, double extrap_val);

/* ---- the reference's own interpolation ----------------------------------
 * Replaces RestrictKernel (EventDrivenMap.cu:769-785, launch :205-206):
 *   x_k = -L + 2L/ngrid * ind_k ;  out = x0 + (T-t0)*(x1-x0)/(t1-t0)   (fp32)
 * Arrays are [spike][realisation] (index m*R + r), n = S*R elements.  `out`
 * may alias t0 (the reference works in place).  Device pointers.
 */
mi_status mi_restrict_f32_dev(mi_ctx* ctx, const float* t0, const uint16_t* i0, const float* t1,
                              const uint16_t* i1, float final_time, float half_length,
                              uint32_t ngrid, float* out, size_t n);
/* Host convenience (what the arma::fvec wrapper calls): uploads the four event arrays, runs, downloads
 * (synchronous).  out may alias t0. */
mi_status mi_restrict_f32_host(mi_ctx* ctx, const float* t0, const uint16_t* i0, const float* t1,
                               const uint16_t* i1, float final_time, float half_length, uint32_t ngrid,
                               float* out, size_t n);
/* Replaces CountRealisationsKernel + realisationReductionKernelBlocks
 * (EventDrivenMap.cu:787-824): mean over accepted realisations, per spike.
 * x: f32[nspikes*nreal] ([spike][realisation]), accept: u32[nreal] (0/1).
 * mean_dev: f32[nspikes]; count_dev: u32[1]; sums_dev (optional, may be
 * NULL): f64[MI_EDM_PARTIAL_LEN(nspikes)], the partial block
 *   [ sum_m over the accepted realisations, m < nspikes | accepted count | x0_m ]
 * whose element-wise sum over the shards of a multi-GPU run is what
 * mi_edm_residual_from_sums turns into the mean.
 * quirk != 0 reproduces the reference's accept[0] clobber
 * (EventDrivenMap.cu:800-802 overwrites accept[0] with the count, :817 then
 * tests accept[index]==1): realisation 0 is left out of the sums -- and
 * counted in the divisor (:822) -- unless count == 1, when it is summed whatever
 * its flag was.  In the partial block the sums never contain realisation 0 when
 * quirk is set and x0_m holds its restricted position (0 without quirk), so the
 * count == 1 rule can be applied to the total.  `accept` is never modified. */
#define MI_EDM_PARTIAL_LEN(nspikes) (2 * (nspikes) + 1)
mi_status mi_masked_mean_f32_dev(mi_ctx* ctx, const float* x, const uint32_t* accept, size_t nreal,
                                 size_t nspikes, int quirk, float* mean_dev, uint32_t* count_dev,
                                 double* sums_dev);
/* Fused Restrict + masked mean: one pass over the 4 event arrays, no
 * intermediate (what ComputeF uses).  restricted_dev may be NULL. */
mi_status mi_restrict_mean_f32_dev(mi_ctx* ctx, const float* t0, const uint16_t* i0, const float* t1,
                                   const uint16_t* i1, const uint32_t* accept, float final_time,
                                   float half_length, uint32_t ngrid, size_t nreal, size_t nspikes,
                                   int quirk, float* restricted_dev, float* mean_dev,
                                   uint32_t* count_dev, double* sums_dev);

/* ---- EventDrivenMap: the residual evaluation ------------------------------
 * Replaces class EventDrivenMap (EventDrivenMap.hpp:11-121,
 * EventDrivenMap.cu:57-404): lift -> evolve -> restrict -> average.
 */
typedef struct mi_edm_params {
    float vth, a1, a2, b1, b2, I, L;   /* parameters.hpp:1-8                     */
    double newton_tol;                 /* parameters.hpp:9  (tol, a double)      */
    uint32_t newton_max_iter;          /* counterMax: undefined in the reference
                                          (EventDrivenMap.cu:564); default 100   */
    uint32_t n_spikes;                 /* parameters.hpp:12 noSpikes (<= 8)      */
    float time_horizon;                /* parameters.hpp:15                      */
    uint32_t n_grid;                   /* mNoThreads (EventDrivenMap.cu:70), <= 1024 */
    uint32_t n_real;                   /* mNoReal                                */
    float beta_mean;                   /* (*pParameters)[0]                      */
    float beta_stddev;                 /* mParStdDev (EventDrivenMap.cu:105)     */
    uint64_t seed;                     /* mSeed                                  */
    int math_mode;                     /* MI_EDM_MATH_EXACT or MI_EDM_MATH_FAST  */
    int mean_quirk;                    /* default 1: average exactly as the
                                          reference does (EventDrivenMap.cu:800-802,
                                          :817,:822: realisation 0 is left out of the
                                          sum but counted in the divisor, unless only
                                          one realisation was accepted); 0: the true
                                          mean over the accepted realisations        */
    uint32_t max_events;               /* hard bound on events per realisation
                                          (termination guarantee; default 2^20)  */
    uint32_t real_offset;              /* global index of this shard's first
                                          realisation (multi-GPU sharding; only
                                          enters the per-neuron beta draw)       */
    uint32_t dedup_identical;          /* opt-in, default 0: with beta_stddev == 0
                                          every realisation is the same computation
                                          (the draw is the only per-realisation
                                          input, EventDrivenMap.cu:179,196); evolve
                                          one and replicate its events to all n_real
                                          rows.  Outputs are bit-identical to the
                                          full evolution; ignored when stddev != 0 */
} mi_edm_params;
#define MI_EDM_MATH_EXACT 0   /* software exp/log, bit-identical to oracle/edm_oracle.c */
#define MI_EDM_MATH_FAST  1   /* v_exp_f32 / v_log_f32 hardware transcendentals          */

void mi_edm_default_params(mi_edm_params* p);
mi_status mi_edm_create(mi_ctx* ctx, const mi_edm_params* p, mi_edm** out);
mi_status mi_edm_destroy(mi_edm* e);
/* setters of EventDrivenMap.hpp:27-51 arrive as a new parameter block */
mi_status mi_edm_set_params(mi_edm* e, const mi_edm_params* p);
/* Tuning / test knob; every choice gives bit-identical results (tests/test_edm_gpu.py runs each case under all of them).
 * waves_per_realisation: 0 = by realisation count (default: a workgroup of four waves per realisation for 48 up to
 * one realisation per CU, one wave per realisation otherwise), 1 or 4 = that form always.  uniform_division: 1 (default) = the
 * exact quotient in three or five operations (multiply by the divisor's rounded reciprocal + one or two correction steps;
 * one only where the device has proved it exact for that divisor) where a divisor is the same for the whole launch,
 * 0 = IEEE division everywhere. */
mi_status mi_edm_set_kernel_choice(mi_edm* e, int waves_per_realisation, int uniform_division);
/* ComputeF (EventDrivenMap.cu:154-240).  z: host, n_spikes doubles (c, Z1..);
 * f: host, n_spikes doubles.  partial (optional, may be NULL): host,
 * MI_EDM_PARTIAL_LEN(n_spikes) doubles receiving this device's partial block
 * [sums | accepted count | x0] (see mi_masked_mean_f32_dev) so that a caller
 * sharding realisations over GPUs can all-reduce them (the shard with
 * real_offset == 0 holds realisation 0 of the ensemble and is the only one that
 * applies mean_quirk); when partial is non-NULL f is still the single-device
 * residual. */
mi_status mi_edm_compute_f(mi_edm* e, const double* z, double* f, double* partial);
/* The same in two halves: _begin enqueues the whole evaluation on the context's stream and returns (with 600 or more
 * realisations it first waits for the lift kernel -- microseconds -- because the evolve kernel's LDS is sized by the
 * lift profile's live slices; the evolve kernel itself, which is > 99.9 % of the evaluation, is never waited for),
 * _end waits for it and forms f (and partial).  Independent evaluations -- the columns of NewtonSolver's
 * finite-difference Jacobian (NewtonSolver.cpp:178-197) -- can then overlap on the device: one mi_ctx (with its own
 * stream) and one mi_edm per evaluation in flight, _begin on all of them, _end on all of them.  One evaluation per
 * handle at a time. */
mi_status mi_edm_compute_f_begin(mi_edm* e, const double* z);
mi_status mi_edm_compute_f_end(mi_edm* e, double* f, double* partial);
/* f from the all-reduced (element-wise summed) partial blocks, MI_EDM_PARTIAL_LEN(n_spikes) doubles: the
 * averaging rule of p->mean_quirk applied to the totals, then the host arithmetic of EventDrivenMap.cu:237-239.
 * With one shard this reproduces mi_edm_compute_f's f bit for bit. */
mi_status mi_edm_residual_from_sums(const mi_edm_params* p, const double* z, const double* sums_and_count,
                                    double* f);
/* Debug taps (SaveLift/SaveEvolve/SaveRestrict, EventDrivenMap.cu:406-503):
 * copy stage outputs of the last compute_f to host.  Any pointer may be NULL.
 * v,s: f32[n_grid]; t0,t1,restricted: f32[S*R]; i0,i1: u16[S*R]; accept: u32[R] */
mi_status mi_edm_debug_read(mi_edm* e, float* v, float* s, float* w, float* t0, uint16_t* i0,
                            float* t1, uint16_t* i1, uint32_t* accept, float* restricted,
                            uint16_t* seed_ind);
/* Decision-coverage taps of the last compute_f (the device-side mirror of the oracle's orc_edm_counters,
 * oracle/edm_oracle.c): the evolve stage is run once more by an instantiation of the kernel that also counts
 *   [0] events of all realisations   [1] most events of one realisation   [2] most Newton iterations of one
 *   firing-time solve (eventTime, EventDrivenMap.cu:561-571)   [3] solves that stopped at newton_max_iter (the
 *   reference's undefined counterMax, :564)   [4] realisations that left the event loop at max_events
 *   [5] accepted realisations   [6] events at which NO neuron will fire (every candidate time is the "never"
 *   value 100.0f: the block arg-min of :843-881 is then decided by its tie rule)   [7] events at which two
 *   neurons share the minimal REAL firing time (lower bound).
 * Debug only: allocates, synchronises, and costs one more evolve. */
#define MI_EDM_N_COUNTERS 8
mi_status mi_edm_debug_counters(mi_edm* e, uint64_t out[MI_EDM_N_COUNTERS]);
/* duration of the stages of the last compute_f in ms (HIP events):
 * [0] lift, [1] evolve, [2] restrict+mean, [3] whole call */
mi_status mi_edm_last_timings(mi_edm* e, float ms[4]);

/* Test hook: evaluate the device math routines of the pipeline on arrays
 * (op: 0 exp, 1 log, 2 pow(a,b), 3 erfinv; 4: a / b[0] the way the kernels divide
 * by a wave-uniform divisor, 5: a / b[0] by IEEE division; 6..11: the firing test
 * will_fire(v0 = a, s0 = b) -> 0/1 on its exact path (even op) and with its hardware
 * pre-decision (odd op) for beta = 13.0589, 1.5, 0.7; 12: a[i ^ 32], the lane-pair
 * exchange the paired firing-time solves rely on, n a multiple of 64) so that tests can
 * compare them with oracle/edm_oracle.c bit for bit.  Device pointers; b_dev may be NULL. */
mi_status mi_edm_math_probe(mi_ctx* ctx, int math_mode, int op, const float* a_dev, const float* b_dev,
                            float* out_dev, size_t n);

/* ---- several GPUs of one node, one host process ----------------------------
 * The reference is single-GPU (EventDrivenMap.cu:80-94 allocates on the current
 * device, Driver.cu:20 builds one problem); BASELINE configs 4-5 shard the
 * realisation axis -- and the query axis of the table interpolation -- over the
 * GPUs of a node.  A group owns one context (device + stream of
 * its own) per shard.  Host code keeps the reference's shape: ONE ComputeF per
 * residual (AbstractNonlinearProblem.hpp:11, Driver.cu:71) -- see
 * mi_group_edm_compute_f -- and one call per interpolation.
 * devices: ndev ordinals, or NULL for 0..ndev-1.  Naming a device more than once
 * is allowed (rehearsal of the shard arithmetic on a box with fewer GPUs); such a
 * group cannot use RCCL and moves data with device-to-device copies instead. */
typedef struct mi_group mi_group;
typedef struct mi_group_grid1 mi_group_grid1;
typedef struct mi_group_edm mi_group_edm;
mi_status mi_group_create(int ndev, const int* devices, mi_group** out);
mi_status mi_group_destroy(mi_group* g);
int mi_group_size(const mi_group* g);
mi_ctx* mi_group_ctx(mi_group* g, int rank);          /* shard `rank`'s context (owned by the group) */
mi_status mi_group_synchronize(mi_group* g);
/* contiguous, balanced split of n units over `world` shards: shard `rank` = [lo, hi); the first n % world shards get
 * one extra unit.  Host arithmetic only. */
void mi_shard_bounds(size_t n, int rank, int world, size_t* lo, size_t* hi);
/* How mi_group_edm_compute_f adds the shards' partial blocks: on the host, where they already are (default), or with
 * ncclAllReduce on the devices (RCCL over xGMI; distinct devices only). */
#define MI_GROUP_REDUCE_HOST 0
#define MI_GROUP_REDUCE_RCCL 1
mi_status mi_group_set_reduce(mi_group* g, int mode);
/* Number of ranks of the group's RCCL communicator as RCCL reports it (ncclCommCount; the communicator is formed by
 * ncclCommInitAll on first use, here if need be).  0: the group cannot use RCCL (a device named twice, or librccl
 * missing); mi_last_error(NULL) says why.  bench.py --backend group prints it as "rccl_ranks". */
int mi_group_rccl_ranks(mi_group* g);

/* 1-D table replicated on every device of the group (x, y: host pointers; flags as mi_grid1_create). */
mi_status mi_group_grid1_create(mi_group* g, const double* x, const double* y, size_t n, unsigned flags,
                                mi_group_grid1** out);
mi_status mi_group_grid1_destroy(mi_group_grid1* t);
/* Query shards from / to host arrays (the arma::vec-facing form): shard r = mi_shard_bounds(nq, r, P) is uploaded to,
 * evaluated on and downloaded from device r, all shards concurrently; returns when yq is complete. */
mi_status mi_group_interp1_f64_host(mi_group* g, const mi_group_grid1* t, const double* xq, double* yq, size_t nq,
                                    double extrap_val);
/* Device-resident shards: xq_dev[r] / yq_dev[r] point to nq_per_shard doubles on device r.  Asynchronous (each shard on
 * its context's stream; mi_group_synchronize waits).  gathered_dev (optional, may be NULL): gathered_dev[r] is a buffer
 * of P * nq_per_shard doubles on device r that receives EVERY shard's results (shard s at offset s * nq_per_shard;
 * in place when yq_dev[r] == gathered_dev[r] + r * nq_per_shard): an RCCL all-gather over xGMI behind the kernels. */
mi_status mi_group_interp1_f64_dev(mi_group* g, const mi_group_grid1* t, const double* const* xq_dev,
                                   double* const* yq_dev, size_t nq_per_shard, double extrap_val,
                                   double* const* gathered_dev);
/* The gather of mi_group_interp1_f64_dev hidden behind its kernels: chunks >= 2 cuts every shard into that many
 * contiguous chunks, and chunk k is exchanged (grouped ncclBroadcast on a second stream per member) while the kernel of
 * chunk k+1 runs.  1 (default): one kernel per shard, then one ncclAllGather.  Results are identical either way. */
mi_status mi_group_set_gather_chunks(mi_group* g, int chunks);

/* 2-D table (BASELINE.json config 3) replicated on every device of the group; arguments as mi_grid2_create (host
 * pointers), the scattered queries sharded exactly as for the 1-D table. */
typedef struct mi_group_grid2 mi_group_grid2;
mi_status mi_group_grid2_create(mi_group* g, const double* x, size_t nx, const double* y, size_t ny, const double* z,
                                unsigned flags, mi_group_grid2** out);
mi_status mi_group_grid2_destroy(mi_group_grid2* t);
mi_status mi_group_interp2_f64_host(mi_group* g, const mi_group_grid2* t, const double* xq, const double* yq, double* zq,
                                    size_t nq, double extrap_val);
mi_status mi_group_interp2_f64_dev(mi_group* g, const mi_group_grid2* t, const double* const* xq_dev,
                                   const double* const* yq_dev, double* const* zq_dev, size_t nq_per_shard,
                                   double extrap_val, double* const* gathered_dev);

/* EventDrivenMap with the realisations sharded over the group: p->n_real is the TOTAL (>= group size); shard r evolves
 * realisations [lo_r, hi_r) = mi_shard_bounds(n_real, r, P) with real_offset = p->real_offset + lo_r, so the per-neuron
 * draws -- and therefore every result -- are those of the unsharded ensemble. */
mi_status mi_group_edm_create(mi_group* g, const mi_edm_params* p, mi_group_edm** out);
mi_status mi_group_edm_destroy(mi_group_edm* e);
mi_status mi_group_edm_set_params(mi_group_edm* e, const mi_edm_params* p);
/* ComputeF (EventDrivenMap.cu:154-240) over all shards: every device runs lift -> evolve -> restrict -> partial sums
 * concurrently, the partial blocks (MI_EDM_PARTIAL_LEN(n_spikes) doubles per shard) are added (see
 * mi_group_set_reduce) and f is formed from the totals exactly as mi_edm_residual_from_sums does.  z, f: host,
 * n_spikes doubles; partial_total (optional): the summed block. */
mi_status mi_group_edm_compute_f(mi_group_edm* e, const double* z, double* f, double* partial_total);
mi_edm* mi_group_edm_shard(mi_group_edm* e, int rank);               /* shard handle, e.g. for mi_edm_debug_read */
mi_status mi_group_edm_shard_bounds(const mi_group_edm* e, int rank, size_t* lo, size_t* hi);

#ifdef __cplusplus
}
#endif
#endif /* MI355_INTERP_H */
