/*
 * lpmp_engine.h — C ABI of the MI355X sweep engine (liblpmp_engine.so).
 *
 * This is the drop-in boundary for ONE path of pawelswoboda/LP_MP: the dual block-coordinate-ascent
 * sweep LP<FMC>::ComputePass and the dual bound LP<FMC>::LowerBound.  The reference's plug-in API is
 * compile-time C++ (FactorContainer / MessageContainer / LP<FMC> templates); the host-side mirrors
 * (lp_mp_amd/include/LP_gpu.hxx for C++, lp_mp_amd/lp.py for Python) keep that surface and talk to the
 * device through the functions below.  Plain pointers and sizes only; no exceptions cross this
 * boundary; every function returns 0 on success or a negative lpmp_status and sets
 * lpmp_last_error().  The host-side mirrors re-throw std::runtime_error, the type the reference throws
 * (reference include/LP_MP.h:458, include/topological_sort.hxx:115).
 *
 * Conventions (mirroring how the reference is used):
 *   - single caller thread per handle (the reference's Solve loop is single-threaded and its static
 *     arenas are not thread-safe, include/factors_messages.hxx:3369-3370).  THREAD AFFINITY: an engine is created,
 *     used and destroyed on ONE thread.  Its streams, its block of pinned words and the staging buffer of its host
 *     copies come from (bounded) per-thread pools and go back to the pool of the thread that calls lpmp_destroy;
 *     different engines may live on different threads;
 *   - host arrays passed in are borrowed for the duration of the call; device arrays passed to
 *     lpmp_upload_model with LPMP_MEM_DEVICE are borrowed until lpmp_destroy / the next upload;
 *   - any structural change on the host side (everything that calls set_flags_dirty in the reference,
 *     include/LP_MP.h:1623) requires a new lpmp_upload_model.
 *
 * Citations are relative to /root/reference.
 */
#ifndef LPMP_ENGINE_H
#define LPMP_ENGINE_H

#include <stdint.h>
#include "lpmp_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lpmp_engine lpmp_engine;
typedef struct lpmp_plan lpmp_plan;

enum lpmp_status {
  LPMP_OK = 0,
  LPMP_ERR_INVALID = -1,     /* bad argument / malformed model (reference: assert / runtime_error) */
  LPMP_ERR_UNSUPPORTED = -2, /* valid in the reference but not executable on the device */
  LPMP_ERR_DEVICE = -3,      /* HIP error, or no GPU */
  LPMP_ERR_STATE = -4        /* call order (e.g. ComputePass before set_reparametrization, LP_MP.h:414,458) */
};

enum lpmp_mem { LPMP_MEM_HOST = 0, LPMP_MEM_DEVICE = 1 };

const char* lpmp_last_error(void);
const char* lpmp_version(void);
/* 0 for the product library.  1 / 2: an experimental build of tools/build_variant.sh (2: with LPMP_ABLATE_* switches that
   remove work from the kernels and compute wrong results) — never what lp_mp_amd/build.py produces. */
int lpmp_experiment_build(void);

/* ---- host-only analysis (no GPU needed) -------------------------------------------------------
 * Replaces LP::SortFactors (include/LP_MP.h:730-797), LP::get_omega (:412-460) and the weight
 * routines (:1086-1154, :1232-1449, :1489-1505); FactorContainer::get_messages
 * (include/factors_messages.hxx:3339-3365).  The cost arrays of the model are not read. */
int lpmp_plan_create(const lpmp_model* m, lpmp_plan** out);
void lpmp_plan_destroy(lpmp_plan* p);
int64_t lpmp_plan_n_factors(const lpmp_plan* p);
int64_t lpmp_plan_n_updated(const lpmp_plan* p, int direction);
int lpmp_plan_get_order(const lpmp_plan* p, int direction, int32_t* out /*[n_factors]*/);          /* forwardOrdering_ */
int lpmp_plan_get_update_order(const lpmp_plan* p, int direction, int32_t* out /*[n_updated]*/);  /* forwardUpdateOrdering_ */
int64_t lpmp_plan_omega_nnz(lpmp_plan* p, int direction);
int64_t lpmp_plan_mask_nnz(lpmp_plan* p, int direction);
int lpmp_plan_get_omega(lpmp_plan* p, int direction, int mode, int64_t* off /*[n_updated+1]*/, double* data);
int lpmp_plan_get_mask(lpmp_plan* p, int direction, int mode, int64_t* off /*[n_updated+1]*/, uint8_t* data);
int lpmp_plan_get_msg_lists(const lpmp_plan* p, int64_t* off /*[n_factors+1]*/, int64_t* entries /*[2*n_messages]: msg*2+role*/);
/* ComputeAnisotropicWeights on an arbitrary ordered factor list (include/LP_MP.h:1232-1415, incl. the
 * strict-subset rules); rows = updated members of the list.  Call with om == NULL to query sizes. */
int lpmp_plan_anisotropic_weights(const lpmp_plan* p, int64_t n, const int32_t* factors, int64_t* n_rows,
                                  int64_t* om_nnz, int64_t* mk_nnz, int64_t* om_off, double* om, int64_t* mk_off,
                                  uint8_t* mk);
/* level schedule of a built-in sweep: levels = dependent steps, launches = kernel launches */
int lpmp_plan_schedule_info(lpmp_plan* p, int direction, int mode, int64_t* n_levels, int64_t* n_launches,
                            int64_t* n_receives, int64_t* n_sends, int64_t* algorithmic_bytes);

/* updated factors of that sweep per device kernel class (LPMP_KCLASS_COUNT entries: generic, dense 4/8/16/32,
 * Potts 4/8/16/32, the run-time-dims forms of the same eight, the streaming class for up to 512 labels, the
 * lane-per-factor class for tiny factors, four classes of updated pairwise factors) */
#define LPMP_KCLASS_COUNT 23
int lpmp_plan_schedule_classes(lpmp_plan* p, int direction, int mode, int64_t* factors /*[LPMP_KCLASS_COUNT]*/);

/* the same summary for an iterator-range pass (LP_MP.h:981-1005) given as factor list + weight rows + receive-mask
 * rows, without a device: what lpmp_schedule_create[_fused] would build.  Arguments as lpmp_compute_pass_custom. */
int lpmp_plan_custom_schedule_info(lpmp_plan* p, int64_t n, const int32_t* factors, const int64_t* omega_off,
                                   const double* omega, const int64_t* mask_off, const uint8_t* mask, int fuse,
                                   int64_t* n_levels, int64_t* n_launches, int64_t* n_receives, int64_t* n_sends,
                                   int64_t* algorithmic_bytes);

/* dependent step (1-based level; 0 = no active message) of every entry of the update order in that sweep (computed alone when the
 * sweep has not been planned: a multi-GPU host asks for the level structure of a global model it never runs as such) */
int lpmp_plan_get_update_levels(lpmp_plan* p, int direction, int mode, int32_t* out /*[n_updated]*/);
/* the same for a whole pass (forward then backward sweep scheduled as one sequence; back-to-back updates
 * of one factor across the two sweeps are folded into one record) */
int lpmp_plan_pass_schedule_info(lpmp_plan* p, int mode, int64_t* n_levels, int64_t* n_launches,
                                 int64_t* n_receives, int64_t* n_sends, int64_t* algorithmic_bytes);

/* chain executor: how a deep sweep (direction 0 / 1, or -1 for the fused forward+backward pass) is run —
 * persistent launches (one per kernel class), their tickets and dependencies, and the launches that stay plain;
 * all 0 when the sweep runs as ordinary launches / graph replay */
int lpmp_plan_chain_info(lpmp_plan* p, int direction, int mode, int64_t* n_chains, int64_t* n_tickets, int64_t* n_dependencies,
                         int64_t* n_plain_launches);
/* ... and how many message vectors of that sweep travel between dependent records through the chain's mailbox (tagged
 * granules polled by the receiving record instead of a completion flag followed by a fetch): rows = sends
 * that also write a mailbox row, receives = receives that poll one.  0 / 0: every hand-over goes through flags. */
int lpmp_plan_mailbox_info(lpmp_plan* p, int direction, int mode, int64_t* n_rows, int64_t* n_receives);

/* 1 when lpmp_compute_pass(n >= 2) joins the tail of a pass with the head of the next one for this mode (2-colour
 * orders: n passes = H, W, (K, W) x (n-1), T) — decided by an op-by-op comparison of the fused
 * schedules; 0 when consecutive passes run one after the other; negative lpmp_status on error */
int lpmp_plan_pass_rotates(lpmp_plan* p, int mode);

/* ---- variable orders and partitions (host only; csrc/graph.cpp) ---------------------------------------------------------------
 * The sweep is Gauss-Seidel over the factor ORDER, which is the caller's input (AddFactorRelation, include/LP_MP.h:698-702; the
 * topological sort of include/topological_sort.hxx:100-144): the engine runs one launch per dependent level of whatever order it
 * is given.  A grid inserted row by row has H + W - 1 levels per directional sweep; in a 2-colour order it has 2 (C3: 14.5 against
 * 5.2 ms per pass).  Another order is another, equally valid trajectory of the dual ascent — not another algorithm.
 *
 * lpmp_plan_suggest_order: rank_of_factor[f] = position of factor f in an order in which the UPDATED factors come colour by colour
 * (two of them conflict when a message joins them or both touch a common factor; component by component: 2 colours where the
 * component has no odd cycle — a grid keeps its 2 levels beside whatever else the model holds —, else a greedy Jones-Plassmann
 * colouring with counter-hash priorities from `seed`), and every other factor keeps its place relative to
 * the updated factors around it (an MRF's pairwise factor between its two unaries, a multicut triplet behind its edges).  The
 * caller turns it into relations as a chain through all factors — AddFactorRelation(by_rank[i], by_rank[i + 1]) for consecutive
 * positions: the only topological order of that chain is the suggested one, forward, and its reverse,
 * backward — and builds its LP with those instead of its own.  n_colours (may be NULL): dependent
 * levels per directional sweep to expect.  The engine prints one line to stderr when a schedule it builds has more than 64 levels
 * (LPMP_QUIET=1 silences it). */
int lpmp_plan_suggest_order(lpmp_plan* p, uint64_t seed, int32_t* rank_of_factor /*[n_factors]*/, int32_t* n_colours);
/* the same colouring for a plain pairwise graph given as an edge list (what synthetic workloads rename their variables by):
 * rank_out[v] = position of variable v in the colour-major order (colour classes in ascending colour, inside a class by index) */
int lpmp_graph_colour_major_order(int64_t n, int64_t m, const int64_t* edge_i, const int64_t* edge_j, uint64_t seed, int64_t* rank_out /*[n]*/,
                                  int32_t* n_colours_out);
/* balanced Kernighan-Lin / label-propagation refinement of a k-way partition of that graph (lp_mp_amd/multi_gpu.py
 * refine_partition, move for move): per round every variable looks at the part most of its neighbours live in; a pseudo-random half
 * of those that would cut fewer edges there move, best gains first, while the target stays within (1 + imbalance) of the mean size */
int lpmp_graph_refine_partition(int64_t n, int64_t m, const int64_t* edge_i, const int64_t* edge_j, int32_t world, int32_t rounds,
                                double imbalance, uint64_t seed, int64_t* part_inout /*[n]*/);

/* ---- device engine --------------------------------------------------------------------------- */
/* LP<FMC>::LP(cmd) (include/LP_MP.h:589-593).  device = HIP device ordinal. */
int lpmp_create(int device, lpmp_engine** out);
void lpmp_destroy(lpmp_engine* e);
/* run all work on this hipStream_t (default: a stream owned by the engine) */
int lpmp_set_stream(lpmp_engine* e, void* hip_stream);

/* add_factor / add_message / AddFactorRelation, flattened (include/LP_MP.h:239-285, :698-702), plus the
 * packed duals as serialize_dual lists them (include/factors_messages.hxx:3196-3223).
 * const_mem / dual_mem say where m->const_data / m->dual_data live.  Device buffers (LPMP_MEM_DEVICE) are borrowed,
 * not copied: whatever fills them must have completed, or be ordered on the engine's stream (lpmp_set_stream), before
 * the next engine call — the engine's own stream is non-blocking and not ordered with the null stream.
 * Size limits of the device kernels (checked when the schedules are built, LPMP_ERR_UNSUPPORTED): a factor that is
 * updated by the wave-per-factor kernels may hold at most 512 doubles of duals and its messages at most 512 entries
 * (unaries with pairwise neighbours: 512 labels; everything else the generic kernel runs: 512 doubles); factors that
 * are only peers (pairwise tables of updated unaries) are bounded by those label counts; at most 32767 active
 * receives and 32767 active sends per updated factor. */
int lpmp_upload_model(lpmp_engine* e, const lpmp_model* m, int const_mem, int dual_mem);

/* LP::set_reparametrization, LP_MP.h:330 (+ the lazy get_omega, :412-460).  Builds the weights of that mode and, for models of up
 * to 2^20 factors, the two directional schedules — a model the device kernels cannot run is refused here (LPMP_ERR_UNSUPPORTED).
 * Larger models get every schedule on first use (lpmp_compute_pass: the fused pass schedule; lpmp_compute_forward_pass / ...: the
 * directional ones; the partitioned drivers: their own iterator-range schedules), and the same refusal then. */
int lpmp_set_reparametrization(lpmp_engine* e, int mode);
/* --reparametrizationType parsed by LP::Begin (LP_MP.h:589-593, :710-722) and switched on in the hot loop (:869-887,
 * :988-1004).  All five run on the device:
 *   shared                 UpdateFactor (factors_messages.hxx:2256-2261)
 *   residual               update_factor_residual (:2270-2279, :2960-3007)
 *   partition              compute_partition_pass over the components of the put_in_same_partition graph
 *                          (lpmp_model.part_pairs; LP_MP.h:1717-1822, :1932-1963), lpmp_set_inner_iterations passes each
 *   overlapping_partition  compute_overlapping_partition_pass (:1824-1843, :1966-2051), then the plain sweeps
 *   adaptive               update_factor_adaptive (:2263-2268, :2860-2926) with the improvement op of
 *                          lpmp_msg_flags; message types without it contribute improvement 0, as in the reference's
 *                          release build, so their factors send nothing
 * LPMP_ERR_UNSUPPORTED: residual / adaptive with batch-capable message ops; adaptive when an updated factor has a
 * message it does not send through (the reference indexes past its weight row there). */
enum lpmp_reparametrization_type {
  LPMP_RTYPE_SHARED = 0, LPMP_RTYPE_RESIDUAL = 1, LPMP_RTYPE_PARTITION = 2, LPMP_RTYPE_OVERLAPPING_PARTITION = 3,
  LPMP_RTYPE_ADAPTIVE = 4
};
int lpmp_set_reparametrization_type(lpmp_engine* e, int rtype);
int lpmp_set_inner_iterations(lpmp_engine* e, int n);       /* --innerIteration, default 5 (LP_MP.h:590) */
/* LP::construct_factor_partition (LP_MP.h:1717-1822): number of partitions; with off != NULL also their factor lists
 * (updated factors only, CSR: off[n_partitions + 1], factors[n_updated]) */
int lpmp_plan_get_partitions(lpmp_plan* p, int64_t* n_partitions, int64_t* off, int32_t* factors);
int lpmp_compute_pass(lpmp_engine* e, int n_passes);        /* LP::ComputePass, LP_MP.h:869-887 ('shared') */
/* optional: build ahead of time what lpmp_compute_pass(e, n_passes) needs that depends on n_passes (the ticket order of
 * n joined passes for the chain executor), e.g. outside of a timed region */
int lpmp_prepare_passes(lpmp_engine* e, int n_passes);
/* ---- passes that run ahead of the caller --------------------------------------------------------------------------
 * The reference's Solver asks for ONE pass per iteration and, by default, for the bound after each (Solver::Iterate /
 * PostIterate, include/solver.hxx:273-284, --lowerBoundComputationInterval 1), while the device is fastest when
 * consecutive passes are one persistent launch (lpmp_compute_pass(e, n): 5.1 against 6.6 ms per pass on C3).  With
 * max_passes_ahead >= 2, lpmp_compute_pass(e, 1) may launch up to that many passes at once (after a snapshot of the
 * duals); the next calls of lpmp_compute_pass(e, 1) only advance a cursor, and lpmp_lower_bound returns the bound after
 * the pass the caller is AT — the launch leaves one row of per-factor bounds per pass.  Every other call first settles:
 * if the caller stopped inside a batch, the duals go back to the snapshot and exactly the passes asked for are run again
 * (n joined passes equal n single ones bit for bit), so results never differ from max_passes_ahead = 0; only time does.
 * The look-ahead adapts to the caller: it doubles while single passes keep coming and restarts at the length of the
 * previous run (MpRoundingSolver: four plain passes between two rounding iterations).  Needs a pass whose consecutive
 * passes join (lpmp_plan_pass_rotates) on an HBM-sized model, send rule `shared`; otherwise every call is executed as it
 * comes.  Default 0 (off; LPMP_SPECULATION=<n> in the environment sets it for engines created afterwards); the solver
 * adapters (lpmp_offload.hxx, LP_gpu.hxx, lp.py) switch it on.  Callers that read a BORROWED dual buffer directly
 * (lpmp_device_duals, LPMP_MEM_DEVICE) call lpmp_synchronize first: it settles.  At most 32. */
int lpmp_set_speculation(lpmp_engine* e, int max_passes_ahead);
int lpmp_speculation_stats(lpmp_engine* e, int64_t* batches, int64_t* passes_launched, int64_t* passes_used, int64_t* rollbacks);
/* device bytes held by the cached ticket lists of joined-pass launches (bounded: LPMP_CHAIN_CACHE_MB, default 2048) */
int64_t lpmp_chain_cache_bytes(const lpmp_engine* e);
int lpmp_compute_forward_pass(lpmp_engine* e);              /* LP::ComputeForwardPass, LP_MP.h:889-900 */
int lpmp_compute_backward_pass(lpmp_engine* e);             /* LP::ComputeBackwardPass, LP_MP.h:902-911 */
/* LP::ComputePass(factorIt, factorItEnd, omegaIt, receive_it), LP_MP.h:981-1005: any factor list with
 * any weights / masks (one row per listed factor). */
int lpmp_compute_pass_custom(lpmp_engine* e, int64_t n, const int32_t* factors, const int64_t* om_off,
                             const double* om, const int64_t* mk_off, const uint8_t* mk);
/* The same, prepared once and replayed: the partition sweeps of the multi-GPU driver are iterator-range
 * passes (main sweep without ghost factors, boundary receive, boundary send) that run every iteration.
 * Precedent in the reference: compute_partition_pass keeps per-partition omega / mask arrays
 * (include/LP_MP.h:1846-1963). */
int lpmp_schedule_create(lpmp_engine* e, int64_t n, const int32_t* factors, const int64_t* om_off, const double* om,
                         const int64_t* mk_off, const uint8_t* mk, int* id_out);
/* as lpmp_schedule_create for a list that concatenates several sweeps (e.g. forward then backward update lists):
 * with fuse != 0 back-to-back updates of one factor are folded into one record (same results) */
int lpmp_schedule_create_fused(lpmp_engine* e, int64_t n, const int32_t* factors, const int64_t* om_off,
                               const double* om, const int64_t* mk_off, const uint8_t* mk, int fuse, int* id_out);
int lpmp_schedule_run(lpmp_engine* e, int id);
int lpmp_schedule_info(lpmp_engine* e, int id, int64_t* n_levels, int64_t* n_launches, int64_t* n_receives,
                       int64_t* n_sends, int64_t* algorithmic_bytes);
int lpmp_schedule_destroy(lpmp_engine* e, int id);
int lpmp_lower_bound(lpmp_engine* e, double* out);          /* LP::LowerBound, LP_MP.h:1507-1518 */
int lpmp_factor_lower_bounds(lpmp_engine* e, double* out /*[n_factors], host*/); /* FactorTypeAdapter::LowerBound */
/* The sweep kernels keep a per-factor lower bound current as a by-product, so lpmp_lower_bound after
 * a pass is a sum over an array.  Call this after changing duals behind the engine's back (writes through
 * lpmp_device_duals or a borrowed dual buffer): the next lpmp_lower_bound recomputes every factor.
 * Order of a direct access to a BORROWED dual buffer: lpmp_synchronize (or lpmp_device_duals) FIRST — it settles passes that ran
 * ahead and, with the rows layout, writes the dense pairwise vectors out to the packed array —, then read / write, then this call.
 * Without the first step, under the rows layout the pairwise vectors read are stale and what was written into them is replaced
 * by the rows' contents here (vector factors live in the packed array only and are not affected).  The lpmp_boundary_* and
 * lpmp_halo_* calls need none of this: they address the rows themselves. */
int lpmp_invalidate_lower_bounds(lpmp_engine* e);
/* How many per-factor bounds the last lpmp_lower_bound / lpmp_factor_lower_bounds had to recompute from the duals (the sweep
 * kernels keep the others current: DESIGN.md 5); the number of factors when it recomputed everything, -1 before the first. */
int64_t lpmp_lower_bound_recomputed(const lpmp_engine* e);
int lpmp_synchronize(lpmp_engine* e);
/* diagnostic: 1 when the uploaded model is streamed with non-temporal loads / stores (tables + duals above 1 GiB, i.e.
 * far larger than L2 + Infinity Cache; LPMP_NT=0/1 in the environment overrides), else 0; -1 without a model */
int lpmp_streaming_access(const lpmp_engine* e);

/* ---- primal rounding inside the sweep -----------------------------------------------------------
 * UpdateFactorPrimal (include/factors_messages.hxx:2332-2373): the same receives and (always 'shared') sends; in
 * between, every factor of a COMPUTE_PRIMAL_SOLUTION type (lpmp_model.ftype_computes_primal) takes the first
 * minimiser of its reparametrised costs as its label unless it already holds one of this time stamp
 * (conditionally_init_primal :3302-3309, MaximizePotentialAndComputePrimal :2382-2389), and the label is copied into
 * the adjacent pairwise factors (propagate_primal_through_messages :2391-2403, :1313-1328).  Built for unary /
 * pairwise models: COMPUTE_PRIMAL vector factors on the left of unary-pairwise messages (LP_MP-MRF's FMC_SRMP), and
 * COMPUTE_PRIMAL pairwise factors (`right` / `full` schedules, MPLP-style): an updated one fills its free sides with
 * the first minimiser in row-major order given the labels its unaries hold and labels those unaries
 * (ComputeLeftFromRightPrimal :1330-1344, then the recursion into their other pairwise factors).  Other message
 * kinds, or two unaries on one side of a pairwise factor, return LPMP_ERR_UNSUPPORTED.  Time stamps as in the
 * reference: 2*iteration+1 / +2. */
int lpmp_compute_forward_pass_and_primal(lpmp_engine* e, uint64_t iteration);   /* LP::ComputeForwardPassAndPrimal, LP_MP.h:914-923 */
int lpmp_compute_backward_pass_and_primal(lpmp_engine* e, uint64_t iteration);  /* LP::ComputeBackwardPassAndPrimal, LP_MP.h:925-934 */
int lpmp_compute_pass_and_primal(lpmp_engine* e, uint64_t iteration);           /* LP::ComputePassAndPrimal, LP_MP.h:936-940 */
int lpmp_check_primal_consistency(lpmp_engine* e, int* consistent);             /* LP::CheckPrimalConsistency, LP_MP.h:1067-1082 */
int lpmp_evaluate_primal(lpmp_engine* e, double* cost);                         /* LP::EvaluatePrimal, LP_MP.h:1521-1536 (+inf when inconsistent / unset) */
/* the factors' primal_ members, what serialize_primal lists: [2*n_factors] int32, vector factor (label, 0), pairwise
 * factor (x0, x1); an unset entry holds the dimension (init_primal) */
int lpmp_download_primal(lpmp_engine* e, int32_t* host_out);
int lpmp_upload_primal(lpmp_engine* e, const int32_t* host_in);

int64_t lpmp_dual_size(const lpmp_engine* e);
/* serialize_dual + save_archive / load_archive (include/serialization.hxx:228-424): packed duals */
int lpmp_download_duals(lpmp_engine* e, double* host_out);
int lpmp_upload_duals(lpmp_engine* e, const double* host_in);
void* lpmp_device_duals(lpmp_engine* e);   /* device pointer of the packed duals (for zero-copy exchange); with the rows layout
                                              the dense pairwise factors' vectors are written out to it first, and the rows are
                                              refreshed from it before the next pass (the caller is assumed to write it) */
/* Rows layout (engine-private): from the NEXT lpmp_upload_model on, every dense pairwise factor lives on the device
 * as one contiguous row [table | m1 | m2] of a buffer of the engine's own, so that the three reads of a receive are one burst
 * (random graphs: C4).  The packed dual array — serialize_dual order, reference factors_messages.hxx:3196-3223 — stays the format
 * of every call that hands duals over (lpmp_download_duals / lpmp_upload_duals / lpmp_device_duals / lpmp_synchronize with a
 * borrowed buffer / the lpmp_boundary_* offsets): the message vectors are copied between the two at those calls, never inside a
 * pass.  Costs the tables a second time in device memory; passes that run ahead of the caller (lpmp_set_speculation) stay off. */
int lpmp_set_rows_layout(lpmp_engine* e, int on);
int lpmp_rows_layout(const lpmp_engine* e);   /* 1 if the uploaded model uses it */

/* Persistent launches (the chain executor and the joined passes in Infinity-Cache order) assume that resident
 * workgroups keep running, i.e. that the device is this process's own.  When several processes time-share one device its scheduler
 * switches queues, ticket holders freeze while their waiters poll, and a run may end in LPMP_ERR_DEVICE ("a dependency wait timed
 * out") with the duals undefined.  A host that shares its device (lpmp_device_identity tells) switches them off for its engine:
 * every schedule then runs launch by launch / as a replayed graph — same results, bit for bit.  Default on; LPMP_NO_CHAIN=1 /
 * LPMP_NO_BLOCKED_PASSES=1 in the environment keep them off whatever is set here.  Callable at any time. */
int lpmp_set_persistent_launches(lpmp_engine* e, int on);
int lpmp_persistent_launches(const lpmp_engine* e);   /* 1 if on */
/* "pci=<domain:bus:device.function> uuid=<hex>" of HIP device ordinal `device` (NUL-terminated, cap >= 64): equal strings = one
 * physical GPU, also when every process has a visibility mask of its own and all of them call their device "0" */
int lpmp_device_identity(int device, char* out, int64_t cap);

const lpmp_plan* lpmp_engine_plan(const lpmp_engine* e);
lpmp_plan* lpmp_engine_plan_mut(lpmp_engine* e);

/* kernel timing with HIP events on the engine's stream (bench.py roofline leg).  While enabled every
 * sweep launch is bracketed by an event pair.  Classes: enum KClass in lp_mp_amd/csrc/plan.hpp. */
int lpmp_enable_kernel_timing(lpmp_engine* e, int on);
int lpmp_get_kernel_timing(lpmp_engine* e, int n_classes, double* ms /*[n]*/, int64_t* launches /*[n]*/,
                           int64_t* factors /*[n]*/, int64_t* receives /*[n]*/, int64_t* bytes /*[n]*/);
int lpmp_reset_kernel_timing(lpmp_engine* e);
/* of the launches reported per class: how many were persistent launches of the chain executor */
int lpmp_get_chain_launches(lpmp_engine* e, int n_classes, int64_t* chain_launches /*[n]*/);

/* ---- boundary step of the partitioned (multi-GPU) sweep, DESIGN.md 7 -------------------------------------------------
 * One process per GPU owns one part of the factor graph (lp_mp_amd/multi_gpu.py builds the parts; a C++ host can do
 * the same from lpmp_model arrays).  Cut messages travel as flat runs of doubles between DEVICE buffers the caller
 * owns: the caller posts the exchange itself (RCCL ncclSend / ncclRecv or all-to-all-v on lpmp_engine_stream, or
 * torch.distributed) between these calls.  Every step is the reference's UpdateFactor of a non-owner endpoint restricted
 * to its cut messages (include/factors_messages.hxx:2256-2261), so the whole schedule replays on the unpartitioned
 * model with LP::ComputePass(factorIt, ...) (include/LP_MP.h:981-1005).
 *   out_*   the cut messages this part OWNS (ghost vectors), in exchange order (by peer, then key)
 *   in_*    the cut messages owned elsewhere that end in a variable of this part, in exchange order; in_omega their send
 *           weights (a row sums to <= 1, LP_MP.h:1008-1014); in_order = indices into in_* grouped by variable and, inside
 *           a variable, in the order its message list holds them (receives and sends happen in that order) */
typedef struct lpmp_boundary lpmp_boundary;
int lpmp_boundary_create(lpmp_engine* e, int64_t n_out, const int64_t* out_dual_off, const int32_t* out_len, int64_t n_in,
                         const int64_t* in_dual_off, const int32_t* in_len, const double* in_omega, const int64_t* in_order,
                         lpmp_boundary** out);
void lpmp_boundary_destroy(lpmp_boundary* b);
int64_t lpmp_boundary_out_doubles(const lpmp_boundary* b);   /* size of the send buffer of pack / the buffer of fold */
int64_t lpmp_boundary_in_doubles(const lpmp_boundary* b);    /* size of the receive buffer / the reply buffer */
/* owner: send[...] = ghost vectors (after the ghost receive schedule ran), ghosts zeroed */
int lpmp_boundary_pack(lpmp_engine* e, lpmp_boundary* b, double* send_dev);
/* non-owner: theta += received (list order); reply = omega * theta after all receives; theta -= reply (list order) */
int lpmp_boundary_reply(lpmp_engine* e, lpmp_boundary* b, const double* recv_dev, double* reply_dev);
/* owner: ghost vectors = reply (then run the ghost send schedule) */
int lpmp_boundary_fold(lpmp_engine* e, lpmp_boundary* b, const double* back_dev);
void* lpmp_engine_stream(lpmp_engine* e);                    /* the hipStream_t all engine work is issued on */

/* ---- halos of the lock-step partitioned sweep, DESIGN.md 7 (lp_mp_amd/lockstep.py) ------------------------------------
 * Several ranks execute THE unpartitioned sweep (LP::ComputePass, include/LP_MP.h:981-1005) level by level; between two runs of
 * levels (lpmp_schedule_run) a rank ships the message vectors it wrote that the next run reads on other ranks.  No arithmetic:
 * pack copies the listed vectors of the dual array into a contiguous DEVICE buffer in exchange order, unpack copies a received
 * buffer into the listed vectors; the caller posts the exchange in between (ncclSend / ncclRecv, all-to-all-v) on
 * lpmp_engine_stream.  Vectors are (offset into the packed dual array = serialize_dual order, length) pairs. */
typedef struct lpmp_halo lpmp_halo;
int lpmp_halo_create(lpmp_engine* e, int64_t n_out, const int64_t* out_dual_off, const int32_t* out_len, int64_t n_in,
                     const int64_t* in_dual_off, const int32_t* in_len, lpmp_halo** out);
void lpmp_halo_destroy(lpmp_halo* h);
int64_t lpmp_halo_out_doubles(const lpmp_halo* h);           /* size of the buffer pack fills */
int64_t lpmp_halo_in_doubles(const lpmp_halo* h);            /* size of the buffer unpack reads */
int lpmp_halo_pack(lpmp_engine* e, lpmp_halo* h, double* send_dev);
int lpmp_halo_unpack(lpmp_engine* e, lpmp_halo* h, const double* recv_dev);

/* synthetic workloads: out[i] = u01(splitmix64(seed + (first+i+1)*GOLDEN)) written on the device */
int lpmp_synth_fill(void* device_ptr, int64_t n, uint64_t seed, uint64_t first, void* hip_stream);
/* the same for n_blocks blocks of block_len values each, block b continuing the stream at first_dev[b] (device array):
 * a rank's scattered share of a global cost stream */
int lpmp_synth_fill_blocks(void* device_ptr, int64_t n_blocks, int64_t block_len, uint64_t seed, const int64_t* first_dev, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LPMP_ENGINE_H */
