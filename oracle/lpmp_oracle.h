/*
 * lpmp_oracle.h — CPU oracle for the LP_MP dual block-coordinate-ascent sweep.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lp_mp_amd/ (the product) may include, link or load
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / reported baseline.
 *
 * It is a plain-C, single-threaded restatement of the reference's algorithm
 * (/root/reference/include/LP_MP.h, factors_messages.hxx, topological_sort.hxx); every function in
 * lpmp_oracle.c cites the reference lines it follows.  See the header of lpmp_oracle.c for how it
 * is pinned.
 */
#ifndef LPMP_ORACLE_H
#define LPMP_ORACLE_H

#include <stdint.h>
#include "../include/lpmp_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc orc_t;

orc_t* orc_create(const lpmp_model* m); /* deep copy; NULL on error */
void orc_destroy(orc_t* o);
const char* orc_last_error(void);

int orc_set_mode(orc_t* o, int repam_mode);             /* LP::set_reparametrization */
int orc_set_reparametrization_type(orc_t* o, int rtype); /* --reparametrizationType, enum lpmp_reparametrization_type numbering: 0 shared, 1 residual, 2 partition, 3 overlapping_partition, 4 adaptive (LP_MP.h:710-722) */
int orc_set_inner_iterations(orc_t* o, int n);           /* --innerIteration (LP_MP.h:590), default 5 */
/* LP::construct_factor_partition (LP_MP.h:1717-1822): the partitions of the put_in_same_partition graph */
int64_t orc_n_partitions(orc_t* o);
int orc_get_partitions(orc_t* o, int64_t* off /*[n_partitions+1]*/, int32_t* factors /*[n_updated]*/);
int orc_compute_pass(orc_t* o, int n_passes);           /* LP::ComputePass, default 'shared' type */
int orc_forward_pass(orc_t* o);                         /* LP::ComputeForwardPass */
int orc_backward_pass(orc_t* o);                        /* LP::ComputeBackwardPass */
double orc_lower_bound(orc_t* o);                       /* LP::LowerBound */
/* primal rounding inside the sweep (LP_MP.h:914-940, 1592-1602; factors_messages.hxx:2332-2373) */
int orc_forward_pass_and_primal(orc_t* o, uint64_t iteration);   /* LP::ComputeForwardPassAndPrimal */
int orc_backward_pass_and_primal(orc_t* o, uint64_t iteration);  /* LP::ComputeBackwardPassAndPrimal */
int orc_compute_pass_and_primal(orc_t* o, uint64_t iteration);   /* LP::ComputePassAndPrimal */
int orc_check_primal_consistency(orc_t* o);                      /* LP::CheckPrimalConsistency, LP_MP.h:1067-1082 */
double orc_evaluate_primal(orc_t* o);                            /* LP::EvaluatePrimal, LP_MP.h:1521-1536 */
void orc_get_primal(orc_t* o, int32_t* out /*[2*n_factors]*/);   /* the factors' primal_ members (unset = dim) */
void orc_get_primal_access(orc_t* o, uint64_t* out /*[n_factors]*/);

int64_t orc_n_factors(orc_t* o);
int64_t orc_dual_size(orc_t* o);
void orc_get_duals(orc_t* o, double* out);              /* packed as lpmp_model.dual_data */
void orc_set_duals(orc_t* o, const double* in);

/* ordering (LP::SortFactors): full order = all factors, update order = those with FactorUpdated() */
void orc_get_order(orc_t* o, int dir, int32_t* out);            /* [n_factors] */
int64_t orc_n_updated(orc_t* o, int dir);
void orc_get_update_order(orc_t* o, int dir, int32_t* out);     /* [n_updated] */

/* weights (LP::get_omega): CSR rows follow the update order */
int64_t orc_omega_nnz(orc_t* o, int dir);
int64_t orc_mask_nnz(orc_t* o, int dir);
int orc_get_omega(orc_t* o, int dir, int mode, int64_t* off, double* data);
int orc_get_mask(orc_t* o, int dir, int mode, int64_t* off, uint8_t* data);

/* per-factor message list (FactorContainer::get_messages): CSR over factors;
 * entry = message index * 2 + role (0: factor is the left factor, 1: right) */
int64_t orc_msg_list_size(orc_t* o);
void orc_get_msg_lists(orc_t* o, int64_t* off /*[n_factors+1]*/, int64_t* entries);

/* LP::ComputePass(factorIt, factorItEnd, omegaIt, receive_it) — the public iterator-range
 * template (LP_MP.h:981-1005): any factor list with any weights/masks. */
int orc_compute_pass_custom(orc_t* o, int64_t n, const int32_t* factors, const int64_t* om_off,
                            const double* om, const int64_t* mk_off, const uint8_t* mk);

/* ComputeAnisotropicWeights on an arbitrary ordered factor list, incl. the strict-subset rules
 * (LP_MP.h:1232-1415).  Output rows: one per updated factor of the list.
 * om_off/mk_off need n+1 entries at most; om/mk sized by orc_sublist_nnz. */
int orc_sublist_nnz(orc_t* o, int64_t n, const int32_t* factors, int64_t* n_rows, int64_t* om_nnz,
                    int64_t* mk_nnz);
int orc_anisotropic_weights_sublist(orc_t* o, int64_t n, const int32_t* factors, int64_t* om_off,
                                    double* om, int64_t* mk_off, uint8_t* mk);

/* executed receives / sends since creation (unit of the throughput metric) */
void orc_get_counters(orc_t* o, int64_t* n_receives, int64_t* n_sends);

/* single-op helpers used by the known-answer tests ------------------------------------- */
/* delta[] = message computed from factor `src` for message `msg` toward the other side,
 * with weight omega, WITHOUT applying it.  to_left != 0: computed by the right factor. */
int orc_message_value(orc_t* o, int64_t msg, int to_left, double omega, double* delta);
double orc_factor_lower_bound(orc_t* o, int64_t f);

/* counter-based generator shared with the engine's synthetic workloads:
 * out[i] = u01(splitmix64(seed + (first + i) * GOLDEN)) in [0,1) */
void orc_synth_u01(double* out, int64_t n, uint64_t seed, uint64_t first);

#ifdef __cplusplus
}
#endif
#endif
