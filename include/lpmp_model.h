/*
 * lpmp_model.h — flat, C-ABI description of an LP_MP factor graph.
 *
 * This is the data format that crosses the drop-in boundary: the host-side
 * mirror of the reference's LP<FMC> (lp_mp_amd/include/LP_gpu.hxx, lp_mp_amd/lp.py)
 * flattens the FactorContainer / MessageContainer objects a user adds through
 * add_factor / add_message / AddFactorRelation into these arrays and hands them
 * to the engine (include/lpmp_engine.h).  The CPU oracle (oracle/) reads the same
 * struct so that parity tests feed identical inputs to both.
 *
 * What each field restates from the reference (paths relative to /root/reference):
 *   - factor / message *types*  = positions in FMC::FactorList / FMC::MessageList
 *     (test/test_model.hxx:130-137; include/factors_messages.hxx:571-578).
 *   - schedule                  = message_passing_schedule (include/config.hxx:43-49).
 *   - n_left / n_right          = NO_OF_LEFT_FACTORS / NO_OF_RIGHT_FACTORS template
 *     constants (0 variable, >0 exact, <0 at most; include/config.hxx:60-66); they
 *     select the message storage and hence the per-factor message iteration order
 *     (include/factors_messages.hxx:2081-2119, LIFO list :2030-2041).
 *   - relations                 = ForwardPassFactorRelation / BackwardPassFactorRelation
 *     (include/LP_MP.h:698-702).
 *   - dual_data                 = what serialize_dual enumerates, factor by factor in
 *     insertion order (include/factors_messages.hxx:3196-3223); matrices are never
 *     part of the dual for the kinds below.
 *
 * All arithmetic is IEEE double (REAL = double, include/config.hxx:28).
 */
#ifndef LPMP_MODEL_H
#define LPMP_MODEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Factor kinds the engine can hold on the device. */
enum lpmp_factor_kind {
  /* cost vector theta[dim0]; dual = theta.  LowerBound = min(theta), or min(0, min theta)
   * when flag LPMP_FF_IMPLICIT_ORIGIN is set.  Covers UnarySimplexFactor
   * (test/simplex.cpp:8-12), labeling_factor (include/factors/labeling_list_factor.hxx:220-275)
   * and test_factor (test/test_model.hxx:10-64). */
  LPMP_F_VECTOR = 0,
  /* const table T[dim0][dim1] row-major; dual = m1[dim0], m2[dim1];
   * cost(a,b) = T[a][b] + m1[a] + m2[b]  (test/simplex.cpp:52-65). */
  LPMP_F_PAIRWISE_DENSE = 1,
  /* const scalar diff; dual = m1[dim0], m2[dim0]; cost(a,b) = diff*[a!=b] + m1[a] + m2[b]
   * (test/potts_factor.cpp:8-72). dim1 == dim0. */
  LPMP_F_PAIRWISE_POTTS = 2
};

enum lpmp_factor_flags {
  LPMP_FF_IMPLICIT_ORIGIN = 1 /* labeling_factor<...,IMPLICIT_ORIGIN=true> */
};

/* Message-op kinds. left/right refer to the MessageContainer's LEFT_FACTOR / RIGHT_FACTOR. */
enum lpmp_msg_kind {
  /* left = VECTOR(theta), right = PAIRWISE_*; param = side (0: left variable of the pair, 1: right).
   * to-left  (computed by right): delta = omega * min_marginal_side(right)
   * to-right (computed by left):  delta = omega * theta
   * (test/simplex_marginalization.cpp:22-41). */
  LPMP_M_UNARY_PAIRWISE = 0,
  /* left = VECTOR[n_l], right = VECTOR[n_r]; param = index of a match table (see tables below).
   * labeling_message (include/factors/labeling_list_factor.hxx:346-506). */
  LPMP_M_LABELING = 1,
  /* left = VECTOR[n], right = VECTOR[n]; delta = omega * (src - min(src))
   * test_message (test/test_model.hxx:66-98). */
  LPMP_M_MINNORM = 2
};

/* Optional members of a message op, detected at compile time in the reference (LP_MP_FUNCTION_EXISTENCE_CLASS,
 * include/factors_messages.hxx:46-81) and declared here per message type.  None of the ops in the reference tree has
 * them; they are hooks for richer factor families, so the device ops behind them are this engine's own definitions:
 *   IMPROVEMENT   send_message_to_left_improvement / send_message_to_right_improvement (:734-747, :795-808), used by
 *                 --reparametrizationType adaptive (:2860-2926).  Device op: the exact change of LowerBound(left) +
 *                 LowerBound(right) a weight-1 send of this message would cause, computed from the current duals.
 *                 Without the flag the reference's release build gets 0 from the container (assert(false); return 0)
 *                 and adaptive updates send nothing; the engine does the same.
 *   BATCH_TO_RIGHT / BATCH_TO_LEFT   static SendMessagesToRight / SendMessagesToLeft(factor, msg_begin, msg_end, omega)
 *                 (:1070-1078, :1180-1188): CallSendMessages hands all active messages of the type to ONE call with
 *                 the sum of their weights when more than one is active (:2709-2720).  Device op: every active message
 *                 gets (omega / number of active messages) times the plain message, all computed from the factor as
 *                 it is on entry. */
enum lpmp_msg_flags {
  LPMP_MF_IMPROVEMENT = 1,
  LPMP_MF_BATCH_TO_RIGHT = 2,
  LPMP_MF_BATCH_TO_LEFT = 4
};

/* message_passing_schedule, same numbering as include/config.hxx:43-49 */
enum lpmp_schedule {
  LPMP_SCHED_LEFT = 0,
  LPMP_SCHED_RIGHT = 1,
  LPMP_SCHED_FULL = 2,
  LPMP_SCHED_ONLY_SEND = 3,
  LPMP_SCHED_NONE = 4
};

/* LPReparametrizationMode, same numbering as include/config.hxx:71 */
enum lpmp_repam_mode {
  LPMP_REPAM_ANISOTROPIC = 0,
  LPMP_REPAM_ANISOTROPIC2 = 1,
  LPMP_REPAM_UNIFORM = 2,
  LPMP_REPAM_DAMPED_UNIFORM = 3,
  LPMP_REPAM_MIXED = 4, /* assert(false) in the reference (LP_MP.h:1455): rejected */
  LPMP_REPAM_COUNT = 4
};

enum lpmp_direction { LPMP_FORWARD = 0, LPMP_BACKWARD = 1 };

/* One entry of FMC::MessageList. */
typedef struct lpmp_msg_type {
  int32_t left_ftype;  /* LEFT_FACTOR_NO  */
  int32_t right_ftype; /* RIGHT_FACTOR_NO */
  int32_t schedule;    /* enum lpmp_schedule */
  int32_t n_left;      /* NO_OF_LEFT_FACTORS  */
  int32_t n_right;     /* NO_OF_RIGHT_FACTORS */
  int32_t kind;        /* enum lpmp_msg_kind */
  int32_t param;       /* side (UNARY_PAIRWISE) or table index (LABELING) */
  int32_t flags;       /* enum lpmp_msg_flags: which OPTIONAL members the message op defines */
} lpmp_msg_type;

/* The whole model. Every pointer is borrowed for the duration of the call it is passed to. */
typedef struct lpmp_model {
  /* --- FMC --- */
  int32_t n_ftypes;                 /* |FMC::FactorList| */
  const uint8_t* ftype_computes_primal; /* [n_ftypes] FactorContainer COMPUTE_PRIMAL flag (affects FactorUpdated only), may be NULL */
  int32_t n_mtypes;                 /* |FMC::MessageList| */
  const lpmp_msg_type* mtypes;      /* [n_mtypes] */

  /* --- labeling match tables (LPMP_M_LABELING) ---
   * table t: for every right labeling r, tab_data[tab_off[t] + r] = index of the matching left
   * labeling, or tab_nleft[t] if none (matching_left_labeling, labeling_list_factor.hxx:384-402). */
  int32_t n_tables;
  const int64_t* tab_off;           /* [n_tables+1] */
  const int32_t* tab_data;
  const int32_t* tab_nleft;         /* [n_tables] */

  /* --- factors, in add_factor order (LP_MP.h:239-253) --- */
  int64_t n_factors;
  const int32_t* f_type;            /* [n_factors] index into FactorList */
  const uint8_t* f_kind;            /* [n_factors] enum lpmp_factor_kind */
  const uint8_t* f_flags;           /* [n_factors] */
  const int32_t* f_dim0;            /* [n_factors] */
  const int32_t* f_dim1;            /* [n_factors] (PAIRWISE_DENSE only; else ignored) */
  /* packed by factor in insertion order; per-factor sizes follow from kind/dims:
   *   const: DENSE dim0*dim1, POTTS 1, VECTOR 0;  dual: VECTOR dim0, DENSE dim0+dim1, POTTS 2*dim0 */
  const double* const_data;
  const double* dual_data;

  /* --- messages, in add_message order (LP_MP.h:267-285) --- */
  int64_t n_messages;
  const int32_t* m_type;            /* [n_messages] index into MessageList */
  const int32_t* m_left;            /* [n_messages] factor index */
  const int32_t* m_right;           /* [n_messages] factor index */

  /* --- ordering relations (LP_MP.h:698-702) --- */
  int64_t n_rel_fwd;
  const int32_t* rel_fwd;           /* [n_rel_fwd][2] : f1 before f2 in the forward pass */
  int64_t n_rel_bwd;
  const int32_t* rel_bwd;           /* [n_rel_bwd][2] */

  double constant;                  /* LP::add_to_constant (LP_MP.h:462) */

  /* --- LP::put_in_same_partition(f1, f2) calls, in call order (LP_MP.h:465): the partition graph of
   * --reparametrizationType partition / overlapping_partition (LP_MP.h:1717-1822) --- */
  int64_t n_part_pairs;
  const int32_t* part_pairs;        /* [n_part_pairs][2] */
} lpmp_model;

/* sizes implied by kind/dims */
static inline int64_t lpmp_factor_const_size(int kind, int dim0, int dim1) {
  return kind == LPMP_F_PAIRWISE_DENSE ? (int64_t)dim0 * dim1 : (kind == LPMP_F_PAIRWISE_POTTS ? 1 : 0);
}
static inline int64_t lpmp_factor_dual_size(int kind, int dim0, int dim1) {
  return kind == LPMP_F_PAIRWISE_DENSE ? (int64_t)dim0 + dim1 : (kind == LPMP_F_PAIRWISE_POTTS ? 2 * (int64_t)dim0 : dim0);
}

#ifdef __cplusplus
}
#endif
#endif /* LPMP_MODEL_H */
