"""Host-side mirror of the reference's plug-in surface for the sweep path, in Python.

Same names, argument meaning and error behaviour as the reference (paths relative to /root/reference):

  FMC / FactorContainer / MessageContainer   test/test_model.hxx:130-137, include/factors_messages.hxx:571-578
  LP<FMC>                                     include/LP_MP.h:239-285 (add_factor/add_message), :698-728, :330,
                                              :869-911 (ComputePass...), :981-1005 (iterator-range ComputePass),
                                              :412-460 (get_omega), :1507-1518 (LowerBound), :462 (add_to_constant)
  Solver::Solve                               include/solver.hxx:230-287
  StandardVisitor                             include/visitors/standard_visitor.hxx:28-199
  LpControl / LPReparametrizationMode         include/config.hxx:71-105

The factor and message OPS that can be plugged in are the device-capable kinds: a user op that is not
one of them is rejected when the FMC is declared (there is no CPU fallback that would silently run it).
The compute itself happens in the HIP engine behind the C ABI (include/lpmp_engine.h).
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import model as M
from .engine import Engine, EngineError


# ---- factor ops (what the reference calls FACTOR_TYPE) -------------------------------------------------
class UnarySimplexFactor:
    """cost vector over the labels of one variable (reference test/simplex.cpp:8-12)."""
    kind = M.F_VECTOR
    implicit_origin = False

    def __init__(self, cost: Sequence[float]):
        self.cost = np.asarray(cost, np.float64).reshape(-1)


class ConstantFactor(UnarySimplexFactor):
    """factor without variables that carries an offset (reference include/factors/constant_factor.hxx:10-30): dual =
    the offset, LowerBound = the offset; on the device a vector factor with one entry."""

    def __init__(self, offset: float = 0.0):
        super().__init__([offset])

    def AddToOffset(self, delta: float):
        self.cost[0] += delta


class test_factor(UnarySimplexFactor):
    """two-label toy factor (reference test/test_model.hxx:10-64)."""

    def __init__(self, x: float, y: float):
        super().__init__([x, y])


class PairwiseSimplexFactor:
    """dense table + one message vector per side (reference test/simplex.cpp:52-65)."""
    kind = M.F_PAIRWISE_DENSE

    def __init__(self, dim1: int, dim2: int, cost=None):
        self.dim1, self.dim2 = int(dim1), int(dim2)
        self.table = np.zeros((dim1, dim2)) if cost is None else np.asarray(cost, np.float64).reshape(dim1, dim2)

    def cost(self, x1, x2):
        return self.table[x1, x2]


class pairwise_potts_factor:
    """diff * [x1 != x2] (reference test/potts_factor.cpp:34-36)."""
    kind = M.F_PAIRWISE_POTTS

    def __init__(self, dim: int, diff_cost: float):
        self.dim, self.diff = int(dim), float(diff_cost)


def labeling_factor(labelings: Sequence[Sequence[int]], implicit_origin: bool):
    """labeling_factor<labelings<...>, IMPLICIT_ORIGIN> (reference include/factors/labeling_list_factor.hxx:220)."""
    labs = [tuple(l) for l in labelings]

    class _LabelingFactor:
        kind = M.F_VECTOR
        labelings = labs

        def __init__(self, cost=None):
            self.cost = np.zeros(len(labs)) if cost is None else np.asarray(cost, np.float64).reshape(len(labs))

    _LabelingFactor.implicit_origin = bool(implicit_origin)
    return _LabelingFactor


# ---- message ops (MESSAGE_TYPE) ------------------------------------------------------------------------
@dataclass(frozen=True)
class UnaryPairwiseMessage:
    """UnaryPairwiseMessage<Chirality> (reference test/simplex_marginalization.cpp:19-20); side 0 = left variable."""
    side: int
    kind: int = M.M_UNARY_PAIRWISE
    flags: int = 0      # M.MF_*: optional members of the op (improvement for adaptive sends, static batch sends)


@dataclass(frozen=True)
class labeling_message:
    """labeling_message<LEFT_LABELINGS, RIGHT_LABELINGS, INDICES...> (labeling_list_factor.hxx:346)."""
    left_labelings: tuple
    right_labelings: tuple
    indices: tuple
    kind: int = M.M_LABELING


@dataclass(frozen=True)
class test_message:
    """min-normalised copy (reference test/test_model.hxx:66-98)."""
    kind: int = M.M_MINNORM


# ---- containers and FMC ----------------------------------------------------------------------------------
@dataclass
class FactorContainer:
    """FactorContainer<FACTOR_TYPE, FMC, FACTOR_NO, COMPUTE_PRIMAL_SOLUTION=false>"""
    factor_type: type
    factor_no: int
    compute_primal: bool = False


@dataclass
class MessageContainer:
    """MessageContainer<MSG, LEFT_NO, RIGHT_NO, message_passing_schedule, NO_LEFT, NO_RIGHT, FMC, MSG_NO>"""
    message_type: object
    left_factor_no: int
    right_factor_no: int
    schedule: int
    no_left_factors: int
    no_right_factors: int
    message_no: int


class FMC:
    def __init__(self, name: str, FactorList: List[FactorContainer], MessageList: List[MessageContainer]):
        self.name, self.FactorList, self.MessageList = name, list(FactorList), list(MessageList)
        for i, f in enumerate(self.FactorList):
            if f.factor_no != i:
                raise RuntimeError("FactorList: factor numbers must be consecutive")
            if not hasattr(f.factor_type, "kind"):
                raise RuntimeError(f"factor type {f.factor_type!r} has no device kind: only the device-capable "
                                   "factor ops can be plugged in (no CPU fallback)")
        for i, m in enumerate(self.MessageList):
            if m.message_no != i:
                raise RuntimeError("MessageList: message numbers must be consecutive")
            if not hasattr(m.message_type, "kind"):
                raise RuntimeError(f"message type {m.message_type!r} has no device kind")


class LPReparametrizationMode:
    Anisotropic, Anisotropic2, Uniform, DampedUniform, Mixed, Undefined = 0, 1, 2, 3, 4, 5


def LPReparametrizationModeConvert(s: str) -> int:
    if s not in M.REPAM_NAMES:
        raise RuntimeError("reparametrization mode " + s + " unknown")   # reference config.hxx:88
    return M.REPAM_NAMES[s]


class LP:
    """LP<FMC> for the sweep path.  Structural calls are collected on the host; the model is flattened and
    uploaded lazily by the first call that needs the device (everything the reference does after
    set_flags_dirty(), LP_MP.h:1623)."""

    REPARAMETRIZATION_TYPES = {"shared": 0, "residual": 1, "partition": 2, "overlapping_partition": 3, "adaptive": 4}

    def __init__(self, fmc: FMC, device: int = 0, reparametrizationType: str = "shared", innerIteration: int = 5, speculation: int = 16):
        """reparametrizationType / innerIteration: the reference's --reparametrizationType and --innerIteration
        (LP_MP.h:589-593); all five types run on the device.  speculation: how many passes the engine may run ahead of a
        Solve loop that asks for one pass and one bound per iteration (Engine.set_speculation; results unchanged)."""
        self._speculation = int(speculation)
        if reparametrizationType not in self.REPARAMETRIZATION_TYPES:
            raise RuntimeError("reparametrization type " + reparametrizationType + " unknown")
        self._rtype = self.REPARAMETRIZATION_TYPES[reparametrizationType]
        self.FMC = fmc
        self._device = device
        self._factors = []       # (container, op)
        self._messages = []      # (container, left, right)
        self._rel_fwd, self._rel_bwd = [], []
        self._partition_graph = []
        self._inner = int(innerIteration)
        self._tables = {}
        self._constant = 0.0
        self._repam = LPReparametrizationMode.Undefined
        self._engine: Optional[Engine] = None
        self._dirty = True
        self._begun = False
        self._duals_host: Optional[np.ndarray] = None

    # -- problem construction (reference LP_MP.h:239-285, :698-702) ---------------------------------------
    def add_factor(self, container: FactorContainer, *args) -> int:
        op = args[0] if len(args) == 1 and isinstance(args[0], container.factor_type) else container.factor_type(*args)
        self._pull_duals()
        self._factors.append((container, op))
        self._dirty = True
        return len(self._factors) - 1

    def add_message(self, container: MessageContainer, left: int, right: int) -> int:
        lf, rf = self._factors[left][0], self._factors[right][0]
        if lf.factor_no != container.left_factor_no or rf.factor_no != container.right_factor_no:
            raise RuntimeError("add_message: factor types do not match the message container")
        self._pull_duals()
        self._messages.append((container, left, right))
        self._dirty = True
        return len(self._messages) - 1

    def AddFactorRelation(self, f1: int, f2: int):
        self.ForwardPassFactorRelation(f1, f2)
        self.BackwardPassFactorRelation(f2, f1)

    def ForwardPassFactorRelation(self, f1: int, f2: int):
        self._rel_fwd.append((f1, f2)); self._dirty = True

    def BackwardPassFactorRelation(self, f1: int, f2: int):
        self._rel_bwd.append((f1, f2)); self._dirty = True

    def suggested_order(self, seed: int = 0):
        """the order the engine suggests for this LP as it stands (lpmp_plan_suggest_order on the host-only plan: no GPU needed):
        (by_rank, n_colours) with by_rank[i] = the factor (add_factor's return value) at position i — the updated factors colour by
        colour, one dependent level per colour.  apply_suggested_order() replaces the relations with it."""
        from .engine import Plan
        rank, k = Plan(self.flat_model()).suggest_order(seed)
        by_rank = np.empty(rank.shape[0], np.int64)
        by_rank[rank] = np.arange(rank.shape[0])
        return by_rank.tolist(), k

    def apply_suggested_order(self, seed: int = 0) -> int:
        """drop this LP's factor relations and chain all factors in the suggested order instead — AddFactorRelation(by_rank[i],
        by_rank[i + 1]), what INTEGRATION.md 2a shows a C++ caller doing: same factors, messages and costs, another (equally valid)
        sweep order with far fewer dependent levels for LPs inserted row by row / chain by chain.  Returns the number of colours."""
        by_rank, k = self.suggested_order(seed)
        self._rel_fwd, self._rel_bwd = [], []
        for a, b in zip(by_rank[:-1], by_rank[1:]):
            self.AddFactorRelation(a, b)
        return k

    def put_in_same_partition(self, f1: int, f2: int):
        """reference LP_MP.h:465"""
        self._partition_graph.append((f1, f2)); self._dirty = True

    def GetNumberOfFactors(self) -> int:
        return len(self._factors)

    def GetNumberOfMessages(self) -> int:
        return len(self._messages)

    def GetFactor(self, i: int):
        return self._factors[i][1]

    def add_to_constant(self, x: float):
        self._constant += float(x); self._dirty = True

    def Begin(self):
        self._repam = LPReparametrizationMode.Undefined      # reference LP_MP.h:707
        self._begun = True

    def End(self):
        pass

    def set_reparametrization(self, r):
        self._repam = LPReparametrizationModeConvert(r) if isinstance(r, str) else int(r)

    # -- flattening --------------------------------------------------------------------------------------
    def _table_id(self, msg_op: labeling_message) -> int:
        key = (msg_op.left_labelings, msg_op.right_labelings, msg_op.indices)
        return self._tables.setdefault(key, len(self._tables))

    def flat_model(self) -> M.FlatModel:
        fmc = self.FMC
        self._tables = {}
        mtypes = []
        for mc in fmc.MessageList:
            op = mc.message_type
            param = op.side if op.kind == M.M_UNARY_PAIRWISE else (self._table_id(op) if op.kind == M.M_LABELING else 0)
            mtypes.append(M.MsgType(mc.left_factor_no, mc.right_factor_no, mc.schedule, mc.no_left_factors,
                                    mc.no_right_factors, op.kind, param, getattr(op, "flags", 0)))
        b = M.ModelBuilder(len(fmc.FactorList), mtypes, [int(f.compute_primal) for f in fmc.FactorList])
        for key, _ in sorted(self._tables.items(), key=lambda kv: kv[1]):
            b.add_labeling_table(*key)
        for c, op in self._factors:
            if op.kind == M.F_VECTOR:
                b.add_vector_factors(c.factor_no, op.cost[None, :], implicit_origin=op.implicit_origin)
            elif op.kind == M.F_PAIRWISE_DENSE:
                b.add_dense_pairwise(c.factor_no, op.table[None])
            else:
                b.add_potts_pairwise(c.factor_no, op.dim, [op.diff])
        for mc, l, r in self._messages:
            b.add_messages(mc.message_no, l, r)
        if self._rel_fwd:
            a = np.asarray(self._rel_fwd, np.int32); b.add_forward_relations(a[:, 0], a[:, 1])
        if self._rel_bwd:
            a = np.asarray(self._rel_bwd, np.int32); b.add_backward_relations(a[:, 0], a[:, 1])
        if self._partition_graph:
            a = np.asarray(self._partition_graph, np.int32); b.put_in_same_partition(a[:, 0], a[:, 1])
        b.constant = self._constant
        m = b.finish()
        if self._duals_host is not None:     # duals of factors that existed before a structural change
            n = min(self._duals_host.shape[0], m.dual_data.shape[0])
            m.dual_data[:n] = self._duals_host[:n]
        return m

    def _pull_duals(self):
        if self._engine is not None and not self._dirty:
            self._duals_host = self._engine.download_duals()

    def _ready(self) -> Engine:
        if len(self._factors) <= 1:
            raise RuntimeError("LP needs more than one factor")           # reference assert, LP_MP.h:708
        if self._engine is None:
            self._engine = Engine(self._device)
            self._engine.set_speculation(self._speculation)
        if self._dirty:
            self._model = self.flat_model()
            self._engine.upload(self._model)
            self._dirty = False
        self._engine.set_inner_iterations(self._inner)
        self._engine.set_reparametrization_type(self._rtype)
        return self._engine

    # -- the hot path ---------------------------------------------------------------------------------------
    def _mode(self) -> int:
        if self._repam == LPReparametrizationMode.Undefined:
            raise RuntimeError("no reparametrization mode set")           # reference LP_MP.h:458
        return self._repam

    def ComputePass(self, iteration=0, *rows):
        """ComputePass(iteration)  or  ComputePass(factors, (omega_off, omega), (mask_off, mask))."""
        e = self._ready()
        if rows:
            factors = iteration
            (om_off, om), (mk_off, mk) = rows
            e.compute_pass_custom(factors, om_off, om, mk_off, mk)
            return
        e.set_reparametrization(self._mode())
        e.compute_pass(1)

    def ComputePasses(self, n: int):
        """n consecutive passes in one call: same results as n x ComputePass, the engine joins them"""
        if n > 0:
            e = self._ready(); e.set_reparametrization(self._mode()); e.compute_pass(int(n))

    def ComputeForwardPass(self):
        e = self._ready(); e.set_reparametrization(self._mode()); e.forward_pass()

    def ComputeBackwardPass(self):
        e = self._ready(); e.set_reparametrization(self._mode()); e.backward_pass()

    def LowerBound(self) -> float:
        return self._ready().lower_bound()

    # -- primal rounding inside the sweep (reference LP_MP.h:914-940, 1067-1082, 1521-1536) ------------------
    def ComputeForwardPassAndPrimal(self, iteration: int):
        e = self._ready(); e.set_reparametrization(self._mode()); e.forward_pass_and_primal(iteration)

    def ComputeBackwardPassAndPrimal(self, iteration: int):
        e = self._ready(); e.set_reparametrization(self._mode()); e.backward_pass_and_primal(iteration)

    def ComputePassAndPrimal(self, iteration: int):
        self.ComputeForwardPassAndPrimal(iteration)
        self.ComputeBackwardPassAndPrimal(iteration)

    def CheckPrimalConsistency(self) -> bool:
        return self._ready().check_primal_consistency()

    def EvaluatePrimal(self) -> float:
        return self._ready().evaluate_primal()

    def primal(self) -> np.ndarray:
        """[n_factors, 2] the factors' primal_ members in serialize_primal order: vector factor (label, 0), pairwise
        factor (x0, x1); an unset entry holds the dimension."""
        return self._ready().download_primal()

    def get_omega(self):
        """omega_storage{forward, backward, receive_mask_forward, receive_mask_backward} as CSR pairs."""
        p = self._ready().plan
        m = self._mode()
        return {"forward": p.omega(0, m), "backward": p.omega(1, m),
                "receive_mask_forward": p.mask(0, m), "receive_mask_backward": p.mask(1, m)}

    def forward_update_ordering(self):
        return self._ready().plan.update_order(0)

    def backward_update_ordering(self):
        return self._ready().plan.update_order(1)

    def duals(self) -> np.ndarray:
        """packed duals in serialize_dual order (reference factors_messages.hxx:3196-3223)."""
        return self._ready().download_duals()


# ---- caller loop --------------------------------------------------------------------------------------------
@dataclass
class LpControl:
    repam: int = LPReparametrizationMode.Undefined
    computePrimal: bool = False
    computeLowerBound: bool = False
    tighten: bool = False
    end: bool = False
    error: bool = False


class StandardVisitor:
    """reference include/visitors/standard_visitor.hxx; same option names and defaults (:32-44)."""

    def __init__(self, maxIter=1000, timeout=None, primalComputationInterval=5, primalComputationStart=1,
                 lowerBoundComputationInterval=1, minDualImprovement=None, minDualImprovementInterval=10,
                 standardReparametrization="anisotropic", roundingReparametrization="damped_uniform", verbosity=0):
        self.maxIter, self.timeout = maxIter, timeout
        self.primalComputationInterval, self.primalComputationStart = primalComputationInterval, primalComputationStart
        self.lowerBoundComputationInterval = lowerBoundComputationInterval
        self.minDualImprovement, self.minDualImprovementInterval = minDualImprovement, minDualImprovementInterval
        self.standardReparametrization = LPReparametrizationModeConvert(standardReparametrization)
        self.roundingReparametrization = LPReparametrizationModeConvert(roundingReparametrization)
        self.verbosity = verbosity
        self.lowerBound_: List[float] = []

    def begin(self, lp) -> LpControl:
        self.remainingIter = self.maxIter
        self.curIter = 0
        self.beginTime = time.monotonic()
        return LpControl(repam=self.standardReparametrization, computeLowerBound=True)

    def visit(self, c: LpControl, lowerBound: float, primalBound: float) -> LpControl:
        self.lowerBound_.append(lowerBound)
        elapsed = time.monotonic() - self.beginTime
        if self.verbosity >= 1 and (c.computePrimal or c.computeLowerBound):
            print(f"iteration = {self.curIter}, lower bound = {lowerBound}, time elapsed = {elapsed:.2f}s")
        self.curIter += 1
        self.remainingIter -= 1
        ret = LpControl()
        if self.remainingIter == 0:
            ret.end = True
            return ret
        if primalBound <= lowerBound + 1e-8:
            ret.end = True
            return ret
        if self.timeout is not None and elapsed >= self.timeout:
            self.remainingIter = min(1, self.remainingIter)
        # (reference standard_visitor.hxx:163-165 starts one visit earlier and then indexes out of bounds; see LP_gpu.hxx)
        if (c.computeLowerBound and len(self.lowerBound_) > self.minDualImprovementInterval and self.minDualImprovement is not None):
            prev = self.lowerBound_[len(self.lowerBound_) - 1 - self.minDualImprovementInterval]
            if self.minDualImprovement > 0 and lowerBound - prev < self.minDualImprovement:
                self.remainingIter = min(1, self.remainingIter)
        if self.remainingIter == 1:
            ret.computePrimal = True
            ret.computeLowerBound = True
            ret.repam = self.roundingReparametrization
            return ret
        ret.repam = self.standardReparametrization
        if self.curIter >= self.primalComputationStart and (self.curIter - self.primalComputationStart) % self.primalComputationInterval == 0:
            ret.computePrimal = True
            ret.repam = self.roundingReparametrization
        if self.curIter % self.lowerBoundComputationInterval == 0:
            ret.computeLowerBound = True
        return ret

    def quiet_iterations(self, c: LpControl) -> int:
        """how many iterations from now on (this one included) ask for neither a lower bound nor a primal: the solver
        may run them as ONE device call and replay the visits afterwards (LP_gpu_solver.hxx has the same rule)"""
        if c.end or c.error or c.computeLowerBound or c.computePrimal or self.timeout is not None:
            return 1
        n = 1
        for j in range(1, max(1, self.remainingIter - 1)):
            it = self.curIter + j
            if self.remainingIter - j <= 1:
                break
            if it >= self.primalComputationStart and (it - self.primalComputationStart) % self.primalComputationInterval == 0:
                break
            if it % self.lowerBoundComputationInterval == 0:
                break
            n += 1
        return n

    def end(self, lower_bound: float, upper_bound: float):
        if self.verbosity >= 1:
            print(f"final lower bound = {lower_bound}, upper bound = {upper_bound}")


class Solver:
    """Solver<LP_TYPE, VISITOR>::Solve with its PreIterate / Iterate / PostIterate / RegisterPrimal hooks (reference
    include/solver.hxx:230-337).  The base class never rounds (bestPrimalCost_ stays +inf unless a derived solver
    registers a primal); the visitor still switches to the rounding reparametrisation where the reference would."""

    def __init__(self, lp: LP, visitor: Optional[StandardVisitor] = None):
        self.lp_ = lp
        self.visitor_ = visitor or StandardVisitor()
        self.lowerBound_ = -np.inf
        self.bestPrimalCost_ = np.inf
        self.solution_ = None
        self.iter = 0

    def GetLP(self) -> LP:
        return self.lp_

    def PreIterate(self, c: LpControl):
        self.lp_.set_reparametrization(c.repam)

    def Iterate(self, c: LpControl):
        self.lp_.ComputePass(self.iter)

    def PostIterate(self, c: LpControl):
        if c.computeLowerBound:
            self.lowerBound_ = self.lp_.LowerBound()

    def RegisterPrimal(self):
        """solver.hxx:320-337: keep the labeling if it is cheaper than the best one so far and consistent"""
        cost = self.lp_.EvaluatePrimal()
        if cost < self.bestPrimalCost_ and self.lp_.CheckPrimalConsistency():
            self.bestPrimalCost_ = cost
            self.solution_ = self.lp_.primal()

    def Solve(self) -> int:
        self.lp_.Begin()
        c = self.visitor_.begin(self.lp_)
        while not c.end and not c.error:
            self.PreIterate(c)
            # iterations in which the visitor asks for nothing run as one device call (the engine joins consecutive
            # passes); only for this class and MpRoundingSolver themselves — a subclass may override Iterate
            quiet = 1
            if type(self) in (Solver, MpRoundingSolver) and hasattr(self.visitor_, "quiet_iterations"):
                quiet = self.visitor_.quiet_iterations(c)
            if quiet > 1:
                self.lp_.ComputePasses(quiet)
                for _ in range(quiet):
                    if c.end or c.error:
                        break
                    c = self.visitor_.visit(c, self.lowerBound_, self.bestPrimalCost_)
                    self.iter += 1
                continue
            self.Iterate(c)
            self.PostIterate(c)
            c = self.visitor_.visit(c, self.lowerBound_, self.bestPrimalCost_)
            self.iter += 1
        if not c.error:
            self.lp_.End()
            if self._rounds:
                self.RegisterPrimal()
            self.lowerBound_ = self.lp_.LowerBound()
            self.visitor_.end(self.lowerBound_, self.bestPrimalCost_)
        return int(not c.error)

    _rounds = False   # the reference calls RegisterPrimal after End() in every solver (solver.hxx:247); without a
                      # rounding pass every primal_ is unset and the cost is +inf, so the base class skips the call

    def lower_bound(self) -> float:
        return self.lowerBound_

    def primal_cost(self) -> float:
        return self.bestPrimalCost_


class MpRoundingSolver(Solver):
    """MpRoundingSolver<SOLVER>: local rounding interleaved with message passing (reference include/solver.hxx:380-400).
    On the iterations the visitor marks computePrimal (every --primalComputationInterval-th, and the last one) the
    pass is run as forward-and-primal, register, backward-and-primal, register."""
    _rounds = True

    def Iterate(self, c: LpControl):
        if c.computePrimal:
            self.lp_.ComputeForwardPassAndPrimal(self.iter)
            self.RegisterPrimal()
            self.lp_.ComputeBackwardPassAndPrimal(self.iter)
            self.RegisterPrimal()
        else:
            super().Iterate(c)
