"""Flat factor-graph model: the Python view of ``include/lpmp_model.h``.

``ModelBuilder`` collects what a user of the reference adds through ``LP<FMC>::add_factor`` /
``add_message`` / ``AddFactorRelation`` (reference: include/LP_MP.h:239-285, :698-702) and
``finish()`` packs it into the arrays the C-ABI engine (include/lpmp_engine.h) takes.
Bulk ``add_*`` calls keep insertion order: ids are handed out consecutively, exactly as the
reference appends to ``f_`` / ``m_``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# enum lpmp_factor_kind
F_VECTOR, F_PAIRWISE_DENSE, F_PAIRWISE_POTTS = 0, 1, 2
FF_IMPLICIT_ORIGIN = 1
# enum lpmp_msg_kind
M_UNARY_PAIRWISE, M_LABELING, M_MINNORM = 0, 1, 2
# enum lpmp_schedule (reference include/config.hxx:43-49)
SCHED_LEFT, SCHED_RIGHT, SCHED_FULL, SCHED_ONLY_SEND, SCHED_NONE = 0, 1, 2, 3, 4
# enum lpmp_repam_mode (reference include/config.hxx:71)
REPAM_ANISOTROPIC, REPAM_ANISOTROPIC2, REPAM_UNIFORM, REPAM_DAMPED_UNIFORM, REPAM_MIXED = 0, 1, 2, 3, 4
REPAM_NAMES = {"anisotropic": 0, "anisotropic2": 1, "uniform": 2, "damped_uniform": 3, "mixed": 4}
FORWARD, BACKWARD = 0, 1
# enum lpmp_msg_flags
MF_IMPROVEMENT, MF_BATCH_TO_RIGHT, MF_BATCH_TO_LEFT = 1, 2, 4
# enum lpmp_reparametrization_type (reference --reparametrizationType, LP_MP.h:710-722)
RTYPE_SHARED, RTYPE_RESIDUAL, RTYPE_PARTITION, RTYPE_OVERLAPPING_PARTITION, RTYPE_ADAPTIVE = 0, 1, 2, 3, 4
RTYPE_NAMES = {"shared": 0, "residual": 1, "partition": 2, "overlapping_partition": 3, "adaptive": 4}
# NO_OF_LEFT/RIGHT_FACTORS shorthands (reference include/config.hxx:60-66)
variableMessageNumber = 0
atMostOneMessage, atMostTwoMessages = -1, -2


class c_msg_type(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("left_ftype", "right_ftype", "schedule", "n_left", "n_right", "kind", "param", "flags")]


class c_model(C.Structure):
    _fields_ = [
        ("n_ftypes", C.c_int32), ("ftype_computes_primal", C.c_void_p),
        ("n_mtypes", C.c_int32), ("mtypes", C.c_void_p),
        ("n_tables", C.c_int32), ("tab_off", C.c_void_p), ("tab_data", C.c_void_p), ("tab_nleft", C.c_void_p),
        ("n_factors", C.c_int64), ("f_type", C.c_void_p), ("f_kind", C.c_void_p), ("f_flags", C.c_void_p),
        ("f_dim0", C.c_void_p), ("f_dim1", C.c_void_p), ("const_data", C.c_void_p), ("dual_data", C.c_void_p),
        ("n_messages", C.c_int64), ("m_type", C.c_void_p), ("m_left", C.c_void_p), ("m_right", C.c_void_p),
        ("n_rel_fwd", C.c_int64), ("rel_fwd", C.c_void_p), ("n_rel_bwd", C.c_int64), ("rel_bwd", C.c_void_p),
        ("constant", C.c_double),
        ("n_part_pairs", C.c_int64), ("part_pairs", C.c_void_p),
    ]


@dataclass
class MsgType:
    """One entry of ``FMC::MessageList`` (reference include/factors_messages.hxx:571-578)."""
    left_ftype: int
    right_ftype: int
    schedule: int = SCHED_LEFT
    n_left: int = variableMessageNumber
    n_right: int = 1
    kind: int = M_UNARY_PAIRWISE
    param: int = 0
    flags: int = 0          # MF_* : optional members of the message op (include/lpmp_model.h, enum lpmp_msg_flags)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


@dataclass
class FlatModel:
    n_ftypes: int
    ftype_computes_primal: np.ndarray
    mtypes: List[MsgType]
    tab_off: np.ndarray
    tab_data: np.ndarray
    tab_nleft: np.ndarray
    f_type: np.ndarray
    f_kind: np.ndarray
    f_flags: np.ndarray
    f_dim0: np.ndarray
    f_dim1: np.ndarray
    const_data: Optional[np.ndarray]   # None: supplied separately as a device buffer
    dual_data: Optional[np.ndarray]
    m_type: np.ndarray
    m_left: np.ndarray
    m_right: np.ndarray
    rel_fwd: np.ndarray                # [n,2] int32
    rel_bwd: np.ndarray
    constant: float = 0.0
    part_pairs: Optional[np.ndarray] = None   # [n,2] int32: put_in_same_partition(f1, f2) calls in call order
    _keep: list = field(default_factory=list, repr=False)

    def __getstate__(self):          # the ctypes views of c_struct() are per process and rebuilt on demand
        d = dict(self.__dict__)
        d["_keep"] = []
        return d

    @property
    def n_factors(self) -> int:
        return int(self.f_type.shape[0])

    @property
    def n_messages(self) -> int:
        return int(self.m_type.shape[0])

    def const_sizes(self) -> np.ndarray:
        d0 = self.f_dim0.astype(np.int64)
        d1 = self.f_dim1.astype(np.int64)
        return np.where(self.f_kind == F_PAIRWISE_DENSE, d0 * d1, np.where(self.f_kind == F_PAIRWISE_POTTS, 1, 0))

    def dual_sizes(self) -> np.ndarray:
        d0 = self.f_dim0.astype(np.int64)
        d1 = self.f_dim1.astype(np.int64)
        return np.where(self.f_kind == F_PAIRWISE_DENSE, d0 + d1, np.where(self.f_kind == F_PAIRWISE_POTTS, 2 * d0, d0))

    def dual_offsets(self) -> np.ndarray:
        return np.concatenate([[0], np.cumsum(self.dual_sizes())]).astype(np.int64)

    def const_offsets(self) -> np.ndarray:
        return np.concatenate([[0], np.cumsum(self.const_sizes())]).astype(np.int64)

    def with_factor_order(self, rank: np.ndarray) -> "FlatModel":
        """the same factors, messages and costs with the factor relations REPLACED by a chain through all factors in the order
        ``rank[f]`` (position of factor f, e.g. Plan.suggest_order): AddFactorRelation(by_rank[i], by_rank[i + 1]) for consecutive
        positions (reference LP_MP.h:698-702: forward sweep in that order, backward sweep in the reverse one; the chain is the only
        topological order, so no tie is left to the sort) — what INTEGRATION.md 2a shows a C++ caller doing with
        lpmp_plan_suggest_order's answer.  Arrays are shared with ``self``."""
        import dataclasses
        rank = np.asarray(rank, np.int64)
        by_rank = np.empty(rank.shape[0], np.int32)
        by_rank[rank] = np.arange(rank.shape[0], dtype=np.int32)
        fwd = np.ascontiguousarray(np.stack([by_rank[:-1], by_rank[1:]], 1))
        return dataclasses.replace(self, rel_fwd=fwd, rel_bwd=np.ascontiguousarray(fwd[:, ::-1]), _keep=[])

    def dump(self, path: str):
        """the model as one flat binary file in the order of include/lpmp_model.h — what the C++ hosts read
        (lp_mp_amd/include/lpmp_lockstep.hxx, model_file; tools/mgpu_rccl_driver.cpp --model-file)"""
        if self.dual_data is None:
            raise ValueError("dump: the model has no host duals")
        const = np.zeros(0) if self.const_data is None else np.ascontiguousarray(self.const_data, np.float64)
        mt = np.array([[t.left_ftype, t.right_ftype, t.schedule, t.n_left, t.n_right, t.kind, t.param, t.flags] for t in self.mtypes], np.int32).reshape(-1, 8)
        head = np.array([0x4C504D504D4F444C, self.n_ftypes, len(self.mtypes), self.tab_nleft.shape[0], self.tab_data.shape[0], self.n_factors,
                         self.n_messages, self.rel_fwd.shape[0], self.rel_bwd.shape[0], const.shape[0], self.dual_data.shape[0]], np.int64)
        with open(path, "wb") as f:
            f.write(head.tobytes()); f.write(np.float64(self.constant).tobytes())
            for a, dt in ((self.ftype_computes_primal, np.uint8), (mt, np.int32), (self.tab_off, np.int64), (self.tab_data, np.int32), (self.tab_nleft, np.int32),
                          (self.f_type, np.int32), (self.f_kind, np.uint8), (self.f_flags, np.uint8), (self.f_dim0, np.int32), (self.f_dim1, np.int32),
                          (const, np.float64), (self.dual_data, np.float64), (self.m_type, np.int32), (self.m_left, np.int32), (self.m_right, np.int32),
                          (self.rel_fwd, np.int32), (self.rel_bwd, np.int32)):
                f.write(np.ascontiguousarray(a, dt).tobytes())

    def c_struct(self) -> c_model:
        """ctypes view; arrays stay owned by ``self`` (borrowed for the duration of a call)."""
        mt = (c_msg_type * max(1, len(self.mtypes)))()
        for i, t in enumerate(self.mtypes):
            mt[i] = c_msg_type(t.left_ftype, t.right_ftype, t.schedule, t.n_left, t.n_right, t.kind, t.param, t.flags)
        self._keep = [mt]
        m = c_model()
        m.n_ftypes = self.n_ftypes
        m.ftype_computes_primal = _ptr(self.ftype_computes_primal)
        m.n_mtypes = len(self.mtypes)
        m.mtypes = C.addressof(mt)
        m.n_tables = int(self.tab_nleft.shape[0])
        m.tab_off, m.tab_data, m.tab_nleft = _ptr(self.tab_off), _ptr(self.tab_data), _ptr(self.tab_nleft)
        m.n_factors = self.n_factors
        m.f_type, m.f_kind, m.f_flags = _ptr(self.f_type), _ptr(self.f_kind), _ptr(self.f_flags)
        m.f_dim0, m.f_dim1 = _ptr(self.f_dim0), _ptr(self.f_dim1)
        m.const_data, m.dual_data = _ptr(self.const_data), _ptr(self.dual_data)
        m.n_messages = self.n_messages
        m.m_type, m.m_left, m.m_right = _ptr(self.m_type), _ptr(self.m_left), _ptr(self.m_right)
        m.n_rel_fwd, m.rel_fwd = int(self.rel_fwd.shape[0]), _ptr(self.rel_fwd)
        m.n_rel_bwd, m.rel_bwd = int(self.rel_bwd.shape[0]), _ptr(self.rel_bwd)
        m.constant = float(self.constant)
        if self.part_pairs is not None and len(self.part_pairs):
            self.part_pairs = np.ascontiguousarray(self.part_pairs, np.int32).reshape(-1, 2)
            m.n_part_pairs, m.part_pairs = int(self.part_pairs.shape[0]), _ptr(self.part_pairs)
        else:
            m.n_part_pairs, m.part_pairs = 0, None
        return m


class ModelBuilder:
    """Collects factors / messages / relations in insertion order (bulk or one at a time)."""

    def __init__(self, n_ftypes: int, mtypes: Sequence[MsgType], ftype_computes_primal: Optional[Sequence[int]] = None):
        self.n_ftypes = int(n_ftypes)
        self.mtypes = list(mtypes)
        self.ftype_computes_primal = np.zeros(self.n_ftypes, np.uint8) if ftype_computes_primal is None \
            else np.asarray(ftype_computes_primal, np.uint8)
        self._tables: List[np.ndarray] = []
        self._tab_nleft: List[int] = []
        self._f = []      # (type, kind, flags, dim0, dim1) array chunks
        self._const = []
        self._dual = []
        self._nf = 0
        self._m = []
        self._nm = 0
        self._rel_fwd = []
        self._rel_bwd = []
        self._part = []
        self.constant = 0.0
        self.skip_const = False  # True: const tables live only on the device (see Engine.upload)
        self.skip_dual = False   # True: so do the duals (upload(dual_dev=...)): no host copy of them is built

    # -- labeling match tables (reference labeling_list_factor.hxx:384-402) ------------------
    def add_labeling_table(self, left_labelings, right_labelings, indices) -> int:
        """``labeling_message<LEFT, RIGHT, INDICES...>``: table[r] = index of the left labeling whose
        labels equal the right labeling's labels at ``indices``, or n_left if none."""
        left = [tuple(l) for l in left_labelings]
        tab = []
        for r in right_labelings:
            sub = tuple(r[i] for i in indices)
            tab.append(left.index(sub) if sub in left else len(left))
        self._tables.append(np.asarray(tab, np.int32))
        self._tab_nleft.append(len(left))
        return len(self._tables) - 1

    # -- factors ---------------------------------------------------------------------------------
    def _add_factors(self, n, ftype, kind, flags, dim0, dim1, const, dual):
        ids = np.arange(self._nf, self._nf + n, dtype=np.int32)
        self._f.append((np.full(n, ftype, np.int32), np.full(n, kind, np.uint8), np.full(n, flags, np.uint8),
                        np.full(n, dim0, np.int32), np.full(n, dim1, np.int32)))
        if const is not None:
            self._const.append(np.ascontiguousarray(const, np.float64).reshape(-1))
        if not self.skip_dual:
            self._dual.append(np.ascontiguousarray(dual, np.float64).reshape(-1))
        self._nf += n
        return ids

    def add_vector_factors(self, ftype: int, costs, implicit_origin: bool = False, shape=None) -> np.ndarray:
        """UnarySimplexFactor / labeling_factor / test_factor; ``costs`` is [n, dim] (None with skip_dual: ``shape`` = (n, dim))."""
        if costs is None:
            assert self.skip_dual and shape is not None
            n, dim = shape
        else:
            costs = np.atleast_2d(np.asarray(costs, np.float64))
            n, dim = costs.shape
        return self._add_factors(n, ftype, F_VECTOR, FF_IMPLICIT_ORIGIN if implicit_origin else 0, dim, 0, None, costs)

    def add_dense_pairwise(self, ftype: int, tables=None, n: Optional[int] = None, dims=None) -> np.ndarray:
        """PairwiseSimplexFactor; ``tables`` is [n, d0, d1] (row-major) or None with skip_const."""
        if tables is not None:
            tables = np.asarray(tables, np.float64)
            if tables.ndim == 2:
                tables = tables[None]
            n, d0, d1 = tables.shape
        else:
            assert self.skip_const and n is not None and dims is not None
            d0, d1 = dims
        return self._add_factors(n, ftype, F_PAIRWISE_DENSE, 0, d0, d1, tables, None if self.skip_dual else np.zeros((n, d0 + d1)))

    def add_potts_pairwise(self, ftype: int, n_labels: int, diffs) -> np.ndarray:
        """pairwise_potts_factor(n_labels, diff)."""
        diffs = np.atleast_1d(np.asarray(diffs, np.float64))
        n = diffs.shape[0]
        return self._add_factors(n, ftype, F_PAIRWISE_POTTS, 0, n_labels, n_labels, diffs, None if self.skip_dual else np.zeros((n, 2 * n_labels)))

    # -- messages / relations ------------------------------------------------------------------------
    def add_messages(self, mtype: int, left, right) -> np.ndarray:
        left = np.atleast_1d(np.asarray(left, np.int32))
        right = np.atleast_1d(np.asarray(right, np.int32))
        assert left.shape == right.shape
        n = left.shape[0]
        ids = np.arange(self._nm, self._nm + n, dtype=np.int64)
        self._m.append((np.full(n, mtype, np.int32), left, right))
        self._nm += n
        return ids

    def add_interleaved_messages(self, mtypes, left, right, return_ids: bool = True):
        """Messages of several types in one insertion sequence (row i has type mtypes[i])."""
        mtypes = np.asarray(mtypes, np.int32)
        left = np.asarray(left, np.int32)
        right = np.asarray(right, np.int32)
        n = left.shape[0]
        ids = np.arange(self._nm, self._nm + n, dtype=np.int64) if return_ids else None
        self._m.append((mtypes, left, right))
        self._nm += n
        return ids

    def add_relations(self, f1, f2):
        """AddFactorRelation(f1, f2): f1 before f2 forward, f2 before f1 backward."""
        f1 = np.atleast_1d(np.asarray(f1, np.int32))
        f2 = np.atleast_1d(np.asarray(f2, np.int32))
        fwd = np.empty((f1.shape[0], 2), np.int32); fwd[:, 0] = f1; fwd[:, 1] = f2
        bwd = np.empty((f1.shape[0], 2), np.int32); bwd[:, 0] = f2; bwd[:, 1] = f1
        self._rel_fwd.append(fwd)
        self._rel_bwd.append(bwd)

    def put_in_same_partition(self, f1, f2):
        """LP::put_in_same_partition(f1, f2) (reference LP_MP.h:465)"""
        self._part.append(np.stack([np.atleast_1d(np.asarray(f1, np.int32)), np.atleast_1d(np.asarray(f2, np.int32))], 1))

    def add_forward_relations(self, f1, f2):
        self._rel_fwd.append(np.stack([np.atleast_1d(np.asarray(f1, np.int32)), np.atleast_1d(np.asarray(f2, np.int32))], 1))

    def add_backward_relations(self, f1, f2):
        self._rel_bwd.append(np.stack([np.atleast_1d(np.asarray(f1, np.int32)), np.atleast_1d(np.asarray(f2, np.int32))], 1))

    def finish(self) -> FlatModel:
        def cat(chunks, dtype, shape=(0,)):
            if len(chunks) == 1:                       # (models of millions of factors: no second copy of a single block)
                return np.ascontiguousarray(chunks[0])
            return np.ascontiguousarray(np.concatenate(chunks)) if chunks else np.zeros(shape, dtype)
        tab_off = np.concatenate([[0], np.cumsum([len(t) for t in self._tables])]).astype(np.int64)
        return FlatModel(
            n_ftypes=self.n_ftypes, ftype_computes_primal=self.ftype_computes_primal, mtypes=self.mtypes,
            tab_off=tab_off, tab_data=cat(self._tables, np.int32), tab_nleft=np.asarray(self._tab_nleft, np.int32),
            f_type=cat([c[0] for c in self._f], np.int32), f_kind=cat([c[1] for c in self._f], np.uint8),
            f_flags=cat([c[2] for c in self._f], np.uint8), f_dim0=cat([c[3] for c in self._f], np.int32),
            f_dim1=cat([c[4] for c in self._f], np.int32),
            const_data=None if self.skip_const else cat(self._const, np.float64),
            dual_data=None if self.skip_dual else cat(self._dual, np.float64),
            m_type=cat([c[0] for c in self._m], np.int32), m_left=cat([c[1] for c in self._m], np.int32),
            m_right=cat([c[2] for c in self._m], np.int32),
            rel_fwd=cat(self._rel_fwd, np.int32, (0, 2)).astype(np.int32, copy=False).reshape(-1, 2),
            rel_bwd=cat(self._rel_bwd, np.int32, (0, 2)).astype(np.int32, copy=False).reshape(-1, 2),
            constant=self.constant,
            part_pairs=cat(self._part, np.int32, (0, 2)).astype(np.int32).reshape(-1, 2) if self._part else None)
