"""ctypes binding of oracle/liblpmp_oracle.so — TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liblpmp_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("lpmp_oracle.c", "lpmp_oracle.h", "gen_mt19937.cpp")] + \
          [os.path.join(_HERE, "..", "include", "lpmp_model.h")]
    stale = force or not os.path.exists(so) or not os.path.exists(os.path.join(_HERE, "gen_mt19937")) or \
        any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_lower_bound.restype = C.c_double
        L.orc_lower_bound.argtypes = [C.c_void_p]
        L.orc_factor_lower_bound.restype = C.c_double
        L.orc_factor_lower_bound.argtypes = [C.c_void_p, C.c_int64]
        for name in ("orc_n_factors", "orc_dual_size", "orc_msg_list_size"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("orc_n_updated", "orc_omega_nnz", "orc_mask_nnz"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_reparametrization_type.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_inner_iterations.argtypes = [C.c_void_p, C.c_int]
        L.orc_n_partitions.restype = C.c_int64
        L.orc_n_partitions.argtypes = [C.c_void_p]
        L.orc_get_partitions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_compute_pass.argtypes = [C.c_void_p, C.c_int]
        L.orc_forward_pass.argtypes = [C.c_void_p]
        L.orc_backward_pass.argtypes = [C.c_void_p]
        for name in ("orc_forward_pass_and_primal", "orc_backward_pass_and_primal", "orc_compute_pass_and_primal"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_uint64]
        L.orc_check_primal_consistency.argtypes = [C.c_void_p]
        L.orc_evaluate_primal.restype = C.c_double
        L.orc_evaluate_primal.argtypes = [C.c_void_p]
        L.orc_get_primal.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_primal_access.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_duals.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_duals.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_get_update_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_get_omega.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_get_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_get_msg_lists.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_compute_pass_custom.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.orc_sublist_nnz.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_anisotropic_weights_sublist.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.orc_get_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_message_value.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_void_p]
        L.orc_synth_u01.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64]
        _LIB = L
    return _LIB


class Oracle:
    """Reference-semantics LP over a FlatModel (lp_mp_amd.model.FlatModel)."""

    def __init__(self, model):
        self.L = lib()
        self.model = model
        cs = model.c_struct()
        self.h = self.L.orc_create(C.addressof(cs))
        if not self.h:
            raise RuntimeError(self.L.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.orc_last_error().decode())

    def set_reparametrization(self, mode: int):
        self._chk(self.L.orc_set_mode(self.h, int(mode)))

    def set_reparametrization_type(self, rtype: int):
        self._chk(self.L.orc_set_reparametrization_type(self.h, int(rtype)))

    def set_inner_iterations(self, n: int):
        """--innerIteration of the partition sweeps (reference LP_MP.h:590)"""
        self._chk(self.L.orc_set_inner_iterations(self.h, int(n)))

    def partitions(self):
        """LP::construct_factor_partition: list of factor-index arrays (updated factors only)"""
        n = self.L.orc_n_partitions(self.h)
        if n < 0:
            self._chk(-1)
        off = np.empty(n + 1, np.int64)
        f = np.empty(max(self.L.orc_n_updated(self.h, 0), 1), np.int32)
        self._chk(self.L.orc_get_partitions(self.h, off.ctypes.data, f.ctypes.data))
        return [f[off[i]:off[i + 1]].copy() for i in range(n)]

    def ComputePass(self, n: int = 1):
        self._chk(self.L.orc_compute_pass(self.h, int(n)))

    def ComputeForwardPass(self):
        self._chk(self.L.orc_forward_pass(self.h))

    def ComputeBackwardPass(self):
        self._chk(self.L.orc_backward_pass(self.h))

    def LowerBound(self) -> float:
        return float(self.L.orc_lower_bound(self.h))

    def ComputeForwardPassAndPrimal(self, iteration: int):
        self._chk(self.L.orc_forward_pass_and_primal(self.h, int(iteration)))

    def ComputeBackwardPassAndPrimal(self, iteration: int):
        self._chk(self.L.orc_backward_pass_and_primal(self.h, int(iteration)))

    def ComputePassAndPrimal(self, iteration: int):
        self._chk(self.L.orc_compute_pass_and_primal(self.h, int(iteration)))

    def CheckPrimalConsistency(self) -> bool:
        return bool(self.L.orc_check_primal_consistency(self.h))

    def EvaluatePrimal(self) -> float:
        return float(self.L.orc_evaluate_primal(self.h))

    def primal(self) -> np.ndarray:
        """[n_factors, 2] primal_ members: vector (label, 0), pairwise (x0, x1); unset = the dim"""
        out = np.empty((self.L.orc_n_factors(self.h), 2), np.int32)
        self.L.orc_get_primal(self.h, out.ctypes.data)
        return out

    def primal_access(self) -> np.ndarray:
        out = np.empty(self.L.orc_n_factors(self.h), np.uint64)
        self.L.orc_get_primal_access(self.h, out.ctypes.data)
        return out

    def factor_lower_bound(self, f: int) -> float:
        return float(self.L.orc_factor_lower_bound(self.h, int(f)))

    def duals(self) -> np.ndarray:
        out = np.empty(self.L.orc_dual_size(self.h), np.float64)
        self.L.orc_get_duals(self.h, out.ctypes.data)
        return out

    def set_duals(self, d: np.ndarray):
        d = np.ascontiguousarray(d, np.float64)
        assert d.shape[0] == self.L.orc_dual_size(self.h)
        self.L.orc_set_duals(self.h, d.ctypes.data)

    def order(self, direction: int) -> np.ndarray:
        out = np.empty(self.L.orc_n_factors(self.h), np.int32)
        self.L.orc_get_order(self.h, direction, out.ctypes.data)
        return out

    def update_order(self, direction: int) -> np.ndarray:
        out = np.empty(self.L.orc_n_updated(self.h, direction), np.int32)
        self.L.orc_get_update_order(self.h, direction, out.ctypes.data)
        return out

    def omega(self, direction: int, mode: int):
        n = self.L.orc_n_updated(self.h, direction)
        off = np.empty(n + 1, np.int64)
        data = np.empty(self.L.orc_omega_nnz(self.h, direction), np.float64)
        self._chk(self.L.orc_get_omega(self.h, direction, mode, off.ctypes.data, data.ctypes.data))
        return off, data

    def mask(self, direction: int, mode: int):
        n = self.L.orc_n_updated(self.h, direction)
        off = np.empty(n + 1, np.int64)
        data = np.empty(self.L.orc_mask_nnz(self.h, direction), np.uint8)
        self._chk(self.L.orc_get_mask(self.h, direction, mode, off.ctypes.data, data.ctypes.data))
        return off, data

    def msg_lists(self):
        off = np.empty(self.L.orc_n_factors(self.h) + 1, np.int64)
        ent = np.empty(self.L.orc_msg_list_size(self.h), np.int64)
        self.L.orc_get_msg_lists(self.h, off.ctypes.data, ent.ctypes.data)
        return off, ent

    def compute_pass_custom(self, factors, om_off, om, mk_off, mk):
        factors = np.ascontiguousarray(factors, np.int32)
        om_off = np.ascontiguousarray(om_off, np.int64)
        om = np.ascontiguousarray(om, np.float64)
        mk_off = np.ascontiguousarray(mk_off, np.int64)
        mk = np.ascontiguousarray(mk, np.uint8)
        self._chk(self.L.orc_compute_pass_custom(self.h, factors.shape[0], factors.ctypes.data, om_off.ctypes.data,
                                                 om.ctypes.data, mk_off.ctypes.data, mk.ctypes.data))

    def anisotropic_weights_sublist(self, factors):
        factors = np.ascontiguousarray(factors, np.int32)
        nr, a, b = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.L.orc_sublist_nnz(self.h, factors.shape[0], factors.ctypes.data, C.addressof(nr),
                                         C.addressof(a), C.addressof(b)))
        om_off = np.empty(nr.value + 1, np.int64)
        mk_off = np.empty(nr.value + 1, np.int64)
        om = np.empty(a.value, np.float64)
        mk = np.empty(b.value, np.uint8)
        self._chk(self.L.orc_anisotropic_weights_sublist(self.h, factors.shape[0], factors.ctypes.data,
                                                         om_off.ctypes.data, om.ctypes.data, mk_off.ctypes.data,
                                                         mk.ctypes.data))
        return om_off, om, mk_off, mk

    def counters(self):
        r, s = C.c_int64(), C.c_int64()
        self.L.orc_get_counters(self.h, C.addressof(r), C.addressof(s))
        return r.value, s.value

    def message_value(self, msg: int, to_left: bool, omega: float = 1.0) -> np.ndarray:
        n = int(self.model.f_dim0[self.model.m_left[msg]])
        out = np.empty(n, np.float64)
        self._chk(self.L.orc_message_value(self.h, msg, 1 if to_left else 0, omega, out.ctypes.data))
        return out


def synth_u01(n: int, seed: int, first: int = 0) -> np.ndarray:
    out = np.empty(n, np.float64)
    lib().orc_synth_u01(out.ctypes.data, n, C.c_uint64(seed), C.c_uint64(first))
    return out


def mt19937_u01(seed: int, count: int) -> np.ndarray:
    """std::mt19937_64(seed) + uniform_real_distribution<double>(0,1) (libstdc++), via oracle/gen_mt19937."""
    import tempfile
    build()
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        subprocess.check_call([os.path.join(_HERE, "gen_mt19937"), str(seed), str(count), f.name])
        return np.fromfile(f.name, np.float64, count)
