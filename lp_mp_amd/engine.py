"""ctypes binding of the C ABI in include/lpmp_engine.h (liblpmp_engine.so).

No CPU fallback: ``Engine`` raises if the HIP extension is missing or no GPU is present.
``Plan`` (host-only analysis: ordering, weights, level schedule) works without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import build as _build
from .model import FlatModel

_LIB = None
N_KCLASS = 23
KCLASS_NAMES = ["generic", "dense4", "dense8", "dense16", "dense32", "potts4", "potts8", "potts16", "potts32",
                "dense_v4", "dense_v8", "dense_v16", "dense_v32", "potts_v4", "potts_v8", "potts_v16", "potts_v32",
                "dense_big", "small", "pairwise4", "pairwise8", "pairwise16", "pairwise32"]
# kernel symbols as rocprofv3 prints them: dense classes run the packed kernel (KMAX 2 at L >= 16, 4 below) whenever a
# launch's factors have at most 8 active messages, else sweep_dense_kernel<L>; the _v classes (any label count up to
# the padded width, rectangular tables) are the same kernels with run-time dims; the exact dense kernels come in a
# plain and a non-temporal-access form (last template argument), chosen by the size of the model: the names below
# are prefixes
KERNEL_NAMES = ["sweep_generic_kernel<64>", "sweep_dense_pk_kernel<4, 4, false", "sweep_dense_pk_kernel<8, 4, false",
                "sweep_dense_pk_kernel<16, 2, false", "sweep_dense_pk_kernel<32, 2, false",
                "sweep_potts_pk_kernel<4, false", "sweep_potts_pk_kernel<8, false", "sweep_potts_pk_kernel<16, false",
                "sweep_potts_pk_kernel<32, false",
                "sweep_dense_pk_kernel<4, 4, true, false>", "sweep_dense_pk_kernel<8, 4, true, false>",
                "sweep_dense_pk_kernel<16, 2, true, false>", "sweep_dense_pk_kernel<32, 2, true, false>",
                "sweep_potts_pk_kernel<4, true, false>", "sweep_potts_pk_kernel<8, true, false>",
                "sweep_potts_pk_kernel<16, true, false>", "sweep_potts_pk_kernel<32, true, false>", "sweep_dense_big_kernel", "sweep_generic_kernel<1>",
                "sweep_pairwise_pk_kernel<4>", "sweep_pairwise_pk_kernel<8>", "sweep_pairwise_pk_kernel<16>",
                "sweep_pairwise_pk_kernel<32>"]
MEM_HOST, MEM_DEVICE = 0, 1

EXPORTS = [
    "lpmp_last_error", "lpmp_version", "lpmp_experiment_build", "lpmp_set_rows_layout", "lpmp_rows_layout", "lpmp_lower_bound_recomputed", "lpmp_plan_create", "lpmp_plan_destroy", "lpmp_plan_n_factors",
    "lpmp_plan_n_updated", "lpmp_plan_get_order", "lpmp_plan_get_update_order", "lpmp_plan_omega_nnz",
    "lpmp_plan_mask_nnz", "lpmp_plan_get_omega", "lpmp_plan_get_mask", "lpmp_plan_get_msg_lists",
    "lpmp_plan_anisotropic_weights", "lpmp_plan_schedule_info", "lpmp_plan_custom_schedule_info", "lpmp_plan_schedule_classes", "lpmp_plan_get_update_levels", "lpmp_plan_pass_schedule_info", "lpmp_plan_pass_rotates", "lpmp_plan_chain_info", "lpmp_plan_mailbox_info", "lpmp_create", "lpmp_destroy", "lpmp_set_stream",
    "lpmp_upload_model", "lpmp_set_reparametrization", "lpmp_set_reparametrization_type", "lpmp_set_inner_iterations", "lpmp_plan_get_partitions", "lpmp_compute_pass", "lpmp_compute_forward_pass",
    "lpmp_compute_backward_pass", "lpmp_compute_pass_custom", "lpmp_schedule_create", "lpmp_schedule_create_fused", "lpmp_schedule_run",
    "lpmp_schedule_info", "lpmp_schedule_destroy", "lpmp_lower_bound", "lpmp_factor_lower_bounds",
    "lpmp_invalidate_lower_bounds", "lpmp_synchronize", "lpmp_dual_size", "lpmp_download_duals", "lpmp_upload_duals", "lpmp_device_duals",
    "lpmp_engine_plan", "lpmp_engine_plan_mut", "lpmp_enable_kernel_timing", "lpmp_get_kernel_timing",
    "lpmp_reset_kernel_timing", "lpmp_get_chain_launches", "lpmp_prepare_passes", "lpmp_synth_fill", "lpmp_compute_forward_pass_and_primal",
    "lpmp_compute_backward_pass_and_primal", "lpmp_compute_pass_and_primal", "lpmp_check_primal_consistency",
    "lpmp_evaluate_primal", "lpmp_download_primal", "lpmp_upload_primal", "lpmp_streaming_access",
    "lpmp_boundary_create", "lpmp_boundary_destroy", "lpmp_boundary_out_doubles", "lpmp_boundary_in_doubles", "lpmp_boundary_pack",
    "lpmp_boundary_reply", "lpmp_boundary_fold", "lpmp_engine_stream", "lpmp_synth_fill_blocks",
    "lpmp_halo_create", "lpmp_halo_destroy", "lpmp_halo_out_doubles", "lpmp_halo_in_doubles", "lpmp_halo_pack", "lpmp_halo_unpack",
    "lpmp_set_speculation", "lpmp_speculation_stats", "lpmp_chain_cache_bytes",
    "lpmp_set_persistent_launches", "lpmp_persistent_launches", "lpmp_device_identity",
    "lpmp_plan_suggest_order", "lpmp_graph_colour_major_order", "lpmp_graph_refine_partition",
]


def library_path() -> str:
    # LPMP_ENGINE_SO: load an experimental build of the same sources (kernel ablations); never a CPU path
    return os.environ.get("LPMP_ENGINE_SO", _build.SO)


def lib():
    """Loads the HIP extension; raises (never falls back) if it is not built."""
    global _LIB
    if _LIB is None:
        # torch ships its own HIP runtime; it must be the one the process loads first, otherwise torch
        # finds no GPU once another libamdhip64 has initialised the device
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        so = library_path()
        if not os.path.exists(so):
            raise RuntimeError(f"{so} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the engine has no CPU fallback)")
        L = C.CDLL(so)
        L.lpmp_last_error.restype = C.c_char_p
        L.lpmp_version.restype = C.c_char_p
        for n in ("lpmp_plan_n_factors", "lpmp_dual_size"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("lpmp_plan_n_updated", "lpmp_plan_omega_nnz", "lpmp_plan_mask_nnz"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [C.c_void_p, C.c_int]
        L.lpmp_plan_create.argtypes = [C.c_void_p, C.c_void_p]
        L.lpmp_plan_destroy.argtypes = [C.c_void_p]
        L.lpmp_plan_get_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.lpmp_plan_get_update_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.lpmp_plan_get_omega.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.lpmp_plan_get_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.lpmp_plan_get_msg_lists.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lpmp_plan_anisotropic_weights.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 8
        L.lpmp_plan_schedule_info.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.lpmp_plan_pass_schedule_info.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.lpmp_plan_pass_rotates.argtypes = [C.c_void_p, C.c_int]
        L.lpmp_plan_chain_info.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.lpmp_plan_mailbox_info.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 2
        L.lpmp_plan_get_update_levels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.lpmp_plan_schedule_classes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.lpmp_plan_custom_schedule_info.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 5
        L.lpmp_create.argtypes = [C.c_int, C.c_void_p]
        L.lpmp_destroy.argtypes = [C.c_void_p]
        L.lpmp_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.lpmp_upload_model.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.lpmp_set_reparametrization.argtypes = [C.c_void_p, C.c_int]
        L.lpmp_set_reparametrization_type.argtypes = [C.c_void_p, C.c_int]
        L.lpmp_set_inner_iterations.argtypes = [C.c_void_p, C.c_int]
        L.lpmp_plan_get_partitions.argtypes = [C.c_void_p] * 4
        L.lpmp_compute_pass.argtypes = [C.c_void_p, C.c_int]
        L.lpmp_compute_forward_pass.argtypes = [C.c_void_p]
        L.lpmp_compute_backward_pass.argtypes = [C.c_void_p]
        L.lpmp_compute_pass_custom.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.lpmp_schedule_create.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 6
        L.lpmp_schedule_create_fused.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        L.lpmp_schedule_run.argtypes = [C.c_void_p, C.c_int]
        L.lpmp_schedule_info.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.lpmp_schedule_destroy.argtypes = [C.c_void_p, C.c_int]
        L.lpmp_lower_bound.argtypes = [C.c_void_p, C.c_void_p]
        L.lpmp_factor_lower_bounds.argtypes = [C.c_void_p, C.c_void_p]
        L.lpmp_invalidate_lower_bounds.argtypes = [C.c_void_p]
        L.lpmp_synchronize.argtypes = [C.c_void_p]
        L.lpmp_download_duals.argtypes = [C.c_void_p, C.c_void_p]
        L.lpmp_upload_duals.argtypes = [C.c_void_p, C.c_void_p]
        L.lpmp_device_duals.restype = C.c_void_p
        L.lpmp_device_duals.argtypes = [C.c_void_p]
        L.lpmp_engine_plan.restype = C.c_void_p
        L.lpmp_engine_plan.argtypes = [C.c_void_p]
        L.lpmp_engine_plan_mut.restype = C.c_void_p
        L.lpmp_engine_plan_mut.argtypes = [C.c_void_p]
        for name in ("lpmp_compute_forward_pass_and_primal", "lpmp_compute_backward_pass_and_primal", "lpmp_compute_pass_and_primal"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_uint64]
        for name in ("lpmp_check_primal_consistency", "lpmp_evaluate_primal", "lpmp_download_primal", "lpmp_upload_primal"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.lpmp_streaming_access.argtypes = [C.c_void_p]
        L.lpmp_enable_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        L.lpmp_get_kernel_timing.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.lpmp_reset_kernel_timing.argtypes = [C.c_void_p]
        L.lpmp_get_chain_launches.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.lpmp_prepare_passes.argtypes = [C.c_void_p, C.c_int]
        if hasattr(L, "lpmp_set_speculation"):       # (absent only in an older experimental build loaded through LPMP_ENGINE_SO for an A/B)
            L.lpmp_set_speculation.argtypes = [C.c_void_p, C.c_int]
            L.lpmp_speculation_stats.argtypes = [C.c_void_p] + [C.c_void_p] * 4
            L.lpmp_chain_cache_bytes.restype = C.c_int64
            L.lpmp_chain_cache_bytes.argtypes = [C.c_void_p]
        if hasattr(L, "lpmp_lower_bound_recomputed"):
            L.lpmp_lower_bound_recomputed.restype = C.c_int64
            L.lpmp_lower_bound_recomputed.argtypes = [C.c_void_p]
        if hasattr(L, "lpmp_set_rows_layout"):
            L.lpmp_set_rows_layout.argtypes = [C.c_void_p, C.c_int]
            L.lpmp_rows_layout.argtypes = [C.c_void_p]
        if hasattr(L, "lpmp_set_persistent_launches"):
            L.lpmp_set_persistent_launches.argtypes = [C.c_void_p, C.c_int]
            L.lpmp_persistent_launches.argtypes = [C.c_void_p]
            L.lpmp_device_identity.argtypes = [C.c_int, C.c_char_p, C.c_int64]
        L.lpmp_plan_suggest_order.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.lpmp_graph_colour_major_order.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.lpmp_graph_refine_partition.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_uint64, C.c_void_p]
        L.lpmp_synth_fill.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.lpmp_synth_fill_blocks.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.lpmp_boundary_create.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.lpmp_boundary_destroy.argtypes = [C.c_void_p]
        for n in ("lpmp_boundary_out_doubles", "lpmp_boundary_in_doubles"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [C.c_void_p]
        L.lpmp_boundary_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lpmp_boundary_reply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lpmp_boundary_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lpmp_engine_stream.restype = C.c_void_p
        L.lpmp_engine_stream.argtypes = [C.c_void_p]
        L.lpmp_halo_create.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lpmp_halo_destroy.argtypes = [C.c_void_p]
        for n in ("lpmp_halo_out_doubles", "lpmp_halo_in_doubles"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [C.c_void_p]
        L.lpmp_halo_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lpmp_halo_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class EngineError(RuntimeError):
    """Mirrors the std::runtime_error the reference throws (LP_MP.h:458)."""

    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


def _chk(rc: int):
    if rc != 0:
        raise EngineError(rc, lib().lpmp_last_error().decode())


class Plan:
    """Host-only analysis of a model: orderings, weights, level schedule (no GPU needed)."""

    def __init__(self, model: Optional[FlatModel] = None, _handle=None, _owner=None):
        self.L = lib()
        self._owner = _owner
        if _handle is not None:
            self.h = _handle
            self._own = False
        else:
            cs = model.c_struct()
            h = C.c_void_p()
            _chk(self.L.lpmp_plan_create(C.addressof(cs), C.addressof(h)))
            self.h = h.value
            self._own = True

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "h", None):
            self.L.lpmp_plan_destroy(self.h)
            self.h = None

    @property
    def n_factors(self) -> int:
        return self.L.lpmp_plan_n_factors(self.h)

    def order(self, d: int) -> np.ndarray:
        out = np.empty(self.n_factors, np.int32)
        _chk(self.L.lpmp_plan_get_order(self.h, d, out.ctypes.data))
        return out

    def update_order(self, d: int) -> np.ndarray:
        out = np.empty(self.L.lpmp_plan_n_updated(self.h, d), np.int32)
        _chk(self.L.lpmp_plan_get_update_order(self.h, d, out.ctypes.data))
        return out

    def omega(self, d: int, mode: int):
        off = np.empty(self.L.lpmp_plan_n_updated(self.h, d) + 1, np.int64)
        data = np.empty(self.L.lpmp_plan_omega_nnz(self.h, d), np.float64)
        _chk(self.L.lpmp_plan_get_omega(self.h, d, mode, off.ctypes.data, data.ctypes.data))
        return off, data

    def mask(self, d: int, mode: int):
        off = np.empty(self.L.lpmp_plan_n_updated(self.h, d) + 1, np.int64)
        data = np.empty(self.L.lpmp_plan_mask_nnz(self.h, d), np.uint8)
        _chk(self.L.lpmp_plan_get_mask(self.h, d, mode, off.ctypes.data, data.ctypes.data))
        return off, data

    def msg_lists(self, n_messages: int):
        off = np.empty(self.n_factors + 1, np.int64)
        ent = np.empty(2 * n_messages, np.int64)
        _chk(self.L.lpmp_plan_get_msg_lists(self.h, off.ctypes.data, ent.ctypes.data))
        return off, ent

    def anisotropic_weights(self, factors):
        factors = np.ascontiguousarray(factors, np.int32)
        nr, a, b = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(self.L.lpmp_plan_anisotropic_weights(self.h, factors.shape[0], factors.ctypes.data, C.addressof(nr),
                                                  C.addressof(a), C.addressof(b), None, None, None, None))
        om_off, mk_off = np.empty(nr.value + 1, np.int64), np.empty(nr.value + 1, np.int64)
        om, mk = np.empty(a.value, np.float64), np.empty(b.value, np.uint8)
        _chk(self.L.lpmp_plan_anisotropic_weights(self.h, factors.shape[0], factors.ctypes.data, None, None, None,
                                                  om_off.ctypes.data, om.ctypes.data, mk_off.ctypes.data, mk.ctypes.data))
        return om_off, om, mk_off, mk

    def update_levels(self, d: int, mode: int) -> np.ndarray:
        out = np.empty(self.L.lpmp_plan_n_updated(self.h, d), np.int32)
        _chk(self.L.lpmp_plan_get_update_levels(self.h, d, mode, out.ctypes.data))
        return out

    def schedule_info(self, d: int, mode: int) -> dict:
        v = [C.c_int64() for _ in range(5)]
        _chk(self.L.lpmp_plan_schedule_info(self.h, d, mode, *[C.addressof(x) for x in v]))
        return dict(zip(("n_levels", "n_launches", "n_receives", "n_sends", "algorithmic_bytes"), [x.value for x in v]))


def _custom_schedule_info(self, factors, om_off, om, mk_off, mk, fuse: bool = False) -> dict:
    """summary of the schedule an iterator-range pass (factor list, weight rows, receive-mask rows) compiles to"""
    factors = np.ascontiguousarray(factors, np.int32)
    om_off = np.ascontiguousarray(om_off, np.int64); om = np.ascontiguousarray(om, np.float64)
    mk_off = np.ascontiguousarray(mk_off, np.int64); mk = np.ascontiguousarray(mk, np.uint8)
    v = [C.c_int64() for _ in range(5)]
    _chk(self.L.lpmp_plan_custom_schedule_info(self.h, factors.shape[0], factors.ctypes.data, om_off.ctypes.data, om.ctypes.data,
                                               mk_off.ctypes.data, mk.ctypes.data, int(bool(fuse)), *[C.addressof(x) for x in v]))
    return dict(zip(("n_levels", "n_launches", "n_receives", "n_sends", "algorithmic_bytes"), [x.value for x in v]))


Plan.custom_schedule_info = _custom_schedule_info


def _schedule_classes(self, d: int, mode: int) -> dict:
    """updated factors of the sweep per device kernel class (only the classes that occur)"""
    out = np.zeros(N_KCLASS, np.int64)
    _chk(self.L.lpmp_plan_schedule_classes(self.h, d, mode, out.ctypes.data))
    return {KCLASS_NAMES[c]: int(out[c]) for c in range(N_KCLASS) if out[c]}


Plan.schedule_classes = _schedule_classes


def _pass_info(self, mode: int) -> dict:
    v = [C.c_int64() for _ in range(5)]
    _chk(self.L.lpmp_plan_pass_schedule_info(self.h, mode, *[C.addressof(x) for x in v]))
    return dict(zip(("n_levels", "n_launches", "n_receives", "n_sends", "algorithmic_bytes"), [x.value for x in v]))


Plan.pass_schedule_info = _pass_info


def _pass_rotates(self, mode: int) -> bool:
    """does lpmp_compute_pass(n >= 2) join consecutive passes at their seam for this mode?"""
    r = self.L.lpmp_plan_pass_rotates(self.h, mode)
    if r < 0:
        _chk(r)
    return bool(r)


Plan.pass_rotates = _pass_rotates


def _chain_info(self, d: int, mode: int) -> dict:
    """how the chain executor runs a deep sweep (d = 0 / 1, or -1: the fused pass): persistent launches, tickets,
    dependencies, launches that stay plain (all 0: ordinary launches / graph replay)"""
    v = [C.c_int64() for _ in range(4)]
    _chk(self.L.lpmp_plan_chain_info(self.h, d, mode, *[C.addressof(x) for x in v]))
    w = [C.c_int64() for _ in range(2)]
    _chk(self.L.lpmp_plan_mailbox_info(self.h, d, mode, *[C.addressof(x) for x in w]))
    return dict(zip(("n_chains", "n_tickets", "n_dependencies", "n_plain_launches", "mailbox_rows", "mailbox_receives"), [x.value for x in v + w]))


Plan.chain_info = _chain_info


def _partitions(self):
    """LP::construct_factor_partition (reference LP_MP.h:1717-1822): the components of the put_in_same_partition
    graph as arrays of (updated) factor indices"""
    n = C.c_int64()
    _chk(self.L.lpmp_plan_get_partitions(self.h, C.addressof(n), None, None))
    off = np.empty(n.value + 1, np.int64)
    f = np.empty(max(1, self.L.lpmp_plan_n_updated(self.h, 0)), np.int32)
    _chk(self.L.lpmp_plan_get_partitions(self.h, C.addressof(n), off.ctypes.data, f.ctypes.data))
    return [f[off[i]:off[i + 1]].copy() for i in range(n.value)]


Plan.partitions = _partitions


def _suggest_order(self, seed: int = 0):
    """(rank_of_factor[n_factors], number of colours): an order of all factors with the updated ones colour by colour
    (include/lpmp_engine.h, lpmp_plan_suggest_order); model.with_factor_order(rank) applies it"""
    rank = np.empty(self.n_factors, np.int32)
    k = C.c_int32()
    _chk(self.L.lpmp_plan_suggest_order(self.h, C.c_uint64(seed), rank.ctypes.data, C.addressof(k)))
    return rank, k.value


Plan.suggest_order = _suggest_order


def graph_colour_major_order(n: int, edge_i, edge_j, seed: int = 0):
    """(rank[n] int64, number of colours): colour-major variable order of a pairwise graph on the planner's threads
    (lpmp_graph_colour_major_order; the numpy statement of the same algorithm is ordering.colour_major_order_numpy)"""
    ei = np.ascontiguousarray(edge_i, np.int64); ej = np.ascontiguousarray(edge_j, np.int64)
    assert ei.shape == ej.shape
    rank = np.empty(int(n), np.int64)
    k = C.c_int32()
    _chk(lib().lpmp_graph_colour_major_order(int(n), ei.shape[0], ei.ctypes.data, ej.ctypes.data, C.c_uint64(seed), rank.ctypes.data, C.addressof(k)))
    return rank, k.value


def graph_refine_partition(n: int, edge_i, edge_j, part, world: int, rounds: int = 30, imbalance: float = 0.03, seed: int = 0) -> np.ndarray:
    """balanced KL refinement of a k-way partition (lpmp_graph_refine_partition; numpy statement: multi_gpu.refine_partition)"""
    ei = np.ascontiguousarray(edge_i, np.int64); ej = np.ascontiguousarray(edge_j, np.int64)
    out = np.array(part, np.int64, copy=True)
    assert out.shape[0] == int(n)
    _chk(lib().lpmp_graph_refine_partition(int(n), ei.shape[0], ei.ctypes.data, ej.ctypes.data, int(world), int(rounds), float(imbalance), C.c_uint64(seed), out.ctypes.data))
    return out


class Engine:
    """Device engine. ``const_dev`` / ``dual_dev``: optional device pointers (ints) of caller-owned HBM
    buffers holding the packed pairwise tables / duals (zero-copy; the caller keeps them alive)."""

    def __init__(self, device: int = 0):
        self.L = lib()
        h = C.c_void_p()
        _chk(self.L.lpmp_create(int(device), C.addressof(h)))
        self.h = h.value
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.lpmp_destroy(self.h)
            self.h = None
            self._keep = None          # the caller-owned device buffers the engine borrowed may go now

    def __del__(self):
        self.close()

    def set_stream(self, stream_ptr: int):
        _chk(self.L.lpmp_set_stream(self.h, C.c_void_p(stream_ptr)))

    def upload(self, model: FlatModel, const_dev: Optional[int] = None, dual_dev: Optional[int] = None, keep=None, rows_layout: Optional[bool] = None):
        """``rows_layout``: dense pairwise factors as [table | m1 | m2] rows of an engine-private buffer (lpmp_set_rows_layout);
        None: whatever LPMP_ROWS_LAYOUT says (default off)"""
        if rows_layout is not None:
            _chk(self.L.lpmp_set_rows_layout(self.h, 1 if rows_layout else 0))
        cs = model.c_struct()
        if const_dev is not None:
            cs.const_data = const_dev
        if dual_dev is not None:
            cs.dual_data = dual_dev
        self._keep = keep
        self.model = model
        _chk(self.L.lpmp_upload_model(self.h, C.addressof(cs), MEM_DEVICE if const_dev is not None else MEM_HOST,
                                      MEM_DEVICE if dual_dev is not None else MEM_HOST))

    def lower_bound_recomputed(self) -> int:
        """per-factor bounds the last lower_bound() had to recompute (the rest were tracked by the sweep kernels)"""
        return int(self.L.lpmp_lower_bound_recomputed(self.h))

    @property
    def rows_layout(self) -> bool:
        return bool(self.L.lpmp_rows_layout(self.h))

    @property
    def plan(self) -> Plan:
        return Plan(_handle=self.L.lpmp_engine_plan_mut(self.h), _owner=self)

    def set_reparametrization(self, mode: int):
        _chk(self.L.lpmp_set_reparametrization(self.h, int(mode)))

    def set_reparametrization_type(self, rtype: int):
        """reference --reparametrizationType: 0 shared, 1 residual, 2 partition, 3 overlapping_partition, 4 adaptive"""
        _chk(self.L.lpmp_set_reparametrization_type(self.h, int(rtype)))

    def set_inner_iterations(self, n: int):
        """reference --innerIteration: passes per partition in the partition sweeps (default 5)"""
        _chk(self.L.lpmp_set_inner_iterations(self.h, int(n)))

    def compute_pass(self, n: int = 1):
        _chk(self.L.lpmp_compute_pass(self.h, int(n)))

    def set_speculation(self, max_passes_ahead: int):
        """let compute_pass(1) run up to that many passes ahead of the caller (include/lpmp_engine.h); 0 = off"""
        _chk(self.L.lpmp_set_speculation(self.h, int(max_passes_ahead)))

    def speculation_stats(self) -> dict:
        v = [C.c_int64() for _ in range(4)]
        _chk(self.L.lpmp_speculation_stats(self.h, *[C.addressof(x) for x in v]))
        return dict(zip(("batches", "passes_launched", "passes_used", "rollbacks"), [x.value for x in v]))

    def chain_cache_bytes(self) -> int:
        return self.L.lpmp_chain_cache_bytes(self.h)

    def set_persistent_launches(self, on: bool):
        """chain executor / joined passes as persistent launches (include/lpmp_engine.h): off for an engine whose device is shared
        with other processes — every schedule then runs launch by launch, same results"""
        _chk(self.L.lpmp_set_persistent_launches(self.h, 1 if on else 0))

    @property
    def persistent_launches(self) -> bool:
        return bool(self.L.lpmp_persistent_launches(self.h))

    def prepare_passes(self, n: int):
        """build ahead of time what compute_pass(n) needs that depends on n (outside of a timed region)"""
        _chk(self.L.lpmp_prepare_passes(self.h, int(n)))

    def forward_pass(self):
        _chk(self.L.lpmp_compute_forward_pass(self.h))

    def backward_pass(self):
        _chk(self.L.lpmp_compute_backward_pass(self.h))

    def compute_pass_custom(self, factors, om_off, om, mk_off, mk):
        factors = np.ascontiguousarray(factors, np.int32)
        om_off = np.ascontiguousarray(om_off, np.int64)
        om = np.ascontiguousarray(om, np.float64)
        mk_off = np.ascontiguousarray(mk_off, np.int64)
        mk = np.ascontiguousarray(mk, np.uint8)
        _chk(self.L.lpmp_compute_pass_custom(self.h, factors.shape[0], factors.ctypes.data, om_off.ctypes.data,
                                             om.ctypes.data, mk_off.ctypes.data, mk.ctypes.data))

    def schedule_create(self, factors, om_off, om, mk_off, mk, fuse: bool = False) -> int:
        """Prepare an iterator-range pass (reference LP_MP.h:981-1005) for repeated replay.  ``fuse``: the list
        concatenates several sweeps; back-to-back updates of one factor are folded into one record."""
        factors = np.ascontiguousarray(factors, np.int32)
        om_off = np.ascontiguousarray(om_off, np.int64)
        om = np.ascontiguousarray(om, np.float64)
        mk_off = np.ascontiguousarray(mk_off, np.int64)
        mk = np.ascontiguousarray(mk, np.uint8)
        sid = C.c_int()
        _chk(self.L.lpmp_schedule_create_fused(self.h, factors.shape[0], factors.ctypes.data, om_off.ctypes.data,
                                               om.ctypes.data, mk_off.ctypes.data, mk.ctypes.data, 1 if fuse else 0,
                                               C.addressof(sid)))
        return sid.value

    def schedule_run(self, sid: int):
        _chk(self.L.lpmp_schedule_run(self.h, int(sid)))

    def schedule_info(self, sid: int) -> dict:
        v = [C.c_int64() for _ in range(5)]
        _chk(self.L.lpmp_schedule_info(self.h, int(sid), *[C.addressof(x) for x in v]))
        return dict(zip(("n_levels", "n_launches", "n_receives", "n_sends", "algorithmic_bytes"), [x.value for x in v]))

    def schedule_destroy(self, sid: int):
        _chk(self.L.lpmp_schedule_destroy(self.h, int(sid)))

    def lower_bound(self) -> float:
        out = C.c_double()
        _chk(self.L.lpmp_lower_bound(self.h, C.addressof(out)))
        return out.value

    # ---- primal rounding inside the sweep (reference LP_MP.h:914-940, 1067-1082, 1521-1536) ----
    def forward_pass_and_primal(self, iteration: int):
        _chk(self.L.lpmp_compute_forward_pass_and_primal(self.h, int(iteration)))

    def backward_pass_and_primal(self, iteration: int):
        _chk(self.L.lpmp_compute_backward_pass_and_primal(self.h, int(iteration)))

    def compute_pass_and_primal(self, iteration: int):
        _chk(self.L.lpmp_compute_pass_and_primal(self.h, int(iteration)))

    def check_primal_consistency(self) -> bool:
        out = C.c_int()
        _chk(self.L.lpmp_check_primal_consistency(self.h, C.addressof(out)))
        return bool(out.value)

    def evaluate_primal(self) -> float:
        out = C.c_double()
        _chk(self.L.lpmp_evaluate_primal(self.h, C.addressof(out)))
        return out.value

    def download_primal(self) -> np.ndarray:
        """[n_factors, 2] primal_ members: vector (label, 0), pairwise (x0, x1); unset = the dimension"""
        out = np.empty((self.model.n_factors, 2), np.int32)
        _chk(self.L.lpmp_download_primal(self.h, out.ctypes.data))
        return out

    def upload_primal(self, primal: np.ndarray):
        primal = np.ascontiguousarray(primal, np.int32)
        assert primal.shape == (self.model.n_factors, 2)
        _chk(self.L.lpmp_upload_primal(self.h, primal.ctypes.data))

    def factor_lower_bounds(self) -> np.ndarray:
        out = np.empty(self.model.n_factors, np.float64)
        _chk(self.L.lpmp_factor_lower_bounds(self.h, out.ctypes.data))
        return out

    def invalidate_lower_bounds(self):
        """duals were changed outside the engine (borrowed buffer): recompute every factor's bound next time"""
        _chk(self.L.lpmp_invalidate_lower_bounds(self.h))

    def synchronize(self):
        _chk(self.L.lpmp_synchronize(self.h))

    def download_duals(self) -> np.ndarray:
        out = np.empty(self.L.lpmp_dual_size(self.h), np.float64)
        _chk(self.L.lpmp_download_duals(self.h, out.ctypes.data))
        return out

    def upload_duals(self, d: np.ndarray):
        d = np.ascontiguousarray(d, np.float64)
        assert d.shape[0] == self.L.lpmp_dual_size(self.h)
        _chk(self.L.lpmp_upload_duals(self.h, d.ctypes.data))

    def device_duals_ptr(self) -> int:
        return self.L.lpmp_device_duals(self.h)

    # ---- boundary step of the partitioned sweep (include/lpmp_engine.h, csrc/boundary.hip) ----
    def boundary_create(self, out_dual_off, out_len, in_dual_off, in_len, in_omega, in_order) -> int:
        a = [np.ascontiguousarray(out_dual_off, np.int64), np.ascontiguousarray(out_len, np.int32),
             np.ascontiguousarray(in_dual_off, np.int64), np.ascontiguousarray(in_len, np.int32),
             np.ascontiguousarray(in_omega, np.float64), np.ascontiguousarray(in_order, np.int64)]
        h = C.c_void_p()
        _chk(self.L.lpmp_boundary_create(self.h, a[0].shape[0], a[0].ctypes.data, a[1].ctypes.data, a[2].shape[0],
                                         a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data, a[5].ctypes.data, C.addressof(h)))
        return h.value

    def boundary_destroy(self, b: int):
        self.L.lpmp_boundary_destroy(b)

    def boundary_sizes(self, b: int):
        return self.L.lpmp_boundary_out_doubles(b), self.L.lpmp_boundary_in_doubles(b)

    def boundary_pack(self, b: int, send_ptr: int):
        _chk(self.L.lpmp_boundary_pack(self.h, b, C.c_void_p(send_ptr)))

    def boundary_reply(self, b: int, recv_ptr: int, reply_ptr: int):
        _chk(self.L.lpmp_boundary_reply(self.h, b, C.c_void_p(recv_ptr), C.c_void_p(reply_ptr)))

    def boundary_fold(self, b: int, back_ptr: int):
        _chk(self.L.lpmp_boundary_fold(self.h, b, C.c_void_p(back_ptr)))

    # ---- halos of the lock-step sweep (csrc/boundary.hip): vectors as (packed dual offset, length), exchange order ----
    def halo_create(self, out_dual_off, out_len, in_dual_off, in_len) -> int:
        a = [np.ascontiguousarray(out_dual_off, np.int64), np.ascontiguousarray(out_len, np.int32),
             np.ascontiguousarray(in_dual_off, np.int64), np.ascontiguousarray(in_len, np.int32)]
        h = C.c_void_p()
        _chk(self.L.lpmp_halo_create(self.h, a[0].shape[0], a[0].ctypes.data, a[1].ctypes.data, a[2].shape[0], a[2].ctypes.data,
                                     a[3].ctypes.data, C.addressof(h)))
        return h.value

    def halo_destroy(self, h: int):
        self.L.lpmp_halo_destroy(h)

    def halo_sizes(self, h: int):
        return self.L.lpmp_halo_out_doubles(h), self.L.lpmp_halo_in_doubles(h)

    def halo_pack(self, h: int, send_ptr: int):
        _chk(self.L.lpmp_halo_pack(self.h, h, C.c_void_p(send_ptr)))

    def halo_unpack(self, h: int, recv_ptr: int):
        _chk(self.L.lpmp_halo_unpack(self.h, h, C.c_void_p(recv_ptr)))

    def enable_kernel_timing(self, on: bool):
        _chk(self.L.lpmp_enable_kernel_timing(self.h, 1 if on else 0))

    def reset_kernel_timing(self):
        _chk(self.L.lpmp_reset_kernel_timing(self.h))

    def kernel_timing(self) -> dict:
        ms = np.zeros(N_KCLASS, np.float64)
        arrs = [np.zeros(N_KCLASS, np.int64) for _ in range(4)]
        _chk(self.L.lpmp_get_kernel_timing(self.h, N_KCLASS, ms.ctypes.data, *[a.ctypes.data for a in arrs]))
        chain = np.zeros(N_KCLASS, np.int64)
        _chk(self.L.lpmp_get_chain_launches(self.h, N_KCLASS, chain.ctypes.data))
        out = {}
        for c in range(N_KCLASS):
            if arrs[0][c] > 0:
                name = KERNEL_NAMES[c]
                if chain[c] > 0:             # joined passes as persistent launches: plain table loads, agent-scope dual accesses
                    name = name.replace("sweep_", "chain_")
                    if not name.endswith(">") and "dense" in name:
                        name += ", false>"
                    elif not name.endswith(">"):
                        name += ">"
                    out[KCLASS_NAMES[c]] = dict(kernel=name, ms=float(ms[c]), launches=int(arrs[0][c]), chain_launches=int(chain[c]),
                                                factors=int(arrs[1][c]), receives=int(arrs[2][c]), bytes=int(arrs[3][c]))
                    continue
                if not name.endswith(">") and "<" in name:       # exact dense kernels: plain / non-temporal form
                    name += ", true>" if self.L.lpmp_streaming_access(self.h) == 1 else ", false>"
                elif name == "sweep_dense_big_kernel":
                    name += "<true>" if self.L.lpmp_streaming_access(self.h) == 1 else "<false>"
                out[KCLASS_NAMES[c]] = dict(kernel=name, ms=float(ms[c]), launches=int(arrs[0][c]),
                                            factors=int(arrs[1][c]), receives=int(arrs[2][c]), bytes=int(arrs[3][c]))
        return out


def device_identity(device: int = 0) -> str:
    """"pci=... uuid=..." of a HIP device ordinal: equal strings = one physical GPU, whatever each process calls it"""
    buf = C.create_string_buffer(128)
    _chk(lib().lpmp_device_identity(int(device), buf, 128))
    return buf.value.decode()


def synth_fill_blocks(device_ptr: int, n_blocks: int, block_len: int, seed: int, first_dev_ptr: int, stream_ptr: int = 0):
    """block b of ``block_len`` values continues the global u01 stream at first[b] (int64 device array)"""
    _chk(lib().lpmp_synth_fill_blocks(C.c_void_p(device_ptr), n_blocks, block_len, C.c_uint64(seed), C.c_void_p(first_dev_ptr), C.c_void_p(stream_ptr)))


def synth_fill(device_ptr: int, n: int, seed: int, first: int = 0, stream_ptr: int = 0):
    _chk(lib().lpmp_synth_fill(C.c_void_p(device_ptr), n, C.c_uint64(seed), C.c_uint64(first), C.c_void_p(stream_ptr)))
