"""The reference's other --reparametrizationType values on the device, against the oracle bit for bit:
partition / overlapping_partition sweeps (LP_MP.h:1717-2051), adaptive sends (factors_messages.hxx:2263-2268, 2860-2926)
with the improvement op, and the static batch-send dispatch of CallSendMessages (factors_messages.hxx:2709-2726)."""
import numpy as np
import pytest

from lp_mp_amd import engine as E
from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle

pytestmark = pytest.mark.gpu
MODES = (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM)


def grid(H, W, L, pairwise="dense", order="row_major", seed=1, flags=0, blocks=0, sched=M.SCHED_LEFT):
    """grid MRF; ``blocks`` > 0: put_in_same_partition for every edge inside one of ``blocks`` column bands"""
    var = S.grid_variable_order(H, W, order).reshape(-1)
    a, bb = S.grid_edges(H, W)
    i, j = np.minimum(var[a], var[bb]), np.maximum(var[a], var[bb])
    mts = [M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 0, flags), M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 1, flags)]
    b = M.ModelBuilder(2, mts)
    u = b.add_vector_factors(0, S.u01(H * W * L, seed).reshape(-1, L))
    if pairwise == "dense":
        p = b.add_dense_pairwise(1, S.u01(len(a) * L * L, seed + 1).reshape(-1, L, L))
    else:
        p = b.add_potts_pairwise(1, L, S.u01(len(a), seed + 1) - 0.3)
    b.add_interleaved_messages(np.tile(np.array([0, 1], np.int32), len(a)), np.stack([u[i], u[j]], 1).reshape(-1), np.repeat(p, 2))
    b.add_relations(np.stack([u[i], p], 1).reshape(-1), np.stack([p, u[j]], 1).reshape(-1))
    if blocks:
        band = (np.arange(H * W) % W) * blocks // W
        for k in range(len(a)):
            if band[a[k]] == band[bb[k]]:
                b.put_in_same_partition(u[var[a[k]]], u[var[bb[k]]])
                if k % 3 == 0:
                    b.put_in_same_partition(p[k], u[var[a[k]]])      # non-updated factors may be named too
    return b.finish()


def _run_both(m, rtype, mode, passes=(1, 2), inner=None, monotone=True):
    o = Oracle(m)
    e = E.Engine(0)
    try:
        if inner is not None:
            o.set_inner_iterations(inner); e.set_inner_iterations(inner)
        o.set_reparametrization_type(rtype); o.set_reparametrization(mode)
        e.upload(m); e.set_reparametrization_type(rtype); e.set_reparametrization(mode)
        lb = e.lower_bound()
        for n in passes:
            o.ComputePass(n); e.compute_pass(n)
            assert np.array_equal(e.download_duals(), o.duals()), (rtype, mode, n)
            lb2, lbo = e.lower_bound(), o.LowerBound()
            assert abs(lb2 - lbo) <= 1e-9 * max(1.0, abs(lbo))
            assert not monotone or lb2 >= lb - 1e-9 * max(1.0, abs(lb))
            lb = lb2
        return o, e
    except Exception:
        e.close()
        raise


# ---- partition sweeps -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rtype", [M.RTYPE_PARTITION, M.RTYPE_OVERLAPPING_PARTITION])
@pytest.mark.parametrize("inner", [1, 3, 5])
def test_partition_sweeps_three_components(rtype, inner):
    for pw, L, order in (("dense", 8, "row_major"), ("potts", 4, "colour_major"), ("dense", 21, "row_major")):
        m = grid(6, 9, L, pw, order, seed=L, blocks=3)
        o, e = _run_both(m, rtype, M.REPAM_ANISOTROPIC, inner=inner)
        po, pe = o.partitions(), e.plan.partitions()
        assert len(po) == 3 and all(np.array_equal(x, y) for x, y in zip(po, pe))
        e.close()


@pytest.mark.parametrize("rtype", [M.RTYPE_PARTITION, M.RTYPE_OVERLAPPING_PARTITION])
def test_partition_sweeps_degenerate_partitions(rtype):
    # no put_in_same_partition call at all: every updated factor is a partition of its own; and ONE partition
    m = grid(4, 5, 4, seed=3)
    o, e = _run_both(m, rtype, M.REPAM_DAMPED_UNIFORM, inner=2)
    assert len(e.plan.partitions()) == 20
    e.close()
    m = grid(4, 5, 4, seed=3, blocks=1)
    o, e = _run_both(m, rtype, M.REPAM_ANISOTROPIC, inner=2)
    assert len(e.plan.partitions()) == 1
    e.close()


@pytest.mark.parametrize("seed", range(12))
def test_partition_sweeps_random_models(seed):
    """random factor graphs of every device kind and schedule with a random partition graph"""
    from tests.test_fuzz_gpu import random_model
    rng = np.random.default_rng(21000 + seed)
    m = random_model(rng)
    n = m.n_factors
    k = int(rng.integers(0, 2 * n))
    m.part_pairs = rng.integers(0, n, size=(k, 2)).astype(np.int32)
    for rtype in (M.RTYPE_PARTITION, M.RTYPE_OVERLAPPING_PARTITION):
        # (no monotonicity claim here: the generator draws implicit-origin flags at random, and a labeling message between
        # a left factor without origin and a right factor with unmatched labelings is not an ascent step in the reference either)
        o, e = _run_both(m, rtype, MODES[seed % 4], inner=int(rng.integers(1, 4)), monotone=False)
        po, pe = o.partitions(), e.plan.partitions()
        assert len(po) == len(pe) and all(np.array_equal(x, y) for x, y in zip(po, pe))
        # switching the type on a live engine
        e.set_reparametrization_type(M.RTYPE_SHARED); o.set_reparametrization_type(M.RTYPE_SHARED)
        e.compute_pass(1); o.ComputePass(1)
        assert np.array_equal(e.download_duals(), o.duals())
        e.close()


# ---- adaptive sends ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pairwise,L", [("dense", 5), ("dense", 8), ("dense", 32), ("potts", 4), ("potts", 16)])
@pytest.mark.parametrize("order", ["row_major", "colour_major"])
def test_adaptive_sends_with_improvement_op(pairwise, L, order):
    m = grid(6, 5, L, pairwise, order, seed=L, flags=M.MF_IMPROVEMENT)
    for mode in MODES:
        o, e = _run_both(m, M.RTYPE_ADAPTIVE, mode, passes=(1, 2, 1))
        assert set(e.plan.schedule_classes(M.FORWARD, mode)) <= {"generic", "small"}      # the adaptive rule lives in the generic kernels
        # anisotropic weights send along messages the factor has NOT just received through: positive improvements.
        # (uniform modes receive through every message first; sending straight back then improves nothing — exactly 0 —
        # and the rescaled weights stay 0, in the reference too)
        # (Potts tables with a positive coupling and no messages yet: min_ab (diff [a != b] + theta[a]) = min theta, again 0)
        if mode in (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2) and pairwise == "dense":
            assert o.counters()[1] > 0
        e.close()


def test_adaptive_sends_without_improvement_op_send_nothing():
    # the reference's release build: the container returns improvement 0, the rescaled weights stay 0
    m = grid(5, 6, 8, seed=2)
    o, e = _run_both(m, M.RTYPE_ADAPTIVE, M.REPAM_ANISOTROPIC)
    assert o.counters()[1] == 0
    e.close()


def test_adaptive_sends_other_kinds_and_roles():
    # updated pairwise factors (right / full schedules: the min-marginal is what is sent), labeling lists, tiny factors
    for sched in (M.SCHED_RIGHT, M.SCHED_FULL):
        m = grid(5, 4, 6, seed=sched, flags=M.MF_IMPROVEMENT, sched=sched)
        for mode in (M.REPAM_ANISOTROPIC, M.REPAM_UNIFORM):
            o, e = _run_both(m, M.RTYPE_ADAPTIVE, mode)
            e.close()
    b = S.multicut_builder()
    for t in b.mtypes:
        t.flags = M.MF_IMPROVEMENT
    rng = np.random.default_rng(4)
    e_ids = b.add_vector_factors(0, rng.uniform(-1, 1, (12, 1)), implicit_origin=True)
    for _ in range(9):
        t = b.add_vector_factors(1, rng.uniform(-0.3, 0.3, (1, 4)), implicit_origin=True)[0]
        for k, ei in enumerate(rng.choice(12, 3, replace=False)):
            b.add_messages(k, e_ids[ei], t); b.add_relations(e_ids[ei], t)
    m = b.finish()
    for mode in MODES:
        o, e = _run_both(m, M.RTYPE_ADAPTIVE, mode, passes=(1, 1, 2))
        assert "small" in e.plan.schedule_classes(M.BACKWARD, mode) or "small" in e.plan.schedule_classes(M.FORWARD, mode)
        e.close()


def test_adaptive_and_residual_refusals():
    from tests.test_plan_host import _full_schedule_model
    e = E.Engine(0)
    try:
        e.upload(_full_schedule_model())                    # factors with messages they do not send through
        with pytest.raises(E.EngineError) as ei:
            e.set_reparametrization_type(M.RTYPE_ADAPTIVE)
        assert ei.value.code == -2
        o = Oracle(_full_schedule_model())
        with pytest.raises(RuntimeError):
            o.set_reparametrization_type(M.RTYPE_ADAPTIVE)
        m = grid(3, 3, 4, flags=M.MF_BATCH_TO_RIGHT)
        e.upload(m)
        for rt in (M.RTYPE_RESIDUAL, M.RTYPE_ADAPTIVE):
            with pytest.raises(E.EngineError):
                e.set_reparametrization_type(rt)
        e.set_reparametrization_type(M.RTYPE_ADAPTIVE if False else M.RTYPE_SHARED)
        # the rule may be chosen before the model arrives (the reference parses it in Begin): refused at upload then
        e2 = E.Engine(0)
        e2.set_reparametrization_type(M.RTYPE_RESIDUAL)
        with pytest.raises(E.EngineError):
            e2.upload(m)
        e2.close()
    finally:
        e.close()


# ---- batch sends ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(10))
def test_batch_send_dispatch(seed):
    """ops with a static SendMessagesToRight / ToLeft: more than one active message of a type -> one call with the
    sum of the weights; iterator-range passes with unequal random weights make the difference to plain sends visible"""
    from tests.test_fuzz_gpu import random_rows
    rng = np.random.default_rng(23000 + seed)
    L = int(rng.choice([3, 4, 8, 16]))
    sched = int(rng.choice([M.SCHED_LEFT, M.SCHED_FULL]))
    flags = int(rng.choice([M.MF_BATCH_TO_RIGHT, M.MF_BATCH_TO_RIGHT | M.MF_BATCH_TO_LEFT, M.MF_BATCH_TO_LEFT]))
    m = grid(5, 6, L, "dense" if seed % 2 else "potts", "row_major", seed=seed, flags=flags, sched=sched)
    plain = grid(5, 6, L, "dense" if seed % 2 else "potts", "row_major", seed=seed, flags=0, sched=sched)
    e = E.Engine(0)
    try:
        differs = False
        for mode in MODES:
            o = Oracle(m); o.set_reparametrization(mode)
            op = Oracle(plain); op.set_reparametrization(mode)
            e.upload(m); e.set_reparametrization(mode)
            e.compute_pass(2); o.ComputePass(2); op.ComputePass(2)
            assert np.array_equal(e.download_duals(), o.duals())
            for _ in range(3):
                rows = random_rows(rng, None, o, m)
                e.compute_pass_custom(*rows); o.compute_pass_custom(*rows); op.compute_pass_custom(*rows)
                assert np.array_equal(e.download_duals(), o.duals()), (seed, mode)
            differs = differs or not np.array_equal(o.duals(), op.duals())
            sid = e.schedule_create(*rows, fuse=False)
            e.schedule_run(sid); o.compute_pass_custom(*rows)
            assert np.array_equal(e.download_duals(), o.duals())
        assert differs == bool(flags & M.MF_BATCH_TO_RIGHT) or sched == M.SCHED_FULL   # the batch rule really changed something
    finally:
        e.close()
