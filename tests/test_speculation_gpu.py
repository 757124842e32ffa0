"""Passes that run ahead of the caller (lpmp_set_speculation, include/lpmp_engine.h): a reference-shaped caller asks for ONE
pass per iteration and the bound after each (Solver::Iterate / PostIterate, reference include/solver.hxx:273-284); the
engine may launch several passes as one persistent launch, hand out each pass's own bound, and roll back when the caller
does something else.  Whatever the caller does, every observable value must equal the engine without speculation BIT FOR
BIT (and the oracle's)."""
import numpy as np
import pytest

from lp_mp_amd import engine as E
from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle

pytestmark = pytest.mark.gpu


def _pair(m, depth, mode=M.REPAM_ANISOTROPIC):
    a, b = E.Engine(0), E.Engine(0)
    for e in (a, b):
        e.upload(m); e.set_reparametrization(mode)
        e.lower_bound()        # from here on both engines sum TRACKED per-factor bounds (the first call scans every table)
    b.set_speculation(depth)
    return a, b


@pytest.mark.parametrize("pairwise,L,H,W", [("dense", 32, 40, 36), ("dense", 8, 60, 70), ("dense", 21, 30, 31)])
def test_solver_loop_gets_joined_passes_and_the_same_bound_history(pairwise, L, H, W, monkeypatch):
    """the reference's Solve loop — set_reparametrization(same mode), ComputePass(iter), LowerBound() every iteration — with
    speculation: the bounds are those of single calls, the passes ran as a few multi-pass launches"""
    monkeypatch.setenv("LPMP_ROT_BANDS", "6")          # the joined chain on a small model (as test_joined_passes_... does)
    m = S.grid_model(H, W, L, pairwise=pairwise, order="colour_major", seed=L)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    a, b = _pair(m, 8)
    try:
        hist_a, hist_b, hist_o = [], [], []
        for it in range(37):
            for e, h in ((a, hist_a), (b, hist_b)):
                e.set_reparametrization(M.REPAM_ANISOTROPIC)
                e.compute_pass(1)
                h.append(e.lower_bound())
            o.ComputePass(1); hist_o.append(o.LowerBound())
        assert hist_a == hist_b                                   # bit for bit
        assert np.allclose(hist_b, hist_o, rtol=1e-9, atol=0)
        st = b.speculation_stats()
        # 2 + 4 + 8 + 8 + 8 + 8 = 38 >= 37: six launches instead of 37, nothing rolled back inside the loop
        assert st["batches"] == 6 and st["passes_used"] == 37 and st["passes_launched"] == 38 and st["rollbacks"] == 0, st
        # the caller stops one pass short of the last batch: End() / the dual download must see the state after 37 passes
        assert np.array_equal(b.download_duals(), a.download_duals())
        assert np.array_equal(b.download_duals(), o.duals())
        assert b.speculation_stats()["rollbacks"] == 1
        assert b.lower_bound() == a.lower_bound()
        assert np.array_equal(b.factor_lower_bounds(), a.factor_lower_bounds())
    finally:
        a.close(); b.close()


def test_rounding_solver_pattern_is_learnt(monkeypatch):
    """MpRoundingSolver: four plain passes, then a rounding iteration in another weight mode (standard_visitor.hxx:172-185,
    solver.hxx:387-397).  The first interruption costs a rollback; from then on a batch is exactly one run long."""
    monkeypatch.setenv("LPMP_ROT_BANDS", "5")
    m = S.grid_model(36, 40, 16, order="colour_major", seed=3, compute_primal=True)
    a, b = _pair(m, 16)
    try:
        rec = {id(a): [], id(b): []}
        for it in range(30):
            for e in (a, b):
                if it % 5 == 4:
                    e.set_reparametrization(M.REPAM_DAMPED_UNIFORM)
                    e.forward_pass_and_primal(it); rec[id(e)].append(e.evaluate_primal())
                    e.backward_pass_and_primal(it); rec[id(e)].append(e.evaluate_primal())
                else:
                    e.set_reparametrization(M.REPAM_ANISOTROPIC)
                    e.compute_pass(1)
                rec[id(e)].append(e.lower_bound())
        assert rec[id(a)] == rec[id(b)]
        assert np.array_equal(a.download_duals(), b.download_duals()) and np.array_equal(a.download_primal(), b.download_primal())
        st = b.speculation_stats()
        # run 1: batches of 2 and 4 (two passes of the second unused: one rollback); runs 2 ... 6: one batch of 4 each
        assert st["rollbacks"] == 1 and st["batches"] == 2 + 5 and st["passes_used"] == 24, st
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_call_sequences_with_and_without_speculation(seed, monkeypatch):
    """any interleaving of passes, bounds, directional sweeps, mode changes, dual round trips, custom passes: every value the
    caller can see is the same with passes running ahead as without"""
    monkeypatch.setenv("LPMP_ROT_BANDS", "4")
    rng = np.random.default_rng(seed)
    L = int(rng.choice([4, 8, 16, 32]))
    m = S.grid_model(int(rng.integers(12, 40)), int(rng.integers(12, 40)), L, order="colour_major", seed=seed)
    a, b = _pair(m, int(rng.choice([2, 3, 8, 32])))
    try:
        mode = M.REPAM_ANISOTROPIC
        for step in range(60):
            r = rng.random()
            seen = []
            for e in (a, b):
                if r < 0.55:
                    e.compute_pass(1)
                elif r < 0.75:
                    seen.append(e.lower_bound())
                elif r < 0.80:
                    e.forward_pass() if step % 2 else e.backward_pass()
                elif r < 0.85:
                    e.compute_pass(3)
                elif r < 0.90:
                    e.set_reparametrization([M.REPAM_ANISOTROPIC, M.REPAM_UNIFORM, M.REPAM_ANISOTROPIC2][step % 3])
                elif r < 0.94:
                    d = e.download_duals(); seen.append(d); e.upload_duals(d * 0.5)
                elif r < 0.97:
                    seen.append(e.factor_lower_bounds())
                else:
                    e.synchronize(); seen.append(e.download_duals())
            # duals: bit for bit.  Bounds: a bound recomputed from scratch (after upload_duals) and a tracked one add the same
            # three numbers in another order — 1 ulp apart; with and without speculation that choice may fall differently
            for x, y in zip(seen[: len(seen) // 2], seen[len(seen) // 2:]):
                if np.ndim(x) == 0 or (0.85 <= r < 0.90) or (0.94 <= r < 0.97):
                    assert np.allclose(np.asarray(x), np.asarray(y), rtol=1e-12, atol=1e-12), (seed, step, r)
                else:
                    assert np.array_equal(np.asarray(x), np.asarray(y)), (seed, step, r)
        assert np.array_equal(a.download_duals(), b.download_duals()) and abs(a.lower_bound() - b.lower_bound()) <= 1e-12 * abs(a.lower_bound())
        assert b.speculation_stats()["batches"] > 0
    finally:
        a.close(); b.close()


def test_models_without_joined_passes_run_every_call_as_it_comes(monkeypatch):
    """Potts steps, row-major orders, cache-resident models: no multi-pass launch exists, speculation changes nothing"""
    for m in (S.grid_model(20, 24, 8, pairwise="potts", order="colour_major", seed=1), S.grid_model(16, 16, 16, order="row_major", seed=2),
              S.grid_model(24, 24, 32, order="colour_major", seed=3)):
        a, b = _pair(m, 8)
        try:
            for _ in range(7):
                a.compute_pass(1); b.compute_pass(1)
                assert a.lower_bound() == b.lower_bound()
            assert np.array_equal(a.download_duals(), b.download_duals())
            assert b.speculation_stats()["batches"] == 0
        finally:
            a.close(); b.close()
