"""Mailbox of the dense chain executor (plan.cpp: which message vectors travel as tagged granules and which dependencies
that covers; kernels.hip: mailbox_put / mailbox_take) against the oracle, bit for bit — with the mailbox, with every
hand-over through completion flags (LPMP_NO_MAILBOX=1), and against each other."""
import numpy as np
import pytest

from lp_mp_amd import engine as E
from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle

pytestmark = pytest.mark.gpu
LB_RTOL = 1e-5


def banded_model(n, L, offsets, seed=3, compute_primal=True):
    """variables 0 .. n-1 in index order, an edge (i, i + o) for every offset: every sweep is n levels deep, a variable
    receives from up to len(offsets) earlier and sends to as many later ones (several mailbox rows per record)"""
    ei = np.concatenate([np.arange(0, n - o) for o in offsets]); ej = np.concatenate([np.arange(0, n - o) + o for o in offsets])
    o = np.lexsort((ej, ei)); ei, ej = ei[o], ej[o]
    return S.mrf_model(n, L, ei, ej, S.u01(n * L, seed, 0), tables=S.u01(ei.shape[0] * L * L, seed, n * L), compute_primal=compute_primal)


MODELS = {
    "banded 8 labels, offsets 1 5 17": lambda: banded_model(600, 8, (1, 5, 17)),
    "banded 32 labels, offsets 1 2 3 4": lambda: banded_model(160, 32, (1, 2, 3, 4)),     # 4 + 4 messages: a full packet
    "banded 4 labels, offset 1 (a chain)": lambda: banded_model(900, 4, (1,)),
    "random graph, 8 labels (mailbox and flags mixed)": lambda: S.random_graph_model(3000, 3000, 8, seed=2, compute_primal=True),
    "row-major grid, 16 labels": lambda: S.grid_model(40, 30, 16, order="row_major", seed=5, compute_primal=True),
    "row-major Potts grid, 8 labels": lambda: S.grid_model(60, 50, 8, pairwise="potts", order="row_major", seed=6, compute_primal=True),
    "row-major Potts grid, 32 labels": lambda: S.grid_model(30, 40, 32, pairwise="potts", order="row_major", seed=7, compute_primal=True),
    "row-major grid, 21 labels (run-time dims)": lambda: S.grid_model(30, 26, 21, order="row_major", seed=8, compute_primal=True),
    "row-major Potts grid, 5 labels (run-time dims)": lambda: S.grid_model(44, 50, 5, pairwise="potts", order="row_major", seed=9, compute_primal=True),
    "banded 3 labels, offsets 1 4 (run-time dims)": lambda: banded_model(500, 3, (1, 4), seed=10),
    "random Potts graph, 4 labels (mailbox and flags mixed)": lambda: S.random_graph_model(3000, 3000, 4, seed=2, pairwise="potts", compute_primal=True),
}


@pytest.mark.parametrize("name", list(MODELS))
def test_mailbox_chains_equal_the_oracle(name, monkeypatch):
    m = MODELS[name]()
    monkeypatch.delenv("LPMP_NO_MAILBOX", raising=False)
    ci = E.Plan(m).chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert ci["n_chains"] == 1 and ci["mailbox_rows"] > 0 and ci["mailbox_receives"] >= ci["mailbox_rows"], ci
    monkeypatch.setenv("LPMP_NO_MAILBOX", "1")
    cf = E.Plan(m).chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert cf["mailbox_rows"] == 0 and cf["n_dependencies"] > ci["n_dependencies"], (ci, cf)
    o = Oracle(m)
    engines = []
    for env in (None, "1"):
        if env: monkeypatch.setenv("LPMP_NO_MAILBOX", env)
        else: monkeypatch.delenv("LPMP_NO_MAILBOX", raising=False)
        e = E.Engine(0); e.upload(m); engines.append(e)
    def same(what):
        for k, e in enumerate(engines):
            assert np.array_equal(e.download_duals(), o.duals()), (what, k)
            lb, lbo = e.lower_bound(), o.LowerBound()
            assert abs(lb - lbo) <= LB_RTOL * max(1.0, abs(lbo)), (what, k)
            # the tracked bounds (incl. the marks a mailbox send leaves to its reader) against a full recomputation
            tracked = e.lower_bound(); e.invalidate_lower_bounds()
            assert abs(e.lower_bound() - tracked) <= 1e-12 * max(1.0, abs(tracked)), (what, k)
    try:
        for mode in (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM):
            o.set_reparametrization(mode)
            for e in engines: e.set_reparametrization(mode)
            o.ComputeForwardPass(); o.ComputeBackwardPass()
            for e in engines: e.forward_pass(); e.backward_pass()
            same(("directional", mode))
            for n in (1, 3):
                o.ComputePass(n)
                for e in engines: e.compute_pass(n)
                same(("passes", mode, n))
        # the residual rule adds to a sent vector once more: the mailbox carries the final value
        o.set_reparametrization_type(1)
        for e in engines: e.set_reparametrization_type(1)
        o.ComputeForwardPass(); o.ComputePass(1)
        for e in engines: e.forward_pass(); e.compute_pass(1)
        same("residual")
        o.set_reparametrization_type(0)
        for e in engines: e.set_reparametrization_type(0)
        # rounding passes run in the same chains
        o.set_reparametrization(M.REPAM_ANISOTROPIC)
        for e in engines: e.set_reparametrization(M.REPAM_ANISOTROPIC)
        for it in (1, 2):
            o.ComputePassAndPrimal(it)
            for k, e in enumerate(engines):
                e.compute_pass_and_primal(it)
                assert np.array_equal(e.download_primal(), o.primal()), (it, k)
                c, co = e.evaluate_primal(), o.EvaluatePrimal()
                assert abs(c - co) <= 1e-9 * max(1.0, abs(co))
        same("after rounding")
    finally:
        for e in engines: e.close()


def test_mailbox_tags_survive_many_runs_and_new_duals():
    """a granule is valid for ONE launch (its tag is the launch's epoch): forty launches back to back on one schedule, new
    duals uploaded in between, the fused pass and the directional sweeps interleaved — always the flags-only result"""
    import os
    m = banded_model(400, 16, (1, 3, 11), seed=8, compute_primal=False)
    rng = np.random.default_rng(4)
    res = {}
    for env in (None, "1"):
        if env: os.environ["LPMP_NO_MAILBOX"] = env
        else: os.environ.pop("LPMP_NO_MAILBOX", None)
        try:
            e = E.Engine(0); e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
            d0 = e.download_duals()
            out = []
            r = np.random.default_rng(4)
            for k in range(40):
                if k % 7 == 3: e.upload_duals(d0 * r.uniform(0.5, 1.5))
                if k % 3 == 0: e.forward_pass()
                elif k % 3 == 1: e.backward_pass()
                else: e.compute_pass(2)
                out.append(e.download_duals().copy())
            res[env] = out
            e.close()
        finally:
            os.environ.pop("LPMP_NO_MAILBOX", None)
    for a, b in zip(res[None], res["1"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["row-major grid, 16 labels", "row-major Potts grid, 8 labels", "banded 32 labels, offsets 1 2 3 4"])
def test_mailbox_planned_schedule_on_the_launch_by_launch_paths(name, monkeypatch):
    """the mailbox plan edits the PACKET copies of the ops it covers (plan.hpp OP_MAILBOX: a send's peer_const, the bits of a
    receive's omega).  The same packets are read by the plain kernels whenever such a schedule runs launch by launch — per-launch
    kernel timing, LPMP_NO_CHAIN=1 (hipGraph replay), primal passes: those bodies must not read the edited fields.  Duals of the
    oracle on each of these paths."""
    m = MODELS[name]()
    monkeypatch.delenv("LPMP_NO_MAILBOX", raising=False)
    assert E.Plan(m).chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)["mailbox_rows"] > 0
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    timed = E.Engine(0); timed.upload(m); timed.set_reparametrization(M.REPAM_ANISOTROPIC)
    monkeypatch.setenv("LPMP_NO_CHAIN", "1")
    nochain = E.Engine(0); nochain.upload(m); nochain.set_reparametrization(M.REPAM_ANISOTROPIC)
    monkeypatch.delenv("LPMP_NO_CHAIN")
    try:
        timed.enable_kernel_timing(True)
        for n in (1, 2):
            o.ComputePass(n); timed.compute_pass(n); nochain.compute_pass(n)
            assert np.array_equal(timed.download_duals(), o.duals()) and np.array_equal(nochain.download_duals(), o.duals())
        o.ComputeForwardPass(); timed.forward_pass(); nochain.forward_pass()
        assert np.array_equal(timed.download_duals(), o.duals()) and np.array_equal(nochain.download_duals(), o.duals())
        timed.enable_kernel_timing(False)
        kt = timed.kernel_timing()
        assert sum(v["launches"] for v in kt.values()) > 9 and all(v.get("chain_launches", 0) == 0 for v in kt.values())   # launch by launch indeed
        o.ComputePassAndPrimal(7); timed.compute_pass_and_primal(7); nochain.compute_pass_and_primal(7)
        assert np.array_equal(timed.download_duals(), o.duals()) and np.array_equal(nochain.download_duals(), o.duals())
        assert np.array_equal(timed.download_primal(), o.primal())
    finally:
        timed.close(); nochain.close()


def test_mailbox_that_does_not_fit_the_device_is_planned_away(monkeypatch):
    """16 bytes per label and mailbox send (C3 row-major: 2 GB).  The engine gives every schedule of a model a budget (half of
    the device memory free at upload; LPMP_MAILBOX_MB overrides): a class whose rows would not fit keeps its completion flags
    instead of failing at upload — same duals, more dependencies"""
    m = MODELS["row-major grid, 16 labels"]()
    monkeypatch.delenv("LPMP_NO_MAILBOX", raising=False)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    monkeypatch.setenv("LPMP_MAILBOX_MB", "0")
    tight = E.Engine(0); tight.upload(m); tight.set_reparametrization(M.REPAM_ANISOTROPIC)
    monkeypatch.delenv("LPMP_MAILBOX_MB")
    roomy = E.Engine(0); roomy.upload(m); roomy.set_reparametrization(M.REPAM_ANISOTROPIC)
    try:
        ct, cr = tight.plan.chain_info(M.FORWARD, M.REPAM_ANISOTROPIC), roomy.plan.chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)
        assert ct["n_chains"] == 1 and ct["mailbox_rows"] == 0 and cr["mailbox_rows"] > 0 and ct["n_dependencies"] > cr["n_dependencies"]
        o.ComputePass(3); tight.compute_pass(3); roomy.compute_pass(3)
        assert np.array_equal(tight.download_duals(), o.duals()) and np.array_equal(roomy.download_duals(), o.duals())
    finally:
        tight.close(); roomy.close()
