"""The Python mirror of the reference's LP<FMC> / Solver surface (lp_mp_amd/lp.py).  The GPU tests read like
the reference's own tests (test/test_model.cpp, test/graphical_model.cpp)."""
import numpy as np
import pytest

from lp_mp_amd import lp as LPM
from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S


def make_test_fmc():
    factor = LPM.FactorContainer(LPM.test_factor, 0)
    message = LPM.MessageContainer(LPM.test_message(), 0, 0, M.SCHED_LEFT, M.variableMessageNumber,
                                   M.variableMessageNumber, 0)
    return LPM.FMC("test model", [factor], [message]), factor, message


def FMC_SRMP():
    U = LPM.FactorContainer(LPM.UnarySimplexFactor, 0)
    P = LPM.FactorContainer(LPM.PairwiseSimplexFactor, 1)
    ML = LPM.MessageContainer(LPM.UnaryPairwiseMessage(0), 0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, 0)
    MR = LPM.MessageContainer(LPM.UnaryPairwiseMessage(1), 0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, 1)
    return LPM.FMC("SRMP", [U, P], [ML, MR]), U, P, ML, MR


def _grid_through_lp(H, W, L, seed):
    """the synthetic grid built call by call through add_factor / add_message / AddFactorRelation"""
    fmc, U, P, ML, MR = FMC_SRMP()
    lp = LPM.LP(fmc)
    ref = S.grid_model(H, W, L, seed=seed)
    n = H * W
    un = ref.dual_data[: n * L].reshape(n, L)
    u = [lp.add_factor(U, un[i]) for i in range(n)]
    a, b = S.grid_edges(H, W)
    tabs = ref.const_data.reshape(-1, L, L)
    for k in range(len(a)):
        p = lp.add_factor(P, L, L, tabs[k])
        lp.add_message(ML, u[a[k]], p)
        lp.add_message(MR, u[b[k]], p)
        lp.AddFactorRelation(u[a[k]], p)
        lp.AddFactorRelation(p, u[b[k]])
    return lp, ref


def test_flattening_matches_bulk_builder_up_to_insertion_order():
    lp, ref = _grid_through_lp(4, 5, 3, 2)
    m = lp.flat_model()
    assert m.n_factors == ref.n_factors and m.n_messages == ref.n_messages
    # same factors, but pairwise factors are interleaved with nothing here: unaries first, then edges
    assert np.array_equal(np.sort(m.const_data), np.sort(ref.const_data))
    assert lp.GetNumberOfFactors() == ref.n_factors and lp.GetNumberOfMessages() == ref.n_messages


def test_error_behaviour_mirrors_reference():
    fmc, factor, message = make_test_fmc()
    with pytest.raises(RuntimeError):
        LPM.LPReparametrizationModeConvert("bogus")                # config.hxx:88 throws runtime_error
    class UserFactor:                                              # an op without a device kind is rejected
        pass
    with pytest.raises(RuntimeError):
        LPM.FMC("x", [LPM.FactorContainer(UserFactor, 0)], [])
    lp = LPM.LP(fmc)
    f1 = lp.add_factor(factor, 0, 1)
    fmc2, U, P, ML, MR = FMC_SRMP()
    lp2 = LPM.LP(fmc2)
    a = lp2.add_factor(U, [0, 1]); b = lp2.add_factor(U, [0, 1])
    with pytest.raises(RuntimeError):
        lp2.add_message(ML, a, b)                                   # right factor is not a pairwise factor


@pytest.mark.gpu
def test_test_model_like_reference():
    """reference test/test_model.cpp:18-48"""
    fmc, factor, message = make_test_fmc()
    s = LPM.Solver(LPM.LP(fmc), LPM.StandardVisitor())
    lp = s.GetLP()
    f1 = lp.add_factor(factor, 0, 1)
    f2 = lp.add_factor(factor, 1, 0)
    f3 = lp.add_factor(factor, 0, 0)
    lp.add_message(message, f1, f2)
    lp.add_message(message, f1, f3)
    assert lp.GetNumberOfFactors() == 3
    assert lp.GetNumberOfMessages() == 2
    with pytest.raises(RuntimeError):
        lp.ComputePass(0)                                           # no reparametrization mode set (LP_MP.h:458)
    s.Solve()
    assert abs(s.GetLP().LowerBound() - 1.0) <= 1e-8
    assert s.iter == 1000                                           # default --maxIter


@pytest.mark.gpu
def test_solver_on_grid_matches_oracle_with_mode_switches():
    """the visitor switches to the rounding reparametrisation every 5th iteration (standard_visitor.hxx:172-185):
    both weight sets are resident and the device follows the same sequence as the oracle."""
    from oracle.binding import Oracle
    lp, ref = _grid_through_lp(6, 7, 4, 3)
    vis = LPM.StandardVisitor(maxIter=12)
    s = LPM.Solver(lp, vis)
    s.Solve()
    o = Oracle(lp.flat_model())
    v2 = LPM.StandardVisitor(maxIter=12)
    c = v2.begin(None)
    lbs = []
    while not c.end:
        o.set_reparametrization(c.repam)
        o.ComputePass(1)
        lbs.append(o.LowerBound())
        c = v2.visit(c, lbs[-1], np.inf)
    assert len(vis.lowerBound_) == len(lbs) == 12
    assert np.allclose(vis.lowerBound_, lbs, rtol=1e-9)
    assert np.array_equal(lp.duals(), o.duals())
    om = lp.get_omega()
    assert set(om) == {"forward", "backward", "receive_mask_forward", "receive_mask_backward"}


def test_quiet_iterations_are_what_the_visits_would_return():
    """StandardVisitor.quiet_iterations: the run it announces consists of iterations whose control asks for nothing,
    and it is maximal — the rule that lets Solver run them as one device call"""
    for kw in (dict(maxIter=40, lowerBoundComputationInterval=7, primalComputationInterval=11),
               dict(maxIter=25), dict(maxIter=33, lowerBoundComputationInterval=100, primalComputationInterval=3),
               dict(maxIter=3, lowerBoundComputationInterval=2, primalComputationInterval=50),
               dict(maxIter=30, lowerBoundComputationInterval=5, timeout=1000)):
        v = LPM.StandardVisitor(**kw)
        c = v.begin(None)
        visits = batched = 0
        while not c.end:
            q = v.quiet_iterations(c)
            assert q >= 1 and (q == 1 or "timeout" not in kw)
            first = c
            for _ in range(q):
                if q > 1:
                    assert not c.computeLowerBound and not c.computePrimal and not c.end and c.repam == first.repam
                    batched += 1
                c = v.visit(c, 0.0, np.inf)
                visits += 1
            if q > 1:
                assert c.end or c.computeLowerBound or c.computePrimal
        assert visits == kw["maxIter"]
        if kw.get("lowerBoundComputationInterval") == 7:
            assert batched > 20


@pytest.mark.gpu
def test_solver_batches_quiet_iterations_without_changing_the_result():
    """--lowerBoundComputationInterval 4: three of four iterations run as one joined device call; duals, bound
    history length and iteration count equal the oracle driven pass by pass with the same visitor"""
    from oracle.binding import Oracle
    lp, ref = _grid_through_lp(6, 7, 4, 5)
    kw = dict(maxIter=23, lowerBoundComputationInterval=4, primalComputationInterval=9)
    vis = LPM.StandardVisitor(**kw)
    s = LPM.Solver(lp, vis)
    calls = []
    orig = lp.ComputePasses
    lp.ComputePasses = lambda n: (calls.append(n), orig(n))[1]
    s.Solve()
    assert calls and max(calls) == 3 and s.iter == 23
    o = Oracle(lp.flat_model())
    v2 = LPM.StandardVisitor(**kw)
    c = v2.begin(None)
    lb = -np.inf
    while not c.end:
        o.set_reparametrization(c.repam)
        o.ComputePass(1)
        if c.computeLowerBound:
            lb = o.LowerBound()
        c = v2.visit(c, lb, np.inf)
    assert np.array_equal(lp.duals(), o.duals())
    assert len(vis.lowerBound_) == len(v2.lowerBound_) == 23
    assert np.allclose(vis.lowerBound_, v2.lowerBound_, rtol=1e-9)


@pytest.mark.gpu
def test_multicut_style_labeling_model_through_lp():
    from oracle.binding import Oracle
    edge = LPM.labeling_factor([(1,)], True)
    trip = LPM.labeling_factor(S.TRIPLET_LABELINGS, True)
    E_ = LPM.FactorContainer(edge, 0)
    T_ = LPM.FactorContainer(trip, 1)
    msgs = [LPM.MessageContainer(LPM.labeling_message(((1,),), tuple(S.TRIPLET_LABELINGS), (k,)), 0, 1, M.SCHED_LEFT,
                                 M.variableMessageNumber, 1, k) for k in range(3)]
    lp = LPM.LP(LPM.FMC("multicut", [E_, T_], msgs))
    e = [lp.add_factor(E_, [c]) for c in (0.7, -0.4, -0.9, 0.2, -0.3)]
    for tri in ((0, 1, 2), (1, 2, 3), (2, 3, 4)):
        t = lp.add_factor(T_)
        for k, ei in enumerate(tri):
            lp.add_message(msgs[k], e[ei], t)
            lp.AddFactorRelation(e[ei], t)
    lp.Begin()
    lp.set_reparametrization("anisotropic")
    o = Oracle(lp.flat_model())
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    for _ in range(5):
        lp.ComputePass(0)
        o.ComputePass(1)
    assert np.array_equal(lp.duals(), o.duals())
    assert abs(lp.LowerBound() - o.LowerBound()) <= 1e-12


@pytest.mark.gpu
def test_mp_rounding_solver_follows_the_oracle_iteration_by_iteration():
    """MpRoundingSolver (reference solver.hxx:380-400): on computePrimal iterations forward-and-primal, register,
    backward-and-primal, register; the same loop replayed on the oracle gives the same bounds, costs and labels."""
    from oracle.binding import Oracle
    U = LPM.FactorContainer(LPM.UnarySimplexFactor, 0, True)          # COMPUTE_PRIMAL_SOLUTION, as in FMC_SRMP
    P = LPM.FactorContainer(LPM.PairwiseSimplexFactor, 1)
    ML = LPM.MessageContainer(LPM.UnaryPairwiseMessage(0), 0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, 0)
    MR = LPM.MessageContainer(LPM.UnaryPairwiseMessage(1), 0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, 1)
    lp = LPM.LP(LPM.FMC("SRMP", [U, P], [ML, MR]))
    H, W, L = 5, 6, 4
    ref = S.grid_model(H, W, L, seed=8)
    n = H * W
    u = [lp.add_factor(U, ref.dual_data[i * L:(i + 1) * L]) for i in range(n)]
    a, b = S.grid_edges(H, W)
    tabs = ref.const_data.reshape(-1, L, L)
    for k in range(len(a)):
        p = lp.add_factor(P, L, L, tabs[k])
        lp.add_message(ML, u[a[k]], p); lp.add_message(MR, u[b[k]], p)
        lp.AddFactorRelation(u[a[k]], p); lp.AddFactorRelation(p, u[b[k]])
    vis = LPM.StandardVisitor(maxIter=17)
    s = LPM.MpRoundingSolver(lp, vis)
    s.Solve()

    o = Oracle(lp.flat_model())
    v2 = LPM.StandardVisitor(maxIter=17)
    c = v2.begin(None)
    best, it, lbs, sol = np.inf, 0, [], None
    while not c.end:
        o.set_reparametrization(c.repam)
        if c.computePrimal:
            for step in (o.ComputeForwardPassAndPrimal, o.ComputeBackwardPassAndPrimal):
                step(it)
                cost = o.EvaluatePrimal()
                if cost < best and o.CheckPrimalConsistency():
                    best, sol = cost, o.primal().copy()
        else:
            o.ComputePass(1)
        lbs.append(o.LowerBound())
        c = v2.visit(c, lbs[-1], best)
        it += 1
    cost = o.EvaluatePrimal()
    if cost < best and o.CheckPrimalConsistency():
        best, sol = cost, o.primal().copy()
    assert np.allclose(vis.lowerBound_, lbs, rtol=1e-9)
    assert np.isfinite(best) and abs(s.primal_cost() - best) <= 1e-9 * max(1.0, abs(best))
    assert np.array_equal(s.solution_, sol)
    assert np.array_equal(lp.duals(), o.duals())
    assert s.primal_cost() >= s.lower_bound() - 1e-9


@pytest.mark.gpu
def test_constant_factor_offsets_the_bound():
    """ConstantFactor (reference include/factors/constant_factor.hxx): no variables, no messages, dual = the offset"""
    U = LPM.FactorContainer(LPM.UnarySimplexFactor, 0)
    P = LPM.FactorContainer(LPM.PairwiseSimplexFactor, 1)
    K = LPM.FactorContainer(LPM.ConstantFactor, 2)
    ML = LPM.MessageContainer(LPM.UnaryPairwiseMessage(0), 0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, 0)
    MR = LPM.MessageContainer(LPM.UnaryPairwiseMessage(1), 0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, 1)
    lp = LPM.LP(LPM.FMC("SRMP + constant", [U, P, K], [ML, MR]))
    u1, u2 = lp.add_factor(U, [0.0, 1.0]), lp.add_factor(U, [1.0, 0.0])
    p = lp.add_factor(P, 2, 2, [[0.0, 1.0], [1.0, 0.0]])
    lp.add_factor(K, 1.5)
    lp.add_message(ML, u1, p); lp.add_message(MR, u2, p)
    lp.AddFactorRelation(u1, p); lp.AddFactorRelation(p, u2)
    lp.Begin()
    lp.set_reparametrization(LPM.LPReparametrizationMode.Anisotropic)
    assert lp.LowerBound() == pytest.approx(1.5)
    for it in range(5):
        lp.ComputePass(it)
    assert lp.LowerBound() == pytest.approx(2.5)


def test_lp_mirror_applies_the_engines_suggested_order():
    """an LP built call by call in row-major order (as `_grid_through_lp` inserts it: 10 dependent levels per direction on a 5 x 6
    grid) asks the engine for an order and applies it with AddFactorRelation calls: 2 levels per direction, same factors, messages
    and costs (host only: the plan needs no GPU)"""
    from lp_mp_amd.engine import Plan
    lp, ref = _grid_through_lp(5, 6, 3, 4)
    m0 = lp.flat_model()
    before = Plan(m0)
    assert [before.schedule_info(d, M.REPAM_ANISOTROPIC)["n_levels"] for d in (0, 1)] == [10, 10]
    by_rank, k = lp.suggested_order()
    assert sorted(by_rank) == list(range(lp.GetNumberOfFactors())) and k == 2
    assert lp.apply_suggested_order() == 2
    m = lp.flat_model()
    assert [Plan(m).schedule_info(d, M.REPAM_ANISOTROPIC)["n_levels"] for d in (0, 1)] == [2, 2]
    assert m.rel_fwd.shape[0] == m.n_factors - 1
    assert np.array_equal(m.const_data, m0.const_data) and np.array_equal(m.dual_data, m0.dual_data) and np.array_equal(m.m_left, m0.m_left)
