"""Primal rounding inside the sweep, CPU side: the oracle's restatement of UpdateFactorPrimal /
propagate_primal_through_messages / EvaluatePrimal (reference factors_messages.hxx:2332-2403, 1313-1344, 3302-3309;
LP_MP.h:914-940, 1067-1082, 1521-1536) checked against properties that hold for any correct implementation:
the primal cost is the energy of the rounded labeling on the ORIGINAL costs (reparametrisation invariance), it is
never below the lower bound, labels are first minimisers, time stamps gate re-initialisation."""
import itertools

import numpy as np
import pytest

from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle


def _random_mrf(rng, n, L, m_edges, potts=False):
    e = set()
    while len(e) < m_edges:
        i, j = sorted(rng.choice(n, 2, replace=False))
        e.add((int(i), int(j)))
    e = np.array(sorted(e))
    un = rng.uniform(0, 1, (n, L))
    if potts:
        d = rng.uniform(-0.3, 1, len(e))
        tabs = np.stack([dd * (1 - np.eye(L)) for dd in d])
        return S.mrf_model(n, L, e[:, 0], e[:, 1], un, potts=d, compute_primal=True), un, e, tabs
    tabs = rng.uniform(0, 1, (len(e), L, L))
    return S.mrf_model(n, L, e[:, 0], e[:, 1], un, tables=tabs, compute_primal=True), un, e, tabs


def _energy(un, e, tabs, x):
    return un[np.arange(len(x)), x].sum() + tabs[np.arange(len(e)), x[e[:, 0]], x[e[:, 1]]].sum()


@pytest.mark.parametrize("potts", [False, True])
@pytest.mark.parametrize("mode", [M.REPAM_ANISOTROPIC, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM])
def test_primal_cost_is_the_energy_of_the_rounded_labeling(potts, mode):
    rng = np.random.default_rng(11 + mode)
    m, un, e, tabs = _random_mrf(rng, 14, 4, 25, potts)
    o = Oracle(m)
    o.set_reparametrization(mode)
    assert o.EvaluatePrimal() == np.inf                      # nothing rounded yet: every primal_ is unset
    for it in range(5):
        for step in (o.ComputeForwardPassAndPrimal, o.ComputeBackwardPassAndPrimal):
            step(it)
            assert o.CheckPrimalConsistency()
            pr = o.primal()
            x = pr[:14, 0]
            assert np.all(x < 4)
            cost = o.EvaluatePrimal()
            assert abs(cost - _energy(un, e, tabs, x)) <= 1e-9
            assert cost >= o.LowerBound() - 1e-9
            # pairwise factors hold the pair of their endpoints' labels
            assert np.array_equal(pr[14:, 0], x[e[:, 0]]) and np.array_equal(pr[14:, 1], x[e[:, 1]])
        assert np.array_equal(o.primal_access(), np.full(m.n_factors, 2 * it + 2, np.uint64))


def test_tree_is_solved_to_optimality_and_rounded_to_it():
    # a chain: the LP relaxation is tight; after convergence the rounding reaches the brute-force optimum
    rng = np.random.default_rng(5)
    n, L = 6, 3
    e = np.array([(i, i + 1) for i in range(n - 1)])
    un = rng.uniform(0, 1, (n, L))
    tabs = rng.uniform(0, 1, (n - 1, L, L))
    m = S.mrf_model(n, L, e[:, 0], e[:, 1], un, tables=tabs, compute_primal=True)
    best = min(_energy(un, e, tabs, np.array(x)) for x in itertools.product(range(L), repeat=n))
    o = Oracle(m)
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    o.ComputePass(20)
    assert abs(o.LowerBound() - best) <= 1e-9
    o.ComputePassAndPrimal(20)
    assert abs(o.EvaluatePrimal() - best) <= 1e-9


def test_labels_are_first_minimisers_of_the_state_after_the_receives():
    # an isolated COMPUTE_PRIMAL unary is updated (FactorUpdated, factors_messages.hxx:3125-3130) and takes the first
    # of its tied minima; a unary with one edge rounds from theta + received min-marginal
    b = M.ModelBuilder(2, S.mrf_mtypes(), [1, 0])
    u = b.add_vector_factors(0, np.array([[3.0, 1.0, 1.0, 2.0], [0.5, 0.0, 0.25, 0.0], [0.0, 0.0, 0.0, 0.0]]))
    T = np.zeros((4, 4)); T[:, 3] = -1.0                       # column 3 is attractive for u1 whatever u0 takes
    p = b.add_dense_pairwise(1, T[None])[0]
    b.add_messages(0, u[0], p); b.add_messages(1, u[1], p)
    b.add_relations(u[0], p); b.add_relations(p, u[1])
    m = b.finish()
    o = Oracle(m)
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    o.ComputeForwardPassAndPrimal(0)
    pr = o.primal()
    assert pr[u[0], 0] == 1                                    # first of the two 1.0 entries
    assert pr[u[1], 0] == 3                                    # 0.0 - 1.0 beats the other 0.0 at index 1
    assert pr[u[2], 0] == 0                                    # isolated, all tied
    assert tuple(pr[p]) == (1, 3)
    assert abs(o.EvaluatePrimal() - (1.0 + 0.0 - 1.0 + 0.0)) <= 1e-12


def test_time_stamps_gate_reinitialisation():
    rng = np.random.default_rng(3)
    m, un, e, tabs = _random_mrf(rng, 10, 3, 15)
    o = Oracle(m)
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    o.ComputeForwardPassAndPrimal(4)
    first = o.primal().copy()
    # same time stamp again: conditionally_init_primal does nothing, MaximizePotentialAndComputePrimal keeps the
    # labels although the duals moved on
    o.ComputePass(3)
    o.ComputeForwardPassAndPrimal(4)
    assert np.array_equal(o.primal(), first)
    # a later one re-rounds from the new duals
    o.ComputeForwardPassAndPrimal(5)
    x = o.primal()[:10, 0]
    assert abs(o.EvaluatePrimal() - _energy(un, e, tabs, x)) <= 1e-9


def test_primal_passes_move_the_duals_like_plain_shared_passes():
    rng = np.random.default_rng(8)
    m, *_ = _random_mrf(rng, 12, 5, 20)
    a, b = Oracle(m), Oracle(m)
    for o in (a, b):
        o.set_reparametrization(M.REPAM_ANISOTROPIC)
    b.set_reparametrization_type(1)                            # UpdateFactorPrimal ignores the residual send rule
    a.ComputePass(2)
    b.ComputeForwardPassAndPrimal(0); b.ComputeBackwardPassAndPrimal(0); b.ComputePassAndPrimal(1)
    assert np.array_equal(a.duals(), b.duals())


def test_void_primal_hooks_on_both_sides_do_not_terminate():
    # test_message's ComputeRightFromLeftPrimal / ComputeLeftFromRightPrimal return void: the reference recurses
    # without end (factors_messages.hxx:1323-1327, 1339-1343); the oracle reports it instead of overflowing the stack
    mt = [M.MsgType(0, 0, M.SCHED_LEFT, 0, 0, M.M_MINNORM, 0)]
    b = M.ModelBuilder(1, mt, [1])
    f = b.add_vector_factors(0, np.array([[0.0, 1.0], [1.0, 0.0]]))
    b.add_messages(0, f[0], f[1])
    o = Oracle(b.finish())
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    with pytest.raises(RuntimeError, match="does not terminate"):
        o.ComputeForwardPassAndPrimal(0)
