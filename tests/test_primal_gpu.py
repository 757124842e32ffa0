"""Primal rounding inside the sweep on the device (lpmp_compute_*_pass_and_primal, lpmp_evaluate_primal,
lpmp_check_primal_consistency) against the oracle: labels bit for bit, duals bit for bit, cost within 1e-9."""
import numpy as np
import pytest

from lp_mp_amd import engine as E
from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle

pytestmark = pytest.mark.gpu
MODES = (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM)


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(0)
    yield e
    e.close()


def _same(eng, o, what):
    assert np.array_equal(eng.download_primal(), o.primal()), what
    assert np.array_equal(eng.download_duals(), o.duals()), what
    assert eng.check_primal_consistency() == o.CheckPrimalConsistency()
    c, co = eng.evaluate_primal(), o.EvaluatePrimal()
    assert (c == co) if np.isinf(co) else abs(c - co) <= 1e-9 * max(1.0, abs(co)), (what, c, co)
    return c


def _run(eng, m, mode, iterations=3, rtype=0):
    o = Oracle(m)
    o.set_reparametrization_type(rtype); o.set_reparametrization(mode)
    eng.upload(m)
    eng.set_reparametrization_type(rtype); eng.set_reparametrization(mode)
    _same(eng, o, "unset")
    best = np.inf
    for it in range(iterations):
        eng.forward_pass_and_primal(it); o.ComputeForwardPassAndPrimal(it)
        best = min(best, _same(eng, o, ("forward", it)))
        eng.backward_pass_and_primal(it); o.ComputeBackwardPassAndPrimal(it)
        best = min(best, _same(eng, o, ("backward", it)))
        eng.compute_pass(1); o.ComputePass(1)                 # plain passes in between, as the solver loop does
    lb = eng.lower_bound()
    assert best >= lb - 1e-9 * max(1.0, abs(lb))
    eng.set_reparametrization_type(0)
    return best


@pytest.mark.parametrize("L", [2, 4, 5, 8, 16, 21, 32, 40, 70])
@pytest.mark.parametrize("pairwise", ["dense", "potts"])
@pytest.mark.parametrize("order", ["row_major", "colour_major"])
def test_grids_every_kernel_class(eng, L, pairwise, order):
    m = S.grid_model(9, 11, L, pairwise=pairwise, order=order, seed=40 + L, compute_primal=True)
    for mode in MODES:
        _run(eng, m, mode)


@pytest.mark.parametrize("L,pairwise", [(8, "dense"), (32, "dense"), (5, "dense"), (16, "potts")])
def test_rounding_passes_in_the_chain_executor(L, pairwise, monkeypatch):
    """rounding passes run the chain form of the packed kernels too: deep (row-major) sweeps as one persistent launch,
    big colour steps as a banded chain (forced here on a small grid), labels and duals against the oracle"""
    monkeypatch.setenv("LPMP_BAND_MIN_BYTES", "1000"); monkeypatch.setenv("LPMP_BAND_BYTES", "20000")
    e = E.Engine(0)
    try:
        for order, H, W in (("row_major", 14, 17), ("colour_major", 24, 26)):
            m = S.grid_model(H, W, L, pairwise=pairwise, order=order, seed=L + H, compute_primal=True)
            for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
                _run(e, m, mode, iterations=2)
                # (an anisotropic directional sweep of a 2-colour grid has one table-reading step: not banded)
                if pairwise == "dense" and (order == "row_major" or (mode == M.REPAM_DAMPED_UNIFORM and L in (8, 32))):
                    assert e.plan.chain_info(M.FORWARD, mode)["n_chains"] == 1
    finally:
        e.close()


def test_ties_take_the_first_minimum(eng):
    H, W, L = 8, 8, 8
    un = np.round(S.u01(H * W * L, 3) * 2.0) / 2.0            # costs in {0, 0.5, 1}: ties everywhere
    for pairwise, extra in (("potts", dict(potts=np.where(S.u01(112, 4) < 0.5, 0.5, 1.0))),
                            ("dense", dict(tables=np.round(S.u01(112 * L * L, 5) * 2.0) / 2.0))):
        m = S.grid_model(H, W, L, pairwise=pairwise, unaries=un, compute_primal=True, **extra)
        _run(eng, m, M.REPAM_ANISOTROPIC)
        _run(eng, m, M.REPAM_UNIFORM)


def test_random_graph_high_degree_and_residual_rule(eng):
    # degrees above the packet limits (indirect mode / generic kernel), and the residual send rule, which primal
    # passes ignore (UpdateFactorPrimal always calls SendMessages, reference factors_messages.hxx:2357-2359)
    for L, pw in ((16, "dense"), (8, "potts"), (11, "dense")):
        m = S.random_graph_model(60, 400, L, seed=L, pairwise=pw, compute_primal=True)
        _run(eng, m, M.REPAM_ANISOTROPIC, rtype=1)
        _run(eng, m, M.REPAM_DAMPED_UNIFORM)


def test_same_time_stamp_keeps_labels_later_one_rerounds(eng):
    m = S.grid_model(7, 6, 5, seed=9, compute_primal=True)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.upload(m); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.forward_pass_and_primal(3); o.ComputeForwardPassAndPrimal(3)
    first = eng.download_primal()
    eng.compute_pass(2); o.ComputePass(2)
    eng.forward_pass_and_primal(3); o.ComputeForwardPassAndPrimal(3)
    assert np.array_equal(eng.download_primal(), first)
    _same(eng, o, "same stamp")
    eng.compute_pass_and_primal(4); o.ComputePassAndPrimal(4)
    _same(eng, o, "later stamp")
    with pytest.raises(E.EngineError, match="must not decrease"):   # the reference asserts primal_access_ <= timestamp
        eng.forward_pass_and_primal(4)


def test_mixed_edge_kinds_isolated_unaries_and_rectangular_tables(eng):
    rng = np.random.default_rng(21)
    dims = [3, 6, 4, 9, 2, 6, 6, 5]
    b = M.ModelBuilder(2, S.mrf_mtypes(), [1, 0])
    u = [b.add_vector_factors(0, rng.uniform(0, 1, (1, d)))[0] for d in dims]
    for i, j in ((0, 1), (1, 2), (2, 3), (3, 4), (1, 5), (5, 6)):
        if dims[i] == dims[j] and rng.uniform() < 0.7:
            p = b.add_potts_pairwise(1, dims[i], [0.4])[0]
        else:
            p = b.add_dense_pairwise(1, rng.uniform(0, 1, (1, dims[i], dims[j])))[0]
        b.add_messages(0, u[i], p); b.add_messages(1, u[j], p)
        b.add_relations(u[i], p); b.add_relations(p, u[j])
    m = b.finish()                                           # u[7] has no edge: still updated and rounded
    for mode in MODES:
        _run(eng, m, mode)
    assert eng.download_primal()[u[7], 0] < dims[7]


def test_without_compute_primal_types_nothing_is_rounded(eng):
    m = S.grid_model(5, 5, 4, seed=2)                          # COMPUTE_PRIMAL false everywhere
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.upload(m); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.compute_pass_and_primal(0); o.ComputePassAndPrimal(0)
    _same(eng, o, "no primal types")
    assert eng.evaluate_primal() == np.inf


def test_upload_primal_and_evaluate(eng):
    H, W, L = 6, 5, 4
    m = S.grid_model(H, W, L, seed=12, compute_primal=True)
    eng.upload(m); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.compute_pass(3)
    n = H * W
    var = S.grid_variable_order(H, W, "row_major").reshape(-1)
    a, bb = S.grid_edges(H, W)
    i, j = np.minimum(var[a], var[bb]), np.maximum(var[a], var[bb])
    x = (np.arange(n) * 7) % L
    pr = np.zeros((m.n_factors, 2), np.int32)
    pr[:n, 0] = x; pr[n:, 0] = x[i]; pr[n:, 1] = x[j]
    eng.upload_primal(pr)
    un = S.u01(n * L, 12, 0).reshape(n, L)
    T = S.u01(len(i) * L * L, 12, n * L).reshape(-1, L, L)
    energy = un[np.arange(n), x].sum() + T[np.arange(len(i)), x[i], x[j]].sum()
    assert eng.check_primal_consistency()
    assert abs(eng.evaluate_primal() - energy) <= 1e-9       # invariant under the reparametrisation of 3 passes
    pr[n, 0] = (pr[n, 0] + 1) % L
    eng.upload_primal(pr)
    assert not eng.check_primal_consistency() and eng.evaluate_primal() == np.inf


def test_unsupported_models_are_refused(eng):
    m = S.multicut_triangle_model(6, 4, seed=1)
    eng.upload(m); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    with pytest.raises(E.EngineError) as ei:
        eng.forward_pass_and_primal(0)
    assert ei.value.code == -2                                 # LPMP_ERR_UNSUPPORTED
    b = M.ModelBuilder(2, S.mrf_mtypes(), [1, 1])             # two unaries on one side of a pairwise factor
    u = b.add_vector_factors(0, np.zeros((2, 3)))
    p = b.add_dense_pairwise(1, np.zeros((1, 3, 3)))[0]
    b.add_messages(0, u[0], p); b.add_messages(0, u[1], p)
    eng.upload(b.finish()); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    with pytest.raises(E.EngineError) as ei:
        eng.evaluate_primal()
    assert ei.value.code == -2


def test_full_size_c3_rounding_properties():
    # C3 (1024 x 1024, 32 labels, dense): no oracle at this size; cost >= bound, consistent, cost = energy of the labels
    import torch
    H = W = 1024; L = 32
    m = S.grid_model(H, W, L, order="colour_major", seed=1, device_const=True, compute_primal=True)
    n_e = len(S.grid_edges(H, W)[0])
    const = torch.empty(n_e * L * L, dtype=torch.float64, device="cuda:0")
    E.synth_fill(const.data_ptr(), const.numel(), 1, H * W * L, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    eng = E.Engine(0)
    try:
        eng.upload(m, const_dev=const.data_ptr(), keep=(const,))
        eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        eng.compute_pass(10)
        costs = []
        for it in range(3):
            eng.compute_pass_and_primal(it)
            assert eng.check_primal_consistency()
            costs.append(eng.evaluate_primal())
        lb = eng.lower_bound()
        assert all(np.isfinite(c) and c >= lb for c in costs)
        pr = eng.download_primal()
        x = pr[:H * W, 0].astype(np.int64)
        assert x.min() >= 0 and x.max() < L
        # energy on the original costs: unaries on the host, tables gathered on the device
        var = S.grid_variable_order(H, W, "colour_major").reshape(-1)
        a, bb = S.grid_edges(H, W)
        i, j = np.minimum(var[a], var[bb]), np.maximum(var[a], var[bb])
        un = S.u01(H * W * L, 1, 0).reshape(-1, L)
        idx = torch.from_numpy(np.arange(len(i)) * L * L + x[i] * L + x[j]).to("cuda:0")
        energy = un[np.arange(H * W), x].sum() + float(const[idx].sum().item())
        assert abs(costs[-1] - energy) <= 1e-7 * abs(energy)
    finally:
        eng.close()


@pytest.mark.parametrize("sched", [M.SCHED_FULL, M.SCHED_RIGHT])
def test_rounding_when_the_pairwise_factors_are_updated_too(eng, sched):
    """`full` / `right` schedules: the pairwise factors are updated (no primal of their own), the unaries round — under
    `right` the unaries have no active message at all and are updated only because their type computes a primal"""
    L, H, W = 6, 5, 6
    mt = [M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 1)]
    b = M.ModelBuilder(2, mt, [1, 0])
    rng = np.random.default_rng(17)
    var = S.grid_variable_order(H, W, "colour_major").reshape(-1)
    a, bb = S.grid_edges(H, W)
    i, j = np.minimum(var[a], var[bb]), np.maximum(var[a], var[bb])
    u = b.add_vector_factors(0, rng.uniform(0, 1, (H * W, L)))
    p = b.add_dense_pairwise(1, rng.uniform(0, 1, (len(a), L, L)))
    b.add_interleaved_messages(np.tile(np.array([0, 1], np.int32), len(a)), np.stack([u[i], u[j]], 1).reshape(-1), np.repeat(p, 2))
    b.add_relations(np.stack([u[i], p], 1).reshape(-1), np.stack([p, u[j]], 1).reshape(-1))
    m = b.finish()
    for mode in MODES:
        _run(eng, m, mode)


def _mrf(sched, computes, L, edges, n, rng, pairwise="dense", relations="chain", dims=None):
    """unaries 0..n-1, one pairwise factor per edge (i < j), messages of schedule `sched`; relations unary -> pairwise ->
    unary in variable order"""
    mt = [M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 1)]
    b = M.ModelBuilder(2, mt, list(computes))
    dims = dims if dims is not None else [L] * n
    u = np.concatenate([b.add_vector_factors(0, rng.uniform(0, 1, (1, dims[k]))) for k in range(n)])
    for (i, j) in edges:
        if pairwise == "potts" and dims[i] == dims[j]:
            p = b.add_potts_pairwise(1, dims[i], rng.uniform(0.1, 1, 1))[0]
        else:
            p = b.add_dense_pairwise(1, rng.uniform(0, 1, (1, dims[i], dims[j])))[0]
        b.add_messages(0, u[i], p); b.add_messages(1, u[j], p)
        b.add_relations(u[i], p); b.add_relations(p, u[j])
    return b.finish()


def _grid_edges_list(H, W):
    a, bb = S.grid_edges(H, W)
    return [(int(min(x, y)), int(max(x, y))) for x, y in zip(a, bb)]


@pytest.mark.parametrize("sched", [M.SCHED_FULL, M.SCHED_RIGHT])
@pytest.mark.parametrize("computes", [(0, 1), (1, 1)])
@pytest.mark.parametrize("L,pairwise", [(3, "dense"), (6, "dense"), (32, "dense"), (40, "dense"), (5, "potts"), (16, "potts")])
def test_pairwise_factors_that_round_themselves(eng, sched, computes, L, pairwise):
    """MPLP-style models (reference factors_messages.hxx:2332-2373 with a COMPUTE_PRIMAL_SOLUTION pairwise type): the
    updated pairwise factor fills its free sides given the labels its unaries already hold and labels them; the
    recursion of propagate_primal_through_messages then reaches the unaries' other pairwise factors"""
    rng = np.random.default_rng(L + 7 * sched + computes[0])
    H, W = (5, 6) if L <= 16 else (3, 4)
    m = _mrf(sched, computes, L, _grid_edges_list(H, W), H * W, rng, pairwise)
    for mode in MODES:
        best = _run(eng, m, mode, iterations=2)
        assert np.isfinite(best)                                   # every factor ends up labelled, consistently


def test_pairwise_rounding_trees_random_graphs_and_mixed_label_counts(eng):
    rng = np.random.default_rng(5)
    for trial in range(12):
        n = int(rng.integers(2, 14))
        dims = [int(x) for x in rng.integers(2, 9, n)]
        if trial % 2 == 0:                                          # tree
            edges = [(int(rng.integers(0, k)), k) for k in range(1, n)]
        else:
            pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
            pick = rng.choice(len(pairs), min(len(pairs), int(rng.integers(1, 3 * n))), replace=False)
            edges = [pairs[k] for k in sorted(pick)]
        for sched in (M.SCHED_FULL, M.SCHED_RIGHT, M.SCHED_LEFT):
            for computes in ((0, 1), (1, 1)):
                m = _mrf(sched, computes, 0, edges, n, rng, "potts" if trial % 3 == 0 else "dense", dims=dims)
                for mode in (M.REPAM_ANISOTROPIC, M.REPAM_UNIFORM):
                    _run(eng, m, mode, iterations=2)


def test_pairwise_rounding_ties_and_sides_without_a_unary(eng):
    """all-zero costs: every restricted minimiser is the first in row-major order; a pairwise factor with a message on
    one side only keeps the label of the other side in its own slot"""
    mt = [M.MsgType(0, 1, M.SCHED_FULL, 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, M.SCHED_FULL, 0, 1, M.M_UNARY_PAIRWISE, 1)]
    b = M.ModelBuilder(2, mt, [0, 1])
    u = b.add_vector_factors(0, np.zeros((3, 4)))
    p = b.add_dense_pairwise(1, np.zeros((3, 4, 4)))
    b.add_messages(0, u[0], p[0]); b.add_messages(1, u[1], p[0])
    b.add_messages(0, u[1], p[1]); b.add_messages(1, u[2], p[1])
    b.add_messages(1, u[2], p[2])                                   # p[2]: nothing on side 0
    b.add_relations(u[0], p[0]); b.add_relations(p[0], u[1]); b.add_relations(u[1], p[1]); b.add_relations(p[1], u[2]); b.add_relations(u[2], p[2])
    m = b.finish()
    for mode in MODES:
        _run(eng, m, mode, iterations=2)
    rng = np.random.default_rng(3)                                   # and with costs: the free side of p[2] is a real argmin
    b = M.ModelBuilder(2, mt, [0, 1])
    u = b.add_vector_factors(0, rng.uniform(0, 1, (3, 4)))
    p = b.add_dense_pairwise(1, rng.uniform(0, 1, (3, 4, 4)))
    b.add_messages(0, u[0], p[0]); b.add_messages(1, u[1], p[0]); b.add_messages(0, u[1], p[1]); b.add_messages(1, u[2], p[1]); b.add_messages(1, u[2], p[2])
    b.add_relations(u[0], p[0]); b.add_relations(p[0], u[1]); b.add_relations(u[1], p[1]); b.add_relations(p[1], u[2]); b.add_relations(u[2], p[2])
    m = b.finish()
    for mode in MODES:
        _run(eng, m, mode, iterations=2)
