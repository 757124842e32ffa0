"""csrc/graph.cpp behind the C ABI against its numpy statements (lp_mp_amd/ordering.py, multi_gpu.refine_partition), bit for bit,
and lpmp_plan_suggest_order: the order the engine suggests for a model inserted in a deep order."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lp_mp_amd import engine as E, model as M, multi_gpu as MG, ordering as O, synthetic as S  # noqa: E402
from oracle.binding import Oracle  # noqa: E402


@pytest.mark.parametrize("n,m,seed", [(20000, 100000, 1), (3000, 4000, 2), (500, 3000, 3), (64, 40, 4), (5, 0, 0)])
def test_colour_major_order_equals_the_numpy_statement_on_random_graphs(n, m, seed):
    ei, ej = S.counter_graph_edges(n, m, seed) if m else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    want = O.colour_major_order_numpy(n, ei, ej, seed)
    got, k = E.graph_colour_major_order(n, ei, ej, seed)
    assert np.array_equal(got, want) and sorted(got.tolist()) == list(range(n))
    assert np.array_equal(O.colour_major_order(n, ei, ej, seed), want)
    # a proper colouring: the two ends of an edge never share a colour class (classes are contiguous ranges of the order)
    col = np.searchsorted(np.cumsum(np.bincount(np.sort(_colours(got, n, ei, ej, seed)))), got, side="right") if m else None
    assert m == 0 or np.all(col[ei] != col[ej])
    assert 1 <= k <= 20


def _colours(rank, n, ei, ej, seed):
    col = O.two_colouring_by_component(n, ei, ej)
    return col if (col >= 0).all() else O.greedy_colouring(n, ei, ej, seed, colour=col)


def test_components_without_an_odd_cycle_keep_two_colours_beside_components_that_need_more():
    """a 2-colourable grid beside a clique and a triangle with a tail (one graph, several components): the grid's vertices get the
    BFS parity colours 0 / 1 as if the grid were alone, the others a greedy colouring — C++ and numpy alike"""
    gi, gj = S.grid_edges(6, 7)                                      # vertices 0..41
    k5 = [(42 + a, 42 + b) for a in range(5) for b in range(a + 1, 5)]
    tri = [(47, 48), (48, 49), (47, 49), (49, 50), (50, 51)]
    ei = np.concatenate([gi, [a for a, _ in k5 + tri]]); ej = np.concatenate([gj, [b for _, b in k5 + tri]])
    n = 54                                                           # 52, 53 isolated
    perm = np.random.default_rng(3).permutation(n)                   # scatter the components over the index range
    ei, ej = perm[ei], perm[ej]
    col = O.two_colouring_by_component(n, ei, ej)
    grid_v = perm[:42]
    alone = O.two_colouring(n, perm[gi], perm[gj])                   # the grid (+ isolated vertices) on its own
    assert np.array_equal(col[grid_v], alone[grid_v]) and set(col[perm[42:52]]) == {-1} and list(col[perm[52:]]) == [0, 0]
    want = O.colour_major_order_numpy(n, ei, ej, 5)
    got, k = E.graph_colour_major_order(n, ei, ej, 5)
    assert np.array_equal(got, want) and k == 5                      # the clique needs 5 colours; the grid's vertices sit in the first two classes
    full = _colours(got, n, ei, ej, 5)
    assert set(full[grid_v]) == {0, 1} and np.all(full[ei] != full[ej])


def test_bipartite_graphs_get_two_colours_and_self_loops_are_handled_like_numpy():
    gi, gj = S.grid_edges(30, 31)
    got, k = E.graph_colour_major_order(930, gi, gj, 0)
    assert k == 2 and np.array_equal(got, O.colour_major_order_numpy(930, gi, gj, 0))
    # several components, isolated vertices: the first vertex of every component is colour 0
    ei = np.array([5, 7, 1], np.int64); ej = np.array([6, 8, 9], np.int64)
    got, k = E.graph_colour_major_order(12, ei, ej, 3)
    assert k == 2 and np.array_equal(got, O.colour_major_order_numpy(12, ei, ej, 3))
    # an odd cycle / a self loop: not bipartite, the greedy colouring (which ignores the loop)
    for ei, ej in ((np.array([0, 1, 2]), np.array([1, 2, 0])), (np.array([0, 1, 3]), np.array([1, 2, 3]))):
        got, k = E.graph_colour_major_order(5, ei, ej, 1)
        assert np.array_equal(got, O.colour_major_order_numpy(5, ei, ej, 1))
    with pytest.raises(E.EngineError):
        E.graph_colour_major_order(3, np.array([0]), np.array([3]), 0)


def test_c5_shape_hyper_colouring_equals_the_numpy_statement():
    rng = np.random.default_rng(5)
    n = 4000
    tri = np.sort(rng.integers(0, n - 64, (1500, 1)) + rng.choice(64, 3, replace=False)[None, :] * 1, axis=1)
    quad = rng.integers(0, n - 64, (700, 1)) + np.array([[0, 5, 11, 40]])
    ei, ej = [], []
    for mem in (tri, quad):
        for a in range(mem.shape[1]):
            for b in range(a + 1, mem.shape[1]):
                ei.append(mem[:, a]); ej.append(mem[:, b])
    ei, ej = np.concatenate(ei), np.concatenate(ej)
    want = O.colour_major_order_numpy(n, ei, ej, 9)
    assert np.array_equal(O.colour_major_order_hyper(n, [tri, quad], 9), want)
    # the C5 generator's colour-major edge variables: every labeling-list factor's members in different colour classes = levels
    m = S.c5_model(16, 16, 4, 600, 300, 100, seed=5, window=16, colour_edge_vars=True)
    p = E.Plan(m)
    assert max(p.schedule_info(d, M.REPAM_ANISOTROPIC)["n_levels"] for d in (0, 1)) <= 24


@pytest.mark.parametrize("world", [2, 4, 8])
def test_partition_refinement_equals_the_numpy_statement_move_for_move(world):
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    for n, m, seed in ((20000, 100000, 1), (1500, 9000, 2)):
        ei, ej = S.counter_graph_edges(n, m, seed)
        a = coo_matrix((np.ones(m, np.float32), (ei, ej)), shape=(n, n)).tocsr(); a = (a + a.T).tocsr()
        order = reverse_cuthill_mckee(a, symmetric_mode=True)
        part = np.empty(n, np.int64); part[order] = (np.arange(n) * world) // n
        for rounds, imb, sd in ((30, 0.03, 0), (4, 0.10, 7)):
            want = MG.refine_partition(a, part, world, rounds, imb, sd)
            got = E.graph_refine_partition(n, ei, ej, part, world, rounds, imb, sd)
            assert np.array_equal(got, want), (n, world, rounds)
        assert np.array_equal(MG.graph_partition(n, ei, ej, world, method="builtin"), MG.refine_partition(a, part, world, 30, 0.03, 0))
        assert (want[ei] != want[ej]).mean() < (part[ei] != part[ej]).mean()


def _levels(model, mode=M.REPAM_ANISOTROPIC):
    p = E.Plan(model)
    return [p.schedule_info(d, mode)["n_levels"] for d in (0, 1)]


@pytest.mark.parametrize("pairwise", ["dense", "potts"])
def test_suggested_order_turns_a_row_major_grid_into_two_levels_per_direction(pairwise):
    """a C3-shaped grid inserted row by row: H + W - 1 dependent levels per direction; the order lpmp_plan_suggest_order hands
    back, applied as AddFactorRelation calls message by message (FlatModel.with_factor_order), has 2 — and the relations it
    yields are exactly the ones of a grid built in colour-major order: u_i -> p_ij -> u_j with the black cells first"""
    H, W, L = 14, 11, 4
    m = S.grid_model(H, W, L, pairwise=pairwise, order="row_major", seed=3)
    assert _levels(m) == [H + W - 1, H + W - 1]
    rank, k = E.Plan(m).suggest_order(0)
    assert k == 2 and sorted(rank.tolist()) == list(range(m.n_factors))
    m2 = m.with_factor_order(rank)
    assert _levels(m2) == [2, 2]
    n = H * W
    # every pairwise factor sits between its two unaries; all black unaries before all white ones
    l, r = m.m_left.reshape(-1, 2), m.m_right.reshape(-1, 2)
    assert np.all(rank[l].min(1) < rank[r[:, 0]]) and np.all(rank[r[:, 0]] < rank[l].max(1))
    rr, cc = np.divmod(np.arange(n), W)
    black = (rr + cc) % 2 == 0
    assert rank[:n][black].max() < rank[:n][~black].min()
    # the sweep in that order: the oracle's duals after 3 passes differ from the row-major run's (another trajectory), the bound grows
    o = Oracle(m2); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    lb0 = o.LowerBound(); o.ComputePass(3)
    assert o.LowerBound() > lb0


def test_suggested_order_keeps_higher_factors_behind_all_their_variables_where_the_caller_put_them_there():
    """multicut / C5 convention: relations edge variable -> labeling-list factor for ALL members; the suggestion keeps every such
    factor behind all its members and brings the local-triple chains down to one level per colour"""
    m = S.c5_model(12, 12, 4, 500, 260, 90, seed=7, window=16)
    before = _levels(m)
    rank, k = E.Plan(m).suggest_order(5)
    m2 = m.with_factor_order(rank)
    after = _levels(m2)
    assert max(after) <= k <= 24 and max(before) > 3 * max(after)
    is_right = np.zeros(m.n_factors, bool); is_right[m.m_right] = True
    lab = np.array([t.kind for t in m.mtypes])[m.m_type] == M.M_LABELING
    assert np.all(rank[m.m_left[lab]] < rank[m.m_right[lab]])            # members before their labeling-list factor, as inserted
    # same problem, another trajectory: both orders ascend from the same initial bound
    o1, o2 = Oracle(m), Oracle(m2)
    assert abs(o1.LowerBound() - o2.LowerBound()) < 1e-9
    for o in (o1, o2):
        o.set_reparametrization(M.REPAM_ANISOTROPIC); o.ComputePass(2)
    assert o2.LowerBound() > -1e30 and o1.LowerBound() > -1e30


def test_deep_schedule_note_names_the_remedy(capfd):
    """planning a sweep of more than 64 levels through the ENGINE prints one line to stderr; host-only plans stay silent"""
    m = S.grid_model(40, 40, 4, pairwise="potts", order="row_major", seed=1)
    E.Plan(m).schedule_info(0, M.REPAM_ANISOTROPIC)
    assert "lpmp_plan_suggest_order" not in capfd.readouterr().err


def _random_models(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 4
    if kind == 0:
        return S.grid_model(int(rng.integers(3, 12)), int(rng.integers(3, 12)), int(rng.choice([2, 3, 5])), pairwise=str(rng.choice(["dense", "potts"])), order="row_major", seed=seed)
    if kind == 1:
        return S.counter_graph_model(int(rng.integers(30, 200)), int(rng.integers(60, 500)), int(rng.choice([2, 4])), seed)
    if kind == 2:
        return S.c5_model(int(rng.integers(2, 6)), int(rng.integers(2, 6)), 3, int(rng.integers(20, 90)), int(rng.integers(8, 50)), int(rng.integers(2, 20)), seed=seed, window=int(rng.integers(6, 30)))
    return S.multicut_triangle_model(int(rng.integers(8, 40)), int(rng.integers(6, 60)), seed=seed)


@pytest.mark.parametrize("seed", range(16))
def test_suggested_order_on_random_models(seed):
    """random grids, graphs, C5-style and multicut models: the suggestion is a permutation; in that order no sweep has more dependent
    levels than colours; every message keeps its direction relative to the higher factor it ends in (a factor that came after k
    of its updated neighbours comes after k of them again); and the oracle's bound ascends from the same start as in the
    caller's order"""
    m = _random_models(seed)
    p = E.Plan(m)
    rank, k = p.suggest_order(seed)
    assert sorted(rank.tolist()) == list(range(m.n_factors))
    m2 = m.with_factor_order(rank)
    p2 = E.Plan(m2)
    for mode in (M.REPAM_ANISOTROPIC, M.REPAM_UNIFORM):
        lv = [p2.schedule_info(d, mode)["n_levels"] for d in (0, 1)]
        assert max(lv) <= max(k, 1), (seed, lv, k)
    # how many of its updated neighbours precede every non-updated factor: unchanged
    upd0 = np.zeros(m.n_factors, bool); upd0[p.update_order(M.FORWARD)] = True
    pos0 = np.empty(m.n_factors, np.int64); pos0[p.order(M.FORWARD)] = np.arange(m.n_factors)
    pos1 = np.empty(m.n_factors, np.int64); pos1[p2.order(M.FORWARD)] = np.arange(m.n_factors)
    assert np.array_equal(pos1, rank)                                   # the chain's only topological order is the suggestion
    l, r = m.m_left.astype(np.int64), m.m_right.astype(np.int64)
    for a, b in ((l, r), (r, l)):                                       # a: updated end, b: other end not updated
        sel = upd0[a] & ~upd0[b]
        before0 = np.bincount(b[sel], weights=(pos0[a[sel]] < pos0[b[sel]]), minlength=m.n_factors)
        before1 = np.bincount(b[sel], weights=(pos1[a[sel]] < pos1[b[sel]]), minlength=m.n_factors)
        assert np.array_equal(before0, before1), seed
    o1, o2 = Oracle(m), Oracle(m2)
    assert abs(o1.LowerBound() - o2.LowerBound()) <= 1e-9 * max(1.0, abs(o1.LowerBound()))
    o2.set_reparametrization(M.REPAM_ANISOTROPIC)
    last = o2.LowerBound()
    for _ in range(3):
        o2.ComputePass(1)
        assert o2.LowerBound() >= last - 1e-9 * max(1.0, abs(last))
        last = o2.LowerBound()


@pytest.mark.parametrize("seed", range(40))
def test_suggested_order_on_models_of_every_schedule(seed):
    """models with `right` / `full` / `only send` messages (higher factors updated too, messages between vector factors, labeling
    lists: tests/test_fuzz_gpu.random_model): two updates conflict whenever what they touch overlaps — also around an UPDATED
    factor — so in the suggested order no sweep of any weight mode has more dependent levels than colours"""
    from tests.test_fuzz_gpu import random_model
    m = random_model(np.random.default_rng(7000 + seed))
    rank, k = E.Plan(m).suggest_order(seed)
    assert sorted(rank.tolist()) == list(range(m.n_factors))
    p2 = E.Plan(m.with_factor_order(rank))
    for mode in (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM):
        assert max(p2.schedule_info(d, mode)["n_levels"] for d in (0, 1)) <= max(k, 1), (seed, mode, k)
