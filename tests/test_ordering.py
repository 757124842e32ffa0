import numpy as np

from lp_mp_amd import engine as E, model as M, ordering, synthetic as S


def _levels(n, L, ei, ej, rank):
    i, j = rank[ei], rank[ej]
    m = S.mrf_model(n, L, np.minimum(i, j), np.maximum(i, j), np.zeros(n * L), potts=np.ones(ei.shape[0]))
    p = E.Plan(m)
    return [p.schedule_info(d, M.REPAM_ANISOTROPIC)["n_levels"] for d in (0, 1)]


def test_grid_gets_two_levels_and_random_graph_few():
    H, W = 12, 17
    a, b = S.grid_edges(H, W)
    rank = ordering.colour_major_order(H * W, a, b)
    assert _levels(H * W, 4, a, b, rank) == [2, 2]
    assert _levels(H * W, 4, a, b, np.arange(H * W)) == [H + W - 1, H + W - 1]
    g = S.random_graph_model(3000, 15000, 4, seed=2, pairwise="potts")
    ei, ej = g.m_left[0::2].astype(np.int64), g.m_left[1::2].astype(np.int64)
    col = ordering.greedy_colouring(3000, ei, ej)
    assert np.all(col[ei] != col[ej]) and col.min() == 0
    rank = ordering.colour_major_order(3000, ei, ej)
    lv = _levels(3000, 4, ei, ej, rank)
    lv_index = _levels(3000, 4, ei, ej, np.arange(3000))
    assert max(lv) <= int(col.max()) + 1 and max(lv) < min(lv_index)
    # an odd cycle is not bipartite
    assert ordering.two_colouring(3, np.array([0, 1, 0]), np.array([1, 2, 2])) is None


def test_colour_major_order_for_higher_order_factors():
    """variables that share a triplet / quadruple factor get different colours: in colour-major order no factor has
    two members in the same colour class, and the classes are few"""
    from lp_mp_amd.ordering import colour_major_order_hyper, greedy_colouring
    rng = np.random.default_rng(4)
    n = 5000
    base = rng.integers(0, n - 32, size=(3000, 1))
    tri = base[:2000] + np.argsort(rng.random((2000, 32)), axis=1)[:, :3]
    quad = base[2000:] + np.argsort(rng.random((1000, 32)), axis=1)[:, :4]
    rank = colour_major_order_hyper(n, [tri, quad], seed=1)
    assert sorted(rank.tolist()) == list(range(n))
    # recover the colour classes from the order: the rank sequence is sorted by colour
    ei = np.concatenate([tri[:, a] for a in range(3) for b in range(a + 1, 3)] + [quad[:, a] for a in range(4) for b in range(a + 1, 4)])
    ej = np.concatenate([tri[:, b] for a in range(3) for b in range(a + 1, 3)] + [quad[:, b] for a in range(4) for b in range(a + 1, 4)])
    col = greedy_colouring(n, ei, ej, 1)
    assert np.all(col[ei] != col[ej]) and col.max() < 24
    order = np.argsort(rank)
    assert np.all(np.diff(col[order]) >= 0)
