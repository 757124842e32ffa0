"""Synthetic factor graphs for the measurement workloads of SURVEY.md 8(d).

The MRF layout mirrors how LP_MP-MRF's FMC_SRMP uses the reference containers (SURVEY.md A.5 and
Appendix B): factor type 0 = unary simplex, factor type 1 = pairwise (dense or Potts); two message
types, one per pairwise side, unary = left factor, schedule ``left``, unary side variable count,
pairwise side exactly 1; relations ``u_i -> p_ij -> u_j`` for i < j in the variable order.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import model as M

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def u01(n: int, seed: int, first: int = 0) -> np.ndarray:
    """Counter-based uniforms in [0,1): splitmix64 finaliser of seed + (first+i+1)*GOLDEN.
    Bit-identical to the engine's device generator (lpmp_synth_fill) and the oracle's orc_synth_u01."""
    if n >= (1 << 22):                       # long streams (the tables of an HBM-sized test model): block by block on a few threads
        from concurrent.futures import ThreadPoolExecutor
        import os
        out = np.empty(n, np.float64)
        nb = (n + (1 << 20) - 1) >> 20

        def fill(k):
            lo, hi = k << 20, min(n, (k + 1) << 20)
            w = np.empty(hi - lo, np.uint64)
            _u64_block(w, seed, first + lo)
            np.multiply(w >> np.uint64(11), 1.0 / 9007199254740992.0, out=out[lo:hi])
        with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as ex:
            list(ex.map(fill, range(nb)))
        return out
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(first + 1, first + n + 1, dtype=np.uint64)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _u64_block(out: np.ndarray, seed: int, first: int):
    n = out.shape[0]
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(first + 1, first + n + 1, dtype=np.uint64)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        np.bitwise_xor(z, z >> np.uint64(31), out=out)


def u64(n: int, seed: int, first: int = 0) -> np.ndarray:
    """the 64-bit words behind u01 (same counter-based stream).  Long streams (the 20 M words behind the C4 edge list) are
    filled block by block on a few threads: numpy releases the GIL inside its loops, and the stream is a pure function of
    the counter"""
    out = np.empty(n, np.uint64)
    if n < (1 << 21):
        _u64_block(out, seed, first)
        return out
    from concurrent.futures import ThreadPoolExecutor
    import os
    nt = max(1, min(8, os.cpu_count() or 1))
    cuts = [n * k // nt for k in range(nt + 1)]
    with ThreadPoolExecutor(nt) as ex:
        list(ex.map(lambda k: _u64_block(out[cuts[k]: cuts[k + 1]], seed, first + cuts[k]), range(nt)))
    return out


_EDGE_CACHE: dict = {}


def counter_graph_edges(n: int, m: int, seed: int, rank: Optional[np.ndarray] = None):
    """C4 (BASELINE.json configs[3]) structure from the counter generator alone, so that every rank of a partitioned
    run derives the same edge list without any communication: edge e joins a = h(2e) mod n and b = a + 1 + h(2e+1) mod
    (n - 1) (never a self loop); returned as (min, max) in edge order.  Uniform random edges; the expected handful of
    repeated pairs at m << n^2 / 2 are kept (two pairwise factors between the same variables are a valid model).
    ``rank``: the same graph with its variables renamed (variable v becomes rank[v]) — e.g. ordering.colour_major_order,
    which turns the 30 dependent levels per sweep of the index order into one level per colour."""
    key = (int(n), int(m), int(seed))
    if _EDGE_CACHE.get("key") != key:                  # (a run asks for the same graph several times: order, partition, parts)
        h = u64(2 * m, seed ^ 0x5DEECE66D, 0)
        a = (h[0::2] % np.uint64(n)).astype(np.int64)
        b = (a + 1 + (h[1::2] % np.uint64(n - 1)).astype(np.int64)) % n
        _EDGE_CACHE.clear(); _EDGE_CACHE.update(key=key, a=a, b=b)
    a, b = _EDGE_CACHE["a"], _EDGE_CACHE["b"]
    if rank is not None:
        rank = np.asarray(rank, np.int64)
        a, b = rank[a], rank[b]
    return np.minimum(a, b), np.maximum(a, b)


def counter_graph_model(n: int, m: int, L: int, seed: int = 1, device_const: bool = False, rank: Optional[np.ndarray] = None) -> M.FlatModel:
    """the unpartitioned C4-style model over counter_graph_edges: unaries u01 stream [0, n L), table of edge e at
    [n L + e L^2, ...) — the layout the per-rank generator of multi_gpu.graph_local_part reproduces piecewise"""
    i, j = counter_graph_edges(n, m, seed, rank)
    un = u01(n * L, seed, 0)
    if device_const:
        return mrf_model(n, L, i, j, un, device_const=True)
    return mrf_model(n, L, i, j, un, tables=u01(m * L * L, seed, n * L))


def mrf_mtypes():
    return [M.MsgType(0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, M.M_UNARY_PAIRWISE, 0),
            M.MsgType(0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, M.M_UNARY_PAIRWISE, 1)]


def grid_variable_order(H: int, W: int, order: str) -> np.ndarray:
    """var[r, c] = index of the variable in the insertion / relation order."""
    if order == "row_major":
        return np.arange(H * W, dtype=np.int64).reshape(H, W)
    if order == "colour_major":
        rr, cc = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        black = ((rr + cc) % 2 == 0).reshape(-1)
        var = np.empty(H * W, np.int64)
        nb = int(black.sum())
        var[black] = np.arange(nb)
        var[~black] = nb + np.arange(H * W - nb)
        return var.reshape(H, W)
    raise ValueError(order)


def grid_edges(H: int, W: int) -> Tuple[np.ndarray, np.ndarray]:
    """Edges in row-major node order, right edge then down edge per node; returns (a, b) as flat r*W+c."""
    rr, cc = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    node = (rr * W + cc)
    right_ok = cc < W - 1
    down_ok = rr < H - 1
    # interleave: for each node, right then down
    a = np.stack([node, node], -1).reshape(-1)
    b = np.stack([node + 1, node + W], -1).reshape(-1)
    ok = np.stack([right_ok, down_ok], -1).reshape(-1)
    return a[ok], b[ok]


def mrf_model(n_vars: int, L: int, edge_i: np.ndarray, edge_j: np.ndarray, unaries: np.ndarray,
              tables: Optional[np.ndarray] = None, potts: Optional[np.ndarray] = None,
              device_const: bool = False, compute_primal: bool = False, device_dual: bool = False) -> M.FlatModel:
    """MRF over variables 0..n_vars-1 (already in variable order) with edges (i<j required).
    ``compute_primal``: the unary FactorContainer's COMPUTE_PRIMAL_SOLUTION flag (as in LP_MP-MRF's FMC_SRMP);
    needed for the ...AndPrimal passes.
    ``tables`` [E,L,L] (T[e,a,b]: a = label of i) or ``potts`` [E] diffs. With ``device_const`` the dense
    tables are not materialised on the host (they are generated in HBM; see Engine.upload); with ``device_dual`` neither are the
    duals (``unaries`` is ignored: the caller fills the device buffer it hands to Engine.upload)."""
    edge_i = np.asarray(edge_i, np.int64)
    edge_j = np.asarray(edge_j, np.int64)
    assert np.all(edge_i < edge_j)
    b = M.ModelBuilder(2, mrf_mtypes(), [1, 0] if compute_primal else None)
    b.skip_const = device_const
    b.skip_dual = device_dual
    u = b.add_vector_factors(0, None, shape=(n_vars, L)) if device_dual else b.add_vector_factors(0, np.asarray(unaries, np.float64).reshape(n_vars, L))
    E = edge_i.shape[0]
    if potts is not None:
        assert not device_const
        p = b.add_potts_pairwise(1, L, potts)
    elif device_const:
        p = b.add_dense_pairwise(1, None, n=E, dims=(L, L))
    else:
        p = b.add_dense_pairwise(1, np.asarray(tables, np.float64).reshape(E, L, L))
    # add_message<ML>(u_i, p); add_message<MR>(u_j, p) per edge, interleaved like the reference's MRF constructor
    # (interleaved int32 arrays filled in place: at 10 M edges every temporary is 80 MB and a pass over it)
    ui, uj = u[edge_i], u[edge_j]
    mt = np.empty(2 * E, np.int32); mt[0::2] = 0; mt[1::2] = 1
    left = np.empty(2 * E, np.int32); left[0::2] = ui; left[1::2] = uj
    right = np.empty(2 * E, np.int32); right[0::2] = p; right[1::2] = p
    b.add_interleaved_messages(mt, left, right, return_ids=False)
    # AddFactorRelation(u_i, p); AddFactorRelation(p, u_j)
    f1 = np.empty(2 * E, np.int32); f1[0::2] = ui; f1[1::2] = p
    f2 = np.empty(2 * E, np.int32); f2[0::2] = p; f2[1::2] = uj
    b.add_relations(f1, f2)
    return b.finish()


def grid_model(H: int, W: int, L: int, pairwise: str = "dense", order: str = "row_major", seed: int = 1,
               unaries: Optional[np.ndarray] = None, tables: Optional[np.ndarray] = None,
               potts: Optional[np.ndarray] = None, device_const: bool = False, compute_primal: bool = False) -> M.FlatModel:
    """H x W grid MRF.  Random costs are U(0,1) from the counter-based generator: unaries first
    (variable order), then the pairwise data edge by edge (edge order of grid_edges)."""
    var = grid_variable_order(H, W, order).reshape(-1)
    a, bb = grid_edges(H, W)
    va, vb = var[a], var[bb]
    i, j = np.minimum(va, vb), np.maximum(va, vb)
    n = H * W
    E = i.shape[0]
    if unaries is None:
        unaries = u01(n * L, seed, 0)
    if pairwise == "dense":
        if tables is None and not device_const:
            tables = u01(E * L * L, seed, n * L)
        return mrf_model(n, L, i, j, unaries, tables=tables, device_const=device_const, compute_primal=compute_primal)
    if pairwise == "potts":
        if potts is None:
            potts = u01(E, seed, n * L)
        return mrf_model(n, L, i, j, unaries, potts=potts, compute_primal=compute_primal)
    raise ValueError(pairwise)


def chain_model(n: int, L: int, diff: float = 1.0, seed: int = 1) -> M.FlatModel:
    """C1: Potts chain, n variables, L labels (BASELINE.json configs[0])."""
    i = np.arange(n - 1)
    return mrf_model(n, L, i, i + 1, u01(n * L, seed, 0), potts=np.full(n - 1, diff))


def random_graph_model(n: int, m: int, L: int, seed: int = 1, pairwise: str = "dense", compute_primal: bool = False) -> M.FlatModel:
    """C4-style G(n, m): m distinct uniform random edges without self loops, variable order = index."""
    rng = np.random.Generator(np.random.PCG64(seed))
    got = np.zeros((0, 2), np.int64)
    while got.shape[0] < m:
        need = int((m - got.shape[0]) * 1.2) + 16
        e = rng.integers(0, n, size=(need, 2))
        e = e[e[:, 0] != e[:, 1]]
        e = np.stack([e.min(1), e.max(1)], 1)
        got = np.unique(np.concatenate([got, e]), axis=0)
    perm = rng.permutation(got.shape[0])[:m]
    e = got[np.sort(perm)]
    un = u01(n * L, seed, 0)
    if pairwise == "dense":
        return mrf_model(n, L, e[:, 0], e[:, 1], un, tables=u01(m * L * L, seed, n * L), compute_primal=compute_primal)
    return mrf_model(n, L, e[:, 0], e[:, 1], un, potts=u01(m, seed, n * L), compute_primal=compute_primal)


# ---- labeling-list (multicut-style) models: reference include/factors/labeling_list_factor.hxx ----
EDGE_LABELINGS = [(1,)]
TRIPLET_LABELINGS = [(0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1)]


def multicut_mtypes():
    """edge factor (type 0, left) <-> triplet factor (type 1, right), one message type per triplet position;
    tables are added in the same order by multicut_builder."""
    return [M.MsgType(0, 1, M.SCHED_LEFT, M.variableMessageNumber, 1, M.M_LABELING, k) for k in range(3)]


def multicut_builder() -> M.ModelBuilder:
    b = M.ModelBuilder(2, multicut_mtypes())
    for k in range(3):
        b.add_labeling_table(EDGE_LABELINGS, TRIPLET_LABELINGS, (k,))
    return b


def multicut_triangle_model(n_nodes: int, n_triangles: int, seed: int = 1) -> M.FlatModel:
    """Random multicut-style instance: edge factors (1 labeling, implicit origin) on the edges of random
    triangles, triplet factors (4 labelings, implicit origin), costs U(-1,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tris = np.sort(np.stack([rng.choice(n_nodes, 3, replace=False) for _ in range(n_triangles)]), 1)
    edges = {}
    for t in tris:
        for (x, y) in ((t[0], t[1]), (t[0], t[2]), (t[1], t[2])):
            edges.setdefault((int(x), int(y)), len(edges))
    b = multicut_builder()
    ecost = 2.0 * u01(len(edges), seed, 0) - 1.0
    e_ids = b.add_vector_factors(0, ecost.reshape(-1, 1), implicit_origin=True)
    tcost = np.zeros((n_triangles, 4))
    t_ids = b.add_vector_factors(1, tcost, implicit_origin=True)
    for ti, t in enumerate(tris):
        for k, (x, y) in enumerate(((t[0], t[1]), (t[0], t[2]), (t[1], t[2]))):
            e = e_ids[edges[(int(x), int(y))]]
            b.add_messages(k, e, t_ids[ti])
            b.add_relations(e, t_ids[ti])
    return b.finish()


QUAD_LABELINGS = [(1, 1, 0, 0), (0, 1, 1, 0), (0, 0, 1, 1), (1, 0, 0, 1), (1, 1, 1, 1), (1, 0, 1, 0), (0, 1, 0, 1)]


def c5_model(H: int, W: int, L: int, n_edge_vars: int, n_triplets: int, n_quads: int, seed: int = 1,
             order: str = "colour_major", window: int = 64, colour_edge_vars: bool = False) -> M.FlatModel:
    """C5 (BASELINE.json configs[4]): a C2-style Potts grid plus labeling-list higher-order factors of mixed arity
    in ONE factor graph — binary edge variables (1 labeling, implicit origin, costs U(-1,1)), triplet factors
    (4 labelings) on random local triples of them and quadruple factors (7 labelings) on random local quads,
    relations edge -> higher-order factor (reference include/factors/labeling_list_factor.hxx:220, 346).
    ``window``: a factor's members are drawn from ``window`` consecutive edge variables.  Small windows make long
    chains of edge variables that share factors — tens of thousands of dependent steps per sweep, whatever executes
    it; ``window = n_edge_vars`` (global triples) gives a handful.  ``colour_edge_vars``: the same factors with the
    edge variables inserted in a colour-major order (variables sharing a factor get different colours,
    ``ordering.colour_major_order_hyper``): the chains disappear, one level per colour."""
    mt = mrf_mtypes() + [M.MsgType(2, 3, M.SCHED_LEFT, 0, 1, M.M_LABELING, k) for k in range(3)] + \
        [M.MsgType(2, 4, M.SCHED_LEFT, 0, 1, M.M_LABELING, 3 + k) for k in range(4)]
    b = M.ModelBuilder(5, mt)
    for k in range(3):
        b.add_labeling_table(EDGE_LABELINGS, TRIPLET_LABELINGS, (k,))
    for k in range(4):
        b.add_labeling_table(EDGE_LABELINGS, QUAD_LABELINGS, (k,))
    n = H * W
    var = grid_variable_order(H, W, order).reshape(-1)
    a, bb = grid_edges(H, W)
    i, j = np.minimum(var[a], var[bb]), np.maximum(var[a], var[bb])
    u = b.add_vector_factors(0, u01(n * L, seed, 0).reshape(n, L))
    p = b.add_potts_pairwise(1, L, u01(len(a), seed, n * L))
    b.add_interleaved_messages(np.tile(np.array([0, 1], np.int32), len(a)), np.stack([u[i], u[j]], 1).reshape(-1), np.repeat(p, 2))
    b.add_relations(np.stack([u[i], p], 1).reshape(-1), np.stack([p, u[j]], 1).reshape(-1))
    rng = np.random.Generator(np.random.PCG64(seed))
    e = b.add_vector_factors(2, (2.0 * u01(n_edge_vars, seed + 1, 0) - 1.0).reshape(-1, 1), implicit_origin=True)

    def local_sets(count, arity):
        base = rng.integers(0, max(1, n_edge_vars - window), size=count)
        off = rng.integers(0, window, size=(count, arity))                     # distinct offsets inside the window
        while True:
            srt = np.sort(off, axis=1)
            bad = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
            if bad.size == 0:
                break
            off[bad] = rng.integers(0, window, size=(bad.size, arity))
        return np.minimum(base[:, None] + off, n_edge_vars - 1)

    sets = {3: local_sets(n_triplets, 3), 4: local_sets(n_quads, 4)}
    if colour_edge_vars:
        from .ordering import colour_major_order_hyper
        rank = colour_major_order_hyper(n_edge_vars, [v for v in sets.values() if v.shape[0]], seed)
        sets = {k: rank[v] for k, v in sets.items()}
    for ftype, dim, count, arity, first_mt in ((3, 4, n_triplets, 3, 2), (4, 7, n_quads, 4, 5)):
        if count == 0:
            continue
        f = b.add_vector_factors(ftype, np.zeros((count, dim)), implicit_origin=True)
        members = e[sets[arity]]
        b.add_interleaved_messages(np.tile(np.arange(first_mt, first_mt + arity, dtype=np.int32), count),
                                   members.reshape(-1), np.repeat(f, arity))
        b.add_relations(members.reshape(-1), np.repeat(f, arity))
    return b.finish()
