"""Variable orders that expose parallelism.

The sweep is Gauss-Seidel over the factor ORDER, which is input data (AddFactorRelation, reference
include/LP_MP.h:698-702): the engine runs whatever order it is given, level by level.  A grid in row-major
variable order has H+W-1 dependent levels per direction; the same grid in a 2-colour order has 2.  These helpers
compute a colour-major variable order for an arbitrary pairwise graph: variables sorted by colour, so that with
relations u_i -> p_ij -> u_j (i < j) every colour class is one level.  Different order = different (equally valid)
trajectory of the dual ascent, not a different algorithm.
"""
from __future__ import annotations

import numpy as np


def two_colouring(n: int, ei: np.ndarray, ej: np.ndarray):
    """colours in {0,1} if the graph is bipartite, else None.  One connected-components call on the bipartite double cover
    (vertices (v, 0) = v and (v, 1) = n + v, an edge u - v becomes (u, 0) - (v, 1) and (u, 1) - (v, 0)): the graph is bipartite iff
    no v has both copies in one component, and then "which copy of v lies in the component labelled first" is a proper
    colouring in which the first vertex of every component of the graph gets colour 0 (= BFS depth parity from that vertex)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    ei = np.asarray(ei, np.int64); ej = np.asarray(ej, np.int64)
    rows = np.concatenate([ei, ei + n]); cols = np.concatenate([ej + n, ej])
    cover = coo_matrix((np.ones(rows.shape[0], np.int8), (rows, cols)), shape=(2 * n, 2 * n)).tocsr()
    _, comp = connected_components(cover, directed=False)
    if np.any(comp[:n] == comp[n:]):
        return None
    return (comp[:n] > comp[n:]).astype(np.int64)


def two_colouring_by_component(n: int, ei: np.ndarray, ej: np.ndarray) -> np.ndarray:
    """colours in {0, 1} for the vertices of every connected component without an odd cycle (BFS depth parity from the component's
    first vertex, as two_colouring gives it), -1 for the vertices of all other components.  A model of several parts — a
    2-colourable grid beside higher-order factors — keeps the 2 levels of its grid whatever the rest needs."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    ei = np.asarray(ei, np.int64); ej = np.asarray(ej, np.int64)
    rows = np.concatenate([ei, ei + n]); cols = np.concatenate([ej + n, ej])
    cover = coo_matrix((np.ones(rows.shape[0], np.int8), (rows, cols)), shape=(2 * n, 2 * n)).tocsr()
    _, comp = connected_components(cover, directed=False)
    graph = coo_matrix((np.ones(ei.shape[0], np.int8), (ei, ej)), shape=(n, n)).tocsr()
    _, part = connected_components(graph, directed=False)
    odd = np.zeros(int(part.max()) + 1 if n else 0, bool)
    odd[part[comp[:n] == comp[n:]]] = True                              # both copies of a vertex in one cover component: an odd cycle
    return np.where(odd[part], -1, (comp[:n] > comp[n:]).astype(np.int64))


def greedy_colouring(n: int, ei: np.ndarray, ej: np.ndarray, seed: int = 0, colour=None) -> np.ndarray:
    """Luby / Jones-Plassmann style parallel greedy colouring with numpy: in every round the uncoloured vertices that
    beat all their uncoloured neighbours (random priorities) take the smallest colour their neighbours do not use.
    At most 63 colours (max degree < 63 is plenty for sparse MRFs).  A round looks only at the directed edges whose
    first end is still uncoloured (most vertices are coloured in the first few rounds)."""
    # priorities: the counter hash of the vertex (ties by index) as a permutation — the generator the C++ form shares
    # (csrc/graph.cpp greedy_colouring; tests/test_graph_host.py holds the two against each other)
    from .synthetic import u64
    prio = np.empty(n, np.int64)
    prio[np.argsort(u64(n, seed), kind="stable")] = np.arange(n)
    # (``colour``: vertices that already hold one — the 2-colourable components — keep it; they are adjacent to none of the others)
    colour = np.full(n, -1, np.int64) if colour is None else np.array(colour, np.int64)
    a = np.concatenate([ei, ej]); b = np.concatenate([ej, ei])          # directed both ways
    while True:
        un = colour < 0
        if not un.any():
            return colour
        live = un[a]
        if not live.all():
            a, b = a[live], b[live]
        unb = un[b]
        loser = np.zeros(n, bool)
        loser[a[unb & (prio[a] < prio[b])]] = True
        cand = un & ~loser
        used = np.zeros(n, np.uint64)
        m = cand[a] & ~unb                                              # coloured neighbours of candidates
        if m.any():
            np.bitwise_or.at(used, a[m], np.uint64(1) << colour[b[m]].astype(np.uint64))
        free = ~used[cand]
        low = free & (~free + np.uint64(1))                             # lowest set bit
        c = np.log2(low.astype(np.float64)).astype(np.int64)
        if np.any(c >= 63):
            raise RuntimeError("graph needs more than 63 colours")
        colour[cand] = c


def colour_major_order(n: int, ei: np.ndarray, ej: np.ndarray, seed: int = 0) -> np.ndarray:
    """rank[v] = position of variable v in a colour-major order (2 colours for every component without an odd cycle, a greedy
    colouring for the others): computed on the planner's
    threads behind the C ABI (lpmp_graph_colour_major_order) — a colouring of 2 M variables / 10 M edges is 6 s of numpy on the
    GPU box and a fraction of a second there; colour_major_order_numpy is the same algorithm as readable numpy, same result"""
    from . import engine as E
    return E.graph_colour_major_order(n, ei, ej, seed)[0]


def colour_major_order_numpy(n: int, ei: np.ndarray, ej: np.ndarray, seed: int = 0) -> np.ndarray:
    ei = np.asarray(ei, np.int64); ej = np.asarray(ej, np.int64)
    col = two_colouring_by_component(n, ei, ej)
    if (col < 0).any():
        col = greedy_colouring(n, ei, ej, seed, colour=col)
    order = np.argsort(col, kind="stable")
    rank = np.empty(n, np.int64)
    rank[order] = np.arange(n)
    return rank


def colour_major_order_hyper(n: int, members: list, seed: int = 0) -> np.ndarray:
    """The same for higher-order factors: ``members`` is a list of [count, arity] index arrays (one per factor
    family); variables that share a factor get different colours (every factor is a clique of conflicts)."""
    ei, ej = [], []
    for mem in members:
        mem = np.asarray(mem, np.int64)
        for a in range(mem.shape[1]):
            for b in range(a + 1, mem.shape[1]):
                ei.append(mem[:, a]); ej.append(mem[:, b])
    ei = np.concatenate(ei); ej = np.concatenate(ej)
    keep = ei != ej
    # (a factor of three or more variables is a triangle of conflicts: never 2-colourable, so this is the greedy colouring; two
    # variables per factor at most: the 2-colouring where one exists)
    return colour_major_order(n, ei[keep], ej[keep], seed)
