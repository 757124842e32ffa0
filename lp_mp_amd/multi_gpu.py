"""Partitioned sweep: one process per GPU, cut-edge messages exchanged once per directional sweep.

The factor graph (unary / pairwise MRF) is split by variables.  Each pairwise factor is owned by the
part of its earlier endpoint; for a cut edge the owner keeps a zero-cost GHOST unary in place of the
remote endpoint.  One directional sweep is, in the reference's own terms (the public iterator-range
``LP::ComputePass(factorIt, factorItEnd, omegaIt, receive_it)``, reference include/LP_MP.h:981-1005, the
same mechanism compute_partition_pass uses, :1932-1963):

  1. main sweep   — every part runs its level-scheduled update list with its own anisotropic weights
                    (``ComputeAnisotropicWeights`` on the part's sub-graph, :1232-1415); ghost factors are
                    not updated, so parts touch disjoint memory and may run concurrently;
  2. boundary step — ``UpdateFactor`` of every non-owner endpoint u_j restricted to its cut messages:
                    receive the owner's min-marginal (weight 1), then send back omega_b * theta_j
                    (as two iterator-range passes: all receives, then all sends — the same thing for an MRF, where
                    a pairwise factor has at most one remote endpoint; with higher-order factors two remote
                    variables may share a factor).
     On the device the two halves of that update live on different GPUs, so the ghost carries the
     message: owner  ghost <- min-marginal (receive-only pass on the ghosts), ship ghost -> remote,
              remote theta_j += delta; delta' = omega_b * theta_j; theta_j -= delta'; ship delta' back
              (omega_b = 1/(k+1) for an endpoint with k cut messages),
              owner  ghost <- delta'; send-only pass (omega 1) folds it into the pairwise factor.
     Two RCCL all-to-all exchanges of (cut edges x L) doubles per directional sweep.

Because every step is an iterator-range pass of the reference, the whole schedule can be replayed by
the oracle on the unpartitioned model (tests/test_multi_gpu.py does exactly that).
"""
from __future__ import annotations

import os

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import model as M
from . import synthetic as S


@dataclass
class LocalPart:
    rank: int
    world: int
    L: int
    model: M.FlatModel                  # local unaries, ghosts, owned pairwise (mrf_model layout)
    n_local: int                        # local unaries = factors [0, n_local)
    n_ghost: int                        # ghosts = factors [n_local, n_local + n_ghost)
    local_to_global: np.ndarray         # local factor id -> global factor id (ghost -> remote unary)
    local_msg_to_global: np.ndarray     # local message id -> global message id
    out_peer: np.ndarray                # owned cut edges, sorted by (peer, key): remote rank
    out_ghost: np.ndarray               #   local ghost factor id
    out_key: np.ndarray                 #   global edge id
    in_peer: np.ndarray                 # cut edges owned elsewhere, sorted by (peer, key): owner rank
    in_unary: np.ndarray                #   local unary factor id of the remote endpoint
    in_key: np.ndarray                  #   global edge id
    const_fill: Optional[list] = None   # [(offset, count, seed, first)] when tables are generated in HBM
    dual_fill: Optional[list] = None
    in_pos: Optional[np.ndarray] = None # position of each incoming cut message in its left factor's global message list
    key_is_msg: bool = False            # keys are global message ids (partition_model) instead of MRF edge ids
    ghost_order: Optional[np.ndarray] = None   # order in which the ghost passes visit the ghosts (default: by index)


def _sorted_by_peer_key(peer, *cols, key):
    order = np.lexsort((key, peer))
    return [np.ascontiguousarray(peer[order])] + [np.ascontiguousarray(c[order]) for c in cols] + [np.ascontiguousarray(key[order])]


def partition_mrf(n_vars: int, L: int, edge_i: np.ndarray, edge_j: np.ndarray, part: np.ndarray, world: int,
                  unaries: Optional[np.ndarray] = None, tables: Optional[np.ndarray] = None, potts: Optional[np.ndarray] = None,
                  only: Optional[int] = None, stream_seed: Optional[int] = None) -> List[LocalPart]:
    """General partitioner for an MRF given as in synthetic.mrf_model (edge e between variables
    edge_i[e] < edge_j[e]); ``part[v]`` = owning rank of variable v.
    ``only``: build this rank's part alone (what a rank of a multi-process run needs).
    ``stream_seed``: the costs are not given as host arrays but live in the counter-based stream of
    synthetic.counter_graph_model (unaries at [0, n L), the table of edge e at [n L + e L^2, ...)): the part's model
    is built without cost arrays and carries block-fill descriptors (``const_fill`` / ``dual_fill`` =
    ("blocks", block_len, seed, first[])) for lpmp_synth_fill_blocks — no rank ever holds the global costs."""
    edge_i = np.asarray(edge_i, np.int64)
    edge_j = np.asarray(edge_j, np.int64)
    part = np.asarray(part, np.int64)
    if stream_seed is None:
        unaries = np.asarray(unaries, np.float64).reshape(n_vars, L)
    n_edges = edge_i.shape[0]
    owner = part[edge_i]
    parts = []
    for k in (range(world) if only is None else [only]):
        lv = np.nonzero(part == k)[0]
        gmap = np.full(n_vars, -1, np.int64)
        gmap[lv] = np.arange(lv.shape[0])
        le = np.nonzero(owner == k)[0]                      # owned edges, global order
        cut = part[edge_j[le]] != k
        n_ghost = int(cut.sum())
        li = gmap[edge_i[le]]
        lj = gmap[edge_j[le]].copy()
        lj[cut] = lv.shape[0] + np.arange(n_ghost)
        const_fill = dual_fill = None
        if stream_seed is not None:
            m = S.mrf_model(lv.shape[0] + n_ghost, L, li, lj, None, device_const=True, device_dual=True)
            const_fill = [("blocks", L * L, stream_seed, (n_vars * L + le * (L * L)).astype(np.int64))]
            dual_fill = [("blocks", L, stream_seed, (lv * L).astype(np.int64))]
        else:
            un = np.concatenate([unaries[lv], np.zeros((n_ghost, L))])
            kw = {}
            if potts is not None:
                kw["potts"] = np.asarray(potts, np.float64)[le]
            else:
                kw["tables"] = np.asarray(tables, np.float64).reshape(n_edges, L, L)[le]
            m = S.mrf_model(lv.shape[0] + n_ghost, L, li, lj, un, **kw)
        n_vec = lv.shape[0] + n_ghost
        l2g = np.concatenate([lv, edge_j[le][cut], n_vars + le]).astype(np.int64)
        lm2g = np.stack([2 * le, 2 * le + 1], 1).reshape(-1).astype(np.int64)
        out_peer, out_ghost, out_key = _sorted_by_peer_key(part[edge_j[le][cut]], (lv.shape[0] + np.arange(n_ghost)).astype(np.int32), key=le[cut])
        ine = np.nonzero((part[edge_j] == k) & (owner != k))[0]
        in_peer, in_unary, in_key = _sorted_by_peer_key(owner[ine], gmap[edge_j[ine]].astype(np.int32), key=ine)
        assert n_vec + le.shape[0] == m.n_factors
        parts.append(LocalPart(k, world, L, m, lv.shape[0], n_ghost, l2g, lm2g, out_peer, out_ghost, out_key,
                               in_peer, in_unary, in_key, const_fill, dual_fill))
    return parts


def graph_local_part(n: int, m: int, L: int, rank: int, world: int, seed: int = 1, part: Optional[np.ndarray] = None,
                     var_rank: Optional[np.ndarray] = None) -> LocalPart:
    """this rank's part of the C4-style model synthetic.counter_graph_model(n, m, L, seed): structure from the counter
    generator (every rank derives the same edge list without communication), costs generated in this rank's HBM.
    ``part``: the variable -> rank map; a multi-process run computes it ONCE on rank 0 and broadcasts it
    (broadcast_partition); without it every caller runs the partitioner itself (same result, deterministic)."""
    ei, ej = S.counter_graph_edges(n, m, seed, var_rank)          # (var_rank: the variables renamed, counter_graph_model(..., rank=var_rank))
    if world == 1:
        # one part = the whole model: what partition_mrf builds for it, without its index arithmetic over 10 M edges
        mdl = S.mrf_model(n, L, ei, ej, None, device_const=True, device_dual=True)
        e = np.arange(m, dtype=np.int64)
        none_i, none_32 = np.zeros(0, np.int64), np.zeros(0, np.int32)
        return LocalPart(0, 1, L, mdl, n, 0, np.arange(n + m, dtype=np.int64), np.arange(2 * m, dtype=np.int64), none_i, none_32, none_i,
                         none_i, none_32, none_i, [("blocks", L * L, seed, n * L + e * (L * L))], [("blocks", L, seed, np.arange(n, dtype=np.int64) * L)])
    if part is None:
        part = graph_partition(n, ei, ej, world)
    return partition_mrf(n, L, ei, ej, part, world, only=rank, stream_seed=seed)[0]


_HOST_GROUP = {}


def host_group(dist):
    """a gloo group beside the RCCL one, for host-side hand-offs: a rank waiting in it for rank 0's host work (the partitioner:
    the better part of a minute at 2 M variables, more on a loaded box) does not sit inside an RCCL kernel under the NCCL
    watchdog (10 minutes by default), and its wait has a timeout of its own (an hour)"""
    if dist.get_backend() == "gloo":
        return None                                        # the default group already is one
    key = dist.get_world_size()
    if key not in _HOST_GROUP:
        import datetime
        _HOST_GROUP[key] = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=1))
    return _HOST_GROUP[key]


def broadcast_partition(torch, dist, n: int, device, compute) -> np.ndarray:
    """the partition of a multi-process run: rank 0 calls ``compute()`` (host work), the others receive the result — one
    8 n-byte broadcast over the host group instead of N identical partitioner runs"""
    rank = dist.get_rank()
    t = torch.from_numpy(np.ascontiguousarray(compute(), np.int64)) if rank == 0 else torch.empty(n, dtype=torch.int64)
    dist.broadcast(t, 0, group=host_group(dist))
    return t.numpy()


def broadcast_string(dist, text: Optional[str]) -> Optional[str]:
    """rank 0's string on every rank (host group): which partitioner ran there"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return text
    box = [text]
    dist.broadcast_object_list(box, 0, group=host_group(dist))
    return box[0]


def partition_model(gm: M.FlatModel, part: np.ndarray, world: int) -> List[LocalPart]:
    """General partitioner: any factor graph whose messages all have the `left` schedule and whose factors are either
    "variables" (left factor of their messages: MRF unaries, multicut edge factors, ...) or "higher factors" (right
    factor: pairwise, triplet, ... factors), with unary-pairwise or labeling messages (the left factor sends
    omega * theta through both).  ``part[f]`` = owning rank of variable f (ignored for higher factors).  A higher
    factor belongs to the part of its lowest-numbered variable; every cut message gets a zero ghost copy of its remote
    variable on the owner's side.  Local factor order: variables, ghosts, higher factors; messages keep the global
    insertion order; a relation survives when both ends are local (a remote variable is represented by its ghost in
    relations with the ghost's own higher factor)."""
    from . import engine as E
    part = np.asarray(part, np.int64)
    nf, nm = gm.n_factors, gm.n_messages
    ml, mr = gm.m_left.astype(np.int64), gm.m_right.astype(np.int64)
    for t in gm.mtypes:
        if t.schedule != M.SCHED_LEFT or t.kind not in (M.M_UNARY_PAIRWISE, M.M_LABELING):
            raise ValueError("partition_model: only `left`-schedule unary-pairwise / labeling messages")
    is_right = np.zeros(nf, bool); is_right[mr] = True
    is_left = np.zeros(nf, bool); is_left[ml] = True
    if np.any(is_left & is_right):
        raise ValueError("partition_model: a factor is both left and right of messages")
    if np.any(gm.f_kind[is_left] != M.F_VECTOR):
        raise ValueError("partition_model: variables must be vector factors")
    first_var = np.full(nf, nf, np.int64)
    np.minimum.at(first_var, mr, ml)
    owner = np.where(is_right, part[np.minimum(first_var, nf - 1)], part)      # of every factor
    m_owner = owner[mr]
    coff, doff = gm.const_offsets(), gm.dual_offsets()
    g_off, g_ent = E.Plan(gm).msg_lists(nm)
    pos_in_list = np.zeros(nm, np.int64)                   # position of message m in its left factor's message list
    lens = np.diff(g_off)
    within = np.arange(int(g_off[-1])) - np.repeat(g_off[:-1], lens)
    left_entry = (g_ent % 2) == 0                            # role 0: the list's factor is the message's left factor
    pos_in_list[g_ent[left_entry] // 2] = within[left_entry]

    def take(data, off, idx):
        if data is None or idx.shape[0] == 0:
            return np.zeros(0)
        lens = off[idx + 1] - off[idx]
        first = np.concatenate([[0], np.cumsum(lens)])
        return data[np.repeat(off[idx], lens) + (np.arange(int(first[-1])) - np.repeat(first[:-1], lens))]

    parts = []
    for k in range(world):
        lv = np.nonzero(~is_right & (part == k))[0]
        rk = np.nonzero(is_right & (owner == k))[0]
        mk = np.nonzero(m_owner == k)[0]                    # messages into owned higher factors, global order
        cut = part[ml[mk]] != k
        n_ghost = int(cut.sum())
        ghost_of = ml[mk][cut]                              # remote variable behind every ghost
        n_local = lv.shape[0]
        if n_local + rk.shape[0] == 0:
            raise ValueError(f"partition_model: part {k} owns no factor")
        lmap = np.full(nf, -1, np.int64)
        lmap[lv] = np.arange(n_local)
        lmap[rk] = n_local + n_ghost + np.arange(rk.shape[0])
        left_loc = lmap[ml[mk]].copy()
        left_loc[cut] = n_local + np.arange(n_ghost)
        src = np.concatenate([lv, ghost_of, rk])            # global factor every local factor copies its shape from
        dual = np.concatenate([take(gm.dual_data, doff, lv), np.zeros(int((doff[ghost_of + 1] - doff[ghost_of]).sum())),
                               take(gm.dual_data, doff, rk)])
        const = None if gm.const_data is None else take(gm.const_data, coff, rk)

        gkey = ghost_of * nf + mr[mk][cut]                  # (remote variable, its higher factor) of every ghost
        gorder = np.argsort(gkey, kind="stable")
        gkey_s = gkey[gorder]

        def ghost_lookup(v, r):
            if gkey_s.shape[0] == 0:
                return np.full(v.shape[0], -1, np.int64)
            key = v * nf + r
            i = np.minimum(np.searchsorted(gkey_s, key), gkey_s.shape[0] - 1)
            return np.where(gkey_s[i] == key, n_local + gorder[i], -1)

        def map_rel(rel):
            rel = np.asarray(rel, np.int64).reshape(-1, 2)
            a, b = rel[:, 0], rel[:, 1]
            la, lb = lmap[a].copy(), lmap[b].copy()
            la = np.where(la < 0, ghost_lookup(a, b), la)
            lb = np.where(lb < 0, ghost_lookup(b, a), lb)
            keep = (la >= 0) & (lb >= 0)
            return np.ascontiguousarray(np.stack([la[keep], lb[keep]], 1).astype(np.int32)).reshape(-1, 2)

        m = M.FlatModel(
            n_ftypes=gm.n_ftypes, ftype_computes_primal=gm.ftype_computes_primal, mtypes=gm.mtypes,
            tab_off=gm.tab_off, tab_data=gm.tab_data, tab_nleft=gm.tab_nleft,
            f_type=np.ascontiguousarray(gm.f_type[src]), f_kind=np.ascontiguousarray(gm.f_kind[src]),
            f_flags=np.ascontiguousarray(gm.f_flags[src]), f_dim0=np.ascontiguousarray(gm.f_dim0[src]),
            f_dim1=np.ascontiguousarray(gm.f_dim1[src]), const_data=const, dual_data=np.ascontiguousarray(dual),
            m_type=np.ascontiguousarray(gm.m_type[mk]), m_left=left_loc.astype(np.int32), m_right=lmap[mr[mk]].astype(np.int32),
            rel_fwd=map_rel(gm.rel_fwd), rel_bwd=map_rel(gm.rel_bwd), constant=gm.constant if k == 0 else 0.0)
        out_peer, out_ghost, out_key = _sorted_by_peer_key(part[ghost_of], (n_local + np.arange(n_ghost)).astype(np.int32), key=mk[cut])
        inm = np.nonzero((part[ml] == k) & (m_owner != k))[0]
        in_peer, in_unary, in_pos, in_key = _sorted_by_peer_key(m_owner[inm], lmap[ml[inm]].astype(np.int32), pos_in_list[inm], key=inm)
        # a higher factor may have several remote variables: the ghost passes visit its ghosts in the order a pass over
        # the remote variables themselves (by global index, cut messages in list order) would touch the factor
        ghost_order = np.lexsort((pos_in_list[mk[cut]], ghost_of))
        parts.append(LocalPart(k, world, 0, m, n_local, n_ghost, src, mk.astype(np.int64), out_peer, out_ghost, out_key,
                               in_peer, in_unary, in_key, in_pos=in_pos, key_is_msg=True, ghost_order=ghost_order))
    return parts


def graph_partition(n_vars: int, edge_i: np.ndarray, edge_j: np.ndarray, world: int, refine_rounds: int = 30,
                    imbalance: float = 0.03, seed: int = 0, method: str = "auto", return_method: bool = False):
    """k-way partition of a sparse variable graph.  ``method``:
      "metis"     METIS_PartGraphKway through `pymetis` or a `libmetis.so` found by the loader (metis_partition); an error when
                  neither is there;
      "builtin"   no METIS (not in this image; SURVEY 8e allows a built-in partitioner): a reverse Cuthill-McKee order (bandwidth
                  reducing, scipy.sparse.csgraph) cut into ``world`` contiguous chunks, then refined by balanced Kernighan-Lin /
                  label-propagation moves (``refine_partition``): every round each variable looks at the part most of its
                  neighbours live in and moves there if that cuts fewer edges and the target stays within (1 + imbalance) of the
                  mean size.  Graphs with locality get few cut edges; G(n, m) random graphs have little to exploit (any balanced
                  partition cuts most of the edges), the refinement recovers a few percent there;
      "auto"      METIS when present, else the built-in one (LPMP_PARTITIONER in the environment overrides "auto").
    Deterministic for a given method: every rank computes the same partition from the same edge list (multi-process drivers
    still compute it once, on rank 0, and broadcast it).  ``return_method``: also the name of what ran (bench.py prints it)."""
    if method == "auto":
        method = os.environ.get("LPMP_PARTITIONER", "auto")
    if method not in ("auto", "metis", "builtin"):
        raise ValueError("graph_partition: method must be auto, metis or builtin")
    edge_i = np.asarray(edge_i, np.int64); edge_j = np.asarray(edge_j, np.int64)
    metis_failed = None
    if method in ("auto", "metis") and world > 1:
        try:
            got = metis_partition(n_vars, edge_i, edge_j, world, imbalance, seed, required=method == "metis")
        except Exception as ex:                       # "auto": an installed METIS that fails must not cost the run its partition
            if method == "metis":
                raise
            got, metis_failed = None, f"{type(ex).__name__}: {ex}"
        if got is not None:
            return (got[0], got[1]) if return_method else got[0]
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    a = coo_matrix((np.ones(edge_i.shape[0], np.float32), (edge_i, edge_j)), shape=(n_vars, n_vars)).tocsr()
    a = (a + a.T).tocsr()
    order = reverse_cuthill_mckee(a, symmetric_mode=True)
    part = np.empty(n_vars, np.int64)
    part[order] = (np.arange(n_vars) * world) // n_vars
    if world > 1 and refine_rounds > 0:
        # on the planner's threads behind the C ABI (lpmp_graph_refine_partition); refine_partition below is the numpy statement
        # of the same moves (20 s against 1 at 2 M variables)
        from . import engine as E
        part = E.graph_refine_partition(n_vars, edge_i, edge_j, part, world, refine_rounds, imbalance, seed)
    how = "builtin (reverse Cuthill-McKee + balanced KL refinement)"
    if metis_failed:
        how += f"; a METIS was found but failed: {metis_failed[:200]}"
    return (part, how) if return_method else part


_METIS = {}
# where metis.h's moptions_et puts the two options set here: 5.1.x has SEED = 8, UFACTOR = 16; 5.2.x inserted NIPARTS and ONDISK
# before them (SEED = 9, UFACTOR = 17).  Slot numbers are never assumed: the layout is PROBED (below), and when neither answers the
# call is made with METIS' own defaults (options = NULL: 3 % imbalance for k-way, its fixed seed).
_METIS_OPTION_LAYOUTS = (("5.1", 8, 16), ("5.2", 9, 17))

_METIS_PROBE = (
    "import ctypes, sys, numpy as np\n"
    "L = ctypes.CDLL(sys.argv[1]); bits = int(sys.argv[2]); I = np.int32 if bits == 32 else np.int64\n"
    "seed_slot, ufactor_slot = int(sys.argv[3]), int(sys.argv[4])\n"
    "n = 64; xadj = (2 * np.arange(n + 1)).astype(I); adj = np.stack([(np.arange(n) - 1) % n, (np.arange(n) + 1) % n], 1).reshape(-1).astype(I)\n"
    "nv = np.array([n], I); nc = np.array([1], I); k = np.array([2], I); obj = np.zeros(1, I); part = np.full(n, -1, I)\n"
    "p = lambda a: a.ctypes.data_as(ctypes.c_void_p)\n"
    "opts = None\n"
    "if seed_slot >= 0:\n"
    "    o = np.zeros(64, I); L.METIS_SetDefaultOptions(p(o)); o[seed_slot] = 0; o[ufactor_slot] = 30; opts = p(o)\n"
    "rc = L.METIS_PartGraphKway(p(nv), p(nc), p(xadj), p(adj), None, None, None, p(k), None, None, opts, p(obj), p(part))\n"
    "ok = rc == 1 and set(part.tolist()) == {0, 1} and abs(int((part == 0).sum()) - 32) <= 8 and 0 < int(obj[0]) <= 16\n"
    "sys.exit(0 if ok else 1)\n")


def _metis_library():
    """a libmetis the loader finds (LPMP_METIS_LIB names one explicitly), checked once in CHILD processes on a 64-ring: idx_t may be
    32 or 64 bits wide depending on how the library was built, and the slots of the option array differ between METIS 5.1 and 5.2 —
    a wrong guess must neither take this process down nor hand METIS a zero where it wants a count.  The probe makes the real
    call's kind of call: first without options (index width), then with the seed / imbalance options in each known layout, the
    values the default call sets (seed 0, UFACTOR 30: a 5.2 library reads a 5.1 layout's seed slot as NCUTS = 0 and refuses; a 5.1
    library reads a 5.2 layout's imbalance slot as NUMBERING = 30 and refuses).  Returns (ctypes library, idx dtype, name,
    (layout name, seed slot, ufactor slot) or None when options cannot be set safely) or None."""
    if "lib" in _METIS:
        return _METIS["lib"]
    import ctypes, ctypes.util, subprocess, sys
    _METIS["lib"] = None
    name = os.environ.get("LPMP_METIS_LIB") or ctypes.util.find_library("metis")
    if not name:
        return None

    def probe(bits, seed_slot, ufactor_slot):
        try:
            return subprocess.run([sys.executable, "-c", _METIS_PROBE, name, str(bits), str(seed_slot), str(ufactor_slot)], timeout=60,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0
        except Exception:
            return False
    for bits in (32, 64):
        if probe(bits, -1, -1):
            ok = [lay for lay in _METIS_OPTION_LAYOUTS if probe(bits, lay[1], lay[2])]
            # exactly one layout may answer; a library that accepts both (it ignores options, or checks nothing) gets none set
            _METIS["lib"] = (ctypes.CDLL(name), np.int32 if bits == 32 else np.int64, name, ok[0] if len(ok) == 1 else None)
            break
    return _METIS["lib"]


def metis_partition(n_vars: int, edge_i, edge_j, world: int, imbalance: float = 0.03, seed: int = 0, required: bool = False):
    """METIS k-way partition (edge cut objective) of the variable graph, or None when no METIS is installed (``required``: an
    error instead).  Returns (part[int64], name of what ran).  `pymetis` first, then libmetis through ctypes.  Seed and imbalance
    are handed over only in an option layout the probe has seen this library accept (_metis_library); otherwise the call runs with
    METIS' defaults and the name says so."""
    edge_i = np.asarray(edge_i, np.int64); edge_j = np.asarray(edge_j, np.int64)
    from scipy.sparse import coo_matrix
    def csr():
        a = coo_matrix((np.ones(2 * edge_i.shape[0], np.int8), (np.concatenate([edge_i, edge_j]), np.concatenate([edge_j, edge_i]))), shape=(n_vars, n_vars)).tocsr()
        a.sum_duplicates()
        a.setdiag(0); a.eliminate_zeros()
        return a
    try:
        import pymetis
    except ImportError:
        pymetis = None
    if pymetis is not None:
        a = csr()
        _, membership = pymetis.part_graph(world, xadj=a.indptr.tolist(), adjncy=a.indices.tolist())
        return np.asarray(membership, np.int64), "metis (pymetis %s)" % getattr(pymetis, "version", "")
    lib = _metis_library()
    if lib is None:
        if required:
            raise RuntimeError("graph_partition(method='metis'): neither pymetis nor a loadable libmetis (LPMP_METIS_LIB) is installed")
        return None
    import ctypes
    L, I, name, layout = lib
    a = csr()
    if I == np.int32 and (a.indices.shape[0] >= 2**31 or n_vars >= 2**31):
        raise RuntimeError("metis_partition: this libmetis has 32-bit indices, the graph needs 64")
    xadj, adj = np.ascontiguousarray(a.indptr, I), np.ascontiguousarray(a.indices, I)
    nv, nc, k, obj = np.array([n_vars], I), np.array([1], I), np.array([world], I), np.zeros(1, I)
    part = np.zeros(n_vars, I)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    opts = None
    if layout is not None:
        opts = np.zeros(64, I)
        L.METIS_SetDefaultOptions(p(opts))
        opts[layout[1]] = seed                                       # METIS_OPTION_SEED
        opts[layout[2]] = max(1, int(round(1000 * imbalance)))       # METIS_OPTION_UFACTOR: allowed imbalance in 1/1000
    rc = L.METIS_PartGraphKway(p(nv), p(nc), p(xadj), p(adj), None, None, None, p(k), None, None, None if opts is None else p(opts), p(obj), p(part))
    if rc != 1:
        raise RuntimeError("METIS_PartGraphKway failed with code %d" % rc)
    part = part.astype(np.int64)
    if part.min() < 0 or part.max() >= world or np.bincount(part, minlength=world).min() == 0:
        raise RuntimeError("METIS_PartGraphKway returned a partition with an empty or out-of-range part")
    how = "options in the %s layout" % layout[0] if layout is not None else "METIS' default options"
    return part, "metis (%s, %d-bit idx_t, %s)" % (os.path.basename(name), 32 if I == np.int32 else 64, how)


def load_partition_file(path: str, n_vars: int, world: int) -> np.ndarray:
    """a partition handed in as a file: n_vars entries (variable -> part) as raw little-endian int64 or int32 (`*.bin`), as text
    (one integer per line / whitespace separated: the format METIS' own gpmetis writes), or a numpy `*.npy`.  Checked: length,
    range, no empty part (every rank takes part in every exchange)."""
    if path.endswith(".npy"):
        part = np.load(path)
    elif path.endswith(".bin"):
        raw = np.fromfile(path, np.uint8)
        if raw.shape[0] == 8 * n_vars:
            part = raw.view("<i8")
        elif raw.shape[0] == 4 * n_vars:
            part = raw.view("<i4")
        else:
            raise ValueError(f"{path}: {raw.shape[0]} bytes is neither {n_vars} int64 nor {n_vars} int32 entries")
    else:
        part = np.loadtxt(path, dtype=np.int64, ndmin=1)
    part = np.ascontiguousarray(part, np.int64).reshape(-1)
    if part.shape[0] != n_vars:
        raise ValueError(f"{path}: {part.shape[0]} entries, the model has {n_vars} variables")
    if part.min() < 0 or part.max() >= world:
        raise ValueError(f"{path}: parts must lie in [0, {world})")
    if np.bincount(part, minlength=world).min() == 0:
        raise ValueError(f"{path}: a part without variables")
    return part


def refine_partition(adj, part: np.ndarray, world: int, rounds: int = 30, imbalance: float = 0.03, seed: int = 0) -> np.ndarray:
    """balanced label-propagation / Kernighan-Lin style refinement of a k-way partition (``adj``: symmetric scipy CSR
    adjacency with edge multiplicities).  Per round: gain of moving v to the part holding most of its neighbours;
    a pseudo-random half of the variables with positive gain is considered (neighbours moving at once could undo each
    other), best gains first, as long as the target part has room."""
    adj = adj.tocsr()
    n = adj.shape[0]
    part = part.copy()
    cap = int(np.ceil(n / world * (1.0 + imbalance)))
    idx = np.arange(n)
    row_w = np.repeat(idx, np.diff(adj.indptr)) * world               # directed edges v -> u: slot of v's counters
    col, mult = adj.indices.astype(np.int64), adj.data.astype(np.float64)
    with np.errstate(over="ignore"):
        coin = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)) >> np.uint64(40)
    for r in range(rounds):
        cnt = np.bincount(row_w + part[col], weights=mult, minlength=n * world).reshape(n, world)   # neighbours of v in every part
        cur = cnt[idx, part]
        best = np.argmax(cnt, axis=1)
        gain = cnt[idx, best] - cur
        cand = np.nonzero((gain > 0) & (((coin >> np.uint64(r % 20)) & np.uint64(1)) == (r & 1)))[0]
        if cand.size == 0:
            if r > 2:
                break
            continue
        size = np.bincount(part, minlength=world)
        moved = 0
        order = cand[np.argsort(-gain[cand], kind="stable")]
        tgt = best[order]
        for t in range(world):                                           # per target: the best moves that fit
            sel = order[tgt == t]
            room = cap - size[t]
            if room <= 0 or sel.size == 0:
                continue
            sel = sel[:room]
            np.subtract.at(size, part[sel], 1)
            size[t] += sel.size
            part[sel] = t
            moved += sel.size
        if moved == 0 and r > 2:
            break
    return part


def graph_partition_model(gm: M.FlatModel, world: int, method: str = "auto", return_method: bool = False):
    """``part`` for partition_model: the variables of a factor graph (left factors of its messages) split by
    graph_partition on the graph that links the variables of every higher factor in a chain; entries of higher
    factors are unused (0).  ``method`` / ``return_method`` as graph_partition."""
    ml, mr = gm.m_left.astype(np.int64), gm.m_right.astype(np.int64)
    order = np.lexsort((ml, mr))                              # messages grouped by higher factor
    a, b, same = ml[order][:-1], ml[order][1:], mr[order][:-1] == mr[order][1:]
    is_right = np.zeros(gm.n_factors, bool); is_right[mr] = True
    var = np.nonzero(~is_right)[0]
    rank_of = np.full(gm.n_factors, -1, np.int64); rank_of[var] = np.arange(var.shape[0])
    ei, ej = rank_of[a[same]], rank_of[b[same]]
    keep = ei != ej
    p, how = graph_partition(var.shape[0], np.minimum(ei, ej)[keep], np.maximum(ei, ej)[keep], world, method=method, return_method=True)
    part = np.zeros(gm.n_factors, np.int64)
    part[var] = p
    return (part, how) if return_method else part


# ---- row-strip grids: closed-form local parts (no global model is ever materialised) ------------
def strip_sizes(H: int, W: int):
    return H * W, H * (W - 1) + W * (H - 1)


def strip_global_edges(H: int, W: int, world: int, order: str):
    """Global enumeration: strip-major variables (local order inside a strip); per strip its internal
    edges (synthetic.grid_edges order) then the W cut edges to the next strip (column order)."""
    n_loc, e_int = strip_sizes(H, W)
    var = S.grid_variable_order(H, W, order).reshape(-1)
    a, b = S.grid_edges(H, W)
    ei, ej, seg = [], [], []
    for k in range(world):
        va, vb = var[a] + k * n_loc, var[b] + k * n_loc
        ei.append(np.minimum(va, vb)); ej.append(np.maximum(va, vb))
        if k < world - 1:
            cols = np.arange(W)
            ei.append(k * n_loc + var[(H - 1) * W + cols]); ej.append((k + 1) * n_loc + var[cols])
    return np.concatenate(ei), np.concatenate(ej)


def strip_costs(H, W, L, world, pairwise, seed):
    """Host arrays of the global cost streams (tests only; sizes grow with world)."""
    n_loc, e_int = strip_sizes(H, W)
    n_vars = world * n_loc
    n_edges = world * e_int + (world - 1) * W
    un = S.u01(n_vars * L, seed, 0)
    if pairwise == "dense":
        return un, S.u01(n_edges * L * L, seed, n_vars * L), None
    return un, None, S.u01(n_edges, seed, n_vars * L)


def strip_local_part(H: int, W: int, L: int, pairwise: str, order: str, rank: int, world: int, seed: int,
                     device_const: bool = False) -> LocalPart:
    """The same LocalPart that partition_mrf gives for the strip partition of the global grid, built
    from closed-form index arithmetic on this rank's strip only."""
    n_loc, e_int = strip_sizes(H, W)
    n_vars = world * n_loc
    var = S.grid_variable_order(H, W, order).reshape(-1)
    a, b = S.grid_edges(H, W)
    va, vb = var[a], var[b]
    li, lj = np.minimum(va, vb), np.maximum(va, vb)
    has_down = rank < world - 1
    has_up = rank > 0
    n_ghost = W if has_down else 0
    cols = np.arange(W)
    if has_down:
        li = np.concatenate([li, var[(H - 1) * W + cols]])
        lj = np.concatenate([lj, n_loc + cols])
    n_own = li.shape[0]
    e_first = rank * (e_int + W)                               # global id of this strip's first edge
    esz = L * L if pairwise == "dense" else 1
    un_first = rank * n_loc * L
    pw_first = n_vars * L + e_first * esz
    unaries = None if device_const else np.concatenate([S.u01(n_loc * L, seed, un_first).reshape(n_loc, L), np.zeros((n_ghost, L))])
    if device_const:
        un_host = np.zeros((n_loc + n_ghost) * L)
        if pairwise == "dense":
            m = S.mrf_model(n_loc + n_ghost, L, li, lj, un_host, device_const=True)
        else:
            m = S.mrf_model(n_loc + n_ghost, L, li, lj, un_host, potts=np.zeros(n_own))
        const_fill = [(0, n_own * esz, seed, pw_first)]
        dual_fill = [(0, n_loc * L, seed, un_first)]
    else:
        const_fill = dual_fill = None
        if pairwise == "dense":
            m = S.mrf_model(n_loc + n_ghost, L, li, lj, unaries, tables=S.u01(n_own * esz, seed, pw_first))
        else:
            m = S.mrf_model(n_loc + n_ghost, L, li, lj, unaries, potts=S.u01(n_own, seed, pw_first))
    ge = e_first + np.arange(n_own)
    l2g = np.concatenate([rank * n_loc + np.arange(n_loc), (rank + 1) * n_loc + var[cols] if has_down else np.zeros(0, np.int64),
                          n_vars + ge]).astype(np.int64)
    lm2g = np.stack([2 * ge, 2 * ge + 1], 1).reshape(-1).astype(np.int64)
    out_peer = np.full(n_ghost, rank + 1, np.int64)
    out_ghost = (n_loc + cols[:n_ghost]).astype(np.int32)
    out_key = e_first + e_int + cols[:n_ghost]
    if has_up:
        in_peer = np.full(W, rank - 1, np.int64)
        in_unary = var[cols].astype(np.int32)
        in_key = (rank - 1) * (e_int + W) + e_int + cols
    else:
        in_peer = np.zeros(0, np.int64); in_unary = np.zeros(0, np.int32); in_key = np.zeros(0, np.int64)
    return LocalPart(rank, world, L, m, n_loc, n_ghost, l2g, lm2g, out_peer, out_ghost, out_key, in_peer, in_unary,
                     in_key, const_fill, dual_fill)


# ---- communication ---------------------------------------------------------------------------------
class DistComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests)."""

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        # gloo: CPU tests, and GPU smoke runs without RCCL (ranks sharing one device) — device rows are staged through the host.
        # LPMP_DIST_DEVICE_COLLECTIVES=1 hands DEVICE tensors to the backend whatever it is (gloo's own CUDA collectives): the code
        # path of an RCCL run — device all-to-all-v, the asynchronous exchange_begin / exchange_end — with real data between two
        # processes on a box where RCCL cannot run two ranks (tests/test_round6_gpu.py)
        self.stage_cpu = dist.get_backend() == "gloo" and not os.environ.get("LPMP_DIST_DEVICE_COLLECTIVES")

    def exchange(self, send, send_counts, recv_counts):
        """rows of ``send`` are grouped by destination rank (send_counts[r] rows each); returns the rows
        received, grouped by source rank.  One all-to-all-v."""
        if self.stage_cpu and send.is_cuda:
            dev = send.device
            return self.exchange(send.cpu(), send_counts, recv_counts).to(dev)
        # zero-row views of 1-row buffers: a rank without cut edges toward anybody still takes part in the collective,
        # and its tensors keep a valid device pointer
        n_out = int(sum(recv_counts))
        tail = tuple(send.shape[1:])
        out = send.new_empty((max(n_out, 1),) + tail)[:n_out]
        src = send.contiguous() if send.shape[0] > 0 else send.new_empty((1,) + tail)[:0]
        self.dist.all_to_all_single(out, src, output_split_sizes=list(map(int, recv_counts)),
                                    input_split_sizes=list(map(int, send_counts)))
        return out

    def exchange_begin(self, send, send_counts, recv_counts):
        """post the all-to-all-v and return at once (RCCL: the collective runs on the process group's own stream, ordered behind
        what the current stream has been given so far; the caller keeps ``send`` untouched until exchange_end).  Staged through the
        CPU (gloo) there is nothing to overlap: the exchange happens here."""
        if self.stage_cpu or not send.is_cuda:
            return ("done", self.exchange(send, send_counts, recv_counts))
        n_out = int(sum(recv_counts))
        tail = tuple(send.shape[1:])
        out = send.new_empty((max(n_out, 1),) + tail)[:n_out]
        src = send.contiguous() if send.shape[0] > 0 else send.new_empty((1,) + tail)[:0]
        work = self.dist.all_to_all_single(out, src, output_split_sizes=list(map(int, recv_counts)),
                                           input_split_sizes=list(map(int, send_counts)), async_op=True)
        return ("posted", out, work, src)

    def exchange_end(self, pending):
        """the rows exchange_begin's transfer delivers; the current stream waits for the collective (no host synchronisation)"""
        if pending[0] == "done":
            return pending[1]
        pending[2].wait()
        return pending[1]

    def all_reduce_sum(self, x: float) -> float:
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.stage_cpu else self._dev)
        self.dist.all_reduce(t)
        return float(t.item())

    _dev = "cpu"


class ExchangeProbe:
    """Where an N-rank pass spends its time, per rank: every exchange (pack -> collective -> unpack) is bracketed by a pair of
    events on the stream the engine works on, the whole call by another pair; compute = total - exchanges.  The span of an
    exchange INCLUDES waiting for the slowest peer (the collective synchronises the ranks): the rank with the largest compute
    time is the one the others wait for.  Used in an untimed repetition of the timed passes (bench.py), never inside them.
    HIP events when the duals live on the device, the host clock for stand-in engines on the CPU (tests)."""

    def __init__(self, torch, on_device: bool):
        self.torch, self.on_device = torch, bool(on_device)
        self.spans, self.bytes_out, self.bytes_in, self.n_exchanges = [], 0, 0, 0
        self.post_spans = []
        self._t0 = self._t1 = self._b = self._p = None

    def _mark(self):
        if self.on_device:
            ev = self.torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        import time
        return time.perf_counter()

    def start(self):
        self._t0 = self._mark()

    def stop(self):
        self._t1 = self._mark()

    def begin_exchange(self):
        self._b = self._mark()

    def end_exchange(self, doubles_out: int, doubles_in: int):
        self.spans.append((self._b, self._mark()))
        self.bytes_out += 8 * int(doubles_out); self.bytes_in += 8 * int(doubles_in); self.n_exchanges += 1

    def begin_post(self):
        """overlapped program: pack + copy + posting the collective (halo_begin) — exchange work on the engine's stream that is not
        hidden behind anything, booked apart from both the compute time and the awaited rest of the exchange (halo_end)"""
        self._p = self._mark()

    def end_post(self):
        self.post_spans.append((self._p, self._mark()))

    def _ms(self, a, b) -> float:
        return float(a.elapsed_time(b)) if self.on_device else (b - a) * 1e3

    def result(self, n_passes: int) -> dict:
        if self.on_device:
            self.torch.cuda.synchronize()
        total = self._ms(self._t0, self._t1)
        exch = sum(self._ms(a, b) for a, b in self.spans)
        post = sum(self._ms(a, b) for a, b in self.post_spans)
        n = max(1, int(n_passes))
        # exchange_ms = everything of the exchanges that sits on the engine's stream: the posts of an overlapped program (pack,
        # copy, posting) + the awaited spans — so compute_ms (and scaling_model's t_run) means the same with and without
        # --overlap-exchange; exchange_post_ms says how much of it was the posts (0 for the plain program)
        return {"total_ms_per_pass": total / n, "exchange_ms_per_pass": (exch + post) / n, "exchange_post_ms_per_pass": post / n,
                "compute_ms_per_pass": (total - exch - post) / n,
                "exchanges_per_pass": self.n_exchanges / n, "exchange_bytes_out_per_pass": self.bytes_out / n,
                "exchange_bytes_in_per_pass": self.bytes_in / n}


def gather_rank_stats(dist, torch, comm, stats: dict) -> dict:
    """every rank's ExchangeProbe.result() (+ whatever else the driver put into ``stats``: numbers only) on every rank, as
    {"per_rank": {key: [v_0 ... v_{N-1}]}, "max": {...}, "mean": {...}, "slowest_rank": argmax compute_ms_per_pass}"""
    keys = sorted(stats)
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    dev = "cpu" if (comm is None or comm.stage_cpu) else comm._dev
    t = torch.zeros(world, len(keys), dtype=torch.float64, device=dev)
    rank = dist.get_rank() if world > 1 or (dist is not None and dist.is_initialized()) else 0
    t[rank] = torch.tensor([float(stats[k]) for k in keys], dtype=torch.float64, device=dev)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t)
    t = t.cpu()
    per = {k: [float(x) for x in t[:, i]] for i, k in enumerate(keys)}
    out = {"per_rank": per, "max": {k: max(v) for k, v in per.items()}, "mean": {k: sum(v) / len(v) for k, v in per.items()}}
    if "compute_ms_per_pass" in per:
        c = per["compute_ms_per_pass"]
        out["slowest_rank"] = int(max(range(len(c)), key=lambda r: c[r]))
    return out


class LocalComm:
    """In-process stand-in used to run several parts on ONE device (tests on the 1-GPU box): the parts'
    sweeps are stepped in lockstep by ``run_lockstep`` and exchange through this mailbox."""

    def __init__(self, world):
        self.world = world
        self.box = {}


BOUNDARY_SHARE = 0.375
# Part of its send weight a variable with k cut messages and d local ones keeps back in the main sweeps (GraphSweep):
# every send row is scaled by 1 - BOUNDARY_RESERVE * k / (d + k), so that what the boundary step then shares out is not
# only what it just pulled in.  G(20 000, 100 000), 16 labels, 8 passes, gap to the unpartitioned bound without -> with
# 0.75 (boundary step after each sweep): 2 parts (31 % cut) 1.10 -> 1.06 %, 4 parts (49 %) 2.20 -> 1.69 %, 8 parts (61 %) 3.37 -> 2.33 %;
# with the boundary step BEFORE each sweep (what program() does): 0.97 / 1.54 / 2.15 % (tests/gap_probe.py)
BOUNDARY_RESERVE = 0.75


# ---- the sweep ---------------------------------------------------------------------------------------
class PartitionedSweep:
    """One part of the partitioned sweep.  ``engine`` is an lp_mp_amd.engine.Engine (or, in CPU tests, an
    object with the same methods backed by the oracle); ``theta_all`` is a torch view of the engine's dual
    buffer, so the boundary arithmetic is done in place with torch ops on the engine's stream."""

    def __init__(self, torch, part: LocalPart, engine, dual_tensor, mode: int = M.REPAM_ANISOTROPIC,
                 omega_b: Optional[float] = None, boundary_every: str = "sweep", reserve: float = 0.0):
        """boundary_every: "sweep" — one boundary step after each directional main sweep (tighter bound per pass:
        on a random graph with 69 % cut edges the gap to the unpartitioned sweep after 8 passes is 3.6 % against
        18 % for "pass"; on 4 strips of 32 rows 0.07 % against 0.31 %); "pass" — one after the forward+backward
        main sweeps, which then run as one fused schedule (every cut message is received once and sent once per
        pass, like every other message under anisotropic weights) — what bench.py uses for row strips."""
        assert boundary_every in ("pass", "sweep")
        self.boundary_every = boundary_every
        self.torch, self.part, self.engine, self.mode = torch, part, engine, mode
        # send weight of the boundary update of a non-owner endpoint with k cut messages: BOUNDARY_SHARE / k each (the
        # row must sum to <= 1, reference omega_valid LP_MP.h:1008-1014) unless given.  The share is a tuning knob of the
        # partitioned schedule, not of the reference: on the C4-shaped graph (20 000 nodes, 2 parts, 31 % of the edges
        # cut, 8 passes) the gap to the unpartitioned bound is 2.2 % with 1 / (k + 1) each, 1.5 % with 0.2 / k, 1.1 % with
        # 0.35 ... 0.4 / k, 1.3 % with 0.5 / k, 2.7 % with 0.7 / k (what the variable keeps feeds its own next sweep)
        k_cut = np.bincount(part.in_unary, minlength=part.n_local + part.n_ghost)[part.in_unary] if part.in_unary.size else np.zeros(0)
        self.in_omega = (BOUNDARY_SHARE / np.maximum(k_cut, 1.0)) if omega_b is None else np.full(part.in_unary.shape[0], float(omega_b))
        if np.any(np.bincount(part.in_unary, weights=self.in_omega) > 1.0 + 1e-8) if part.in_unary.size else False:
            raise ValueError("boundary send weights of one unary sum to more than 1")
        p = part
        n_vec = p.n_local + p.n_ghost
        self.dual = dual_tensor
        if p.L > 0:                                          # MRF parts: vectors first, one label count
            self.theta = dual_tensor[: n_vec * p.L].view(n_vec, p.L)
        # the boundary arithmetic works on flat element indices of the dual buffer (factors may differ in size)
        doff = p.model.dual_offsets()
        dim = p.model.f_dim0.astype(np.int64)

        def flat(factors):
            factors = np.asarray(factors, np.int64)
            lens = dim[factors]
            first = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            return np.repeat(doff[factors], lens) + (np.arange(int(first[-1])) - np.repeat(first[:-1], lens)), lens, first
        dev = dual_tensor.device
        plan = engine.plan
        ghost = np.zeros(p.model.n_factors, bool)
        ghost[p.n_local: n_vec] = True
        # 1. main sweeps: the part's own update lists and anisotropic rows, ghost factors dropped
        self.main = []
        self.main_rows = []
        for d in (M.FORWARD, M.BACKWARD):
            upd = plan.update_order(d)
            om_off, om = plan.omega(d, mode)
            mk_off, mk = plan.mask(d, mode)
            keep = ~ghost[upd]
            rows = _select_rows(upd, om_off, om, mk_off, mk, keep)
            scale = getattr(part, "main_send_scale", None)      # (tests/gap_probe.py sets it for experiments)
            if scale is None and d == M.BACKWARD:
                scale = getattr(part, "main_send_scale_backward", None)
            if scale is None and reserve > 0.0 and part.in_unary.size:
                nf = part.model.n_factors
                deg = np.bincount(part.model.m_left, minlength=nf).astype(np.float64)
                kc = np.bincount(part.in_unary, minlength=nf).astype(np.float64)
                scale = np.ones(nf)
                has = kc > 0
                scale[has] = 1.0 - reserve * kc[has] / (deg[has] + kc[has])
            if scale is not None:
                lens = rows[1][1:] - rows[1][:-1]
                rows = (rows[0], rows[1], rows[2] * np.repeat(scale[rows[0]], lens), rows[3], rows[4])
            self.main_rows.append(rows)
            self.main.append(engine.schedule_create(*rows))
        # forward then backward main sweep as one (fused) sequence, and its steady-state variants: the first level
        # of the forward sweep that touches nothing the boundary step touches (X) commutes with the boundary step,
        # so it is run at the END of the previous pass, where it fuses with the backward sweep's last level
        # (same factors): pass k = [forward minus X, backward, X of pass k+1].
        f0, f1 = self.main_rows
        self.rows = {"F": f0, "B": f1, "FB": _cat_rows(f0, f1)}
        self.sched = {"F": self.main[0], "B": self.main[1]}
        if boundary_every == "pass":
            self.sched["FB"] = engine.schedule_create(*self.rows["FB"], fuse=True)
            lev = plan.update_levels(M.FORWARD, mode)[~ghost[plan.update_order(M.FORWARD)]]
            touched = np.zeros(p.model.n_factors, bool)            # what the boundary step reads or writes
            touched[p.n_local: n_vec] = True
            touched[p.in_unary] = True
            m_l, m_r = p.model.m_left, p.model.m_right
            cut_pw = np.zeros(p.model.n_factors, bool)
            cut_pw[m_r[ghost[m_l]]] = True                            # pairwise factors adjacent to a ghost
            touched |= cut_pw
            near = np.zeros(p.model.n_factors, bool)                  # factors with a message to a touched factor
            near[m_l[touched[m_r]]] = True
            near[m_r[touched[m_l]]] = True
            x = (lev == 1) & ~touched[f0[0]] & ~near[f0[0]]
            if x.any() and not x.all():
                fx, frest = _subset_rows(f0, x), _subset_rows(f0, ~x)
                self.rows["first"] = _cat_rows(f0, f1, fx)
                self.rows["mid"] = _cat_rows(frest, f1, fx)
                self.rows["last"] = _cat_rows(frest, f1)
                for k in ("first", "mid", "last"):
                    self.sched[k] = engine.schedule_create(*self.rows[k], fuse=True)
        # 2. boundary passes on the ghosts: every ghost has exactly one message (side 1 of its cut edge)
        g = np.arange(p.n_local, n_vec, dtype=np.int32)
        if p.ghost_order is not None:
            g = np.ascontiguousarray(g[p.ghost_order])
        ones_off = np.arange(g.shape[0] + 1, dtype=np.int64)
        self.ghost_rows_recv = (g, ones_off, np.zeros(g.shape[0]), ones_off, np.ones(g.shape[0], np.uint8))
        self.ghost_rows_send = (g, ones_off, np.ones(g.shape[0]), ones_off, np.zeros(g.shape[0], np.uint8))
        self.ghost_recv = engine.schedule_create(*self.ghost_rows_recv)
        self.ghost_send = engine.schedule_create(*self.ghost_rows_send)
        # 3. exchange plan (counts in doubles)
        out_e, out_len, _ = flat(p.out_ghost)
        in_e, in_len, in_first = flat(p.in_unary)
        self.out_counts = np.bincount(p.out_peer, weights=out_len, minlength=p.world).astype(np.int64)
        self.in_counts = np.bincount(p.in_peer, weights=in_len, minlength=p.world).astype(np.int64)
        self.out_elems_t = torch.from_numpy(out_e).to(dev)
        self.n_in_elems = int(in_first[-1])
        # rounds: a non-owner factor with several cut messages receives / sends them in its message-list order
        # (MRF parts: side-1 messages in LIFO storage => descending global edge id, reference
        # factors_messages.hxx:2030-2041; general parts carry the position explicitly)
        self.rounds = []
        if p.in_unary.shape[0]:
            pos = p.in_pos if p.in_pos is not None else -p.in_key
            order = np.lexsort((pos, p.in_unary))
            u_sorted = p.in_unary[order]
            first = np.r_[True, u_sorted[1:] != u_sorted[:-1]]
            start = np.maximum.accumulate(np.where(first, np.arange(order.shape[0]), 0))
            rnd = np.arange(order.shape[0]) - start
            for r in range(int(rnd.max()) + 1):
                sel = order[rnd == r]
                tgt = flat(p.in_unary[sel])[0]
                src = np.repeat(in_first[sel], in_len[sel]) + (np.arange(int(in_len[sel].sum())) -
                                                              np.repeat(np.concatenate([[0], np.cumsum(in_len[sel])[:-1]]), in_len[sel]))
                self.rounds.append((torch.from_numpy(tgt).to(dev), torch.from_numpy(src.astype(np.int64)).to(dev)))
            self.in_elems_t = torch.from_numpy(in_e).to(dev)
            self.in_omega_t = torch.from_numpy(np.repeat(self.in_omega, in_len)).to(dev)
        # the boundary arithmetic as device kernels behind the C ABI (csrc/boundary.hip) when the engine is the HIP
        # engine; the torch index ops below remain for engine stand-ins (CPU tests on the oracle)
        self.bd = None
        if hasattr(engine, "boundary_create") and dual_tensor.is_cuda:
            in_order = np.zeros(0, np.int64)
            if p.in_unary.shape[0]:
                pos = p.in_pos if p.in_pos is not None else -p.in_key
                in_order = np.lexsort((pos, p.in_unary))
            self.bd = engine.boundary_create(doff[np.asarray(p.out_ghost, np.int64)], dim[np.asarray(p.out_ghost, np.int64)],
                                             doff[np.asarray(p.in_unary, np.int64)], dim[np.asarray(p.in_unary, np.int64)],
                                             self.in_omega, in_order)
            n_out, n_in = engine.boundary_sizes(self.bd)
            assert n_out == int(self.out_counts.sum()) and n_in == int(self.in_counts.sum())
            self._send = dual_tensor.new_empty(max(n_out, 1))[:n_out]
            self._reply = dual_tensor.new_empty(max(n_in, 1))[:n_in]
        self.info = [engine.schedule_info(s) for s in self.main]
        self.info_ghost = [engine.schedule_info(self.ghost_recv), engine.schedule_info(self.ghost_send)]

    # -- the steps of n passes: ("run", key of self.rows / self.sched) and ("boundary",) ------------------------
    def program(self, n: int):
        if self.boundary_every == "sweep":
            # the boundary step BEFORE each directional sweep: what it pulls in is sent on by the sweep that follows
            # (after the sweeps instead: gap 1.06 % instead of 0.97 % on the C4-shaped graph in 2 parts)
            return [step for _ in range(n) for step in (("boundary",), ("run", "F"), ("boundary",), ("run", "B"))]
        if n >= 2 and "mid" in self.sched:
            keys = ["first"] + ["mid"] * (n - 2) + ["last"]
        else:
            keys = ["FB"] * n
        return [step for k in keys for step in (("run", k), ("boundary",))]

    def run(self, key):
        self.engine.schedule_run(self.sched[key])

    def boundary_pack(self):
        """owner: ghost <- min-marginal toward the remote variable; returns the doubles to ship (by peer, key)."""
        if self.part.n_ghost == 0:
            return self.dual.new_zeros((0,))
        self.engine.schedule_run(self.ghost_recv)
        if self.bd is not None:
            self.engine.boundary_pack(self.bd, self._send.data_ptr())
            return self._send
        send = self.dual[self.out_elems_t]
        self.dual[self.out_elems_t] = 0.0
        return send

    def boundary_reply(self, recv):
        """non-owner: theta_j += delta (message-list order), delta' = omega_b * theta_j, theta_j -= delta'."""
        if not self.rounds:
            return recv.new_zeros((0,))
        if self.bd is not None:
            recv = recv.contiguous()
            self.engine.boundary_reply(self.bd, recv.data_ptr(), self._reply.data_ptr())
            return self._reply
        for tgt, src in self.rounds:
            self.dual[tgt] += recv[src]
        reply = self.in_omega_t * self.dual[self.in_elems_t]
        for tgt, src in self.rounds:
            self.dual[tgt] -= reply[src]
        return reply

    def boundary_fold(self, recv):
        """owner: ghost <- delta'; a weight-1 send folds it into the cut edge's pairwise factor."""
        if self.part.n_ghost == 0:
            return
        if self.bd is not None:
            recv = recv.contiguous()
            self.engine.boundary_fold(self.bd, recv.data_ptr())
        else:
            self.dual[self.out_elems_t] = recv
        self.engine.schedule_run(self.ghost_send)

    # -- stand-alone driver over a DistComm -------------------------------------------------------------
    def boundary_step(self, comm, probe=None):
        # (the ghost receive / ghost send schedules inside pack and fold are part of the boundary step: what a probe brackets)
        if probe is not None:
            probe.begin_exchange()
        send = self.boundary_pack()
        recv = comm.exchange(send, self.out_counts, self.in_counts)
        reply = self.boundary_reply(recv)
        back = comm.exchange(reply, self.in_counts, self.out_counts)
        self.boundary_fold(back)
        if probe is not None:
            n = int(self.out_counts.sum()) + int(self.in_counts.sum())
            probe.end_exchange(n, n); probe.n_exchanges += 1          # two all-to-alls per boundary step

    def compute_pass(self, comm, n=1, probe=None):
        if probe is not None:
            probe.start()
        for step in self.program(n):
            if step[0] == "run":
                self.run(step[1])
            else:
                self.boundary_step(comm, probe)
        if probe is not None:
            probe.stop()

    def exchange_counts(self):
        """split sizes of this part's largest exchange (doubles per peer rank): what a self test of the collective uses"""
        return self.out_counts, self.in_counts

    def local_lower_bound(self):
        if hasattr(self.engine, "invalidate_lower_bounds"):
            self.engine.invalidate_lower_bounds()   # the boundary step edits theta with torch ops
        return self.engine.lower_bound()

    def updates_per_pass(self):
        """executed receives + sends per pass on this part: main sweeps + 2 boundary steps (1 receive and
        1 send per cut edge each, counted on the non-owner side where the reference would execute them)."""
        n = sum(i["n_receives"] + i["n_sends"] for i in self.info)
        steps = 2 if self.boundary_every == "sweep" else 1
        return n + steps * 2 * int(self.part.in_unary.shape[0])

    def bytes_per_pass(self):
        steps = 2 if self.boundary_every == "sweep" else 1
        return sum(i["algorithmic_bytes"] for i in self.info) + steps * sum(i["algorithmic_bytes"] for i in self.info_ghost)


def _cat_rows(*rows):
    """concatenate (factors, om_off, om, mk_off, mk) row sets into one sequence"""
    f = np.concatenate([r[0] for r in rows])
    def offs(k):
        out, base = [np.zeros(1, np.int64)], 0
        for r in rows:
            out.append(base + r[k][1:]); base += int(r[k][-1])
        return np.concatenate(out).astype(np.int64)
    return (f, offs(1), np.concatenate([r[2] for r in rows]), offs(3), np.concatenate([r[4] for r in rows]))


def _subset_rows(rows, keep):
    return _select_rows(rows[0], rows[1], rows[2], rows[3], rows[4], keep)


def _select_rows(upd, om_off, om, mk_off, mk, keep):
    """CSR rows of the kept factors."""
    idx = np.nonzero(keep)[0]
    f = upd[idx].astype(np.int32)
    ol = (om_off[1:] - om_off[:-1])[idx]
    ml = (mk_off[1:] - mk_off[:-1])[idx]
    n_off = np.concatenate([[0], np.cumsum(ol)]).astype(np.int64)
    k_off = np.concatenate([[0], np.cumsum(ml)]).astype(np.int64)
    def gather(data, off, lens, new_off):
        out = np.empty(int(new_off[-1]), data.dtype)
        if out.shape[0]:
            src = np.repeat(off[:-1][idx], lens) + (np.arange(int(new_off[-1])) - np.repeat(new_off[:-1], lens))
            out[:] = data[src]
        return out
    return f, n_off, gather(om, om_off, ol, n_off), k_off, gather(mk, mk_off, ml, k_off)


def run_lockstep(sweeps: List[PartitionedSweep], n_passes: int):
    """Drive all parts inside one process (LocalComm): same steps as PartitionedSweep.sweep, with the
    all-to-all replaced by in-process row shuffles.  Used to test the partition schedule on one GPU."""
    torch = sweeps[0].torch
    world = len(sweeps)

    def shuffle(rows, counts_out, counts_in):
        # rows[r] grouped by destination; deliver grouped by source
        offs = [np.concatenate([[0], np.cumsum(c)]) for c in counts_out]
        got = []
        for dst in range(world):
            pieces = [rows[src][offs[src][dst]: offs[src][dst + 1]] for src in range(world)]
            got.append(torch.cat(pieces) if pieces else rows[dst].new_zeros((0,) + tuple(rows[dst].shape[1:])))
            assert got[-1].shape[0] == int(sum(counts_in[dst]))
        return got

    programs = [s.program(n_passes) for s in sweeps]
    assert all(len(p) == len(programs[0]) and [x[0] for x in p] == [x[0] for x in programs[0]] for p in programs)
    for i, step in enumerate(programs[0]):
        if step[0] == "run":
            for s, prog in zip(sweeps, programs):
                s.run(prog[i][1])
        else:
            sent = [s.boundary_pack() for s in sweeps]
            recv = shuffle(sent, [s.out_counts for s in sweeps], [s.in_counts for s in sweeps])
            rep = [s.boundary_reply(r) for s, r in zip(sweeps, recv)]
            back = shuffle(rep, [s.in_counts for s in sweeps], [s.out_counts for s in sweeps])
            for s, b in zip(sweeps, back):
                s.boundary_fold(b)


class SetupLaps:
    """seconds of the steps of a driver's constructor (bench.py puts them into `setup_s`: what the time before the first pass is)"""

    def __init__(self):
        import time
        self.t, self.laps, self._clock = time.perf_counter(), {}, time.perf_counter

    def __call__(self, what: str):
        now = self._clock()
        self.laps[what] = self.laps.get(what, 0.0) + now - self.t
        self.t = now


class DriverStats:
    """what the bench.py drivers of all three schedules share: an untimed repetition of the passes under an ExchangeProbe, the
    split sizes of the largest exchange, and the switches of an engine whose device is shared with other ranks.  The drivers read
    and write the (borrowed) dual buffer between passes — directly or through the lpmp_halo_* / lpmp_boundary_* kernels — so
    passes that run ahead of the caller are off, and the rows layout is whatever the driver asked for explicitly (never the
    environment's LPMP_ROWS_LAYOUT)."""
    redundant_fraction = 0.0

    def own_the_engine(self, engine):
        if hasattr(engine, "set_speculation"):
            engine.set_speculation(0)
        # ranks that sit on the SAME physical device (smoke runs of an N-rank job on one GPU) must not use persistent launches:
        # found here, by the PCI address of every rank's device, not by the caller's guess from device ordinals
        dist = getattr(self, "dist", None)
        if hasattr(engine, "set_persistent_launches") and dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            from . import engine as E
            ids = [None] * dist.get_world_size()
            dist.all_gather_object(ids, E.device_identity(self.torch.cuda.current_device()), group=host_group(dist))
            self.shared_device = len(set(ids)) < len(ids)
            if self.shared_device and not os.environ.get("LPMP_BENCH_KEEP_PERSISTENT"):
                engine.set_persistent_launches(False)

    shared_device = False

    def set_shared_device(self, shared: bool):
        """ranks that time-share one GPU: persistent launches off for this rank's engine (include/lpmp_engine.h)"""
        if hasattr(self.engine, "set_persistent_launches"):
            self.engine.set_persistent_launches(not shared)

    def probe_passes(self, n: int) -> dict:
        """n passes with every exchange bracketed by events: this rank's compute / exchange split (ExchangeProbe.result)"""
        probe = ExchangeProbe(self.torch, bool(getattr(self.dualt, "is_cuda", False)))
        self.compute_pass(n, probe=probe)
        out = probe.result(n)
        out["redundant_fraction"] = float(self.redundant_fraction)
        return out

    def exchange_counts(self):
        sw = getattr(self, "sweep", None)
        if sw is None or getattr(self, "comm", None) is None:
            return None
        return sw.exchange_counts()


class StripSweep(DriverStats):
    """bench.py driver: this rank's H x W strip of a (world*H) x W grid on its own GPU."""

    def __init__(self, torch, dist, H, W, L, pairwise, order, mode, seed=1, omega_b=None, boundary_every="pass"):
        from . import engine as E
        self.torch, self.dist = torch, dist
        self.comm = DistComm(dist, torch)
        rank, world = self.comm.rank, self.comm.world
        dev = torch.device("cuda", torch.cuda.current_device())
        self.comm._dev = dev
        part = strip_local_part(H, W, L, pairwise, order, rank, world, seed, device_const=True)
        m = part.model
        stream = torch.cuda.current_stream().cuda_stream
        n_const = int(m.const_sizes().sum())
        self.const = torch.empty(max(n_const, 2), dtype=torch.float64, device=dev)
        self.dualt = torch.zeros(int(m.dual_sizes().sum()), dtype=torch.float64, device=dev)
        fill_device_costs(torch, E, part, self.const, self.dualt, stream)
        self.engine = E.Engine(torch.cuda.current_device())
        self.engine.set_stream(stream)
        self.engine.upload(m, const_dev=self.const.data_ptr(), dual_dev=self.dualt.data_ptr(), keep=(self.const, self.dualt), rows_layout=False)
        self.own_the_engine(self.engine)
        self.sweep = PartitionedSweep(torch, part, self.engine, self.dualt, mode, omega_b, boundary_every)
        t = torch.tensor([self.sweep.updates_per_pass(), self.sweep.bytes_per_pass()], dtype=torch.float64,
                         device="cpu" if self.comm.stage_cpu else dev)
        dist.all_reduce(t)
        self.global_updates_per_pass = int(t[0].item())
        self.global_bytes_per_pass = int(t[1].item())
        self.levels = [i["n_levels"] for i in self.sweep.info]
        self.cut_fraction = (world - 1) * W / max(1, world * strip_sizes(H, W)[1] + (world - 1) * W)

    def compute_pass(self, n=1, probe=None):
        self.sweep.compute_pass(self.comm, n, probe=probe)

    def lower_bound(self):
        return self.comm.all_reduce_sum(self.sweep.local_lower_bound())


def fill_device_costs(torch, E, part: LocalPart, const_t, dual_t, stream):
    """run a part's fill descriptors: (offset, count, seed, first) contiguous runs, or ("blocks", block_len, seed,
    first[]) scattered blocks of the global stream (rows of consecutive blocks in the buffer)"""
    keep = []
    for buf, fills in ((const_t, part.const_fill), (dual_t, part.dual_fill)):
        for f in fills or []:
            if f[0] == "blocks":
                _, blen, sd, first = f
                fd = torch.from_numpy(np.ascontiguousarray(first, np.int64)).to(buf.device)
                keep.append(fd)
                E.synth_fill_blocks(buf.data_ptr(), int(first.shape[0]), int(blen), sd, fd.data_ptr(), stream)
            else:
                off, cnt, sd, first = f
                E.synth_fill(buf.data_ptr() + 8 * off, cnt, sd, first, stream)
    torch.cuda.synchronize()
    return keep


class GraphSweep(DriverStats):
    """bench.py driver for C4 (BASELINE.json configs[3]): this rank's part of the random sparse graph on its own GPU.
    The global model is never materialised: structure from the counter generator, costs generated in HBM.  Random
    graphs cut most of their edges under any balanced partition, so the boundary step runs after every directional
    sweep (boundary_every="sweep")."""

    def __init__(self, torch, dist, n, m, L, mode, seed=1, omega_b=None, boundary_every=None, rows_layout=None, order="index",
                 part_of=None, partitioner="auto"):
        """``order``: "index" = the generator's variable order; "colour_major" = the variables renamed by
        ordering.colour_major_order (9 dependent levels per directional sweep instead of 33 on the C4 shape): the model is then
        synthetic.counter_graph_model(..., rank=self.rank_of) — what lockstep.LockstepGraph runs by default.
        ``part_of``: a partition handed in (variable -> rank, in the ORDERED variable numbering); else ``partitioner``
        (graph_partition's ``method``) computes one on rank 0"""
        from . import engine as E
        self.torch, self.dist = torch, dist
        self.comm = DistComm(dist, torch) if dist is not None and dist.is_initialized() else None
        if rows_layout and self.comm is not None and self.comm.world > 1:
            raise ValueError("rows layout: the partitioned sweep indexes the packed dual buffer between passes")
        rank, world = (self.comm.rank, self.comm.world) if self.comm else (0, 1)
        dev = torch.device("cuda", torch.cuda.current_device())
        if self.comm:
            self.comm._dev = dev
        lap = self.setup_laps = SetupLaps()
        self.order, self.rank_of = order, None
        if order == "colour_major":
            from . import ordering as O
            compute = lambda: O.colour_major_order(n, *S.counter_graph_edges(n, m, seed), seed=seed)
            self.rank_of = broadcast_partition(torch, dist, n, dev, compute) if self.comm and world > 1 else compute()
        elif order != "index":
            raise ValueError(order)
        lap("edges_and_variable_order_s")
        self.partitioner = "given" if part_of is not None else ("none (1 part)" if world == 1 else None)
        if part_of is None and self.comm and world > 1:  # partition once, on rank 0
            used = []
            def compute_part():
                p, how = graph_partition(n, *S.counter_graph_edges(n, m, seed, self.rank_of), world, method=partitioner, return_method=True)
                used.append(how)
                return p
            part_of = broadcast_partition(torch, dist, n, dev, compute_part)
            self.partitioner = broadcast_string(dist, used[0] if used else None)
        lap("partition_s")
        part = graph_local_part(n, m, L, rank, world, seed, part_of, self.rank_of)
        lap("local_part_model_s")
        self.part = part
        mdl = part.model
        stream = torch.cuda.current_stream().cuda_stream
        self.const = torch.empty(max(int(mdl.const_sizes().sum()), 2), dtype=torch.float64, device=dev)
        self.dualt = torch.zeros(int(mdl.dual_sizes().sum()), dtype=torch.float64, device=dev)
        fill_device_costs(torch, E, part, self.const, self.dualt, stream)
        lap("costs_in_hbm_s")
        self.engine = E.Engine(torch.cuda.current_device())
        self.engine.set_stream(stream)
        self.engine.upload(mdl, const_dev=self.const.data_ptr(), dual_dev=self.dualt.data_ptr(), keep=(self.const, self.dualt), rows_layout=bool(rows_layout))
        lap("plan_and_upload_s")
        if self.comm is not None:
            self.own_the_engine(self.engine)
        n_cut = int(part.out_ghost.shape[0] + part.in_unary.shape[0])
        self.cut_fraction = n_cut / max(1, int(mdl.n_messages) // 2 + int(part.in_unary.shape[0]))
        if boundary_every is None:                       # few cut edges: once per pass (fused sweeps); many: every sweep
            boundary_every = "sweep" if self.cut_fraction > 0.10 else "pass"
        self.boundary_every = boundary_every
        self.engine.set_reparametrization(mode)
        lap("weights_s")
        if self.comm is None:
            # one GPU, no process group: the engine's own pass schedules are the sweep (no boundary schedules to build beside them)
            # (message updates, algorithmic bytes and levels per directional sweep come from the directional schedules, which a run of
            # plain passes never builds for a model of this size: query_info() plans them — bench.py asks after its timed region)
            self.sweep = None
            self._mode = mode
            self.global_cut_fraction = 0.0
            self.global_updates_per_pass = self.global_bytes_per_pass = self.levels = None
            return
        self.sweep = PartitionedSweep(torch, part, self.engine, self.dualt, mode, omega_b, boundary_every, BOUNDARY_RESERVE)
        vals = [self.sweep.updates_per_pass(), self.sweep.bytes_per_pass(), part.out_ghost.shape[0], m]
        t = torch.tensor(vals[:3], dtype=torch.float64, device="cpu" if self.comm.stage_cpu else dev)
        dist.all_reduce(t)
        vals[:3] = [float(x) for x in t]
        self.global_updates_per_pass = int(vals[0])
        self.global_bytes_per_pass = int(vals[1])
        self.global_cut_fraction = vals[2] / m
        self.levels = [i["n_levels"] for i in self.sweep.info]

    def query_info(self):
        """one GPU: message updates / algorithmic bytes per pass (summed over the two directional sweeps) and dependent
        levels per direction, from the directional schedules (host planning only; seconds at the full C4 size)"""
        if self.sweep is None and self.levels is None:
            info = [self.engine.plan.schedule_info(d, self._mode) for d in (M.FORWARD, M.BACKWARD)]
            self.global_updates_per_pass = sum(int(i["n_receives"] + i["n_sends"]) for i in info)
            self.global_bytes_per_pass = sum(int(i["algorithmic_bytes"]) for i in info)
            self.levels = [i["n_levels"] for i in info]
        return self

    def prepare_passes(self, n):
        """what the first call of compute_pass(n) would build (one GPU: the engine's pass schedule): outside a timed region"""
        if self.sweep is None:
            self.engine.prepare_passes(n)

    def compute_pass(self, n=1, probe=None):
        if self.sweep is not None:
            self.sweep.compute_pass(self.comm, n, probe=probe)
        else:                                           # one part: no boundary, plain engine passes
            if probe is not None:
                probe.start()
            self.engine.compute_pass(n)
            if probe is not None:
                probe.stop()

    def lower_bound(self):
        if self.sweep is None:
            return self.engine.lower_bound()
        return self.comm.all_reduce_sum(self.sweep.local_lower_bound())


class ModelSweep(DriverStats):
    """Driver for an arbitrary partitioned model: this rank's part of ``global_model`` (partition_model) on its own
    GPU, cut messages exchanged through torch.distributed.  Every rank derives the partition from the same inputs."""

    def __init__(self, torch, dist, global_model: M.FlatModel, part_of: np.ndarray, mode, omega_b=None,
                 boundary_every="sweep", engine_factory=None):
        self.torch, self.dist = torch, dist
        self.comm = DistComm(dist, torch)
        self.part = partition_model(global_model, part_of, self.comm.world)[self.comm.rank]
        m = self.part.model
        if engine_factory is None:                        # the HIP engine on this rank's device
            from . import engine as E
            dev = torch.device("cuda", torch.cuda.current_device())
            self.comm._dev = dev
            self.dualt = torch.from_numpy(m.dual_data.copy()).to(dev)
            self.engine = E.Engine(torch.cuda.current_device())
            self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
            self.engine.upload(m, dual_dev=self.dualt.data_ptr(), keep=self.dualt, rows_layout=False)
            self.own_the_engine(self.engine)
        else:                                             # tests: an engine stand-in on a host buffer
            self.dualt, self.engine = engine_factory(m)
        self.sweep = PartitionedSweep(torch, self.part, self.engine, self.dualt, mode, omega_b, boundary_every)

    def compute_pass(self, n=1, probe=None):
        self.sweep.compute_pass(self.comm, n, probe=probe)

    def lower_bound(self):
        return self.comm.all_reduce_sum(self.sweep.local_lower_bound())
