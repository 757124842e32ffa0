"""Lock-step partitioned sweep: several GPUs run THE unpartitioned sweep, not an approximation of it.

multi_gpu.PartitionedSweep gives every part its own sub-problem and reconciles the cut messages in a boundary step; its
dual bound after n passes stays below the unpartitioned sweep's (1 - 2 % on a random graph).  This module
does the other thing the level schedule allows: updates of one dependency level commute, so the global
level-sorted sequence of the reference's sweep (LP::ComputePass, reference include/LP_MP.h:981-1005) can be executed by
several ranks at once — every rank runs the updates of ITS variables with the GLOBAL weights (LP::get_omega, :412-460,
computed once on the global structure) level by level, and between two levels the ranks exchange the message vectors the
next level reads across the cut.  The result is the unpartitioned sweep's, bit for bit, on any number of ranks: the gap
to the unpartitioned bound is zero by construction, and every step is an iterator-range pass of the reference.

What makes the exchange a plain halo copy: a pairwise factor's dual is [side 0 | side 1] and side s is written ONLY by the
updates of endpoint s (its receives rewrite its own side, its sends add to its own side), read by endpoint 1 - s.  Both
ranks of a cut edge hold a copy of the pairwise factor (table and dual) and a never-updated ghost of the remote variable;
after a level in which endpoint s wrote, its rank ships side s to the other copy.

The same holds for any model whose messages all have the `left` schedule (lockstep_model: multicut triplets, C5's labeling-list
factors, mixed label counts): the unit shipped is the slice of a higher factor's dual one message writes — one side of a pairwise
factor, the WHOLE dual of a labeling-list factor (every message into it rewrites all of it, so such a write also needs the other
ranks' last writes first) — and it goes to every other rank that holds a copy of that factor.

Levels are merged into SEGMENTS greedily (identically on every rank, from the global structure): a segment ends before the
first level that reads a cut vector written inside it.  A colour-major grid in row strips: two segments per pass.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import model as M
from . import synthetic as S


@dataclass
class LockstepPart:
    rank: int
    world: int
    L: int
    model: M.FlatModel                 # local variables + ghosts (by global index), every pairwise factor touching a local variable
    vars_global: np.ndarray            # local vector factor -> global variable
    is_ghost: np.ndarray               # [n local vector factors]
    edges_global: np.ndarray           # local pairwise factor (in local edge order) -> global edge
    owned: np.ndarray                  # [n local factors] bool: counted in this rank's share of the lower bound
    rows: List[List[tuple]]            # [direction][sub-level] = (factors, om_off, om, mk_off, mk) of this rank's updates (sub-level: LockstepSchedule.program)
    const_fill: Optional[list] = None
    dual_fill: Optional[list] = None
    # the exchange units this part holds a copy of: global vector id (ascending), first element in the local dual array, length
    vec_ids: Optional[np.ndarray] = None
    vec_start: Optional[np.ndarray] = None
    vec_len: Optional[np.ndarray] = None
    factors_global: Optional[np.ndarray] = None   # lockstep_model: local factor -> global factor


@dataclass
class LockstepSchedule:
    """what every rank computes identically from the global structure"""
    n_levels: Tuple[int, int]
    # cut vectors (one per message: an MRF's 2 * edge + side) written / read per (direction, level), who writes a vector and
    # which other ranks hold a copy of it
    written: List[List[np.ndarray]]
    read: List[List[np.ndarray]]
    writer: np.ndarray                 # [n_vecs] rank of the message's left factor (the only writer of that slice)
    dest_off: np.ndarray               # [n_vecs + 1] CSR: the other ranks holding the vector's factor (an MRF: the other endpoint's rank, if it differs)
    dest: np.ndarray
    n_vecs: int
    _programs: Dict[int, list] = field(default_factory=dict)
    _halo_ids: Dict[bytes, int] = field(default_factory=dict)     # distinct exchanges (sets of vectors) of all programs -> small integer

    def program_overlapped(self, n_passes: int):
        """program(n) with every exchange taken off the critical path where the sweep allows it: the run that follows an exchange
        begins with the sub-levels that READ nothing the exchange ships (the interior records of the level whose cut-adjacent
        records just ran: same level, so they commute with them) — those are issued between ("halo_begin", ...) — pack, collective
        posted — and ("halo_end", ...) — collective awaited, unpack —, so that the transfer runs while they compute.  Same records
        in an order that differs only between commuting updates: the result is program(n)'s bit for bit.  Where a run begins with
        a reader of the shipped vectors (a random graph with 60 % of its edges cut: C4) the exchange stays a plain ("halo", ...)."""
        key = ("overlapped", n_passes)
        if key in self._programs:
            return self._programs[key]
        steps = self.program(n_passes)
        out, i = [], 0
        while i < len(steps):
            st = steps[i]
            if st[0] == "halo" and i + 1 < len(steps) and steps[i + 1][0] == "run":
                shipped = np.zeros(self.n_vecs, bool); shipped[st[1]] = True
                seq = steps[i + 1][1]
                k = 0
                while k < len(seq) and not shipped[self.read[seq[k][0]][seq[k][1]]].any():
                    k += 1
                if k > 0:
                    out.append(("halo_begin", st[1], st[2]))
                    out.append(("run", tuple(seq[:k])))
                    out.append(("halo_end", st[1], st[2]))
                    if k < len(seq):
                        out.append(("run", tuple(seq[k:])))
                    i += 2
                    continue
            out.append(st)
            i += 1
        self._programs[key] = out
        return out

    def program(self, n_passes: int):
        """steps of n passes: ("run", ((d, sub-level), ...)) and ("halo", vectors to ship: sorted global ids, id of that set).
        Sub-level 2 l = the records of level l + 1 that touch a cut edge, 2 l + 1 = the others (they write no cut vector).
        An exchange is needed before the first sub-level that reads a cut vector written since the last one; it is then
        moved BACK over the sub-levels that wrote nothing it ships (they read nothing it ships either, or it would have
        come earlier): on strips of a 2-colour grid that puts it right behind the boundary rows of a colour step, and the
        bulk of that step lands in the same schedule as the colour's update of the opposite sweep — the two fold."""
        if n_passes in self._programs:
            return self._programs[n_passes]
        seq = [(d, sl) for _ in range(n_passes) for d in (0, 1) for sl in range(2 * self.n_levels[d])]
        dirty = np.zeros(self.n_vecs, bool)
        where, last = [], 0                             # positions in seq an exchange comes before
        for i, (d, sl) in enumerate(seq):
            if dirty[self.read[d][sl]].any():
                pos = i
                while pos - 1 >= last and self.written[seq[pos - 1][0]][seq[pos - 1][1]].size == 0:
                    pos -= 1
                where.append(pos); dirty[:] = False; last = pos
            dirty[self.written[d][sl]] = True
        # what each exchange ships: not everything written since the last one, only what is READ before the next one — a vector
        # that is rewritten before anyone reads it (anisotropic weights: the side a receive rewrites is next read after the same
        # endpoint's send of the opposite sweep) travels once, with its later state.  (Every read still finds its vector shipped:
        # the positions were found with everything shipped, so between a write and the next read of its vector lies an exchange,
        # and the last one before the read takes it.)
        dirty[:] = False
        halos, start = [], 0
        for k, pos in enumerate(where):
            for (d, sl) in seq[start:pos]:
                dirty[self.written[d][sl]] = True
            nxt = where[k + 1] if k + 1 < len(where) else len(seq)
            need = np.zeros(self.n_vecs, bool)
            for (d, sl) in seq[pos:nxt]:
                need[self.read[d][sl]] = True
            ship = np.nonzero(dirty & need)[0]
            dirty[ship] = False
            if ship.size:
                halos.append((pos, ship))
            start = pos
        for (d, sl) in seq[start:]:
            dirty[self.written[d][sl]] = True
        def halo(vecs):                                 # (the id is what a rank's exchange plans are kept under: a step of a pass
            hid = self._halo_ids.setdefault(vecs.tobytes(), len(self._halo_ids))   # must not hash its vector list every time)
            return ("halo", vecs, hid)
        steps, start = [], 0
        for pos, vecs in halos:
            if pos > start:
                steps.append(("run", tuple(seq[start:pos])))
            steps.append(halo(vecs)); start = pos
        if start < len(seq):
            steps.append(("run", tuple(seq[start:])))
        if dirty.any():                                 # the copies agree again when the call returns
            steps.append(halo(np.nonzero(dirty)[0]))
        self._programs[n_passes] = steps
        return steps


def _small_ints(a):
    """the same values in the narrowest unsigned type that holds them: numpy's stable argsort is a radix sort for 8- and 16-bit
    keys (levels of a sweep: a handful to a few thousand), a merge sort otherwise"""
    if a.size and a.min() >= 0:
        m = int(a.max())
        if m < 256:
            return a.astype(np.uint8)
        if m < 65536:
            return a.astype(np.uint16)
    return a


def _csr_take(off, data, idx):
    """rows ``idx`` of the CSR array (off, data), back to back: (offsets of the taken rows, their entries)"""
    lens = off[idx + 1] - off[idx]
    first = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(first[-1])
    if total == 0:
        return first, data[:0]
    # positions as a running sum: +1 inside a row, a jump to the next row's start at every row boundary (one pass over the
    # output instead of two repeats and an arange: these arrays hold 20 M entries at the C4 size)
    step = np.ones(total, np.int64)
    nz = lens > 0
    starts = off[idx][nz].astype(np.int64)
    at = first[:-1][nz]
    step[at[0]] = starts[0]
    if at.shape[0] > 1:
        step[at[1:]] = starts[1:] - (starts[:-1] + lens[nz][:-1] - 1)
    return first, data[np.cumsum(step)]


def lockstep_mrf(n_vars: int, L: int, edge_i, edge_j, part, world: int, mode: int, unaries=None, tables=None, potts=None,
                 only: Optional[int] = None, stream_seed: Optional[int] = None, pairwise: str = "dense"):
    """Lock-step parts of the MRF synthetic.mrf_model(n_vars, L, edge_i, edge_j, ...) for the partition ``part[v]``.
    Returns (schedule, parts); ``only``: just that rank's part.  Costs as in multi_gpu.partition_mrf (host arrays, or
    ``stream_seed``: generated in HBM from the counter stream, fill descriptors in the part)."""
    from . import engine as E
    edge_i = np.asarray(edge_i, np.int64); edge_j = np.asarray(edge_j, np.int64); part = np.asarray(part, np.int64)
    n_edges = edge_i.shape[0]
    if part.shape[0] != n_vars or part.min() < 0 or part.max() >= world:
        raise ValueError("lockstep_mrf: part must name a rank in [0, world) for every variable")
    if np.bincount(part, minlength=world).min() == 0:
        raise ValueError("lockstep_mrf: a rank without variables (every rank takes part in every exchange)")
    # the global structure (no costs) and everything the reference derives from it
    if pairwise == "dense":
        gm = S.mrf_model(n_vars, L, edge_i, edge_j, None, device_const=True, device_dual=True)
    else:
        gm = S.mrf_model(n_vars, L, edge_i, edge_j, None, potts=np.zeros(n_edges), device_dual=True)
    gp = E.Plan(gm)
    g_off, g_ent = gp.msg_lists(gm.n_messages)
    msg = g_ent // 2                                         # messages of every factor's list; message 2 e + s: edge e, side s
    writer = np.empty(2 * n_edges, np.int64); writer[0::2] = part[edge_i]; writer[1::2] = part[edge_j]
    reader = np.empty(2 * n_edges, np.int64); reader[0::2] = part[edge_j]; reader[1::2] = part[edge_i]
    is_cut_vec = writer != reader
    per_dir = []
    n_levels, written, read = [], [], []
    for d in (M.FORWARD, M.BACKWARD):
        upd = gp.update_order(d).astype(np.int64)
        om_off, om = gp.omega(d, mode)
        mk_off, mk = gp.mask(d, mode)
        lev = np.maximum(gp.update_levels(d, mode).astype(np.int64), 1)     # (0: no active message; runs with the first level)
        assert np.all(upd < n_vars), "lock-step parts: only the variables are updated (schedule `left`)"
        lens = g_off[upd + 1] - g_off[upd]
        assert np.array_equal(lens, om_off[1:] - om_off[:-1]) and np.array_equal(lens, mk_off[1:] - mk_off[:-1])
        _, vec_own = _csr_take(g_off, msg, upd)              # per row entry: own-side vector id = the message id (2 e + s)
        row_of = np.repeat(np.arange(upd.shape[0]), lens)
        active_w = (om != 0.0) | (mk != 0)
        active_r = mk != 0
        cut = is_cut_vec[vec_own]
        nl = int(lev.max()) if lev.size else 0

        def by_level(sel, flip):                             # cut vectors of the selected row entries, grouped by level
            v, lv = vec_own[sel] ^ flip, lev[row_of[sel]]
            order = np.argsort(_small_ints(lv), kind="stable")
            bounds = np.searchsorted(lv[order], np.arange(1, nl + 2))
            out = []
            for l in range(nl):                              # sub-levels: the records that touch a cut edge, then the others
                out += [np.unique(v[order[bounds[l]: bounds[l + 1]]]), np.zeros(0, np.int64)]
            return out
        w_l, r_l = by_level(cut & active_w, 0), by_level(cut & active_r, 1)
        n_levels.append(nl); written.append(w_l); read.append(r_l)
        touches_cut = np.zeros(upd.shape[0], bool)
        touches_cut[row_of[cut & active_w]] = True
        per_dir.append((upd, om_off, om, mk_off, mk, lev, touches_cut))
    dest_off = np.concatenate([[0], np.cumsum(is_cut_vec)]).astype(np.int64)
    sched = LockstepSchedule((n_levels[0], n_levels[1]), written, read, writer, dest_off, reader[is_cut_vec], 2 * n_edges)

    parts = []
    for k in (range(world) if only is None else [only]):
        local = part == k
        le = np.nonzero(local[edge_i] | local[edge_j])[0]               # every edge touching a local variable, global order
        vk = np.unique(np.concatenate([np.nonzero(local)[0], edge_i[le], edge_j[le]]))
        lmap = np.full(n_vars, -1, np.int64); lmap[vk] = np.arange(vk.shape[0])
        li, lj = lmap[edge_i[le]], lmap[edge_j[le]]
        ghost = ~local[vk]
        const_fill = dual_fill = None
        esz = L * L if pairwise == "dense" else 1
        if stream_seed is not None:
            if pairwise == "dense":
                m = S.mrf_model(vk.shape[0], L, li, lj, None, device_const=True, device_dual=True)
            else:
                m = S.mrf_model(vk.shape[0], L, li, lj, None, potts=np.zeros(le.shape[0]), device_dual=True)
            const_fill = [("blocks", esz, stream_seed, (n_vars * L + le * esz).astype(np.int64))]
            dual_fill = [("blocks", L, stream_seed, (vk * L).astype(np.int64))]
        else:
            un = np.asarray(unaries, np.float64).reshape(n_vars, L)[vk]
            if pairwise == "dense":
                m = S.mrf_model(vk.shape[0], L, li, lj, un, tables=np.asarray(tables, np.float64).reshape(n_edges, L, L)[le])
            else:
                m = S.mrf_model(vk.shape[0], L, li, lj, un, potts=np.asarray(potts, np.float64)[le])
        owned = np.concatenate([~ghost, part[edge_i[le]] == k])        # a pairwise factor counts where its earlier endpoint lives
        rows = []
        for (upd, om_off, om, mk_off, mk, lev, touches_cut) in per_dir:
            mine = np.nonzero(local[upd])[0]
            sub = 2 * (lev[mine] - 1) + (~touches_cut[mine])      # sub-level: by level, the cut-touching records first
            order = mine[np.argsort(sub, kind="stable")]          # sequence order inside a sub-level
            nl = int(lev.max()) if lev.size else 0
            bounds = np.searchsorted(np.sort(sub, kind="stable"), np.arange(0, 2 * nl + 1))
            per_level = []
            for l in range(2 * nl):
                idx = order[bounds[l]: bounds[l + 1]]
                fo, o = _csr_take(om_off, om, idx)
                fm, k_ = _csr_take(mk_off, mk, idx)
                per_level.append((lmap[upd[idx]].astype(np.int32), fo, o, fm, k_))
            rows.append(per_level)
        pw_off = m.dual_offsets()[vk.shape[0]: vk.shape[0] + le.shape[0]]      # dual offset of local pairwise factor e (local edge order)
        parts.append(LockstepPart(k, world, L, m, vk, ghost, le, owned, rows, const_fill, dual_fill,
                                  vec_ids=(2 * le[:, None] + np.arange(2)[None, :]).reshape(-1), vec_start=(pw_off[:, None] + L * np.arange(2)[None, :]).reshape(-1),
                                  vec_len=np.full(2 * le.shape[0], L, np.int64)))
    return sched, parts


def lockstep_model(gm: M.FlatModel, part, world: int, mode: int, only: Optional[int] = None):
    """Lock-step parts of ANY model whose messages all have the `left` schedule and whose factors are either variables (left
    factor of their messages, or no message at all) or higher factors (right factor), with unary-pairwise or labeling messages —
    what multi_gpu.partition_model accepts.  ``part[f]``: rank of variable f (ignored for higher factors).  Returns
    (schedule, parts) like lockstep_mrf; the parts run the unpartitioned sweep of ``gm`` bit for bit.
    A part holds its variables, every higher factor touching one of them with ALL its messages, and never-updated ghosts of the
    remote variables behind those; factors and messages keep the global relative order (message lists of a local variable are
    the global ones).  A higher factor counts in the bound where its lowest-numbered variable lives."""
    from . import engine as E
    part = np.asarray(part, np.int64)
    nf, nm = gm.n_factors, gm.n_messages
    ml, mr = gm.m_left.astype(np.int64), gm.m_right.astype(np.int64)
    for t in gm.mtypes:
        if t.schedule != M.SCHED_LEFT or t.kind not in (M.M_UNARY_PAIRWISE, M.M_LABELING):
            raise ValueError("lockstep_model: only `left`-schedule unary-pairwise / labeling messages")
    is_right = np.zeros(nf, bool); is_right[mr] = True
    is_left = np.zeros(nf, bool); is_left[ml] = True
    if np.any(is_left & is_right):
        raise ValueError("lockstep_model: a factor is both left and right of messages")
    is_var = ~is_right
    if part.shape[0] != nf or part[is_var].min() < 0 or part[is_var].max() >= world:
        raise ValueError("lockstep_model: part must name a rank in [0, world) for every variable")
    if np.bincount(part[is_var], minlength=world).min() == 0:
        raise ValueError("lockstep_model: a rank without variables (every rank takes part in every exchange)")
    gp = E.Plan(gm)
    g_off, g_ent = gp.msg_lists(nm)
    msg_of_entry = g_ent // 2
    # the slice of its higher factor's dual a message writes: side `param` of a pairwise factor, all of a labeling-list factor
    kind = np.array([t.kind for t in gm.mtypes], np.int64)[gm.m_type]
    side = np.array([t.param for t in gm.mtypes], np.int64)[gm.m_type]
    d0 = gm.f_dim0[mr].astype(np.int64)
    d1 = np.where(gm.f_kind[mr] == M.F_PAIRWISE_POTTS, d0, gm.f_dim1[mr].astype(np.int64))
    up = kind == M.M_UNARY_PAIRWISE
    v_off = np.where(up & (side == 1), d0, 0)
    v_len = np.where(up, np.where(side == 1, d1, d0), gm.dual_sizes()[mr])
    whole = ~up
    writer = part[ml]
    # ranks holding a copy of higher factor r: those of its variables; a vector goes to all of them but its writer
    by_r = np.argsort(mr, kind="stable")
    sib_off = np.concatenate([[0], np.cumsum(np.bincount(mr, minlength=nf))]).astype(np.int64)      # messages of every factor as a right factor
    sib = by_r
    hold = np.unique(mr * world + writer)                      # (r, rank) pairs
    hold_off = np.concatenate([[0], np.cumsum(np.bincount(hold // world, minlength=nf))]).astype(np.int64)
    hold_rank = hold % world
    first, q = _csr_take(hold_off, hold_rank, mr)              # per message: the holders of its factor
    kq = np.repeat(np.arange(nm), np.diff(first))
    far = q != writer[kq]
    dest = q[far]
    dest_off = np.concatenate([[0], np.cumsum(np.bincount(kq[far], minlength=nm))]).astype(np.int64)
    is_cut_vec = np.diff(dest_off) > 0

    per_dir = []
    n_levels, written, read = [], [], []
    for d in (M.FORWARD, M.BACKWARD):
        upd = gp.update_order(d).astype(np.int64)
        om_off, om = gp.omega(d, mode)
        mk_off, mk = gp.mask(d, mode)
        lev = np.maximum(gp.update_levels(d, mode).astype(np.int64), 1)
        if np.any(is_right[upd]):
            raise ValueError("lockstep_model: only variables may be updated (a higher factor computes a primal or sends)")
        lens = g_off[upd + 1] - g_off[upd]
        assert np.array_equal(lens, om_off[1:] - om_off[:-1]) and np.array_equal(lens, mk_off[1:] - mk_off[:-1])
        _, k_own = _csr_take(g_off, msg_of_entry, upd)        # per row entry: its message
        row_of = np.repeat(np.arange(upd.shape[0]), lens)
        writes = (om != 0.0) | (mk != 0)
        reads = (mk != 0) | (writes & whole[k_own])           # a write into a labeling-list factor rewrites all of it
        nl = int(lev.max()) if lev.size else 0

        def by_level(v, lv):
            order = np.argsort(_small_ints(lv), kind="stable")
            bounds = np.searchsorted(lv[order], np.arange(1, nl + 2))
            out = []
            for l in range(nl):                              # sub-levels: the records that touch a cut vector, then the others
                out += [np.unique(v[order[bounds[l]: bounds[l + 1]]]), np.zeros(0, np.int64)]
            return out
        sel_w = writes & is_cut_vec[k_own]
        w_l = by_level(k_own[sel_w], lev[row_of[sel_w]])
        # what an entry reads across the cut: the vectors of its factor's other messages written on another rank
        rd = np.nonzero(reads)[0]
        f2, k2 = _csr_take(sib_off, sib, mr[k_own[rd]])
        e2 = np.repeat(rd, np.diff(f2))
        remote = writer[k2] != writer[k_own[e2]]
        r_l = by_level(k2[remote], lev[row_of[e2[remote]]])
        n_levels.append(nl); written.append(w_l); read.append(r_l)
        touches_cut = np.zeros(upd.shape[0], bool)
        touches_cut[row_of[sel_w]] = True
        per_dir.append((upd, om_off, om, mk_off, mk, lev, touches_cut))
    sched = LockstepSchedule((n_levels[0], n_levels[1]), written, read, writer, dest_off, dest, nm)

    first_var = np.full(nf, nf, np.int64)
    np.minimum.at(first_var, mr, ml)
    owner = np.where(is_right, part[np.minimum(first_var, nf - 1)], part)
    coff, doff = gm.const_offsets(), gm.dual_offsets()

    def take(data, off, idx):
        if data is None:
            return None
        _, out = _csr_take(off, data, idx)
        return np.ascontiguousarray(out)

    parts = []
    for k in (range(world) if only is None else [only]):
        local = is_var & (part == k)
        in_r = np.zeros(nf, bool); in_r[mr[local[ml]]] = True             # higher factors touching a local variable
        mk_sel = np.nonzero(in_r[mr])[0]                                   # all their messages, global order
        keep = local | in_r
        keep[ml[mk_sel]] = True                                            # + ghosts of the remote variables behind them
        fk = np.nonzero(keep)[0]
        lmap = np.full(nf, -1, np.int64); lmap[fk] = np.arange(fk.shape[0])
        ghost = is_var[fk] & ~local[fk]

        def map_rel(rel):
            rel = np.asarray(rel, np.int64).reshape(-1, 2)
            a, b = lmap[rel[:, 0]], lmap[rel[:, 1]]
            ok = (a >= 0) & (b >= 0)
            return np.ascontiguousarray(np.stack([a[ok], b[ok]], 1).astype(np.int32)).reshape(-1, 2)
        m = M.FlatModel(
            n_ftypes=gm.n_ftypes, ftype_computes_primal=gm.ftype_computes_primal, mtypes=gm.mtypes,
            tab_off=gm.tab_off, tab_data=gm.tab_data, tab_nleft=gm.tab_nleft,
            f_type=np.ascontiguousarray(gm.f_type[fk]), f_kind=np.ascontiguousarray(gm.f_kind[fk]), f_flags=np.ascontiguousarray(gm.f_flags[fk]),
            f_dim0=np.ascontiguousarray(gm.f_dim0[fk]), f_dim1=np.ascontiguousarray(gm.f_dim1[fk]),
            const_data=take(gm.const_data, coff, fk), dual_data=take(gm.dual_data, doff, fk),
            m_type=np.ascontiguousarray(gm.m_type[mk_sel]), m_left=lmap[ml[mk_sel]].astype(np.int32), m_right=lmap[mr[mk_sel]].astype(np.int32),
            rel_fwd=map_rel(gm.rel_fwd), rel_bwd=map_rel(gm.rel_bwd), constant=gm.constant if k == 0 else 0.0)
        owned = owner[fk] == k
        rows = []
        for (upd, om_off, om, mk_off, mk, lev, touches_cut) in per_dir:
            mine = np.nonzero(local[upd])[0]
            sub = 2 * (lev[mine] - 1) + (~touches_cut[mine])
            order = mine[np.argsort(sub, kind="stable")]
            nl = int(lev.max()) if lev.size else 0
            bounds = np.searchsorted(np.sort(sub, kind="stable"), np.arange(0, 2 * nl + 1))
            per_level = []
            for l in range(2 * nl):
                idx = order[bounds[l]: bounds[l + 1]]
                fo, o = _csr_take(om_off, om, idx)
                fm, k_ = _csr_take(mk_off, mk, idx)
                per_level.append((lmap[upd[idx]].astype(np.int32), fo, o, fm, k_))
            rows.append(per_level)
        ldoff = m.dual_offsets()
        parts.append(LockstepPart(k, world, 0, m, fk[is_var[fk]], ghost[is_var[fk]], np.zeros(0, np.int64), owned, rows,
                                  vec_ids=mk_sel, vec_start=ldoff[lmap[mr[mk_sel]]] + v_off[mk_sel], vec_len=v_len[mk_sel], factors_global=fk))
    return sched, parts


def _cat(rows):
    f = np.concatenate([r[0] for r in rows])
    def offs(i):
        out, base = [np.zeros(1, np.int64)], 0
        for r in rows:
            out.append(base + r[i][1:]); base += int(r[i][-1])
        return np.concatenate(out)
    return f, offs(1), np.concatenate([r[2] for r in rows]), offs(3), np.concatenate([r[4] for r in rows])


class LockstepSweep:
    """one rank of the lock-step sweep.  ``engine``: lp_mp_amd.engine.Engine with the part's model uploaded (or a stand-in
    with the same methods in CPU tests); ``dual_tensor``: torch view of the engine's dual buffer."""

    def __init__(self, torch, part: LockstepPart, sched: LockstepSchedule, engine, dual_tensor, overlap_exchange: bool = False):
        """``overlap_exchange``: run LockstepSchedule.program_overlapped — the collective of an exchange is posted asynchronously and
        awaited only before the first reader of what it ships (same results)"""
        self.torch, self.part, self.sched, self.engine, self.dual = torch, part, sched, engine, dual_tensor
        self.overlap_exchange = bool(overlap_exchange)
        self._open = None
        self._sids: Dict[tuple, int] = {}
        self._sid_of: Dict[int, tuple] = {}
        self._halo: Dict[int, tuple] = {}
        self._device_halos = hasattr(engine, "halo_create")          # the HIP engine (CPU tests run oracle-backed stand-ins without it)
        self._send = None
        self.info = {}

    def _schedule(self, seg: tuple) -> int:
        hit = self._sid_of.get(id(seg))                     # (the steps of a cached program are the same tuple objects every call: no
        if hit is not None:                                 # hashing of a run of thousands of sub-levels per step)
            return hit[1]
        sid = self._schedule_of(seg)
        self._sid_of[id(seg)] = (seg, sid)                  # (the tuple is kept: its id stays its own)
        return sid

    def _schedule_of(self, seg: tuple) -> int:
        if seg not in self._sids:
            rows = [self.part.rows[d][l] for (d, l) in seg]
            n_sweeps = 1 + sum(1 for a, b in zip(seg[:-1], seg[1:]) if b[0] != a[0] or b[1] < a[1])
            f, oo, om, mo, mk = _cat(rows)
            if f.shape[0] == 0:
                self._sids[seg] = -1
            else:
                self._sids[seg] = self.engine.schedule_create(f, oo, om, mo, mk, fuse=n_sweeps > 1)
                if hasattr(self.engine, "schedule_info"):
                    self.info[seg] = self.engine.schedule_info(self._sids[seg])
        return self._sids[seg]

    def run(self, seg: tuple):
        sid = self._schedule(seg)
        if sid >= 0:
            self.engine.schedule_run(sid)

    def _halo_plan(self, vecs: np.ndarray, key: Optional[int] = None):
        if key is None:
            key = self.sched._halo_ids.setdefault(vecs.tobytes(), len(self.sched._halo_ids))
        if key not in self._halo:
            p, s = self.part, self.sched
            lens = s.dest_off[vecs + 1] - s.dest_off[vecs]
            _, q = _csr_take(s.dest_off, s.dest, vecs)             # (vector, rank holding another copy of it) pairs
            v = np.repeat(vecs, lens)
            at = np.searchsorted(p.vec_ids, v)
            at[at >= p.vec_ids.shape[0]] = 0
            here = p.vec_ids[at] == v if p.vec_ids.shape[0] else np.zeros(v.shape[0], bool)     # the vectors this part holds
            src = s.writer[v]

            def plan(sel, peer):                                   # the selected pairs by peer, then vector id: (first element, length) each
                idx = np.nonzero(sel)[0]
                idx = idx[np.lexsort((v[idx], peer[idx]))]
                assert here[idx].all()
                return p.vec_start[at[idx]].astype(np.int64), p.vec_len[at[idx]].astype(np.int64), peer[idx]
            # (a part built from a proxy world, strips_lockstep_part: its ranks are shifted into the true world)
            shift, world = getattr(p, "rank_shift", 0), getattr(p, "true_world", p.world)
            count = lambda peer, ln: np.bincount(peer + shift, weights=ln, minlength=world).astype(np.int64)
            o_start, o_len, o_peer = plan(src == p.rank, q)
            i_start, i_len, i_peer = plan(q == p.rank, src)
            if self._device_halos:
                # the HIP engine: copies by its halo kernels (include/lpmp_engine.h, lpmp_halo_*), the send buffer kept with the plan
                h = self.engine.halo_create(o_start, o_len, i_start, i_len)
                self._halo[key] = (h, count(o_peer, o_len), int(o_len.sum()), count(i_peer, i_len))
            else:
                def elems(start, ln):                              # (stand-in engines on the CPU: index tensors into the dual array)
                    first = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
                    return np.repeat(start, ln) + (np.arange(int(first[-1])) - np.repeat(first[:-1], ln))
                dev = self.dual.device
                self._halo[key] = (self.torch.from_numpy(elems(o_start, o_len)).to(dev), count(o_peer, o_len),
                                   self.torch.from_numpy(elems(i_start, i_len)).to(dev), count(i_peer, i_len))
        return self._halo[key]

    def halo_pack(self, vecs, key=None):
        src, out_counts, buf, in_counts = self._halo_plan(vecs, key)
        if self._device_halos:
            # one send buffer for all exchanges of this part (work on a stream is ordered: the previous exchange has read it)
            if self._send is None or self._send.shape[0] < buf:
                self._send = self.torch.empty(max(buf, 1), dtype=self.torch.float64, device=self.dual.device)
            self.engine.halo_pack(src, self._send.data_ptr())
            return self._send[:buf], out_counts, in_counts
        return self.dual[src], out_counts, in_counts

    def halo_unpack(self, vecs, recv, key=None):
        h, _, dst, in_counts = self._halo_plan(vecs, key)
        if self._device_halos:
            if int(in_counts.sum()):
                recv = recv.contiguous()
                assert recv.shape[0] == int(in_counts.sum())
                self.engine.halo_unpack(h, recv.data_ptr())
                self._last_recv = recv                             # (alive until the copy has run: the allocator is stream-ordered)
            return
        if dst.shape[0]:
            self.dual[dst] = recv

    def close(self):
        if self._device_halos:
            for h in self._halo.values():
                self.engine.halo_destroy(h[0])
        self._halo = {}

    def compute_pass(self, comm, n=1, probe=None):
        """``probe``: multi_gpu.ExchangeProbe — every pack -> all-to-all -> unpack span and the whole call get event pairs"""
        if probe is not None:
            probe.start()
        for step in self.steps(n):
            if step[0] == "run":
                self.run(step[1])
            elif step[0] == "halo_begin":                # pack, post the collective; the interior records that follow run meanwhile
                if probe is not None:
                    probe.begin_post()                   # (exchange work on this stream that nothing hides: booked as such)
                send, out_counts, in_counts = self.halo_pack(step[1], step[2])
                # (a send buffer of its own: the next pack may come before this transfer has read it)
                self._open = (comm.exchange_begin(send.clone() if self._device_halos else send, out_counts, in_counts), out_counts, in_counts)
                if probe is not None:
                    probe.end_post()
            elif step[0] == "halo_end":
                if probe is not None:
                    probe.begin_exchange()               # (what is left of the exchange on the critical path)
                pending, out_counts, in_counts = self._open
                self._open = None
                self.halo_unpack(step[1], comm.exchange_end(pending), step[2])
                if probe is not None:
                    probe.end_exchange(out_counts.sum(), in_counts.sum())
            else:
                if probe is not None:
                    probe.begin_exchange()
                send, out_counts, in_counts = self.halo_pack(step[1], step[2])
                self.halo_unpack(step[1], comm.exchange(send, out_counts, in_counts), step[2])
                if probe is not None:
                    probe.end_exchange(out_counts.sum(), in_counts.sum())
        if probe is not None:
            probe.stop()

    def steps(self, n):
        return self.sched.program_overlapped(n) if self.overlap_exchange else self.sched.program(n)

    def exchange_counts(self, n=2):
        """split sizes (doubles per peer rank) of the largest exchange of an n-pass call: for a self test of the collective"""
        # (every rank must name the SAME exchange: the one that ships the most vectors over all ranks — the program is global)
        best = None
        for step in self.sched.program(n):               # (the overlapped program ships the same sets)
            if step[0] == "halo" and (best is None or step[1].shape[0] > best[1].shape[0]):
                best = step
        world = getattr(self.part, "true_world", self.part.world)
        if best is None:
            return np.zeros(world, np.int64), np.zeros(world, np.int64)
        _, out_counts, _, in_counts = self._halo_plan(best[1], best[2])
        return out_counts, in_counts

    def local_lower_bound(self) -> float:
        if hasattr(self.engine, "invalidate_lower_bounds"):
            self.engine.invalidate_lower_bounds()                  # the halo copies edit pairwise duals behind the engine's back
        flb = np.asarray(self.engine.factor_lower_bounds())
        return float(flb[self.part.owned].sum())

    def updates_per_pass(self) -> int:
        return sum(int((r[2] != 0).sum()) + int(r[4].sum()) for d in (0, 1) for r in self.part.rows[d])


def run_lockstep(sweeps: List[LockstepSweep], n_passes: int):
    """all parts inside one process (tests, several parts on one GPU): the all-to-all as in-process row shuffles"""
    torch = sweeps[0].torch
    world = len(sweeps)
    held = None
    for step in sweeps[0].steps(n_passes):
        if step[0] == "run":
            for s in sweeps:
                s.run(step[1])
            continue
        if step[0] == "halo_begin":                      # (in one process nothing overlaps: the packed buffers wait for halo_end)
            held = [(p[0].clone(), p[1], p[2]) for p in (s.halo_pack(step[1], step[2]) for s in sweeps)]
            continue
        packed = held if step[0] == "halo_end" else [s.halo_pack(step[1], step[2]) for s in sweeps]
        held = None
        offs = [np.concatenate([[0], np.cumsum(p[1])]) for p in packed]
        for dst, s in enumerate(sweeps):
            pieces = [packed[src][0][offs[src][dst]: offs[src][dst + 1]] for src in range(world)]
            recv = torch.cat(pieces)
            assert recv.shape[0] == int(packed[dst][2].sum())
            s.halo_unpack(step[1], recv, step[2])


# ---- drivers (one process per GPU) --------------------------------------------------------------------------------
def strips_lockstep_part(H: int, W: int, L: int, pairwise: str, order: str, rank: int, world: int, mode: int, seed: int,
                         proxy: bool = True):
    """This rank's lock-step part of the (world * H) x W strip grid of multi_gpu.strip_global_edges, costs generated in HBM.
    The global structure of `world` strips is never built when world > 3: strips are translates of each other, so a strip
    with a neighbour on both sides looks the same in any world — the part and the schedule come from a 3-strip PROXY
    (first, interior, last strip) and only the positions in the cost stream and the peer ranks are shifted
    (tests/test_lockstep.py compares with the parts of the true global structure)."""
    from . import multi_gpu as MG
    n_loc, e_int = MG.strip_sizes(H, W)
    # (only where the level structure is the same in every strip: 2-colour orders.  A row-major order chains its levels
    # through the strips — the global structure is built then)
    pw = world if (not proxy or world <= 3 or order != "colour_major") else 3
    pr = rank if pw == world else (0 if rank == 0 else 2 if rank == world - 1 else 1)
    ei, ej = MG.strip_global_edges(H, W, pw, order)
    part_of = np.repeat(np.arange(pw), n_loc)
    sched, parts = lockstep_mrf(pw * n_loc, L, ei, ej, part_of, pw, mode, only=pr, stream_seed=seed, pairwise=pairwise,
                                unaries=None if pairwise == "dense" else np.zeros(pw * n_loc * L), potts=None if pairwise == "dense" else np.zeros(ei.shape[0]))
    p = parts[0]
    shift = rank - pr                                             # strips between the proxy's and the true position
    n_true = world * n_loc
    esz = L * L if pairwise == "dense" else 1
    var_true = p.vars_global + shift * n_loc
    edge_true = p.edges_global + shift * (e_int + W)
    p.const_fill = [("blocks", esz, seed, (n_true * L + edge_true * esz).astype(np.int64))]
    p.dual_fill = [("blocks", L, seed, (var_true * L).astype(np.int64))]
    p.rank_shift, p.true_world = shift, world
    return sched, p


from .multi_gpu import DriverStats


class _Driver(DriverStats):
    def _setup(self, torch, dist, part, sched, mode, fill=True, rows_layout=False, engine_factory=None):
        """``rows_layout``: dense pairwise factors as [table | m1 | m2] rows (lpmp_set_rows_layout) — the halo kernels address the
        rows themselves, so the lock-step exchange works on either layout; never taken from the environment here.
        ``engine_factory(part) -> (dual tensor, engine)``: tests run the driver logic (partition hand-over, program, exchange,
        statistics) on a stand-in engine over gloo; the product path below creates the HIP engine and fails without a GPU"""
        from . import engine as E
        from . import multi_gpu as MG
        self.torch, self.dist, self.part, self.sched = torch, dist, part, sched
        self.comm = MG.DistComm(dist, torch) if dist is not None and dist.is_initialized() else None
        if engine_factory is not None:
            self.dualt, self.engine = engine_factory(part)
            self.engine.set_reparametrization(mode)
            self.sweep = LockstepSweep(torch, part, sched, self.engine, self.dualt)
            vals = torch.tensor([float(self.sweep.updates_per_pass()), 0.0], dtype=torch.float64)
            if self.comm:
                dist.all_reduce(vals)
            self.global_updates_per_pass, self.global_bytes_per_pass, self.levels = int(vals[0].item()), 0, list(sched.n_levels)
            return
        dev = torch.device("cuda", torch.cuda.current_device())
        if self.comm:
            self.comm._dev = dev
        m = part.model
        stream = torch.cuda.current_stream().cuda_stream
        if fill:
            self.const = torch.empty(max(int(m.const_sizes().sum()), 2), dtype=torch.float64, device=dev)
            self.dualt = torch.zeros(int(m.dual_sizes().sum()), dtype=torch.float64, device=dev)
            if m.const_data is not None and part.const_fill is None:
                self.const[: m.const_data.shape[0]] = torch.from_numpy(m.const_data).to(dev)
            MG.fill_device_costs(torch, E, part, self.const, self.dualt, stream)
        else:
            self.const = torch.from_numpy(np.ascontiguousarray(m.const_data if m.const_data is not None and m.const_data.size else np.zeros(2))).to(dev)
            self.dualt = torch.from_numpy(m.dual_data.copy()).to(dev)
        self.engine = E.Engine(torch.cuda.current_device())
        self.engine.set_stream(stream)
        self.engine.upload(m, const_dev=self.const.data_ptr(), dual_dev=self.dualt.data_ptr(), keep=(self.const, self.dualt), rows_layout=bool(rows_layout))
        self.own_the_engine(self.engine)
        self.engine.set_reparametrization(mode)
        self.sweep = LockstepSweep(torch, part, sched, self.engine, self.dualt, overlap_exchange=getattr(self, "overlap_exchange", False))
        # the schedules of the steady state (built here, outside any timed region) and their algorithmic bytes per pass
        n_probe = 4
        prog = self.sweep.steps(n_probe)
        by = 0
        for step in prog:
            if step[0] == "run" and self.sweep._schedule(step[1]) >= 0:
                by += self.sweep.info[step[1]]["algorithmic_bytes"]
        vals = torch.tensor([float(self.sweep.updates_per_pass()), by / n_probe], dtype=torch.float64, device="cpu" if (self.comm is None or self.comm.stage_cpu) else dev)
        if self.comm:
            dist.all_reduce(vals)
        self.global_updates_per_pass = int(vals[0].item())
        self.global_bytes_per_pass = int(vals[1].item())
        self.levels = list(sched.n_levels)

    def prepare_passes(self, n):
        """what depends on the pass count of a call (schedules of its segments, exchange plans): outside a timed region"""
        for step in self.sweep.steps(n):
            if step[0] == "run":
                self.sweep._schedule(step[1])
            elif step[0] != "halo_end":
                self.sweep._halo_plan(step[1], step[2])

    def compute_pass(self, n=1, probe=None):
        if self.comm is None:
            if probe is not None:
                probe.start()
            for step in self.sweep.steps(n):
                if step[0] == "run":
                    self.sweep.run(step[1])
            if probe is not None:
                probe.stop()
            return
        self.sweep.compute_pass(self.comm, n, probe=probe)

    def lower_bound(self):
        lb = self.sweep.local_lower_bound()
        return self.comm.all_reduce_sum(lb) if self.comm else lb

    def halo_steps_per_pass(self, n=4):
        return sum(1 for s in self.sched.program(n) if s[0] == "halo") / n

    def close(self):
        """exchange plans (device arrays behind lpmp_halo_*) and the engine"""
        self.sweep.close()
        self.engine.close()


class LockstepStrips(_Driver):
    """bench.py driver: this rank's H x W strip of the (world * H) x W grid, run in lock step with the other strips —
    the result is the single-GPU sweep of the whole grid, bit for bit."""

    def __init__(self, torch, dist, H, W, L, pairwise, order, mode, seed=1, proxy=True, overlap_exchange=False):
        self.overlap_exchange = overlap_exchange
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None and dist.is_initialized() else (0, 1)
        sched, part = strips_lockstep_part(H, W, L, pairwise, order, rank, world, mode, seed, proxy)
        self._setup(torch, dist, part, sched, mode)
        from . import multi_gpu as MG
        self.cut_fraction = (world - 1) * W / max(1, world * MG.strip_sizes(H, W)[1] + (world - 1) * W)


class LockstepGraph(_Driver):
    """the same for the C4-style random graph synthetic.counter_graph_model(n, m, L, seed); every rank derives the global
    structure from the counter generator (no costs), the partition comes from rank 0.
    ``order``: "colour_major" (default) renames the variables by ordering.colour_major_order first — one dependent level per
    colour (9 per directional sweep on the C4 shape instead of 30), i.e. 18 exchanges per pass instead of 60; "index": the
    generator's own order.  The order is part of the problem (another order is another, equally valid sweep): the
    unpartitioned sweep the result equals bit for bit is the one of counter_graph_model(..., rank=self.rank_of)."""

    def __init__(self, torch, dist, n, m, L, mode, seed=1, part_of=None, order="colour_major", partitioner="auto", rows_layout=False,
                 engine_factory=None, overlap_exchange=False):
        """``part_of``: a partition handed in (variable -> rank in the ORDERED numbering, e.g. multi_gpu.load_partition_file); else
        ``partitioner`` (multi_gpu.graph_partition's ``method``: auto / metis / builtin) computes one on rank 0"""
        from . import multi_gpu as MG
        from . import ordering as O
        on = dist is not None and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
        self.order = order
        self.overlap_exchange = overlap_exchange
        self.rank_of = None
        lap = self.setup_laps = MG.SetupLaps()
        self.partitioner = "given" if part_of is not None else "none (1 part)"
        if order == "colour_major":
            compute = lambda: O.colour_major_order(n, *S.counter_graph_edges(n, m, seed), seed=seed)
            # (a colouring of 2 M variables / 10 M edges is a minute of numpy: once, on rank 0)
            self.rank_of = MG.broadcast_partition(torch, dist, n, None, compute) if on and world > 1 else compute()
        elif order != "index":
            raise ValueError(order)
        ei, ej = S.counter_graph_edges(n, m, seed, self.rank_of)
        lap("edges_and_variable_order_s")
        if part_of is None:
            if world > 1:
                used = []
                def compute_part():
                    p, how = MG.graph_partition(n, ei, ej, world, method=partitioner, return_method=True)
                    used.append(how)
                    return p
                part_of = MG.broadcast_partition(torch, dist, n, None, compute_part)
                self.partitioner = MG.broadcast_string(dist, used[0] if used else None)
            else:
                part_of = np.zeros(n, np.int64)
        part_of = np.asarray(part_of, np.int64)
        self.part_of = part_of
        self.cut_fraction = float((part_of[ei] != part_of[ej]).mean())
        lap("partition_s")
        sched, parts = lockstep_mrf(n, L, ei, ej, part_of, world, mode, only=rank, stream_seed=seed)
        lap("global_plan_and_local_part_s")
        self._setup(torch, dist, parts[0], sched, mode, rows_layout=rows_layout, engine_factory=engine_factory)
        lap("costs_plan_upload_schedules_s")


class LockstepModel(_Driver):
    """one rank of the lock-step sweep of an arbitrary `left`-schedule model (lockstep_model): C5's grid + labeling-list factors,
    multicut triplets, ...  Every rank holds ``global_model`` on the host (structure and costs) and takes its own part of it;
    ``part_of[f]``: rank of variable f (e.g. multi_gpu.graph_partition_model, computed once and broadcast)."""

    def __init__(self, torch, dist, global_model: M.FlatModel, part_of, mode, engine_factory=None, overlap_exchange=False):
        self.overlap_exchange = overlap_exchange
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None and dist.is_initialized() else (0, 1)
        sched, parts = lockstep_model(global_model, part_of, world, mode, only=rank)
        self._setup(torch, dist, parts[0], sched, mode, fill=False, engine_factory=engine_factory)
        # share of the cut: messages whose vector has a copy on another rank
        self.cut_fraction = float((np.diff(sched.dest_off) > 0).mean()) if sched.n_vecs else 0.0

