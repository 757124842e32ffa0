"""Overlapped strips: several GPUs run THE unpartitioned sweep of a big grid, each one as plain joined passes.

The lock-step sweep (lockstep.py) reproduces the unpartitioned sweep (LP::ComputePass, reference include/LP_MP.h:869-887,
981-1005) exactly, but pays for it with an exchange behind every dependent level — on the headline grid two per pass, each
of which cuts the persistent joined-pass launch in two and with it the Infinity-Cache reuse of the pairwise
tables: 6.5 ms per pass and part against 5.05 on one GPU.  A grid in a 2-colour order allows something cheaper, the
ghost-zone trick of stencil codes: information travels ONE grid row per half sweep (a variable's update reads its four
pairwise factors, whose other sides were written by its neighbours' previous updates), two rows per pass.  So a rank that
holds, besides its own H rows, g more rows on either side can run n = g / 2 - 1 passes WITHOUT any exchange: whatever is
wrong at the rim of its window (the outermost row misses its neighbours) has moved 2 n rows inwards by then and has not
reached a row the rank owns.  Then the ranks refresh each other's ghost rows with the owners' values — one exchange per n
passes instead of 2 n — and go on.

    * the global model is the (world * H) x W grid in ONE global colour-major order (synthetic.grid_model): at world = 1
      the headline model itself;
    * rank r holds rows [r H - g, (r + 1) H + g) (clipped to the grid) as a grid model of its own in colour-major order —
      g even, so the window's colouring is the global one and the relative order of any two neighbours the global one:
      the anisotropic weights the engine derives from the window (LP::ComputeAnisotropicWeights, LP_MP.h:1232-1415:
      a function of the relative order of a factor's neighbours and of their neighbours) are the global weights for every
      variable that has all its neighbours in the window;
    * every rank's work is `lpmp_compute_pass(n)` — the single-GPU hot path: n joined passes as one persistent launch in
      Infinity-Cache order — over H + 2 g rows instead of H (g = 12, H = 1024: + 2.3 %);
    * the owned rows equal the unpartitioned sweep's bit for bit (tests/test_overlap.py: oracle on the unpartitioned grid),
      the bound is the sum over the factors a rank owns: the dual-bound gap is 0.

Only for 2-colour grids (`order="colour_major"`, `left` schedule MRF): on a random graph the 2 n-hop neighbourhood of a
part is the whole graph; those take the lock-step sweep.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import model as M
from . import synthetic as S


# ---- closed-form enumeration of the global grid (no global array is ever built) -----------------------------------------
def _blacks_in_rows(r, W: int):
    """number of black cells ((row + col) even) in rows [0, r) of a W-column grid"""
    r = np.asarray(r, np.int64)
    return (r * W + ((r & 1) if (W & 1) else 0)) // 2


def global_var_index(rows, cols, GH: int, W: int):
    """synthetic.grid_variable_order(GH, W, "colour_major")[rows, cols] without building the grid"""
    rows = np.asarray(rows, np.int64); cols = np.asarray(cols, np.int64)
    black = ((rows + cols) & 1) == 0
    nb_before = _blacks_in_rows(rows, W) + np.where((rows & 1) == 0, (cols + 1) // 2, cols // 2)
    k = rows * W + cols
    return np.where(black, nb_before, int(_blacks_in_rows(GH, W)) + (k - nb_before))


def global_edge_index(rows, cols, down, GH: int, W: int):
    """position in synthetic.grid_edges(GH, W) of the right (down = 0) or down (down = 1) edge of node (rows, cols)"""
    rows = np.asarray(rows, np.int64); cols = np.asarray(cols, np.int64); down = np.asarray(down, np.int64)
    inner = rows * (2 * W - 1) + 2 * cols + np.where(down == 1, np.where(cols < W - 1, 1, 0), 0)
    last = (GH - 1) * (2 * W - 1) + cols
    return np.where(rows < GH - 1, inner, last)


def window_rows(rank: int, world: int, H: int, g: int) -> Tuple[int, int]:
    return max(0, rank * H - g), min(world * H, (rank + 1) * H + g)


@dataclass
class OverlapPart:
    rank: int
    world: int
    H: int
    W: int
    L: int
    g: int
    r0: int
    r1: int
    model: M.FlatModel                  # the window as a grid model of its own (colour-major)
    vars_global: np.ndarray             # local variable -> global variable
    edges_global: np.ndarray            # local edge -> global edge
    owned: np.ndarray                   # [n local factors] bool
    var_owner: np.ndarray               # rank that owns each local variable / edge
    edge_owner: np.ndarray
    var_row: np.ndarray                 # global row of each local variable / of each local edge's first node
    edge_row: np.ndarray
    edge_row2: np.ndarray               # global row of each local edge's second node
    const_fill: Optional[list] = None
    dual_fill: Optional[list] = None


def strip_window_part(H: int, W: int, L: int, pairwise: str, rank: int, world: int, g: int, seed: int = 1,
                      costs: Optional[dict] = None, compute_primal: bool = False) -> OverlapPart:
    """rank's window of the (world * H) x W grid of synthetic.grid_model(world * H, W, L, pairwise, "colour_major", seed).
    ``costs``: host arrays of the GLOBAL model (tests: {"unaries": [n, L], "tables": [E, L, L] or "potts": [E]}); else the
    costs are generated in HBM from the counter stream (fill descriptors, multi_gpu.fill_device_costs)."""
    if g % 2 or g < 4:
        raise ValueError("overlap: the ghost depth must be even (the window keeps the global colouring) and at least 4 (one pass between exchanges needs 4 rows)")
    if H % 2:
        raise ValueError("overlap: strips need an even number of rows (every window starts on an even global row)")
    if world > 1 and g > H:
        raise ValueError("overlap: ghost rows reach beyond the neighbouring strip (g <= H)")
    GH = world * H
    r0, r1 = window_rows(rank, world, H, g)
    h = r1 - r0
    n_g = GH * W
    var_l = S.grid_variable_order(h, W, "colour_major").reshape(-1)          # node (row-major, local) -> local variable
    rr, cc = np.divmod(np.arange(h * W, dtype=np.int64), W)
    vars_global = np.empty(h * W, np.int64)
    vars_global[var_l] = global_var_index(rr + r0, cc, GH, W)
    var_row = np.empty(h * W, np.int64); var_row[var_l] = rr + r0
    a, b = S.grid_edges(h, W)
    down = (b - a == W).astype(np.int64)
    ar, ac = np.divmod(a, W)
    edges_global = global_edge_index(ar + r0, ac, down, GH, W)
    edge_row = ar + r0
    edge_row2 = b // W + r0
    # the relative order inside the window is the global one (what makes the window's own weights the global ones)
    assert np.all(np.diff(vars_global) > 0) and np.all(np.diff(edges_global) > 0)
    va, vb = var_l[a], var_l[b]
    li, lj = np.minimum(va, vb), np.maximum(va, vb)
    n_l, E_l = h * W, a.shape[0]
    esz = L * L if pairwise == "dense" else 1
    const_fill = dual_fill = None
    if costs is None:
        if pairwise == "dense":
            m = S.mrf_model(n_l, L, li, lj, np.zeros(n_l * L), device_const=True, compute_primal=compute_primal)
            const_fill = [("blocks", esz, seed, (n_g * L + edges_global * esz).astype(np.int64))]
        else:                                                              # Potts: one scalar per edge, host side
            pos = n_g * L + edges_global
            m = S.mrf_model(n_l, L, li, lj, np.zeros(n_l * L), potts=_u01_at(pos, seed), compute_primal=compute_primal)
        dual_fill = [("blocks", L, seed, (vars_global * L).astype(np.int64))]
    else:
        un = np.asarray(costs["unaries"], np.float64).reshape(n_g, L)[vars_global]
        if pairwise == "dense":
            m = S.mrf_model(n_l, L, li, lj, un, tables=np.asarray(costs["tables"], np.float64).reshape(-1, L, L)[edges_global], compute_primal=compute_primal)
        else:
            m = S.mrf_model(n_l, L, li, lj, un, potts=np.asarray(costs["potts"], np.float64)[edges_global], compute_primal=compute_primal)
    var_owner = var_row // H
    edge_owner = edge_row // H                                              # a pairwise factor lives with its upper / left endpoint
    owned = np.concatenate([var_owner == rank, edge_owner == rank])
    return OverlapPart(rank, world, H, W, L, g, r0, r1, m, vars_global, edges_global, owned, var_owner, edge_owner, var_row, edge_row,
                       edge_row2, const_fill, dual_fill)


def _u01_at(pos: np.ndarray, seed: int) -> np.ndarray:
    """the counter stream at scattered positions (a pure function of the counter)"""
    pos = np.asarray(pos, np.int64)
    out = np.empty(pos.shape[0])
    if pos.size == 0:
        return out
    order = np.argsort(pos, kind="stable"); srt = pos[order]
    cuts = np.nonzero(np.diff(srt) != 1)[0] + 1
    for s, e in zip(np.r_[0, cuts], np.r_[cuts, srt.shape[0]]):
        out[order[s:e]] = S.u01(int(e - s), seed, int(srt[s]))
    return out


def max_passes_between_exchanges(g: int) -> int:
    """what is wrong at the rim of a window moves two rows per pass (one per directional sweep); the outermost row is wrong
    from the first update on and one more row of margin keeps the cut edge's far side exact: g >= 2 n + 2"""
    if g < 4:
        raise ValueError("overlap: fewer than 4 ghost rows allow no pass at all between two exchanges")
    return (g - 2) // 2


class OverlapSweep:
    """one rank.  ``engine``: lp_mp_amd.engine.Engine with the window uploaded and the mode set (or a stand-in with
    compute_pass / factor_lower_bounds in CPU tests); ``dual_tensor``: torch view of the engine's dual buffer."""

    def __init__(self, torch, part: OverlapPart, engine, dual_tensor, chunk: Optional[int] = None):
        self.torch, self.part, self.engine, self.dual = torch, part, engine, dual_tensor
        self.chunk = max_passes_between_exchanges(part.g) if chunk is None else int(chunk)
        if part.world > 1 and self.chunk > max_passes_between_exchanges(part.g):
            raise ValueError("overlap: %d passes between exchanges need %d ghost rows, the window has %d" % (self.chunk, 2 * self.chunk + 2, part.g))
        p, L = part, part.L
        doff = p.model.dual_offsets()
        n_l = p.vars_global.shape[0]
        dev = dual_tensor.device

        def elements(var_sel, edge_sel):
            """flat dual elements of the selected variables (global order) followed by the selected edges (global order)"""
            v = np.nonzero(var_sel)[0]; e = np.nonzero(edge_sel)[0]               # local order IS global order inside a window
            ve = (doff[v][:, None] + np.arange(L)[None, :]).reshape(-1)
            ee = (doff[n_l + e][:, None] + np.arange(2 * L)[None, :]).reshape(-1)
            return np.concatenate([ve, ee]).astype(np.int64)

        self.peers: List[int] = []
        self._send: Dict[int, object] = {}
        self._recv: Dict[int, object] = {}
        self.send_counts = np.zeros(p.world, np.int64)
        self.recv_counts = np.zeros(p.world, np.int64)
        for q in (p.rank - 1, p.rank + 1):
            if q < 0 or q >= p.world:
                continue
            q0, q1 = window_rows(q, p.world, p.H, p.g)
            # what I own and q holds: variables in q's rows, edges with both nodes in q's rows
            s = elements((p.var_owner == p.rank) & (p.var_row >= q0) & (p.var_row < q1),
                          (p.edge_owner == p.rank) & (p.edge_row >= q0) & (p.edge_row2 < q1))
            # what q owns and I hold
            r = elements(p.var_owner == q, p.edge_owner == q)
            self.peers.append(q)
            self._send[q] = torch.from_numpy(s).to(dev); self._recv[q] = torch.from_numpy(r).to(dev)
            self.send_counts[q] = s.shape[0]; self.recv_counts[q] = r.shape[0]
        self.exchanges = 0

    # ---- the exchange: owners' values into the neighbours' ghost rows -------------------------------------------------
    def pack(self):
        if not self.peers:
            return self.dual[:0]
        return self.torch.cat([self.dual[self._send[q]] for q in self.peers])          # by destination rank (ascending)

    def unpack(self, recv):
        at = 0
        for q in self.peers:
            n = int(self.recv_counts[q])
            self.dual[self._recv[q]] = recv[at: at + n]
            at += n
        self.exchanges += 1

    def chunks(self, n: int) -> List[int]:
        if self.part.world == 1:
            return [n]
        out = []
        while n > 0:
            k = min(n, self.chunk); out.append(k); n -= k
        return out

    def compute_pass(self, comm, n: int = 1, probe=None):
        """``probe``: multi_gpu.ExchangeProbe — every pack -> all-to-all -> unpack span and the whole call get event pairs"""
        if probe is not None:
            probe.start()
        for k in self.chunks(n):
            self.engine.compute_pass(k)
            if self.part.world > 1:
                if probe is not None:
                    probe.begin_exchange()
                self.unpack(comm.exchange(self.pack(), self.send_counts, self.recv_counts))
                if probe is not None:
                    probe.end_exchange(self.send_counts.sum(), self.recv_counts.sum())
        if probe is not None:
            probe.stop()

    def exchange_counts(self):
        return self.send_counts, self.recv_counts

    def local_lower_bound(self) -> float:
        if hasattr(self.engine, "invalidate_lower_bounds"):
            self.engine.invalidate_lower_bounds()                  # the exchange edits duals behind the engine's back
        flb = np.asarray(self.engine.factor_lower_bounds())
        return float(flb[self.part.owned].sum())


def run_overlapped(sweeps: List[OverlapSweep], n_passes: int):
    """all parts inside one process (tests, several parts on one GPU): the exchange as in-process copies"""
    torch = sweeps[0].torch
    for k in sweeps[0].chunks(n_passes):
        for s in sweeps:
            s.engine.compute_pass(k)
        if len(sweeps) == 1:
            continue
        packed = [s.pack() for s in sweeps]
        offs = [np.concatenate([[0], np.cumsum(s.send_counts)]) for s in sweeps]
        for dst, s in enumerate(sweeps):
            pieces = [packed[src][offs[src][dst]: offs[src][dst + 1]] for src in s.peers]
            s.unpack(torch.cat(pieces) if pieces else packed[dst][:0])


def grid_pass_counts(GH: int, W: int, L: int, pairwise: str = "dense") -> Tuple[int, int]:
    """message updates and algorithmic bytes of one anisotropic pass over the GH x W grid: every message is
    received once and sent once"""
    E = GH * (W - 1) + (GH - 1) * W
    n = GH * W
    per_msg = (8 * L * L if pairwise == "dense" else 8) + 40 * L
    return 4 * E, 2 * E * per_msg + 32 * L * n


# ---- driver (one process per GPU) -----------------------------------------------------------------------------------
from .multi_gpu import DriverStats


class OverlapStrips(DriverStats):
    """bench.py driver: this rank's window of the (world * H) x W grid.  The result is the single-GPU sweep of the whole
    grid, bit for bit; between two exchanges a rank runs plain joined passes."""

    def __init__(self, torch, dist, H, W, L, pairwise, mode, seed=1, g=12, chunk=None):
        from . import engine as E
        from . import multi_gpu as MG
        self.torch, self.dist = torch, dist
        on = dist is not None and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
        self.comm = MG.DistComm(dist, torch) if on else None
        dev = torch.device("cuda", torch.cuda.current_device())
        if self.comm:
            self.comm._dev = dev
        part = strip_window_part(H, W, L, pairwise, rank, world, g, seed, compute_primal=True)
        self.part = part
        m = part.model
        stream = torch.cuda.current_stream().cuda_stream
        self.const = torch.empty(max(int(m.const_sizes().sum()), 2), dtype=torch.float64, device=dev)
        self.dualt = torch.zeros(int(m.dual_sizes().sum()), dtype=torch.float64, device=dev)
        if m.const_data is not None and part.const_fill is None and m.const_data.size:
            self.const[: m.const_data.shape[0]] = torch.from_numpy(m.const_data).to(dev)
        MG.fill_device_costs(torch, E, part, self.const, self.dualt, stream)
        self.engine = E.Engine(torch.cuda.current_device())
        self.engine.set_stream(stream)
        # (the exchange reads and writes the borrowed packed dual buffer directly between passes: packed layout, no passes ahead)
        self.engine.upload(m, const_dev=self.const.data_ptr(), dual_dev=self.dualt.data_ptr(), keep=(self.const, self.dualt), rows_layout=False)
        if self.comm is not None:
            self.own_the_engine(self.engine)
        self.engine.set_reparametrization(mode)
        self.sweep = OverlapSweep(torch, part, self.engine, self.dualt, chunk)
        self.global_updates_per_pass, self.global_bytes_per_pass = grid_pass_counts(world * H, W, L, pairwise)
        info = [self.engine.plan.schedule_info(d, mode) for d in (0, 1)]
        self.levels = [i["n_levels"] for i in info]
        self.window_rows = (part.r0, part.r1)
        # work this rank does beyond its share (ghost rows are updated too)
        self.redundant_fraction = (part.r1 - part.r0) / H - 1.0
        self.cut_fraction = (world - 1) * W / max(1, (world * H) * (W - 1) + (world * H - 1) * W)

    def prepare_passes(self, n):
        for k in sorted(set(self.sweep.chunks(n))):
            self.engine.prepare_passes(k)

    def compute_pass(self, n=1, probe=None):
        if self.comm is None:
            if probe is not None:
                probe.start()
            self.engine.compute_pass(n)
            if probe is not None:
                probe.stop()
        else:
            self.sweep.compute_pass(self.comm, n, probe=probe)

    def lower_bound(self):
        lb = self.sweep.local_lower_bound()
        return self.comm.all_reduce_sum(lb) if self.comm else lb

    def exchanges_per_pass(self, n=20):
        return (len(self.sweep.chunks(n)) / n) if self.part.world > 1 else 0.0
