"""Test-only helpers for the partitioned sweep: an oracle-backed stand-in for the device engine (so the
partition / exchange logic can run on CPU, incl. 2-process gloo runs) and the replay of the partition
schedule on the UNPARTITIONED model with the oracle's iterator-range ComputePass."""
import numpy as np

from lp_mp_amd import engine as E
from lp_mp_amd import model as M
from oracle.binding import Oracle


class OracleEngine:
    """Same methods as lp_mp_amd.engine.Engine, computed by the oracle on a shared numpy dual buffer.
    Lives in tests/ only: the product never routes through the oracle."""

    def __init__(self, model, dual_np):
        self.o = Oracle(model)
        self.dual = dual_np
        self.model = model
        self.plan = E.Plan(model)
        self.rows = []

    def schedule_create(self, factors, om_off, om, mk_off, mk, fuse=False):
        self.rows.append((np.array(factors), np.array(om_off), np.array(om), np.array(mk_off), np.array(mk)))
        return len(self.rows) - 1

    def schedule_run(self, sid):
        self.o.set_duals(self.dual)
        self.o.compute_pass_custom(*self.rows[sid])
        self.dual[:] = self.o.duals()

    def set_reparametrization(self, mode):
        self.o.set_reparametrization(mode)

    def compute_pass(self, n=1):
        self.o.set_duals(self.dual)
        self.o.ComputePass(n)
        self.dual[:] = self.o.duals()

    def schedule_info(self, sid):
        f, oo, om, mo, mk = self.rows[sid]
        return dict(n_levels=0, n_launches=0, n_receives=int(mk.sum()), n_sends=int((om != 0).sum()), algorithmic_bytes=0)

    def lower_bound(self):
        self.o.set_duals(self.dual)
        return self.o.LowerBound()

    def factor_lower_bounds(self):
        self.o.set_duals(self.dual)
        return np.array([self.o.factor_lower_bound(f) for f in range(self.model.n_factors)])


def global_replay(global_model, parts, sweeps, n_passes):
    """Runs the partition schedule on the UNPARTITIONED model with the oracle: every ("run", key) step of the parts'
    programs becomes one iterator-range ComputePass over the concatenation of the parts' row sets (expanded to the
    global message lists), every ("boundary",) step one ComputePass over the non-owner boundary unaries with only
    their cut slots active."""
    o = Oracle(global_model)
    g_off, g_ent = o.msg_lists()

    def glist(f):
        return g_ent[g_off[f]:g_off[f + 1]] // 2          # message ids in list order (MRF: every entry sends+receives)

    def main_step(i, programs):
        F, OM, MK, off = [], [], [], [0]
        for p, sw, prog in zip(parts, sweeps, programs):
            f_loc, om_off, om, mk_off, mk = sw.rows[prog[i][1]]
            l_off, l_ent = p._local_lists
            for r, fl in enumerate(f_loc):
                g = int(p.local_to_global[fl])
                lm = p.local_msg_to_global[l_ent[l_off[fl]:l_off[fl + 1]] // 2]
                gm = glist(g)
                o_row = np.zeros(gm.shape[0]); m_row = np.zeros(gm.shape[0], np.uint8)
                pos = {int(x): k for k, x in enumerate(gm)}
                lo = om[om_off[r]:om_off[r + 1]]; lk = mk[mk_off[r]:mk_off[r + 1]]
                for j, x in enumerate(lm):
                    o_row[pos[int(x)]] = lo[j]; m_row[pos[int(x)]] = lk[j]
                F.append(g); OM.append(o_row); MK.append(m_row); off.append(off[-1] + gm.shape[0])
        o.compute_pass_custom(np.array(F, np.int32), off, np.concatenate(OM), off, np.concatenate(MK))

    def boundary():
        cut = {}
        for p, sw in zip(parts, sweeps):
            for u, key, w in zip(p.in_unary, p.in_key, sw.in_omega):
                msg = int(key) if p.key_is_msg else 2 * int(key) + 1      # MRF parts key by edge: side-1 message
                cut.setdefault(int(p.local_to_global[u]), {})[msg] = float(w)
        # two iterator-range passes over the boundary variables (global index order): all receives, then all sends
        for recv in (True, False):
            F, OM, MK, off = [], [], [], [0]
            for g in sorted(cut):
                gm = glist(g)
                o_row = np.zeros(gm.shape[0]); m_row = np.zeros(gm.shape[0], np.uint8)
                for k, x in enumerate(gm):
                    if int(x) in cut[g]:
                        if recv:
                            m_row[k] = 1
                        else:
                            o_row[k] = cut[g][int(x)]
                F.append(g); OM.append(o_row); MK.append(m_row); off.append(off[-1] + gm.shape[0])
            if F:
                o.compute_pass_custom(np.array(F, np.int32), off, np.concatenate(OM), off, np.concatenate(MK))

    for n in ([n_passes] if isinstance(n_passes, int) else n_passes):      # one entry per compute_pass call
        programs = [sw.program(n) for sw in sweeps]
        for i, step in enumerate(programs[0]):
            if step[0] == "run":
                main_step(i, programs)
            else:
                boundary()
    return o


def gather_global_duals(global_model, parts, local_duals):
    """Scatter every part's local duals (ghosts excluded) into the global packed layout."""
    g_off = global_model.dual_offsets()
    out = np.full(int(g_off[-1]), np.nan)
    for p, d in zip(parts, local_duals):
        l_off = p.model.dual_offsets()
        for fl in range(p.model.n_factors):
            if p.n_local <= fl < p.n_local + p.n_ghost:
                assert np.all(d[l_off[fl]:l_off[fl + 1]] == 0.0)      # ghosts are empty between sweeps
                continue
            g = int(p.local_to_global[fl])
            out[g_off[g]:g_off[g + 1]] = d[l_off[fl]:l_off[fl + 1]]
    assert not np.isnan(out).any()
    return out


def attach_local_lists(parts):
    for p in parts:
        p._local_lists = E.Plan(p.model).msg_lists(p.model.n_messages)


def materialise_fills(part):
    """host copies of the costs a part would generate in HBM from its fill descriptors (multi_gpu.fill_device_costs):
    lets the per-rank generator of the C4 workload run on the oracle-backed stand-in engine"""
    from lp_mp_amd import synthetic as S
    m = part.model
    for name, fills, size in (("const_data", part.const_fill, int(m.const_sizes().sum())), ("dual_data", part.dual_fill, int(m.dual_sizes().sum()))):
        if fills is None:                                   # costs already on the host (Potts scalars of overlap.py's windows)
            continue
        buf = np.zeros(size)
        for f in fills or []:
            if f[0] == "blocks":
                _, blen, sd, first = f
                if len(first):
                    w = np.arange(blen, dtype=np.int64)[None, :] + np.asarray(first, np.int64)[:, None]     # stream positions
                    # u01(position) for scattered positions: the generator is a pure function of the counter
                    flat = w.reshape(-1)
                    vals = np.empty(flat.shape[0])
                    order = np.argsort(flat, kind="stable")
                    srt = flat[order]
                    runs = np.nonzero(np.diff(srt) != 1)[0] + 1
                    for a, b in zip(np.r_[0, runs], np.r_[runs, srt.shape[0]]):
                        vals[order[a:b]] = S.u01(b - a, sd, int(srt[a]))
                    buf[: flat.shape[0]] = vals
            else:
                off, cnt, sd, first = f
                buf[off: off + cnt] = S.u01(cnt, sd, first)
        setattr(m, name, buf)
    return part
