"""Dual-bound gap of the partitioned sweep against the unpartitioned one, on the CPU (oracle-backed engines, lock-stepped
parts): the experiment behind BOUNDARY_SHARE and the boundary schedule.
    python tests/gap_probe.py [n] [m] [L] [world] [passes] [share,share,...] [every]
The last line is the LOCK-STEP sweep of the same parts (lp_mp_amd/lockstep.py): the unpartitioned sweep itself, level by
level with halo copies in between — gap 0 by construction, one exchange per level that reads across the cut.
GAP_RESERVE=x overrides multi_gpu.BOUNDARY_RESERVE (0: the main sweeps keep nothing back); GAP_VARIANT: experiments."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # lives in tests/: it runs the oracle (test infrastructure)
import numpy as np
import torch
from lp_mp_amd import model as M, multi_gpu as MG, synthetic as S, lockstep as LS
from oracle.binding import Oracle
from mgpu_helpers import OracleEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
world = int(sys.argv[4]) if len(sys.argv) > 4 else 2
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 8
shares = [float(x) for x in sys.argv[6].split(",")] if len(sys.argv) > 6 else [MG.BOUNDARY_SHARE]
every = sys.argv[7] if len(sys.argv) > 7 else "sweep"

class DirectionalSweep(MG.PartitionedSweep):
    """experiment: the non-owner endpoint of a cut edge is always the LATER one, so the forward sweep's boundary step only
    pulls (receive, nothing sent back) and the backward sweep's only pushes (no pull)"""
    step_no = 0

    def boundary_pack(self):
        self.pull = (self.step_no % 2 == 0)
        self.step_no += 1
        if not self.pull:
            return self.dual.new_zeros((int(self.out_counts.sum()),))
        return super().boundary_pack()

    def boundary_reply(self, recv):
        if not self.rounds:
            return recv.new_zeros((0,))
        pull = self.pull
        for tgt, src in self.rounds:
            self.dual[tgt] += recv[src]
        w = self.in_omega_t * (0.0 if pull else 1.0)
        reply = w * self.dual[self.in_elems_t]
        for tgt, src in self.rounds:
            self.dual[tgt] -= reply[src]
        return reply

    def boundary_fold(self, recv):
        if self.part.n_ghost == 0:
            return
        if self.pull:
            return                              # nothing was sent back
        self.dual[self.out_elems_t] = recv
        self.engine.schedule_run(self.ghost_send)


t0 = time.time()
g = S.counter_graph_model(n, m, L, 1)
ei = g.m_left[0::2].astype(np.int64); ej = g.m_left[1::2].astype(np.int64)
ref = Oracle(g); ref.set_reparametrization(M.REPAM_ANISOTROPIC)
lb0 = ref.LowerBound()
ref.ComputePass(passes)
lb_ref = ref.LowerBound()
part_of = MG.graph_partition(n, ei, ej, world)
cut = float((part_of[ei] != part_of[ej]).mean())
print(f"G({n},{m}) L={L}, {world} parts, cut {100 * cut:.1f} %, {passes} passes: unpartitioned LB {lb0:.3f} -> {lb_ref:.3f}  ({time.time() - t0:.0f} s)", flush=True)
for share in shares:
    MG.BOUNDARY_SHARE = share
    parts = MG.partition_mrf(n, L, ei, ej, part_of, world, g.dual_data[: n * L], tables=g.const_data)
    sweeps = []
    variant = os.environ.get("GAP_VARIANT", "")
    for p in parts:
        if variant.startswith("reserve"):             # main sweeps keep back the cut messages' share of a boundary variable
            nf = p.model.n_factors
            deg = np.bincount(p.model.m_left, minlength=nf).astype(np.float64)
            k = np.bincount(p.in_unary, minlength=nf).astype(np.float64)
            sc = np.ones(nf)
            has = k > 0
            alpha = float(variant.split(":")[1]) if ":" in variant else 1.0
            sc[has] = 1.0 - alpha * k[has] / (deg[has] + k[has])
            p.main_send_scale = sc
        dual = p.model.dual_data.copy()
        eng = OracleEngine(p.model, dual)
        cls = MG.PartitionedSweep
        if variant.startswith("double"):              # two boundary steps after each directional sweep
            class cls(MG.PartitionedSweep):
                def program(self, n):
                    return [step for _ in range(n) for step in (("run", "F"), ("boundary",), ("boundary",), ("run", "B"), ("boundary",), ("boundary",))]
        if variant.startswith("before"):              # the boundary step BEFORE each directional sweep (same number of exchanges)
            class cls(MG.PartitionedSweep):
                def program(self, n):
                    return [step for _ in range(n) for step in (("boundary",), ("run", "F"), ("boundary",), ("run", "B"))]
        if variant.startswith("pre"):                 # a boundary step before AND after each directional sweep
            class cls(MG.PartitionedSweep):
                def program(self, n):
                    return [("boundary",)] + [step for _ in range(n) for step in (("run", "F"), ("boundary",), ("run", "B"), ("boundary",))]
        if variant.startswith("dir"):                 # direction-aware boundary: pull after the forward sweep, push after the backward one
            cls = DirectionalSweep
            nf = p.model.n_factors
            deg = np.bincount(p.model.m_left, minlength=nf).astype(np.float64)
            k = np.bincount(p.in_unary, minlength=nf).astype(np.float64)
            alpha = float(variant.split(":")[1]) if ":" in variant else 1.0
            sc = np.ones(nf); has = k > 0
            sc[has] = 1.0 - alpha * k[has] / (deg[has] + k[has])
            p.main_send_scale_backward = sc
        sweeps.append(cls(torch, p, eng, torch.from_numpy(dual), M.REPAM_ANISOTROPIC, None, every, float(os.environ.get("GAP_RESERVE", MG.BOUNDARY_RESERVE))))
    MG.run_lockstep(sweeps, passes)
    lb = sum(s.local_lower_bound() for s in sweeps)
    print(f"  share {share:.3f} every {every}: LB {lb:.3f}  gap {100 * (lb_ref - lb) / abs(lb_ref):.3f} %  ({time.time() - t0:.0f} s)", flush=True)

sched, lparts = LS.lockstep_mrf(n, L, ei, ej, part_of, world, M.REPAM_ANISOTROPIC, g.dual_data[: n * L], g.const_data)
lsw = []
for p in lparts:
    dual = p.model.dual_data.copy()
    lsw.append(LS.LockstepSweep(torch, p, sched, OracleEngine(p.model, dual), torch.from_numpy(dual)))
LS.run_lockstep(lsw, passes)
lb = sum(s.local_lower_bound() for s in lsw)
n_halo = sum(1 for s in sched.program(passes) if s[0] == "halo")
print(f"  lock step ({sched.n_levels[0]} + {sched.n_levels[1]} levels per pass, {n_halo / passes:.1f} exchanges per pass): LB {lb:.3f}  gap {100 * (lb_ref - lb) / abs(lb_ref):.3e} %  ({time.time() - t0:.0f} s)", flush=True)
