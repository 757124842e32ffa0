"""C4-shaped random graph on ONE GPU, `parts` parts as separate engines stepped in process (no RCCL): the lock-step sweep
(lockstep.py: the unpartitioned sweep itself) against the partitioned sweep with boundary steps (multi_gpu.py) and the
unpartitioned engine — ms per pass and part, exchanges per pass, dual-bound gap after `passes` passes.
    python tools/lockstep_graph_probe.py [n] [m] [labels] [parts] [passes]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, multi_gpu as MG, lockstep as LS, synthetic as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 4
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 8
mode = M.REPAM_ANISOTROPIC
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
ei, ej = S.counter_graph_edges(n, m, 1)
part_of = MG.graph_partition(n, ei, ej, parts)
out = {"n": n, "m": m, "labels": L, "parts": parts, "passes": passes, "cut_fraction": round(float((part_of[ei] != part_of[ej]).mean()), 4)}

def timed(run):
    run(); torch.cuda.synchronize()                      # builds the schedules
    t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes * 1e3

def device_part(p):
    mdl = p.model
    const = torch.empty(max(int(mdl.const_sizes().sum()), 2), dtype=torch.float64, device=dev)
    dual = torch.zeros(int(mdl.dual_sizes().sum()), dtype=torch.float64, device=dev)
    MG.fill_device_costs(torch, E, p, const, dual, stream)
    e = E.Engine(0); e.set_stream(stream)
    e.upload(mdl, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
    e.set_reparametrization(mode)
    return e, dual, const

# the unpartitioned engine: 2 x passes passes (the timed call follows the warm-up call, as below)
g = S.counter_graph_model(n, m, L, 1, device_const=True)
const0 = torch.empty(m * L * L, dtype=torch.float64, device=dev)
dual0 = torch.zeros(n * L + m * 2 * L, dtype=torch.float64, device=dev)
E.synth_fill(const0.data_ptr(), const0.numel(), 1, n * L, stream); E.synth_fill(dual0.data_ptr(), n * L, 1, 0, stream); torch.cuda.synchronize()
e0 = E.Engine(0); e0.set_stream(stream); e0.upload(g, const_dev=const0.data_ptr(), dual_dev=dual0.data_ptr(), keep=(const0, dual0)); e0.set_reparametrization(mode)
e0.compute_pass(passes); lb_mid = e0.lower_bound()
t0 = time.perf_counter(); e0.compute_pass(passes); e0.synchronize(); out["unpartitioned_ms_per_pass"] = round((time.perf_counter() - t0) / passes * 1e3, 3)
lb_ref = e0.lower_bound(); e0.close(); del const0, dual0

t0 = time.perf_counter()
sched, lparts = LS.lockstep_mrf(n, L, ei, ej, part_of, parts, mode, stream_seed=1)
out["lockstep_setup_s"] = round(time.perf_counter() - t0, 1)
sw, keep = [], []
for p in lparts:
    e, dual, const = device_part(p); keep.append((dual, const))
    sw.append(LS.LockstepSweep(torch, p, sched, e, dual))
ms = timed(lambda: LS.run_lockstep(sw, passes))
lb = sum(s.local_lower_bound() for s in sw)
prog = sched.program(passes)
out["lockstep"] = {"ms_per_pass_and_part": round(ms / parts, 3), "exchanges_per_pass": sum(1 for s in prog if s[0] == "halo") / passes,
                   "levels": list(sched.n_levels), "gap_percent": 100 * (lb_ref - lb) / abs(lb_ref), "lb": lb}
for s in sw: s.engine.close()
del sw, keep

pparts = MG.partition_mrf(n, L, ei, ej, part_of, parts, only=None, stream_seed=1)
sw, keep = [], []
for p in pparts:
    e, dual, const = device_part(p); keep.append((dual, const))
    sw.append(MG.PartitionedSweep(torch, p, e, dual, mode, None, "sweep", MG.BOUNDARY_RESERVE))
ms = timed(lambda: MG.run_lockstep(sw, passes))
lb = sum(s.local_lower_bound() for s in sw)
out["boundary_steps"] = {"ms_per_pass_and_part": round(ms / parts, 3), "exchanges_per_pass": 4, "gap_percent": 100 * (lb_ref - lb) / abs(lb_ref), "lb": lb}
out["lb_unpartitioned"] = lb_ref
print(json.dumps(out))
