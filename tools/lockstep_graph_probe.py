"""C4-shaped random graph on ONE GPU, `parts` parts as separate engines stepped in process (no RCCL): the lock-step sweep
(lockstep.py: the unpartitioned sweep itself) against the partitioned sweep with boundary steps (multi_gpu.py) and the
unpartitioned engine — ms per pass and part, exchanges per pass, dual-bound gap after `passes` passes.
    python tools/lockstep_graph_probe.py [n] [m] [labels] [parts] [passes] [colour_major|index]
The lock-step sweep and its unpartitioned reference run in the given variable order (default colour_major: one level per colour);
the boundary-step schedule keeps the generator's index order (its own reference too)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, multi_gpu as MG, lockstep as LS, synthetic as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 4
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 8
order = sys.argv[6] if len(sys.argv) > 6 else "colour_major"
mode = M.REPAM_ANISOTROPIC
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
ei0, ej0 = S.counter_graph_edges(n, m, 1)
part0 = MG.graph_partition(n, ei0, ej0, parts)
rank_of = None
if order == "colour_major":
    from lp_mp_amd import ordering as O
    t0 = time.perf_counter(); rank_of = O.colour_major_order(n, ei0, ej0, seed=1); t_order = time.perf_counter() - t0
ei, ej = S.counter_graph_edges(n, m, 1, rank_of)
part_of = part0 if rank_of is None else MG.graph_partition(n, ei, ej, parts)
out = {"n": n, "m": m, "labels": L, "parts": parts, "passes": passes, "order": order, "cut_fraction": round(float((part_of[ei] != part_of[ej]).mean()), 4)}
if rank_of is not None:
    out["ordering_s"] = round(t_order, 2)

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

# the unpartitioned engines: 2 x passes passes (the timed call follows the warm-up call, as below)
def unpartitioned(rank):
    g = S.counter_graph_model(n, m, L, 1, device_const=True, rank=rank)
    const0 = torch.empty(m * L * L, dtype=torch.float64, device=dev)
    dual0 = torch.zeros(n * L + m * 2 * L, dtype=torch.float64, device=dev)
    E.synth_fill(const0.data_ptr(), const0.numel(), 1, n * L, stream); E.synth_fill(dual0.data_ptr(), n * L, 1, 0, stream); torch.cuda.synchronize()
    e0 = E.Engine(0); e0.set_stream(stream); e0.upload(g, const_dev=const0.data_ptr(), dual_dev=dual0.data_ptr(), keep=(const0, dual0)); e0.set_reparametrization(mode)
    info = [e0.plan.schedule_info(d, mode)["n_levels"] for d in (0, 1)]
    e0.compute_pass(passes); e0.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.compute_pass(passes); e0.synchronize(); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / passes * 1e3
    lb = e0.lower_bound(); e0.close()
    return ms, lb, info
ms_i, lb_index, lev_i = unpartitioned(None)
out["unpartitioned_index_order"] = {"ms_per_pass": round(ms_i, 3), "levels": lev_i, "lb": lb_index}
if rank_of is not None:
    ms_c, lb_ref, lev_c = unpartitioned(rank_of)
    out["unpartitioned_" + order] = {"ms_per_pass": round(ms_c, 3), "levels": lev_c, "lb": lb_ref}
else:
    lb_ref = lb_index
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
# where the time goes: the runs alone (no exchange: the duals are garbage afterwards, nothing is read from them), one part alone,
# and the exchanges alone (pack + in-process shuffle + unpack of every halo step)
def runs_only(sweeps):
    for step in prog:
        if step[0] == "run":
            for s_ in sweeps: s_.run(step[1])
def halos_only():
    for step in prog:
        if step[0] == "halo":
            packed = [s_.halo_pack(step[1], step[2]) for s_ in sw]
            offs = [np.concatenate([[0], np.cumsum(p_[1])]) for p_ in packed]
            for dst, s_ in enumerate(sw):
                s_.halo_unpack(step[1], torch.cat([packed[src][0][offs[src][dst]: offs[src][dst + 1]] for src in range(parts)]), step[2])
out["lockstep"]["ms_runs_only_per_pass_and_part"] = round(timed(lambda: runs_only(sw)) / parts, 3)
out["lockstep"]["ms_runs_only_one_part_alone"] = round(timed(lambda: runs_only(sw[:1])), 3)
out["lockstep"]["ms_exchanges_only_per_pass_and_part"] = round(timed(halos_only) / parts, 3)
out["lockstep"]["halo_MB_per_pass_and_part"] = round(sum(int(s_.halo_pack(st[1], st[2])[0].numel()) for st in prog if st[0] == "halo" for s_ in sw) * 8 / 1e6 / passes / parts, 2)
for s in sw: s.engine.close()
del sw, keep

pparts = MG.partition_mrf(n, L, ei0, ej0, part0, parts, only=None, stream_seed=1)
sw, keep = [], []
for p in pparts:
    e, dual, const = device_part(p); keep.append((dual, const))
    sw.append(MG.PartitionedSweep(torch, p, e, dual, mode, None, "sweep", MG.BOUNDARY_RESERVE))
ms = timed(lambda: MG.run_lockstep(sw, passes))
lb = sum(s.local_lower_bound() for s in sw)
out["boundary_steps"] = {"ms_per_pass_and_part": round(ms / parts, 3), "exchanges_per_pass": 4, "order": "index", "gap_percent": 100 * (lb_index - lb) / abs(lb_index), "lb": lb}
print(json.dumps(out))
