"""C5 (BASELINE.json configs[4]: Potts grid + 100 k labeling-list factors of mixed arity, one factor graph) in `parts` lock-step parts
as separate engines on ONE GPU (lockstep.lockstep_model; no RCCL): ms per pass and part, exchanges per pass, bound against the
unpartitioned engine — and what taking the exchanges off the critical path (LockstepSchedule.program_overlapped) can buy:
  lockstep_ms_per_pass_and_part        the plain program (every exchange between two runs)
  overlapped_ms_per_pass_and_part      the overlapped program, in process: nothing overlaps here, so this is the COST of the split
                                       (one more launch boundary per exchange)
  runs_only_ms_per_pass_and_part       the plain program with the exchanges left out (wrong duals; time only): what is left when
                                       ALL of the exchange time is hidden — the ceiling of any overlap
  exchange model                       n_exchanges * latency + bytes / bandwidth per pass and part under the stated assumptions
    python tools/lockstep_c5_probe.py [parts] [passes] [latency_us] [GBps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, multi_gpu as MG, lockstep as LS, synthetic as S

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lat_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
gbps = float(sys.argv[4]) if len(sys.argv) > 4 else 400.0
mode = M.REPAM_ANISOTROPIC
torch.cuda.set_device(0); dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream

def timed(run):
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes * 1e3

def runs_only(sweeps, n):
    for step in sweeps[0].sched.program(n):
        if step[0] == "run":
            for s in sweeps:
                s.run(step[1])

for window, coloured, name in ((64, False, "local triples"), (64, True, "local triples, colour-major edge variables"), (150000, False, "global triples")):
    gm = S.c5_model(512, 512, 8, 150000, 70000, 30000, seed=4, window=window, colour_edge_vars=coloured)
    e0 = E.Engine(0); e0.set_stream(stream); e0.upload(gm); e0.set_reparametrization(mode)
    ms0 = timed(lambda: e0.compute_pass(passes)); e0.synchronize(); lb0 = e0.lower_bound(); e0.close()
    t0 = time.perf_counter()
    part_of = MG.graph_partition_model(gm, parts)
    sched, lparts = LS.lockstep_model(gm, part_of, parts, mode)
    setup = time.perf_counter() - t0
    res = {}
    for overlap in (False, True):
        sw = []
        for p in lparts:
            dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
            e = E.Engine(0); e.set_stream(stream); e.upload(p.model, dual_dev=dual.data_ptr(), keep=dual, rows_layout=False); e.set_reparametrization(mode)
            sw.append(LS.LockstepSweep(torch, p, sched, e, dual, overlap_exchange=overlap))
        ms = timed(lambda: LS.run_lockstep(sw, passes))
        lb = sum(s.local_lower_bound() for s in sw)
        res[overlap] = (ms, lb)
        if not overlap:
            ms_runs = timed(lambda: runs_only(sw, passes))
            by = [sum(int(s._halo_plan(st[1], st[2])[2]) for st in sched.program(passes) if st[0] == "halo") * 8 / passes for s in sw]
        for s in sw: s.close(); s.engine.close()
    prog, over = sched.program(passes), sched.program_overlapped(passes)
    n_ex = sum(1 for s in prog if s[0] == "halo") / passes
    exch_model_ms = n_ex * lat_us * 1e-3 + max(by) / (gbps * 1e9) * 1e3
    t_run = ms_runs / parts
    print(json.dumps({"c5": name, "parts": parts, "passes": passes, "unpartitioned_ms_per_pass": round(ms0, 3), "levels": list(sched.n_levels),
                      "lockstep_ms_per_pass_and_part": round(res[False][0] / parts, 3), "overlapped_ms_per_pass_and_part": round(res[True][0] / parts, 3),
                      "runs_only_ms_per_pass_and_part": round(t_run, 3),
                      "exchanges_per_pass": n_ex, "exchanges_split_for_overlap_per_pass": sum(1 for s in over if s[0] == "halo_begin") / passes,
                      "exchange_bytes_per_pass_and_part_max": int(max(by)),
                      "assumed_latency_us": lat_us, "assumed_GBps": gbps, "exchange_model_ms_per_pass": round(exch_model_ms, 4),
                      "projected_ms_per_pass_plain": round(t_run + exch_model_ms, 3),
                      "ceiling_of_overlap_percent": round(100 * exch_model_ms / (t_run + exch_model_ms), 2),
                      "cut_vectors": int((np.diff(sched.dest_off) > 0).sum()), "messages": int(gm.n_messages),
                      "partition_and_parts_s": round(setup, 1), "lb_unpartitioned_after_2x_passes": lb0, "lb_lockstep": res[False][1],
                      "lb_overlapped": res[True][1], "gap_percent": 100 * (lb0 - res[False][1]) / abs(lb0)}), flush=True)
