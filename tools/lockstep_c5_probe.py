"""C5 (BASELINE.json configs[4]: Potts grid + 100 k labeling-list factors of mixed arity, one factor graph) in `parts` lock-step parts
as separate engines on ONE GPU (lockstep.lockstep_model; no RCCL): ms per pass and part, exchanges per pass, bound against the
unpartitioned engine.   python tools/lockstep_c5_probe.py [parts] [passes]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, multi_gpu as MG, lockstep as LS, synthetic as S

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mode = M.REPAM_ANISOTROPIC
torch.cuda.set_device(0); dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream

def timed(run):
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes * 1e3

for window, coloured, name in ((64, False, "local triples"), (64, True, "local triples, colour-major edge variables"), (150000, False, "global triples")):
    gm = S.c5_model(512, 512, 8, 150000, 70000, 30000, seed=4, window=window, colour_edge_vars=coloured)
    e0 = E.Engine(0); e0.set_stream(stream); e0.upload(gm); e0.set_reparametrization(mode)
    ms0 = timed(lambda: e0.compute_pass(passes)); e0.synchronize(); lb0 = e0.lower_bound(); e0.close()
    t0 = time.perf_counter()
    part_of = MG.graph_partition_model(gm, parts)
    sched, lparts = LS.lockstep_model(gm, part_of, parts, mode)
    setup = time.perf_counter() - t0
    sw = []
    for p in lparts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        e = E.Engine(0); e.set_stream(stream); e.upload(p.model, dual_dev=dual.data_ptr(), keep=dual); e.set_reparametrization(mode)
        sw.append(LS.LockstepSweep(torch, p, sched, e, dual))
    ms = timed(lambda: LS.run_lockstep(sw, passes))
    lb = sum(s.local_lower_bound() for s in sw)
    prog = sched.program(passes)
    print(json.dumps({"c5": name, "parts": parts, "passes": passes, "unpartitioned_ms_per_pass": round(ms0, 3), "levels": list(sched.n_levels),
                      "lockstep_ms_per_pass_and_part": round(ms / parts, 3), "lockstep_ms_per_pass_all_parts_in_turn": round(ms, 3),
                      "exchanges_per_pass": sum(1 for s in prog if s[0] == "halo") / passes,
                      "cut_vectors": int((np.diff(sched.dest_off) > 0).sum()), "messages": int(gm.n_messages),
                      "partition_and_parts_s": round(setup, 1), "lb_unpartitioned_after_2x_passes": lb0, "lb_lockstep": lb,
                      "gap_percent": 100 * (lb0 - lb) / abs(lb0)}), flush=True)
    for s in sw: s.close(); s.engine.close()
