"""Cost of the lock-step partitioned sweep on ONE GPU: `parts` strips of the headline grid as separate engines on the one
device, stepped in process (no RCCL) — ms per pass and part, exchanges per pass, bytes shipped; and the bound against the
partitioned sweep with boundary steps (multi_gpu.StripSweep's schedule) at a small size where the oracle is affordable.
    python tools/lockstep_probe.py [grid] [labels] [parts] [passes]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, multi_gpu as MG, lockstep as LS

g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 10
mode = M.REPAM_ANISOTROPIC
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
sweeps, keep = [], []
for r in range(parts):
    sched, p = LS.strips_lockstep_part(g, g, L, "dense", "colour_major", r, parts, mode, 1)
    m = p.model
    const = torch.empty(max(int(m.const_sizes().sum()), 2), dtype=torch.float64, device=dev)
    dual = torch.zeros(int(m.dual_sizes().sum()), dtype=torch.float64, device=dev)
    MG.fill_device_costs(torch, E, p, const, dual, stream)
    e = E.Engine(0); e.set_stream(stream)
    e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
    e.set_reparametrization(mode)
    sweeps.append(LS.LockstepSweep(torch, p, sched, e, dual)); keep.append((const, dual))
setup = time.perf_counter() - t0
LS.run_lockstep(sweeps, passes); torch.cuda.synchronize()            # builds the schedules
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    LS.run_lockstep(sweeps, passes); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / passes * 1e3)
prog = sweeps[0].sched.program(passes)
halos = [s for s in prog if s[0] == "halo"]
out = {"grid_per_part": g, "labels": L, "parts": parts, "passes_per_call": passes, "setup_s": round(setup, 2),
       "ms_per_pass_all_parts": [round(t, 3) for t in ts], "ms_per_pass_and_part": round(min(ts) / parts, 3),
       "exchanges_per_pass": len(halos) / passes, "vectors_per_exchange": int(np.mean([h[1].shape[0] for h in halos])) if halos else 0,
       "updates_per_pass_and_part": sweeps[0].updates_per_pass(), "lower_bound": sum(s.local_lower_bound() for s in sweeps)}
print(json.dumps(out))
