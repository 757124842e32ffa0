"""Deep schedules: the chain executor (one persistent launch per pass, dependency flags between workgroups) against
hipGraph replay of one launch per level.  python tools/chain_probe.py [grid] [labels] [dense|potts] [passes]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, synthetic as S
import bench as B

g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pw = sys.argv[3] if len(sys.argv) > 3 else "dense"
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 10
torch.cuda.set_device(0)
sp = torch.cuda.current_stream().cuda_stream
out = {}
duals = {}
for name, env, no_mb in (("chain", "0", None), ("chain_flags_only", "0", "1"), ("graph", "1", None)):
    os.environ["LPMP_NO_CHAIN"] = env
    os.environ.pop("LPMP_NO_MAILBOX", None)
    if no_mb: os.environ["LPMP_NO_MAILBOX"] = no_mb          # every hand-over through a completion flag (plan.cpp, mailbox)
    m, const, dual = B.build_device_grid(torch, g, g, L, pw, "row_major", 1, E, S, sp)
    e = E.Engine(0); e.set_stream(sp)
    e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
    e.set_reparametrization(M.REPAM_ANISOTROPIC)
    info = e.plan.pass_schedule_info(M.REPAM_ANISOTROPIC)
    e.compute_pass(2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        e.compute_pass(1)
    e.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / passes
    out[name] = {"ms_per_pass": dt * 1e3, "us_per_level": dt * 1e6 / info["n_levels"], "levels": info["n_levels"], "lb": e.lower_bound()}
    duals[name] = dual.clone()
    e.close(); del e, const, dual
out["bit_identical"] = bool(torch.equal(duals["chain"], duals["graph"])) and bool(torch.equal(duals["chain_flags_only"], duals["graph"]))
print(json.dumps(out))
