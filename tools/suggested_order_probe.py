"""A grid inserted row by row, run in the order the engine suggests for it (lpmp_plan_suggest_order applied as a chain of relations),
against the same grid built in colour-major order: ms per pass for joined passes and for single passes, whether consecutive passes
join.   python tools/suggested_order_probe.py [grid] [labels] [passes]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, synthetic as S

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mode = M.REPAM_ANISOTROPIC
torch.cuda.set_device(0); dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
n, n_e = G * G, len(S.grid_edges(G, G)[0])

def run(m, name):
    const = torch.empty(n_e * L * L, dtype=torch.float64, device=dev)
    dual = torch.zeros(n * L + n_e * 2 * L, dtype=torch.float64, device=dev)
    E.synth_fill(const.data_ptr(), const.numel(), 1, n * L, stream); E.synth_fill(dual.data_ptr(), n * L, 1, 0, stream); torch.cuda.synchronize()
    e = E.Engine(0); e.set_stream(stream)
    t0 = time.perf_counter()
    e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual)); e.set_reparametrization(mode)
    e.prepare_passes(passes); e.prepare_passes(1); e.synchronize()
    setup = time.perf_counter() - t0
    e.compute_pass(passes); e.synchronize()
    t0 = time.perf_counter(); e.compute_pass(passes); e.synchronize(); joined = (time.perf_counter() - t0) / passes * 1e3
    for _ in range(3): e.compute_pass(1)
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes): e.compute_pass(1)
    e.synchronize(); single = (time.perf_counter() - t0) / passes * 1e3
    out = {"order": name, "grid": G, "labels": L, "levels": [e.plan.schedule_info(d, mode)["n_levels"] for d in (0, 1)], "pass_rotates": e.plan.pass_rotates(mode),
           "ms_per_pass_joined": round(joined, 3), "ms_per_pass_single_calls": round(single, 3), "setup_s": round(setup, 2), "lower_bound": e.lower_bound()}
    e.close()
    print(json.dumps(out), flush=True)

run(S.grid_model(G, G, L, order="colour_major", device_const=True), "colour_major (as built)")
rm = S.grid_model(G, G, L, order="row_major", device_const=True)
t0 = time.perf_counter(); rank, k = E.Plan(rm).suggest_order(0); t_s = time.perf_counter() - t0
print(json.dumps({"suggest_order_s": round(t_s, 2), "colours": k}), flush=True)
run(rm.with_factor_order(rank), "row_major insertion, suggested order")
if len(sys.argv) > 4:
    run(rm, "row_major (as built)")
