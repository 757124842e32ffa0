"""where the seconds before the first pass of `bench.py --workload c4` go (one GPU): stage by stage, LPMP_PLAN_TIMES=1 for the
planner's own laps.   python tools/c4_setup_times.py [n] [m] [labels]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t00 = time.perf_counter()
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, multi_gpu as MG, synthetic as S
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
m = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
out = {"n": n, "m": m, "labels": L, "threads": os.environ.get("LPMP_PLAN_THREADS"), "imports_s": round(time.perf_counter() - t00, 2)}
def lap(name, t0):
    torch.cuda.synchronize(); out[name] = round(time.perf_counter() - t0, 2); return time.perf_counter()
torch.cuda.set_device(0); dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
torch.zeros(1, device=dev); t = time.perf_counter()
ei, ej = S.counter_graph_edges(n, m, 1); t = lap("edges_s", t)
part = MG.partition_mrf(n, L, ei, ej, np.zeros(n, np.int64), 1, only=0, stream_seed=1)[0]; t = lap("part_model_s", t)
mdl = part.model
const = torch.empty(max(int(mdl.const_sizes().sum()), 2), dtype=torch.float64, device=dev)
dual = torch.zeros(int(mdl.dual_sizes().sum()), dtype=torch.float64, device=dev); t = lap("device_buffers_s", t)
MG.fill_device_costs(torch, E, part, const, dual, stream); t = lap("fill_costs_s", t)
eng = E.Engine(0); eng.set_stream(stream); t = lap("engine_create_s", t)
eng.upload(mdl, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual), rows_layout=True); t = lap("upload_plan_s", t)
eng.set_reparametrization(M.REPAM_ANISOTROPIC); t = lap("set_mode_weights_s", t)
sw = MG.PartitionedSweep(torch, part, eng, dual, M.REPAM_ANISOTROPIC, None, "sweep", MG.BOUNDARY_RESERVE); t = lap("partitioned_sweep_object_s", t)
lb = eng.lower_bound(); t = lap("first_lower_bound_s", t)
eng.compute_pass(1); eng.synchronize(); t = lap("first_pass_schedules_s", t)
eng.compute_pass(1); eng.synchronize(); t = lap("second_pass_s", t)
print(json.dumps(out))
