"""ms per pass of the unpartitioned C4-style random graph (LPMP_ENGINE_SO selects a build)
python tools/graph_time.py [n] [m] [labels] [passes]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lp_mp_amd import engine as E, model as M, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 8
torch.cuda.set_device(0); dev = torch.device("cuda:0"); sp = torch.cuda.current_stream().cuda_stream
g = S.counter_graph_model(n, m, L, 1, device_const=True)
const = torch.empty(m * L * L, dtype=torch.float64, device=dev)
dual = torch.zeros(n * L + m * 2 * L, dtype=torch.float64, device=dev)
E.synth_fill(const.data_ptr(), const.numel(), 1, n * L, sp); E.synth_fill(dual.data_ptr(), n * L, 1, 0, sp); torch.cuda.synchronize()
e = E.Engine(0); e.set_stream(sp); e.upload(g, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual)); e.set_reparametrization(M.REPAM_ANISOTROPIC)
e.compute_pass(passes); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); e.compute_pass(passes); e.synchronize(); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / passes * 1e3)
ci = e.plan.chain_info(-1, M.REPAM_ANISOTROPIC)
print(json.dumps({"so": os.path.basename(os.environ.get("LPMP_ENGINE_SO", "default")), "ms_per_pass": [round(t, 3) for t in ts], "lb": e.lower_bound(), "chain": ci}))
