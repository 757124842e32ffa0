"""ms per pass of a row-major grid on the chain executor (one leg of tools/chain_probe.py; LPMP_ENGINE_SO selects a build)
python tools/row_major_time.py [grid] [labels] [dense|potts] [passes]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lp_mp_amd import engine as E, model as M, synthetic as S
import bench as B
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pw = sys.argv[3] if len(sys.argv) > 3 else "dense"
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 20
torch.cuda.set_device(0)
sp = torch.cuda.current_stream().cuda_stream
m, const, dual = B.build_device_grid(torch, g, g, L, pw, "row_major", 1, E, S, sp)
e = E.Engine(0); e.set_stream(sp)
e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
e.set_reparametrization(M.REPAM_ANISOTROPIC)
e.compute_pass(2); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(passes): e.compute_pass(1)
    e.synchronize(); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / passes * 1e3)
print(json.dumps({"so": os.path.basename(os.environ.get("LPMP_ENGINE_SO", "default")), "ms_per_pass": [round(t, 3) for t in ts], "lb": e.lower_bound()}))
