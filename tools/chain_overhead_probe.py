"""What the chain executor costs on a throughput-bound schedule: the colour-major C3 pass (3 launches: H, W, T) as plain
launches against the same pass as one persistent chain launch (LPMP_CHAIN_MIN=1).  python tools/chain_overhead_probe.py [grid] [passes]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
out = {}
for name, cmin in (("plain", "1000000"), ("chain", "1")):
    code = f"""
import os, sys, time, json
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import torch
from lp_mp_amd import engine as E, model as M, synthetic as S
import bench as B
torch.cuda.set_device(0)
sp = torch.cuda.current_stream().cuda_stream
m, const, dual = B.build_device_grid(torch, {g}, {g}, 32, "dense", "colour_major", 1, E, S, sp)
e = E.Engine(0); e.set_stream(sp)
e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
e.set_reparametrization(M.REPAM_ANISOTROPIC)
for _ in range(3): e.compute_pass(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range({passes}): e.compute_pass(1)
e.synchronize(); torch.cuda.synchronize()
print(json.dumps({{"ms_per_pass": (time.perf_counter() - t0) / {passes} * 1e3, "lb": e.lower_bound()}}))
"""
    import subprocess
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LPMP_CHAIN_MIN=cmin), capture_output=True, text=True)
    out[name] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-400:]}
print(json.dumps(out))
