"""Temporal blocking through the Infinity Cache, first experiment: the colour-major C3 pass (H, W, T) as one chain launch
whose tickets are taken in a skewed band order (LPMP_CHAIN_BANDS) so that T re-reads the tables W just read.
python tools/mall_blocking_probe.py [grid] [passes]"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
code = f"""
import os, sys, time, json
sys.path.insert(0, {ROOT!r})
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
b = dual.view(torch.int64)
print(json.dumps({{"ms_per_pass": (time.perf_counter() - t0) / {passes} * 1e3, "lb": e.lower_bound(), "dual_sum": int(b.sum().item())}}))
"""
out = {}
for name, env in (("plain", {"LPMP_CHAIN_MIN": "1000000"}), ("chain", {"LPMP_CHAIN_MIN": "1"}),
                  ("chain_bands256", {"LPMP_CHAIN_MIN": "1", "LPMP_CHAIN_BANDS": "256"}),
                  ("chain_bands512", {"LPMP_CHAIN_MIN": "1", "LPMP_CHAIN_BANDS": "512"}),
                  ("chain_bands512_nt0", {"LPMP_CHAIN_MIN": "1", "LPMP_CHAIN_BANDS": "512", "LPMP_NT": "0"}),
                  ("chain_bands1024_lag3_nt0", {"LPMP_CHAIN_MIN": "1", "LPMP_CHAIN_BANDS": "1024", "LPMP_CHAIN_LAG": "3", "LPMP_NT": "0"})):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    out[name] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:]}
    print(name, out[name], flush=True)
