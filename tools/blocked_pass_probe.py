"""Joined passes as one persistent chain launch with the Infinity-Cache ticket order: ms per pass on C3 for several
(bands, lag, depth) against one launch per step.  python tools/blocked_pass_probe.py [grid] [passes] [configs...]
config = bands:lag:depth (0 bands = automatic; lag / depth "a" = the engine's own choice), tT:depth (tiled order, T blocks per tile),
or "off" """
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfgs = sys.argv[3:] or ["off", "0:2:4", "0:2:2", "0:2:3", "0:2:6", "512:2:4", "2048:2:4", "1024:3:4", "1024:2:8"]
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
t0 = time.perf_counter(); e.prepare_passes(3); e.prepare_passes({passes}); prep = time.perf_counter() - t0
e.compute_pass(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
e.compute_pass({passes})
e.synchronize(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
b = dual.view(torch.int64)
t1 = time.perf_counter()
for _ in range(5): e.compute_pass(1)
e.synchronize(); torch.cuda.synchronize()
print(json.dumps({{"ms_per_pass": dt / {passes} * 1e3, "ms_single_pass_calls": (time.perf_counter() - t1) / 5 * 1e3, "prepare_s": prep, "lb": e.lower_bound(), "dual_sum": int(b.sum().item())}}))
"""
for c in cfgs:
    env = dict(os.environ, LPMP_ROT_VERBOSE="1")
    if c == "off":
        env["LPMP_NO_BLOCKED_PASSES"] = "1"
    elif c.startswith("t"):
        t, d = c[1:].split(":")
        env.update(LPMP_ROT_TILES=t, LPMP_ROT_DEPTH=d)
    else:
        b, l, d = c.split(":")
        env.update(LPMP_ROT_BANDS=b)
        if l != "a": env.update(LPMP_ROT_LAG=l)
        if d != "a": env.update(LPMP_ROT_DEPTH=d)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    out = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else {"error": r.stderr[-400:]}
    out["chain"] = [l[6:] for l in r.stderr.splitlines() if l.startswith("lpmp: ")]
    print(c, json.dumps(out), flush=True)
