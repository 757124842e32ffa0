"""Where a level of the level loop spends its time (engine.cpp, LPMP_LEVEL_TRACE): C5 with local triples, backward sweep.
Stamps per level (lane 0's record): 0 level start, 1 own duals landed, 2 receives done, 3 ops done, 4 body done
(stores issued), 5 stores drained.   python tools/level_trace.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = "/tmp/level_trace.bin"
code = f"""
import sys
sys.path.insert(0, {ROOT!r})
from lp_mp_amd import engine as E, model as M, synthetic as S
m = S.c5_model(512, 512, 8, 150000, 70000, 30000, seed=4, window=64)
e = E.Engine(0); e.upload(m); e.set_reparametrization(0)
e.compute_pass(1); e.backward_pass(); e.synchronize()
"""
r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LPMP_LEVEL_TRACE=path), capture_output=True, text=True)
if r.returncode != 0:
    print(r.stderr[-1500:]); sys.exit(1)
a = np.fromfile(path, np.int64)[8:].reshape(-1, 8)
a = a[(a[:, 0] > 0) & (a[:, 5] > 0)][50:]                    # skip the first levels (cold)
us = lambda x: x * 0.01
q = lambda v: "median %.2f  p10 %.2f  p90 %.2f us" % tuple(us(np.percentile(v, [50, 10, 90])))
print(len(a), "levels")
print("level to level            :", q(np.diff(a[:, 0])))
print("start -> own duals landed :", q(a[:, 1] - a[:, 0]))
has2 = a[:, 2] > 0
print("own landed -> receives done:", q((a[:, 2] - a[:, 1])[has2]), "(%d levels with a snapshot point)" % has2.sum())
print("receives done -> ops done :", q((a[:, 3] - a[:, 2])[has2]))
print("ops done -> body done     :", q(a[:, 4] - a[:, 3]))
print("body done -> stores drained:", q(a[:, 5] - a[:, 4]))

# round 3 (staged levels: kernels.hip label_ops_body_staged / level_loop_kernel<1>): slot 1 = the peers' costs of wave 0's first
# eight records have landed, 2 = their receives are done, 4 = body done; slots 6 / 7 = the wave that runs ahead starts / ends its stage
ok = (a[:, 1] > a[:, 0]) & (a[:, 2] > a[:, 1]) & (a[:, 4] > a[:, 2])
print("staged levels with all stamps: %d of %d" % (ok.sum(), len(a)))
if ok.any():
    print("  start -> costs landed   :", q((a[:, 1] - a[:, 0])[ok]))
    print("  costs landed -> receives:", q((a[:, 2] - a[:, 1])[ok]))
    print("  receives -> body done   :", q((a[:, 4] - a[:, 2])[ok]))
st = (a[:, 6] > 0) & (a[:, 7] > a[:, 6])
if st.any(): print("run-ahead wave: start -> stage done :", q((a[:, 7] - a[:, 6])[st]))
