"""generic-kernel throughput on a multicut-style labeling-list model (BASELINE configs[4]'s higher-order part)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lp_mp_amd import engine as E, synthetic as S, model as M
n_nodes, n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 3000, int(sys.argv[2]) if len(sys.argv) > 2 else 100000
t0 = time.time(); m = S.multicut_triangle_model(n_nodes, n_tri, seed=1); print("build %.1fs, factors %d" % (time.time() - t0, m.n_factors))
e = E.Engine(0)
t0 = time.time(); e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC); print("upload+plan %.1fs" % (time.time() - t0))
for d in (0, 1): print(d, e.plan.schedule_info(d, 0), e.plan.schedule_classes(d, 0))
lb0 = e.lower_bound(); e.compute_pass(2); e.synchronize()
t0 = time.perf_counter(); e.compute_pass(20); e.synchronize(); dt = (time.perf_counter() - t0) / 20
info = [e.plan.schedule_info(d, 0) for d in (0, 1)]
upd = sum(i["n_receives"] + i["n_sends"] for i in info)
print("ms/pass %.3f  msg-updates/s %.3e  LB %.3f -> %.3f" % (dt * 1e3, upd / dt, lb0, e.lower_bound()))
