"""MPLP-style schedule (pairwise factors are the updated ones: `right` schedule) on a dense grid: generic kernel speed"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lp_mp_amd import engine as E, synthetic as S, model as M
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mt = [M.MsgType(0, 1, M.SCHED_RIGHT, 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, M.SCHED_RIGHT, 0, 1, M.M_UNARY_PAIRWISE, 1)]
b = M.ModelBuilder(2, mt)
n = H * W
var = S.grid_variable_order(H, W, "colour_major").reshape(-1)
a, bb = S.grid_edges(H, W)
i, j = np.minimum(var[a], var[bb]), np.maximum(var[a], var[bb])
u = b.add_vector_factors(0, S.u01(n * L, 1, 0).reshape(n, L))
p = b.add_dense_pairwise(1, S.u01(len(a) * L * L, 1, n * L).reshape(-1, L, L))
b.add_interleaved_messages(np.tile(np.array([0, 1], np.int32), len(a)), np.stack([u[i], u[j]], 1).reshape(-1), np.repeat(p, 2))
b.add_relations(np.stack([u[i], p], 1).reshape(-1), np.stack([p, u[j]], 1).reshape(-1))
m = b.finish()
e = E.Engine(0); e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
info = [e.plan.schedule_info(d, 0) for d in (0, 1)]
print([e.plan.schedule_classes(d, 0) for d in (0, 1)], [x["n_levels"] for x in info])
lb0 = e.lower_bound(); e.compute_pass(2); e.synchronize()
t0 = time.perf_counter(); e.compute_pass(5); e.synchronize(); dt = (time.perf_counter() - t0) / 5
print("%dx%d L=%d right schedule: %.3f ms per pass, %.0f GB/s algorithmic, LB %.3f -> %.3f" %
      (H, W, L, dt * 1e3, sum(x["algorithmic_bytes"] for x in info) / dt / 1e9, lb0, e.lower_bound()))
