"""C5 (grid + 100 k labeling-list factors) on one GPU: time per pass for local and global triples"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lp_mp_amd import engine as E, synthetic as S, model as M
for window, coloured in ((64, False), (64, True), (150000, False)):
    m = S.c5_model(512, 512, 8, 150000, 70000, 30000, seed=4, window=window, colour_edge_vars=coloured)
    e = E.Engine(0); e.upload(m); e.set_reparametrization(0)
    info = [e.plan.schedule_info(d, 0) for d in (0, 1)]
    t0 = time.perf_counter(); e.compute_pass(2); e.synchronize(); first = time.perf_counter() - t0
    t0 = time.perf_counter(); e.compute_pass(10); e.synchronize(); dt = (time.perf_counter() - t0) / 10
    upd = sum(i["n_receives"] + i["n_sends"] for i in info)
    print("window %d%s: levels %s, first 2 passes %.2f s, then %.3f ms per pass, %.3e msg-updates/s, LB %.3f" %
          (window, " colour-major edge variables" if coloured else "", [i["n_levels"] for i in info], first, dt * 1e3, upd / dt, e.lower_bound()))
    e.close()
