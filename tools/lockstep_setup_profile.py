"""where the host time of a lock-step part's setup goes (cProfile of lockstep.lockstep_mrf for one rank of `world` on the C4-shaped
graph; no GPU work).   python tools/lockstep_setup_profile.py [n] [m] [world]"""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lp_mp_amd import synthetic as S, lockstep as LS, multi_gpu as MG, engine as E, model as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
t = time.perf_counter()
ei0, ej0 = S.counter_graph_edges(n, m, 1)
rank, k = E.graph_colour_major_order(n, ei0, ej0, 1)
ei, ej = S.counter_graph_edges(n, m, 1, rank)
print("edges + order %.2f s" % (time.perf_counter() - t)); t = time.perf_counter()
part = MG.graph_partition(n, ei, ej, world)
print("partition %.2f s" % (time.perf_counter() - t))
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
sched, parts = LS.lockstep_mrf(n, 16, ei, ej, part, world, M.REPAM_ANISOTROPIC, only=3, stream_seed=1)
print("lockstep_mrf %.2f s" % (time.perf_counter() - t))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
