"""GPU stress (not collected by pytest): random models of every kind — grids inserted row by row, random graphs, C5-style and multicut
models (tests/test_graph_host._random_models), models of every message schedule (tests/test_fuzz_gpu.random_model) — run in the order
lpmp_plan_suggest_order gives for them (applied as a chain of relations): duals and bound of the oracle in that order, bit for bit,
plain and residual sends, joined passes included; and no sweep has more dependent levels than colours.
    python tests/stress_suggested_order.py [minutes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lp_mp_amd import engine as E, model as M
from oracle.binding import Oracle
from tests.test_graph_host import _random_models
from tests.test_fuzz_gpu import random_model

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
t_end = time.time() + 60 * minutes
seed, n = 50000, 0
eng = E.Engine(0)
while time.time() < t_end:
    m = _random_models(seed) if seed % 2 == 0 else random_model(np.random.default_rng(seed))
    rank, k = E.Plan(m).suggest_order(seed)
    m2 = m.with_factor_order(rank)
    o = Oracle(m2)
    eng.upload(m2)
    rng = np.random.default_rng(seed)
    for rtype in (0, 1):
        o.set_reparametrization_type(rtype); eng.set_reparametrization_type(rtype)
        mode = [M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM][int(rng.integers(0, 4))]
        o.set_reparametrization(mode); eng.set_reparametrization(mode)
        assert max(eng.plan.schedule_info(d, mode)["n_levels"] for d in (0, 1)) <= max(k, 1), (seed, k)
        for npass in rng.integers(1, 4, 2):
            o.ComputePass(int(npass)); eng.compute_pass(int(npass))
            assert np.array_equal(eng.download_duals(), o.duals()), (seed, rtype, mode)
        assert abs(eng.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound())), seed
    seed += 1; n += 1
eng.close()
print(f"stress_suggested_order: {n} random models in the suggested order, {minutes} minutes: 0 mismatches")
