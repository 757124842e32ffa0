"""GPU stress (not collected by pytest): the randomised lock-step check of tests/test_lockstep.py for many more seeds (MRFs and labeling-list models), and random
deep dense / Potts chains (banded graphs with random offsets and label counts) through the mailbox against the flags-only
executor and the oracle.   python tests/stress_lockstep_mailbox.py [minutes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lp_mp_amd import engine as E, model as M, synthetic as S
from oracle.binding import Oracle
from tests.test_lockstep import test_lockstep_random_graphs_partitions_and_modes_on_device as lockstep_case
from tests.test_lockstep import test_lockstep_random_general_models_partitions_and_modes_on_device as general_case

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
t_end = time.time() + 60 * minutes
n_ls = n_mb = n_gen = 0
seed = 1000
while time.time() < t_end:
    lockstep_case(seed); n_ls += 1
    general_case(seed); n_gen += 1
    rng = np.random.default_rng(seed)
    L = int(rng.choice([2, 3, 4, 7, 8, 16, 21, 32])); n = int(rng.integers(60, 500))
    offs = sorted(set(int(x) for x in rng.integers(1, 24, int(rng.integers(1, 5)))))
    ei = np.concatenate([np.arange(0, n - o) for o in offs]); ej = np.concatenate([np.arange(0, n - o) + o for o in offs])
    o = np.lexsort((ej, ei)); ei, ej = ei[o], ej[o]
    potts = rng.uniform() < 0.4
    kw = dict(potts=S.u01(ei.shape[0], seed, n * L)) if potts else dict(tables=S.u01(ei.shape[0] * L * L, seed, n * L))
    m = S.mrf_model(n, L, ei, ej, S.u01(n * L, seed, 0), compute_primal=True, **kw)
    ref = Oracle(m)
    mode = [M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM][int(rng.integers(0, 4))]
    rtype = int(rng.integers(0, 2))
    engines = []
    for env in (None, "1"):
        os.environ.pop("LPMP_NO_MAILBOX", None)
        if env: os.environ["LPMP_NO_MAILBOX"] = env
        e = E.Engine(0); e.upload(m); e.set_reparametrization_type(rtype); e.set_reparametrization(mode); engines.append(e)
    os.environ.pop("LPMP_NO_MAILBOX", None)
    ref.set_reparametrization_type(rtype); ref.set_reparametrization(mode)
    for k in rng.integers(1, 4, 2):
        ref.ComputePass(int(k)); ref.ComputeForwardPass()
        for e in engines:
            e.compute_pass(int(k)); e.forward_pass()
            assert np.array_equal(e.download_duals(), ref.duals()), (seed, L, offs, potts, mode, rtype)
            assert abs(e.lower_bound() - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound()))
    ref.ComputePassAndPrimal(1)
    for e in engines:
        e.compute_pass_and_primal(1)
        assert np.array_equal(e.download_primal(), ref.primal()), seed
        e.close()
    n_mb += 1; seed += 1
print(f"stress_lockstep_mailbox: {n_ls} lock-step cases on random MRFs, {n_gen} on random labeling-list models, {n_mb} mailbox chains against flags-only and the oracle, {minutes} minutes: 0 mismatches")
