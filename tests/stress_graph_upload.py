"""stress of freshly uploaded schedules first read by graph-launched kernels: re-upload the model, run one
iterator-range pass (upload / run / free), then the first compute_pass of the engine (pass schedule upload + hipGraph
capture + launch); duals against the oracle.  python tests/stress_graph_upload.py ITERATIONS [SEED]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import test_fuzz_gpu as T
from lp_mp_amd import engine as E
from oracle.binding import Oracle

iters = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 79921
rng = np.random.default_rng(13000 + seed)
m = T.random_mrf(rng, primal=True) if seed % 2 else T.random_mrf_any_labels(rng, primal=True)
bad = 0; t0 = time.time()
eng = E.Engine(0)
for it in range(iters):
    o = Oracle(m); o.set_reparametrization(0)
    eng.upload(m); eng.set_reparametrization(0)
    if it % 2:
        eng.forward_pass_and_primal(0); o.ComputeForwardPassAndPrimal(0)
    rows = T.random_rows(rng, None, o, m)
    eng.compute_pass_custom(*rows); o.compute_pass_custom(*rows)
    eng.compute_pass(1); o.ComputePass(1)
    if not np.array_equal(eng.download_duals(), o.duals()):
        bad += 1; print("MISMATCH iteration", it)
eng.close()
print("done", iters, "iterations,", bad, "mismatches, %.0f s" % (time.time() - t0))
