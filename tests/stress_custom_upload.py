"""stress of the upload-then-run path (iterator-range passes compile, upload and run a fresh schedule per call): many
random row sets on a few small models, duals against the oracle after every call.
python tests/stress_custom_upload.py N_MODELS CALLS_PER_MODEL"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import test_fuzz_gpu as T
from lp_mp_amd import engine as E
from oracle.binding import Oracle

n_models, calls = int(sys.argv[1]), int(sys.argv[2])
bad = 0; t0 = time.time()
for k in range(n_models):
    rng = np.random.default_rng(500 + k)
    m = T.random_mrf(rng, primal=True) if k % 2 else T.random_mrf_any_labels(rng, primal=True)
    eng = E.Engine(0); o = Oracle(m)
    eng.upload(m); eng.set_reparametrization(0); o.set_reparametrization(0)
    for c in range(calls):
        rows = T.random_rows(rng, None, o, m)
        eng.compute_pass_custom(*rows); o.compute_pass_custom(*rows)
        if c % 4 == 3:
            eng.compute_pass(1); o.ComputePass(1)
        if not np.array_equal(eng.download_duals(), o.duals()):
            bad += 1; print("MISMATCH model", k, "call", c)
            eng.upload_duals(o.duals())
    eng.close()
print("done", n_models * calls, "calls,", bad, "mismatches, %.0f s" % (time.time() - t0))
