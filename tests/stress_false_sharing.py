"""stress for sub-cache-line factors: tiny label counts (vectors of 16-40 B, several factors per 128-B line), deep
schedules whose levels scatter over the XCDs; duals against the oracle after EVERY pass.
python tests/stress_false_sharing.py PASSES"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lp_mp_amd import engine as E, model as M, synthetic as S
from oracle.binding import Oracle

passes = int(sys.argv[1])
bad = 0; t0 = time.time()
cases = [("dense", 2, "row_major", 24), ("dense", 3, "row_major", 20), ("potts", 5, "row_major", 24), ("dense", 2, "colour_major", 40),
         ("potts", 3, "colour_major", 40)]
for pw, L, order, n in cases:
    m = S.grid_model(n, n, L, pairwise=pw, order=order, seed=L, compute_primal=True)
    for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
        eng = E.Engine(0); o = Oracle(m)
        eng.upload(m); eng.set_reparametrization(mode); o.set_reparametrization(mode)
        first_bad = None
        for p in range(passes):
            if p % 3 == 2:
                eng.forward_pass(); eng.backward_pass(); o.ComputeForwardPass(); o.ComputeBackwardPass()
            else:
                eng.compute_pass(1); o.ComputePass(1)
            if not np.array_equal(eng.download_duals(), o.duals()):
                bad += 1
                if first_bad is None:
                    first_bad = p; d = eng.download_duals(); r = o.duals()
                    w = np.nonzero(d != r)[0]
                    print("MISMATCH", pw, L, order, "mode", mode, "pass", p, "elements", w[:8], "of", d.shape[0], "max diff", np.abs(d - r).max())
                eng.upload_duals(o.duals())
        eng.close()
print("done", passes, "passes x", len(cases) * 2, "runs,", bad, "mismatching passes, %.0f s" % (time.time() - t0))
