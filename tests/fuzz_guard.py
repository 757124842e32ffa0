"""hunt for the one-in-48000 oracle argument error seen in a long fuzz run: the primal test family with copies of the
row arrays taken before the engine call and compared after it; dumps what changed.  python tests/fuzz_guard.py FIRST COUNT"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import test_fuzz_gpu as T
from lp_mp_amd import engine as E
from oracle.binding import Oracle

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(13000 + seed)
    m = T.random_mrf(rng, primal=True) if seed % 2 else T.random_mrf_any_labels(rng, primal=True)
    eng = E.Engine(0)
    try:
        for mode in T.MODES:
            o = Oracle(m); o.set_reparametrization(mode)
            eng.upload(m); eng.set_reparametrization(mode)
            for step in range(6):
                rows = T.random_rows(rng, None, o, m)
                keep = [r.copy() for r in rows]
                off0, ent0 = o.msg_lists()
                eng.compute_pass_custom(*rows)
                changed = [i for i, (a, b) in enumerate(zip(rows, keep)) if not np.array_equal(a, b)]
                off1, ent1 = o.msg_lists()
                if changed or not (np.array_equal(off0, off1) and np.array_equal(ent0, ent1)):
                    bad += 1; print("CORRUPTION seed", seed, "mode", mode, "step", step, "rows changed", changed,
                                    "oracle lists changed", not np.array_equal(ent0, ent1))
                try:
                    o.compute_pass_custom(*rows)
                except RuntimeError as e:
                    bad += 1; print("ORACLE ERROR seed", seed, mode, step, e, "rows changed", changed)
                    o.compute_pass_custom(*keep)
                if not np.array_equal(eng.download_duals(), o.duals()):
                    bad += 1; print("DUAL MISMATCH seed", seed, mode, step)
    finally:
        eng.close()
print("done", count, "seeds,", bad, "events")
