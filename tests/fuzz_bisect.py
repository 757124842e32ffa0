import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)  # debugging aid for test_fuzz_gpu.py
import numpy as np
from tests.test_fuzz_gpu import random_model
from lp_mp_amd import engine as E, model as M
from oracle.binding import Oracle
seed = int(sys.argv[1]); mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(1000 + seed); m = random_model(rng)
o = Oracle(m); o.set_reparametrization(mode)
e = E.Engine(0); e.upload(m); e.set_reparametrization(mode)
def cmp(tag):
    d, do = e.download_duals(), o.duals()
    bad = np.nonzero(d != do)[0]
    off = m.dual_offsets()
    fs = sorted(set(int(np.searchsorted(off, b, side="right") - 1) for b in bad))
    print(tag, "equal" if bad.size == 0 else "DIFF factors %s kinds %s types %s" % (fs[:10], [int(m.f_kind[f]) for f in fs[:10]], [int(m.f_type[f]) for f in fs[:10]]))
e.forward_pass(); o.ComputeForwardPass(); cmp("fwd1")
e.backward_pass(); o.ComputeBackwardPass(); cmp("bwd1")
e.compute_pass(1); o.ComputePass(1); cmp("pass(1)")
e.compute_pass(2); o.ComputePass(2); cmp("pass(2)")
