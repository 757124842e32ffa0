"""Generates tests/golden/c5_small.npz: the CPU oracle on the model of `bench.py --workload c5 --c5-small` (BASELINE.json configs[4]
in miniature: 64 x 64 Potts grid with 8 labels + 2 000 binary edge variables, 900 triplet and 400 quadruple labeling-list
factors, one factor graph; anisotropic weights) in the edge-variable orders bench.py offers (--c5-order index / colour_major /
suggested: the index model in the order lpmp_plan_suggest_order gives for it).

    python tests/golden/make_c5_small.py

Stored per order: the lower bound after 0 ... 32 passes and two exact checksums of the packed duals (wrapping uint64 sums over
the IEEE bit patterns, plain and position-weighted: tests/golden/make_c3_full.py), which bench.py's `oracle_check` recomputes
from the state the timed passes leave on the device — on several ranks every rank adds up the factors it owns.
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench                                         # noqa: E402  (the model is built by bench.py's own function)
from lp_mp_amd import model as M, synthetic as S     # noqa: E402
from oracle.binding import Oracle                    # noqa: E402
from tests.golden.make_c3_full import dual_checksums  # noqa: E402

PASSES = list(range(0, 33))


def main():
    out = {"passes": np.array(PASSES)}
    for order in ("index", "colour_major", "suggested"):
        args = types.SimpleNamespace(c5_small=True, c5_labels=8, c5_window=64, c5_order=order, c5_grid=512, c5_edge_vars=150000,
                                     c5_triplets=70000, c5_quads=30000)
        o = Oracle(bench.c5_global_model(args, S))
        o.set_reparametrization(M.REPAM_ANISOTROPIC)
        lb, s0, s1 = [], [], []
        done = 0
        for n in PASSES:
            if n > done:
                o.ComputePass(n - done); done = n
            a, b = dual_checksums(o.duals())
            lb.append(o.LowerBound()); s0.append(a); s1.append(b)
        out[f"lb_{order}"] = np.array(lb)
        out[f"dual_sum_{order}"] = np.array(s0, np.uint64)
        out[f"dual_wsum_{order}"] = np.array(s1, np.uint64)
        print(order, "lb after 0 / 8 / 32 passes:", lb[0], lb[8], lb[32], flush=True)
    np.savez(os.path.join(ROOT, "tests", "golden", "c5_small.npz"), **out)


if __name__ == "__main__":
    main()
