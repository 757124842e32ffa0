"""Generates tests/golden/oracle_regression.npz: lower-bound sequences, dual checksums and rounded labels that THIS
REPO'S ORACLE produced for a handful of seeded models.  These are NOT outputs of the reference (those are in
survey_grids.npz); they freeze the oracle's behaviour so that a later edit of oracle/lpmp_oracle.c that changes any
result — the checker every GPU parity test leans on — is noticed by the CPU suite.
Run from the repo root:  python tests/golden/make_oracle_regression.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lp_mp_amd import model as M, synthetic as S  # noqa: E402
from oracle.binding import Oracle  # noqa: E402


def cases():
    yield "grid_dense_L5_row", S.grid_model(7, 6, 5, seed=11, compute_primal=True)
    yield "grid_potts_L8_colour", S.grid_model(8, 8, 8, pairwise="potts", order="colour_major", seed=12, compute_primal=True)
    yield "random_graph_L16", S.random_graph_model(40, 120, 16, seed=13, compute_primal=True)
    yield "multicut", S.multicut_triangle_model(14, 25, seed=3)
    yield "c5_mini", S.c5_model(5, 6, 4, 40, 25, 10, seed=2, window=12)


def run(m, primal):
    out = {}
    for mode, name in ((M.REPAM_ANISOTROPIC, "anisotropic"), (M.REPAM_UNIFORM, "uniform"), (M.REPAM_DAMPED_UNIFORM, "damped")):
        for rtype in (0, 1):
            o = Oracle(m)
            o.set_reparametrization_type(rtype); o.set_reparametrization(mode)
            lbs = [o.LowerBound()]
            for _ in range(4):
                o.ComputePass(1); lbs.append(o.LowerBound())
            key = f"{name}_r{rtype}"
            out[key + "_lb"] = np.array(lbs)
            out[key + "_dual_sum_abs"] = np.array([np.abs(o.duals()).sum(), (o.duals() * np.arange(1, o.duals().shape[0] + 1)).sum()])
            if primal and rtype == 0:
                o.ComputePassAndPrimal(4)
                out[key + "_labels"] = o.primal()[:, 0].copy()
                out[key + "_primal_cost"] = np.array([o.EvaluatePrimal()])
    return out


if __name__ == "__main__":
    res = {}
    for name, m in cases():
        for k, v in run(m, primal=m.ftype_computes_primal.any()).items():
            res[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_regression.npz"), **res)
    print("wrote oracle_regression.npz with", len(res), "arrays")
