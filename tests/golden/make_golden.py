"""Generates tests/golden/survey_grids.npz.

Inputs: the random costs of the reference runs recorded in SURVEY.md 8(c) / 8(a5) — the surveyor
ran the reference's LP<FMC> in this container on grids drawn from std::mt19937_64(12345) +
uniform_real_distribution(0,1) (unaries first, then per edge the LxL table, edges row-major
right-then-down) — re-created with oracle/gen_mt19937 (libstdc++).  Expected outputs: the lower
bounds printed by that reference run, copied from SURVEY.md (they are NOT produced by this repo).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import mt19937_u01  # noqa: E402
from lp_mp_amd import synthetic as S  # noqa: E402

out = {}
for (H, W, L) in ((8, 8, 4), (16, 16, 4)):
    n = H * W
    E = len(S.grid_edges(H, W)[0])
    out[f"costs_{H}x{W}_L{L}"] = mt19937_u01(12345, n * L + E * L * L)
# reference outputs recorded in SURVEY.md 8(c): LB before, after 1 pass, after 4 passes (anisotropic)
out["lb_8x8_L4_pass0_1_4"] = np.array([18.6574210744, 46.2832640826, 47.6298489968])
# 16x16 grid, LB before and after 1 pass for anisotropic / uniform / damped_uniform
out["lb_16x16_L4_start"] = np.array([79.0639824217])
out["lb_16x16_L4_pass1_aniso_uniform_damped"] = np.array([203.0008353407, 191.2885211469, 185.2649565271])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "survey_grids.npz"), **out)
print("wrote survey_grids.npz")
