"""Generates tests/golden/c3_full_lb.npz: the CPU oracle run ONCE on the exact C3 workload of BASELINE.json
configs[2] (1024 x 1024 grid, 32 labels, dense tables, colour-major order, anisotropic weights) — the inputs
bench.py (seed 1) and tests/test_engine_gpu.py::test_full_size_properties (seed 3) generate in HBM.

Needs ~40 GB of host memory and 7-14 s per pass on one core; run in the build container, not on the GPU box:

    python tests/golden/make_c3_full.py

Stored per seed: the lower bound after the listed pass counts and two exact checksums of the packed duals
(wrapping uint64 sums over the IEEE bit patterns, plain and position-weighted), so the device result can be
compared bit for bit without shipping 1.3 GB of duals.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lp_mp_amd import model as M, synthetic as S   # noqa: E402
from oracle.binding import Oracle                   # noqa: E402

H = W = 1024
L = 32
# seed -> pass counts at which the state is recorded.  Seed 1 is bench.py's model: EVERY pass count up to 48 is kept,
# so that whatever --warmup / --steps the caller of bench.py chooses (the round driver runs 5 + 20), the state the
# timed call leaves in HBM has an oracle value to be compared with.
RUNS = {1: list(range(0, 49)), 3: [0, 1, 2, 3]}


def dual_checksums(d: np.ndarray):
    """(sum of bit patterns, sum of bit pattern * (2 i + 1)) mod 2^64 — what tests compute on the device with
    wrapping int64 arithmetic."""
    b = np.ascontiguousarray(d).view(np.uint64)
    with np.errstate(over="ignore"):
        s0 = np.add.reduce(b, dtype=np.uint64)
        w = np.arange(b.shape[0], dtype=np.uint64) * np.uint64(2) + np.uint64(1)
        s1 = np.add.reduce(b * w, dtype=np.uint64)
    return np.uint64(s0), np.uint64(s1)


def main():
    out = {"H": H, "W": W, "L": L}
    path = os.path.join(ROOT, "tests", "golden", "c3_full_lb.npz")
    only = [int(a) for a in sys.argv[1:]]            # e.g. `make_c3_full.py 1`: redo seed 1, keep the other seeds' entries
    if only and os.path.exists(path):
        old = np.load(path)
        out.update({k: old[k] for k in old.files if k not in ("H", "W", "L")})
    for seed, marks in RUNS.items():
        if only and seed not in only:
            continue
        t0 = time.time()
        m = S.grid_model(H, W, L, order="colour_major", seed=seed)
        o = Oracle(m)
        o.model = None
        del m                                        # the oracle holds its own copy of the tables
        o.set_reparametrization(M.REPAM_ANISOTROPIC)
        print(f"seed {seed}: model + oracle ready after {time.time() - t0:.0f} s", flush=True)
        lbs, c0, c1 = [], [], []
        done = 0
        for k in marks:
            if k > done:
                o.ComputePass(k - done)
                done = k
            lbs.append(o.LowerBound())
            a, b = dual_checksums(o.duals())
            c0.append(a); c1.append(b)
            print(f"seed {seed}: {k} passes, LB {lbs[-1]!r}, {time.time() - t0:.0f} s", flush=True)
        out[f"passes_seed{seed}"] = np.array(marks, np.int64)
        out[f"lb_seed{seed}"] = np.array(lbs, np.float64)
        out[f"dual_sum_seed{seed}"] = np.array(c0, np.uint64)
        out[f"dual_wsum_seed{seed}"] = np.array(c1, np.uint64)
        del o
    np.savez(path, **out)
    print("written")


if __name__ == "__main__":
    main()
