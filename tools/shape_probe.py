"""Message-passing rate on model SHAPES other than the benchmark's: what a reference user may bring (round 6: looking for cliffs).
  python tools/shape_probe.py potts2d G L | grid3d N L | grid8 G L | strip H W L      [passes]
Variables are numbered colour by colour where the shape is 2-colourable (3-D grids: parity of x + y + z); the 8-connected grid is
inserted row by row and run in the order the ENGINE suggests (lpmp_plan_suggest_order: 4 colours).  Costs are generated in HBM."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lp_mp_amd import engine as E, model as M, synthetic as S

shape = sys.argv[1]
a = [int(x) for x in sys.argv[2:]]
potts = False
if shape == "potts2d":
    G, L = a[:2]; passes = a[2] if len(a) > 2 else 20
    ei, ej = S.grid_edges(G, G); rank = S.grid_variable_order(G, G, "colour_major").reshape(-1); n = G * G; potts = True
elif shape == "strip":
    H, W, L = a[:3]; passes = a[3] if len(a) > 3 else 10
    ei, ej = S.grid_edges(H, W); rank = S.grid_variable_order(H, W, "colour_major").reshape(-1); n = H * W
elif shape == "grid3d":
    N, L = a[:2]; passes = a[2] if len(a) > 2 else 10
    idx = np.arange(N ** 3, dtype=np.int64).reshape(N, N, N)
    ei = np.concatenate([idx[:-1].ravel(), idx[:, :-1].ravel(), idx[:, :, :-1].ravel()])
    ej = np.concatenate([idx[1:].ravel(), idx[:, 1:].ravel(), idx[:, :, 1:].ravel()])
    z, y, x = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
    black = ((x + y + z) % 2 == 0).ravel(); n = N ** 3
    rank = np.empty(n, np.int64); rank[black] = np.arange(int(black.sum())); rank[~black] = int(black.sum()) + np.arange(n - int(black.sum()))
elif shape == "grid8":
    G, L = a[:2]; passes = a[2] if len(a) > 2 else 10
    idx = np.arange(G * G, dtype=np.int64).reshape(G, G); n = G * G
    ei = np.concatenate([idx[:, :-1].ravel(), idx[:-1].ravel(), idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel()])
    ej = np.concatenate([idx[:, 1:].ravel(), idx[1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()])
    rank = np.arange(n, dtype=np.int64)
else:
    raise SystemExit(__doc__)
va, vb = rank[ei], rank[ej]
i, j = np.minimum(va, vb), np.maximum(va, vb)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sp = torch.cuda.current_stream().cuda_stream
n_e = i.shape[0]
t0 = time.perf_counter()
if potts:
    m = S.mrf_model(n, L, i, j, S.u01(n * L, 1, 0), potts=S.u01(n_e, 1, n * L))
    const = dual = None
else:
    m = S.mrf_model(n, L, i, j, None, device_const=True, device_dual=True)
    const = torch.empty(n_e * L * L, dtype=torch.float64, device=dev)
    dual = torch.zeros(n * L + n_e * 2 * L, dtype=torch.float64, device=dev)
    E.synth_fill(const.data_ptr(), const.numel(), 1, n * L, sp)
    E.synth_fill(dual.data_ptr(), n * L, 1, 0, sp)
    torch.cuda.synchronize()
suggested = None
if shape == "grid8":
    r, k = E.Plan(m).suggest_order(0)
    m = m.with_factor_order(r); suggested = k
e = E.Engine(0); e.set_stream(sp)
if potts:
    e.upload(m)
else:
    e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
e.set_reparametrization(M.REPAM_ANISOTROPIC)
setup = time.perf_counter() - t0
info = [e.plan.schedule_info(d, M.REPAM_ANISOTROPIC) for d in (0, 1)]
lb0 = e.lower_bound()
e.prepare_passes(3); e.prepare_passes(passes)
e.compute_pass(3); torch.cuda.synchronize()
t0 = time.perf_counter(); e.compute_pass(passes); e.synchronize(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
e.enable_kernel_timing(True); e.compute_pass(passes); torch.cuda.synchronize(); kt = e.kernel_timing(); e.enable_kernel_timing(False)
by = sum(x["algorithmic_bytes"] for x in info)
print(json.dumps({"shape": shape, "args": a, "variables": n, "edges": int(n_e), "labels": L, "levels": [x["n_levels"] for x in info], "suggested_colours": suggested,
                  "ms_per_pass": dt / passes * 1e3, "algorithmic_GBps": by * passes / dt / 1e9, "frac_of_8TBps": by * passes / dt / 8e12,
                  "msg_updates_per_s": sum(x["n_receives"] + x["n_sends"] for x in info) * passes / dt, "kernels": {k: v["kernel"] for k, v in kt.items()},
                  "pass_rotates": bool(e.plan.pass_rotates(M.REPAM_ANISOTROPIC)), "setup_s": setup, "lb": [lb0, e.lower_bound()],
                  "peak_GB": torch.cuda.max_memory_allocated() / 1e9}))
