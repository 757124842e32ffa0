#!/usr/bin/env python3
"""C4-style workload probe (BASELINE configs[3] on ONE GPU): G(n, m) random graph, 16 labels, dense tables
generated in HBM; checks LB monotonicity + energy invariance and times passes.
usage: c4_probe.py [n_nodes] [n_edges] [labels] [passes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
m_e = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 5
colour = len(sys.argv) > 5 and sys.argv[5] == "colour"
t0 = time.time()
rng = np.random.Generator(np.random.PCG64(1))
e = rng.integers(0, n, size=(int(m_e * 1.02) + 16, 2))
e = e[e[:, 0] != e[:, 1]]
key = np.minimum(e[:, 0], e[:, 1]) * n + np.maximum(e[:, 0], e[:, 1])
key = np.unique(key)[:m_e]
ei, ej = key // n, key % n
m_e = ei.shape[0]
if colour:   # relabel the variables in a colour-major order (lp_mp_amd/ordering.py): few, wide levels
    from lp_mp_amd import ordering
    tc = time.time()
    rank = ordering.colour_major_order(n, ei, ej)
    ri, rj = rank[ei], rank[ej]
    ei, ej = np.minimum(ri, rj), np.maximum(ri, rj)
    print("colour-major order %.1fs" % (time.time() - tc), flush=True)
model = S.mrf_model(n, L, ei, ej, np.zeros(n * L), device_const=True)
print("model %.1fs  factors %d messages %d" % (time.time() - t0, model.n_factors, model.n_messages), flush=True)
dev = torch.device("cuda:0")
const = torch.empty(m_e * L * L, dtype=torch.float64, device=dev)
dual = torch.zeros(n * L + m_e * 2 * L, dtype=torch.float64, device=dev)
sp = torch.cuda.current_stream().cuda_stream
E.synth_fill(const.data_ptr(), const.numel(), 1, n * L, sp)
E.synth_fill(dual.data_ptr(), n * L, 1, 0, sp)
torch.cuda.synchronize()
t0 = time.time()
eng = E.Engine(0); eng.set_stream(sp)
eng.upload(model, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
eng.set_reparametrization(0)
info = [eng.plan.schedule_info(d, 0) for d in (0, 1)]
pinfo = eng.plan.pass_schedule_info(0)
print("plan+upload %.1fs levels %s pass-levels %d launches %d" % (time.time() - t0, [i["n_levels"] for i in info], pinfo["n_levels"], pinfo["n_launches"]), flush=True)
eit, ejt = torch.from_numpy(ei).to(dev), torch.from_numpy(ej).to(dev)
T = const.view(m_e, L, L)
gen = torch.Generator(device="cpu").manual_seed(0)
x = torch.randint(0, L, (n,), generator=gen).to(dev)
def energy():
    th = dual[: n * L].view(n, L); pw = dual[n * L:].view(m_e, 2 * L)
    ar_n = torch.arange(n, device=dev); ar_e = torch.arange(m_e, device=dev)
    xi, xj = x[eit], x[ejt]
    return (th[ar_n, x].sum() + T[ar_e, xi, xj].sum() + pw[ar_e, xi].sum() + pw[ar_e, L + xj].sum()).item()
e0 = energy(); lb = [eng.lower_bound()]
eng.compute_pass(1); torch.cuda.synchronize(); lb.append(eng.lower_bound())
t0 = time.perf_counter(); eng.compute_pass(passes); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / passes
lb.append(eng.lower_bound()); e1 = energy()
upd = sum(i["n_receives"] + i["n_sends"] for i in info); by = sum(i["algorithmic_bytes"] for i in info)
print("ms/pass %.3f  msg-updates/s %.3e  algorithmic GB/s %.0f" % (dt * 1e3, upd / dt, by / dt / 1e9))
print("LB", lb, "energy drift", abs(e1 - e0) / abs(e0), "weak duality", lb[-1] <= e1)
assert lb[0] <= lb[1] <= lb[2] and abs(e1 - e0) <= 1e-9 * abs(e0) and lb[-1] <= e1
print("ok")
