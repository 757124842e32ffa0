#!/usr/bin/env python3
"""Interleaved A/B timing of engine variants in ONE process (cdna_hip_programming.md 5.4 rule 24).
Each variant = a dict of environment variables read at lpmp_create (LPMP_NO_FUSE, LPMP_NO_PACKED,
LPMP_NT, ...).  Variants share the read-only tables and have their own duals.

usage: ab_bench.py [--grid 1024] [--labels 32] [--order colour_major] [--rounds 5] [--steps 10] VAR=VAL[,VAR=VAL] ...
"""
import argparse, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--labels", type=int, default=32)
ap.add_argument("--order", default="colour_major")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
import torch
from lp_mp_amd import engine as E, model as M, synthetic as S
H = W = a.grid; L = a.labels
dev = torch.device("cuda:0")
m = S.grid_model(H, W, L, order=a.order, seed=1, device_const=True)
n = H * W; n_e = len(S.grid_edges(H, W)[0])
const = torch.empty(n_e * L * L, dtype=torch.float64, device=dev)
sp = torch.cuda.current_stream().cuda_stream
E.synth_fill(const.data_ptr(), const.numel(), 1, n * L, sp)
engines = []
for v in a.variants:
    env = dict(kv.split("=") for kv in v.split(",") if "=" in kv)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    dual = torch.zeros(n * L + n_e * 2 * L, dtype=torch.float64, device=dev)
    E.synth_fill(dual.data_ptr(), n * L, 1, 0, sp)
    torch.cuda.synchronize()
    e = E.Engine(0); e.set_stream(sp)
    e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
    e.set_reparametrization(0); e.compute_pass(2); torch.cuda.synchronize()
    for k, o in old.items():
        if o is None: os.environ.pop(k, None)
        else: os.environ[k] = o
    engines.append((v, e, []))
for r in range(a.rounds):
    for v, e, ts in engines:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.compute_pass(a.steps); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / a.steps * 1e3)
lbs = [e.lower_bound() for _, e, _ in engines]
for (v, e, ts), lb in zip(engines, lbs):
    print(f"{v:40s} ms/pass median {statistics.median(ts):.3f} min {min(ts):.3f} max {max(ts):.3f}  LB {lb:.6f}")
