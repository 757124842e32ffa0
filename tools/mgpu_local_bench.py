#!/usr/bin/env python3
"""Times the partitioned (multi-GPU) pass on ONE GPU: `world` row strips of H x W lock-stepped in process
(in-process row shuffles stand in for the all-to-all).  Reported per-strip time = what one rank of a real run
spends per pass, excluding RCCL latency.   usage: mgpu_local_bench.py [world] [H] [passes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from lp_mp_amd import engine as E, model as M, multi_gpu as MG
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 10
W, L = H, 32
dev = torch.device("cuda:0"); sp = torch.cuda.current_stream().cuda_stream
sweeps, keep = [], []
for r in range(world):
    part = MG.strip_local_part(H, W, L, "dense", "colour_major", r, world, 1, device_const=True)
    m = part.model
    const = torch.empty(int(m.const_sizes().sum()), dtype=torch.float64, device=dev)
    dual = torch.zeros(int(m.dual_sizes().sum()), dtype=torch.float64, device=dev)
    for (off, cnt, sd, first) in part.const_fill: E.synth_fill(const.data_ptr() + 8 * off, cnt, sd, first, sp)
    for (off, cnt, sd, first) in part.dual_fill: E.synth_fill(dual.data_ptr() + 8 * off, cnt, sd, first, sp)
    torch.cuda.synchronize()
    e = E.Engine(0); e.set_stream(sp)
    e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
    sweeps.append(MG.PartitionedSweep(torch, part, e, dual, M.REPAM_ANISOTROPIC, None, "pass")); keep.append((const, dual, e))
print("program", [s[1] for s in sweeps[0].program(4) if s[0] == "run"])
MG.run_lockstep(sweeps, 2); torch.cuda.synchronize()
t0 = time.perf_counter(); MG.run_lockstep(sweeps, passes); torch.cuda.synchronize(); dt = time.perf_counter() - t0
upd = sum(s.updates_per_pass() for s in sweeps)
print("ms per pass per strip %.3f   aggregate msg-updates/s if the strips ran on %d GPUs: %.3e" % (dt / passes / world * 1e3, world, upd / (dt / passes / world)))
print("LB", sum(s.local_lower_bound() for s in sweeps))
# breakdown on strip 0: main schedules alone vs boundary pieces
s0 = sweeps[0]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for k in ("FB", "first", "mid", "last"):
    if k in s0.sched: print("schedule %-5s %.3f ms  %s" % (k, timeit(lambda: s0.run(k)), s0.engine.schedule_info(s0.sched[k])))
print("ghost recv+send %.3f ms" % timeit(lambda: (s0.engine.schedule_run(s0.ghost_recv), s0.engine.schedule_run(s0.ghost_send))))
print("boundary_pack %.3f ms  reply %.3f ms" % (timeit(lambda: s0.boundary_pack()), timeit(lambda: sweeps[1].boundary_reply(torch.zeros((sweeps[1].n_in_elems,), dtype=torch.float64, device=dev)))))
