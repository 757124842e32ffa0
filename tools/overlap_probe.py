"""Cost of the overlapped-strips schedule (lp_mp_amd/overlap.py) on ONE GPU: `parts` windows of the (parts * grid) x grid
headline grid as separate engines on the one device, run one after the other per chunk, exchanges as in-process copies — ms per
pass and part (= what one rank of a real run spends per pass, without RCCL latency), exchanges per pass, and the bound.
    python tools/overlap_probe.py [grid] [labels] [parts] [passes] [ghost_rows] [chunk]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lp_mp_amd import engine as E, model as M, multi_gpu as MG, overlap as OV

g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ghost = int(sys.argv[5]) if len(sys.argv) > 5 else 12
chunk = int(sys.argv[6]) if len(sys.argv) > 6 else None
mode = M.REPAM_ANISOTROPIC
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
sweeps, keep = [], []
for r in range(parts):
    p = OV.strip_window_part(g, g, L, "dense", r, parts, ghost, 1)
    m = p.model
    const = torch.empty(max(int(m.const_sizes().sum()), 2), dtype=torch.float64, device=dev)
    dual = torch.zeros(int(m.dual_sizes().sum()), dtype=torch.float64, device=dev)
    MG.fill_device_costs(torch, E, p, const, dual, stream)
    e = E.Engine(0); e.set_stream(stream)
    e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
    e.set_reparametrization(mode)
    sweeps.append(OV.OverlapSweep(torch, p, e, dual, chunk)); keep.append((const, dual))
setup = time.perf_counter() - t0
t0 = time.perf_counter()
for s in sweeps:
    for k in sorted(set(s.chunks(passes))):
        s.engine.prepare_passes(k)
prep = time.perf_counter() - t0
OV.run_overlapped(sweeps, passes); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    OV.run_overlapped(sweeps, passes); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / passes * 1e3)
# the exchange alone
t0 = time.perf_counter()
for _ in range(10):
    if parts > 1:
        packed = [s.pack() for s in sweeps]
        offs = [np.concatenate([[0], np.cumsum(s.send_counts)]) for s in sweeps]
        for dst, s in enumerate(sweeps):
            s.unpack(torch.cat([packed[src][offs[src][dst]: offs[src][dst + 1]] for src in s.peers]))
torch.cuda.synchronize()
ex_ms = (time.perf_counter() - t0) / 10 * 1e3
lb = sum(s.local_lower_bound() for s in sweeps)
out = {"grid_per_part": g, "labels": L, "parts": parts, "passes_per_call": passes, "ghost_rows": ghost, "passes_between_exchanges": sweeps[0].chunk,
       "setup_s": round(setup, 2), "prepare_passes_s": round(prep, 2),
       "ms_per_pass_all_parts": [round(t, 3) for t in ts], "ms_per_pass_and_part": round(min(ts) / parts, 3),
       "exchanges_per_pass": len(sweeps[0].chunks(passes)) / passes if parts > 1 else 0.0, "exchange_ms_all_parts": round(ex_ms, 3),
       "doubles_sent_per_exchange_by_part": [int(s.send_counts.sum()) for s in sweeps],
       "window_rows": [(s.part.r0, s.part.r1) for s in sweeps], "lower_bound": lb, "total_passes": 4 * passes}
print(json.dumps(out))
