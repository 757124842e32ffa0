"""Projected scaling table from ONE GPU: what a pass costs per part when the model runs in N parts (all parts of one process on the
one device: tools/overlap_probe.py, lockstep_graph_probe.py, lockstep_c5_probe.py) + a stated model of the exchange,
    t_N = t_run(N) + n_exchanges(N) * latency + bytes(N) / bandwidth,
against the single-GPU time.  A PROJECTION, not a measurement: it is what the first run on an 8-GPU node confirms or refutes
(bench.py prints the same model beside the measured time in every N-rank line).

    python tools/scaling_projection.py collect      # on the GPU box: runs the probes for 1 / 2 / 4 / 8 parts -> gpurun_out/r05_scaling_probe_*.txt
    python tools/scaling_projection.py table [latency_us] [GBps]   # anywhere: reads those files -> JSON table on stdout"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def collect():
    os.makedirs(OUT, exist_ok=True)
    run = lambda args, name: open(os.path.join(OUT, name), "a").write(subprocess.run([sys.executable] + args, capture_output=True, text=True, cwd=ROOT, timeout=3000).stdout)
    for f in ("r05_scaling_probe_c3.txt", "r05_scaling_probe_c4.txt", "r05_scaling_probe_c5.txt"):
        open(os.path.join(OUT, f), "w").close()
    for parts in (1, 2, 3):                                 # (3 parts: the middle one has two neighbours, like every inner rank of 8)
        run(["tools/overlap_probe.py", "1024", "32", str(parts), "10", "12"], "r05_scaling_probe_c3.txt")
    for parts in (2, 4, 8):
        run(["tools/lockstep_graph_probe.py", "2000000", "10000000", "16", str(parts), "6"], "r05_scaling_probe_c4.txt")
        run(["tools/lockstep_c5_probe.py", str(parts), "6"], "r05_scaling_probe_c5.txt")


def lines(name):
    p = os.path.join(OUT, name)
    if not os.path.exists(p):
        p = os.path.join(ROOT, "profiles", name)
    return [json.loads(l) for l in open(p) if l.startswith("{")]


def table(lat_us=30.0, gbps=400.0):
    lam, bw = lat_us * 1e-3, gbps * 1e9
    out = {"assumed_latency_us_per_exchange": lat_us, "assumed_GBps_per_rank": gbps, "formula": "t_run + n_exchanges * latency + bytes / bandwidth",
           "note": "projection from per-part times on ONE GPU; no multi-GPU hardware was available"}
    c3 = {d["parts"]: d for d in lines("r05_scaling_probe_c3.txt")}
    t1 = c3[1]["ms_per_pass_and_part"]
    rows = {"1": {"ms_per_pass": t1, "efficiency": 1.0}}
    inner = c3[3]                                           # per part with 3 windows: 2 edge windows + 1 inner one
    for n in (2, 4, 8):
        d = c3[2] if n == 2 else inner
        ex_per_pass = d["exchanges_per_pass"]
        byts = max(d["doubles_sent_per_exchange_by_part"]) * 8             # (3 windows: the inner one ships to both neighbours, like every inner rank of 8)
        t = d["ms_per_pass_and_part"] + ex_per_pass * lam + ex_per_pass * byts / bw * 1e3
        rows[str(n)] = {"ms_per_pass": round(t, 4), "efficiency": round(t1 / t, 4), "aggregate_speedup_weak": round(n * t1 / t, 3)}
    out["c3 (weak scaling: one 1024 x 1024 grid per GPU, overlap schedule)"] = rows
    c4 = {d["parts"]: d for d in lines("r05_scaling_probe_c4.txt")}
    t1 = min(d["unpartitioned_colour_major"]["ms_per_pass"] for d in c4.values())
    rows = {"1": {"ms_per_pass": t1, "speedup": 1.0}}
    for n in sorted(c4):
        ls = c4[n]["lockstep"]
        t = ls["ms_runs_only_per_pass_and_part"] + ls["exchanges_per_pass"] * lam + ls["halo_MB_per_pass_and_part"] * 1e6 / bw * 1e3
        # t_run: the runs of a part while the other parts' runs interleave with it on the one GPU (1.80 ms at 8 parts) — a part running
        # ALONE takes 1.44, but then the pack / unpack copies of its 16 exchanges (0.2 - 0.46 ms, inside `ms_exchanges_only` here) come
        # on top: the interleaved figure stands for both
        rows[str(n)] = {"ms_per_pass": round(t, 4), "speedup": round(t1 / t, 3), "t_run_ms": ls["ms_runs_only_per_pass_and_part"],
                        "t_run_one_part_alone_ms": ls["ms_runs_only_one_part_alone"], "in_process_exchange_copies_ms": ls["ms_exchanges_only_per_pass_and_part"],
                        "exchanges_per_pass": round(ls["exchanges_per_pass"], 2), "MB_per_pass_and_rank": ls["halo_MB_per_pass_and_part"], "cut_fraction": c4[n]["cut_fraction"]}
    out["c4 (strong scaling: G(2 M, 10 M), 16 labels, lock step)"] = rows
    for name in ("local triples", "local triples, colour-major edge variables"):
        c5 = {d["parts"]: d for d in lines("r05_scaling_probe_c5.txt") if d["c5"] == name}
        if not c5:
            continue
        t1 = min(d["unpartitioned_ms_per_pass"] for d in c5.values())
        rows = {"1": {"ms_per_pass": t1, "speedup": 1.0}}
        for n in sorted(c5):
            d = c5[n]
            rows[str(n)] = {"ms_per_pass": d["projected_ms_per_pass_plain"] if (lat_us, gbps) == (d["assumed_latency_us"], d["assumed_GBps"]) else
                            round(d["runs_only_ms_per_pass_and_part"] + d["exchanges_per_pass"] * lam + d["exchange_bytes_per_pass_and_part_max"] / bw * 1e3, 4),
                            "t_run_ms": d["runs_only_ms_per_pass_and_part"], "exchanges_per_pass": round(d["exchanges_per_pass"], 2)}
            rows[str(n)]["speedup"] = round(t1 / rows[str(n)]["ms_per_pass"], 3)
        out[f"c5, {name} (strong scaling, lock step)"] = rows
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "collect":
        collect()
    else:
        print(json.dumps(table(*(float(x) for x in sys.argv[2:4])), indent=1))
