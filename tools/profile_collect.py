#!/usr/bin/env python3
"""tools/profile_collect.py TAG [kernel substring] [last_n skip_tail] [workload] (last_n = 0: the largest launch of the kernel) — summaries of tools/profile_round.sh's runs
(gpurun_out/prof_TAG/) written into profiles/:
  TAG_bench_c3_default.json, TAG_bench_c3_under_rocprof.json   bench lines (plain / under the kernel trace)
  TAG_bench_c3_kernel_stats.csv                                per-kernel calls / total / average of the kernel trace
  TAG_pmc_{fetch,write}_counter_collection.csv                 the sweep kernels' PMC rows
  TAG_pmc_c3_dense32.json                                      HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes
  + a cross-check printed: in-bench HIP-event average against the profiler's average of the same launches"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "sweep_dense_pk_kernel<32"
last_n, skip_tail = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (41, 8)
# workload name inside the file names: c3 (default) / c4 ...; the PMC summary is TAG_pmc_<WL>_<class>.json
wl = sys.argv[5] if len(sys.argv) > 5 else "c3"
wl_class = {"c3": "c3_dense32", "c4": "c4_dense16"}.get(wl, wl)
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def bench_line(log, out):
    for line in open(os.path.join(src, log)):
        if line.startswith('{"metric"'):
            open(os.path.join(dst, out), "w").write(line)
            return json.loads(line)
    raise SystemExit(f"no bench line in {log}")


plain = bench_line("bench_plain.log", f"{tag}_bench_{wl}_default.json")
prof = bench_line("bench_stats.log", f"{tag}_bench_{wl}_under_rocprof.json")
def db(d):
    """the database of that run — the NEWEST one: gpurun merges a call's output into gpurun_out/, so an earlier run of the same tag
    leaves its database beside the new one"""
    return max(glob.glob(os.path.join(src, d, "*", "*.db")), key=os.path.getmtime)
tool = os.path.join(ROOT, "tools", "rocpd_summary.py")
subprocess.check_call([sys.executable, tool, "stats", db("stats"), os.path.join(dst, f"{tag}_bench_{wl}_kernel_stats.csv")])
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    subprocess.check_call([sys.executable, tool, "pmc", db("pmc_" + name), counter, os.path.join(dst, f"{tag}_pmc_{name}_counter_collection.csv")])
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(dst, f"{tag}_pmc_fetch_counter_collection.csv"),
                       os.path.join(dst, f"{tag}_pmc_write_counter_collection.csv"), kernel, os.path.join(dst, f"{tag}_pmc_{wl_class}.json"),
                       str(last_n), str(skip_tail)])
# a chain launch covers as many passes as the call had: keep that with the byte count (bench.py scales by it)
pj = os.path.join(dst, f"{tag}_pmc_{wl_class}.json")
d = json.load(open(pj))
d["passes_per_launch"] = plain["roofline"].get("passes_per_launch")
# provenance: the library that ran under the profiler (bench.py compares it with the running library's stamp and reports
# `traffic: null` on a mismatch) and the kernel's full name as the profiler printed it
def _hash_of(log):
    try:
        for line in open(os.path.join(src, log)):
            if line.startswith('{"metric"'):
                return json.loads(line).get("library", {}).get("source_hash")
    except OSError:
        pass
    return None
hashes = {h for h in (_hash_of("bench_fetch.log"), _hash_of("bench_write.log"), prof.get("library", {}).get("source_hash"), plain.get("library", {}).get("source_hash")) if h}
if len(hashes) != 1:
    raise SystemExit(f"the runs of this profile were not made with ONE library: {hashes}")
d["library_source_hash"] = hashes.pop()
import csv
names = sorted({r["Kernel_Name"] for r in csv.DictReader(open(os.path.join(dst, f"{tag}_pmc_fetch_counter_collection.csv"))) if kernel in r["Kernel_Name"]})
d["kernel_full_names"] = names
d["variable_order"] = plain.get("config", {}).get("variable_order")
d["note"] = ("FETCH_SIZE / WRITE_SIZE are the L2's fabric-side request counters: reads served by the 256 MiB Infinity Cache are "
             "counted like reads served by HBM (MI355X_MICROARCH.md, HBM / rocprofv3 section)")
json.dump(d, open(pj, "w"))
c = sqlite3.connect(db("stats"))
v = [r[0] for r in c.execute("select duration from kernels where name like ? order by start", ("%" + kernel + "%",))]
leg = v[:len(v) - skip_tail][-last_n:] if last_n > 0 else [max(v)]          # last_n = 0: the longest launch (the timed multi-pass chain launch)
print(json.dumps({"plain_ms_per_step": plain["ms_per_step"], "plain_frac": plain["roofline"]["frac"], "oracle_check": plain.get("oracle_check"),
                  "under_rocprof_in_bench_avg_launch_ms": prof["roofline"]["avg_launch_ms"],
                  "rocprof_kernel_trace_avg_ms_same_launches": sum(leg) / len(leg) / 1e6, "launches": len(leg)}))
