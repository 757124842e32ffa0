#!/usr/bin/env python3
"""Summaries of rocprofv3's SQLite output (ROCm 7.2 writes <pid>_results.db by default) in the CSV shapes the round-1
profiles used:
  rocpd_summary.py stats  <results.db> <out_kernel_stats.csv>                  (= --kernel-trace --stats: per-kernel calls / total / average / min / max)
  rocpd_summary.py pmc    <results.db> <counter> <out_counter_collection.csv>  (per dispatch of the sweep / chain kernels: counter value)"""
import csv
import sqlite3
import statistics
import sys


def main():
    what, db = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    if what == "stats":
        rows = c.execute("select name, duration from kernels").fetchall()
        by = {}
        for n, d in rows:
            by.setdefault(n, []).append(d)
        total = sum(sum(v) for v in by.values())
        with open(sys.argv[3], "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v), statistics.pstdev(v) if len(v) > 1 else 0.0])
    else:
        counter, out = sys.argv[3], sys.argv[4]
        cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
        rows = c.execute("select * from counters_collection").fetchall()
        idx = {k: i for i, k in enumerate(cols)}
        name_col = "kernel_name" if "kernel_name" in idx else ("name" if "name" in idx else None)
        with open(out, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            for r in rows:
                kn = r[idx[name_col]]
                if not ("sweep_" in kn or "chain_" in kn):
                    continue
                if r[idx["counter_name"]] != counter:
                    continue
                w.writerow([r[idx["dispatch_id"]], kn, r[idx["counter_name"]], r[idx["value"]], r[idx.get("start", 0)], r[idx.get("end", 0)]])


if __name__ == "__main__":
    main()
