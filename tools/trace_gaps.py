#!/usr/bin/env python3
"""kernel durations and inter-kernel gaps from a rocprofv3 kernel trace CSV (one stream)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gap = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
gap = [g for g in gap if g < 100]
import statistics as st
print("n", len(dur), "dur us: mean %.2f median %.2f min %.2f max %.2f" % (st.mean(dur), st.median(dur), min(dur), max(dur)))
print("gap us: mean %.2f median %.2f min %.2f max %.2f" % (st.mean(gap), st.median(gap), min(gap), max(gap)))
