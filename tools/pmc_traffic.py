#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs with
--kernel-trace only) into per-launch traffic of one kernel as the L2 requests it from the fabric (Infinity Cache + HBM).

Units and gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section) and
cdna_hip_programming.md section 7: counters are in KiB; FETCH_SIZE reports exactly half the bytes of a
wide coalesced streaming read on gfx950, WRITE_SIZE is exact:  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <out.json> [last_n [skip_tail]]
last_n: only the last n launches of the kernel (e.g. the event-timed leg of bench.py), after dropping the final
skip_tail launches (bench.py ends with two rounding passes = 8 launches of the directional sweeps)
"""
import csv
import json
import sys


def per_launch(path, counter, kernel):
    vals = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
    return vals


def main():
    fetch_csv, write_csv, kernel, out = sys.argv[1:5]
    f = per_launch(fetch_csv, "FETCH_SIZE", kernel)
    w = per_launch(write_csv, "WRITE_SIZE", kernel)
    if len(sys.argv) > 6 and int(sys.argv[6]) > 0:
        f, w = f[:-int(sys.argv[6])], w[:-int(sys.argv[6])]
    if len(sys.argv) > 5 and int(sys.argv[5]) == 0:
        # last_n = 0: the LARGEST launch of the kernel (the timed multi-pass chain launch: other legs of bench.py launch the
        # same kernel for single sweeps)
        k = max(range(len(f)), key=lambda i: f[i])
        f, w = [f[k]], [w[k]]
    elif len(sys.argv) > 5:
        f, w = f[-int(sys.argv[5]):], w[-int(sys.argv[5]):]
    assert f and w and len(f) == len(w), (len(f), len(w))
    by = [(2.0 * a + b) * 1024.0 for a, b in zip(f, w)]
    res = {"kernel": kernel, "launches": len(by), "fetch_size_kib_avg": sum(f) / len(f), "write_size_kib_avg": sum(w) / len(w),
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
           "counts": "L2 fabric-side requests: reads served by the Infinity Cache are counted like reads served by HBM",
           "fabric_bytes_per_launch_avg": sum(by) / len(by), "fabric_bytes_per_launch_min": min(by), "fabric_bytes_per_launch_max": max(by)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
