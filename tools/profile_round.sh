#!/bin/bash
# tools/profile_round.sh TAG — the evidence bench.py's roofline object cites, produced on the GPU box:
#   profiles/TAG_bench_c3_default.json        the plain bench line
#   profiles/TAG_bench_c3_under_rocprof.json  the bench line of the kernel-trace run (its in-bench HIP-event average must agree
#                                             with the profiler's)
#   profiles/TAG_bench_c3_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of that same command
#   profiles/TAG_pmc_{fetch,write}_counter_collection.csv + TAG_pmc_c3_dense32.json   HBM bytes per launch from the PMC
#       counters, FETCH_SIZE and WRITE_SIZE in SEPARATE passes with --kernel-trace only, corrected as
#       /opt/skills/guides/MI355X_MICROARCH.md prescribes (tools/pmc_traffic.py)
# The program after `--` is python3 itself (no env / shell wrapper: the profiler initialises the GPU before the program starts).
set -u
TAG=${1:-r02}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT" profiles
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench_plain.log" 2>&1 && tail -n 1 "$OUT/bench_plain.log" > profiles/${TAG}_bench_c3_default.json
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_stats.log" 2>&1
tail -n 1 "$OUT/bench_stats.log" > profiles/${TAG}_bench_c3_under_rocprof.json
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp "$f" profiles/${TAG}_bench_c3_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_write.log" 2>&1
ff=$(find "$OUT/pmc_fetch" -name "*counter_collection.csv" | head -n 1); fw=$(find "$OUT/pmc_write" -name "*counter_collection.csv" | head -n 1)
if [ -n "$ff" ] && [ -n "$fw" ]; then
  python3 - "$ff" "$fw" "$TAG" <<'PY'
import csv, sys
# keep only the sweep kernel's rows (the full collections are tens of MB)
for src, name in ((sys.argv[1], "fetch"), (sys.argv[2], "write")):
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if "sweep_" in r["Kernel_Name"] or "chain_" in r["Kernel_Name"]]
    with open(f"profiles/{sys.argv[3]}_pmc_{name}_counter_collection.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
PY
  python3 tools/pmc_traffic.py profiles/${TAG}_pmc_fetch_counter_collection.csv profiles/${TAG}_pmc_write_counter_collection.csv "sweep_dense_pk_kernel<32" profiles/${TAG}_pmc_c3_dense32.json 41 8
fi
ls -la profiles | grep "$TAG"
