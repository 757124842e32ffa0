#!/bin/bash
# tools/profile_round.sh TAG [bench args...] — on the GPU box: the runs behind bench.py's roofline object.
#   gpurun_out/prof_TAG/bench_plain.log                      the plain bench line
#   gpurun_out/prof_TAG/stats/   + bench_stats.log           rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/prof_TAG/pmc_fetch/, pmc_write/               PMC passes, FETCH_SIZE and WRITE_SIZE in SEPARATE runs with --kernel-trace only
# (gpurun_out/ is what comes back from the box; tools/profile_collect.py TAG then writes the summaries into profiles/).
# The program after `--` is python3 itself: the profiler initialises the GPU before the program starts, so no env / shell hop.
set -u
TAG=${1:-r02}; shift || true
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py "$@" > "$OUT/bench_plain.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_write.log" 2>&1
grep -h '^{"metric"' "$OUT/bench_plain.log" | cut -c1-400
