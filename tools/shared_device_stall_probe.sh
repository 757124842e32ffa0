#!/bin/bash
# tools/shared_device_stall_probe.sh [attempts] — the experiment behind profiles/r04_8rank_stall_*.txt, on the GPU box:
# 8 rank processes of bench.py on the ONE GPU with the persistent joined-pass launches left ON (bench.py itself switches them off
# when ranks share a device: LPMP_BENCH_KEEP_PERSISTENT=1 overrides that here), wait bound 6 s.  About every third run stalls:
# a chain launch's wait gives up and says which ticket it waited for.  With LPMP_CHAIN_TRACE the dump of the aborted run is kept
# (<path>.aborted) and tools/chain_stall_report.py shows what its open tickets were doing.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
N=${1:-6}
mkdir -p gpurun_out /tmp/ctrace
fails=0
for attempt in $(seq 1 "$N"); do
  rm -f /tmp/ctrace/*
  LPMP_BENCH_KEEP_PERSISTENT=1 LPMP_ROT_EXPLICIT=1 LPMP_CHAIN_TIMEOUT_S=6 LPMP_CHAIN_TRACE=/tmp/ctrace/t_%p.bin timeout 900 \
    python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-compare-schedules > gpurun_out/stall_probe_$attempt.json 2> gpurun_out/stall_probe_$attempt.err
  rc=$?
  echo "attempt $attempt rc=$rc"
  if [ $rc -ne 0 ]; then
    fails=$((fails + 1))
    grep -h "EngineError: " gpurun_out/stall_probe_$attempt.err | sort | uniq -c | head -3
    for f in /tmp/ctrace/*.aborted; do [ -f "$f" ] && python tools/chain_stall_report.py "$f" | head -12; done
  fi
done
echo "stalled: $fails of $N"
