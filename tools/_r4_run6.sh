#!/bin/bash
# the 8-rank stall: trace the chain runs (explicit ticket lists), report what the open tickets of a stalled run were doing
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out /tmp/ctrace
for attempt in 1 2 3; do
  rm -f /tmp/ctrace/*
  LPMP_ROT_EXPLICIT=1 LPMP_CHAIN_TIMEOUT_S=8 LPMP_CHAIN_TRACE=/tmp/ctrace/t_%p.bin timeout 900 python bench.py --gpus 8 --steps 6 --warmup 3 --no-cpu-baseline --no-compare-schedules > gpurun_out/r4_6_try$attempt.json 2> gpurun_out/r4_6_try$attempt.err
  rc=$?
  echo "attempt $attempt rc=$rc"
  if [ $rc -ne 0 ]; then
    grep -h "EngineError" gpurun_out/r4_6_try$attempt.err | sort | uniq -c | head
    python tools/chain_stall_report.py /tmp/ctrace/*.bin 2>&1 | tee gpurun_out/r4_6_stall_report.txt | head -80
    break
  fi
done
# the same without the persistent launch (one launch per step): must pass
LPMP_NO_BLOCKED_PASSES=1 timeout 900 python bench.py --gpus 8 --steps 6 --warmup 3 --no-cpu-baseline --no-compare-schedules > gpurun_out/r4_6_noblocked.json 2> gpurun_out/r4_6_noblocked.err
echo "no blocked passes rc=$?"
