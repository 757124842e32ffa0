#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
dmesg 2>/dev/null | tail -5
fails=0
for attempt in 1 2 3 4 5 6 7 8; do
  LPMP_CHAIN_TIMEOUT_S=6 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-compare-schedules > gpurun_out/r4_10_a$attempt.json 2> gpurun_out/r4_10_a$attempt.err
  rc=$?
  echo "attempt $attempt rc=$rc $(python -c "import json;d=json.loads(open('gpurun_out/r4_10_a$attempt.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['dual_bound_gap'])" 2>/dev/null)"
  if [ $rc -ne 0 ]; then fails=$((fails+1)); grep -h "EngineError" gpurun_out/r4_10_a$attempt.err | sort | uniq -c | head -3; fi
done
echo "failures: $fails of 8"
dmesg 2>/dev/null | tail -5
