#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out /tmp/ctrace
for attempt in 1 2 3; do
  LPMP_CHAIN_TIMEOUT_S=8 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-compare-schedules > gpurun_out/r4_7_a$attempt.json 2> gpurun_out/r4_7_a$attempt.err
  echo "periodic attempt $attempt rc=$?"; grep -h "EngineError" gpurun_out/r4_7_a$attempt.err | sort | uniq -c | head -3
done
for attempt in 1 2 3; do
  rm -f /tmp/ctrace/*
  LPMP_ROT_EXPLICIT=1 LPMP_CHAIN_TIMEOUT_S=8 LPMP_CHAIN_TRACE=/tmp/ctrace/t_%p.bin timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-compare-schedules > gpurun_out/r4_7_b$attempt.json 2> gpurun_out/r4_7_b$attempt.err
  rc=$?
  echo "explicit+trace attempt $attempt rc=$rc"
  if [ $rc -ne 0 ]; then
    grep -h "EngineError" gpurun_out/r4_7_b$attempt.err | sort | uniq -c | head
    python tools/chain_stall_report.py /tmp/ctrace/*.bin 2>&1 | tee gpurun_out/r4_7_stall_report.txt | head -80
    break
  fi
done
