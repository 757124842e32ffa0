#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
fails=0
for attempt in 1 2 3 4 5 6 7 8; do
  LPMP_CHAIN_TIMEOUT_S=6 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-compare-schedules > gpurun_out/r4_9_a$attempt.json 2> gpurun_out/r4_9_a$attempt.err
  rc=$?
  echo "attempt $attempt rc=$rc $(python -c "import json;d=json.loads(open('gpurun_out/r4_9_a$attempt.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['dual_bound_gap'])" 2>/dev/null)"
  if [ $rc -ne 0 ]; then fails=$((fails+1)); grep -h "EngineError" gpurun_out/r4_9_a$attempt.err | sort | uniq -c | head -3; fi
done
echo "failures: $fails of 8"
# the headline must not have moved (same box A/B would need the old binary; here: the absolute number)
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_9_bench_c3.json 2>/dev/null
python -c "import json;d=json.loads(open('gpurun_out/r4_9_bench_c3.json').read().strip().splitlines()[-1]);print('c3', d['ms_per_step'], d['setup_s'], d['oracle_check']['duals_bit_identical_to_oracle'])"
timeout 600 python bench.py --order row_major --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_9_bench_c3_rowmajor.json 2>/dev/null
python -c "import json;d=json.loads(open('gpurun_out/r4_9_bench_c3_rowmajor.json').read().strip().splitlines()[-1]);print('c3 row-major', d['ms_per_step'])"
timeout 1200 python -m pytest tests/test_engine_gpu.py -q -m gpu -x -k "joined or blocked or full_size or bench_times" 2>&1 | tail -3
