#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_16_pytest.log 2>&1 ) 2> gpurun_out/r4_16_pytest_time.txt
echo "pytest rc=$?"; tail -4 gpurun_out/r4_16_pytest.log; tail -3 gpurun_out/r4_16_pytest_time.txt
R03=$PWD/build/exp/liblpmp_engine_r03.so
for i in 1 2; do
  LPMP_ENGINE_SO=$R03 python tools/row_major_time.py 1024 32 dense 10 2>/dev/null | tail -1
  python tools/row_major_time.py 1024 32 dense 10 2>/dev/null | tail -1
  LPMP_ENGINE_SO=$R03 python tools/row_major_time.py 512 8 potts 20 2>/dev/null | tail -1
  python tools/row_major_time.py 512 8 potts 20 2>/dev/null | tail -1
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_16_bench_c3.json 2>/dev/null
python -c "import json;d=json.loads(open('gpurun_out/r4_16_bench_c3.json').read().strip().splitlines()[-1]);print('c3', d['ms_per_step'], d['setup_s'], d['oracle_check']['duals_bit_identical_to_oracle'])"
LPMP_ENGINE_SO=$R03 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c3 r03', d['ms_per_step'], d['setup_s'])"
timeout 600 python tools/c5_probe.py 2>/dev/null | tail -3
LPMP_ENGINE_SO=$R03 timeout 600 python tools/c5_probe.py 2>/dev/null | tail -3
