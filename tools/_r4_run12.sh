#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_rows_layout_gpu.py tests/test_speculation_gpu.py tests/test_multi_gpu.py -q -m gpu -x > gpurun_out/r4_12_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r4_12_pytest.log
# C4 A/B in ONE box: packed vs rows layout, alternating processes
for i in 1 2; do
  for lay in off on; do
    timeout 900 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --rows-layout $lay > gpurun_out/r4_12_c4_${lay}_$i.json 2> gpurun_out/r4_12_c4_${lay}_$i.err
    python -c "import json;d=json.loads(open('gpurun_out/r4_12_c4_${lay}_$i.json').read().strip().splitlines()[-1]);print('c4 rows $lay', d['ms_per_step'], d['roofline']['frac'], d['lower_bound_after'], d['setup_s'])" 2>&1 | tail -1
  done
done
# C3 with rows (does it cost the headline anything / gain?)
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rows-layout on > gpurun_out/r4_12_c3_rows.json 2> gpurun_out/r4_12_c3_rows.err
python -c "import json;d=json.loads(open('gpurun_out/r4_12_c3_rows.json').read().strip().splitlines()[-1]);print('c3 rows on', d['ms_per_step'], d['oracle_check'])" 2>&1 | tail -1
LPMP_PLAN_TIMES=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_12_c3.json 2> gpurun_out/r4_12_c3.err
grep "make_schedule" gpurun_out/r4_12_c3.err | head -40
python -c "import json;d=json.loads(open('gpurun_out/r4_12_c3.json').read().strip().splitlines()[-1]);print('c3', d['ms_per_step'], d['setup_s'])"
