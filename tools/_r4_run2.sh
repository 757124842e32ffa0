#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_overlap.py -q -m gpu -x -k full_size > gpurun_out/r4_2_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_2_pytest.log
tail -3 gpurun_out/r4_2_pytest.log
( time timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_2_bench_gpus2.json 2> gpurun_out/r4_2_bench_gpus2.err ) 2>&1 | tail -4
echo "bench rc=$?"
tail -c 3000 gpurun_out/r4_2_bench_gpus2.json
tail -5 gpurun_out/r4_2_bench_gpus2.err
