#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
( time timeout 1700 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_4_bench_gpus8_c3.json 2> gpurun_out/r4_4_bench_gpus8_c3.err ) 2> gpurun_out/r4_4_time_c3.txt
echo "c3 rc=$?"; cat gpurun_out/r4_4_time_c3.txt
tail -c 2500 gpurun_out/r4_4_bench_gpus8_c3.json
grep -v "hostname of the client\|^\[Gloo\]\|amdgpu.ids\|^\[rank . stdout\]$" gpurun_out/r4_4_bench_gpus8_c3.err | tail -5
timeout 1500 python -m pytest tests/test_mailbox_gpu.py tests/test_multi_gpu.py tests/test_bench_contract.py tests/test_speculation_gpu.py -q -m gpu -x > gpurun_out/r4_4_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r4_4_pytest.log
