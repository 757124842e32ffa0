#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
( time timeout 1700 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_5_bench_gpus8_c3.json 2> gpurun_out/r4_5_bench_gpus8_c3.err ) 2> gpurun_out/r4_5_time_c3.txt
echo "c3 rc=$?"; cat gpurun_out/r4_5_time_c3.txt
tail -c 1800 gpurun_out/r4_5_bench_gpus8_c3.json | cut -c1-1800
grep -v "hostname of the client\|^\[Gloo\]\|amdgpu.ids\|^\[rank . stdout\]$" gpurun_out/r4_5_bench_gpus8_c3.err | grep -i "error\|Traceback" | head -5
timeout 1500 python -m pytest tests/test_mailbox_gpu.py tests/test_multi_gpu.py tests/test_bench_contract.py tests/test_speculation_gpu.py tests/test_cpp_dropin.py -q -m gpu -x > gpurun_out/r4_5_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r4_5_pytest.log
g++ -std=c++17 -O2 -I lp_mp_amd/include -I tests/cpp tools/offload_solver_loop.cpp -L lp_mp_amd/csrc -llpmp_engine -Wl,-rpath,$PWD/lp_mp_amd/csrc -o build/offload_solver_loop
timeout 900 ./build/offload_solver_loop --grid 1024 --labels 32 --iterations 60 --warm 25 --rounding 1 > gpurun_out/r4_5_solver_cycle.json 2> gpurun_out/r4_5_solver_cycle.err
echo "cycle rc=$?"; cat gpurun_out/r4_5_solver_cycle.json; tail -3 gpurun_out/r4_5_solver_cycle.err
timeout 900 ./build/offload_solver_loop --grid 1024 --labels 32 --iterations 64 --warm 40 > gpurun_out/r4_5_solver_plain.json 2>> gpurun_out/r4_5_solver_cycle.err
cat gpurun_out/r4_5_solver_plain.json
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_5_bench_c3.json 2> gpurun_out/r4_5_bench_c3.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_5_bench_c3.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","setup_s","rounding","cpu_baseline","oracle_check")})
PY
