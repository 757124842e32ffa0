#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
# 1. the whole GPU suite
( time timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_11_pytest.log 2>&1 ) 2> gpurun_out/r4_11_pytest_time.txt
echo "pytest rc=$?"; tail -4 gpurun_out/r4_11_pytest.log; tail -3 gpurun_out/r4_11_pytest_time.txt
# 2. 8 ranks on the one GPU, full size (plain launches: the ranks share the device)
( time timeout 1700 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_11_bench_gpus8_c3.json 2> gpurun_out/r4_11_bench_gpus8_c3.err ) 2> gpurun_out/r4_11_time_c3.txt
echo "8-rank c3 rc=$?"; tail -3 gpurun_out/r4_11_time_c3.txt; rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
# 3. the MpRoundingSolver cycle through the off-load adapter
g++ -std=c++17 -O2 -I lp_mp_amd/include -I tests/cpp tools/offload_solver_loop.cpp -L lp_mp_amd/csrc -llpmp_engine -Wl,-rpath,$PWD/lp_mp_amd/csrc -o build/offload_solver_loop
timeout 900 ./build/offload_solver_loop --grid 1024 --labels 32 --iterations 60 --warm 25 --rounding 1 > gpurun_out/r4_11_solver_cycle.json 2> gpurun_out/r4_11_solver_cycle.err
echo "cycle rc=$?"; cat gpurun_out/r4_11_solver_cycle.json; tail -3 gpurun_out/r4_11_solver_cycle.err
# 4. setup time of the periodic chains
LPMP_ROT_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_11_bench_c3.json 2> gpurun_out/r4_11_bench_c3.err
grep "lpmp:" gpurun_out/r4_11_bench_c3.err | head -20
python -c "import json;d=json.loads(open('gpurun_out/r4_11_bench_c3.json').read().strip().splitlines()[-1]);print('c3', d['ms_per_step'], d['setup_s'], d['oracle_check'])"
