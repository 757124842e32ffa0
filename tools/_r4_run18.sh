#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_18_pytest.log 2>&1 ) 2> gpurun_out/r4_18_pytest_time.txt
echo "pytest rc=$?"; tail -6 gpurun_out/r4_18_pytest.log
R03=$PWD/build/exp/liblpmp_engine_r03.so
for i in 1 2; do
  LPMP_ENGINE_SO=$R03 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c3 r03', d['ms_per_step'])"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c3 now', d['ms_per_step'], d['setup_s'], d['rounding'])"
  LPMP_ENGINE_SO=$R03 python tools/row_major_time.py 1024 32 dense 10 2>/dev/null | tail -1
  python tools/row_major_time.py 1024 32 dense 10 2>/dev/null | tail -1
  LPMP_ENGINE_SO=$R03 python tools/row_major_time.py 512 8 potts 20 2>/dev/null | tail -1
  python tools/row_major_time.py 512 8 potts 20 2>/dev/null | tail -1
done
g++ -std=c++17 -O2 -I lp_mp_amd/include -I tests/cpp tools/offload_solver_loop.cpp -L lp_mp_amd/csrc -llpmp_engine -Wl,-rpath,$PWD/lp_mp_amd/csrc -o build/offload_solver_loop
timeout 900 ./build/offload_solver_loop --grid 1024 --labels 32 --iterations 60 --warm 25 --rounding 1 2>/dev/null
