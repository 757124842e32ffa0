#!/bin/bash
# 8 rank processes on the ONE GPU of the box (gloo, CPU-staged exchanges): the N-rank code path at full size.  Timings mean nothing.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
( time timeout 1700 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_3_bench_gpus8_c3.json 2> gpurun_out/r4_3_bench_gpus8_c3.err ) 2> gpurun_out/r4_3_time_c3.txt
echo "c3 rc=$?"; cat gpurun_out/r4_3_time_c3.txt; rocm-smi --showmeminfo vram 2>/dev/null | tail -3
( time timeout 2400 python bench.py --gpus 8 --workload c4 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r4_3_bench_gpus8_c4_boundary.json 2> gpurun_out/r4_3_bench_gpus8_c4_boundary.err ) 2> gpurun_out/r4_3_time_c4b.txt
echo "c4 boundary rc=$?"; cat gpurun_out/r4_3_time_c4b.txt
( time timeout 2400 python bench.py --gpus 8 --workload c4 --schedule lockstep --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r4_3_bench_gpus8_c4_lockstep.json 2> gpurun_out/r4_3_bench_gpus8_c4_lockstep.err ) 2> gpurun_out/r4_3_time_c4l.txt
echo "c4 lockstep rc=$?"; cat gpurun_out/r4_3_time_c4l.txt
for f in gpurun_out/r4_3_bench_gpus8_*.json; do echo $f; tail -c 1500 $f; echo; done
for f in gpurun_out/r4_3_bench_gpus8_*.err; do echo $f; grep -v "hostname of the client\|Gloo\|amdgpu.ids" $f | tail -5; done
