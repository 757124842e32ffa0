#!/bin/bash
# tools/profile_round6.sh [part ...] — round 6 on the GPU box (everything under gpurun_out/; tools/profile_collect.py + copies go into profiles/).
#   counters   C3 and C4 bench lines plain / under the kernel trace / under the two PMC passes (program directly after `--`), C5 lines in both orders
#   probes     overlap and lock-step probes (parts of one process on the one GPU), the C5 probe with the overlapped program, the C++ hosts
#   solver     the reference-shaped Solve loop through offloaded<>: colour-major, row-major and the engine's suggested order
#   sizes      the headline kernel away from the headline size: grids of 1536^2 ... 3072^2 (167 GB), 33 ... 256 labels, C4 at four times the size
#   ranks      bench.py --gpus 2 / 8 for c3, c4, c5 on the one GPU (gloo, persistent launches off: their timings mean nothing; what is
#              checked is that an N-rank launch runs, its self test passes and the line carries the per-rank keys)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
parts=${*:-counters probes solver sizes ranks}
for part in $parts; do case $part in
counters)
  bash tools/profile_round.sh r06d --steps 20 --warmup 5
  bash tools/profile_round.sh r06d_c4 --workload c4 --steps 10 --warmup 3
  timeout 600 python bench.py --workload c5 --steps 10 --warmup 3 > gpurun_out/r06d_c5_bench_c5_default.json 2>/dev/null
  timeout 600 python bench.py --workload c5 --c5-order colour_major --steps 20 --warmup 5 > gpurun_out/r06d_c5_bench_c5_colour_major.json 2>/dev/null
  timeout 600 python bench.py --workload c5 --c5-order suggested --steps 20 --warmup 5 > gpurun_out/r06d_c5_bench_c5_suggested.json 2>/dev/null
  timeout 600 python bench.py --grid 512 --labels 8 --pairwise potts --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_c2.json 2>/dev/null
  timeout 600 python bench.py --order row_major --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_c3_row_major.json 2> gpurun_out/r06_bench_c3_row_major.err
  ;;
probes)
  for p in 1 2; do timeout 600 python tools/overlap_probe.py 1024 32 $p 10 12; done > gpurun_out/r06_overlap_probe.txt 2>/dev/null
  timeout 1500 python tools/lockstep_graph_probe.py 2000000 10000000 16 8 6 > gpurun_out/r06_lockstep_graph_probe_full_size.json 2>/dev/null
  for p in 4 8; do timeout 1200 python tools/lockstep_c5_probe.py $p 6 2>/dev/null; done > gpurun_out/r06_lockstep_c5_probe.txt
  python -c "from lp_mp_amd import build as B; B.build_mgpu_driver()" > /dev/null
  ( export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
    ./build/mgpu_rccl_driver --H 1024 --W 1024 --L 32 --parts-per-rank 2 --schedule overlap --passes 5 --time 10 2>/dev/null | grep driver
    ./build/mgpu_rccl_driver --H 1024 --W 1024 --L 32 --parts-per-rank 2 --schedule lockstep --passes 2 --time 10 2>/dev/null | grep driver
    ./build/mgpu_rccl_driver --graph 2000000 10000000 --L 16 --parts-per-rank 8 --schedule lockstep --passes 2 --time 6 2>/dev/null | grep driver ) > gpurun_out/r06_cpp_host_driver.txt
  ;;
solver)
  g++ -std=c++17 -O2 -I lp_mp_amd/include -I tests/cpp tools/offload_solver_loop.cpp -L lp_mp_amd/csrc -llpmp_engine -Wl,-rpath,$PWD/lp_mp_amd/csrc -o build/offload_solver_loop
  for o in colour_major row_major suggested; do
    timeout 1500 ./build/offload_solver_loop --grid 1024 --labels 32 --iterations 64 --warm 24 --order $o 2>> gpurun_out/r06_solver_orders.err
  done > gpurun_out/r06_solver_orders.json
  timeout 900 ./build/offload_solver_loop --grid 1024 --labels 32 --iterations 60 --warm 26 --rounding 1 > gpurun_out/r06_solver_cycle.json 2>/dev/null
  ;;
sizes)
  for g in 1536 2048 3072; do LPMP_ROT_VERBOSE=1 timeout 900 python bench.py --grid $g --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_c3_grid${g}.json 2> gpurun_out/r06_bench_c3_grid${g}.err; done
  for cfg in "1024 33" "1024 40" "1024 48" "1024 64" "768 96" "512 128" "256 256" "2048 16"; do set -- $cfg
    timeout 900 python bench.py --grid $1 --labels $2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_c3shape_grid$1_L$2.json 2>/dev/null; done
  bash tools/tile_sweep.sh > gpurun_out/r06_tile_sweep.txt 2>&1
  timeout 1200 python bench.py --workload c4 --c4-nodes 8000000 --c4-edges 40000000 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_c4_x4_8M_40M.json 2>/dev/null
  for f in gpurun_out/r06_bench_c3_grid*.json gpurun_out/r06_bench_c3shape_*.json gpurun_out/r06_bench_c4_x4_8M_40M.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 3), d["roofline"]["kernel"], round(d["peak_device_memory_GB_rank0"], 1))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done > gpurun_out/r06_sizes_summary.txt; cat gpurun_out/r06_sizes_summary.txt
  ;;
ranks)
  # two ranks hand DEVICE tensors to the backend (gloo's CUDA collectives): the code path of an RCCL run with real data between two processes
  ( export LPMP_DIST_DEVICE_COLLECTIVES=1
    timeout 900 python bench.py --gpus 2 --workload c5 --c5-small --steps 4 --warmup 2 --no-cpu-baseline --overlap-exchange > gpurun_out/r06_bench_gpus2_c5_small_device_collectives_overlapped.json 2>/dev/null
    timeout 900 python bench.py --gpus 2 --grid 512 --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_gpus2_c3_512_device_collectives.json 2>/dev/null )
  for n in 2 8; do
    ( time timeout 1700 python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_gpus${n}_c3.json 2> gpurun_out/r06_bench_gpus${n}_c3.err ) 2> gpurun_out/r06_bench_gpus${n}_c3.time
    ( time timeout 2400 python bench.py --gpus $n --workload c4 --no-compare-schedules --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_gpus${n}_c4.json 2> gpurun_out/r06_bench_gpus${n}_c4.err ) 2> gpurun_out/r06_bench_gpus${n}_c4.time
    ( time timeout 2400 python bench.py --gpus $n --workload c5 --c5-order colour_major --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_gpus${n}_c5.json 2> gpurun_out/r06_bench_gpus${n}_c5.err ) 2> gpurun_out/r06_bench_gpus${n}_c5.time
  done
  ( time timeout 2400 python bench.py --gpus 8 --workload c5 --steps 4 --warmup 2 --no-cpu-baseline --overlap-exchange > gpurun_out/r06_bench_gpus8_c5_index_overlapped.json 2> gpurun_out/r06_bench_gpus8_c5_index_overlapped.err ) 2> gpurun_out/r06_bench_gpus8_c5_index_overlapped.time
  for f in gpurun_out/r06_bench_gpus*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["n_gpus"], round(d["ms_per_step"], 3), d["dual_bound_gap"], d["launch"]["self_test"], d["compute_ms_per_pass"], d["exchange_ms_per_pass"], d["slowest_rank"], d["config"]["partitioner"], d["setup_s"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
  ;;
esac; done
ls -la gpurun_out | tail -40
