#!/bin/bash
# tools/profile_round4.sh — round 4 on the GPU box: C3 and C4 bench lines plain / under the kernel trace / under the two PMC passes, the overlap and lock-step probes, the solver cycle, N ranks on the one GPU (everything under gpurun_out/; profile_collect.py + copies go into profiles/)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile_round.sh r04b --steps 20 --warmup 5
bash tools/profile_round.sh r04b_c4 --workload c4 --steps 10 --warmup 3
# probes
for parts in 1 2 3; do timeout 600 python tools/overlap_probe.py 1024 32 $parts 10 12; done > gpurun_out/r04_overlap_probe.txt 2>/dev/null
timeout 600 python tools/overlap_probe.py 1024 32 2 16 18 >> gpurun_out/r04_overlap_probe.txt 2>/dev/null
timeout 600 python tools/lockstep_probe.py 1024 32 2 10 >> gpurun_out/r04_overlap_probe.txt 2>/dev/null
timeout 600 python tools/mgpu_local_bench.py 2 1024 10 2>/dev/null | grep "ms per pass per strip" >> gpurun_out/r04_overlap_probe.txt
for parts in 4 8; do timeout 1200 python tools/lockstep_graph_probe.py 200000 1000000 16 $parts 8 colour_major; done > gpurun_out/r04_lockstep_graph_probe.txt 2>/dev/null
timeout 1200 python tools/lockstep_graph_probe.py 200000 1000000 16 4 8 index >> gpurun_out/r04_lockstep_graph_probe.txt 2>/dev/null
timeout 1500 python tools/lockstep_graph_probe.py 2000000 10000000 16 8 6 > gpurun_out/r04_lockstep_graph_probe_full_size.json 2>/dev/null
for parts in 4 8; do timeout 800 python tools/lockstep_c5_probe.py $parts 6 2>/dev/null; done > gpurun_out/r04_lockstep_c5_probe.txt
# the C++ host (parts of one rank on the one GPU): overlap and lock-step strips of the headline grid, lock step on the C4 graph (index order, contiguous parts)
python -c "from lp_mp_amd import build as B; B.build_mgpu_driver()" > /dev/null
( export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
  ./build/mgpu_rccl_driver --H 1024 --W 1024 --L 32 --parts-per-rank 2 --schedule overlap --passes 5 --time 10 2>/dev/null | grep driver
  ./build/mgpu_rccl_driver --H 1024 --W 1024 --L 32 --parts-per-rank 2 --schedule lockstep --passes 2 --time 10 2>/dev/null | grep driver
  ./build/mgpu_rccl_driver --H 1024 --W 1024 --L 32 --parts-per-rank 2 --passes 2 --time 10 2>/dev/null | grep driver
  ./build/mgpu_rccl_driver --graph 2000000 10000000 --L 16 --parts-per-rank 8 --schedule lockstep --passes 2 --time 6 2>/dev/null | grep driver
  python - <<'PY'
import numpy as np
from lp_mp_amd import ordering as O, multi_gpu as MG, synthetic as S
n, m = 2000000, 10000000
r = O.colour_major_order(n, *S.counter_graph_edges(n, m, 1), seed=1)
r.astype(np.int64).tofile("/tmp/c4_order.bin"); MG.graph_partition(n, *S.counter_graph_edges(n, m, 1, r), 8).astype(np.int64).tofile("/tmp/c4_part.bin")
PY
  ./build/mgpu_rccl_driver --graph 2000000 10000000 --L 16 --parts-per-rank 8 --schedule lockstep --passes 2 --time 6 --order-file /tmp/c4_order.bin --part-file /tmp/c4_part.bin 2>/dev/null | grep driver
  python - <<'PY'
import numpy as np
from lp_mp_amd import multi_gpu as MG, synthetic as S
gm = S.c5_model(512, 512, 8, 150000, 70000, 30000, seed=4, window=64)
gm.dump("/tmp/c5_model.bin"); MG.graph_partition_model(gm, 8).astype(np.int64).tofile("/tmp/c5_part8.bin")
PY
  ./build/mgpu_rccl_driver --schedule lockstep --model-file /tmp/c5_model.bin --part-file /tmp/c5_part8.bin --parts-per-rank 8 --passes 2 --time 6 2>/dev/null | grep driver ) > gpurun_out/r04_cpp_host_driver.txt
# the solver cycle
g++ -std=c++17 -O2 -I lp_mp_amd/include -I tests/cpp tools/offload_solver_loop.cpp -L lp_mp_amd/csrc -llpmp_engine -Wl,-rpath,$PWD/lp_mp_amd/csrc -o build/offload_solver_loop
timeout 900 ./build/offload_solver_loop --grid 1024 --labels 32 --iterations 60 --warm 25 --rounding 1 > gpurun_out/r04_solver_cycle.json 2>/dev/null
timeout 900 ./build/offload_solver_loop --grid 1024 --labels 32 --iterations 64 --warm 40 > gpurun_out/r04_solver_plain.json 2>/dev/null
# N ranks on the one GPU
( time timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_gpus2_c3.json 2> gpurun_out/r04_bench_gpus2_c3.err ) 2> gpurun_out/r04_bench_gpus2_c3.time
( time timeout 1700 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_gpus8_c3.json 2> gpurun_out/r04_bench_gpus8_c3.err ) 2> gpurun_out/r04_bench_gpus8_c3.time
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" > gpurun_out/r04_bench_gpus8_c3.vram
( time timeout 2400 python bench.py --gpus 8 --workload c4 --schedule boundary --no-compare-schedules --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_gpus8_c4_boundary.json 2> gpurun_out/r04_bench_gpus8_c4_boundary.err ) 2> gpurun_out/r04_bench_gpus8_c4_boundary.time
( time timeout 2400 python bench.py --gpus 8 --workload c4 --no-compare-schedules --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_gpus8_c4_lockstep.json 2> gpurun_out/r04_bench_gpus8_c4_lockstep.err ) 2> gpurun_out/r04_bench_gpus8_c4_lockstep.time
( time timeout 1500 python bench.py --gpus 2 --workload c4 --schedule lockstep --c4-nodes 200000 --c4-edges 1000000 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_gpus2_c4_lockstep_small.json 2>/dev/null ) 2>/dev/null
# other configs on one GPU
timeout 600 python bench.py --grid 512 --labels 8 --pairwise potts --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_c2.json 2>/dev/null
timeout 600 python bench.py --order row_major --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_c3_row_major.json 2>/dev/null
timeout 600 python tools/c5_probe.py > gpurun_out/r04_c5_probe.log 2>/dev/null
for f in gpurun_out/r04_bench_gpus*.time; do echo $f; grep real $f; done
for f in gpurun_out/r04_bench_gpus*.json gpurun_out/r04_bench_c2.json gpurun_out/r04_bench_c3_row_major.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["n_gpus"], round(d["ms_per_step"], 3), d["dual_bound_gap"], d.get("schedules"), d.get("setup_s"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat gpurun_out/r04_overlap_probe.txt gpurun_out/r04_lockstep_graph_probe.txt gpurun_out/r04_lockstep_graph_probe_full_size.json gpurun_out/r04_lockstep_c5_probe.txt gpurun_out/r04_cpp_host_driver.txt gpurun_out/r04_solver_cycle.json gpurun_out/r04_solver_plain.json | cut -c1-700
