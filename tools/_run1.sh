cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --workload c4 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r03_bench_c4_last_binaries.json
timeout 600 python bench.py --grid 512 --labels 8 --pairwise potts --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r03_bench_c2_last_binaries.json
timeout 600 python bench.py --grid 512 --labels 8 --pairwise potts --order row_major --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_c2_row_major_last_binaries.json
timeout 900 python tools/c5_probe.py > gpurun_out/r03_c5_probe_last_binaries.log 2>&1
for f in r03_bench_c4_last_binaries r03_bench_c2_last_binaries r03_bench_c2_row_major_last_binaries; do python3 -c "
import json; d=json.load(open('gpurun_out/$f.json')); print('$f', round(d['ms_per_step'],3), d['value'], d['roofline'].get('frac'))"; done
tail -4 gpurun_out/r03_c5_probe_last_binaries.log
