cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a_bench_driver_command.log 2>&1
grep -h '^{"metric"' gpurun_out/r03a_bench_driver_command.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({k:d[k] for k in ('value','ms_per_step','oracle_check','setup_s','roofline','cpu_baseline')}, indent=1))"
