cd $GRAFT_REPO_ROOT
export LPMP_DIST_BACKEND=gloo
for sched in boundary lockstep; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --grid 256 --steps 6 --warmup 2 --no-cpu-baseline --schedule $sched 2>&1 | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sched', 'c3', d['n_gpus'], round(d['ms_per_step'],3), d['dual_bound_gap'], d['config']['parallelism'])"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --workload c4 --c4-nodes 20000 --c4-edges 100000 --steps 8 --warmup 2 --no-cpu-baseline --schedule $sched 2>&1 | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sched', 'c4', d['n_gpus'], round(d['ms_per_step'],3), d['dual_bound_gap'], d['config']['parallelism'])"
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 4 --grid 128 --steps 6 --warmup 2 --no-cpu-baseline --schedule lockstep 2>&1 | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lockstep c3 x4', d['n_gpus'], round(d['ms_per_step'],3), d['dual_bound_gap'], d['config']['parallelism'])"
