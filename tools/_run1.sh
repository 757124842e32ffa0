cd $GRAFT_REPO_ROOT
LPMP_ROT_VERBOSE=1 timeout 900 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -E "table stream|^\{" | cut -c1-200
