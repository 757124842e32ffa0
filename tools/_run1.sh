cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_bench_contract.py -x -q -m gpu -k "distributed_branch" 2>&1 | tail -12
