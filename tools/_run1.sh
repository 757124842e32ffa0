cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "chain_executor or many_levels or tiny_levels or multi_pass or residual or primal or few_big" 2>&1 | tail -6
timeout 600 python tools/chain_probe.py 1024 32 dense 10 2>&1 | tail -3
LPMP_STRANDS=0 timeout 600 python tools/chain_probe.py 1024 32 dense 10 2>&1 | tail -3
