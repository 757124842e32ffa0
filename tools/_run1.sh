cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "chain or level or deep or row_major or many" 2>&1 | tail -5
timeout 300 python tools/chain_probe.py 1024 32 dense 10 2>&1 | tail -2
timeout 300 python tools/chain_probe.py 512 8 dense 10 2>&1 | tail -2
