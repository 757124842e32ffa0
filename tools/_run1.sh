cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mailbox_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "chain or level or deep or row_major or many or labels" 2>&1 | tail -2
timeout 300 python tools/chain_probe.py 1024 21 dense 10 2>&1 | tail -1
timeout 300 python tools/chain_probe.py 1024 7 potts 10 2>&1 | tail -1
