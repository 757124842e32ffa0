cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mailbox_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_fuzz_gpu.py tests/test_primal_gpu.py tests/test_send_rules_gpu.py -x -q -m gpu 2>&1 | tail -3
