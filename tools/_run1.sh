cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lockstep.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/lockstep_probe.py 1024 32 2 10 2>&1 | tail -1
timeout 900 python tools/lockstep_probe.py 1024 32 3 10 2>&1 | tail -1
