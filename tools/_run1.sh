cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mailbox_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
