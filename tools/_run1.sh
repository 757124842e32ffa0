cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python tools/solver_loop_time.py 2>&1 | grep -E "^C2|^C3"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
