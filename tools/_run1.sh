cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>&1 | tail -1 | cut -c1-300
