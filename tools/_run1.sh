cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r03b --steps 20 --warmup 5 2>&1 | tail -2
