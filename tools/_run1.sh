cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py tests/test_fuzz_gpu.py tests/test_send_rules_gpu.py -x -q -m gpu -k "tiny_levels or c5 or labeling or multicut or fuzz or send_rules" 2>&1 | tail -8
for k in 1 2; do
python tools/c5_probe.py 2>&1 | grep "window 64:" 
LPMP_ENGINE_SO=$PWD/build/exp/liblpmp_engine_r02.so python tools/c5_probe.py 2>&1 | grep "window 64:"
done
python tools/level_trace.py 2>&1 | tail -6
