cd $GRAFT_REPO_ROOT
python -m pytest tests/test_multi_gpu.py -x -q -m gpu -k "cpp_rccl" 2>&1 | tail -15
for v in base notab norv nosv novec nolb nored; do
  if [ $v = base ]; then unset LPMP_ENGINE_SO; else export LPMP_ENGINE_SO=$PWD/build/exp/liblpmp_engine_$v.so; fi
  echo "== $v"; timeout 600 python tools/c4_probe.py 2000000 10000000 16 20 2>&1 | grep -E "ms/pass|Error|error" 
done
