cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mailbox_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "chain or level or deep or row_major or many" 2>&1 | tail -3
timeout 300 python tools/chain_probe.py 1024 32 dense 10 2>&1 | tail -1
for k in 1 2; do
LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_base.so timeout 300 python tools/row_major_time.py 2>&1 | tail -1
timeout 300 python tools/row_major_time.py 2>&1 | tail -1
done
