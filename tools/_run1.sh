cd $GRAFT_REPO_ROOT
export LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_pre2.so
timeout 900 python -m pytest tests/test_mailbox_gpu.py -x -q -m gpu 2>&1 | tail -3
unset LPMP_ENGINE_SO
for k in 1 2; do
LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_base.so timeout 300 python tools/row_major_time.py 2>&1 | tail -1
LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_pre.so timeout 300 python tools/row_major_time.py 2>&1 | tail -1
LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_pre2.so timeout 300 python tools/row_major_time.py 2>&1 | tail -1
done
LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_pre2.so timeout 600 python tools/chain_trace.py run 1024 32 row_major 2>&1 | tail -9
