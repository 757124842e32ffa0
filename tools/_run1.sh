cd $GRAFT_REPO_ROOT
for k in 1 2; do
timeout 300 python tools/row_major_time.py 2>&1 | tail -1
LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_k1s2.so timeout 300 python tools/row_major_time.py 2>&1 | tail -1
done
LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_k1s2.so timeout 600 python tools/chain_trace.py run 1024 32 row_major 2>&1 | tail -9
LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_k1s2.so timeout 300 python tools/chain_probe.py 1024 32 dense 5 2>&1 | tail -1 | cut -c1-200
