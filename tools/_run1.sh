cd $GRAFT_REPO_ROOT
for k in 1 2; do
LPMP_ENGINE_SO=build/exp/liblpmp_engine_mb_base.so timeout 300 python tools/row_major_time.py 2>&1 | tail -1
timeout 300 python tools/row_major_time.py 2>&1 | tail -1
done
timeout 300 python tools/chain_probe.py 1024 32 dense 10 2>&1 | tail -1
timeout 600 python tools/chain_trace.py run 1024 32 row_major 2>&1 | tail -14
mkdir -p gpurun_out; cp /tmp/chain_trace.bin gpurun_out/chain_trace_c3rm.bin
timeout 900 python -m pytest tests/test_mailbox_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "mailbox or chain or level or deep or row_major or many" 2>&1 | tail -2
