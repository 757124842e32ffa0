cd $GRAFT_REPO_ROOT
timeout 600 python tools/chain_trace.py run 1024 32 row_major 2>&1 | tail -2
mkdir -p gpurun_out; cp /tmp/chain_trace.bin gpurun_out/chain_trace_c3rm.bin
