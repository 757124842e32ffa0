#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
for parts in 4 8; do
  timeout 1200 python tools/lockstep_graph_probe.py 200000 1000000 16 $parts 8 colour_major >> gpurun_out/r4_13_lockstep_graph.log 2>> gpurun_out/r4_13_lockstep_graph.err
done
timeout 1200 python tools/lockstep_graph_probe.py 200000 1000000 16 4 8 index >> gpurun_out/r4_13_lockstep_graph.log 2>> gpurun_out/r4_13_lockstep_graph.err
cat gpurun_out/r4_13_lockstep_graph.log
tail -3 gpurun_out/r4_13_lockstep_graph.err
timeout 900 python -m pytest tests/test_lockstep.py -q -m gpu -x 2>&1 | tail -3
