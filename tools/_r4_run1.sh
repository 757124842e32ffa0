#!/bin/bash
# round 4, first GPU call: overlap parity on the device, the self-launching bench, the overlap probe at full size
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_overlap.py tests/test_bench_contract.py -q -m gpu -x > gpurun_out/r4_1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_1_pytest.log
for parts in 1 2 3; do
  timeout 600 python tools/overlap_probe.py 1024 32 $parts 10 12 >> gpurun_out/r4_1_overlap_probe.log 2>&1
done
timeout 600 python tools/overlap_probe.py 1024 32 2 20 22 >> gpurun_out/r4_1_overlap_probe.log 2>&1
timeout 600 python tools/overlap_probe.py 1024 32 2 16 18 >> gpurun_out/r4_1_overlap_probe.log 2>&1
tail -5 gpurun_out/r4_1_pytest.log
cat gpurun_out/r4_1_overlap_probe.log
