cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
MALLOC_CHECK_=3 timeout 1300 python tests/fuzz_split.py 500000 1000000 --minutes 20 --families 0123456 --in-process-oracle > gpurun_out/r06_fuzz_soak_extended.log 2>&1
tail -1 gpurun_out/r06_fuzz_soak_extended.log
LPMP_STRESS_OVERLAP=1 timeout 500 python tests/stress_lockstep_mailbox.py 6 > gpurun_out/r06_stress_lockstep_overlapped_extended.log 2>&1
tail -1 gpurun_out/r06_stress_lockstep_overlapped_extended.log
