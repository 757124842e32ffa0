cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MALLOC_CHECK_=3 timeout 1500 python tests/fuzz_split.py 0 1000000 --minutes 10.5 --families 0123456 --in-process-oracle > gpurun_out/r03_fuzz_soak_last_binaries.log 2>&1
tail -3 gpurun_out/r03_fuzz_soak_last_binaries.log
LPMP_CHAIN_MIN=2 LPMP_CHAIN_ALL=1 MALLOC_CHECK_=3 timeout 900 python tests/fuzz_split.py 500000 1000000 --minutes 6 --families 0123456 --in-process-oracle > gpurun_out/r03_fuzz_chains_and_mailbox_forced.log 2>&1
tail -3 gpurun_out/r03_fuzz_chains_and_mailbox_forced.log
