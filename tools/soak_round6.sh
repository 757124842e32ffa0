#!/bin/bash
# round 6: the >= 10-minute soak of the shipped configuration, the forced-chain soak, the lock-step / mailbox stress, with this round's last binaries
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
MALLOC_CHECK_=3 timeout 800 python tests/fuzz_split.py 0 1000000 --minutes 10.5 --families 0123456 --in-process-oracle > gpurun_out/r06_fuzz_soak_last_binaries.log 2>&1
tail -2 gpurun_out/r06_fuzz_soak_last_binaries.log
LPMP_CHAIN_MIN=2 LPMP_CHAIN_ALL=1 LPMP_BAND_MIN_BYTES=1000 MALLOC_CHECK_=3 timeout 500 python tests/fuzz_split.py 0 1000000 --minutes 5 --families 0123456 --in-process-oracle > gpurun_out/r06_fuzz_chains_and_mailbox_forced.log 2>&1
tail -2 gpurun_out/r06_fuzz_chains_and_mailbox_forced.log
LPMP_ROWS_LAYOUT=1 MALLOC_CHECK_=3 timeout 400 python tests/fuzz_split.py 0 1000000 --minutes 4 --families 0123456 --in-process-oracle > gpurun_out/r06_fuzz_rows_layout.log 2>&1
tail -2 gpurun_out/r06_fuzz_rows_layout.log
timeout 400 python tests/stress_lockstep_mailbox.py 4 > gpurun_out/r06_stress_lockstep_mailbox.log 2>&1
tail -2 gpurun_out/r06_stress_lockstep_mailbox.log
# the overlapped lock-step program and the rows layout under the lock-step stress (seeds of their own)
LPMP_STRESS_OVERLAP=1 LPMP_STRESS_ROWS=1 timeout 300 python tests/stress_lockstep_mailbox.py 3 > gpurun_out/r06_stress_lockstep_overlapped_rows.log 2>&1
tail -2 gpurun_out/r06_stress_lockstep_overlapped_rows.log
# the tiled ticket order of the joined passes forced on the fuzz models (LPMP_ROT_VERBOSE counts the launches that took it)
LPMP_ROT_BANDS=3 LPMP_ROT_TILES=5 LPMP_ROT_DEPTH=4 LPMP_ROT_VERBOSE=1 MALLOC_CHECK_=3 timeout 500 python tests/fuzz_split.py 700000 1000000 --minutes 6 --families 0123456 --in-process-oracle > gpurun_out/r06_fuzz_tiled_order_forced.log 2> gpurun_out/r06_fuzz_tiled_order_forced.err
tail -2 gpurun_out/r06_fuzz_tiled_order_forced.log; grep -c "tiled order" gpurun_out/r06_fuzz_tiled_order_forced.err
