cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tests/stress_lockstep_mailbox.py 4 2>&1 | tail -3 | tee gpurun_out/r03_stress_lockstep_mailbox.log
