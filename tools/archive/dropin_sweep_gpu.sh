# DEV TOOL (GPU box): chunk-schedule sweep of the host-buffer path (tests/tools/dropin_sweep.py); log in gpurun_out/r03sw/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03sw
timeout 500 python tests/tools/dropin_sweep.py 2>&1 | tee gpurun_out/r03sw/dropin_sweep.txt
