R=$GRAFT_REPO_ROOT; cd $R
for m in "NFR=40" "NFR=1" "DEVSYNC=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=2" "AMD_DIRECT_DISPATCH=0"; do env $m timeout 120 python tools/r06/alias_exit_check.py stream 2>&1 | tail -2; echo "env $m rc=$?"; done
