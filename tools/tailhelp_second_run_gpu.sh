# DEV TOOL (GPU box): the tail-help experiment's launch end (wave_tail.py on the -DLHIP_TAIL_HELP -DLHIP_WAVE_TIMES build) and a short randomised
# sweep of the -DLHIP_TAIL_HELP build against the oracle (two-channel families: the only ones it changes).  gpurun_out/r03th/.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03th; mkdir -p $O
timeout 40 python tests/tools/wave_tail.py 100000 2>&1 | grep -v "^$" | head -3 | tee $O/wave_tail_tailhelp.txt
export LAMEJS_HIP_LIB=$GRAFT_REPO_ROOT/lamejs_amd/lib/variants/liblamejs_hip_tailhelp.so
( timeout 38 python tests/tools/fuzz_gpu.py 260 77111 mpeg1 stereo 2>&1 | tail -1 | sed 's/^/mpeg1 stereo 260 77111: /' ) | tee $O/fuzz_tailhelp.txt
( timeout 30 python tests/tools/fuzz_gpu.py 160 77112 mpeg1 joint 2>&1 | tail -1 | sed 's/^/mpeg1 joint 160 77112: /' ) | tee -a $O/fuzz_tailhelp.txt
