# DEV TOOL (GPU box): tests/tools/wave_tail.py with the -DLHIP_WAVE_TIMES build (lamejs_amd/lib/variants/liblamejs_hip_wavetimes.so); gpurun_out/r03wt/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03wt
timeout 300 python tests/tools/wave_tail.py 2>&1 | tee gpurun_out/r03wt/wave_tail.txt
