R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_cand_prof; mkdir -p $O; cd $R
for v in never all; do cp lamejs_amd/lib/prof_$v.so lamejs_amd/lib/liblamejs_hip_prof.so; timeout 200 python tests/tools/frame_prof.py 150 > $O/prof_$v.txt 2>&1; done
rm -f lamejs_amd/lib/liblamejs_hip_prof.so
bash tools/r05/gpu_ab_calls.sh > $O/calls_ab.txt 2>&1
