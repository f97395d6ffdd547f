# DEV TOOL (GPU box): candidate helpers: their statistics in the production code (tests/tools/handoff_prof.py), a short one-frame sweep, the call pattern shipped vs variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_cand2; mkdir -p $O
cd $R
timeout 200 python tests/tools/handoff_prof.py 2>&1 | tee $O/handoff_prof.txt | cut -c1-400
echo "  fuzz -> $(timeout 600 python tests/tools/fuzz_gpu.py 120 660011 mpeg1 framecalls 2>&1 | tail -1)" | tee $O/fuzz.txt
bash tools/r05/gpu_ab_calls.sh 2>&1 | tee $O/calls_ab.txt
