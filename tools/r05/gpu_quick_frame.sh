# DEV TOOL (GPU box): quick check of a change to the one-frame launch: the 1152-sample call pattern through Node (shipped library vs lamejs_amd/lib/variants/*.so) and a
# short randomised sweep of the one-frame-per-call path against the oracle.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_quick_frame; mkdir -p $O
cd $R
{
for spec in "120 770001 mpeg1 framecalls" "80 770002 lsf framecalls" "40 770003 resample framecalls" "60 770005 mpeg1 joint framecalls" "40 770006 mpeg1 reservoir framecalls"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 300 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz.txt
bash tools/r05/gpu_ab_calls.sh
