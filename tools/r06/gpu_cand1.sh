# DEV TOOL (GPU box): candidate helpers (next-gain evaluation beside the owner's), first device run: one-frame sweeps against the oracle, then the 1152-sample call pattern shipped vs variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_cand1; mkdir -p $O
cd $R
{
for spec in "150 660001 mpeg1 framecalls" "100 660002 lsf framecalls" "80 660003 mpeg1 joint framecalls" "60 660004 mpeg1 reservoir framecalls" "40 660005 lowrate framecalls" "40 660006 resample framecalls"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 600 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz.txt
bash tools/r05/gpu_ab_calls.sh 2>&1 | tee $O/calls_ab.txt
