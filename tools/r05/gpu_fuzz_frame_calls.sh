# DEV TOOL (GPU box): randomised sweep of the one-frame-per-call path (g_frame with count helpers, psyB beside the search) on the device: every case fed a frame's worth of
# samples per call, all families and both extensions, GPU output vs the CPU oracle.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_fuzz_frame_calls; mkdir -p $O
cd $R
{
echo "GPU fuzz of the one-frame-per-call path on the final code of round 5 (tests/tools/fuzz_gpu.py <n> <seed> <family> framecalls)"
for spec in "900 660001 mpeg1 framecalls" "500 660002 lsf framecalls" "250 660003 resample framecalls" "150 660004 lowrate framecalls" "400 660005 mpeg1 joint framecalls" "250 660006 mpeg1 reservoir framecalls" "200 660007 lsf joint reservoir framecalls"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 600 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz.txt
