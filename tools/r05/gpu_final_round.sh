# DEV TOOL (GPU box): the round's closing measurements on the final code: the driver's sequence (smoke, GPU tier, default bench line), the one-frame profile,
# the hand-over legs, kernel statistics of the headline configuration, and randomised sweeps of the paths this round's second half changed.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_final_round; mkdir -p $O
cd $R
bash tools/r05/gpu_final_check.sh 2>&1 | tee $O/final_check.txt
cp $R/gpurun_out/r05_final_check/bench.json $O/bench.json; cp $R/gpurun_out/r05_final_check/pytest_gpu.txt $O/pytest_gpu.txt
timeout 200 python tests/tools/frame_prof.py 200 > $O/frame_prof.txt 2>&1; grep -E "^==|g_frame launch" $O/frame_prof.txt
timeout 100 python tests/tools/handoff_prof.py > $O/handoff_prof.txt 2>&1; tail -4 $O/handoff_prof.txt | cut -c1-200
{
echo "GPU fuzz on the final code of round 5 (tests/tools/fuzz_gpu.py <n> <seed> <family> [framecalls])"
for spec in "700 990001 mpeg1 framecalls" "400 990002 lsf framecalls" "200 990003 resample framecalls" "120 990004 lowrate framecalls" "300 990005 mpeg1 joint framecalls" "200 990006 mpeg1 reservoir framecalls" "150 990007 lsf joint reservoir framecalls" \
            "500 990011 mpeg1" "300 990012 lsf" "150 990013 resample" "250 990015 mpeg1 joint" "300 990016 mpeg1 reservoir" "150 990017 lsf joint reservoir"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 600 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz.txt
