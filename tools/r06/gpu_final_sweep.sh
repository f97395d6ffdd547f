# DEV TOOL (GPU box): randomised sweeps on the round's last library (after the look-ahead was switched off and the count helper's record grew)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final_sweep; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
{
echo "GPU fuzz on the last library of round 6 (tests/tools/fuzz_gpu.py <n> <seed> <family> [framecalls]: random material, random chunking / one frame per call, GPU output vs the CPU oracle)"
for spec in "1200 690001 mpeg1" "600 690002 lsf" "300 690003 resample" "200 690004 lowrate" "500 690005 mpeg1 joint" "300 690006 mpeg1 reservoir" "200 690007 lsf joint reservoir" \
            "700 690011 mpeg1 framecalls" "400 690012 lsf framecalls" "200 690013 resample framecalls" "300 690015 mpeg1 joint framecalls" "300 690016 mpeg1 reservoir framecalls" "150 690017 lsf joint reservoir framecalls"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 600 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz.txt
