# DEV TOOL (GPU box): the randomised GPU-vs-oracle sweep on the final code of round 3 (tests/tools/fuzz_gpu.py); log in gpurun_out/r03fz/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03fz; mkdir -p $O
cd $R
{
echo "GPU fuzz on the final code of round 3 (tests/tools/fuzz_gpu.py <n> <seed> <family>: random material -- tones, filtered / white noise, clicks,"
echo "bursts, silence gaps and level steps -- random chunking, GPU output vs the CPU oracle, which is pinned to the live reference by"
echo "tests/tools/fuzz_ref.py on the same generator); git HEAD $(cat $R/gpurun_out/.head 2>/dev/null)"
for spec in "3000 535351 mpeg1" "1600 535352 lsf" "800 535353 resample" "400 535354 lowrate" "1200 535355 mpeg1 joint" "510 535356 mpeg1 reservoir"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 900 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz_gpu_final_code.txt
