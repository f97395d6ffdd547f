# DEV TOOL (GPU box): after the launch-geometry change of the psychoacoustic kernels: GPU tier + a randomised sweep over all families
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_pass2; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
{
for spec in "600 670001 mpeg1" "400 670002 lsf" "200 670003 resample" "100 670004 lowrate" "300 670005 mpeg1 joint" "150 670006 mpeg1 reservoir" "100 670007 lsf joint reservoir" "300 670008 mpeg1 framecalls" "150 670009 lsf framecalls"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 600 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz.txt
