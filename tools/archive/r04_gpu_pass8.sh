# DEV TOOL (GPU box), round 4 pass 8: workgroup-aggregated counters in g_validate_fast -- the GPU tests around the seed chain, then the A/B of pass 7's shape
# (shipped library against lamejs_amd/lib/variants/*), then validate_stats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pass8; mkdir -p $O
cd $R
timeout 300 python -m pytest tests -m gpu -q -x --timeout 200 -k "seed or random or repair or goldens or long or chunk or async" > $O/pytest_gpu_subset.txt 2>&1; tail -3 $O/pytest_gpu_subset.txt
bash tools/r04_gpu_pass7.sh > /dev/null 2>&1; cp gpurun_out/r04_pass7/ab.txt $O/ab.txt; cat $O/ab.txt | cut -c1-200
timeout 200 python tests/tools/validate_stats.py 2>&1 | grep -v "amdgpu.ids" | tee $O/validate_stats.txt
