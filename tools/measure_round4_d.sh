# DEV TOOL (GPU box): round 4, re-verification after g_fixup went to two workgroups per CU: GPU tier, bench line, a short randomised sweep on the repair-heavy families.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m3; mkdir -p $O
cd $R
timeout 300 python -m pytest tests -m gpu -q --timeout 200 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04m3/bench_default.json').read().strip().splitlines()[0])
print('value', d['value'], 'ms', d['ms_per_step'], 'bit_exact_full', d['config']['bit_exact_full'])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','bit_exact_full','error')})
print(d['kernels_ms'])
PY
{
echo "GPU fuzz after g_fixup went to two workgroups per CU (tests/tools/fuzz_gpu.py <n> <seed> <family>)"
for spec in "600 660001 mpeg1" "300 660002 lsf" "200 660005 mpeg1 joint" "200 660007 mpeg1 stereo whole"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 120 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz_gpu.txt
