# DEV TOOL (GPU box): the driver's round-end sequence (round 6) -- smoke(), the GPU tier, the default bench line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_check; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -c 200 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06_check/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'bit_exact_full', d['config']['bit_exact_full'], 'roofline', d['roofline']['frac'], d['roofline_compute']['frac_at_kernel_occupancy'], d['roofline_compute']['source'][:40])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','bit_exact_full','error')})
print(json.dumps(d['cpu_baseline']['reference_node'])[:300])
PY
