# DEV TOOL (GPU box): round 6, first pass -- the new GPU tests (interleaved live encoders, two aliased contexts, Node interleaving), then the whole tier and the bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_pass1; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -x -q -m gpu -k "interleaved or two_devices" > $O/pytest_new.txt 2>&1; tail -5 $O/pytest_new.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -c 300 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06_pass1/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], d['config']['devices_per_rank'], d['config']['distinct_devices'])
cb=d['cpu_baseline']; print('cpu_baseline', cb['value'], cb['kind'], 'port', cb.get('port',{}).get('value'), 'shapes', {k:v.get('value') for k,v in cb.get('shapes',{}).items()})
for k,v in d.get('other_configs',{}).items():
    if '1152' in k: print(k, {a:b for a,b in v.items() if a in ('frames_per_s','ms_per_call','bit_exact_full','speedup_vs_reference_same_pattern','reference_same_pattern_frames_per_s','error')})
PY
