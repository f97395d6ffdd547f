# DEV TOOL (GPU box), round 4 pass 1: GPU tier on the new host code, the bench line (drop-in lines through Node included), A/B shipped vs tail-help.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pass1; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04_pass1/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'bit_exact_full', d['config']['bit_exact_full'])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','bit_exact_full','vs_device_resident','vs_c_abi_host_call','samples_s','constructor_ms','constructor_ms_each','error','result_is_exact_arraybuffer')})
print(d['kernels_ms'])
PY
bash tools/ab_variants.sh r04_pass1_ab > /dev/null 2>&1; cat $R/gpurun_out/r04_pass1_ab/ab.txt
