# DEV TOOL (GPU box), round 4 pass 2: tail-help + validation digest as the shipped build -- GPU tier, bench line, kernel stats of configs 3 and 2.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pass2; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04_pass2/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'bit_exact_full', d['config']['bit_exact_full'])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','bit_exact_full','vs_device_resident','vs_c_abi_host_call','samples_s','constructor_ms','constructor_ms_each','error')})
print(d['kernels_ms'])
PY
cd /tmp && export TMPDIR=/tmp
for c in 3 2; do
  rocprofv3 --kernel-trace --stats -d $O/prof_c$c -o p -- python $R/bench.py --config $c --no-extras --cpu-seconds 0 --steps 5 --warmup 1 --check-frames 0 > /dev/null 2>&1
  f=$(find $O/prof_c$c -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_config$c.csv; head -12 $f | cut -c1-150
done
rm -rf $O/prof_c3 $O/prof_c2
