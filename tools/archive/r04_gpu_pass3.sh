# DEV TOOL (GPU box), round 4 pass 3: pipeline mode, validation quads, grouped multi-stream host path, reservoir 256 / 512 streams.
# Every step under its own timeout (pass 2 lost 25 GPU-minutes to an unbounded rocprofv3 run).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pass3; mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -q -x --durations=5 --timeout 200 > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
timeout 420 python bench.py --steps 6 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04_pass3/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'single', d['config'].get('ms_per_step_one_batch_in_flight'), 'bit_exact_full', d['config']['bit_exact_full'])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','bit_exact_full','vs_device_resident','vs_c_abi_host_call','samples_s','error')})
print(d['kernels_ms'])
PY
for c in 3 2; do
  timeout 100 python bench.py --config $c --no-extras --cpu-seconds 0 --steps 6 --warmup 1 --check-frames 0 --no-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config$c no-pipeline: step_ms', d['ms_per_step'], 'kernels', {k:v['ms'] for k,v in d['kernels_ms'].items()})"
done
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 4 --check-frames 0 > $O/kt3.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt3 $O/kernel_stats_config3.csv; head -14 $O/kernel_stats_config3.csv | cut -c1-160
