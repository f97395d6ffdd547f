# DEV TOOL (GPU box), round 4 pass 4 (first pass of the re-created container): GPU tier, bench line, kernel stats (pipelined default + one batch in flight),
# phase cycles of g_quant / g_psyA.  Every step under its own timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pass4; mkdir -p $O
cd $R
timeout 420 python -m pytest tests -m gpu -q -x --durations=6 --timeout 200 > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
timeout 420 python bench.py --steps 6 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04_pass4/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'single', d['config'].get('ms_per_step_one_batch_in_flight'), 'bit_exact_full', d['config']['bit_exact_full'])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','bit_exact_full','vs_device_resident','vs_c_abi_host_call','samples_s','error','constructor_ms')})
print(d['kernels_ms']); print(d.get('roofline')); print(d.get('cpu_baseline'))
PY
timeout 60 python tests/tools/phase_prof.py 20000 > $O/quant_phase_cycles.txt 2>&1; grep -v "^   [cnq]_" $O/quant_phase_cycles.txt | head -60
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 4 --check-frames 0 > $O/kt3.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt3 $O/kernel_stats_config3_pipelined.csv; head -16 $O/kernel_stats_config3_pipelined.csv | cut -c1-160
for c in 3 2; do
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt${c}n -- python $R/bench.py --config $c --cpu-seconds 0 --no-extras --steps 4 --check-frames 0 --no-pipeline > $O/kt${c}n.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt${c}n $O/kernel_stats_config${c}_one_in_flight.csv; head -16 $O/kernel_stats_config${c}_one_in_flight.csv | cut -c1-160
done
