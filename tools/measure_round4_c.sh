# DEV TOOL (GPU box): round 4, re-verification after the last kernel change (workgroup-aggregated counters in g_validate_fast): GPU tier, bench line, kernel stats of
# configs 3 and 2, a shorter randomised sweep.  Lands in gpurun_out/r04m2/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m2; mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -q --timeout 200 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 420 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04m2/bench_default.json').read().strip().splitlines()[0])
print('value', d['value'], 'ms', d['ms_per_step'], 'bit_exact_full', d['config']['bit_exact_full'])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','bit_exact_full','vs_device_resident','vs_c_abi_host_call','error')})
print(d['kernels_ms']); print(d.get('roofline'))
PY
{
echo "GPU fuzz after the last kernel change of round 4 (tests/tools/fuzz_gpu.py <n> <seed> <family>: random material, random chunking, GPU output vs the CPU oracle)"
for spec in "1000 650001 mpeg1" "500 650002 lsf" "200 650003 resample" "300 650005 mpeg1 joint" "200 650007 mpeg1 stereo whole"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 300 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz_gpu.txt
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 6 --check-frames 0 > $O/kt3.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt3 $O/kernel_stats_config3.csv; head -11 $O/kernel_stats_config3.csv | cut -c1-150
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 6 --check-frames 0 --config 2 > $O/kt2.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt2 $O/kernel_stats_config2.csv; head -11 $O/kernel_stats_config2.csv | cut -c1-150
