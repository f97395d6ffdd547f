# DEV TOOL (GPU box): round 6's final pass on the final code -- GPU tier, bench line, the RCCL path at world 1, the randomised sweep, kernel stats of configs 3 and 2.
# Lands in gpurun_out/r06m/; every step under its own timeout.  (tools/measure_round6_a.sh, the counter pass, runs first.)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -q --timeout 200 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06m/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'bit_exact_full', d['config']['bit_exact_full'])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','bit_exact_full','vs_device_resident','vs_c_abi_host_call','call_us_median','error')})
print(d['kernels_ms']); print(d.get('roofline')); print(json.dumps(d.get('cpu_baseline'))[:900])
PY
LAMEJS_BENCH_FORCE_DIST=1 timeout 200 python bench.py --config 3 --no-extras --cpu-seconds 0 --steps 2 > $O/bench_nccl_world1_config3.json 2> $O/bench_nccl_world1_config3.err; tail -c 200 $O/bench_nccl_world1_config3.err
{
echo "GPU fuzz on the final code of round 6 (tests/tools/fuzz_gpu.py <n> <seed> <family>: random material, random chunking, GPU output vs the CPU oracle)"
for spec in "3000 650001 mpeg1" "1600 650002 lsf" "800 650003 resample" "400 650004 lowrate" "1200 650005 mpeg1 joint" "510 650006 mpeg1 reservoir" "400 650007 mpeg1 stereo whole" "300 650008 lsf joint reservoir"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 400 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz_gpu_final_code.txt
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 6 --check-frames 0 > $O/kt3.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt3 $O/kernel_stats_config3.csv; head -12 $O/kernel_stats_config3.csv | cut -c1-150
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 6 --check-frames 0 --config 2 > $O/kt2.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt2 $O/kernel_stats_config2.csv; head -12 $O/kernel_stats_config2.csv | cut -c1-150
timeout 200 python $R/tests/tools/frame_prof.py 300 > $O/frame_prof.txt 2>&1; grep -E '^==|g_frame launch|hand-over' $O/frame_prof.txt
ls $O
timeout 150 python $R/tests/tools/handoff_prof.py > $O/handoff_prof.txt 2>&1; tail -4 $O/handoff_prof.txt | cut -c1-300
