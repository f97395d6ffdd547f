# DEV TOOL (GPU box), round 4 pass 5: scratch-free-ish g_quant (lane_anew / const_here), two batches in flight with the quantization kernels ordered,
# the 96-register probe (what 5 waves per SIMD would leave g_quant; run at 4), HBM traffic of the new build.  Every step under its own timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pass5; mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -q -x --timeout 200 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$1', 'step_ms', d['ms_per_step'], 'other_mode_ms', c.get('ms_per_step_two_batches_in_flight', c.get('ms_per_step_one_batch_in_flight')), 'in_flight', c['batches_in_flight'], 'quant_ms', d['kernels_ms']['quant']['ms'], 'validate_ms', d['kernels_ms']['validate']['ms'], 'bit_exact_full', c['bit_exact_full'])"; }
B="python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 1 --check-frames 0"
{
for rep in 1 2; do
  for lib in lamejs_amd/lib/liblamejs_hip.so lamejs_amd/lib/variants/*.so; do
    for c in 3 2; do
      LAMEJS_HIP_LIB=$R/$lib timeout 120 $B --config $c 2>/dev/null | line "$lib config$c rep$rep"
    done
  done
done
for c in 3 2 bursts; do
  timeout 120 $B --config $c --pipeline 2>/dev/null | line "shipped config$c --pipeline (quant kernels ordered)"
  LAMEJS_HIP_PIPE_QUANT_ORDER=0 timeout 120 $B --config $c --pipeline 2>/dev/null | line "shipped config$c --pipeline LAMEJS_HIP_PIPE_QUANT_ORDER=0 (side by side)"
done
} | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
B3="python $R/bench.py --cpu-seconds 0 --steps 1 --warmup 1 --check-frames 0 --no-extras"
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf3 -- $B3 > $O/pf3.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw3 -- $B3 > $O/pw3.log 2>&1
python $R/tools/pmc_summary.py traffic /tmp/pf3 /tmp/pw3 $O/pmc_traffic_config3.json "SURVEY 8d config 3, 99999 frames, 1 stream, 1x MI355X" && python -c "
import json; d=json.load(open('$O/pmc_traffic_config3.json'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['hbm_bytes_per_launch'])[:8]: print(k, round(v['hbm_bytes_per_launch']/1e9,3), 'GB')"
