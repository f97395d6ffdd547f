# DEV TOOL (GPU box), round 5 pass 1: baseline of the reference's documented call pattern (1152 samples per encodeBuffer call) before any
# change -- where a call spends its time (profiling build), the kernel's own duration (rocprofv3), the new bench lines, the live-reference test.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_pass1; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python tests/tools/frame_prof.py 300 > $O/frame_prof.txt 2>&1; cat $O/frame_prof.txt
for ch in 2 1; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_calls_ch$ch -o calls -- node $R/tests/tools/bench_dropin.js calls $ch 128 sine 600 1 > $O/calls_ch$ch.json 2> $O/calls_ch$ch.err )
  cut -c1-700 $O/calls_ch$ch.json
  f=$(find $O/prof_calls_ch$ch -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_calls_ch$ch.csv && head -8 $f
done
LAMEJS_HIP_NO_FRAME_KERNEL=1 timeout 120 node tests/tools/bench_dropin.js calls 2 128 sine 600 1 | cut -c1-600 > $O/calls_ch2_separate_kernels.json; cat $O/calls_ch2_separate_kernels.json
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "live_reference or fixture_md5s" > $O/pytest_live_reference.txt 2>&1; tail -5 $O/pytest_live_reference.txt
timeout 900 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05_pass1/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'bit_exact_full', d['config']['bit_exact_full'])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','call_us_median','call_us_min','bit_exact_full','samples_s','error')})
print(json.dumps(d['cpu_baseline'])[:1500])
print(d['kernels_ms'])
PY
