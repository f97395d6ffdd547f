# DEV TOOL (GPU box), round 5 pass 2: the one-frame call after the host small-call path, the overlapped stages and the count helper.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_pass2; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python tests/tools/frame_prof.py 300 > $O/frame_prof.txt 2>&1; head -60 $O/frame_prof.txt
calls() { for a in "2 128 fixture" "1 128 fixture" "2 128 sine 1000" "1 128 sine 1000"; do timeout 120 node tests/tools/bench_dropin.js calls $a 3 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$a', 'frames/s', d['frames_per_s'], 'ms/call', d['ms_per_call'], 'median_us', d['call_us_median'], 'min_us', d['call_us_min'], d['md5'][:8])"; done; }
{
calls shipped
LAMEJS_HIP_LIB=$R/lamejs_amd/lib/variants/nopipe.so calls no_count_helper
LAMEJS_HIP_NO_SMALL_CALLS=1 calls no_small_call_path
} 2>&1 | tee $O/calls_ab.txt
timeout 900 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
timeout 900 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05_pass2/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'bit_exact_full', d['config']['bit_exact_full'])
for k,v in d.get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','frames_per_s','ms_per_step','ms_per_call','call_us_median','call_us_min','bit_exact_full','error')})
print(d['kernels_ms'])
PY
