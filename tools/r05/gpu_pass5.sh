# DEV TOOL (GPU box), round 5 pass 5: the reservoir configurations and the call pattern with a pause in the helpers' polls.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_pass5; mkdir -p $O
cd $R
for c in reservoir reservoir256 reservoir512; do
  timeout 200 python bench.py --no-extras --cpu-seconds 0 --steps 4 --warmup 1 --check-frames 0 --config $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', 'ms_per_step', d['ms_per_step'], 'frames/s', d['value'], 'bit_exact_full', d['config']['bit_exact_full'])"
done 2>&1 | tee $O/configs.txt
for a in "2 128 fixture" "1 128 fixture" "2 128 sine 1000" "1 128 sine 1000"; do timeout 120 node tests/tools/bench_dropin.js calls $a 3 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('calls', '$a', 'frames/s', d['frames_per_s'], 'ms/call', d['ms_per_call'], 'median_us', d['call_us_median'], d['md5'][:8])"; done 2>&1 | tee $O/calls.txt
