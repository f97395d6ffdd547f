# DEV TOOL (GPU box): the 1152-sample call pattern through Node, shipped library vs every library under lamejs_amd/lib/variants/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_ab_calls; mkdir -p $O
cd $R
calls() { for a in "2 128 fixture" "1 128 fixture" "2 128 sine 1000" "1 128 sine 1000"; do timeout 120 node tests/tools/bench_dropin.js calls $a 3 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$a', 'frames/s', d['frames_per_s'], 'ms/call', d['ms_per_call'], 'median_us', d['call_us_median'], 'min_us', d['call_us_min'], d['md5'][:8])"; done; }
{
for rep in 1 2; do
calls shipped
for v in lamejs_amd/lib/variants/*.so; do LAMEJS_HIP_LIB=$R/$v calls $(basename $v .so); done
done
} 2>&1 | tee $O/calls_ab.txt
