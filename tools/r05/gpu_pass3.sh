# DEV TOOL (GPU box), round 5 pass 3: the count helper with short polls -- phases (profiling build) and the call pattern, shipped vs variants.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_pass3; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python tests/tools/frame_prof.py 300 > $O/frame_prof.txt 2>&1; grep -E "^==|host side|g_frame|quant |search|quantize|count|noise|balance|helper|total" $O/frame_prof.txt
calls() { for a in "2 128 fixture" "1 128 fixture" "2 128 sine 1000" "1 128 sine 1000"; do timeout 120 node tests/tools/bench_dropin.js calls $a 3 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$a', 'frames/s', d['frames_per_s'], 'ms/call', d['ms_per_call'], 'median_us', d['call_us_median'], 'min_us', d['call_us_min'], d['md5'][:8])"; done; }
{
calls shipped
for v in lamejs_amd/lib/variants/*.so; do LAMEJS_HIP_LIB=$R/$v calls $(basename $v .so); done
} 2>&1 | tee $O/calls_ab.txt
