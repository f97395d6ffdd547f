# DEV TOOL (GPU box), round 5 pass 6: the call pattern with psyB beside the quantization; the launch's own duration (rocprofv3 kernel trace) next to the call's.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_pass6; mkdir -p $O
cd $R
export TMPDIR=/tmp
for a in "2 128 fixture" "1 128 fixture" "2 128 sine 1000" "1 128 sine 1000"; do timeout 120 node tests/tools/bench_dropin.js calls $a 3 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('calls', '$a', 'frames/s', d['frames_per_s'], 'ms/call', d['ms_per_call'], 'median_us', d['call_us_median'], d['md5'][:8])"; done 2>&1 | tee $O/calls.txt
for ch in 2 1; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_calls_ch$ch -o calls -- node $R/tests/tools/bench_dropin.js calls $ch 128 sine 600 1 > $O/calls_ch$ch.json 2> $O/calls_ch$ch.err )
  for f in $(find $O/prof_calls_ch$ch -name '*stats.csv'); do echo "== $f"; head -6 $f; done
done 2>&1 | tee $O/rocprof.txt
