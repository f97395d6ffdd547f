# DEV TOOL (GPU box): the bit-reservoir configurations (bench lines, md5-checked) and a short randomised sweep of the reservoir paths after a change to g_resv_stream / g_frame<1>.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_resv_check; mkdir -p $O
cd $R
{
for spec in "120 880001 mpeg1 reservoir" "60 880002 lsf reservoir" "80 880003 mpeg1 joint reservoir" "80 880004 mpeg1 reservoir framecalls" "40 880005 lsf joint reservoir framecalls"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 300 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
for c in reservoir reservoir256 reservoir512; do
  timeout 200 python bench.py --no-extras --cpu-seconds 0 --steps 4 --warmup 1 --check-frames 0 --config $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', 'ms_per_step', d['ms_per_step'], 'frames/s', d['value'], 'bit_exact_full', d['config']['bit_exact_full'])"
done
} 2>&1 | tee $O/resv.txt
