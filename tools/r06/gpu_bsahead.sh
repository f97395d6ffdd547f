# DEV TOOL (GPU box): the bin search's look-ahead on the count helper: sweeps of the device code, then the call pattern and the reservoir configurations shipped vs -DLHIP_BS_AHEAD=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_bsahead; mkdir -p $O; cd $R
{
for spec in "300 680001 mpeg1 framecalls" "150 680002 lsf framecalls" "100 680003 mpeg1 joint framecalls" "150 680004 mpeg1 reservoir framecalls" "150 680005 mpeg1 reservoir" "80 680006 lsf joint reservoir" "60 680007 lowrate framecalls" "60 680008 resample framecalls"; do
  echo "  python tests/tools/fuzz_gpu.py $spec    -> $(timeout 600 python tests/tools/fuzz_gpu.py $spec 2>&1 | tail -1)"
done
} | tee $O/fuzz.txt
bash tools/r05/gpu_ab_calls.sh 2>&1 | tee $O/calls_ab.txt
B="python bench.py --no-extras --cpu-seconds 0 --steps 3 --warmup 1 --check-frames 0"
{
for rep in 1 2; do for lib in lamejs_amd/lib/liblamejs_hip.so lamejs_amd/lib/variants/nobsahead.so; do for c in reservoir reservoir256; do
  LAMEJS_HIP_LIB=$R/$lib timeout 200 $B --config $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$c', 'frames/s', d['value'], 'ms', d['ms_per_step'], d['config']['bit_exact_full'])"
done; done; done
} | tee $O/resv_ab.txt
