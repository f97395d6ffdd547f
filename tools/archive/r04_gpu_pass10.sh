# DEV TOOL (GPU box), round 4 pass 10 (the round's last 2.8 GPU-minutes): A/B of the two experiment patches (tools/experiments/, variant libraries built by try_patch.sh)
# against the shipped library on configs 3 and 2, two interleaved repetitions, every run under a short timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pass10; mkdir -p $O
cd $R
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$1', 'step_ms', d['ms_per_step'], 'quant_ms', d['kernels_ms']['quant']['ms'], 'validate_ms', d['kernels_ms']['validate']['ms'], 'bit_exact_full', c['bit_exact_full'])"; }
B="python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 1 --check-frames 0"
{
for rep in 1 2; do
  for lib in lamejs_amd/lib/liblamejs_hip.so lamejs_amd/lib/variants/*.so; do
    for c in 3 2; do
      LAMEJS_HIP_LIB=$R/$lib timeout 40 $B --config $c 2>/dev/null | line "$lib config$c rep$rep"
    done
  done
done
} | tee $O/ab.txt
