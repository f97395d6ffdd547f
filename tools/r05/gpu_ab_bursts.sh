# DEV TOOL (GPU box): the `bursts` configuration (validation + repair on the device), shipped library vs lamejs_amd/lib/variants/*.so, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_ab_bursts; mkdir -p $O
cd $R
run() { timeout 200 python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 2 --check-frames 0 --config $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'config $2', 'ms_per_step', d['ms_per_step'], 'frames/s', d['value'], 'bit_exact_full', d['config']['bit_exact_full'], {k: round(v, 3) for k, v in (d.get('kernel_ms') or {}).items()} if isinstance(d.get('kernel_ms'), dict) else '')"; }
{
for rep in 1 2; do
  for c in bursts; do
    unset LAMEJS_HIP_LIB; run shipped $c
    for v in lamejs_amd/lib/variants/*.so; do export LAMEJS_HIP_LIB=$R/$v; run $(basename $v .so) $c; done
  done
done
} 2>&1 | tee $O/ab.txt
