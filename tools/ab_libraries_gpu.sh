# DEV TOOL (GPU box): A/B of library variants under lamejs_amd/lib/variants/ against the shipped library (step, quantization and validation times,
# bit-exactness) on configs 3, 2 and bursts, two interleaved repetitions.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O
cd $R
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
k=d['kernels_ms']
print('$1', 'step_ms', d['ms_per_step'], 'quant_ms', k['quant']['ms'], 'psyA_ms', k['psyA']['ms'], 'psyB_ms', k['psyB']['ms'], 'poly+mdct_ms', round(k['polyphase']['ms'] + k['mdct']['ms'], 4), 'validate_ms', k['validate']['ms'], 'bits_ms', k['bits']['ms'], 'repaired', c['seed_repaired_frames'], 'bit_exact_full', c['bit_exact_full'])"; }
B="python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 1 --check-frames 0"
{
for rep in 1 2; do
  for lib in lamejs_amd/lib/liblamejs_hip.so lamejs_amd/lib/variants/*.so; do
    for c in 3 2 bursts; do
      LAMEJS_HIP_LIB=$R/$lib timeout 120 $B --config $c 2>/dev/null | line "$lib config$c rep$rep"
    done
  done
done
} | tee $O/ab.txt
