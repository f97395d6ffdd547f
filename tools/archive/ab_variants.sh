# DEV TOOL (GPU box): A/B of library variants (lamejs_amd/lib/variants/*.so against the shipped library): g_quant / validation time, step time,
# repaired frames and bit-exactness of four workloads, two repetitions each, interleaved.  Lands in gpurun_out/$1.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ab}; mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in lamejs_amd/lib/liblamejs_hip.so lamejs_amd/lib/variants/*.so; do
    for c in 3 2 bursts 5; do
      LAMEJS_HIP_LIB=$R/$lib timeout 120 python bench.py --config $c --no-extras --cpu-seconds 0 --steps 4 --warmup 1 --check-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'config$c', 'rep$rep', 'quant_ms', d['kernels_ms']['quant']['ms'], 'validate_ms', d['kernels_ms']['validate']['ms'], 'step_ms', d['ms_per_step'], 'repaired', d['config']['seed_repaired_frames'], d['config']['repair_iterations'], 'bit_exact_full', d['config']['bit_exact_full'])" | tee -a $O/ab.txt
    done
  done
done
