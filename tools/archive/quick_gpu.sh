# DEV TOOL (GPU box): a short check of the current build -- headline workloads (g_quant time, bit-exactness), phase cycles, GPU tests.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-quick}; mkdir -p $O
cd $R
for c in 3 2 bursts; do
  timeout 120 python bench.py --config $c --no-extras --cpu-seconds 0 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config$c', 'quant_ms', d['kernels_ms']['quant']['ms'], 'psyA', d['kernels_ms']['psyA']['ms'], 'psyB', d['kernels_ms']['psyB']['ms'], 'step_ms', d['ms_per_step'], 'value', d['value'], 'bit_exact_full', d['config']['bit_exact_full'], 'prefix', d['config']['bit_exact_prefix_vs_oracle'], 'repaired', d['config']['seed_repaired_frames'])" | tee -a $O/quick.txt
done
timeout 120 python tests/tools/phase_prof.py 20000 > $O/quant_phase_cycles.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
