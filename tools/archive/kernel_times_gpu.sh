cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03i
for c in 3 2; do timeout 120 python bench.py --config $c --no-extras --cpu-seconds 0 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config$c', {k:v['ms'] for k,v in d['kernels_ms'].items()}, d['ms_per_step'], d['config']['bit_exact_full'])" | tee -a gpurun_out/r03i/k.txt; done
