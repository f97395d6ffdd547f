cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b20
timeout 400 python bench.py --steps 20 --warmup 2 > gpurun_out/r03b20/bench_steps20.json 2> gpurun_out/r03b20/err.txt
python -c "
import json; d=json.load(open('gpurun_out/r03b20/bench_steps20.json')); print(d['value'], d['ms_per_step'], d['steps'], d['config']['bit_exact_full'], d['roofline_compute']['frac_at_kernel_occupancy'])"
