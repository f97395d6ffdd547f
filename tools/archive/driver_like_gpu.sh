# DEV TOOL (GPU box): what the driver runs at round end, in its order: the GPU test tier (-x), smoke(), the default bench line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-drv}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3 | tee $O/bench_time.txt
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['config']['bit_exact_full'], d['roofline']['frac'], d['roofline_compute']['frac_at_kernel_occupancy'], d['cpu_baseline']['value'], d['cpu_baseline']['aggregate']['value'])"
