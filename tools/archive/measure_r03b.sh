# DEV TOOL (GPU box): quick validation pass -- GPU tests, bench line, kernel stats, phase cycles.  Lands in gpurun_out/$1 (default r03b).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03b}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
timeout 120 python tests/tools/state_diff.py > $O/state_diff.txt 2>&1; head -12 $O/state_diff.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
timeout 120 python tests/tools/phase_prof.py 20000 > $O/quant_phase_cycles.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/bench.py --cpu-seconds 0 --no-extras > $O/kt3.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt3 $O/kernel_stats_config3.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --cpu-seconds 0 --no-extras --config 2 > $O/kt2.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt2 $O/kernel_stats_config2.csv
cd $R
head -c 1200 $O/bench_default.json; cat $O/kernel_stats_config3.csv | head -20
