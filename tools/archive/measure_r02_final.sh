# DEV TOOL (GPU box): last pass of round 2 -- rocprofv3 kernel stats of both headline configs on the final code, then the GPU test tier.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 45 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/bench.py --cpu-seconds 0 --no-extras --check-frames 0 > $O/kt3.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt3 $O/kernel_stats_config3.csv
timeout 30 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --cpu-seconds 0 --no-extras --check-frames 0 --config 2 > $O/kt2.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt2 $O/kernel_stats_config2.csv
cd $R
timeout 75 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
