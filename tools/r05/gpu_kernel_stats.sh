# DEV TOOL (GPU box): rocprofv3 kernel-trace statistics of bench.py, configs 3 and 2 (the CSVs kept under profiles/).  usage: gpu_kernel_stats.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r05}; O=$R/gpurun_out/${T}_kernel_stats; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in 3 2; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$c -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 6 --warmup 1 --check-frames 0 --config $c > $O/kt$c.log 2>&1
  f=$(find /tmp/kt$c -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats_config$c.csv; tail -1 $O/kt$c.log | cut -c1-200; head -14 $O/kernel_stats_config$c.csv
done
