# DEV TOOL (GPU box): the measurement pass behind profiles/r02_*.  Everything lands in gpurun_out/r02m/; copy what is to be kept.
#   bash tools/measure_round.sh        (about 4 GPU-minutes)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O
cd $R
timeout 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 60 tools/_build/ubench_issue $O/ubench_issue.json > $O/ubench_issue.txt 2>&1
timeout 60 python tests/tools/call_latency.py > $O/call_latency.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B3="python $R/bench.py --cpu-seconds 0 --steps 1 --warmup 1 --check-frames 0 --no-extras"
B2="$B3 --config 2"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/bench.py --cpu-seconds 0 --no-extras > $O/kt3.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt3 $O/kernel_stats_config3.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --cpu-seconds 0 --no-extras --config 2 > $O/kt2.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt2 $O/kernel_stats_config2.csv
for c in 3 2; do
  eval B=\$B$c
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf$c -- $B > $O/pf$c.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw$c -- $B > $O/pw$c.log 2>&1
  python $R/tools/pmc_summary.py traffic /tmp/pf$c /tmp/pw$c $O/pmc_traffic_config$c.json "SURVEY 8d config $c, 99999 frames, 1 stream, 1x MI355X"
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/ps$c -- $B > $O/ps$c.log 2>&1
  python $R/tools/pmc_summary.py sq /tmp/ps$c $O/pmc_sq_config$c.json "SURVEY 8d config $c, 99999 frames, 1 stream, 1x MI355X"
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 --kernel-trace --output-format csv -d /tmp/pm$c -- $B > $O/pm$c.log 2>&1
  python $R/tools/pmc_summary.py sq /tmp/pm$c $O/pmc_mix_config$c.json "SURVEY 8d config $c, 99999 frames, 1 stream, 1x MI355X"
done
ls -la $O; head -c 1500 $O/bench_default.json; head -20 $O/kernel_stats_config3.csv
