# DEV TOOL (GPU box): the measurement pass behind profiles/r03_*.  Everything lands in gpurun_out/r03m/; copy what is to be kept.
#   bash tools/measure_round3.sh        (about 14 GPU-minutes)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
timeout 60 tools/_build/ubench_issue $O/ubench_issue.json > $O/ubench_issue.txt 2>&1
timeout 90 python tests/tools/call_latency.py > $O/call_latency.txt 2>&1
timeout 120 python tests/tools/phase_prof.py 20000 > $O/quant_phase_cycles.txt 2>&1
# the real RCCL path on one GPU: forced process group at world = 1, and once through torch.distributed.run
for c in 3 4 5 shard3; do
  LAMEJS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config $c --no-extras --cpu-seconds 0 --steps 2 > $O/bench_nccl_world1_config$c.json 2> $O/bench_nccl_world1_config$c.err
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --config 3 --no-extras --cpu-seconds 0 --steps 2 > $O/bench_torchrun_world1_config3.json 2> $O/bench_torchrun_world1_config3.err
cd /tmp && export TMPDIR=/tmp
B3="python $R/bench.py --cpu-seconds 0 --steps 1 --warmup 1 --check-frames 0 --no-extras"
B2="$B3 --config 2"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/bench.py --cpu-seconds 0 --no-extras > $O/kt3.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt3 $O/kernel_stats_config3.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --cpu-seconds 0 --no-extras --config 2 > $O/kt2.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt2 $O/kernel_stats_config2.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktr -- python $R/bench.py --cpu-seconds 0 --no-extras --config reservoir --steps 2 > $O/ktr.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/ktr $O/kernel_stats_reservoir.csv
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
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pl3 -- $B3 > $O/pl3.log 2>&1
python $R/tools/pmc_summary.py sq /tmp/pl3 $O/pmc_lanes_config3.json "SURVEY 8d config 3, 99999 frames, 1 stream, 1x MI355X"
rocprofv3 -L 2>/dev/null | grep -iE "LDS" | head -40 > $O/avail_lds_counters.txt
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pd3 -- $B3 > $O/pd3.log 2>&1
python $R/tools/pmc_summary.py sq /tmp/pd3 $O/pmc_lds_config3.json "SURVEY 8d config 3, 99999 frames, 1 stream, 1x MI355X" || tail -5 $O/pd3.log
cd $R
ls -la $O; head -c 1500 $O/bench_default.json; head -12 $O/kernel_stats_config3.csv
