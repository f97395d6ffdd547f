# DEV TOOL (GPU box): first GPU pass of round 3 -- GPU tests on the new reservoir path, bench line, the RCCL path at world = 1,
# VALU lane utilisation of the kernels.  Everything lands in gpurun_out/r03a/.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 400 python bench.py --steps 5 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
# the real RCCL path on one GPU: forced process group at world = 1, and once through torch.distributed.run
for c in 3 4 5 shard3; do
  LAMEJS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config $c --no-extras --cpu-seconds 0 --steps 2 > $O/bench_nccl_world1_config$c.json 2> $O/bench_nccl_world1_config$c.err
  tail -c 300 $O/bench_nccl_world1_config$c.err
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --config 3 --no-extras --cpu-seconds 0 --steps 2 > $O/bench_torchrun_world1_config3.json 2> $O/bench_torchrun_world1_config3.err
tail -c 300 $O/bench_torchrun_world1_config3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -iE "VALU|THREAD_CYCLES|Utilization" | head -60 > $O/avail_valu_counters.txt
B3="python $R/bench.py --cpu-seconds 0 --steps 1 --warmup 1 --check-frames 0 --no-extras"
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pl3 -- $B3 > $O/pl3.log 2>&1
python $R/tools/pmc_summary.py sq /tmp/pl3 $O/pmc_lanes_config3.json "SURVEY 8d config 3, 99999 frames, 1 stream, 1x MI355X" || tail -5 $O/pl3.log
timeout 200 rocprofv3 --pmc VALUUtilization VALUBusy SALUBusy --kernel-trace --output-format csv -d /tmp/pu3 -- $B3 > $O/pu3.log 2>&1
python $R/tools/pmc_summary.py sq /tmp/pu3 $O/pmc_util_config3.json "SURVEY 8d config 3, 99999 frames, 1 stream, 1x MI355X" || tail -5 $O/pu3.log
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktr -- python $R/bench.py --cpu-seconds 0 --no-extras --config reservoir --steps 2 > $O/ktr.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/ktr $O/kernel_stats_reservoir.csv
cd $R
timeout 120 python tests/tools/phase_prof.py 20000 > $O/quant_phase_cycles.txt 2>&1
ls -la $O; head -c 2500 $O/bench_default.json
