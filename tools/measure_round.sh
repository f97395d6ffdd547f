set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
python bench.py > $O/bench_mono.json 2> $O/bench_mono.err
python bench.py --channels 2 --cpu-frames 20000 > $O/bench_st128.json 2>> $O/bench_mono.err
python bench.py --channels 2 --kbps 320 --cpu-frames 0 > $O/bench_st320.json 2>> $O/bench_mono.err
python bench.py --streams 128 --frames 1000 --cpu-frames 0 > $O/bench_128streams.json 2>> $O/bench_mono.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --cpu-frames 0 > $O/kt.log 2>&1
python $R/tools/pmc_summary.py stats /tmp/kt $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python $R/bench.py --cpu-frames 0 --steps 1 --warmup 1 --check-frames 0 > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python $R/bench.py --cpu-frames 0 --steps 1 --warmup 1 --check-frames 0 > $O/pw.log 2>&1
python $R/tools/pmc_summary.py traffic /tmp/pf /tmp/pw $O/pmc_traffic.json "BASELINE configs[1]: mono 44.1kHz 128kbps, 99999 frames, 1 stream, 1x MI355X"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ps -- python $R/bench.py --cpu-frames 0 --steps 1 --warmup 1 --check-frames 0 > $O/ps.log 2>&1
python $R/tools/pmc_summary.py sq /tmp/ps $O/pmc_sq.json "BASELINE configs[1]: mono 44.1kHz 128kbps, 99999 frames, 1 stream, 1x MI355X"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ps2 -- python $R/bench.py --channels 2 --cpu-frames 0 --steps 1 --warmup 1 --check-frames 0 > $O/ps2.log 2>&1
python $R/tools/pmc_summary.py sq /tmp/ps2 $O/pmc_sq_st128.json "BASELINE configs[2]: stereo 44.1kHz 128kbps, 99999 frames, 1 stream, 1x MI355X"
ls -la $O; head -5 $O/kernel_stats.csv
