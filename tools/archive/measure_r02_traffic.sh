# DEV TOOL (GPU box): HBM traffic of the stereo config after the polyphase / bit-packing staging changes (two PMC passes).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-seconds 0 --steps 1 --warmup 1 --check-frames 0 --no-extras"
timeout 18 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf3 -- $B > $O/pf3.log 2>&1
timeout 18 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw3 -- $B > $O/pw3.log 2>&1
python $R/tools/pmc_summary.py traffic /tmp/pf3 /tmp/pw3 $O/pmc_traffic_config3.json "SURVEY 8d config 3, 99999 frames, 1 stream, 1x MI355X" > $O/sum.log 2>&1
tail -2 $O/sum.log
