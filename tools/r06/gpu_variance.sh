# DEV TOOL (GPU box): run-to-run spread of the default bench line -- ten runs on one lease
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_variance; mkdir -p $O; cd $R
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 600 python bench.py --cpu-seconds 0 > $O/bench_$i.json 2> $O/bench_$i.err; echo "run $i exit $?"; done
python tools/variance_summary.py $O > $O/summary.json; head -c 1500 $O/summary.json
