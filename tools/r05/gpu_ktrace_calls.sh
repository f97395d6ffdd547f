# DEV TOOL (GPU box): kernel durations (rocprofv3) of the 1152-sample call pattern, mono + stereo `sine`, shipped library vs lamejs_amd/lib/variants/*.so
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_ktrace_calls; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in shipped $R/lamejs_amd/lib/variants/*.so; do
  for a in "1 128 sine 600" "2 128 sine 300"; do
    n=$(basename $lib .so)_$(echo $a | tr ' ' '_')
    if [ $lib = shipped ]; then unset LAMEJS_HIP_LIB; else export LAMEJS_HIP_LIB=$lib; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- node $R/tests/tools/bench_dropin.js calls $a 1 > $O/$n.log 2>&1
    f=$(ls $O/$n/*/*kernel_stats.csv 2>/dev/null | head -1)
    echo "== $n"; [ -n "$f" ] && head -4 $f | cut -d, -f1-8
  done
done 2>&1 | tee $O/summary.txt
