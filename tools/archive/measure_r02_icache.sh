# DEV TOOL (GPU box): instruction-cache and scalar-data-cache counters of the stereo config (is the 30 % "waiting for an instruction"
# of g_quant an instruction-fetch problem?  the kernel is 91 KB of code, the cache 64 KB per two CUs).  Lands in gpurun_out/r02i/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-seconds 0 --steps 1 --warmup 1 --check-frames 0 --no-extras"
timeout 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pi1 -- $B > $O/pi1.log 2>&1
timeout 120 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pi2 -- $B > $O/pi2.log 2>&1
for p in pi1 pi2; do python $R/tools/pmc_summary.py sq /tmp/$p $O/pmc_$p.json "BASELINE configs[2]: stereo 44.1kHz 128kbps, 99999 frames, 1 stream, 1x MI355X" >> $O/pmc.log 2>&1; done
tail -3 $O/pi1.log $O/pi2.log $O/pmc.log
