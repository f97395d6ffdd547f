# DEV TOOL (GPU box): round-2 first measurement pass: GPU tests, VALU issue microbench, instruction-class PMC of the stereo config,
# per-call breakdown, bench of the default config.  Everything lands in gpurun_out/r02a/.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 tools/_build/ubench_issue $O/ubench_issue.json > $O/ubench_issue.txt 2>&1
timeout 300 python tests/tools/call_breakdown.py > $O/call_breakdown.txt 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-seconds 0 --steps 1 --warmup 1 --check-frames 0 --no-extras"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d /tmp/p1 -- $B > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/p2 -- $B > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/p3 -- $B > $O/p3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/p4 -- $B > $O/p4.log 2>&1
for p in p1 p2 p3 p4; do python $R/tools/pmc_summary.py sq /tmp/$p $O/pmc_$p.json "BASELINE configs[2]: stereo 44.1kHz 128kbps, 99999 frames, 1 stream, 1x MI355X" >> $O/pmc.log 2>&1; done
ls -la $O; tail -3 $O/pytest.log; cat $O/ubench_issue.txt; cat $O/call_breakdown.txt; head -c 3000 $O/bench_default.json
