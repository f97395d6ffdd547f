# DEV TOOL (GPU box): round 6's counter pass -- issue microbenchmark + PMC traffic / instruction counts / mix of configs 3 and 2 -> gpurun_out/r06m/pmc_config{3,2}.json
# (copied to profiles/r05_pmc_config{3,2}.json, which bench.py's roofline lines read: run BEFORE the final bench pass, tools/measure_round6_b.sh).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O
cd $R
timeout 60 tools/_build/ubench_issue $O/ubench_issue.json > $O/ubench_issue.txt 2>&1; tail -3 $O/ubench_issue.txt
cd /tmp && export TMPDIR=/tmp
B3="python $R/bench.py --cpu-seconds 0 --steps 1 --warmup 1 --check-frames 0 --no-extras"
B2="$B3 --config 2"
for c in 3 2; do
  eval B=\$B$c
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf$c -- $B > $O/pf$c.log 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw$c -- $B > $O/pw$c.log 2>&1
  python $R/tools/pmc_summary.py traffic /tmp/pf$c /tmp/pw$c $O/pmc_traffic_config$c.json "SURVEY 8d config $c, 99999 frames, 1 stream, 1x MI355X"
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/ps$c -- $B > $O/ps$c.log 2>&1
  python $R/tools/pmc_summary.py sq /tmp/ps$c $O/pmc_sq_config$c.json "SURVEY 8d config $c, 99999 frames, 1 stream, 1x MI355X"
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 --kernel-trace --output-format csv -d /tmp/pm$c -- $B > $O/pm$c.log 2>&1
  python $R/tools/pmc_summary.py sq /tmp/pm$c $O/pmc_mix_config$c.json "SURVEY 8d config $c, 99999 frames, 1 stream, 1x MI355X"
  python $R/tools/make_profile_json.py $O $c $O/pmc_config$c.json
done
ls $O
# ---- the one-frame launch under the counters: 300 calls of 1152 samples, sine, two channels and one -- per-launch averages of g_frame
cd /tmp
for ch in 2 1; do
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pfr$ch -- python $R/tools/r06/one_frame_calls.py $ch 300 > $O/pfr$ch.log 2>&1
  python $R/tools/pmc_summary.py sq /tmp/pfr$ch $O/pmc_sq_one_frame_calls_ch$ch.json "300 one-frame calls (1152 samples per lhip_encode), sine, $ch channel(s), 128 kbps: the single launch g_frame per call"
done
python - <<PY
import json, os
O = "$O"
out = {"head": os.environ.get("LAMEJS_SOURCE_HEAD"), "method": "rocprofv3 --pmc SQ_* --kernel-trace over tools/r06/one_frame_calls.py: averages per g_frame launch (one workgroup of eight waves per call)"}
for ch in (2, 1):
    d = json.load(open(f"{O}/pmc_sq_one_frame_calls_ch{ch}.json"))
    k = {n: v for n, v in d["kernels"].items() if n.startswith("g_frame")}
    for n, v in k.items():
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        v["issue_fraction_of_wave_cycles"] = round(v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4); v["wait_any_fraction"] = round(v.get("SQ_WAIT_ANY", 0) / wc, 4); v["wait_for_issue_fraction"] = round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 4)
        v["salu_per_valu"] = round(v.get("SQ_INSTS_SALU", 0) / max(v.get("SQ_INSTS_VALU", 1), 1), 3)
    out[f"channels_{ch}"] = {"workload": d["workload"], "kernels": k, "launches": d.get("launches")}
json.dump(out, open(f"{O}/pmc_one_frame_calls.json", "w"), indent=1)
print(json.dumps(out)[:1200])
PY
