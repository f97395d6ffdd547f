#!/usr/bin/env python3
"""DEV TOOL: merge the rocprofv3 summaries of tools/measure_round.sh into the per-config profile bench.py reads
(profiles/r02_pmc_config<N>.json): HBM traffic per kernel launch, instruction counts and class mix of every kernel, and -- for
the compute roofline -- the issue ceiling of the quantization kernel's instruction mix measured by tools/ubench_issue.hip.
usage: make_profile_json.py <gpurun_out/r02m> <config number> <out.json>"""
import json, sys
d, c, out = sys.argv[1], sys.argv[2], sys.argv[3]
tr = json.load(open(f"{d}/pmc_traffic_config{c}.json"))
sq = json.load(open(f"{d}/pmc_sq_config{c}.json"))
mx = json.load(open(f"{d}/pmc_mix_config{c}.json"))
ub = json.load(open(f"{d}/ubench_issue.json"))
qm = ub["cases"]["quant_mix"]
res = {"workload": tr["workload"], "method": {"traffic": tr["method"], "counters": "rocprofv3 --pmc SQ_* (two passes), per launch averages",
                                               "issue_ceiling": "tools/ubench_issue.hip `quant_mix`: an independent-instruction stream with the quantization kernel's "
                                                                "measured mix (16 VALU = 5 int32 + 2 f64 + 1 cvt + 8 move/compare/select, 10 SALU, 1 LDS read), "
                                                                "G VALU wave-instructions/s over the whole chip at 1, 2, 4, 5, 6, 8 waves per SIMD (wall clock)"},
       "traffic": {k: v for k, v in tr["kernels"].items() if k.startswith("g_")},
       "counters": {k: dict(sq["kernels"][k], **mx["kernels"].get(k, {})) for k in sq["kernels"] if k.startswith("g_")},
       "compute": {}}
res["traffic_total_bytes_per_step"] = sum(v["hbm_bytes_per_launch"] for v in res["traffic"].values())
q = res["counters"]["g_quant"]
res["compute"]["g_quant"] = {"valu_insts_per_launch": q["SQ_INSTS_VALU"], "salu_insts_per_launch": q["SQ_INSTS_SALU"], "lds_insts_per_launch": q["SQ_INSTS_LDS"],
                             "issue_ceiling_ginst_per_s": {"waves_per_simd": ub["waves_per_simd"], "valu_ginst_per_s": qm["ginst_per_s"]},
                             "occupancy_waves_per_simd": 4}
import os
res["head"] = os.environ.get("LAMEJS_SOURCE_HEAD")       # the commit whose library was measured (bench.py: roofline_compute.source_head)
json.dump(res, open(out, "w"), indent=1)
print(out, "total traffic GB", res["traffic_total_bytes_per_step"] / 1e9)
