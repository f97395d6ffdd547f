#!/usr/bin/env python3
"""DEV TOOL: min / median / max over the bench lines of a directory (tools/r06/gpu_variance.sh): headline, every other configuration, every kernel of the
headline step, and the step's slack over the critical-path kernels (step - sum of the kernels on the launch stream: where a slow g_scan_ath would show)."""
import json, statistics, sys
from pathlib import Path
d = Path(sys.argv[1])
lines = []
for f in sorted(d.glob("bench_*.json")):
    try:
        lines.append(json.loads(f.read_text().strip().splitlines()[-1]))
    except Exception:
        pass
def mm(v):
    v = [x for x in v if x is not None]
    return None if not v else {"min": min(v), "median": statistics.median(v), "max": max(v), "spread_pct": round(100.0 * (max(v) - min(v)) / statistics.median(v), 2), "n": len(v)}
out = {"runs": len(lines), "headline_frames_per_s": mm([l["value"] for l in lines]), "headline_ms_per_step": mm([l["ms_per_step"] for l in lines]), "kernels_ms": {}, "other_configs": {}}
for k in lines[0].get("kernels_ms", {}):
    out["kernels_ms"][k] = mm([l["kernels_ms"].get(k, {}).get("ms") for l in lines])
crit = ("load", "psyA", "scan", "psyB", "polyphase", "mdct", "quant", "validate", "repair", "bits", "save")
out["step_minus_kernel_sum_ms"] = mm([round(l["ms_per_step"] - sum(l["kernels_ms"].get(k, {}).get("ms", 0.0) for k in crit if k != "scan"), 3) for l in lines])
out["note"] = "kernels_ms come from one extra step with HIP events around every kernel, everything on the launch stream (g_scan_ath included: 'scan'); in the timed steps g_scan_ath runs on a side stream beside polyphase + mdct"
for k in lines[0].get("other_configs", {}):
    vals = [l.get("other_configs", {}).get(k, {}) for l in lines]
    out["other_configs"][k] = mm([v.get("value", v.get("frames_per_s")) for v in vals])
print(json.dumps(out, indent=1))
