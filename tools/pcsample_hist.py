#!/usr/bin/env python3
"""DEV TOOL (GPU box): aggregate a rocprofv3 PC-sampling CSV into a per-instruction histogram.

usage: pcsample_hist.py <rocprof output dir> <out.txt>
Writes: header + first rows (format probe), then counts per (kernel/dispatch, instruction) sorted by count.
"""
import csv, sys, os, collections, glob
d, out = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)]
with open(out, "w") as o:
    o.write("files: %s\n" % [(f, os.path.getsize(f)) for f in files])
    for f in files:
        if "pc_sampling" not in os.path.basename(f):
            continue
        with open(f, newline="") as fh:
            rd = csv.reader(fh)
            hdr = next(rd)
            o.write("== %s\nheader: %s\n" % (f, hdr))
            cnt = collections.Counter()
            stall = collections.Counter()
            n = 0
            idx = {h: i for i, h in enumerate(hdr)}
            ki = idx.get("Instruction", None)
            ci = idx.get("Instruction_Comment", None)
            for row in rd:
                if n < 5:
                    o.write("row: %s\n" % row)
                n += 1
                key = (row[ki] if ki is not None else "?", row[ci] if ci is not None else "")
                cnt[key] += 1
                for h in ("Stall_Reason", "Instruction_Type", "Wave_Issued_Instruction"):
                    if h in idx:
                        stall[(h, row[idx[h]])] += 1
            o.write("samples: %d distinct: %d\n" % (n, len(cnt)))
            for k, v in sorted(stall.items(), key=lambda kv: -kv[1]):
                o.write("  %s = %s : %d\n" % (k[0], k[1], v))
            for (ins, com), v in cnt.most_common(4000):
                o.write("%8d  %s  ; %s\n" % (v, ins, com))
