#!/usr/bin/env python3
"""DEV TOOL: a kernel's ISA as a readable flow -- one instruction per line, prefixed with the innermost k_quant.h source line of
its inline chain (and the inlined helper's file:line when the instruction comes from lhip_wave.h / lhip_math.h).
usage: isa_flow.py <listing.s> <kernel symbol> [first last]   (first/last: only instructions attributed to that k_quant.h range)"""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 1 << 30)
inside = False; cur = 0; inner = ""
for l in open(path):
    if not inside:
        if l.startswith(kern + ":"): inside = True
        continue
    if l.startswith(".Lfunc_end"): break
    m = re.match(r'\s*\.loc\s+\d+\s+\d+.*?;\s*(.*)$', l)
    if m:
        chain = m.group(1)
        mm = re.findall(r'k_quant\.h:(\d+):', chain)
        if mm: cur = int(mm[0])
        first = re.match(r'\S*?/?(\w+\.(?:h|cpp)):(\d+)', chain)
        inner = "" if not first or first.group(1) == "k_quant.h" else f"{first.group(1)[:-2]}:{first.group(2)}"
        continue
    if re.match(r'^\.L\w+:', l):
        if lo <= cur <= hi: print(f"      {l.strip()}")
        continue
    m = re.match(r'\s+([a-z][a-z0-9_]+.*)', l)
    if m and lo <= cur <= hi and not m.group(1).startswith(('.', ';')):
        print(f"{cur:5d} {inner:14s} {m.group(1).split(';')[0].rstrip()}")
