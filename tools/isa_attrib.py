#!/usr/bin/env python3
"""DEV TOOL: static attribution of g_quant's ISA to k_quant.h source functions.

Build the listing first:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -x hip --cuda-device-only -gline-tables-only -S lamejs_amd/csrc/lhip_api.cpp -o kg.s
usage: isa_attrib.py kg.s [kernel-symbol-substring]
Every instruction is attributed to the last `.loc` that pointed into k_quant.h (helpers inlined from lhip_wave.h /
lhip_math.h are charged to their call site).  Prints VALU / SGPR-spill (v_readlane/v_writelane) / SALU / LDS / VMEM counts
per source function (by line range) -- static counts, to be weighted with the phase call counts of phase_prof.py.
"""
import re, sys, collections
path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "g_quant"
src = open("lamejs_amd/csrc/k_quant.h").read().split("\n")
# function start lines
funcs = []
for i, l in enumerate(src, 1):
    m = re.match(r"^(?:template.*)?LHIP_DEV\s+[\w:<>\*& ]+?\s+(\w+)\s*\(", l)
    if m and not l.rstrip().endswith(";"):
        funcs.append((i, m.group(1)))
def func_of(line):
    name = "?"
    for s, n in funcs:
        if s <= line: name = n
        else: break
    return name
lines = open(path).read().split("\n")
fileno = None
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"k_quant\.h"', l)
    if m: fileno = m.group(1)
inside = False
cur = 0
stat = collections.defaultdict(lambda: collections.Counter())
perline = collections.defaultdict(lambda: collections.Counter())
for l in lines:
    if not inside:
        if re.match(r"^_Z\w*%s\w*:" % kern, l): inside = True
        continue
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        if m.group(1) == fileno: cur = int(m.group(2))
        continue
    m = re.match(r"\s+([a-z][a-z0-9_]+)", l)
    if not m: continue
    op = m.group(1)
    if op.startswith("v_readlane") or op.startswith("v_writelane"): k = "spill"
    elif op.startswith("v_"): k = "valu"
    elif op.startswith("s_"): k = "salu"
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith(("global_", "flat_", "scratch_", "buffer_")): k = "vmem"
    else: continue
    stat[func_of(cur)][k] += 1
    perline[cur][k] += 1
tot = collections.Counter()
print("%-28s %6s %6s %6s %5s %5s" % ("function", "valu", "spill", "salu", "lds", "vmem"))
for f, c in sorted(stat.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["spill"])):
    print("%-28s %6d %6d %6d %5d %5d" % (f, c["valu"], c["spill"], c["salu"], c["lds"], c["vmem"]))
    tot.update(c)
print("%-28s %6d %6d %6d %5d %5d" % ("TOTAL", tot["valu"], tot["spill"], tot["salu"], tot["lds"], tot["vmem"]))
if len(sys.argv) > 3:
    print("\nper line (top 60 by valu+spill):")
    for ln, c in sorted(perline.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["spill"]))[:60]:
        print("%5d %5d %5d %5d  %s" % (ln, c["valu"], c["spill"], c["salu"], src[ln - 1].strip()[:110] if 0 < ln <= len(src) else ""))
