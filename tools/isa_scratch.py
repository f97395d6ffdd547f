#!/usr/bin/env python3
"""DEV TOOL: where a kernel's scratch (register-spill) instructions sit.  Input: the listing of tools/isa_build.sh.
usage: isa_scratch.py kg.s [kernel-symbol-substring]   ->  scratch_load / scratch_store counts per source file:line (innermost frame
of the .loc's inline chain that lies in lamejs_amd/csrc), plus the kernel's private segment size."""
import re, sys, collections
path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "_Z7g_quantILi0EEv5QArgs"
lines = open(path).read().split("\n")
files = {}
inside = False
cur = "?"
cnt = collections.Counter()
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
    if re.match(r'^%s:' % re.escape(kern), l):
        inside = True
        continue
    if inside and re.match(r'\s*\.end_amdhsa_kernel', l):
        break
    if inside and l.startswith(".Lfunc_end"):
        inside = False
    if not inside:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = "%s:%s" % (files.get(m.group(1), m.group(1)), m.group(2))
        mm = re.findall(r'(k_\w+\.h|lhip_\w+\.(?:h|cpp)):(\d+)', l)
        if mm:
            cur = "%s:%s" % mm[0]
        continue
    m = re.match(r'\s+(scratch_(load|store)\w*)', l)
    if m:
        cnt[(cur, m.group(2))] += 1
tot = sum(cnt.values())
print("# %s: %d scratch instructions" % (kern, tot))
def keyf(kv):
    f, ln = kv[0][0].rsplit(":", 1)
    return (f, int(ln) if ln.isdigit() else 0, kv[0][1])
for (loc, kind), n in sorted(cnt.items(), key=keyf):
    print("%-28s %-5s x%d" % (loc, kind, n))
