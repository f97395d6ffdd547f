#!/usr/bin/env python3
"""DEV TOOL: where a kernel's scalar-side overhead sits, by source line: exec-mask triplets (s_and_saveexec / s_cbranch_execz / s_or_b64 exec),
DPP-hazard s_nop, 64-bit literal s_mov_b64 pairs.  Input: the listing of tools/isa_build.sh.
usage: isa_scalar_sites.py kg.s <kernel symbol substring, e.g. g_quantILi0> [top N]"""
import collections, re, sys
path, kern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
inside, cur = False, None
cnt = collections.defaultdict(collections.Counter)
files = {}
for l in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m: files[m.group(1)] = m.group(3)
    if not inside:
        if re.match(r"^_Z\w*%s\w*:" % kern, l): inside = True
        continue
    if l.startswith(".Lfunc_end"): break
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)\s+\d+(.*)$', l)
    if m:
        # innermost frame that lies in k_quant.h / k_quant_tail.h, else the .loc itself
        mm = re.findall(r'(k_quant(?:_tail)?\.h):(\d+):', m.group(3))
        cur = (mm[0][0], int(mm[0][1])) if mm else (files.get(m.group(1), '?').split('/')[-1], int(m.group(2)))
        continue
    m = re.match(r'\s+([a-z][a-z0-9_]+)\s*(.*)', l)
    if not m or cur is None: continue
    op, args = m.group(1), m.group(2)
    c = cnt[cur]
    if op == 's_and_saveexec_b64': c['saveexec'] += 1
    elif op == 's_cbranch_execz': c['execz'] += 1
    elif op == 's_nop': c['nop'] += 1; c['nop_cycles'] += int(args.split()[0]) + 1
    elif op == 's_mov_b64' and ('0x' in args or re.search(r',\s*-?\d', args)): c['mov64lit'] += 1
    elif op == 's_mov_b32' and ('0x' in args or re.search(r',\s*-?\d', args)): c['mov32lit'] += 1
    if op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_') and op not in ('s_waitcnt', 's_nop'): c['salu'] += 1
tot = collections.Counter()
for c in cnt.values(): tot.update(c)
print("totals:", dict(tot))
for key in ('saveexec', 'nop', 'mov64lit', 'mov32lit'):
    print(f"-- top {top} source lines by {key}")
    for (f, ln), c in sorted(cnt.items(), key=lambda kv: -kv[1][key])[:top]:
        if c[key] == 0: break
        print(f"   {f}:{ln:<5d} {key} {c[key]:4d}   (valu {c['valu']}, salu {c['salu']}, nop cycles {c['nop_cycles']})")
