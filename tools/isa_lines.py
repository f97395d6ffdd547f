#!/usr/bin/env python3
"""DEV TOOL: instruction counts per k_quant.h source line (innermost k_quant.h frame of the inline chain): VALU, SGPR spill
traffic (v_readlane / v_writelane), SALU (without s_waitcnt / s_nop), waits+nops, LDS, VMEM.
usage: isa_lines.py <g_quant listing with .loc comments> <first line> <last line>"""
import re, collections, sys
lines = open(sys.argv[1]).read().split('\n')
a, b = int(sys.argv[2]), int(sys.argv[3])
cur = None
cnt = collections.defaultdict(collections.Counter)
for l in lines:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+).*?;\s*(.*)$', l)
    if m:
        mm = re.findall(r'k_quant\.h:(\d+):', m.group(3))
        cur = int(mm[0]) if mm else cur
        continue
    m = re.match(r'\s+([a-z][a-z0-9_]+)', l)
    if m and cur:
        op = m.group(1)
        if op.startswith(('v_readlane', 'v_writelane')): k = 'spill'
        elif op.startswith('v_'): k = 'valu'
        elif op in ('s_waitcnt', 's_nop'): k = 'wait'
        elif op.startswith('s_'): k = 'salu'
        elif op.startswith('ds_'): k = 'lds'
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): k = 'vmem'
        else: k = 'other'
        cnt[cur][k] += 1
src = open('/root/repo/lamejs_amd/csrc/k_quant.h').read().split('\n')
tot = collections.Counter()
print("line  valu spill salu wait lds vmem")
for ln in range(a, b + 1):
    c = cnt.get(ln)
    if c:
        tot.update(c)
        print(f"{ln:5d} {c['valu']:4d} {c['spill']:4d} {c['salu']:5d} {c['wait']:4d} {c['lds']:3d} {c['vmem']:3d}  {src[ln - 1].strip()[:100]}")
print('total', dict(tot))
