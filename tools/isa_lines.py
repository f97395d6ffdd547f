#!/usr/bin/env python3
"""DEV TOOL: VALU instruction count per k_quant.h source line (innermost k_quant.h frame of the inline chain).
usage: isa_lines.py <g_quant listing with .loc comments> <first line> <last line>"""
import re, collections, sys
lines = open(sys.argv[1]).read().split('\n')
a, b = int(sys.argv[2]), int(sys.argv[3])
cur = None; cnt = collections.Counter(); sp = collections.Counter()
for l in lines:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+).*?;\s*(.*)$', l)
    if m:
        mm = re.findall(r'k_quant\.h:(\d+):', m.group(3))
        cur = int(mm[0]) if mm else cur
        continue
    m = re.match(r'\s+([a-z][a-z0-9_]+)', l)
    if m and cur:
        op = m.group(1)
        if op.startswith(('v_readlane', 'v_writelane')): sp[cur] += 1
        elif op.startswith('v_'): cnt[cur] += 1
src = open('/root/repo/lamejs_amd/csrc/k_quant.h').read().split('\n')
t = 0
for ln in range(a, b + 1):
    if cnt[ln] or sp[ln]:
        t += cnt[ln]; print(ln, cnt[ln], sp[ln], src[ln - 1].strip()[:110])
print('total valu', t)
