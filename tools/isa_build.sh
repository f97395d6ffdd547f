#!/bin/bash
# DEV TOOL: device-only assembly listing of the library with line tables (input of tools/isa_attrib.py, isa_lines.py, isa_flow.py) and
# the kernels' resource usage.  usage: tools/isa_build.sh [out.s] [extra hipcc flags...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-/tmp/isa/kg.s}; shift || true
mkdir -p "$(dirname "$O")"
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -x hip --cuda-device-only -gline-tables-only -S "$@" \
    "$R/lamejs_amd/csrc/lhip_api.cpp" -o "$O" 2>/dev/null
python3 - "$O" <<'P'
import re, sys
name = None
for l in open(sys.argv[1]):
    m = re.match(r'\s*\.amdhsa_kernel\s+(\S+)', l)
    if m: name = m.group(1); d = {}
    m = re.match(r'\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)\s+(\S+)', l)
    if m and name: d[m.group(1)] = m.group(2)
    if l.strip() == '.end_amdhsa_kernel' and name:
        print(f"{name[:40]:40s} vgpr {d.get('next_free_vgpr'):>4} accum_off {d.get('accum_offset'):>4} sgpr {d.get('next_free_sgpr'):>4} lds {d.get('group_segment_fixed_size'):>6} scratch {d.get('private_segment_fixed_size'):>5}")
        name = None
P
