#!/bin/bash
# DEV TOOL: registers / LDS / scratch of every kernel in a built library's gfx950 code object (what tests/test_abi.py::test_device_code_resources reads).
# usage: tools/co_resources.sh [library.so]
L=$(realpath ${1:-$(dirname "$0")/../lamejs_amd/lib/liblamejs_hip.so}); D=$(mktemp -d); cd $D; cp "$L" l.so
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading l.so >/dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes *amdgcn* | python3 -c "
import re,sys
for blk in sys.stdin.read().split('- .agpr_count:')[1:]:
    f={k:v for k,v in re.findall(r'\.(name|private_segment_fixed_size|vgpr_count|sgpr_count|group_segment_fixed_size):\s+(\S+)',blk)}
    print(f['name'][:46].ljust(46),'vgpr',f['vgpr_count'].rjust(3),'sgpr',f['sgpr_count'].rjust(3),'lds',f['group_segment_fixed_size'].rjust(6),'scratch',f['private_segment_fixed_size'])
"
rm -rf $D
