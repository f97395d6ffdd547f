#!/bin/bash
# DEV TOOL: the phase-profiling build of the library (tests/tools/phase_prof.py): per-phase cycle counters in g_quant / g_psyA.
# Never loaded by the product or the tests.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = handoff ]; then   # the production code + the count helper's hand-over legs only (tests/tools/handoff_prof.py)
    shift
    exec ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -fPIC -shared -x hip -DLHIP_HANDOFF_PROF "$@" "$R/lamejs_amd/csrc/lhip_api.cpp" -o "$R/lamejs_amd/lib/liblamejs_hip_ho.so"
fi
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -fPIC -shared -x hip -DLHIP_PHASE_PROF \
    "$R/lamejs_amd/csrc/lhip_api.cpp" -o "$R/lamejs_amd/lib/liblamejs_hip_prof.so"
