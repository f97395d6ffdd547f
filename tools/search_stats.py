#!/usr/bin/env python3
"""DEV TOOL (CPU only): the shape of the quantization search per granule-channel on the bench materials -- bin-search depth, step-up evaluations, and the
lengths of the `while (bits > huff_bits) gain++` runs of the outer loop -- from the one-lane host simulation (LAMEJS_SEARCH_STATS=1, k_quant.h SearchStats).
Prices evaluating a run's gains side by side on idle waves of the one-frame launch (VERDICT round 5, next #1a).

    python tools/search_stats.py [frames]          -> one block per material on stderr/stdout
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CODE = r"""
import sys, numpy as np
sys.path.insert(0, r'%(root)s'); sys.path.insert(0, r'%(root)s/tests')
import lamejs_amd, pcm
lib = lamejs_amd.load_library(r'%(root)s/tests/hostsim/_build/liblamejs_hostsim.so')
mat, ch, nfr = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if mat == 'fixture':
    L = np.fromfile(r'%(root)s/tests/golden/left44100_full.s16', dtype=np.int16); R = np.fromfile(r'%(root)s/tests/golden/right44100_full.s16', dtype=np.int16) if ch == 2 else None
else:
    L, R = pcm.CORPORA[mat](1152 * nfr, ch)
enc = lamejs_amd.Mp3Encoder(ch, 44100, 128, lib=lib)
out = enc.encodeBuffer(L, R) + enc.flush()
print(mat, ch, 'channels:', len(L) // 1152, 'frames,', len(out), 'bytes')
"""


def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    subprocess.run(["make", "-C", str(ROOT / "tests" / "hostsim"), "all"], check=True, capture_output=True)
    for mat, ch in (("sine", 2), ("sine", 1), ("fixture", 2), ("fixture", 1), ("bursts", 2)):
        r = subprocess.run([sys.executable, "-c", CODE % {"root": ROOT}, mat, str(ch), str(nfr)], capture_output=True, text=True, env=dict(os.environ, LAMEJS_SEARCH_STATS="1"))
        print("==", r.stdout.strip())
        print(r.stderr.strip())


if __name__ == "__main__":
    main()
