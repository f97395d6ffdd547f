#!/usr/bin/env python3
"""TEST TOOL: dense-noise stereo material at the configurations with the LARGEST frames (1440 bytes at 32 kHz / 320 kbps and
8 kHz / 160 kbps, 1152 at 32 kHz / 256 kbps, 1045 at 44.1 kHz / 320 kbps): full-scale noise fills the whole frame with Huffman
data, which a sine does not -- the bit-packing buffer (BitsLds) must hold the whole frame.  Used by the GPU tier (C ABI on the
device) and, under AddressSanitizer, by the CPU tier (host simulation of the kernel bodies):
    LD_PRELOAD=libasan.so python tests/tools/large_frames.py <liblamejs_hostsim_asan.so>
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

CFGS = [(32000, 320), (32000, 256), (44100, 320), (8000, 160), (48000, 320), (16000, 160), (24000, 160)]


def run(lib=None, nframes=6, cfgs=CFGS):
    import lamejs_amd
    from oracle_py import oracle_encode
    rng = np.random.default_rng(5)
    bad = []
    for sr, kb in cfgs:
        n = 1152 * nframes
        L = rng.integers(-30000, 30000, n).astype(np.int16)
        R = rng.integers(-30000, 30000, n).astype(np.int16)
        enc = lamejs_amd.Mp3Encoder(2, sr, kb, lib=lib)
        got = enc.encodeBuffer(L, R) + enc.flush()
        enc.close()
        want = oracle_encode(2, sr, kb, L, R)
        fb = (144000 if sr >= 32000 else 72000) * kb // sr
        # the material must really use the tail of the frame, or the case proves nothing
        tail_used = any(any(want[k * fb + 1092: k * fb + fb - 4]) for k in range(2, len(want) // fb - 2))    # beyond the old 1088-byte buffer
        if got != want or (fb >= 1100 and not tail_used):
            bad.append((sr, kb, len(got), len(want), tail_used))
    return bad


if __name__ == "__main__":
    import lamejs_amd
    lib = lamejs_amd.load_library(sys.argv[1]) if len(sys.argv) > 1 else None
    b = run(lib)
    print("large-frame cases:", "OK" if not b else b)
    sys.exit(1 if b else 0)
