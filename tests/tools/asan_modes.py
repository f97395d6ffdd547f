"""TEST TOOL: the extension modes (joint stereo, bit reservoir, both), the one-frame-per-stream frame program and the frame-range
sharding calls through a library
given on the command line -- run by tests/test_hostsim_parity.py against the AddressSanitizer build of the host simulation.
usage: python tests/tools/asan_modes.py <liblamejs_hostsim_asan.so>   (prints OK, exit code 0, if every output equals the oracle's)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd, pcm
from oracle_py import oracle_encode

lib = lamejs_amd.load_library(sys.argv[1])
bad = 0
for corpus, ch, sr, kbps, nfr, chunk, joint, resv in [("bursts", 2, 44100, 128, 12, 1152, False, True), ("centre_bursts", 2, 22050, 64, 10, 777, True, True),
                                                      ("bursts", 1, 8000, 8, 30, 576, False, True), ("bursts", 2, 44100, 320, 8, 1152, True, False),
                                                      ("bursts", 1, 48000, 24, 12, 1152, False, True), ("bursts", 1, 44100, 128, 6, 1152, False, False)]:
    L, R = pcm.CORPORA[corpus](1152 * nfr, ch)
    enc = lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib, joint=joint, reservoir=resv)
    out = b""
    for p in range(0, len(L), chunk):
        out += enc.encodeBuffer(L[p:p + chunk], None if R is None else R[p:p + chunk])
    out += enc.flush()
    enc.close()
    if out != oracle_encode(ch, sr, kbps, L, R, joint=joint, reservoir=resv):
        print("MISMATCH", corpus, ch, sr, kbps, joint, resv)
        bad += 1
# frame-range sharding of one stream (lhip_seek / lhip_state_get / lhip_state_set) under the same sanitizer build
sys.path.insert(0, str(ROOT / "tests" / "tools"))
import fuzz_shard
bad += len(fuzz_shard.run(6, 3, lib=lib, verbose=False))
print("OK" if not bad else "FAILED")
sys.exit(1 if bad else 0)
