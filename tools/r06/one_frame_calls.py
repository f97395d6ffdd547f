"""DEV TOOL (GPU box): N calls of 1152 samples through the C ABI (the reference's documented call pattern) -- the process rocprofv3 watches in tools/measure_round6_a.sh"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd, pcm
ch, n = int(sys.argv[1]), int(sys.argv[2])
L, R = pcm.CORPORA["sine"](1152 * (n + 2), ch)
enc = lamejs_amd.Mp3Encoder(ch, 44100, 128)
nb = 0
for c in range(n):
    nb += len(enc.encodeBuffer(L[1152 * c:1152 * (c + 1)], None if R is None else R[1152 * c:1152 * (c + 1)]))
print("calls", n, "bytes", nb)
