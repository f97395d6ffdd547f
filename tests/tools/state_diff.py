"""DEV TOOL (GPU or hostsim): which carried fields differ between the speculated state at a cut (seek + H warm-up frames) and the state
of the stream that encoded up to the cut.  usage: state_diff.py [corpus ch sr kbps nfr H cut...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd, pcm
from state_fields import describe_diff
a = sys.argv[1:]
corpus, ch, sr, kbps, nfr, H = (a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5])) if len(a) >= 6 else ("sine", 2, 44100, 128, 400, 8)
cuts = [int(x) for x in a[6:]] or [100, 200, 300]
import os
lib = lamejs_amd.load_library(os.environ['LAMEJS_LIB']) if os.environ.get('LAMEJS_LIB') else lamejs_amd.load_library()
L, R = pcm.CORPORA[corpus](1152 * nfr, ch)
fs = 1152 if sr >= 32000 else 576
for rep in range(2):
    whole = lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib)
    pos = 0
    for c in cuts:
        whole.encodeBuffer(L[pos:c * fs], None if R is None else R[pos:c * fs]); pos = c * fs
        truth = whole.state_get()
        for HH in (H, 2 * H, 4 * H):
            if c - HH < 2: continue
            e = lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib)
            p0, nt = (c - HH) * fs, e.seek_tail_samples()
            e.seek(p0, L[p0 - nt:p0], None if R is None else R[p0 - nt:p0])
            e.encodeBuffer(L[p0:c * fs], None if R is None else R[p0:c * fs])
            print(f"rep {rep} cut {c} H {HH}: {describe_diff(e.state_get(), truth)}", flush=True)
            e.close()
    whole.close()
