"""DEV TOOL (GPU box): does a process that used two aliased contexts (LHIP_ALIAS_DEVICES=2) exit cleanly?  mode: stream | nostream | release | release_all"""
import ctypes, os, sys, threading
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
mode = sys.argv[1]
os.environ["LHIP_ALIAS_DEVICES"] = "2"
if mode == "nostream":
    os.environ["LHIP_ALIAS_NO_STREAM"] = "1"
import lamejs_amd, pcm
lib = lamejs_amd.load_library()
lib.lhip_set_devices.argtypes = [ctypes.c_uint64]
assert lib.lhip_set_devices(0b11) == 2
NFR = int(os.environ.get("NFR", "300"))
mats = [pcm.CORPORA["bursts"](1152 * NFR, 2, seed=900 + i) for i in range(2)]
got = [None, None]
def work(i):
    for rep in range(2):
        enc = lamejs_amd.Mp3Encoder(2, 44100, 128, device=i)
        L, R = mats[i]
        got[i] = enc.encodeBuffer(L, R) + enc.flush()
        enc.close()
    if os.environ.get("DEVSYNC"):
        ctypes.CDLL("libamdhip64.so").hipDeviceSynchronize()
th = [threading.Thread(target=work, args=(i,)) for i in range(2)] if "threads" not in sys.argv else []
if th:
    for t in th: t.start()
    for t in th: t.join()
else:
    work(0); work(1)
lib.lhip_set_devices(0)
if mode.startswith("release"):
    print("release ctx1", lib.lhip_debug_release_context(1))
    if mode == "release_all":
        print("release ctx0", lib.lhip_debug_release_context(0))
print("done", mode, len(got[0]), len(got[1]))
