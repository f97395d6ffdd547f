"""DEV TOOL (GPU): what the seed-chain validation of a 1e5-frame batch finds -- frames flagged by the memo-only first pass, frames whose replay asked for a
gain the speculative pass never evaluated (g_fixup re-validates those with real bit counts: its time on steady material) -- per workload."""
import ctypes, sys
from pathlib import Path
import numpy as np
import torch          # (before the library: one HIP runtime per process -- torch's)
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd, pcm
lib = lamejs_amd.load_library()
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for corpus, ch, kbps in [("sine", 1, 128), ("sine", 2, 128), ("sine", 2, 320), ("bursts", 2, 128)]:
    L, R = pcm.CORPORA[corpus](1152 * nfr, ch, seed=12345 if corpus == "sine" else 777)
    dl = torch.from_numpy(L).cuda(); dr = torch.from_numpy(R).cuda() if ch == 2 else dl
    out = torch.empty((nfr + 4) * (144000 * kbps // 44100 + 1), dtype=torch.uint8, device="cuda")
    e2 = lamejs_amd.Mp3Encoder(ch, 44100, kbps, device=0)
    H = (ctypes.c_void_p * 1)(e2._h); wr = (ctypes.c_int64 * 1)()
    rc = lib.lhip_encode_batch_device(H, 1, (ctypes.c_void_p * 1)(dl.data_ptr()), (ctypes.c_void_p * 1)(dr.data_ptr()), (ctypes.c_size_t * 1)(len(L)),
                                      (ctypes.c_void_p * 1)(out.data_ptr()), (ctypes.c_size_t * 1)(out.numel()), wr, 1)
    assert rc == 0, lib.lhip_last_error()
    buf = (ctypes.c_int32 * 64)()
    n = lib.lhip_debug_read(8, buf, 256)
    assert n == 256, lib.lhip_last_error()
    print(f"{corpus} ch={ch} {kbps}k frames={nfr}: first pass flagged {buf[0]}, undecided by the memo (re-validated in g_fixup) {buf[1]}; repaired {buf[32]} in {buf[33]} iteration(s)")
    e2.close()
