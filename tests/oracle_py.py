"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/_ref/liblame_oracle.so)."""
import ctypes
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_SO = ROOT / "oracle" / "_ref" / "liblame_oracle.so"
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not _SO.exists():
            subprocess.run(["make", "-C", str(ROOT / "oracle"), "all"], check=True, capture_output=True)
        lib = ctypes.CDLL(str(_SO))
        lib.lo_create.restype = ctypes.c_void_p
        lib.lo_create.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        lib.lo_destroy.argtypes = [ctypes.c_void_p]
        lib.lo_encode.restype = ctypes.c_long
        lib.lo_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        lib.lo_flush.restype = ctypes.c_long
        lib.lo_flush.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        _lib = lib
    return _lib


def oracle_encode(channels, samplerate, kbps, left, right=None, chunk=None, flush=True, joint=False, reservoir=False) -> bytes:
    sys.path.insert(0, str(ROOT))
    from lamejs_amd import tables_blob

    lib = _load()
    blob = tables_blob(channels, samplerate, kbps, joint, reservoir)
    buf = ctypes.create_string_buffer(blob, len(blob))
    h = lib.lo_create(buf, len(blob))
    if not h:
        raise RuntimeError("lo_create failed")
    L = np.ascontiguousarray(left, dtype=np.int16)
    R = L if (channels == 1 or right is None) else np.ascontiguousarray(right, dtype=np.int16)
    n = len(L)
    chunk = chunk or max(n, 1)
    cap = (n // 1152 + 8) * 1500 + 16384
    out = np.empty(cap, dtype=np.uint8)
    off = 0
    try:
        for p in range(0, n, chunk):
            m = min(chunk, n - p)
            w = lib.lo_encode(h, L[p:].ctypes.data, R[p:].ctypes.data, m, out[off:].ctypes.data, cap - off)
            if w < 0:
                raise RuntimeError(f"lo_encode {w}")
            off += w
        if flush:
            w = lib.lo_flush(h, out[off:].ctypes.data, cap - off)
            if w < 0:
                raise RuntimeError(f"lo_flush {w}")
            off += w
    finally:
        lib.lo_destroy(h)
    return out[:off].tobytes()
