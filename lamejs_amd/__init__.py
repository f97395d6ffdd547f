"""lamejs_amd -- MI355X-native MP3 frame-encode path behind the lamejs ``Mp3Encoder`` API.

Python mirror of the reference's operator interface for this path (``src/js/index.js:66-136``):
``Mp3Encoder(channels, samplerate, kbps).encodeBuffer(left[, right]) -> bytes`` and ``.flush()``.
It is a thin ctypes binding over the C ABI of ``include/lamejs_hip.h`` (``lib/liblamejs_hip.so``,
hand-written HIP for gfx950).  There is no CPU fallback: if the shared library is missing or no
HIP device is visible, construction raises.

The production host is JavaScript (``lamejs_amd/js/index.js`` + N-API addon); this mirror exists
so that the pytest suite and ``bench.py`` can drive exactly the same C ABI.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
_LIB_PATH = _PKG / "lib" / "liblamejs_hip.so"
_TABLE_DIR = _PKG / "tables"

__all__ = ["Mp3Encoder", "load_library", "tables_blob", "LhipError", "encode_streams"]


class LhipError(RuntimeError):
    pass


class _Config(ctypes.Structure):
    _fields_ = [("channels", ctypes.c_int32), ("samplerate", ctypes.c_int32), ("kbps", ctypes.c_int32),
                ("device", ctypes.c_int32)]


_lib = None


def load_library(path: os.PathLike | None = None) -> ctypes.CDLL:
    """Load the HIP shared library (built by ``__graft_entry__.build()``); raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else Path(os.environ.get("LAMEJS_HIP_LIB", _LIB_PATH))
    if not p.exists():
        raise LhipError(f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                        "lamejs_amd has no CPU fallback.")
    lib = ctypes.CDLL(str(p))
    lib.lhip_device_count.restype = ctypes.c_int
    lib.lhip_create.restype = ctypes.c_int
    lib.lhip_create.argtypes = [ctypes.POINTER(_Config), ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.lhip_encode.restype = ctypes.c_int64
    lib.lhip_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    lib.lhip_flush.restype = ctypes.c_int64
    lib.lhip_flush.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.lhip_destroy.restype = None
    lib.lhip_destroy.argtypes = [ctypes.c_void_p]
    lib.lhip_max_output_bytes.restype = ctypes.c_size_t
    lib.lhip_max_output_bytes.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.lhip_encode_output_bytes.restype = ctypes.c_int64
    lib.lhip_encode_output_bytes.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    for name in ("lhip_encode_batch", "lhip_flush_batch", "lhip_encode_batch_device"):
        getattr(lib, name).restype = ctypes.c_int
    lib.lhip_set_hip_stream.restype = ctypes.c_int
    lib.lhip_set_hip_stream.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.lhip_last_batch_stats.restype = None
    lib.lhip_last_batch_stats.argtypes = [ctypes.POINTER(ctypes.c_int64)] * 3
    lib.lhip_debug_read.restype = ctypes.c_int64
    lib.lhip_debug_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    lib.lhip_last_error.restype = ctypes.c_char_p
    lib.lhip_version.restype = ctypes.c_char_p
    if path is None:
        _lib = lib
    return lib


def tables_blob(channels: int, samplerate: int, kbps: int, joint: bool = False, reservoir: bool = False) -> bytes:
    """The LHTB table blob for a configuration.

    Built by the host-side JavaScript ``lamejs_amd/js/tables.js`` (so every transcendental comes
    from the same engine the reference uses).  Blobs for the BASELINE configurations are generated
    at build time into ``lamejs_amd/tables/``; other configurations are generated on demand when
    ``node`` is available.  ``joint``: the reference's joint-stereo mode (an extension: its own
    ``Mp3Encoder`` never selects it, index.js:105); only meaningful for two channels.
    """
    joint = bool(joint) and channels == 2
    f = _TABLE_DIR / f"t_{channels}_{samplerate}_{kbps}{'_joint' if joint else ''}{'_resv' if reservoir else ''}.bin"
    gen = _PKG / "js" / "tables.js"
    if not f.exists() or f.stat().st_mtime < gen.stat().st_mtime:      # a cached blob older than its generator is stale
        _TABLE_DIR.mkdir(exist_ok=True)
        try:
            subprocess.run(["node", str(_PKG / "js" / "tables.js"), str(channels), str(samplerate), str(kbps), str(f)] + (["joint"] if joint else []) + (["reservoir"] if reservoir else []),
                           check=True, capture_output=True, text=True)
        except (OSError, subprocess.CalledProcessError) as e:  # pragma: no cover
            msg = getattr(e, "stderr", "") or str(e)
            if isinstance(e, OSError) and f.exists():      # no node on this machine: use the blob that was shipped
                return f.read_bytes()
            raise LhipError(f"no table blob for ({channels},{samplerate},{kbps}{',joint' if joint else ''}) and node could not build it: {msg}")
    return f.read_bytes()


def _as_i16(a) -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=np.int16)
    if arr.ndim != 1:
        raise ValueError("PCM must be a 1-D Int16 array")
    return arr


class Mp3Encoder:
    """Mirror of the reference's ``Mp3Encoder`` (index.js:66-136)."""

    def __init__(self, channels: int = 1, samplerate: int = 44100, kbps: int = 128, device: int = -1, lib=None, joint: bool = False, reservoir: bool = False):
        """``joint`` (extension, not in the reference's wrapper): encode two channels in the reference's joint-stereo mode --
        per frame mid/side or left/right, as its encoder core decides when asked for MPEGMode.JOINT_STEREO.
        ``reservoir`` (extension): encode with the bit reservoir in use (the reference's wrapper disables it, index.js:108); the frames
        of a stream then form a serial chain, so only batches of many streams use the GPU well."""
        self._lib = lib or load_library()
        self.channels, self.samplerate, self.kbps = int(channels), int(samplerate), int(kbps)
        self._resv = bool(reservoir)
        blob = tables_blob(self.channels, self.samplerate, self.kbps, joint, reservoir)
        cfg = _Config(self.channels, self.samplerate, self.kbps, device)
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(blob, len(blob))
        rc = self._lib.lhip_create(ctypes.byref(cfg), buf, len(blob), ctypes.byref(h))
        if rc != 0:
            raise LhipError(f"lhip_create failed ({rc}): {self._lib.lhip_last_error().decode()}")
        self._h = h

    # ---- frame-range sharding of one stream (extension; include/lamejs_hip.h: lhip_seek / lhip_state_get / lhip_state_set) ----
    def seek_tail_samples(self) -> int:
        self._lib.lhip_seek_tail_samples.restype = ctypes.c_size_t
        self._lib.lhip_seek_tail_samples.argtypes = [ctypes.c_void_p]
        return int(self._lib.lhip_seek_tail_samples(self._h))

    def seek(self, sample_pos: int, tail_left, tail_right=None) -> None:
        """Put this FRESH encoder at input position ``sample_pos`` (a whole number >= 2 of frames); ``tail_*``: the
        ``seek_tail_samples()`` input samples in front of that position."""
        l = _as_i16(tail_left)
        if self.channels == 2 and tail_right is None:
            raise ValueError("seek: a two-channel stream needs both tails")
        r = l if self.channels == 1 else _as_i16(tail_right)
        assert len(l) == self.seek_tail_samples() == len(r)
        self._lib.lhip_seek.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        rc = self._lib.lhip_seek(self._h, int(sample_pos), l.ctypes.data, r.ctypes.data)
        if rc != 0:
            raise LhipError(f"lhip_seek failed ({rc}): {self._lib.lhip_last_error().decode()}")

    def state_get(self) -> bytes:
        """The complete carried state of the stream (host counters + device record): equal blobs = equal futures."""
        self._lib.lhip_state_bytes.restype = ctypes.c_size_t
        self._lib.lhip_state_bytes.argtypes = [ctypes.c_void_p]
        n = int(self._lib.lhip_state_bytes(self._h))
        buf = ctypes.create_string_buffer(n)
        self._lib.lhip_state_get.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        rc = self._lib.lhip_state_get(self._h, buf, n)
        if rc != 0:
            raise LhipError(f"lhip_state_get failed ({rc}): {self._lib.lhip_last_error().decode()}")
        return buf.raw

    def state_set(self, blob: bytes) -> None:
        buf = ctypes.create_string_buffer(blob, len(blob))
        self._lib.lhip_state_set.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        rc = self._lib.lhip_state_set(self._h, buf, len(blob))
        if rc != 0:
            raise LhipError(f"lhip_state_set failed ({rc}): {self._lib.lhip_last_error().decode()}")

    def encodeBuffer(self, left, right=None) -> bytes:
        l = _as_i16(left)
        r = l if (self.channels == 1 or right is None) else _as_i16(right)
        if len(l) != len(r):
            raise ValueError("left/right length mismatch")
        if len(l) == 0:
            return b""
        # the N-API binding's protocol: the result array is allocated at lhip_encode_output_bytes() and written in place
        cap = self._lib.lhip_encode_output_bytes(self._h, len(l))
        if cap < 0:
            raise LhipError(f"lhip_encode_output_bytes failed ({cap}): {self._lib.lhip_last_error().decode()}")
        out = np.empty(cap, dtype=np.uint8)
        n = self._lib.lhip_encode(self._h, l.ctypes.data, r.ctypes.data, len(l), out.ctypes.data, cap)
        if n < 0:
            raise LhipError(f"lhip_encode failed ({n}): {self._lib.lhip_last_error().decode()}")
        if not self._resv and n != cap:      # exact without the bit reservoir: that is what lets a binding skip the copy
            raise LhipError(f"lhip_encode returned {n} bytes, lhip_encode_output_bytes promised {cap}")
        return out[:n].tobytes()

    def flush(self) -> bytes:
        cap = self._lib.lhip_max_output_bytes(self._h, 4 * 1152)
        out = np.empty(cap, dtype=np.uint8)
        n = self._lib.lhip_flush(self._h, out.ctypes.data, cap)
        if n < 0:
            raise LhipError(f"lhip_flush failed ({n}): {self._lib.lhip_last_error().decode()}")
        return out[:n].tobytes()

    def last_batch_stats(self):
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._lib.lhip_last_batch_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return {"frames": a.value, "repaired_frames": b.value, "repair_iterations": c.value}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lhip_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def encode_streams(encoders, lefts, rights=None, flush=True):
    """Batch extension (BASELINE config 5): one launch for many independent streams.

    encoders: list of Mp3Encoder with identical configuration; lefts/rights: per-stream Int16 arrays.
    Returns a list of bytes objects (encode [+ flush] output per stream)."""
    lib = encoders[0]._lib
    n = len(encoders)
    L = [_as_i16(a) for a in lefts]
    R = L if rights is None else [_as_i16(a) for a in rights]
    H = (ctypes.c_void_p * n)(*[e._h for e in encoders])
    lp = (ctypes.c_void_p * n)(*[a.ctypes.data for a in L])
    rp = (ctypes.c_void_p * n)(*[a.ctypes.data for a in R])
    ns = (ctypes.c_size_t * n)(*[len(a) for a in L])
    caps = [lib.lhip_max_output_bytes(e._h, len(a)) for e, a in zip(encoders, L)]
    outs = [np.empty(c, dtype=np.uint8) for c in caps]
    op = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    cp = (ctypes.c_size_t * n)(*caps)
    wr = (ctypes.c_int64 * n)()
    rc = lib.lhip_encode_batch(H, n, lp, rp, ns, op, cp, wr)
    if rc != 0:
        raise LhipError(f"lhip_encode_batch failed ({rc}): {lib.lhip_last_error().decode()}")
    res = [outs[i][: wr[i]].tobytes() for i in range(n)]
    if flush:
        caps2 = [lib.lhip_max_output_bytes(e._h, 4 * 1152) for e in encoders]
        outs2 = [np.empty(c, dtype=np.uint8) for c in caps2]
        op2 = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs2])
        cp2 = (ctypes.c_size_t * n)(*caps2)
        rc = lib.lhip_flush_batch(H, n, op2, cp2, wr)
        if rc != 0:
            raise LhipError(f"lhip_flush_batch failed ({rc}): {lib.lhip_last_error().decode()}")
        res = [res[i] + outs2[i][: wr[i]].tobytes() for i in range(n)]
    return res
