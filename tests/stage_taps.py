"""TEST INFRASTRUCTURE: stage-level differential check (SURVEY.md 4 "stage-level tap points").

The oracle is driven frame by frame with its taps enabled (oracle/lo_common.h `lo_tap`: MDCT output xr, block types, the
masking ratios handed to the quantizer, ATH.adjust, and the quantizer's global_gain / part2_3_length / part2_length of the
frame); the library under test -- the HIP build on the GPU, or the host simulation of the same kernel bodies in the CPU tier --
encodes the same PCM in one batch and its intermediate arrays are read back through lhip_debug_read.  Everything is compared
per granule and channel, bit for bit: a mismatch names the first stage that differs instead of "some byte differs".
"""
import ctypes

import numpy as np

from oracle_py import _load as _load_oracle

SFBMAX = 39
TAP = np.dtype([("xr", "<f4", (2, 2, 576)), ("block_type", "<i4", (2, 2)), ("ratio", "<f4", (2, 2, 122)), ("ath_adjust", "<f8"),
                ("l3_xmin", "<f4", (2, 2, SFBMAX)), ("global_gain", "<i4", (2, 2)), ("part2_3_length", "<i4", (2, 2)),
                ("part2_length", "<i4", (2, 2)), ("mode_ext", "<i4"), ("pe", "<f8", (2, 2)), ("pe_MS", "<f8", (2, 2)),
                ("ms_ener_ratio", "<f8", (2,))], align=True)

# struct GrSide (lamejs_amd/csrc/lhip_defs.h), 32-bit fields in declaration order
_GRSIDE_FIELDS = (["part2_3_length", "part2_length", "big_values", "count1", "global_gain", "scalefac_compress", "block_type"] +
                  [f"table_select{i}" for i in range(3)] + [f"subblock_gain{i}" for i in range(3)] +
                  ["region0_count", "region1_count", "preflag", "scalefac_scale", "count1table_select", "sfbmax", "sfbdivide", "active",
                   "bs_start", "bs_step_in", "bs_gain", "targ_bits", "scfsi"])
GRSIDE = np.dtype([(f, "<i4") for f in _GRSIDE_FIELDS] + [("scalefac", "<i4", (SFBMAX,)), ("bs_ntab", "<i4"), ("bs_tab", "<i4", (24,)),
                                                         ("bs_asg", "<i4", (24,)), ("bs_state", "<i4"), ("mode_ext", "<i4")])


def oracle_stages(channels, samplerate, kbps, L, R, joint=False):
    """Per-frame taps of the oracle: list of TAP records (one per emitted frame, flush excluded)."""
    import lamejs_amd
    lib = _load_oracle()
    lib.lo_enable_tap.argtypes = [ctypes.c_void_p]
    lib.lo_get_tap.restype = ctypes.c_void_p
    lib.lo_get_tap.argtypes = [ctypes.c_void_p]
    lib.lo_tap_size.restype = ctypes.c_size_t
    assert lib.lo_tap_size() == TAP.itemsize, (lib.lo_tap_size(), TAP.itemsize)
    blob = lamejs_amd.tables_blob(channels, samplerate, kbps, joint)
    buf = ctypes.create_string_buffer(blob, len(blob))
    h = lib.lo_create(buf, len(blob))
    assert h
    lib.lo_enable_tap(h)
    L = np.ascontiguousarray(L, dtype=np.int16)
    R = L if (channels == 1 or R is None) else np.ascontiguousarray(R, dtype=np.int16)
    out = np.empty(1 << 16, dtype=np.uint8)
    taps = []
    # the smallest chunk that never completes two frames in one call (mode_gr = 1 configurations: 576-sample frames)
    step = 576
    for p in range(0, len(L), step):
        m = min(step, len(L) - p)
        w = lib.lo_encode(h, L[p:].ctypes.data, R[p:].ctypes.data, m, out.ctypes.data, out.size)
        assert w >= 0
        if w > 0:
            taps.append(np.frombuffer(ctypes.string_at(lib.lo_get_tap(h), TAP.itemsize), dtype=TAP)[0].copy())
    lib.lo_destroy(h)
    return taps


def device_stages(lib, channels, samplerate, kbps, L, R, joint=False):
    """One batch through the library under test, then its intermediate arrays (lhip_debug_read taps 0-4)."""
    import lamejs_amd
    enc = lamejs_amd.Mp3Encoder(channels, samplerate, kbps, lib=lib, joint=joint)
    lib = enc._lib
    mp3 = enc.encodeBuffer(L, R)
    nfr = enc.last_batch_stats()["frames"]
    C = channels
    Cp = 4 if (joint and channels == 2) else C          # psy channels: L, R (+ mid, side in joint stereo)
    GR = 2 if samplerate >= 32000 else 1
    ngs, nfs = GR * nfr + 1, nfr + 1

    def read(what, dtype, count):
        a = np.empty(count, dtype=dtype)
        n = lib.lhip_debug_read(what, a.ctypes.data_as(ctypes.c_void_p), a.nbytes)
        assert n == a.nbytes, (what, n, a.nbytes)
        return a

    st = {"nframes": nfr, "GR": GR,
          "xr": read(0, "<f4", ngs * C * 576).reshape(ngs, C, 576),
          "blocktype": read(1, "<i4", ngs * C).reshape(ngs, C),
          "E": read(2, "<f4", ngs * Cp * 122).reshape(ngs, Cp, 122),
          "ath": read(3, "<f8", nfs),
          "side": read(4, GRSIDE, nfr * 2 * C).reshape(nfr, 2, C)}
    enc.close()
    return st, mp3


def compare_stages(lib, channels, samplerate, kbps, L, R, psfb21_start=None, joint=False):
    """Returns a list of human-readable mismatches (empty = every stage of every granule agrees)."""
    taps = oracle_stages(channels, samplerate, kbps, L, R, joint)
    st, _ = device_stages(lib, channels, samplerate, kbps, L, R, joint)
    bad = []
    assert st["nframes"] == len(taps), (st["nframes"], len(taps))
    GR, C = st["GR"], channels
    u32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    for k, t in enumerate(taps):
        if np.float64(t["ath_adjust"]).view(np.uint64) != st["ath"][1 + k].view(np.uint64):
            bad.append(f"frame {k}: ATH.adjust {st['ath'][1 + k]!r} != {t['ath_adjust']!r}")
        ms = 0
        if joint:                                    # the frame's M/S decision (mode_ext 0 / 2); the maskings handed on are then mid / side
            ms = int(t["mode_ext"])
            if int(st["side"][k, 0, 0]["mode_ext"]) != ms:
                bad.append(f"frame {k}: mode_ext {int(st['side'][k, 0, 0]['mode_ext'])} != {ms} (oracle pe {t['pe'].tolist()} pe_MS {t['pe_MS'].tolist()})")
        for gr in range(GR):
            gs = 1 + GR * k + gr
            for ch in range(C):
                if st["blocktype"][gs, ch] != t["block_type"][gr, ch]:
                    bad.append(f"frame {k} gr {gr} ch {ch}: block type {st['blocktype'][gs, ch]} != {t['block_type'][gr, ch]}")
                # masking handed to the quantizer for this granule = thresholds of the previous psy call (slot gs - 1)
                if not np.array_equal(u32(st["E"][gs - 1, ch + ms]), u32(t["ratio"][gr, ch])):
                    i = int(np.nonzero(u32(st["E"][gs - 1, ch + ms]) != u32(t["ratio"][gr, ch]))[0][0])
                    bad.append(f"frame {k} gr {gr} ch {ch}{' (M/S)' if ms else ''}: masking en/thm differs at index {i}: {st['E'][gs - 1, ch + ms, i]!r} != {t['ratio'][gr, ch, i]!r}")
                # MDCT output.  The quantization kernel writes the zeros of the analog-silence rule back into xr (lines of the
                # pseudo bands above sfb21 / sfb12 below the adjusted ATH, Quantize.js:147-202), the oracle taps xr before it
                d, o = st["xr"][gs, ch], t["xr"][gr, ch]
                diff = u32(d) != u32(o)
                if diff.any():
                    idx = np.nonzero(diff)[0]
                    zeroed = (d[idx] == 0)
                    if not zeroed.all():
                        i = int(idx[~zeroed][0])
                        bad.append(f"frame {k} gr {gr} ch {ch}: xr[{i}] {d[i]!r} != {o[i]!r}")
                    elif psfb21_start is not None and st["blocktype"][gs, ch] != 2 and idx.min() < psfb21_start:
                        bad.append(f"frame {k} gr {gr} ch {ch}: xr zeroed below the pseudo bands at line {int(idx.min())}")
                s = st["side"][k, gr, ch]
                for f in ("global_gain", "part2_3_length", "part2_length"):
                    if s[f] != t[f][gr, ch]:
                        bad.append(f"frame {k} gr {gr} ch {ch}: {f} {s[f]} != {t[f][gr, ch]}")
                if s["block_type"] != t["block_type"][gr, ch]:
                    bad.append(f"frame {k} gr {gr} ch {ch}: side block_type {s['block_type']} != {t['block_type'][gr, ch]}")
        if len(bad) > 20:
            break
    return bad
