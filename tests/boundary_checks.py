"""TEST helper shared by the CPU tier (host simulation) and the GPU tier: the error paths of the C ABI (include/lamejs_hip.h) and
what a stream looks like after them.

Contract (INTEGRATION.md "Error codes"): -1 = an output buffer is too small for what the call may produce, -3 = not a live stream
handle, -4 = anything else.  A call that fails has consumed NOTHING: the same call with a larger buffer gives the bytes a stream that
never failed gives.  (The reference's Mp3Encoder sizes its own buffer -- index.js:117-130 -- so its -1, Lame.js:1634-1667 with the
frame already encoded into the internal bitstream, cannot be reached through the drop-in API; the C ABI's callers get the stronger
guarantee.)"""
import ctypes

import numpy as np

ERR_SMALL, ERR_HANDLE, ERR_INTERNAL = -1, -3, -4


def _ptr(a):
    return a.ctypes.data if a is not None else None


def run_boundary_checks(lib, oracle_encode):
    import lamejs_amd
    import pcm

    lib.lhip_flush_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.lhip_encode_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.lhip_state_get.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.lhip_state_bytes.restype = ctypes.c_size_t
    lib.lhip_state_bytes.argtypes = [ctypes.c_void_p]
    for ch, sr, kbps, resv in ((2, 44100, 128, False), (1, 22050, 64, False), (1, 44100, 128, True)):
        nfr = 9
        L, R = pcm.bursts(1152 * nfr + 300, ch, seed=4100 + ch)
        want = oracle_encode(ch, sr, kbps, L, R, reservoir=resv)
        enc = lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib, reservoir=resv)
        h = enc._h
        big = np.empty(lib.lhip_max_output_bytes(h, len(L)), dtype=np.uint8)
        small = np.empty(100, dtype=np.uint8)
        r_ = L if R is None else R
        st0 = enc.state_get()
        # ---- what a binding sizes its result with: exact without the bit reservoir (and said to be), a bound with it; a garbage handle is refused
        assert lib.lhip_output_bytes_is_exact(h) == (0 if resv else 1)
        assert lib.lhip_output_bytes_is_exact(ctypes.c_void_p(0)) == ERR_HANDLE
        if not resv:
            assert lib.lhip_encode_output_bytes(h, len(L)) == len(oracle_encode(ch, sr, kbps, L, R, flush=False))
        else:
            assert lib.lhip_encode_output_bytes(h, len(L)) == lib.lhip_max_output_bytes(h, len(L))
        # ---- -1 from lhip_encode: nothing consumed; the same call with room gives the right bytes
        assert lib.lhip_encode(h, L.ctypes.data, r_.ctypes.data, len(L), small.ctypes.data, len(small)) == ERR_SMALL
        assert b"too small" in lib.lhip_last_error()
        assert enc.state_get() == st0, "a failed lhip_encode changed the stream"
        n1 = lib.lhip_encode(h, L.ctypes.data, r_.ctypes.data, len(L), big.ctypes.data, len(big))
        assert n1 >= 0
        got = big[:n1].tobytes()
        # ---- -1 from lhip_flush: the flush can be repeated
        st1 = enc.state_get()
        assert lib.lhip_flush(h, small.ctypes.data, 10) == ERR_SMALL
        assert enc.state_get() == st1, "a failed lhip_flush changed the stream"
        n2 = lib.lhip_flush(h, big.ctypes.data, len(big))
        assert n2 > 0
        got += big[:n2].tobytes()
        assert got == want, (ch, sr, kbps, resv)
        assert lib.lhip_flush(h, big.ctypes.data, len(big)) == 0                     # a second flush returns nothing
        enc.close()
        # ---- batch: one entry too small fails the whole batch with -1 and consumes nothing of ANY stream
        encs = [lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib, reservoir=resv) for _ in range(3)]
        H = (ctypes.c_void_p * 3)(*[e._h for e in encs])
        lp = (ctypes.c_void_p * 3)(*[L.ctypes.data] * 3)
        rp = (ctypes.c_void_p * 3)(*[r_.ctypes.data] * 3)
        ns = (ctypes.c_size_t * 3)(*[len(L)] * 3)
        outs = [np.empty(len(big), dtype=np.uint8) for _ in range(3)]
        op = (ctypes.c_void_p * 3)(*[o.ctypes.data for o in outs])
        caps = (ctypes.c_size_t * 3)(len(big), 64, len(big))
        wr = (ctypes.c_int64 * 3)()
        states = [e.state_get() for e in encs]
        assert lib.lhip_encode_batch(H, 3, lp, rp, ns, op, caps, wr) == ERR_SMALL
        assert wr[1] == ERR_SMALL and wr[0] < 0 and wr[2] < 0
        assert [e.state_get() for e in encs] == states
        caps[1] = len(big)
        assert lib.lhip_encode_batch(H, 3, lp, rp, ns, op, caps, wr) == 0
        heads = [outs[i][: wr[i]].tobytes() for i in range(3)]
        caps[2] = 8
        assert lib.lhip_flush_batch(H, 3, op, caps, wr) == ERR_SMALL and wr[2] == ERR_SMALL
        caps[2] = len(big)
        assert lib.lhip_flush_batch(H, 3, op, caps, wr) == 0
        for i in range(3):
            assert heads[i] + outs[i][: wr[i]].tobytes() == want
        # a flush batch in which one stream has been flushed already: it contributes nothing, the others are complete (with the
        # bit reservoir this is the case that used to leave the live streams' bitstreams unpadded)
        e2 = [lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib, reservoir=resv) for _ in range(3)]
        for e in e2:
            assert e.encodeBuffer(L, R) == heads[0]
        tail0 = e2[0].flush()
        H2 = (ctypes.c_void_p * 3)(*[e._h for e in e2])
        assert lib.lhip_flush_batch(H2, 3, op, caps, wr) == 0
        assert wr[0] == 0
        for i in (1, 2):
            assert outs[i][: wr[i]].tobytes() == tail0, (ch, sr, kbps, resv, i, wr[i], len(tail0))
        for e in encs + e2:
            e.close()
    # ---- -3: destroyed and garbage handles, on every entry point that takes one
    enc = lamejs_amd.Mp3Encoder(1, 44100, 128, lib=lib)
    h = enc._h
    lib.lhip_destroy(h)
    enc._h = None
    z = np.zeros(2048, dtype=np.int16)
    out = np.empty(8192, dtype=np.uint8)
    garbage = ctypes.create_string_buffer(4096)                                        # zeroed memory: no stream magic in it
    for bad in (h, ctypes.c_void_p(ctypes.addressof(garbage)), None):
        assert lib.lhip_encode(bad, z.ctypes.data, z.ctypes.data, len(z), out.ctypes.data, len(out)) == ERR_HANDLE
        assert lib.lhip_flush(bad, out.ctypes.data, len(out)) == ERR_HANDLE
        assert lib.lhip_state_get(bad, out.ctypes.data, len(out)) == ERR_HANDLE
        assert lib.lhip_state_bytes(bad) == 0 and lib.lhip_max_output_bytes(bad, 1152) == 0
        Hb = (ctypes.c_void_p * 1)(bad)
        one = (ctypes.c_void_p * 1)(z.ctypes.data); n1 = (ctypes.c_size_t * 1)(len(z)); o1 = (ctypes.c_void_p * 1)(out.ctypes.data); c1 = (ctypes.c_size_t * 1)(len(out)); w1 = (ctypes.c_int64 * 1)()
        assert lib.lhip_encode_batch(Hb, 1, one, one, n1, o1, c1, w1) == ERR_HANDLE
        assert lib.lhip_flush_batch(Hb, 1, o1, c1, w1) == ERR_HANDLE
        lib.lhip_destroy(bad)                                                           # a no-op, not a crash
    assert b"bad stream handle" in lib.lhip_last_error()


def run_state_canonical_checks(lib):
    """lhip_state_get is canonical -- a stream fed in odd chunk sizes and one fed in whole frames stand at the same point
    with EQUAL blobs (the samples beyond mf_size are dead); MPEG-2 and MPEG-2.5 blobs are not interchangeable."""
    import lamejs_amd
    import pcm

    L, R = pcm.bursts(1152 * 40, 2, seed=99)
    blobs = []
    for chunk in (1152, 1000, 1700, 333, 1152 * 40):
        e = lamejs_amd.Mp3Encoder(2, 44100, 128, lib=lib)
        for p in range(0, len(L), chunk):
            e.encodeBuffer(L[p:p + chunk], R[p:p + chunk])
        blobs.append(e.state_get())
        e.close()
    assert all(b == blobs[0] for b in blobs), [sum(x != y for x, y in zip(b, blobs[0])) for b in blobs]
    a = lamejs_amd.Mp3Encoder(1, 22050, 32, lib=lib)
    b = lamejs_amd.Mp3Encoder(1, 11025, 32, lib=lib)
    try:
        b.state_set(a.state_get())
    except lamejs_amd.LhipError:
        pass
    else:
        raise AssertionError("an MPEG-2 state blob was accepted by an MPEG-2.5 stream")
    try:
        lamejs_amd.Mp3Encoder(2, 44100, 128, lib=lib).seek(1152 * 4, L[:1680])
    except ValueError:
        pass
    else:
        raise AssertionError("seek of a two-channel stream without the right tail was accepted")
    a.close(); b.close()
