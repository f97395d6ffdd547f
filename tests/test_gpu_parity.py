"""GPU parity tests (run on the MI355X box with `-m gpu`): the hand-written HIP path, called through the
C ABI, must be BIT-EXACT against (a) the committed goldens of the unmodified reference and (b) the CPU
oracle on seeded inputs.  Integer/byte work: tolerance is zero."""
import ctypes
import hashlib

import numpy as np
import pytest

from conftest import ROOT, load_case_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import lamejs_amd
    l = lamejs_amd.load_library()
    assert l.lhip_device_count() > 0, "no HIP device"
    assert b"HIP gfx950" in l.lhip_version()
    return l


def _encode(ch, kbps, L, R, chunk, sr=44100):
    import lamejs_amd
    enc = lamejs_amd.Mp3Encoder(ch, sr, kbps)
    out = b""
    for p in range(0, len(L), chunk):
        out += enc.encodeBuffer(L[p:p + chunk], None if R is None else R[p:p + chunk])
    out += enc.flush()
    assert enc.flush() == b""
    enc.close()
    return out


def test_device_math_matches_v8(lib):
    """log10 / pow(10,.) / sqrt / division / f32 rounding / no-FMA on the device vs the pinned oracle math."""
    import oracle_py
    o = oracle_py._load()
    o.lo_math.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.default_rng(7)
    n = 400000
    pos = np.exp((rng.random(n) - 0.5) * 120.0)
    cases = {0: pos, 1: (rng.random(n) - 0.5) * 60.0, 2: pos, 3: pos * np.where(rng.random(n) < 0.5, -1, 1),
             4: pos * 1e-3, 5: (rng.random(n) - 0.5) * 1e6, 6: pos}
    for op, x in cases.items():
        x = np.ascontiguousarray(x, dtype=np.float64)
        a = np.empty(n); b = np.empty(n)
        assert lib.lhip_debug_math(op, x.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n) == 0
        o.lo_math(op, x.ctypes.data, b.ctypes.data, n)
        bad = np.nonzero(a.view(np.uint64) != b.view(np.uint64))[0]
        assert bad.size == 0, f"op {op}: {bad.size} mismatches, first x={x[bad[0]]!r} gpu={a[bad[0]]!r} ref={b[bad[0]]!r}"


def test_gpu_matches_reference_goldens(lib, golden):
    n = 0
    for case in golden:
        if case.get("outside_envelope"):
            continue
        L, R = load_case_pcm(case)
        mp3 = _encode(case["channels"], case["kbps"], L, R, case["chunk"], case.get("samplerate", 44100))
        assert len(mp3) == case["mp3_len"], case
        assert hashlib.md5(mp3).hexdigest() == case["mp3_md5"], case
        n += 1
    assert n >= 65


def test_gpu_reference_fixture_md5s(lib, golden):
    """the three md5s of the reference's own fixtures (testdata/Left44100.wav [+ Right44100.wav], 287 x 1152-sample
    calls + flush = 288 frames) -- the values Tests.js' outputs have -- reproduced on the GPU through the C ABI."""
    want = {(1, 128): "5a522d307c7593e9cb57bfd31ff2a18c", (2, 128): "444bd5a0b7af22b0498d2a3fdd77b359", (2, 320): "910219d73827803a67e6e1a89a0794a5"}
    seen = 0
    for case in golden:
        if case["corpus"] != "wavfull":
            continue
        assert case["mp3_md5"] == want[(case["channels"], case["kbps"])]
        L, R = load_case_pcm(case)
        for chunk in (1152, len(L)):                    # the reference's call pattern, and one batch call
            mp3 = _encode(case["channels"], case["kbps"], L, R, chunk)
            assert hashlib.md5(mp3).hexdigest() == case["mp3_md5"], (case, chunk)
        seen += 1
    assert seen == 3


def test_gpu_matches_oracle_seeded(lib):
    import pcm
    from oracle_py import oracle_encode
    for corpus, ch, kbps, nfr, seed in [("bursts", 2, 128, 700, 31), ("bursts", 1, 320, 500, 32), ("sine", 2, 192, 300, 33),
                                        ("bursts", 2, 320, 600, 34), ("bursts", 1, 128, 900, 35)]:
        L, R = pcm.CORPORA[corpus](1152 * nfr, ch, seed=seed)
        got = _encode(ch, kbps, L, R, 1152 * nfr)
        want = oracle_encode(ch, 44100, kbps, L, R)
        assert got == want, (corpus, ch, kbps, seed, _first_diff(got, want))


def _first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return f"first diff at byte {i} (lengths {len(a)}, {len(b)})"
    return f"lengths {len(a)} vs {len(b)}"


def test_gpu_edge_cases(lib):
    import lamejs_amd
    from oracle_py import oracle_encode
    # empty input, input shorter than one frame, silence (inactive granules: seed pass-through), max amplitude
    enc = lamejs_amd.Mp3Encoder(1, 44100, 128)
    assert enc.encodeBuffer(np.zeros(0, dtype=np.int16)) == b""
    assert enc.encodeBuffer(np.zeros(100, dtype=np.int16)) == b""
    enc.close()
    z = np.zeros(1152 * 20, dtype=np.int16)
    assert _encode(1, 128, z, None, 1152) == oracle_encode(1, 44100, 128, z)
    sq = np.where((np.arange(1152 * 30) // 50) % 2 == 0, 32767, -32768).astype(np.int16)
    assert _encode(2, 128, sq, sq[::-1].copy(), 5000) == oracle_encode(2, 44100, 128, sq, sq[::-1].copy())
    # silence followed by sound followed by silence
    rng = np.random.default_rng(5)
    x = np.concatenate([np.zeros(1152 * 12, dtype=np.int16), (rng.normal(0, 6000, 1152 * 15)).astype(np.int16), np.zeros(1152 * 9, dtype=np.int16)])
    assert _encode(2, 128, x, x, 1152 * 40) == oracle_encode(2, 44100, 128, x, x)


@pytest.mark.parametrize("sr,kbps", [(22050, 64), (16000, 32), (8000, 16)])
def test_gpu_edge_cases_lsf(lib, sr, kbps):
    """MPEG-2 / 2.5 (576-sample frames): empty and sub-frame input, silence, full-scale square wave, silence-sound-silence,
    and the flush that may emit two frames per 1152-sample bunch."""
    import lamejs_amd
    from oracle_py import oracle_encode
    enc = lamejs_amd.Mp3Encoder(1, sr, kbps)
    assert enc.encodeBuffer(np.zeros(0, dtype=np.int16)) == b""
    assert enc.encodeBuffer(np.zeros(100, dtype=np.int16)) == b""
    enc.close()
    z = np.zeros(576 * 21, dtype=np.int16)
    assert _encode(1, kbps, z, None, 576, sr) == oracle_encode(1, sr, kbps, z)
    sq = np.where((np.arange(1152 * 30) // 50) % 2 == 0, 32767, -32768).astype(np.int16)
    assert _encode(2, kbps, sq, sq[::-1].copy(), 5000, sr) == oracle_encode(2, sr, kbps, sq, sq[::-1].copy())
    rng = np.random.default_rng(5)
    x = np.concatenate([np.zeros(1152 * 12, dtype=np.int16), (rng.normal(0, 6000, 1152 * 15)).astype(np.int16), np.zeros(1152 * 9 + 17, dtype=np.int16)])
    assert _encode(2, kbps, x, x, 1152 * 40, sr) == oracle_encode(2, sr, kbps, x, x)
    for n in (1, 575, 576, 577, 1151, 1153, 1729):            # every flush shape around the 576 / 1152 boundaries
        y = (rng.normal(0, 3000, n)).astype(np.int16)
        assert _encode(1, kbps, y, None, n, sr) == oracle_encode(1, sr, kbps, y), n


def test_gpu_batch_streams_lsf(lib):
    import lamejs_amd, pcm
    from oracle_py import oracle_encode
    streams = [pcm.bursts(576 * (21 + (i % 9)) + 13 * i, 1, seed=2000 + i)[0] for i in range(24)]
    encs = [lamejs_amd.Mp3Encoder(1, 22050, 64) for _ in streams]
    got = lamejs_amd.encode_streams(encs, streams)
    for s, g in zip(streams, got):
        assert g == oracle_encode(1, 22050, 64, s)


def test_gpu_full_size_properties_lsf(lib):
    """1e5 MPEG-2 frames (22.05 kHz mono 64 kbps): frame walk by header (sync, version bit 0, padding -> length), one call ==
    two calls, prefix == oracle."""
    import pcm
    from oracle_py import oracle_encode
    nfr = 100000
    L, _ = pcm.sine(576 * nfr, 1)
    one = _encode(1, 64, L, None, 576 * nfr, 22050)
    two = _encode(1, 64, L, None, 576 * 61234 + 5, 22050)
    assert hashlib.md5(one).hexdigest() == hashlib.md5(two).hexdigest()
    base = 72000 * 64 // 22050
    pos = k = 0
    while pos < len(one):
        assert one[pos] == 0xFF and (one[pos + 1] & 0xFE) == 0xF2, f"lost sync at frame {k}"     # MPEG-2, layer III, no CRC
        pos += base + ((one[pos + 2] >> 1) & 1)
        k += 1
    assert pos == len(one) and k >= nfr + 1
    ref = oracle_encode(1, 22050, 64, L[: 576 * 3000], flush=False)
    assert one[: len(ref)] == ref


def test_gpu_batch_streams(lib):
    """BASELINE config 5 shape (scaled down): many independent mono streams in one launch."""
    import lamejs_amd, pcm
    from oracle_py import oracle_encode
    streams = [pcm.bursts(1152 * (20 + (i % 7)), 1, seed=1000 + i)[0] for i in range(24)]
    encs = [lamejs_amd.Mp3Encoder(1, 44100, 128) for _ in streams]
    got = lamejs_amd.encode_streams(encs, streams)
    for s, g in zip(streams, got):
        assert g == oracle_encode(1, 44100, 128, s)


def test_gpu_full_size_properties(lib):
    """BASELINE full size (1e5 frames, mono 128k): size-independent properties + spot parity.
    - every frame starts with the sync word, has the right length (417/418) and the closed-form padding pattern
    - encoding in one call == encoding in two calls (stream-state carry at scale)
    - the first 3000 frames equal the oracle's (the prefix of a stream does not depend on what follows)"""
    import lamejs_amd, pcm
    from oracle_py import oracle_encode
    nfr = 100000
    L, _ = pcm.sine(1152 * nfr, 1)
    one = _encode(1, 128, L, None, 1152 * nfr)
    two = _encode(1, 128, L, None, 1152 * 61234)
    assert hashlib.md5(one).hexdigest() == hashlib.md5(two).hexdigest()
    pos, k, lag = 0, 0, 42300
    while pos < len(one):
        assert one[pos] == 0xFF and (one[pos + 1] & 0xFE) == 0xFA, f"lost sync at frame {k}"
        pad = (one[pos + 2] >> 1) & 1
        lag -= 42300
        exp = 0
        if lag < 0:
            lag += 44100
            exp = 1
        assert pad == exp
        pos += 417 + pad
        k += 1
    assert k == nfr + 1 and pos == len(one)
    ref = oracle_encode(1, 44100, 128, L[: 1152 * 3000], flush=False)
    assert one[: len(ref)] == ref


@pytest.mark.gpu
@pytest.mark.parametrize("sr,kbps", [(44100, 128), (22050, 64), (8000, 24)])
def test_gpu_seed_repair_path(lib, sr, kbps):
    """Poor speculative seed -> frames flagged by the validation kernel -> repair passes -> reference bytes."""
    import lamejs_amd, pcm
    from oracle_py import oracle_encode
    L, R = pcm.bursts(1152 * 40, 2, seed=78)
    want = oracle_encode(2, sr, kbps, L, R)
    lib.lhip_debug_set_spec_seed.argtypes = [ctypes.c_int, ctypes.c_int]
    try:
        assert lib.lhip_debug_set_spec_seed(255, 1) == 0
        enc = lamejs_amd.Mp3Encoder(2, sr, kbps)
        got = enc.encodeBuffer(L, R)
        stats = enc.last_batch_stats()
        got += enc.flush()
        assert stats["repaired_frames"] > 0, stats
        assert got == want
    finally:
        lib.lhip_debug_set_spec_seed(180, 4)


@pytest.mark.gpu
def test_gpu_random_material(lib):
    """Seeded random material vs the oracle (see tests/tools/fuzz_gpu.py): sparse / quiet frames, clicks, level steps."""
    import sys
    sys.path.insert(0, str(ROOT / "tests" / "tools"))
    import fuzz_gpu
    assert fuzz_gpu.run(84, 2024, verbose=False) == []
    assert fuzz_gpu.run(56, 7, verbose=False) == []
    assert fuzz_gpu.run(96, 31, verbose=False, cfgs=fuzz_gpu.LSF_CFGS) == []               # MPEG-2 / 2.5
    assert fuzz_gpu.run(70, 5, verbose=False, cfgs=fuzz_gpu.RESAMPLE_CFGS) == []           # integer-ratio resampling in front


@pytest.mark.gpu
@pytest.mark.parametrize("kbps", [128, 320])
def test_gpu_stereo_both_quant_paths(lib, kbps):
    """Stereo batches up to 6 x CUs frames take the two-waves-per-frame latency kernel (g_quant_pair), larger ones the persistent
    kernel: the same PCM through one 2600-frame call (persistent), through 300-frame calls (pair) and through the oracle."""
    import lamejs_amd
    import pcm
    from oracle_py import oracle_encode
    n = 1152 * 2600
    L, R = pcm.bursts(n, 2)
    L2, R2 = pcm.sine(n, 2, seed=4242)
    L = (L // 2 + L2 // 2).astype(np.int16); R = (R // 2 + R2 // 2).astype(np.int16)
    want = oracle_encode(2, 44100, kbps, L, R)
    big = _encode(2, kbps, L, R, n)
    small = _encode(2, kbps, L, R, 1152 * 300)
    assert big == want, "persistent kernel differs from the oracle: " + _first_diff(big, want)
    assert small == want, "pair kernel differs from the oracle: " + _first_diff(small, want)


def test_gpu_largest_frames_dense_noise(lib):
    """Full-scale noise at the configurations with the largest frames (1440 bytes): every byte of the frame carries Huffman data."""
    import sys
    sys.path.insert(0, str(ROOT / "tests" / "tools"))
    import large_frames
    assert large_frames.run(nframes=40) == []
