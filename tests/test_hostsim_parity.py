"""Kernel *logic* check without a GPU: the kernel bodies of lamejs_amd/csrc compiled for the host with
one lane (tests/hostsim, test-only) must produce the reference's bytes through the same C ABI."""
import ctypes
import hashlib
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_case_pcm
from oracle_py import oracle_encode


@pytest.fixture(scope="module")
def sim():
    import lamejs_amd
    subprocess.run(["make", "-C", str(ROOT / "tests" / "hostsim"), "all"], check=True, capture_output=True)
    lib = lamejs_amd.load_library(ROOT / "tests" / "hostsim" / "_build" / "liblamejs_hostsim.so")
    assert b"HOST SIMULATION" in lib.lhip_version()
    return lib


def _encode(lib, ch, kbps, L, R, chunk, sr=44100, joint=False, reservoir=False):
    import lamejs_amd
    enc = lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib, joint=joint, reservoir=reservoir)
    out = b""
    for p in range(0, len(L), chunk):
        out += enc.encodeBuffer(L[p:p + chunk], None if R is None else R[p:p + chunk])
    out += enc.flush()
    assert enc.flush() == b""          # second flush returns nothing (Lame.js:1397-1399)
    enc.close()
    return out


def test_hostsim_matches_goldens(sim, golden):
    n = 0
    for case in golden:
        if case["nsamples"] > 1152 * 300:
            continue
        if case.get("outside_envelope"):
            continue
        L, R = load_case_pcm(case)
        mp3 = _encode(sim, case["channels"], case["kbps"], L, R, case["chunk"], case.get("samplerate", 44100))
        assert hashlib.md5(mp3).hexdigest() == case["mp3_md5"], case
        n += 1
    assert n >= 53


def test_hostsim_matches_reference_wav_fixtures_48k_and_stereo(sim, golden_wavfix):
    """testdata/Left.wav + Right.wav (48 kHz) and Stereo44100.wav, reference output vs the kernel logic."""
    for case in golden_wavfix:
        L, R = load_case_pcm(case)
        mp3 = _encode(sim, case["channels"], case["kbps"], L, R, case["chunk"], case["samplerate"])
        assert len(mp3) == case["mp3_len"] and hashlib.md5(mp3).hexdigest() == case["mp3_md5"], case


def test_hostsim_matches_joint_stereo_goldens(sim, golden_joint):
    """SURVEY.md 8f #3 (extension): the kernel logic in joint-stereo mode -- four psy channels, the per-frame M/S decision, mid/side
    quantization -- against the reference's own joint-stereo output (L/R-only, M/S-only and mixed streams; MPEG-1, LSF, resampling)."""
    n = ms = 0
    for case in golden_joint:
        if case["nsamples"] > 1152 * 300:
            continue
        L, R = load_case_pcm(case)
        mp3 = _encode(sim, 2, case["kbps"], L, R, case["chunk"], case.get("samplerate", 44100), joint=True)
        assert hashlib.md5(mp3).hexdigest() == case["mp3_md5"], case
        n += 1
        ms += case["ms_frames"]
    assert n >= 18 and ms >= 2000


def test_hostsim_matches_bit_reservoir_goldens(sim, golden_resv):
    """SURVEY.md 8f #4 (extension): the kernel logic with the bit reservoir in use -- frame-at-a-time launches, the reservoir-dependent
    pre-echo control, on_pe, ResvFrameEnd, main_data_begin and the continuous stream with lazily inserted headers, the stream flush --
    against the reference's own reservoir output (mono / stereo / joint stereo, MPEG-1 / 2 / 2.5, resampling, many-call chunking)."""
    n = 0
    for case in golden_resv:
        if case["nsamples"] > 1152 * 300:
            continue
        L, R = load_case_pcm(case)
        mp3 = _encode(sim, case["channels"], case["kbps"], L, R, case["chunk"], case.get("samplerate", 44100), joint=bool(case.get("joint")), reservoir=True)
        assert hashlib.md5(mp3).hexdigest() == case["mp3_md5"], case
        n += 1
    assert n >= 25


def test_hostsim_bit_reservoir_stream_batch(sim):
    """Many streams side by side with the reservoir in use (the only parallelism that mode has) == every stream on its own."""
    import lamejs_amd, pcm
    streams = [pcm.bursts(1152 * (5 + 2 * i) + 100 * i, 1, seed=3000 + i)[0] for i in range(5)]
    encs = [lamejs_amd.Mp3Encoder(1, 44100, 128, lib=sim, reservoir=True) for _ in streams]
    got = lamejs_amd.encode_streams(encs, streams)
    for s_, g in zip(streams, got):
        assert g == oracle_encode(1, 44100, 128, s_, reservoir=True)


def test_hostsim_batch_streams_match_single(sim):
    import lamejs_amd, pcm
    streams = [pcm.bursts(1152 * (5 + i), 1, seed=1000 + i)[0] for i in range(4)]
    encs = [lamejs_amd.Mp3Encoder(1, 44100, 128, lib=sim) for _ in streams]
    got = lamejs_amd.encode_streams(encs, streams)
    for s, g in zip(streams, got):
        assert g == oracle_encode(1, 44100, 128, s)


@pytest.mark.parametrize("sr,kbps", [(44100, 128), (22050, 64), (8000, 24)])
def test_hostsim_seed_repair_path(sim, sr, kbps):
    """A deliberately poor speculative bin-search seed makes the validation flag frames; the repair passes
    must converge to the reference's bytes (the chain-implied seeds), whatever was speculated."""
    import lamejs_amd, pcm
    L, R = pcm.bursts(1152 * 12, 2, seed=77)
    want = oracle_encode(2, sr, kbps, L, R)
    sim.lhip_debug_set_spec_seed.argtypes = [ctypes.c_int, ctypes.c_int]
    try:
        assert sim.lhip_debug_set_spec_seed(255, 1) == 0
        enc = lamejs_amd.Mp3Encoder(2, sr, kbps, lib=sim)
        got = enc.encodeBuffer(L, R)
        stats = enc.last_batch_stats()
        got += enc.flush()
        assert stats["repaired_frames"] > 0 and stats["repair_iterations"] > 0, stats
        assert got == want
    finally:
        sim.lhip_debug_set_spec_seed(180, 4)


def test_hostsim_interleaved_live_encoders(sim):
    """Nine encoders of different configurations (MPEG-1 / MPEG-2 / resampled, joint stereo, bit reservoir) alive at once and called alternately a frame's
    worth at a time, with calls of 0 and ~3000 samples in between (tests/interleaved.py; the GPU tier runs the same against the device)."""
    import interleaved
    assert interleaved.run(sim, 606061) == []


def test_hostsim_random_material(sim):
    """Seeded random material (tones, coloured noise, clicks, silence gaps, level steps) at every supported rate:
    this is the sweep that exposed the path-dependent table_select leftovers of the bin search (sparse frames)."""
    import sys
    sys.path.insert(0, str(ROOT / "tests" / "tools"))
    import fuzz_gpu
    assert fuzz_gpu.run(42, 2024, lib=sim, verbose=False) == []
    assert fuzz_gpu.run(48, 31, lib=sim, verbose=False, cfgs=fuzz_gpu.LSF_CFGS) == []      # MPEG-2 / 2.5
    assert fuzz_gpu.run(42, 5, lib=sim, verbose=False, cfgs=fuzz_gpu.RESAMPLE_CFGS) == []  # integer-ratio resampling in front
    # the lowest bit budgets (8-32 kbps, MPEG-2 / 2.5): joint stereo's reduce_side takes its "side keeps 125 bits" branch only here
    # (found with tools/gcov_hostsim.sh; the oracle is pinned on this family by tests/tools/fuzz_ref.py ... lowrate)
    assert fuzz_gpu.run(16, 9301, lib=sim, verbose=False, cfgs=fuzz_gpu.LOWRATE_CFGS, joint=True) == []
    assert fuzz_gpu.run(16, 9302, lib=sim, verbose=False, cfgs=fuzz_gpu.LOWRATE_CFGS, joint=True, reservoir=True) == []
    assert fuzz_gpu.run(16, 9303, lib=sim, verbose=False, cfgs=fuzz_gpu.LOWRATE_CFGS) == []


def test_hostsim_asan_largest_frames():
    """The kernel bodies under AddressSanitizer on dense noise at the configurations with the largest frames (1440 bytes at
    32 kHz / 320 kbps): the bit-packing buffer once held 1088 bytes (an out-of-bounds LDS write on the device, found by review,
    invisible to sine goldens whose main data stays short).  Also byte-compared with the oracle."""
    import os, shutil, sys
    gcc = shutil.which("gcc")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip() if gcc else ""
    if not asan or not os.path.isabs(asan):
        pytest.skip("libasan not available")
    subprocess.run(["make", "-C", str(ROOT / "tests" / "hostsim"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "tools" / "large_frames.py"), str(ROOT / "tests" / "hostsim" / "_build" / "liblamejs_hostsim_asan.so")],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    # the extension modes (joint stereo, bit reservoir) and the one-frame-per-stream frame program under the same sanitizer build
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "tools" / "asan_modes.py"), str(ROOT / "tests" / "hostsim" / "_build" / "liblamejs_hostsim_asan.so")],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_hostsim_ubsan_modes():
    """The kernel bodies under UndefinedBehaviorSanitizer (any report aborts the run): the extension modes, the frame program and
    the sharding calls of tests/tools/asan_modes.py.  (It found a left shift of a negative scalefactor in inc_subblock_gain.)"""
    import os, shutil, sys
    gcc = shutil.which("gcc")
    ub = subprocess.run([gcc, "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip() if gcc else ""
    if not ub or not os.path.isabs(ub):
        pytest.skip("libubsan not available")
    subprocess.run(["make", "-C", str(ROOT / "tests" / "hostsim"), "ubsan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=ub, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "tools" / "asan_modes.py"), str(ROOT / "tests" / "hostsim" / "_build" / "liblamejs_hostsim_ubsan.so")],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "OK" in r.stdout and "runtime error" not in r.stderr, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("ch,sr,kbps,nfr", [(2, 44100, 128, 40), (1, 22050, 64, 40)])
def test_hostsim_stage_taps(sim, ch, sr, kbps, nfr):
    """Stage-level differential check of the kernel logic (host simulation) against the oracle's taps: MDCT output, block types,
    masking ratios, ATH.adjust and the quantizer's gain / lengths per granule -- see tests/stage_taps.py (the GPU tier runs the
    same comparison on the device)."""
    import pcm, stage_taps
    L, R = pcm.bursts(1152 * nfr, ch, seed=91)
    assert stage_taps.compare_stages(sim, ch, sr, kbps, L, R) == []


@pytest.mark.parametrize("corpus,sr,kbps,nfr", [("bursts", 44100, 128, 40), ("centre_bursts", 22050, 64, 40)])
def test_hostsim_stage_taps_joint_stereo(sim, corpus, sr, kbps, nfr):
    """The same per-stage comparison in joint-stereo mode: additionally the frame's M/S decision, and the maskings handed to the
    quantizer are the mid / side ones in M/S frames (`bursts` mixes M/S and L/R frames)."""
    import pcm, stage_taps
    L, R = pcm.CORPORA[corpus](1152 * nfr, 2)
    assert stage_taps.compare_stages(sim, 2, sr, kbps, L, R, joint=True) == []


def _encode_in_ranges(lib, ch, sr, kbps, L, R, cuts, H, joint=False):
    """One stream cut at the frame numbers `cuts`; every piece on its own encoder: seek + H warm-up frames, state verified against
    the state the previous piece ended in, transplanted on a miss.  Returns (bytes, cuts whose speculated state missed)."""
    import lamejs_amd
    fs = 1152 if sr >= 32000 else 576
    bounds = [0] + [c * fs for c in cuts] + [len(L)]
    outs, prev, missed = [], None, []
    for r in range(len(bounds) - 1):
        a, b = bounds[r], bounds[r + 1]
        enc = lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib, joint=joint)
        if r > 0:
            p0, nt = a - H * fs, enc.seek_tail_samples()
            enc.seek(p0, L[p0 - nt:p0], None if R is None else R[p0 - nt:p0])
            enc.encodeBuffer(L[p0:a], None if R is None else R[p0:a])          # warm-up frames, output discarded
            got_state = enc.state_get()
            if got_state != prev:
                from state_fields import describe_diff
                missed.append((r, describe_diff(got_state, prev)))
                enc.state_set(prev)
                assert enc.state_get() == prev
        out = enc.encodeBuffer(L[a:b], None if R is None else R[a:b])
        prev = enc.state_get()
        if r == len(bounds) - 2:
            out += enc.flush()
        outs.append(out)
        enc.close()
    return b"".join(outs), missed


@pytest.mark.parametrize("corpus,ch,sr,kbps,nfr,cuts,H,joint", [
    ("sine", 2, 44100, 128, 60, [20, 40], 8, False),        # loud steady material: every speculated cut verifies
    ("bursts", 2, 44100, 128, 90, [30, 61], 8, False),      # silence gaps: the ATH adjustment has not converged at one cut -> transplant
    ("bursts", 1, 22050, 64, 80, [25, 50], 10, False),      # MPEG-2 (576-sample frames), no padding accumulator
    ("bursts", 2, 44100, 128, 90, [30, 61], 2, False),      # a warm-up far too short: both cuts miss, the transplant still gives the bytes
    ("centre_bursts", 2, 44100, 128, 70, [33], 8, True),    # joint stereo (four psy channels in the state)
    ("sine", 1, 44100, 128, 30, [3, 29], 1, False),         # the earliest position a stream can be put at (2 frames) and a cut one frame before the end
    ("sine", 2, 48000, 192, 40, [11, 12, 13], 4, False),    # ranges of a single frame
])
def test_hostsim_frame_range_shards(sim, corpus, ch, sr, kbps, nfr, cuts, H, joint):
    """SURVEY.md 8e, second mode (extension): ONE stream encoded as frame ranges on separate encoders == the stream encoded in one
    piece (the oracle's bytes), whether the speculated state at a cut verifies or has to be transplanted."""
    import pcm
    L, R = pcm.CORPORA[corpus](1152 * nfr, ch)
    got, missed = _encode_in_ranges(sim, ch, sr, kbps, L, R, cuts, H, joint)
    assert got == oracle_encode(ch, sr, kbps, L, R, joint=joint)
    if corpus == "sine" and H >= 8:
        assert missed == []
    if H == 2:
        assert [m[0] for m in missed] == [1, 2]


def test_hostsim_seek_and_state_errors(sim):
    import lamejs_amd, pcm
    L, R = pcm.sine(1152 * 6, 2)
    enc = lamejs_amd.Mp3Encoder(2, 44100, 128, lib=sim)
    nt = enc.seek_tail_samples()
    assert nt == 528 + 1152                               # the samples the encoder holds back + one frame
    with pytest.raises(lamejs_amd.LhipError):
        enc.seek(1152 * 2 + 1, L[:nt], R[:nt])             # not a frame boundary
    with pytest.raises(lamejs_amd.LhipError):
        enc.seek(1152, L[:nt], R[:nt])                     # too close to the start
    enc.encodeBuffer(L, R)
    with pytest.raises(lamejs_amd.LhipError):
        enc.seek(1152 * 3, L[:nt], R[:nt])                 # only a fresh stream can be moved
    st = enc.state_get()
    with pytest.raises(lamejs_amd.LhipError):
        enc.state_set(st[:100])
    with pytest.raises(lamejs_amd.LhipError):
        enc.state_set(b"\0" * len(st))
    wrong = lamejs_amd.Mp3Encoder(2, 44100, 160, lib=sim)   # same layout, another configuration: refused, not silently accepted
    with pytest.raises(lamejs_amd.LhipError):
        wrong.state_set(st)
    wrong.close()
    other = lamejs_amd.Mp3Encoder(2, 44100, 128, lib=sim)
    other.state_set(st)                                     # a clone continues exactly like the original
    assert other.encodeBuffer(L, R) + other.flush() == enc.encodeBuffer(L, R) + enc.flush()
    # a clone taken in the middle of a bit-reservoir stream (the reservoir fill, the header queue and the entropy history travel too)
    M = pcm.bursts(1152 * 9, 1)[0]
    ra = lamejs_amd.Mp3Encoder(1, 44100, 128, lib=sim, reservoir=True)
    head = ra.encodeBuffer(M[:1152 * 5])
    rb = lamejs_amd.Mp3Encoder(1, 44100, 128, lib=sim, reservoir=True)
    rb.state_set(ra.state_get())
    assert head + rb.encodeBuffer(M[1152 * 5:]) + rb.flush() == oracle_encode(1, 44100, 128, M, reservoir=True)
    ra.close(); rb.close()
    for e in (lamejs_amd.Mp3Encoder(1, 44100, 32, lib=sim), lamejs_amd.Mp3Encoder(1, 44100, 128, lib=sim, reservoir=True)):
        with pytest.raises(lamejs_amd.LhipError):           # resampling in front / bit reservoir: no seek
            e.seek(1152 * 4, L[:e.seek_tail_samples()])
        e.close()
    enc.close(); other.close()


def test_hostsim_boundary_error_paths(sim):
    """-1 (output buffer too small) on lhip_encode / lhip_flush / batch entries and the stream state afterwards, -3 on destroyed and
    garbage handles, the flush batch with an already flushed stream (tests/boundary_checks.py; the GPU tier runs the same)."""
    from boundary_checks import run_boundary_checks
    run_boundary_checks(sim, oracle_encode)


def test_hostsim_state_blob_is_canonical(sim):
    from boundary_checks import run_state_canonical_checks
    run_state_canonical_checks(sim)


def test_hostsim_noise_class_shortcut(sim):
    """calc_noise without the logarithm (lhip_math.h noise_class): equals the class from log10 (f64 and Float32 copy) wherever taken."""
    from noise_class_check import check_noise_class
    took, n = check_noise_class(sim)
    assert took > 0.5 * n


def test_hostsim_division_by_reciprocal_is_the_division(sim):
    """calc_noise's division by xmin through its reciprocal (lhip_math.h div_by_f32) against the division itself: bit-identical."""
    from noise_class_check import check_div_by_f32
    assert check_div_by_f32(sim) > 500000


def test_hostsim_mask_add_index_shortcut(sim):
    """mask_add's table index without the logarithm (k_psy.h ma_index16): the logarithm's index wherever it answers."""
    from noise_class_check import check_ma_index
    took, n = check_ma_index(sim)
    assert took > 0.9 * n


def test_hostsim_two_devices_tsan_asan(tmp_path):
    """Two devices inside one process (the GPU tier's two-device test skips on a one-GPU box): the host simulation pretends to have two
    (LHIP_HOSTSIM_DEVICES=2) and tests/tools/two_devices.c checks lhip_set_devices' round-robin placement, lhip_stream_device, a refused
    device outside the mask, and two host threads -- one per device context -- encoding at the same time, every second repetition through
    the chunked host path.  Built with ThreadSanitizer (any report fails the run) and AddressSanitizer; the bytes against the oracle."""
    import os, shutil
    import pcm
    gcc = shutil.which("gcc")
    tsan = subprocess.run([gcc, "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip() if gcc else ""
    if not tsan or not os.path.isabs(tsan):
        pytest.skip("libtsan not available")
    subprocess.run(["make", "-C", str(ROOT / "tests" / "hostsim"), "tsan"], check=True, capture_output=True)
    import lamejs_amd
    blob = tmp_path / "t.bin"
    blob.write_bytes(lamejs_amd.tables_blob(2, 44100, 128))
    mats = []
    for i in range(2):
        L, R = pcm.bursts(1152 * 120, 2, seed=900 + i)
        np.stack([L, R], axis=1).astype("<i2").tofile(tmp_path / f"pcm{i}.s16")
        mats.append((L, R))
    env = dict(os.environ, LHIP_HOSTSIM_DEVICES="2", LAMEJS_HIP_HOST_CHUNK_FRAMES="16,48", TSAN_OPTIONS="halt_on_error=1:exitcode=66", ASAN_OPTIONS="detect_leaks=0")
    for build in ("tsan", "asan"):
        exe = ROOT / "tests" / "hostsim" / "_build" / f"two_devices_{build}"
        r = subprocess.run([str(exe), str(blob), str(tmp_path / "pcm0.s16"), str(tmp_path / "pcm1.s16"), str(tmp_path / "o0.mp3"), str(tmp_path / "o1.mp3")],
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "OK two devices" in r.stdout and "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr, (build, r.stdout[-500:], r.stderr[-3000:])
        for i, (L, R) in enumerate(mats):
            assert (tmp_path / f"o{i}.mp3").read_bytes() == oracle_encode(2, 44100, 128, L, R), (build, i)


def test_hostsim_host_call_in_chunks_random_material(sim):
    """The chunked host path of lhip_encode (long Int16Arrays: chunks copied in, encoded and copied out side by side) on the material
    that upsets the seed chain, in the CPU tier: small chunks forced through LAMEJS_HIP_HOST_CHUNK_FRAMES (first, cap, growth -- so the
    doubling, the cap and the merged short remainder all occur), repairs forced by a poor speculation seed, a second ordinary call and
    flush() after the chunked one, two channels / joint stereo / mono.  lhip_last_batch_stats must cover the whole call.
    And a failure injected in chunk 2 must leave the stream exactly where the call found it (state blob and the bytes that follow)."""
    import os
    import lamejs_amd
    import pcm
    old = os.environ.get("LAMEJS_HIP_HOST_CHUNK_FRAMES")
    # the schedule is read once per process: this test needs its own (a subprocess would also do; the variable is read at the first long call)
    r = subprocess.run([__import__("sys").executable, "-c", """
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, lamejs_amd, pcm
from oracle_py import oracle_encode
sys.path.insert(0, %r)
import fuzz_gpu
lib = lamejs_amd.load_library(%r)
lib.lhip_debug_set_spec_seed(255, 1)                 # a poor seed: the validation flags frames inside the chunks
rng = np.random.default_rng(4242)
for ch, joint, nfr in ((2, False, 420), (2, True, 300), (1, False, 350)):
    L, R = fuzz_gpu.material(rng, 1152 * nfr + 517, ch)
    enc = lamejs_amd.Mp3Encoder(ch, 44100, 128, lib=lib, joint=joint)
    a = enc.encodeBuffer(L, R)
    st = enc.last_batch_stats()
    assert st["frames"] == nfr and st["repaired_frames"] > 0, st
    b = enc.encodeBuffer(L[:4000], None if R is None else R[:4000])
    c = enc.flush()
    LL = np.concatenate([L, L[:4000]]); RR = None if R is None else np.concatenate([R, R[:4000]])
    assert a + b + c == oracle_encode(ch, 44100, 128, LL, RR, joint=joint), (ch, joint)
lib.lhip_debug_set_spec_seed(180, 4)
L, R = pcm.bursts(1152 * 300, 2, seed=5)
enc = lamejs_amd.Mp3Encoder(2, 44100, 128, lib=lib)
pre = enc.encodeBuffer(L[:3000], R[:3000])
s0 = enc.state_get()
os.environ["LHIP_HOSTSIM_FAIL_CHUNK"] = "2"
try:
    enc.encodeBuffer(L[3000:], R[3000:])
    raise SystemExit("the injected failure did not surface")
except lamejs_amd.LhipError as e:
    assert "injected" in str(e)
os.environ["LHIP_HOSTSIM_FAIL_CHUNK"] = ""
assert enc.state_get() == s0
assert pre + enc.encodeBuffer(L[3000:], R[3000:]) + enc.flush() == oracle_encode(2, 44100, 128, L, R)
print("OK chunks")
""" % (str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "tools"), str(ROOT / "tests" / "hostsim" / "_build" / "liblamejs_hostsim.so"))],
                       capture_output=True, text=True, env=dict(os.environ, LAMEJS_HIP_HOST_CHUNK_FRAMES="16,96,2"), timeout=900)
    assert r.returncode == 0 and "OK chunks" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_hostsim_interleaved_async_batches(sim):
    """Asynchronous "device" batches (sync = 0; host pointers here, the simulation runs them at once) of two streams, interleaved, give the
    oracle's bytes."""
    import lamejs_amd
    import pcm
    if True:
        nfr = 12
        mats = [pcm.bursts(1152 * nfr * 2, 2, seed=70 + i) for i in range(2)]
        encs = [lamejs_amd.Mp3Encoder(2, 44100, 128, lib=sim) for _ in range(2)]
        got = [b"", b""]
        cap = (nfr + 4) * 420
        for half in range(2):
            for i in range(2):
                L, R = mats[i]
                l, r = np.ascontiguousarray(L[1152 * nfr * half: 1152 * nfr * (half + 1)]), np.ascontiguousarray(R[1152 * nfr * half: 1152 * nfr * (half + 1)])
                out = np.empty(cap, dtype=np.uint8)
                wr = (ctypes.c_int64 * 1)()
                rc = sim.lhip_encode_batch_device((ctypes.c_void_p * 1)(encs[i]._h), 1, (ctypes.c_void_p * 1)(l.ctypes.data), (ctypes.c_void_p * 1)(r.ctypes.data),
                                                  (ctypes.c_size_t * 1)(len(l)), (ctypes.c_void_p * 1)(out.ctypes.data), (ctypes.c_size_t * 1)(cap), wr, 0)
                assert rc == 0, sim.lhip_last_error()
                got[i] += out[: wr[0]].tobytes()
        for i in range(2):
            got[i] += encs[i].flush()
            encs[i].close()
            assert got[i] == oracle_encode(2, 44100, 128, *mats[i])


def test_hostsim_validation_counters_tap(sim):
    """lhip_debug_read(8): the seed-chain validation's counters of the last batch ([0] frames flagged by the memo-only pass, [1] frames it could not decide).
    With the speculation seed forced far from every chain seed the first pass must flag or leave undecided a good part of a batch; with the usual seed on a
    steady tone nothing is flagged -- and the bytes are the oracle's either way."""
    import lamejs_amd
    import pcm
    sim.lhip_debug_set_spec_seed.argtypes = [ctypes.c_int, ctypes.c_int]
    L, R = pcm.sine(1152 * 30, 2, seed=5)
    want = oracle_encode(2, 44100, 128, L, R)
    buf = (ctypes.c_int32 * 64)()
    try:
        for seed, expect_trouble in (((180, 4), False), ((255, 1), True)):
            sim.lhip_debug_set_spec_seed(*seed)
            enc = lamejs_amd.Mp3Encoder(2, 44100, 128, lib=sim)
            got = enc.encodeBuffer(L, R)
            assert sim.lhip_debug_read(8, buf, 256) == 256, sim.lhip_last_error()
            got += enc.flush()
            enc.close()
            assert got == want
            assert buf[0] >= 0 and buf[1] >= 0
            if not expect_trouble:
                assert buf[0] == 0
    finally:
        sim.lhip_debug_set_spec_seed(180, 4)


def test_experiment_patches_apply(tmp_path):
    """tools/experiments/*.patch are ideas waiting for GPU minutes, kept against HEAD: they must keep applying (a patch that no longer does is stale and goes)."""
    import shutil
    patches = sorted((ROOT / "tools" / "experiments").glob("*.patch"))
    if not patches or not shutil.which("patch"):
        pytest.skip("no experiment patches (or no patch tool)")
    for p in patches:
        dst = tmp_path / p.stem
        shutil.copytree(ROOT / "lamejs_amd" / "csrc", dst / "lamejs_amd" / "csrc")
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", str(p)], cwd=dst, capture_output=True, text=True)
        assert r.returncode == 0, (p.name, r.stdout[-800:], r.stderr[-400:])
