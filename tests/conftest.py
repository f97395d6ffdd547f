import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())["cases"]


@pytest.fixture(scope="session")
def golden_joint():
    """Joint-stereo goldens: the reference's modules driven with gfp.mode = JOINT_STEREO (tests/tools/gen_golden_joint.js)."""
    return json.loads((ROOT / "tests" / "golden" / "golden_joint.json").read_text())["cases"]


@pytest.fixture(scope="session")
def golden_resv():
    """Bit-reservoir goldens: the reference's modules driven with gfp.disable_reservoir = false (tests/tools/gen_golden_resv.js)."""
    return json.loads((ROOT / "tests" / "golden" / "golden_resv.json").read_text())["cases"]


@pytest.fixture(scope="session")
def golden_wavfix():
    """The reference's remaining fixtures: testdata/Left.wav + Right.wav (48 kHz) and Stereo44100.wav, encoded by the
    unmodified reference (tests/tools/gen_golden_wavfix.js)."""
    return json.loads((ROOT / "tests" / "golden" / "golden_wavfix.json").read_text())["cases"]


def load_case_pcm(case):
    """PCM of a golden case: committed excerpt for the reference's fixtures, regenerated for synthetic corpora."""
    import hashlib
    import pcm

    ch, n = case["channels"], case["nsamples"]
    if case["corpus"] in ("wavexcerpt", "wavfull"):     # the reference's own fixtures (testdata/Left44100.wav, Right44100.wav) as raw s16le
        L = np.fromfile(ROOT / "tests" / "golden" / "left44100_full.s16", dtype="<i2")[:n]
        R = np.fromfile(ROOT / "tests" / "golden" / "right44100_full.s16", dtype="<i2")[:n] if ch == 2 else None
    elif case["corpus"] == "wavstereo44100":               # testdata/Stereo44100.wav is Left44100.wav / Right44100.wav interleaved (checked by the generator)
        L = np.fromfile(ROOT / "tests" / "golden" / "left44100_full.s16", dtype="<i2")[:n]
        R = np.fromfile(ROOT / "tests" / "golden" / "right44100_full.s16", dtype="<i2")[:n]
    elif case["corpus"] == "wav48000":                     # testdata/Left.wav, Right.wav
        L = np.fromfile(ROOT / "tests" / "golden" / "left48000_full.s16", dtype="<i2")[:n]
        R = np.fromfile(ROOT / "tests" / "golden" / "right48000_full.s16", dtype="<i2")[:n] if ch == 2 else None
    else:
        L, R = pcm.CORPORA[case["corpus"]](n, ch)
    h = hashlib.md5()
    h.update(L.tobytes())
    if R is not None:
        h.update(R.tobytes())
    assert h.hexdigest() == case["pcm_md5"], "PCM generator drifted from the fixture generator"
    return L, R
