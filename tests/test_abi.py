"""The C-ABI library must load without a GPU and export every symbol include/lamejs_hip.h declares."""
import ctypes
import re

import pytest

from conftest import ROOT


def test_hip_library_exports_declared_symbols():
    import lamejs_amd
    lib = lamejs_amd.load_library()
    hdr = (ROOT / "include" / "lamejs_hip.h").read_text()
    names = set(re.findall(r"\b(lhip_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"missing export {n}"
    assert b"HIP gfx950" in lib.lhip_version()


def test_no_cpu_fallback_without_device():
    """On a box without a HIP device creation must fail loudly (never silently fall back)."""
    import lamejs_amd
    lib = lamejs_amd.load_library()
    if lib.lhip_device_count() > 0:
        return
    try:
        lamejs_amd.Mp3Encoder(1, 44100, 128)
    except lamejs_amd.LhipError as e:
        assert "no HIP device" in str(e) or "failed" in str(e)
    else:
        raise AssertionError("encoder creation succeeded without a HIP device")


def test_product_does_not_reference_oracle():
    for p in list((ROOT / "lamejs_amd").rglob("*.h")) + list((ROOT / "lamejs_amd").rglob("*.cpp")) + list((ROOT / "lamejs_amd").rglob("*.py")) + list((ROOT / "lamejs_amd").rglob("*.js")):
        txt = p.read_text(errors="ignore")
        assert "oracle/" not in txt and "lame_oracle" not in txt and "oracle_py" not in txt, p


def test_configs_outside_the_envelope_fail_loudly(golden):
    """Configurations the reference would resample to an MPEG-2 rate are refused (no silent fallback)."""
    import lamejs_amd
    outside = [c for c in golden if c.get("outside_envelope")]
    assert len(outside) >= 3
    for c in outside:
        with pytest.raises(lamejs_amd.LhipError, match="resampl"):
            lamejs_amd.tables_blob(c["channels"], c.get("samplerate", 44100), c["kbps"])


def test_set_devices_argument_handling():
    """lhip_set_devices without a GPU: fails loudly (no device to allow); with one: rejects masks naming absent devices."""
    import ctypes
    import lamejs_amd
    lib = lamejs_amd.load_library()
    lib.lhip_set_devices.restype = ctypes.c_int
    lib.lhip_set_devices.argtypes = [ctypes.c_uint64]
    n = lib.lhip_device_count()
    if n <= 0:
        assert lib.lhip_set_devices(1) < 0 and b"no HIP device" in lib.lhip_last_error()
    else:
        assert lib.lhip_set_devices(1 << 63) < 0
        assert lib.lhip_set_devices(0) == n
