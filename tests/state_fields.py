"""TEST HELPER: names for the bytes of a lhip_state_get blob (40-byte host header + StreamState, lamejs_amd/csrc/lhip_layout.h), so that
a failed "the speculated state equals the true state" check can say WHICH carried field differed."""
import struct

HDR = 40
FIELDS = [(0, "pcm_tail"), (15232, "sb"), (19840, "E"), (21792, "ecb_s"), (24864, "peaks"), (25056, "loud"), (25064, "tot_ener"),
          (25080, "last_attack"), (25096, "tent"), (25104, "last_bt"), (25112, "ath_adjust"), (25120, "ath_limit"), (25128, "seed"),
          (25144, "rs_old"), (25400, "nb1"), (26424, "nb2"), (27448, "rv"), (28336, "END")]


def describe_diff(a: bytes, b: bytes) -> str:
    if len(a) != len(b):
        return f"sizes differ: {len(a)} vs {len(b)}"
    if len(a) != HDR + FIELDS[-1][0]:
        return f"blob size {len(a)} does not match tests/state_fields.py ({HDR + FIELDS[-1][0]}): update the offsets"
    out = []
    if a[:HDR] != b[:HDR]:
        out.append(f"header: {struct.unpack('<10i', a[:HDR])} vs {struct.unpack('<10i', b[:HDR])}")
    for (o, name), (o2, _) in zip(FIELDS[:-1], FIELDS[1:]):
        fa, fb = a[HDR + o:HDR + o2], b[HDR + o:HDR + o2]
        if fa != fb:
            idx = [i for i in range(0, len(fa), 4) if fa[i:i + 4] != fb[i:i + 4]]
            fmt = "<d" if name.startswith("ath_") else ("<i" if name in ("last_attack", "tent", "last_bt", "seed") else "<f")
            w = 8 if fmt == "<d" else 4
            i0 = idx[0] // w * w
            out.append(f"{name}: {len(idx)} words differ, first at word {i0 // w}: {struct.unpack(fmt, fa[i0:i0 + w])[0]!r} vs {struct.unpack(fmt, fb[i0:i0 + w])[0]!r}")
    return "; ".join(out) if out else "equal"
