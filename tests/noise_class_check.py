"""TEST helper (CPU tier: host simulation; GPU tier: the device): calc_noise's logarithm-free band classification
(lamejs_amd/csrc/lhip_math.h noise_class) must agree with the class the reference derives from log10 -- from the f64 value for a
band evaluated in the call and from the Float32 copy for a cached band -- wherever it does not ask for the logarithm (-1)."""
import ctypes

import numpy as np


def noise_class_cases(n=400000, seed=5):
    rng = np.random.default_rng(seed)
    xs = [10.0 ** rng.uniform(-25, 39, n), 1.0 + rng.uniform(-1e-3, 1e-3, n // 4), 10.0 ** rng.uniform(-0.2, 6, n)]
    k = np.arange(0, 420, dtype=np.float64)
    steps = 10.0 ** ((k - 0.5) / 10.0)                                   # where tmp changes
    for eps in (0.0, 1e-16, 3e-16, 1e-15, 1e-12, 1e-9, 1e-7, 6e-7, 2e-6, 5.5e-6, 1e-5, 3e-5, 4.6e-5, 6e-5, 1e-4, 3e-4, 1e-3):
        xs += [steps * (1 + eps), steps * (1 - eps)]
    one = np.array([1.0])
    for _ in range(6):
        xs.append(one.copy()); one = np.nextafter(one, 2.0)
    one = np.array([1.0])
    for _ in range(6):
        one = np.nextafter(one, 0.0); xs.append(one.copy())
    xs.append(np.array([0.0, -1.0, np.nan, np.inf, 1e-300, 5e-324, 1e37, 9.99e36, 1.0001e37, 1e38, 3e38, 1e39, 1e300]))
    return np.concatenate(xs)


def check_noise_class(lib):
    lib.lhip_debug_math.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    x = np.ascontiguousarray(noise_class_cases(), dtype=np.float64)
    out = np.empty_like(x)
    assert lib.lhip_debug_math(9, x.ctypes.data, out.ctypes.data, len(x)) == 0
    v = np.rint(out).astype(np.int64)
    cf = v // 1000000
    rest = v - cf * 1000000
    cd = (rest + 1) // 1000                       # fast class is -1, 0 .. 401: make the split robust to the -1
    fast = rest - cd * 1000
    took = fast >= 0
    bad = took & ((fast != cd) | (fast != cf))
    assert not bad.any(), (x[bad][:5], fast[bad][:5], cd[bad][:5], cf[bad][:5])
    # the shortcut must actually be taken almost everywhere in the operating range
    mid = (x > 1.0) & (x < 1e30)
    assert took[mid].mean() > 0.9, took[mid].mean()
    return int(took.sum()), len(x)


def check_div_by_f32(lib, n=600000, seed=11):
    """calc_noise divides every band's noise by xmin (a Float32) through xmin's reciprocal (lhip_math.h div_by_f32: a multiply and two fma);
    the result must be the division's, bit for bit: random operands over the whole range the path can see, divisors down to Float32
    subnormals, quotients next to powers of two, noise = 0."""
    lib.lhip_debug_math.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.default_rng(seed)
    a = np.concatenate([10.0 ** rng.uniform(-180, 25, n), rng.uniform(0, 4, n // 4), np.zeros(16), 2.0 ** rng.integers(-300, 60, n // 8).astype(np.float64)])
    b = np.concatenate([10.0 ** rng.uniform(-44.9, 38, n), rng.uniform(0.5, 2, n // 4).astype(np.float32).astype(np.float64), 10.0 ** rng.uniform(-10, 10, 16),
                        (1.0 + rng.integers(0, 1 << 23, n // 8) / float(1 << 23))])
    # quotients that land next to a power of two (where the rounding grid changes): a = b32 * 2^k * (1 +- tiny)
    b32 = b[: n // 8].astype(np.float32).astype(np.float64)
    a = np.concatenate([a, b32 * 2.0 ** rng.integers(-40, 40, n // 8) * (1 + rng.choice([-3, -2, -1, 0, 1, 2, 3], n // 8) * 2.0 ** -52)])
    b = np.concatenate([b, b32])
    rec = np.ascontiguousarray(np.stack([a, b], axis=1).reshape(-1), dtype=np.float64)
    out = np.empty_like(rec)
    assert lib.lhip_debug_math(10, rec.ctypes.data, out.ctypes.data, len(rec)) == 0
    o = out.reshape(-1, 2)
    bad = np.nonzero(o[:, 0].view(np.uint64) != o[:, 1].view(np.uint64))[0]
    assert bad.size == 0, (a[bad][:5], b[bad][:5], o[bad][:5])
    return len(a)


def check_ma_index(lib, n=400000, seed=13):
    """mask_add's table index ToInt32(log10(ratio) * 16) without the logarithm (k_psy.h ma_index16, v_log_f32 behind it on the device): wherever
    the shortcut answers it must be the logarithm's index -- random ratios over the range the psychoacoustic model produces and every step
    10^(k / 16) approached from both sides down to one ulp."""
    lib.lhip_debug_math.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.default_rng(seed)
    xs = [10.0 ** rng.uniform(0, 2, n), 1.0 + rng.uniform(0, 1e-3, n // 4), 10.0 ** rng.uniform(0, 6, n // 4)]
    k = np.arange(0, 97, dtype=np.float64)
    steps = 10.0 ** (k / 16.0)
    for eps in (0.0, 1e-16, 3e-16, 1e-15, 1e-12, 1e-9, 1e-7, 6e-7, 2e-6, 5.5e-6, 1e-5, 3e-5, 4.6e-5, 6e-5, 1e-4, 3e-4, 1e-3):
        xs += [steps * (1 + eps), np.maximum(steps * (1 - eps), 1.0)]
    one = np.array([1.0])
    for _ in range(6):
        xs.append(one.copy()); one = np.nextafter(one, 2.0)
    x = np.ascontiguousarray(np.concatenate(xs), dtype=np.float64)
    out = np.empty_like(x)
    assert lib.lhip_debug_math(11, x.ctypes.data, out.ctypes.data, len(x)) == 0
    v = np.rint(out).astype(np.int64)
    exact = (v + 1) // 1000
    fast = v - exact * 1000
    took = fast >= 0
    bad = took & (fast != exact)
    assert not bad.any(), (x[bad][:5], fast[bad][:5], exact[bad][:5])
    assert took.mean() > 0.9, took.mean()
    return int(took.sum()), len(x)
