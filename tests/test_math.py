"""python_raytracer_amd/csrc/vrt_math.h (the sin / cos / pow the kernels and the oracle's portable mode share) against the
correctly rounded value computed with mpmath at 300 bits, in the argument ranges the trace path uses:
camera half-angles (|x| < 2 rad; lib.py:323-338 via init.py:41-43), (1 + bounces) ** (1 + falloff) (lib.py:450, 465).
The header claims the correctly rounded result; glibc's own functions are only "< 1 ulp" and are counted beside it."""
import ctypes as C

import mpmath
import numpy as np
import pytest

import oracle_lib as ol

mpmath.mp.prec = 300
N = 20000          # per function; tools/math_soak.py runs millions


def _lib():
    L = ol.lib()
    for f in (L.orc_sin, L.orc_cos):
        f.restype = C.c_double
        f.argtypes = [C.c_int, C.c_double]
    L.orc_pow.restype = C.c_double
    L.orc_pow.argtypes = [C.c_int, C.c_double, C.c_double]
    return L


def _correct(fn, *args):
    """Correctly rounded binary64 value of fn at exactly-representable arguments."""
    return float(fn(*[mpmath.mpf(a) for a in args]))     # mpf -> float rounds to nearest even from 300 bits


def test_sin_cos_are_correctly_rounded():
    L = _lib()
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.uniform(-2.0, 2.0, N), rng.uniform(-0.45, 0.45, N),      # half-angles of a 90-degree lens
                         rng.uniform(-1e-3, 1e-3, 500), [0.0, -0.0, 1e-300, np.pi / 4, np.pi / 2, -np.pi / 2, 2.0 ** -30]])
    bad = {"sin": 0, "cos": 0, "glibc_sin": 0, "glibc_cos": 0}
    for x in xs.tolist():
        s, c = _correct(mpmath.sin, x), _correct(mpmath.cos, x)
        bad["sin"] += L.orc_sin(ol.LIBM_PORTABLE, x) != s
        bad["cos"] += L.orc_cos(ol.LIBM_PORTABLE, x) != c
        bad["glibc_sin"] += L.orc_sin(ol.LIBM_GLIBC, x) != s
        bad["glibc_cos"] += L.orc_cos(ol.LIBM_GLIBC, x) != c
    assert bad["sin"] == 0 and bad["cos"] == 0, bad
    # the sign of a zero result follows the argument, like libm
    assert np.signbit(L.orc_sin(ol.LIBM_PORTABLE, -0.0)) and not np.signbit(L.orc_sin(ol.LIBM_PORTABLE, 0.0))


def test_joint_sin_cos_equals_the_single_functions():
    """vrt_sincos2 (the HIP ray generation's call: sin and cos of both lens half-angles in one straight-line block) returns
    bit for bit what vrt_sin / vrt_cos return -- inside its fast block's domain and outside it (zeros, |x| >= pi/4)."""
    L = _lib()
    L.orc_sincos2.restype = C.c_int
    L.orc_sincos2.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
    rng = np.random.default_rng(13)
    a = np.concatenate([rng.uniform(-0.78, 0.78, N), rng.uniform(-2.0, 2.0, N // 4), rng.uniform(-1e-6, 1e-6, 200),
                        [0.0, -0.0, 0.3, np.pi / 4, -np.pi / 4, 0.7853981633974482, 1e-300]])
    b = rng.permutation(a)
    out = (C.c_double * 4)()
    fast = 0
    for x, y in zip(a.tolist(), b.tolist()):
        fast += L.orc_sincos2(x, y, out)
        want = (L.orc_sin(ol.LIBM_PORTABLE, x), L.orc_cos(ol.LIBM_PORTABLE, x), L.orc_sin(ol.LIBM_PORTABLE, y), L.orc_cos(ol.LIBM_PORTABLE, y))
        assert np.array(out[:]).tobytes() == np.array(want).tobytes(), (x, y, out[:], want)
    assert fast > N // 2          # the fast block is what camera lenses use
    assert L.orc_sincos2(0.0, 0.3, out) == 0 and L.orc_sincos2(0.3, 1.0, out) == 0 and L.orc_sincos2(0.3, -0.5, out) == 1


def test_pow_is_correctly_rounded():
    L = _lib()
    rng = np.random.default_rng(12)
    # bases: 1 + sums of material absorptions (a few of 0.05 .. 7); exponents: 1 + falloff in [1, 3]
    xs = np.concatenate([1 + rng.uniform(0, 30, N), 1 + rng.integers(1, 200, N // 4) * 0.25, rng.uniform(0.01, 1.0, N // 4)])
    ys = np.concatenate([rng.uniform(1.0, 3.0, N), np.full(N // 4, 1.25), rng.uniform(0.1, 4.0, N // 4)])
    bad = glibc_bad = 0
    for x, y in zip(xs.tolist(), ys.tolist()):
        v = _correct(mpmath.power, x, y)
        bad += L.orc_pow(ol.LIBM_PORTABLE, x, y) != v
        glibc_bad += L.orc_pow(ol.LIBM_GLIBC, x, y) != v
    assert bad == 0, (bad, glibc_bad)
    assert L.orc_pow(ol.LIBM_PORTABLE, 1.0, 1.25) == 1.0 and L.orc_pow(ol.LIBM_PORTABLE, 4.0, 0.5) == 2.0


def test_known_answers_from_cpython():
    """tests/golden/kat_math.json: math.sin / math.cos / ** of the build container's CPython (glibc) on the reference's
    call pattern ([argument(s)..., value] as hex floats).  The portable functions agree wherever glibc itself is
    correctly rounded, and are the correctly rounded value where it is not."""
    import json
    import os
    kat = json.load(open(os.path.join(ol.GOLDEN, "kat_math.json")))
    L = _lib()
    n = same = 0
    for name, fn, ref in (("sin", lambda a: L.orc_sin(ol.LIBM_PORTABLE, a[0]), mpmath.sin),
                          ("cos", lambda a: L.orc_cos(ol.LIBM_PORTABLE, a[0]), mpmath.cos),
                          ("pow", lambda a: L.orc_pow(ol.LIBM_PORTABLE, a[0], a[1]), mpmath.power)):
        for rec in kat[name]:
            vals = [float.fromhex(v) for v in rec]
            args, want = vals[:-1], vals[-1]
            got = fn(args)
            n += 1
            if got == want:
                same += 1
            else:
                assert got == _correct(ref, *args), (name, args)
    assert n > 100 and same / n > 0.99, (n, same)
