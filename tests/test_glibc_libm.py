"""include/avp_glibc_libm.h (the atan2 / asin / acos / tan / pow(., 2) the device computes the Reeds-Shepp words with)
compiled for the host == this platform's glibc libm == CPython's math module, bit for bit. The 1e9-arguments-per-
distribution run of the same sweep is committed as profiles/r04_glibc_libm_sweep.txt (2.9e10 arguments, 0 mismatches)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

from conftest import ROOT


def args_for(kind, rng, n):
    """Arguments that reach every range of the five functions (shared with the -m gpu device-vs-host test)."""
    u = rng.uniform
    if kind == 0:      # atan2(y, x)
        y = np.concatenate([u(-10, 10, n), 2.0 * np.ones(n // 4), u(0, 8, n // 4), 10.0 ** u(-300, 300, n // 4) * rng.choice([-1, 1], n // 4),
                            np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.inf, np.nan, 1.0, 5e-324, 1e308])])
        x = np.concatenate([u(-10, 10, n), u(-8, 8, n // 4), -2.0 * np.ones(n // 4), 10.0 ** u(-300, 300, n // 4) * rng.choice([-1, 1], n // 4),
                            np.array([0.0, 0.0, -0.0, -1.0, 0.0, -0.0, np.inf, -np.inf, 1.0, 1.0, np.nan, 1e308, 5e-324])])
        return y, x
    if kind in (1, 2):  # asin / acos
        a = np.concatenate([u(-1, 1, n), 10.0 ** u(-20, 0, n // 4) * rng.choice([-1, 1], n // 4), 1.0 - 10.0 ** u(-17, -1, n // 4),
                            -1.0 + 10.0 ** u(-17, -1, n // 4), np.array([0.0, -0.0, 1.0, -1.0, 0.5, 0.125, 0.75, 0.96875, 1.0000001, np.inf, np.nan])])
        return a, None
    if kind == 3:       # tan
        a = np.concatenate([u(-2 * np.pi, 2 * np.pi, n), u(-26, 26, n // 4), u(-1e8, 1e8, n // 4), 10.0 ** u(-12, 0, n // 4), 10.0 ** u(8, 300, n // 4) * rng.choice([-1, 1], n // 4),
                            np.arange(-16, 17) * (np.pi / 2), np.array([0.0, -0.0, 0.0608, 0.787, 25.0, np.inf, np.nan])])
        return a, None
    a = np.concatenate([u(-100, 100, n), 10.0 ** u(-320, 308, n // 4) * rng.choice([-1, 1], n // 4), 1.0 + u(-1e-9, 1e-9, n // 4),
                        np.array([0.0, -0.0, 1.0, -1.0, 1.5e154, 1e-200, 5e-324, np.inf, -np.inf, np.nan])])
    return a, None


def host_libm(kind, a, b=None):
    from automatedvaletparking_amd import _native
    L = C.CDLL(_native.HOSTMATH_PATH)
    a = np.ascontiguousarray(a, dtype=np.float64)
    bb = np.ascontiguousarray(b if b is not None else a, dtype=np.float64)
    out = np.empty_like(a)
    L.avp_host_libm(C.c_int(kind), a.ctypes.data_as(C.c_void_p), bb.ctypes.data_as(C.c_void_p), C.c_long(len(a)), out.ctypes.data_as(C.c_void_p))
    return out


def _py(fn, *cols):
    out = np.empty(len(cols[0]))
    for i, t in enumerate(zip(*cols)):
        try:
            out[i] = fn(*[float(v) for v in t])
        except (ValueError, OverflowError):       # math.asin(2), math.tan(inf), 1e200 ** 2: CPython raises where libm returns
            out[i] = np.nan
    return out


def test_host_build_equals_cpython_math():
    """What the reference calls: math.atan2 / asin / acos / tan and float ** 2 (rs_curve.py:176-664)."""
    rng = np.random.default_rng(2024)
    n = 40_000
    fns = [math.atan2, math.asin, math.acos, math.tan, lambda v: v ** 2]
    for kind, fn in enumerate(fns):
        a, b = args_for(kind, rng, n)
        got = host_libm(kind, a, b)
        want = _py(fn, a, b) if kind == 0 else _py(fn, a)
        ok = np.isfinite(want)                    # where CPython raised, libm returns NaN / inf: compared by the C sweep below
        assert ok.sum() > 0.9 * len(a)
        bad = np.where(got[ok].view(np.uint64) != want[ok].view(np.uint64))[0]
        assert len(bad) == 0, (kind, a[ok][bad[:3]], got[ok][bad[:3]], want[ok][bad[:3]])


def test_sweep_against_platform_libm(tmp_path):
    """scripts/glibc_libm_sweep.c: 30 distributions x 2e6 arguments (all ranges, subnormals, any bit pattern) vs libm."""
    exe = str(tmp_path / "sweep")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fno-builtin", "-fopenmp", "-o", exe,
                           os.path.join(ROOT, "scripts", "glibc_libm_sweep.c"), "-lm"])
    r = subprocess.run([exe, "2000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "total mismatches 0" in r.stdout, r.stdout[-2000:]


def test_pow2_is_not_the_plain_square():
    """Why avp_pow2 exists: libm pow(v, 2.0) -- Python's v ** 2 -- differs from v * v in ~0.08 % of arguments."""
    rng = np.random.default_rng(5)
    v = rng.uniform(-100, 100, 200_000)
    p = host_libm(4, v)
    assert np.array_equal(p, np.array([float(t) ** 2 for t in v]))
    frac = np.mean(p != v * v)
    assert 1e-4 < frac < 5e-3, frac
    assert np.max(np.abs(p - v * v) / np.spacing(v * v)) <= 1.0
