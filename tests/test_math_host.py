"""avp_math.h compiled for the host: bit equality with this host's glibc libm (the arithmetic the
reference's numpy/math calls resolve to) and with CPython float semantics."""
import ctypes as C
import math

import numpy as np


def _lib():
    from automatedvaletparking_amd import _native
    L = C.CDLL(_native.HOSTMATH_PATH)
    return L


def _sincos(x):
    L = _lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    s = np.empty_like(x)
    c = np.empty_like(x)
    L.avp_host_sincos(x.ctypes.data_as(C.c_void_p), C.c_long(len(x)), s.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
    return s, c


def test_sincos_bit_equal_glibc():
    rng = np.random.default_rng(123)
    xs = np.concatenate([rng.uniform(-np.pi, np.pi, 1_500_000), rng.uniform(-7, 7, 300_000), rng.uniform(-1e3, 1e3, 200_000),
                         rng.uniform(-1e8, 1e8, 100_000), rng.uniform(-1e-7, 1e-7, 50_000),
                         np.array([0.0, -0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2, 0.855469, 0.85546875, 2.426265, 0.126, 1e-300]),
                         np.linspace(-np.pi, np.pi, 100_001)])
    s, c = _sincos(xs)
    # np.sin/np.cos == glibc on x86-64 (checked against math.sin below on a subsample)
    assert np.array_equal(s, np.sin(xs)) and np.array_equal(c, np.cos(xs))
    sub = xs[::97]
    assert np.array_equal(_sincos(sub)[0], np.array([math.sin(v) for v in sub]))
    assert np.array_equal(_sincos(sub)[1], np.array([math.cos(v) for v in sub]))


def test_fused_sincos_equals_separate_bit_for_bit():
    """avp_sincos (what the kernels call) == (avp_sin, avp_cos) on every range: tiny, table, pi/2 - x, reduced, huge."""
    L = _lib()
    rng = np.random.default_rng(321)
    xs = np.concatenate([rng.uniform(-np.pi, np.pi, 1_000_000), rng.uniform(-30, 30, 300_000), rng.uniform(-1e8, 1.2e8, 200_000),
                         rng.uniform(-1e-7, 1e-7, 50_000), 10.0 ** rng.uniform(-320, 9, 100_000) * rng.choice([-1, 1], 100_000),
                         np.array([0.0, -0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2, 0.855469, 0.85546875, 2.426265, 2.4262657, 0.126,
                                   1e-300, 105414350.0, 105414357.0, 2e8, np.inf, -np.inf, np.nan])])
    s0, c0 = _sincos(xs)
    s1 = np.empty_like(xs)
    c1 = np.empty_like(xs)
    L.avp_host_sincos_fused(xs.ctypes.data_as(C.c_void_p), C.c_long(len(xs)), s1.ctypes.data_as(C.c_void_p), c1.ctypes.data_as(C.c_void_p))
    assert np.array_equal(s0.view(np.uint64), s1.view(np.uint64)) and np.array_equal(c0.view(np.uint64), c1.view(np.uint64))


def test_hypot_mod_wraps_match_cpython():
    L = _lib()
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.uniform(-30, 30, 200_000), rng.uniform(-1e-3, 1e-3, 20_000)])
    b = np.concatenate([rng.uniform(-30, 30, 200_000), rng.uniform(-1e3, 1e3, 20_000)])
    hyp, mod, p2p, M = [np.empty_like(a) for _ in range(4)]
    L.avp_host_misc(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_long(len(a)),
                    *[v.ctypes.data_as(C.c_void_p) for v in (hyp, mod, p2p, M)])
    assert np.array_equal(hyp, np.array([math.hypot(x, y) for x, y in zip(a, b)]))
    assert np.array_equal(mod, np.array([x % y for x, y in zip(a.tolist(), b.tolist())]))
    twopi = 2.0 * math.pi

    def p2(t):
        while t > math.pi:
            t -= twopi
        while t < -math.pi:
            t += twopi
        return t

    def mm(t):
        phi = t % twopi
        if phi < -math.pi:
            phi += twopi
        if phi > math.pi:
            phi -= twopi
        return phi
    assert np.array_equal(p2p, np.array([p2(t) for t in a.tolist()]))
    assert np.array_equal(M, np.array([mm(t) for t in a.tolist()]))


def test_unified_node_search_equals_reference_routines():
    """avp_node_search (one instruction stream for all four bounds of a footprint AABB) == avp_first_ge / avp_last_le,
    and the strict variants keep their own definitions, on linspace tables like map_position, incl. exact hits,
    values off both ends and a pitch estimate that is off by a few cells."""
    L = _lib()
    rng = np.random.default_rng(11)
    for n, lo, hi in ((290, -28.0, 1.0), (250, -26.0, -1.0), (2, 0.0, 1.0), (620, 4.4e9, 4.4e9 + 62.0)):
        A = np.linspace(lo, hi, n)
        pitch = A[1] - A[0]
        v = np.concatenate([rng.uniform(lo - 3 * pitch, hi + 3 * pitch, 20000), A, A + 1e-12, A - 1e-12,
                            np.nextafter(A, np.inf), np.nextafter(A, -np.inf), [lo - 100, hi + 100]])
        m = len(v)
        out = [np.empty(m, dtype=np.int32) for _ in range(6)]
        for pf in (pitch, pitch * 1.03, pitch * 0.97):            # the pitch is only a hint
            L.avp_host_node_search(A.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_double(lo), C.c_double(pf), v.ctypes.data_as(C.c_void_p),
                                   C.c_long(m), *[o.ctypes.data_as(C.c_void_p) for o in out])
            fge, lle, fgt, llt, ulo, uhi = out
            assert np.array_equal(fge, np.searchsorted(A, v, side="left")) and np.array_equal(lle, np.searchsorted(A, v, side="right") - 1)
            assert np.array_equal(fgt, np.searchsorted(A, v, side="right")) and np.array_equal(llt, np.searchsorted(A, v, side="left") - 1)
            assert np.array_equal(ulo, fge) and np.array_equal(uhi, lle)


def test_linspace_restatement_bit_equal_numpy():
    """avp_linspace0(stop, num, q) == numpy.linspace(0, stop, num)[q] bit for bit (end-point overwrite, num 0/1/2,
    zero and denormal steps): the rasteriser's sample positions (map/costmap.py:239)."""
    L = _lib()
    rng = np.random.default_rng(2)
    cases = [(0.0, 5), (1e-320, 7), (5e-324, 3), (0.05, 0), (0.15, 1), (0.25, 2), (-3.7, 9)]
    cases += [(float(s), int(n)) for s, n in zip(rng.uniform(0, 60, 3000), rng.integers(0, 700, 3000))]
    for stop, num in cases:
        out = np.empty(max(num, 1))
        L.avp_host_linspace0(C.c_double(stop), C.c_int(num), out.ctypes.data_as(C.c_void_p))
        want = np.linspace(0, stop, num)
        assert np.array_equal(out[:num].view(np.uint64), want.view(np.uint64)), (stop, num)
