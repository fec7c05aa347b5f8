"""The restated glibc libm on the DEVICE (include/avp_glibc_libm.h compiled by hipcc for gfx950, tables in global memory)
== the host build of the same header == the platform libm (tests/test_glibc_libm.py), bit for bit, on every range."""
import ctypes as C

import numpy as np
import pytest

from conftest import case_map_from_gold
from test_glibc_libm import args_for, host_libm

pytestmark = pytest.mark.gpu

NAMES = ["atan2", "asin", "acos", "tan", "pow2"]


@pytest.mark.parametrize("kind", range(5), ids=NAMES)
def test_device_libm_bit_equal_host(kind, vehicle, cfg):
    from automatedvaletparking_amd import _native
    dm = _native.DeviceMap(case_map_from_gold(1), vehicle, cfg)
    rng = np.random.default_rng(100 + kind)
    a, b = args_for(kind, rng, 1_000_000)
    ta = dm.dev_tensor(a)
    tb = dm.dev_tensor(b) if b is not None else ta
    out = dm.empty(len(a), dm.torch.float64)
    _native.chk(_native.lib().avp_libm_batch(dm.h, C.c_int32(kind), C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()),
                                             C.c_int64(len(a)), C.c_void_p(out.data_ptr())))
    got = out.cpu().numpy()
    want = host_libm(kind, a, b)
    nan_both = np.isnan(got) & np.isnan(want)
    bad = np.where((got.view(np.uint64) != want.view(np.uint64)) & ~nan_both)[0]
    assert len(bad) == 0, (NAMES[kind], len(bad), a[bad[:3]], got[bad[:3]], want[bad[:3]])
    # and the platform libm itself on the finite range glibc's kernels are restated for
    import math
    sub = slice(0, 100_000)
    if kind == 0:
        w = np.array([math.atan2(y, x) for y, x in zip(a[sub], b[sub])])
    elif kind == 1:
        w = np.array([math.asin(v) for v in a[sub]])
    elif kind == 2:
        w = np.array([math.acos(v) for v in a[sub]])
    elif kind == 3:
        w = np.array([math.tan(v) for v in a[sub]])
    else:
        w = np.array([float(v) ** 2 for v in a[sub]])
    assert np.array_equal(got[sub], w)
