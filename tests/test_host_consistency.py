"""Host-side consistency (no GPU): the status codes of include/avp.h and their Python names, the padding pose of the multi-GPU deal,
the CPU quota bench.py reports its all-core baseline against."""
import os
import re

import numpy as np

from conftest import ROOT


def test_status_names_match_the_header():
    from automatedvaletparking_amd import path_planner
    hdr = open(os.path.join(ROOT, "include", "avp.h")).read()
    enum = dict((name, int(val)) for name, val in re.findall(r"AVP_PLAN_([A-Z_]+) = (-?\d+)", hdr))
    enum.update((name, int(val)) for name, val in re.findall(r"#define AVP_PLAN_([A-Z_]+) \(?(-?\d+)\)?", hdr))
    assert enum["OK"] == 0 and enum["BAD_POSE"] == 7 and enum["DEFERRED"] == 100 and enum["UNFINISHED"] == -1
    for name, val in enum.items():
        assert path_planner.STATUS_NAMES.get(val) == name, (name, val)
    assert set(path_planner.STATUS_NAMES) == set(enum.values())


def test_padding_problems_use_a_pose_inside_the_map():
    """distributed.take_padded: a padding problem is start == goal at the list's FIRST GOAL (a pose inside the map -- the lattice
    set-up walks from the goal to the map's borders), never the origin; an empty list falls back to PAD_POSE."""
    from automatedvaletparking_amd import distributed as avd
    st = np.array([[1.0, 2.0, 0.1], [3.0, 4.0, 0.2], [5.0, 6.0, 0.3]])
    go = np.array([[7.0, 8.0, 0.4], [9.0, 10.0, 0.5], [11.0, 12.0, 0.6]])
    s_l, g_l = avd.take_padded(st, go, [2, -1, 0, -1])
    assert np.array_equal(s_l[0], st[2]) and np.array_equal(g_l[0], go[2]) and np.array_equal(s_l[2], st[0])
    for k in (1, 3):
        assert np.array_equal(s_l[k], go[0]) and np.array_equal(g_l[k], go[0])
    s_e, g_e = avd.take_padded(st[:0], go[:0], [-1, -1])
    assert np.array_equal(s_e, np.tile(np.array(avd.PAD_POSE), (2, 1))) and np.array_equal(s_e, g_e)
    idx, per = avd.deal_slice(5, 1, 2)
    assert per == 3 and list(idx) == [1, 3, -1]


def test_cpu_quota_reads_the_cgroup(tmp_path, monkeypatch):
    import builtins
    import bench
    real_open = builtins.open

    def fake(quota):
        def _open(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                p = tmp_path / "cpu.max"
                p.write_text(quota)
                return real_open(p, *a, **k)
            return real_open(path, *a, **k)
        return _open

    n = os.cpu_count() or 1
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert bench.cpu_quota() <= n
    monkeypatch.setattr(builtins, "open", fake("150000 100000\n"))
    assert bench.cpu_quota() == 1
    monkeypatch.setattr(builtins, "open", fake("%d 100000\n" % (100000 * 10 ** 6)))
    assert bench.cpu_quota() <= n
