"""Portable-libm oracle mode (the arithmetic the device uses for atan2/asin/acos/tan) vs the glibc
mode (the reference's arithmetic): the difference is confined to Reeds-Shepp last-ulp noise and to
the exact ties between mirror-image words that this noise decides."""
import glob
import os

import numpy as np
import pytest

from conftest import CASES, GOLD, gold, case_map_from_gold


def test_rs_portable_vs_reference(vehicle, cfg):
    from oracle import oracle
    g4 = gold("g4_rs.npz")
    o = oracle.Oracle(case_map_from_gold(1), vehicle, cfg)
    maxc = float(g4["maxc"])
    with oracle.portable_libm():
        r = o.rs_optimal(g4["q0"], g4["q1"], maxc, maxpts=int(g4["npts"].max()) + 8)
    assert (r["status"] == 0).all()
    assert np.abs(r["L"] - g4["L"]).max() < 1e-12            # the optimum length never changes
    same = (r["types"] == g4["types"]).all(axis=1)
    assert same.mean() > 0.999, same.mean()          # measured 0.9999: the flips are exact ties decided by glibc's
                                                     # own misroundings (it is not correctly rounded either)
    ns, k = g4["pts"].shape[:2]
    sm = same[:ns]
    d = np.abs(r["pts"][:ns][sm][:, :k] - g4["pts"][sm])
    d[..., 2] = np.minimum(d[..., 2], np.abs(d[..., 2] - 2 * np.pi))
    assert d.max() < 1e-9
    assert np.array_equal(r["npts"][same], g4["npts"][same])
    # every type flip is an exact tie: the reference's candidate list holds the flipped word with the
    # optimum's length (to 1e-12)
    flips = np.where(~same)[0]
    nc, ty, le = o.rs_candidates(g4["q0"][flips], g4["q1"][flips], maxc)
    for j, i in enumerate(flips):
        Ls = np.abs(le[j, :nc[j]]).sum(axis=1) / maxc
        match = [(ty[j, c] == r["types"][i]).all() and abs(Ls[c] - g4["L"][i]) < 1e-12 for c in range(nc[j])]
        assert any(match), (i, r["types"][i], g4["types"][i])


# golden problems on which the portable arithmetic orders two open nodes differently from glibc (measured; see
# DESIGN.md "Numerics"). With the nearly correctly rounded atan2/asin/acos of include/avp_libm.h there is none
# among the finished golden plans (the first, fdlibm-accuracy version diverged on Case18 at pop 466) and one
# among the 117 trace fixtures incl. the unfinished reference runs: in the two-circle-checker variant below a
# Reeds-Shepp length that differs in its last bit (h = 9.4061481238178093 vs ...8111; the same 1-ulp noise shows
# up at pops 483 and 1048 without consequence) swaps two nodes of almost equal f at pop 3506 of 4048.
KNOWN_TIE_DIVERGENCE = {"g10_variant_circle_case4_2.npz"}

TRACE_FIXTURES = sorted(glob.glob(os.path.join(GOLD, "g6_trace_case*.npz")) + glob.glob(os.path.join(GOLD, "g7_random_case*.npz"))
                        + glob.glob(os.path.join(GOLD, "g10_variant_*.npz")))


@pytest.mark.parametrize("path", TRACE_FIXTURES)
def test_trace_portable_vs_reference(path, vehicle, cfg):
    """north_star bar for the device arithmetic, checked on the CPU: the popped grid-id sequence is
    identical to the reference's -- for finished reference runs to the end (and the final path agrees to
    1e-6), for runs that hit the generator's time limit over their whole recorded prefix -- except where an
    exact Reeds-Shepp tie or a twin-node tie (two open nodes whose costs differ by an ulp of libm noise) is
    resolved the other way; those cases are listed, not hidden."""
    import json
    from automatedvaletparking_amd import costmap
    from oracle import oracle
    g = np.load(path)
    gp = g["pops"]
    if str(g["status"]) not in ("ok", "timeout") or len(gp) == 0:
        pytest.skip("no reference pop trace")
    k = int(g["case"])
    case = costmap.Case.read(os.path.join(CASES, f"Case{k}.csv"))
    m = costmap.Map.from_cells(case, g["map_boundary"], int(g["map_nx"]), int(g["map_ny"]), g["map_cells"])
    st, go = (g["start"], g["goal"]) if "start" in g.files else ([case.x0, case.y0, case.theta0], [case.xf, case.yf, case.thetaf])
    c2 = dict(cfg)
    if "cfg_json" in g.files:
        c2.update(json.loads(str(g["cfg_json"])))
    finished = str(g["status"]) == "ok"
    o = oracle.Oracle(m, vehicle, c2, max_pops=0 if finished else len(gp))
    with oracle.portable_libm():
        r = o.plan(st, go, max_trace=len(gp) + 2000)
    name = os.path.basename(path)
    n = min(r["n_pops"], len(gp))
    same_ids = (r["n_pops"] == len(gp) or (not finished and r["n_pops"] >= len(gp))) and np.array_equal(r["trace"][:n, 2], gp[:n, 2])
    if name in KNOWN_TIE_DIVERGENCE:
        assert not same_ids, f"{name} no longer diverges: take it off the list"
        return
    assert same_ids, name
    if finished:
        assert r["final_path"].shape == g["final_path"].shape and np.abs(r["final_path"] - g["final_path"]).max() < 1e-6, name


def test_known_tie_divergence_is_a_one_ulp_swap(vehicle, cfg):
    """The listed divergence, pinned down on the CPU from the reference's OWN trace: at the first differing pop the
    reference pops node A then node B whose keys differ by ONE ulp (f = 14.50108653548671 vs ...6712); the portable
    arithmetic rounds one Reeds-Shepp length of that pair the other way and pops B first. Nothing else differs up to
    there, and the node popped instead is the reference's very next pop."""
    import json
    from automatedvaletparking_amd import costmap
    from oracle import oracle
    (name,) = KNOWN_TIE_DIVERGENCE
    g = np.load(os.path.join(GOLD, name))
    gp = g["pops"]
    case = costmap.Case.read(os.path.join(CASES, f"Case{int(g['case'])}.csv"))
    m = costmap.Map.from_cells(case, g["map_boundary"], int(g["map_nx"]), int(g["map_ny"]), g["map_cells"])
    c2 = dict(cfg)
    c2.update(json.loads(str(g["cfg_json"])))
    with oracle.portable_libm():
        r = oracle.Oracle(m, vehicle, c2, max_pops=len(gp)).plan(g["start"], g["goal"], max_trace=len(gp) + 8)
    t = r["trace"]
    n = min(len(t), len(gp))
    i = int(np.where(t[:n, 2] != gp[:n, 2])[0][0])
    assert i == 3506 and np.array_equal(t[:i, :3], gp[:i, :3])                 # identical node / parent / grid id up to the swap
    fa, fb = float(gp[i, 8]), float(gp[i + 1, 8])                                # the reference's own two keys
    assert fa < fb and abs(fb - fa) <= np.spacing(fa) * 1.0000001                # exactly one unit in the last place apart
    assert t[i, 0] == gp[i + 1, 0] and t[i, 2] == gp[i + 1, 2]                   # the port pops the reference's NEXT node first
    assert abs(float(t[i, 8]) - fb) <= np.spacing(fb)
