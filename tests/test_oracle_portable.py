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


# golden problems on which the portable arithmetic resolves a tie differently from glibc (measured; see
# DESIGN.md "Numerics"). With the nearly correctly rounded atan2/asin/acos of include/avp_libm.h there is
# none among the 33 finished golden plans (the first, fdlibm-accuracy version diverged on Case18 at pop 466).
KNOWN_TIE_DIVERGENCE = set()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g6_trace_case*.npz")) + glob.glob(os.path.join(GOLD, "g7_random_case*.npz"))))
def test_trace_portable_vs_reference(path, vehicle, cfg):
    """north_star bar for the device arithmetic, checked on the CPU: the popped grid-id sequence is
    identical to the reference's and the final path agrees to 1e-6, except where an exact
    Reeds-Shepp tie or a twin-node tie (two open nodes whose poses differ by an ulp and whose costs
    are equal in glibc arithmetic) is resolved the other way; those cases are listed, not hidden."""
    from automatedvaletparking_amd import costmap
    from oracle import oracle
    g = np.load(path)
    if str(g["status"]) != "ok":
        pytest.skip("no finished reference plan")
    k = int(g["case"])
    case = costmap.Case.read(os.path.join(CASES, f"Case{k}.csv"))
    m = costmap.Map.from_cells(case, g["map_boundary"], int(g["map_nx"]), int(g["map_ny"]), g["map_cells"])
    st, go = (g["start"], g["goal"]) if "start" in g.files else ([case.x0, case.y0, case.theta0], [case.xf, case.yf, case.thetaf])
    o = oracle.Oracle(m, vehicle, cfg)
    with oracle.portable_libm():
        r = o.plan(st, go, max_trace=len(g["pops"]) + 2000)
    gp = g["pops"]
    name = os.path.basename(path)
    same_ids = r["n_pops"] == len(gp) and np.array_equal(r["trace"][:, 2], gp[:, 2])
    if name in KNOWN_TIE_DIVERGENCE:
        assert r["status"] == 0
        return
    assert same_ids, name
    assert r["final_path"].shape == g["final_path"].shape and np.abs(r["final_path"] - g["final_path"]).max() < 1e-6, name
