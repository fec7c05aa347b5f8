"""Host logic of `split_path` / `split_path_batch` (segmentation, extension bookkeeping, IndexError) on the CPU: the
collision check of the extension poses is answered by the CPU oracle here (the GPU tests in test_gpu_split.py run the
same fixtures with the HIP checker)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLD, CASES

SPLIT_GOLDENS = sorted(p for p in glob.glob(os.path.join(GOLD, "g*.npz"))
                       if os.path.basename(p).startswith(("g6_", "g7_", "g8_synth_c", "g10_")) and "split_error" in np.load(p).files)


class _OracleChecker:
    def __init__(self, o, kind):
        self.o, self.kind = o, kind

    def check_batch(self, poses, variant=0):
        return self.o.check_batch(np.asarray(poses, dtype=np.float64).reshape(-1, 3), kind=self.kind)

    def check(self, node_x, node_y, theta):
        return bool(self.check_batch([[node_x, node_y, theta]])[0])


@pytest.mark.parametrize("path", SPLIT_GOLDENS)
def test_split_host_logic(path, vehicle, cfg):
    from automatedvaletparking_amd import costmap, path_planner
    from oracle import oracle
    g = np.load(path)
    if "case" in g.files:
        case = costmap.Case.read(os.path.join(CASES, f"Case{int(g['case'])}.csv"))
    else:
        case = costmap.Case()
        case.x0, case.y0, case.theta0, case.xf, case.yf, case.thetaf = [float(v) for v in g["map_poses"]]
    m = costmap.Map.from_cells(case, g["map_boundary"], int(g["map_nx"]), int(g["map_ny"]), g["map_cells"])
    c2 = dict(cfg)
    if "cfg_json" in g.files:
        c2.update(json.loads(str(g["cfg_json"])))
    chk = _OracleChecker(oracle.Oracle(m, vehicle, c2), 1 if c2["collision_check"] == "circle" else 0)
    fp = [[float(v) for v in row] for row in g["final_path"]]
    outs = path_planner.split_path_batch([fp, fp], c2, vehicle, chk)
    if str(g["split_error"]) == "IndexError":
        assert all(isinstance(o, IndexError) for o in outs)
        with pytest.raises(IndexError):
            path_planner.split_path(fp, c2, vehicle, chk)
        return
    for seg, gear in outs + [path_planner.split_path(fp, c2, vehicle, chk)]:
        assert gear == int(g["change_gear"]) and [len(s) for s in seg] == list(g["split_len"])
        assert np.array_equal(np.array(sum(seg, []), dtype=np.float64).reshape(-1, 3), g["split_concat"])
