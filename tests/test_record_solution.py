"""Solution wire format (animation/record_solution.py:23-51): byte-identical to what the reference's
pandas call writes (pandas is the checker here; the module itself does not need it)."""
import os

import numpy as np
import pytest


def _pandas_bytes(tmp_path, traj):
    pd = pytest.importorskip("pandas")
    df = pd.DataFrame(traj)
    df.columns = ['x', 'y', 'theta', 'v', 'a', 'sigma', 'omega', 't']
    p = tmp_path / "ref.tsv"
    df.to_csv(str(p), index='True', sep='\t')
    return open(p, 'rb').read()


@pytest.mark.parametrize("kind", ["random", "special", "intcol"])
def test_record_bytes_match_pandas(tmp_path, kind):
    from automatedvaletparking_amd.record_solution import DataRecorder
    rng = np.random.default_rng(3)
    if kind == "random":
        traj = rng.normal(size=(50, 8)).tolist()
    elif kind == "special":
        traj = [[0.0, -0.0, 1e-5, 1e22, 123456789.125, -1.5e-300, 2.5, 1.0], [1 / 3, 2 / 3, np.pi, -np.pi, 1e16, 1e15, 0.1, 100.0]]
    else:
        traj = [[float(i), 0.5 * i, 0.0, 1, 0.0, 0.0, 0.0, i] for i in range(5)]      # int columns stay ints
    out = DataRecorder.record(str(tmp_path / "solution"), "Case1.csv", traj)
    assert os.path.basename(out) == "Solution_Case1.csv"
    assert open(out, 'rb').read() == _pandas_bytes(tmp_path, traj)
    back = DataRecorder.read(out)
    assert np.array_equal(back, np.array(traj, dtype=np.float64))


def test_record_rejects_wrong_width(tmp_path):
    from automatedvaletparking_amd.record_solution import DataRecorder
    with pytest.raises(AssertionError):
        DataRecorder.record(str(tmp_path), "x", [[0.0, 1.0, 2.0]])


def test_waypoints_to_trajectory():
    from automatedvaletparking_amd.record_solution import waypoints_to_trajectory
    t = waypoints_to_trajectory([[1.0, 2.0, 0.5], [1.5, 2.5, 0.6]])
    assert t == [[1.0, 2.0, 0.5, 0, 0, 0, 0, 0], [1.5, 2.5, 0.6, 0, 0, 0, 0, 0]]
