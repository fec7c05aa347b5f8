"""Trajectory wire format of the reference (animation/record_solution.py:23-51).

`DataRecorder.record(save_path, save_name, trajectory)` writes `<save_path>/Solution_<save_name>` as a tab
separated table with the header `\\tx\\ty\\ttheta\\tv\\ta\\tsigma\\tomega\\tt` and one indexed row per sample -- the
bytes `pandas.DataFrame(trajectory, columns=...).to_csv(file, index=True, sep='\\t')` produces for float rows,
without needing pandas at run time. `read` parses such a file back (what animation/curve_plot.py does with
`pd.read_csv(sep='\\t')`). Pure host I/O: nothing here touches the GPU."""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np

COLUMNS = ['x', 'y', 'theta', 'v', 'a', 'sigma', 'omega', 't']


def _fmt(v) -> str:
    # pandas writes floats with repr() (shortest round-trip) and integers without a decimal point; a column
    # is integer only when every value in it is (DataFrame dtype inference) -- handled by the caller
    return repr(float(v))


class DataRecorder:
    def __init__(self) -> None:
        pass

    @staticmethod
    def record(save_path: str, save_name: str, trajectory: Sequence[Sequence[float]]):
        """trajectory rows: x, y, theta, v, a, sigma, omega, t (record_solution.py:31-36)."""
        assert len(trajectory[0]) == 8, 'the trajectory size should be 8'
        rows = [list(r) for r in trajectory]
        ncol = len(COLUMNS)
        # per-column dtype as pandas infers it: all-int columns print as integers
        int_col = [all(isinstance(r[c], (int, np.integer)) and not isinstance(r[c], bool) for r in rows) for c in range(ncol)]
        if not os.path.exists(save_path):
            os.makedirs(save_path)
        file_name = os.path.join(save_path, 'Solution_' + save_name)
        with open(file_name, 'w', newline='') as f:
            f.write('\t' + '\t'.join(COLUMNS) + '\n')
            for i, r in enumerate(rows):
                f.write(str(i) + '\t' + '\t'.join(str(int(r[c])) if int_col[c] else _fmt(r[c]) for c in range(ncol)) + '\n')
        return file_name

    @staticmethod
    def read(file_name: str) -> np.ndarray:
        """Rows of a recorded solution as float64 [n, 8] (index column dropped)."""
        with open(file_name) as f:
            header = f.readline().rstrip('\n').split('\t')
            assert header[1:] == COLUMNS, header
            data = [[float(v) for v in line.rstrip('\n').split('\t')[1:]] for line in f if line.strip()]
        return np.array(data, dtype=np.float64).reshape(-1, 8)

    @staticmethod
    def save_gif():
        pass


def waypoints_to_trajectory(path: Sequence[Sequence[float]]) -> List[List[float]]:
    """Planner way-points [x, y, theta] as 8-column rows with the not-yet-optimised columns (v, a, sigma,
    omega, t) zero: lets the planning-only driver emit files in the solution layout."""
    return [[float(p[0]), float(p[1]), float(p[2]), 0.0, 0.0, 0.0, 0.0, 0.0] for p in path]
