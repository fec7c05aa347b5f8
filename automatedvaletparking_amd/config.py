"""Planner configuration: the flat dict the reference's `config/read_config.py:16-23` returns.

`default_config()` carries the values of the reference's `config/config.yaml` for the keys the
hot path consumes (hybrid-A* discretisation, costs, collision margins) plus the downstream keys
so that the same dict can be handed to the reference's later stages unchanged.
`read_config(name, directory)` loads a YAML file with the same flat layout and overlays it.
"""
from __future__ import annotations

import os
from typing import Optional


def default_config() -> dict:
    return {
        # hybrid A*
        'steering_angle_num': 5, 'dt': 0.6, 'Benchmark_path': 'BenchmarkCases', 'trajectory_dt': 0.2,
        'map_discrete_size': 0.1, 'flag_radius': 18, 'extended_num': 1,
        # costs
        'cost_gear': 1, 'cost_heading_change': 0.5, 'cost_scale': 10,
        # collision check
        'safe_side_dis': 0.1, 'safe_fr_dis': 0.1, 'collision_check': 'distance', 'draw_collision': False,
        # downstream stages (not consumed by the hot path)
        'expand_dis': 0.8, 'smooth_cost': 5, 'compact_cost': 3, 'offset_cost': 0.8, 'slack_cost': 1,
        'velocity_func_type': 'sin_func', 'velocity_plan_num': 100,
        'cost_steering_angle': 10, 'cost_omega': 10, 'cost_acceleration': 10, 'cost_velocity': 10,
        'cost_time': 100, 'save_path': './solution', 'pic_path': './pictures',
    }


def read_config(config_name: Optional[str] = None, directory: Optional[str] = None) -> dict:
    cfg = default_config()
    if config_name is None:
        return cfg
    import yaml
    path = os.path.join(directory or os.getcwd(), config_name + '.yaml')
    with open(path, 'r', encoding='utf-8') as f:
        loaded = yaml.load(f.read(), Loader=yaml.FullLoader)
    if loaded:
        cfg.update(loaded)
    return cfg
