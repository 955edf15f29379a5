"""Minimum-jerk reference streams (the closed forms behind behaviours H1/H2 of SURVEY.md 8(a)).

Follows /root/reference/include/inria_wbc/trajs/trajectory_generator.hpp:23-78
(`minimum_jerk_polynom<ZERO|FIRST|SECOND>`, `min_jerk_trajectory`) and the way
`behaviors::humanoid::MoveCom` (/root/reference/src/behaviors/humanoid/move_com.cpp:8-61) chains
segments and loops over them.
"""
from __future__ import annotations

import numpy as np


def minimum_jerk_polynom(x0, xf, t: float, duration: float, order: int = 0) -> np.ndarray:
    x0 = np.asarray(x0, dtype=np.float64)
    xf = np.asarray(xf, dtype=np.float64)
    td = t / duration
    if order == 0:
        td3 = td ** 3
        return x0 + (xf - x0) * (6 * td * td * td3 - 15 * td * td3 + 10 * td3)
    if order == 1:
        td2 = td * td
        return (xf - x0) * (30 * td * td * td2 - 60 * td * td2 + 30 * td2) / duration
    if order == 2:
        return (xf - x0) * (120 * td * td * td - 180 * td * td + 60 * td) / duration ** 2
    raise ValueError("minimum_jerk_polynom is not implemented for derivative order %d" % order)


def min_jerk_trajectory(start, dest, dt: float, duration: float, order: int = 0) -> np.ndarray:
    """[n_steps, dim] samples at t = dt*i, n_steps = floor(duration/dt)."""
    n_steps = int(np.floor(duration / dt))
    return np.stack([minimum_jerk_polynom(start, dest, dt * i, duration, order) for i in range(n_steps)])


def move_com_stream(task_init, targets, mask: str, dt: float, duration: float, loop: bool = True, absolute: bool = False):
    """Position / velocity / acceleration arrays that `MoveCom` precomputes in its constructor
    (move_com.cpp:22-45). For etc/talos/squat.yaml: targets [[0,0,-0.2]], mask '001', 2 s, loop
    -> 2000 + 2000 samples at dt = 1 ms."""
    task_init = np.asarray(task_init, dtype=np.float64)
    targets = [list(t) for t in targets]
    if loop:
        targets.append(list(task_init) if absolute else [0.0, 0.0, 0.0])
    pos, vel, acc = [], [], []
    start = task_init.copy()
    for tgt in targets:
        end = task_init.copy()
        for j in range(3):
            if mask[j] == "1":
                end[j] = tgt[j] if absolute else tgt[j] + task_init[j]
        pos.append(min_jerk_trajectory(start, end, dt, duration, 0))
        vel.append(min_jerk_trajectory(start, end, dt, duration, 1))
        acc.append(min_jerk_trajectory(start, end, dt, duration, 2))
        start = end
    return np.concatenate(pos), np.concatenate(vel), np.concatenate(acc)
