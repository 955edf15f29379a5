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


def _angle_axis(R: np.ndarray):
    """Eigen::AngleAxisd(Matrix3d): matrix -> quaternion -> (angle, axis); identity gives (0, x)."""
    t = np.trace(R)
    if t > 0.0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        v = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s])
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        v = np.zeros(3)
        v[i] = 0.5 * s
        s = 0.5 / s
        w = (R[k, j] - R[j, k]) * s
        v[j] = (R[j, i] + R[i, j]) * s
        v[k] = (R[k, i] + R[i, k]) * s
    n = np.linalg.norm(v)
    if n == 0.0:
        return 0.0, np.array([1.0, 0.0, 0.0])
    angle = 2.0 * np.arctan2(n, abs(w))
    return angle, v / (-n if w < 0.0 else n)


def min_jerk_se3(R0: np.ndarray, p0: np.ndarray, R1: np.ndarray, p1: np.ndarray, dt: float, duration: float):
    """Poses (R [n,3,3], p [n,3]) and 6-D first / second derivatives (linear, angular) of the SE(3) min-jerk move of
    /root/reference/include/inria_wbc/trajs/trajectory_generator.hpp:80-147: translation by the polynomial, rotation about
    the fixed axis of R0' R1 with a min-jerk angle."""
    angle, axis = _angle_axis(R0.T @ R1)
    n = int(np.floor(duration / dt))
    Rs, ps, d1, d2 = np.zeros((n, 3, 3)), np.zeros((n, 3)), np.zeros((n, 6)), np.zeros((n, 6))
    K = np.array([[0.0, -axis[2], axis[1]], [axis[2], 0.0, -axis[0]], [-axis[1], axis[0], 0.0]])
    for i in range(n):
        t = dt * i
        ps[i] = minimum_jerk_polynom(p0, p1, t, duration, 0)
        a = minimum_jerk_polynom([0.0], [angle], t, duration, 0)[0]
        Rs[i] = R0 @ (np.eye(3) + np.sin(a) * K + (1.0 - np.cos(a)) * (K @ K))
        for order, out in ((1, d1), (2, d2)):
            out[i, :3] = minimum_jerk_polynom(p0, p1, t, duration, order)
            out[i, 3:] = R0 @ (minimum_jerk_polynom([0.0], [angle], t, duration, order)[0] * axis)
    return Rs, ps, d1, d2


def cartesian_stream(R_init: np.ndarray, p_init: np.ndarray, rel_pos, dt: float, duration: float, loop: bool = True, rel_rpy=None):
    """The sample stream `generic::cartesian` precomputes for one task (cartesian.cpp:28-61): init -> target (-> init when
    looping).  Returns (R, p, vel, acc) concatenated over the segments."""
    Rf, pf = R_init.copy(), p_init + np.asarray(rel_pos, dtype=np.float64)
    if rel_rpy is not None and len(rel_rpy) == 3:
        r, pch, y = rel_rpy
        cz, sz, cy, sy, cx, sx = np.cos(y), np.sin(y), np.cos(pch), np.sin(pch), np.cos(r), np.sin(r)
        rot = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Rf = rot @ R_init
    segs = [min_jerk_se3(R_init, p_init, Rf, pf, dt, duration)]
    if loop:
        segs.append(min_jerk_se3(Rf, pf, R_init, p_init, dt, duration))
    return tuple(np.concatenate([s[k] for s in segs]) for k in range(4))
