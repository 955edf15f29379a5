"""Constant structure of a whole-body QP ("task stack"): what inria_wbc fixes at controller
construction time and what therefore is NOT streamed per control tick.

Mirrors what `PosTracker::parse_tasks` (/root/reference/src/controllers/pos_tracker.cpp:161-189)
builds from a `tasks.yaml` through the factories of /root/reference/src/controllers/tasks.cpp:38-404:
which level-1 rows exist and which weight they carry, which level-0 constraints exist and in which
order they were added (yaml-cpp iterates the file in order), and the constant blocks of each
6-D contact (tsid `Contact6d`: force generator, friction pyramid, force regularisation).

Row inventories follow SURVEY.md Appendix B; the numeric constants are the ones in
/root/reference/etc/{talos,icub,franka,tiago}/tasks.yaml.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

INEQ_BOUNDS = 0
INEQ_ACTUATION = 1
INEQ_FORCE = 2

HESSIAN_REGULARIZATION = 1e-8  # tsid DEFAULT_HESSIAN_REGULARIZATION
MAX_ITER = 1000  # eiquadprog-fast DEFAULT_MAX_ITER
W_FORCE_FEET = 1e-3  # /root/reference/include/inria_wbc/controllers/tasks.hpp:23
CONTACT6D_FORCE_REG_WEIGHTS = (1.0, 1.0, 1e-3, 2.0, 2.0, 2.0)  # tsid Contact6d ctor default


def _skew(p: Sequence[float]) -> np.ndarray:
    x, y, z = p
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def contact6d_points(lxn: float, lyn: float, lxp: float, lyp: float, lz: float) -> np.ndarray:
    """3x4 contact points in the order of tasks.cpp:353-358."""
    return np.array([[-lxn, -lxn, lxp, lxp], [-lyn, lyp, -lyn, lyp], [lz, lz, lz, lz]], dtype=np.float64)


def contact6d_force_generator(points: np.ndarray) -> np.ndarray:
    """T (6x12) = per point [I3; skew(p_i)]  (tsid Contact6d::updateForceGeneratorMatrix)."""
    T = np.zeros((6, 12))
    for i in range(4):
        T[0:3, 3 * i:3 * i + 3] = np.eye(3)
        T[3:6, 3 * i:3 * i + 3] = _skew(points[:, i])
    return T


def contact6d_friction(normal: Sequence[float], mu: float, fmin: float, fmax: float) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """B (17x12), lb, ub of tsid Contact6d::updateForceInequalityConstraints:
    rows 0-3 of the first point = (-/+t1 - mu n)', (-/+t2 - mu n)', replicated block-diagonally for the
    4 points with lb = -1e10, ub = 0; row 16 = [n' n' n' n'] with lb = fmin, ub = fmax."""
    n = np.asarray(normal, dtype=np.float64)
    t1 = np.cross(n, [1.0, 0.0, 0.0])
    if np.linalg.norm(t1) < 1e-5:
        t1 = np.cross(n, [0.0, 1.0, 0.0])
    t2 = np.cross(n, t1)
    t1 = t1 / np.linalg.norm(t1)
    t2 = t2 / np.linalg.norm(t2)
    B = np.zeros((17, 12))
    B[0, 0:3] = -t1 - mu * n
    B[1, 0:3] = t1 - mu * n
    B[2, 0:3] = -t2 - mu * n
    B[3, 0:3] = t2 - mu * n
    for i in range(1, 4):
        B[4 * i:4 * i + 4, 3 * i:3 * i + 3] = B[0:4, 0:3]
    for i in range(4):
        B[16, 3 * i:3 * i + 3] = n
    lb = -1e10 * np.ones(17)
    ub = np.zeros(17)
    lb[16] = fmin
    ub[16] = fmax
    return B, lb, ub


@dataclass
class Contact:
    name: str
    points: np.ndarray  # 3x4
    normal: Tuple[float, float, float]
    mu: float
    fmin: float
    fmax: float
    force_reg_weights: Tuple[float, ...] = CONTACT6D_FORCE_REG_WEIGHTS

    @property
    def T(self) -> np.ndarray:
        return contact6d_force_generator(self.points)

    @property
    def forcereg(self) -> np.ndarray:
        """diag(w_f) * T  (tsid Contact6d::updateForceRegularizationTask)."""
        return np.diag(self.force_reg_weights) @ self.T


@dataclass
class Structure:
    """One task stack. Field meanings match `wbcqp_structure` in include/wbcqp.h."""
    name: str
    nv: int
    na: int
    contacts: List[Contact]
    # level 1
    task_names: List[str]  # index = position in the per-QP weight vector w
    default_weights: np.ndarray  # [n_tasks]
    dense_row_task: np.ndarray  # [n_dense] int32
    sel_col: np.ndarray  # [n_sel] int32
    sel_task: np.ndarray  # [n_sel] int32
    forcereg_task: np.ndarray  # [nc] int32
    # level 0
    bound_col: np.ndarray  # [n_bound] int32
    act_bounds: bool
    ineq_blocks: List[Tuple[int, int]]  # (kind, arg) in task-stack order
    hessian_reg: float = HESSIAN_REGULARIZATION
    max_iter: int = MAX_ITER
    kp: Dict[str, float] = field(default_factory=dict)  # task gains (only used by reference-stream generators)
    # level-1 tasks that make H dense (registered by the reference's factory, in none of its shipped stacks):
    # "torque" (tasks.cpp:227-271) rows scale_j [M_a(joint_j, :) | -J_a(:, joint_j)'] over the mask's ones, and
    # "cop" (tasks.cpp:156-178) 3 rows over all force variables, given per QP (field Acop)
    acteq_joint: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))  # [n_acteq] actuated joint in [0, na)
    acteq_scale: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))  # [n_acteq] `scaling:` entry, 1 without
    acteq_task: int = -1  # index into w
    cop_task: int = -1  # index into w, -1: none

    # ---- sizes -------------------------------------------------------------------------
    @property
    def nc(self) -> int:
        return len(self.contacts)

    @property
    def k(self) -> int:
        return 12 * self.nc

    @property
    def n(self) -> int:
        return self.nv + self.k

    @property
    def nu(self) -> int:
        return self.nv - self.na

    @property
    def n_dense(self) -> int:
        return int(self.dense_row_task.size)

    @property
    def n_sel(self) -> int:
        return int(self.sel_col.size)

    @property
    def n_bound(self) -> int:
        return int(self.bound_col.size)

    @property
    def n_tasks(self) -> int:
        return len(self.task_names)

    @property
    def neq(self) -> int:
        return self.nu + 6 * self.nc

    @property
    def nin(self) -> int:
        """two-sided inequality rows (tsid nIn)."""
        t = 0
        for kind, _ in self.ineq_blocks:
            t += {INEQ_BOUNDS: self.n_bound, INEQ_ACTUATION: self.na, INEQ_FORCE: 17}[kind]
        return t

    @property
    def nin2(self) -> int:
        """one-sided rows of eiquadprog's CI."""
        return 2 * self.nin

    @property
    def n_acteq(self) -> int:
        return int(self.acteq_joint.size)

    @property
    def n_cop(self) -> int:
        return 3 if self.cop_task >= 0 else 0

    @property
    def dense_h(self) -> bool:
        """H is no longer block diagonal [dv | f_1 | f_2 ...]: the stack runs the full layout with a dense H (wbcqp_layout.dense_h)."""
        return self.n_acteq > 0 or self.cop_task >= 0

    @property
    def r1(self) -> int:
        return self.n_dense + self.n_sel + 6 * self.nc + self.n_acteq + self.n_cop

    # ---- constant blocks ---------------------------------------------------------------
    def force_gen(self) -> np.ndarray:
        return np.ascontiguousarray(np.stack([c.T for c in self.contacts]) if self.contacts else np.zeros((0, 6, 12)))

    def forcereg_mat(self) -> np.ndarray:
        return np.ascontiguousarray(np.stack([c.forcereg for c in self.contacts]) if self.contacts else np.zeros((0, 6, 12)))

    def friction(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        if not self.contacts:
            return np.zeros((0, 17, 12)), np.zeros((0, 17)), np.zeros((0, 17))
        Bs, lbs, ubs = zip(*[contact6d_friction(c.normal, c.mu, c.fmin, c.fmax) for c in self.contacts])
        return np.ascontiguousarray(np.stack(Bs)), np.ascontiguousarray(np.stack(lbs)), np.ascontiguousarray(np.stack(ubs))

    # ---- per-QP record layout (element counts; include/wbcqp.h documents the same) ------
    def field_lengths(self) -> Dict[str, int]:
        return {
            "M": self.nv * (self.nv + 1) // 2,
            "h": self.nv,
            "A": self.n_dense * self.nv,
            "b1": self.r1,
            "Ac": self.nc * 6 * self.nv,
            "bc": self.nc * 6,
            "blb": self.n_bound,
            "bub": self.n_bound,
            "tlb": self.na if self.act_bounds else 0,
            "tub": self.na if self.act_bounds else 0,
            "w": self.n_tasks,
            "Acop": 3 * self.k if self.cop_task >= 0 else 0,
        }

    def flops_estimate(self, iters: float = 0.0) -> float:
        """Analytic count of the fp64 flops one QP needs on the blocked path (multiply and add counted separately):
        assembly, elimination H -> J, x0, B = J'N, Householder QR, J Q, and `iters` active-set iterations."""
        nv, n, m, k = self.nv, self.n, self.neq, self.k
        asm = self.n_dense * nv * (nv + 1)
        elim = 4.0 * nv ** 3 / 3.0 + self.nc * 4.0 * 12 ** 3 / 3.0
        x0 = 2.0 * (nv * nv + self.nc * 144)
        b = m * (nv * (nv + 1) + self.nc * 12 * 13)
        qr = 2.0 * n * m * m - 2.0 * m ** 3 / 3.0
        jq = 4.0 * n * n * m
        per_iter = 2.0 * (n * n + n * (n - m) + n * (n - m)) + 2.0 * self.na * n + 24.0 * 17 * 2 * self.nc
        if self.dense_h:  # H assembled and eliminated as one n x n matrix
            asm += (self.n_acteq + self.n_cop) * n * (n + 1)
            elim = 4.0 * n ** 3 / 3.0
            x0 = 2.0 * n * n
        return asm + elim + x0 + b + qr + jq + iters * per_iter

    def algorithmic_bytes(self, itemsize: int = 8) -> int:
        """Compact-boundary bytes per QP: inputs + x + tau + status/iters."""
        n_in = sum(self.field_lengths().values())
        return itemsize * (n_in + self.n + self.na) + 8


def _mk(name, nv, na, contacts, dense_tasks, posture, sc_tasks, has_bounds, has_act, level0_order, kp=None):
    """dense_tasks: list of (name, rows, weight) in yaml order (se3/com/momentum);
    posture: (name, weight) or None, with its yaml position given by `order` below;
    sc_tasks: list of (name, weight) self-collision tasks (1 dense row each).
    The weight vector w is ordered like tsid's level-1 list (order of addMotionTask/addRigidContact calls)."""
    task_names: List[str] = []
    weights: List[float] = []
    dense_row_task: List[int] = []
    sel_col: List[int] = []
    sel_task: List[int] = []
    forcereg_task: List[int] = []
    for item in dense_tasks:
        if item[0] == "__posture__":
            _, pname, w = item
            t = len(task_names)
            task_names.append(pname)
            weights.append(w)
            for j in range(na):
                sel_col.append(nv - na + j)
                sel_task.append(t)
        elif item[0] == "__contacts__":
            for c in contacts:
                t = len(task_names)
                task_names.append("forcereg_" + c.name)
                weights.append(W_FORCE_FEET)
                forcereg_task.append(t)
        else:
            tname, rows, w = item
            t = len(task_names)
            task_names.append(tname)
            weights.append(w)
            dense_row_task.extend([t] * rows)
    for tname, w in sc_tasks:
        t = len(task_names)
        task_names.append(tname)
        weights.append(w)
        dense_row_task.append(t)
    if contacts and not forcereg_task:
        raise ValueError("contacts given but no __contacts__ marker")
    bound_col = np.arange(nv - na, nv, dtype=np.int32) if has_bounds else np.zeros(0, np.int32)
    return Structure(
        name=name, nv=nv, na=na, contacts=contacts, task_names=task_names,
        default_weights=np.asarray(weights, dtype=np.float64),
        dense_row_task=np.asarray(dense_row_task, dtype=np.int32),
        sel_col=np.asarray(sel_col, dtype=np.int32), sel_task=np.asarray(sel_task, dtype=np.int32),
        forcereg_task=np.asarray(forcereg_task, dtype=np.int32),
        bound_col=bound_col, act_bounds=has_act, ineq_blocks=list(level0_order), kp=kp or {})


def talos_structure(single_support: bool = False) -> Structure:
    """/root/reference/etc/talos/tasks.yaml with talos.urdf: nv 50, na 44, two 6-D foot contacts
    -> n 74, nEq 18, nIn 122 (244 one-sided), 97 level-1 rows.
    single_support=True drops contact_lfoot like WalkOnSpot does (walk_on_spot.cpp:165-184):
    n 62, nEq 12, nIn 105, 91 level-1 rows."""
    pts = contact6d_points(lxn=0.11, lyn=0.069, lxp=0.1, lyp=0.069, lz=0.107)
    cl = Contact("contact_lfoot", pts, (0.0, 0.0, 1.0), 0.3, 5.0, 1500.0)
    cr = Contact("contact_rfoot", pts, (0.0, 0.0, 1.0), 0.3, 5.0, 1500.0)
    contacts = [cr] if single_support else [cl, cr]
    dense = [("head", 2, 1.0), ("head_pitch", 1, 100.0), ("head_yaw", 1, 100.0), ("lh", 6, 10.0), ("rh", 6, 10.0),
             ("torso", 2, 10.0), ("lf", 6, 1000.0), ("rf", 6, 1000.0), ("com", 3, 1000.0),
             ("__posture__", "posture", 1.75), ("momentum", 2, 0.0 if single_support else 1000.0), ("__contacts__",)]
    sc = [("self_collision-left", 2000.0), ("self_collision-right", 2000.0), ("self_collision-elbow-right", 1000.0),
          ("self_collision-elbow-left", 1000.0), ("self_collision-wrist-right", 1000.0), ("self_collision-wrist-left", 1000.0)]
    level0 = [(INEQ_BOUNDS, 0), (INEQ_ACTUATION, 0)] + [(INEQ_FORCE, c) for c in range(len(contacts))]
    kp = {"com": 30.0, "lf": 30.0, "rf": 30.0, "posture": 10.0}
    return _mk("talos_single_support" if single_support else "talos", 50, 44, contacts, dense, None, sc, True, True, level0, kp)


def icub_structure(single_support: bool = False) -> Structure:
    """/root/reference/etc/icub/tasks.yaml: nv 38, na 32, two contacts (normal 0,0,-1), bounds but no
    actuation bounds -> n 62, nEq 18, nIn 66 (132), 83 level-1 rows.
    single_support=True drops contact_lfoot the way WalkOnSpot does (walk_on_spot.cpp:165-184, as talos_structure above): n 50, nEq 12,
    nIn 49, 77 level-1 rows -- the one humanoid stack whose workgroup is small enough (52.9 KB of LDS) for three on a CU."""
    pts = contact6d_points(lxn=0.06, lyn=0.045, lxp=0.14, lyp=0.045, lz=0.065)
    cl = Contact("contact_lfoot", pts, (0.0, 0.0, -1.0), 0.3, 5.0, 1500.0)
    cr = Contact("contact_rfoot", pts, (0.0, 0.0, -1.0), 0.3, 5.0, 1500.0)
    contacts = [cr] if single_support else [cl, cr]
    dense = [("lh", 6, 1.0), ("rh", 6, 1.0), ("lf", 6, 1000.0), ("rf", 6, 10.0), ("com", 3, 3000.0), ("momentum", 2, 0.0 if single_support else 1000.0),
             ("__posture__", "posture", 0.05), ("torso", 3, 1.0), ("head", 5, 10.0), ("__contacts__",)]
    sc = [("self_collision-left", 500.0), ("self_collision-right", 500.0)]
    level0 = [(INEQ_BOUNDS, 0)] + [(INEQ_FORCE, c) for c in range(len(contacts))]
    return _mk("icub_single_support" if single_support else "icub", 38, 32, contacts, dense, None, sc, True, False, level0, {"com": 50.0, "posture": 10.0})


def franka_structure() -> Structure:
    """/root/reference/etc/franka/tasks.yaml: fixed base, nv = na = 9, no constraints, 15 level-1 rows."""
    dense = [("ee", 6, 100.0), ("__posture__", "posture", 0.75)]
    return _mk("franka", 9, 9, [], dense, None, [], False, False, [], {"ee": 30.0, "posture": 30.0})


def tiago_structure(nv: int = 12) -> Structure:
    """/root/reference/etc/tiago/tasks.yaml: fixed base, bounds only; nv is not derivable without the URDF
    (SURVEY App. B takes 12): 6 + 3 + nv + 4 level-1 rows, nIn = nv."""
    dense = [("ee", 6, 1500.0), ("head", 3, 500.0), ("__posture__", "posture", 0.1)]
    sc = [("sc-gripper", 1000.0), ("sc-wrist", 1000.0), ("sc-forearm", 1000.0), ("sc-elbow", 1000.0)]
    return _mk("tiago", nv, nv, [], dense, None, sc, True, False, [(INEQ_BOUNDS, 0)])


def three_contact_structure(nv: int = 36, na: int = 30) -> Structure:
    """Not a shipped inria_wbc stack: a floating-base robot with THREE 6-D contacts (two feet and a hand), odd contact
    count and 24 equalities. Exercises the code paths the shipped stacks do not reach (sequential equality phase,
    unpaired contact block)."""
    pts = contact6d_points(lxn=0.06, lyn=0.045, lxp=0.14, lyp=0.045, lz=0.065)
    contacts = [Contact("contact_lfoot", pts, (0.0, 0.0, 1.0), 0.4, 5.0, 1200.0),
                Contact("contact_rfoot", pts, (0.0, 0.0, 1.0), 0.4, 5.0, 1200.0),
                Contact("contact_lhand", pts, (0.0, 0.0, 1.0), 0.5, 1.0, 500.0)]
    dense = [("lh", 6, 10.0), ("rh", 6, 10.0), ("lf", 6, 1000.0), ("rf", 6, 1000.0), ("com", 3, 1000.0),
             ("__posture__", "posture", 0.5), ("momentum", 2, 100.0), ("__contacts__",)]
    level0 = [(INEQ_BOUNDS, 0), (INEQ_ACTUATION, 0), (INEQ_FORCE, 0), (INEQ_FORCE, 1), (INEQ_FORCE, 2)]
    return _mk("three_contact", nv, na, contacts, dense, None, [], True, True, level0)


def with_torque_task(st: Structure, weight: float = 1.0, mask: Sequence[int] = None, scaling: Sequence[float] = None,
                     name: str = "torque") -> Structure:
    """The stack with a `type: torque` task added (tasks.cpp:227-271): mask over the na actuated joints (default all ones),
    `scaling` = the task's weight vector (default ones), reference zero.  Appended to the level-1 list like a task at the end
    of tasks.yaml."""
    import dataclasses
    if st.n_acteq:
        raise ValueError("one torque task per stack")
    m = np.ones(st.na, np.int64) if mask is None else np.asarray(mask, np.int64)
    if m.size != st.na:
        raise ValueError("wrong size in torque mask, expected:%d got:%d" % (st.na, m.size))
    sc = np.ones(st.na) if scaling is None else np.asarray(scaling, np.float64)
    if sc.size != st.na:
        raise ValueError("wrong size in torque scaling, expected:%d got:%d" % (st.na, sc.size))
    joints = np.where(m != 0)[0].astype(np.int32)
    return dataclasses.replace(st, name=st.name + "+torque", task_names=st.task_names + [name],
                               default_weights=np.append(st.default_weights, weight), acteq_joint=joints,
                               acteq_scale=np.ascontiguousarray(sc[joints]), acteq_task=len(st.task_names))


def with_cop_task(st: Structure, weight: float = 1.0, name: str = "cop") -> Structure:
    """The stack with a `type: cop` task added (tasks.cpp:156-178): 3 rows over all contact-force variables."""
    import dataclasses
    if st.cop_task >= 0:
        raise ValueError("one cop task per stack")
    if st.nc == 0:
        raise ValueError("a cop task needs a contact")
    return dataclasses.replace(st, name=st.name + "+cop", task_names=st.task_names + [name],
                               default_weights=np.append(st.default_weights, weight), cop_task=len(st.task_names))


def with_posture_mask(st: Structure, mask: Sequence[int]) -> Structure:
    """The stack with `mask:` on its posture task (tasks.cpp:205-214): one character per actuated joint, the selection rows
    are the ones."""
    import dataclasses
    m = np.asarray(mask, np.int64)
    if m.size != st.na:
        raise ValueError("wrong size in posture mask, expected:%d got:%d" % (st.na, m.size))
    if st.n_sel != st.na:
        raise ValueError("the stack's posture task already carries a mask")
    keep = m != 0
    return dataclasses.replace(st, name=st.name + "+posture_mask", sel_col=np.ascontiguousarray(st.sel_col[keep]),
                               sel_task=np.ascontiguousarray(st.sel_task[keep]))


def cop_rows(st: Structure, placements: Sequence[Tuple[np.ndarray, np.ndarray]], cop_ref=(0.0, 0.0, 0.0),
             normal=(0.0, 0.0, 1.0)) -> np.ndarray:
    """3 x k matrix of tsid TaskCopEquality::compute [UPSTREAM-RECALL]: contact c's frame has placement (R, p) in the world,
    its contact point i sits at p_w = R p_i + p; with d = p_w - cop_ref and the forces expressed in the contact frame, the moment of
    the contact forces about the reference point has no component in the plane normal to n:
    n x sum_i (d_i x R f_i) = sum_i (d_i n' - (n . d_i) I) R f_i = 0."""
    n = np.asarray(normal, np.float64)
    A = np.zeros((3, st.k))
    for c, contact in enumerate(st.contacts):
        R, p = placements[c]
        for i in range(4):
            d = R @ contact.points[:, i] + p - np.asarray(cop_ref, np.float64)
            A[:, 12 * c + 3 * i:12 * c + 3 * i + 3] = (np.outer(d, n) - float(n @ d) * np.eye(3)) @ R
    return A


STRUCTURES = {
    "talos": talos_structure,
    "talos_single_support": lambda: talos_structure(single_support=True),
    "icub": icub_structure,
    "icub_single_support": lambda: icub_structure(single_support=True),
    "franka": franka_structure,
    "tiago": tiago_structure,
    "three_contact": three_contact_structure,
    # the two task types of the factory no shipped stack uses (weights: a torque-minimisation term and a CoP term one would add)
    "talos_torque": lambda: with_torque_task(talos_structure(), 1e-2),
    "talos_cop": lambda: with_cop_task(talos_structure(), 10.0),
    "talos_torque_cop": lambda: with_cop_task(with_torque_task(talos_structure(), 1e-2), 10.0),
    "icub_torque": lambda: with_torque_task(icub_structure(), 1e-3),
}
