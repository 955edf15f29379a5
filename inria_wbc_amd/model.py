"""Kinematic trees and task bindings as DATA: what the step before the QP needs.

The reference gets its tree from a URDF through pinocchio (`robots::RobotWrapper`, /root/reference/src/controllers/
pos_tracker.cpp:83) and its task bindings from tasks.yaml (/root/reference/src/controllers/tasks.cpp:38-404).  Neither
the URDFs nor pinocchio exist here, so a `Model` carries the parsed form directly -- parents, joint types, joint
placements, spatial inertias, named frames, limits -- and `talos_like()` is a stand-in with Talos' topology and joint
names (/root/reference/etc/talos/configurations.srdf:4-48) and plausible, seeded geometry.  A `TaskMap` is the per-task
binding (tracked frame, mask, gains) in the order of /root/reference/etc/talos/tasks.yaml; together with a `Structure`
(structure.py) it says which rows the device writes into the QP record of include/wbcqp.h.

Host-side initialisation only: `frame_placements` / `com` below are the numpy forward kinematics the constructor-time code
needs (the reference initialises every task reference to the current placement, tasks.cpp:64-80,109).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .structure import Structure

J_FREEFLYER, J_RX, J_RY, J_RZ, J_PX, J_PY, J_PZ = range(7)
T_SE3, T_COM, T_MOMENTUM, T_SELFCOLLISION = range(4)
REF_LEN = {T_SE3: 24, T_COM: 9, T_MOMENTUM: 12, T_SELFCOLLISION: 0}


def _rot(axis: int, a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    R = np.eye(3)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    R[i, i] = c
    R[i, j] = -s
    R[j, i] = s
    R[j, j] = c
    return R


def quat_to_rot(x: float, y: float, z: float, w: float) -> np.ndarray:
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pack_se3(R: np.ndarray, p: Sequence[float]) -> np.ndarray:
    """R row-major (9) then p (3): the layout of every placement array at the C boundary."""
    return np.concatenate([np.asarray(R, dtype=np.float64).reshape(9), np.asarray(p, dtype=np.float64)])


def se3_ref(R: np.ndarray, p: Sequence[float]) -> np.ndarray:
    """Reference placement as tsid's SE3ToVector / src/trajs/loader.cpp:11-53 write it: translation, rotation column-major."""
    return np.concatenate([np.asarray(p, dtype=np.float64), np.asarray(R, dtype=np.float64).T.reshape(9)])


@dataclass
class Model:
    name: str
    floating_base: bool
    parent: np.ndarray  # [nbody] int32, parent[i] < i, -1 for body 0
    jtype: np.ndarray  # [nbody] int32
    placement: np.ndarray  # [nbody, 12]
    inertia: np.ndarray  # [nbody, 10] mass, com (3), I_c xx xy xz yy yz zz
    joint_names: List[str]
    frame_names: List[str]
    frame_body: np.ndarray  # [nframe] int32
    frame_placement: np.ndarray  # [nframe, 12]
    q_lb: np.ndarray
    q_ub: np.ndarray
    dq_max: np.ndarray
    tau_max: np.ndarray
    q0: np.ndarray  # reference configuration (srdf group_state)
    gravity: Tuple[float, float, float] = (0.0, 0.0, -9.81)

    @property
    def nbody(self) -> int:
        return int(self.parent.size)

    @property
    def nv(self) -> int:
        return self.nbody + (5 if self.floating_base else 0)

    @property
    def nq(self) -> int:
        return self.nbody + (6 if self.floating_base else 0)

    @property
    def na(self) -> int:
        return self.nv - (6 if self.floating_base else 0)

    @property
    def nframe(self) -> int:
        return int(self.frame_body.size)

    def frame(self, name: str) -> int:
        try:
            return self.frame_names.index(name)
        except ValueError:
            raise KeyError("Unknown frame or joint [%s]" % name)  # tasks.cpp:68-69

    def idx_q(self, i: int) -> int:
        return (0 if i == 0 else 6 + i) if self.floating_base else i

    def idx_v(self, i: int) -> int:
        return (0 if i == 0 else 5 + i) if self.floating_base else i

    def subtree_last(self) -> np.ndarray:
        last = np.arange(self.nbody, dtype=np.int32)
        for i in range(self.nbody - 1, 0, -1):
            last[self.parent[i]] = max(last[self.parent[i]], last[i])
        return last

    def depth(self) -> np.ndarray:
        d = np.zeros(self.nbody, dtype=np.int32)
        for i in range(1, self.nbody):
            d[i] = d[self.parent[i]] + 1
        return d

    def validate(self) -> None:
        nb = self.nbody
        assert self.parent[0] == -1 and all(0 <= self.parent[i] < i for i in range(1, nb)), "parents must precede children"
        assert (self.jtype[0] == J_FREEFLYER) == self.floating_base and not np.any(self.jtype[1:] == J_FREEFLYER)
        last = self.subtree_last()
        for i in range(1, nb):
            assert last[self.parent[i]] >= last[i]
        # depth-first numbering: the subtree of body i is the contiguous range [i, last[i]]
        for i in range(nb):
            for j in range(i + 1, last[i] + 1):
                b = j
                while b > i:
                    b = self.parent[b]
                assert b == i, "bodies are not numbered depth-first"
        for a in (self.q_lb, self.q_ub, self.dq_max, self.tau_max):
            assert a.shape == (self.na,)
        assert self.q0.shape == (self.nq,)

    # ---- numpy forward kinematics (host, initialisation time) -----------------------------------
    def body_placements(self, q: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        nb = self.nbody
        R = np.zeros((nb, 3, 3))
        p = np.zeros((nb, 3))
        for i in range(nb):
            Rp, pp = self.placement[i, :9].reshape(3, 3), self.placement[i, 9:]
            jt, qi = int(self.jtype[i]), q[self.idx_q(i):]
            if jt == J_FREEFLYER:
                Rj, pj = quat_to_rot(*qi[3:7]), qi[0:3]
            elif jt <= J_RZ:
                Rj, pj = _rot(jt - J_RX, qi[0]), np.zeros(3)
            else:
                Rj, pj = np.eye(3), np.eye(3)[jt - J_PX] * qi[0]
            Rl, pl = Rp @ Rj, Rp @ pj + pp
            if self.parent[i] >= 0:
                R[i], p[i] = R[self.parent[i]] @ Rl, R[self.parent[i]] @ pl + p[self.parent[i]]
            else:
                R[i], p[i] = Rl, pl
        return R, p

    def frame_placements(self, q: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        R, p = self.body_placements(q)
        Rf = np.zeros((self.nframe, 3, 3))
        pf = np.zeros((self.nframe, 3))
        for f in range(self.nframe):
            b = self.frame_body[f]
            Rf[f] = R[b] @ self.frame_placement[f, :9].reshape(3, 3)
            pf[f] = R[b] @ self.frame_placement[f, 9:] + p[b]
        return Rf, pf

    def com(self, q: np.ndarray) -> np.ndarray:
        R, p = self.body_placements(q)
        m = self.inertia[:, 0]
        c = np.einsum("bij,bj->bi", R, self.inertia[:, 1:4]) + p
        return (m[:, None] * c).sum(axis=0) / m.sum()


@dataclass
class TaskBlock:
    name: str
    kind: int
    frame: int = 0
    mask: int = 0
    kp: float = 0.0
    kd: float = 0.0
    ref: int = 0
    avoided: List[Tuple[int, float]] = field(default_factory=list)
    radius: float = 0.0
    margin: float = 0.0
    m: float = 0.0

    @property
    def rows(self) -> int:
        return 1 if self.kind == T_SELFCOLLISION else bin(self.mask).count("1")


@dataclass
class TaskMap:
    blocks: List[TaskBlock]
    sel_col: np.ndarray
    posture_kp: float
    posture_kd: float
    posture_ref: int
    contact_frame: np.ndarray
    contact_kp: np.ndarray
    contact_kd: np.ndarray
    contact_ref: np.ndarray
    n_bound: int
    dt: float
    nref: int
    n_acteq: int = 0  # rows of a `torque` task (tasks.cpp:227-271): zero right-hand sides, nothing to compute per tick
    cop: bool = False  # a `cop` task (tasks.cpp:156-178): three rows over the contact forces from the contact frames' placements
    contact_points: np.ndarray = field(default_factory=lambda: np.zeros(0))  # [ncontact][4][3] (tasks.cpp:353-358)

    @property
    def n_dense(self) -> int:
        return sum(b.rows for b in self.blocks)

    @property
    def ncontact(self) -> int:
        return int(self.contact_frame.size)

    def used_frames(self) -> List[int]:
        fr = []
        for b in self.blocks:
            if b.kind in (T_SE3, T_SELFCOLLISION):
                fr.append(b.frame)
            fr.extend(f for f, _ in b.avoided)
        fr.extend(int(f) for f in self.contact_frame)
        return sorted(set(fr))

    def algorithmic_bytes(self, model: Model, st: Structure, itemsize: int = 8) -> int:
        """Bytes one instance of the rows kernel must move: state and references in, QP record out."""
        L = st.field_lengths()
        out = L["M"] + L["h"] + L["A"] + L["b1"] + L["Ac"] + L["bc"] + L["blb"] + L["bub"] + L["Acop"]
        return itemsize * (model.nq + model.nv + self.nref + out)


def _mask_bits(s: str) -> int:
    """tasks.cpp:27-35 convert_mask: character i of the string is row i."""
    return sum(1 << i for i, ch in enumerate(s) if ch != "0")


def build_taskmap(model: Model, st: Structure, stack: Sequence[dict], dt: float = 1e-3) -> TaskMap:
    """`stack` lists the task nodes of a tasks.yaml in file order (dicts with the yaml's own keys).  Gains follow the
    factories: Kd = 2 sqrt(Kp) for se3 / com / posture / momentum / contact (tasks.cpp:57,107,140,204,360), kp and kd both
    given for self-collision (tasks.cpp:398-399)."""
    blocks: List[TaskBlock] = []
    sc: List[TaskBlock] = []
    off = 0
    posture = None
    contacts = []
    has_bounds = False
    for node in stack:
        ty, name = node["type"], node["name"]
        if ty == "se3":
            kp = float(node["kp"])
            blocks.append(TaskBlock(name, T_SE3, model.frame(node["tracked"]), _mask_bits(node["mask"]), kp, 2.0 * np.sqrt(kp), off))
            off += REF_LEN[T_SE3]
        elif ty == "com":
            kp = float(node["kp"])
            assert len(node["mask"]) == 3, "CoM masks needs to be 3D (x y z)"  # tasks.cpp:101
            blocks.append(TaskBlock(name, T_COM, 0, _mask_bits(node["mask"]), kp, 2.0 * np.sqrt(kp), off))
            off += REF_LEN[T_COM]
        elif ty == "momentum":
            kp = float(node["kp"])
            assert len(node["mask"]) == 6, "Momentum mask needs to be 6D"  # tasks.cpp:134
            blocks.append(TaskBlock(name, T_MOMENTUM, 0, _mask_bits(node["mask"]), kp, 2.0 * np.sqrt(kp), off))
            off += REF_LEN[T_MOMENTUM]
        elif ty == "self-collision":
            sc.append(TaskBlock(name, T_SELFCOLLISION, model.frame(node["tracked"]), 1, float(node["kp"]), float(node["kd"]), 0,
                                [(model.frame(k), float(r)) for k, r in node["avoided"].items()],
                                float(node["radius"]), float(node["margin"]), float(node["m"])))
        elif ty == "posture":
            posture = node
        elif ty == "contact":
            contacts.append(node)
        elif ty == "bounds":
            has_bounds = True
        elif ty == "actuation-bounds":
            pass  # constant torque limits: part of the QP record (tlb / tub), nothing to compute per tick
        elif ty in ("torque", "cop"):
            pass  # which rows exist is the structure's business (st.n_acteq, st.cop_task); the cop rows need the contact frames, below
        else:
            raise KeyError("unknown task type [%s]" % ty)
    blocks = blocks + sc  # structure.py keeps the self-collision rows behind the se3 / com / momentum rows
    pkp = float(posture["kp"]) if posture else 0.0
    posture_ref = off
    if posture:
        off += model.na
    cref = []
    for _ in contacts:
        cref.append(off)
        off += 24  # a full sample: placement 12, velocity 6, acceleration 6
    ckp = np.array([float(c["kp"]) for c in contacts], dtype=np.float64)
    tm = TaskMap(blocks=blocks, sel_col=st.sel_col.copy(), posture_kp=pkp, posture_kd=2.0 * np.sqrt(pkp), posture_ref=posture_ref,
                 contact_frame=np.array([model.frame(c["joint"]) for c in contacts], dtype=np.int32),
                 contact_kp=ckp, contact_kd=2.0 * np.sqrt(ckp), contact_ref=np.array(cref, dtype=np.int32),
                 n_bound=model.na if has_bounds else 0, dt=dt, nref=off, n_acteq=st.n_acteq, cop=st.cop_task >= 0,
                 contact_points=np.ascontiguousarray(np.stack([c.points.T for c in st.contacts])) if st.nc else np.zeros(0))
    assert st.n_acteq == sum(int(np.sum(np.asarray([ch != "0" for ch in n.get("mask", "1" * model.na)]))) for n in stack if n["type"] == "torque")
    assert (st.cop_task >= 0) == any(n["type"] == "cop" for n in stack)
    # the structure (which rows exist) and the map (how they are computed) must describe the same stack
    assert tm.n_dense == st.n_dense, (tm.n_dense, st.n_dense)
    assert [b.rows for b in tm.blocks] == [int((st.dense_row_task == t).sum()) for t in dict.fromkeys(st.dense_row_task.tolist())]
    assert tm.ncontact == st.nc and tm.n_bound == st.n_bound and (st.n_sel == 0 or posture is not None)
    assert model.nv == st.nv and model.na == st.na
    return tm


JOINT_TYPE_NAMES = ("freeflyer", "RX", "RY", "RZ", "PX", "PY", "PZ")


def to_yaml(model: Model, path: str, skip_frames: Sequence[str] = (), ref_name: str = "start") -> None:
    """Writes the tree in the YAML subset the C++ facade reads (csrc/host/include/inria_wbc/robots/robot_wrapper.hpp): the
    stand-in for the URDF the reference hands to pinocchio.  Joint frames are implicit (one per joint, as pinocchio adds
    them); `skip_frames` are left to a frames.yaml in the reference's own schema."""
    fmt = lambda a: "[" + ", ".join(repr(float(x)) for x in a) + "]"
    with open(path, "w") as f:
        f.write("# kinematic tree of %s (written by inria_wbc_amd/model.py::to_yaml)\n" % model.name)
        f.write("name: %s\nfloating_base: %s\ngravity: %s\njoints:\n" % (model.name, "true" if model.floating_base else "false", fmt(model.gravity)))
        ja = 0
        for i, nm in enumerate(model.joint_names):
            f.write("  %s:\n    parent: %s\n    type: %s\n" % (nm, "universe" if model.parent[i] < 0 else model.joint_names[model.parent[i]],
                                                               JOINT_TYPE_NAMES[int(model.jtype[i])]))
            f.write("    placement: %s\n    inertia: %s\n" % (fmt(model.placement[i]), fmt(model.inertia[i])))
            if model.jtype[i] != J_FREEFLYER:
                f.write("    limits: %s\n" % fmt([model.q_lb[ja], model.q_ub[ja], model.dq_max[ja], model.tau_max[ja]]))
                ja += 1
        extra = [k for k in range(model.nframe) if model.frame_names[k] not in model.joint_names and model.frame_names[k] not in skip_frames]
        if extra:
            f.write("frames:\n")
            for k in extra:
                f.write("  %s:\n    parent: %s\n    placement: %s\n" % (model.frame_names[k], model.joint_names[model.frame_body[k]], fmt(model.frame_placement[k])))
        f.write("reference_configurations:\n  %s: %s\n" % (ref_name, fmt(model.q0)))


# ---- the shipped stacks, as data ---------------------------------------------------------------------------------

def _sc(name, tracked, radius, avoided, kp, kd=250.0):
    return dict(name=name, type="self-collision", tracked=tracked, radius=radius, avoided=avoided, kp=kp, kd=kd, margin=0.02, m=0.2)


def talos_stack() -> List[dict]:
    """/root/reference/etc/talos/tasks.yaml in file order."""
    big = lambda other, arm5: {other: 0.1, arm5: 0.05, "v_leg_right_3": 0.15, "v_leg_left_3": 0.15, "torso_2_joint": 0.1,
                               "base_link": 0.15, "v_base_link_left": 0.115, "v_base_link_right": 0.115, "leg_right_1_joint": 0.15,
                               "leg_left_1_joint": 0.15, "leg_right_4_joint": 0.1, "leg_left_4_joint": 0.1}
    small = {"base_link": 0.15, "v_base_link_left": 0.115, "v_base_link_right": 0.115}
    foot = dict(type="contact", kp=30.0, lxp=0.1, lxn=0.11, lyp=0.069, lyn=0.069, lz=0.107, fmin=5.0, fmax=1500.0, mu=0.3)
    return [
        dict(name="head", type="se3", tracked="head_1_joint", kp=1.0, mask="110000"),
        dict(name="head_pitch", type="se3", tracked="head_1_joint", kp=30.0, mask="000010"),
        dict(name="head_yaw", type="se3", tracked="head_2_joint", kp=30.0, mask="000001"),
        dict(name="lh", type="se3", tracked="gripper_left_joint", kp=30.0, mask="111111"),
        dict(name="rh", type="se3", tracked="gripper_right_joint", kp=30.0, mask="111111"),
        dict(name="torso", type="se3", tracked="torso_2_link", kp=30.0, mask="000110"),
        dict(name="lf", type="se3", tracked="leg_left_6_joint", kp=30.0, mask="111111"),
        dict(name="rf", type="se3", tracked="leg_right_6_joint", kp=30.0, mask="111111"),
        dict(name="com", type="com", kp=30.0, mask="111"),
        dict(name="posture", type="posture", kp=10.0),
        dict(name="momentum", type="momentum", kp=30.0, mask="000110"),
        dict(name="bounds", type="bounds"),
        dict(name="actuation_bounds", type="actuation-bounds"),
        dict(name="contact_lfoot", joint="leg_left_6_joint", **foot),
        dict(name="contact_rfoot", joint="leg_right_6_joint", **foot),
        _sc("self_collision-left", "gripper_left_joint", 0.1, big("gripper_right_joint", "arm_right_5_joint"), 50.0),
        _sc("self_collision-right", "gripper_right_joint", 0.1, big("gripper_left_joint", "arm_left_5_joint"), 50.0),
        _sc("self_collision-elbow-right", "arm_right_4_joint", 0.15, small, 50.0),
        _sc("self_collision-elbow-left", "arm_left_4_joint", 0.15, small, 150.0),
        _sc("self_collision-wrist-right", "arm_right_5_joint", 0.10, small, 50.0),
        _sc("self_collision-wrist-left", "arm_left_5_joint", 0.10, small, 150.0),
    ]


def icub_stack() -> List[dict]:
    """/root/reference/etc/icub/tasks.yaml in file order (virtual frames of etc/icub/frames.yaml)."""
    foot = dict(type="contact", kp=30.0, lxp=0.14, lxn=0.06, lyp=0.045, lyn=0.045, lz=0.065, fmin=5.0, fmax=1500.0, mu=0.3)
    av = lambda other: {other: 0.05, "v_leg_right": 0.08, "v_leg_left": 0.08, "torso_pitch": 0.11, "l_hip_pitch": 0.08,
                        "r_hip_pitch": 0.08, "l_knee": 0.08, "r_knee": 0.08}
    return [
        dict(name="lh", type="se3", tracked="l_hand", kp=30.0, mask="111111"),
        dict(name="rh", type="se3", tracked="r_hand", kp=30.0, mask="111111"),
        dict(name="lf", type="se3", tracked="left_foot", kp=30.0, mask="111111"),
        dict(name="rf", type="se3", tracked="right_foot", kp=30.0, mask="111111"),
        dict(name="com", type="com", kp=50.0, mask="111"),
        dict(name="momentum", type="momentum", kp=30.0, mask="000110"),
        dict(name="posture", type="posture", kp=10.0),
        dict(name="torso", type="se3", tracked="chest", kp=30.0, mask="000111"),
        dict(name="head", type="se3", tracked="head", kp=30.0, mask="110111"),
        dict(name="bounds", type="bounds"),
        dict(name="contact_lfoot", joint="l_ankle_roll", **foot),
        dict(name="contact_rfoot", joint="r_ankle_roll", **foot),
        _sc("self_collision-left", "l_wrist_yaw", 0.05, av("r_wrist_yaw"), 50.0),
        _sc("self_collision-right", "r_wrist_yaw", 0.05, av("l_wrist_yaw"), 50.0),
    ]


def icub_like(seed: int = 9) -> Model:
    """iCub's tree: free-flyer + 32 revolute joints (etc/icub/configurations.srdf:4-52: legs 6 + 6, torso 3, arms 7 + 7,
    neck 3), numbered depth-first.  Child-sized geometry (total 33 kg), NOT icub.urdf; the ankle-roll frames have z pointing
    DOWN, which is what the stack's contact normal (0, 0, -1) presumes (etc/icub/tasks.yaml:55-80); the reference posture is a
    crouch chosen for this geometry (feet flat), not the srdf's joint values, whose axis conventions belong to the real URDF."""
    rng = np.random.default_rng(seed)
    names, parent, jtype, place, inert = [], [], [], [], []

    def add(name, par, jt, xyz, mass, size, com, R=None):
        names.append(name)
        parent.append(par)
        jtype.append(jt)
        place.append(pack_se3(np.eye(3) if R is None else R, xyz))
        inert.append(_inertia(rng, mass, size, com))
        return len(names) - 1

    base = add("root_joint", -1, J_FREEFLYER, (0, 0, 0), 5.0, (0.15, 0.2, 0.12), (0.0, 0.0, 0.0))
    for side, sy in (("l", 1.0), ("r", -1.0)):
        j = add("%s_hip_pitch" % side, base, J_RY, (0.0, sy * 0.068, -0.12), 0.7, (0.06, 0.06, 0.06), (0, 0, 0))
        j = add("%s_hip_roll" % side, j, J_RX, (0, 0, 0), 0.5, (0.06, 0.06, 0.06), (0, 0, -0.01))
        j = add("%s_hip_yaw" % side, j, J_RZ, (0, 0, 0), 1.5, (0.09, 0.09, 0.2), (0, 0, -0.1))
        j = add("%s_knee" % side, j, J_RY, (0, 0, -0.22), 1.3, (0.08, 0.08, 0.2), (0, 0, -0.1))
        j = add("%s_ankle_pitch" % side, j, J_RY, (0, 0, -0.21), 0.3, (0.05, 0.05, 0.05), (0, 0, 0))
        add("%s_ankle_roll" % side, j, J_RX, (0, 0, 0), 0.6, (0.2, 0.09, 0.04), (0.03, 0.0, 0.04), R=_rot(0, np.pi))
    t = add("torso_pitch", base, J_RY, (0, 0, 0.05), 1.0, (0.1, 0.1, 0.08), (0, 0, 0.02))
    t = add("torso_roll", t, J_RX, (0, 0, 0.03), 1.0, (0.1, 0.1, 0.08), (0, 0, 0.02))
    chest = add("torso_yaw", t, J_RZ, (0, 0, 0.03), 7.0, (0.18, 0.25, 0.2), (0.0, 0.0, 0.1))
    for side, sy in (("l", 1.0), ("r", -1.0)):
        j = add("%s_shoulder_pitch" % side, chest, J_RY, (0.0, sy * 0.11, 0.16), 0.5, (0.06, 0.06, 0.06), (0, sy * 0.02, 0))
        j = add("%s_shoulder_roll" % side, j, J_RX, (0, sy * 0.02, 0), 0.3, (0.05, 0.05, 0.05), (0, 0, 0))
        j = add("%s_shoulder_yaw" % side, j, J_RZ, (0, 0, 0), 0.9, (0.06, 0.06, 0.15), (0, 0, -0.07))
        j = add("%s_elbow_joint" % side, j, J_RY, (0.0, 0, -0.15), 0.5, (0.05, 0.05, 0.12), (0, 0, -0.05))
        j = add("%s_wrist_prosup" % side, j, J_RZ, (0, 0, -0.07), 0.3, (0.05, 0.05, 0.08), (0, 0, -0.03))
        j = add("%s_wrist_pitch" % side, j, J_RY, (0, 0, -0.07), 0.1, (0.04, 0.04, 0.04), (0, 0, 0))
        add("%s_wrist_yaw" % side, j, J_RX, (0, 0, 0), 0.25, (0.06, 0.03, 0.1), (0, 0, -0.04))
    n = add("neck_pitch", chest, J_RY, (0, 0, 0.22), 0.2, (0.04, 0.04, 0.04), (0, 0, 0))
    n = add("neck_roll", n, J_RX, (0, 0, 0.02), 0.2, (0.04, 0.04, 0.04), (0, 0, 0))
    add("neck_yaw", n, J_RZ, (0, 0, 0.02), 1.3, (0.14, 0.14, 0.16), (0.01, 0, 0.07))
    nb = len(names)
    fnames, fbody = list(names), list(range(nb))
    fplace = [pack_se3(np.eye(3), (0, 0, 0)) for _ in range(nb)]

    def frame(name, body_name, xyz, R=None):
        fnames.append(name)
        fbody.append(names.index(body_name))
        fplace.append(pack_se3(np.eye(3) if R is None else R, xyz))

    frame("l_hand", "l_wrist_yaw", (0, 0, -0.06))
    frame("r_hand", "r_wrist_yaw", (0, 0, -0.06))
    frame("left_foot", "l_ankle_roll", (0.03, 0, 0.04))
    frame("right_foot", "r_ankle_roll", (0.03, 0, 0.04))
    frame("chest", "torso_yaw", (0, 0, 0.1))
    frame("head", "neck_yaw", (0, 0, 0.08))
    frame("v_leg_right", "r_hip_yaw", (0.0, -0.12, 0.0))  # etc/icub/frames.yaml
    frame("v_leg_left", "l_hip_yaw", (0.0, -0.12, 0.0))
    na = nb - 1
    leg = [-0.45, 0.0, 0.0, 0.9, -0.45, 0.0]  # crouch: hip + knee + ankle pitch = 0 keeps the sole level
    arm = [-0.6, 0.4, 0.0, 0.6, 0.0, 0.0, 0.0]
    q0 = np.array([0.0, 0.0, 0.5, 0.0, 0.0, 0.0, 1.0] + leg + leg + [0.0, 0.0, 0.0] + arm + [-0.6, -0.4, 0.0, 0.6, 0.0, 0.0, 0.0] + [0.0, 0.0, 0.0])
    m = Model(name="icub_like", floating_base=True, parent=np.array(parent, dtype=np.int32), jtype=np.array(jtype, dtype=np.int32),
              placement=np.stack(place), inertia=np.stack(inert), joint_names=names, frame_names=fnames,
              frame_body=np.array(fbody, dtype=np.int32), frame_placement=np.stack(fplace),
              q_lb=np.minimum(-rng.uniform(1.0, 2.0, na), q0[7:] - 0.3), q_ub=np.maximum(rng.uniform(1.0, 2.0, na), q0[7:] + 0.3),
              dq_max=rng.uniform(2.0, 8.0, na), tau_max=np.full(na, 40.0), q0=q0)
    m.validate()
    # stand on the ground: put the soles at z = 0
    _, pf = m.frame_placements(q0)
    m.q0[2] -= pf[m.frame("left_foot")][2] - 0.0
    return m


def franka_stack() -> List[dict]:
    """/root/reference/etc/franka/tasks.yaml."""
    return [dict(name="ee", type="se3", tracked="panda_joint7", kp=30.0, mask="111111"),
            dict(name="posture", type="posture", kp=30.0)]


def _inertia(rng, mass: float, size: Sequence[float], com: Sequence[float]) -> np.ndarray:
    """A box-like body of the given extents, slightly rotated: mass, com, I_c (xx xy xz yy yz zz)."""
    sx, sy, sz = size
    d = mass / 12.0 * np.array([sy * sy + sz * sz, sx * sx + sz * sz, sx * sx + sy * sy])
    w = 0.2 * rng.standard_normal(3)
    Rr = _rot(0, w[0]) @ _rot(1, w[1]) @ _rot(2, w[2])
    I = Rr @ np.diag(d) @ Rr.T
    return np.array([mass, com[0], com[1], com[2], I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]])


def talos_like(seed: int = 7) -> Model:
    """Talos' tree: free-flyer + 44 revolute joints in the order of etc/talos/configurations.srdf:4-48 (legs 6 + 6,
    torso 2, arms 7 + 7 each followed by its 7 gripper joints, head 2).  Link lengths, axes and masses are plausible
    humanoid values (total 95 kg), NOT those of talos.urdf, which is not available here."""
    rng = np.random.default_rng(seed)
    names, parent, jtype, place, inert = [], [], [], [], []

    def add(name, par, jt, xyz, mass, size, com, rpy=(0.0, 0.0, 0.0)):
        names.append(name)
        parent.append(par)
        jtype.append(jt)
        R = _rot(2, rpy[2]) @ _rot(1, rpy[1]) @ _rot(0, rpy[0])
        place.append(pack_se3(R, xyz))
        inert.append(_inertia(rng, mass, size, com))
        return len(names) - 1

    base = add("root_joint", -1, J_FREEFLYER, (0, 0, 0), 15.4, (0.25, 0.3, 0.2), (-0.05, 0.0, -0.03))
    for side, sy in (("left", 1.0), ("right", -1.0)):
        j = add("leg_%s_1_joint" % side, base, J_RZ, (-0.02, sy * 0.085, -0.27), 1.8, (0.1, 0.1, 0.1), (0.02, sy * 0.01, 0.03))
        j = add("leg_%s_2_joint" % side, j, J_RX, (0, 0, 0), 1.9, (0.12, 0.12, 0.12), (-0.01, sy * 0.0, -0.02))
        j = add("leg_%s_3_joint" % side, j, J_RY, (0, 0, 0), 6.2, (0.15, 0.15, 0.4), (0.0, sy * 0.05, -0.17))
        j = add("leg_%s_4_joint" % side, j, J_RY, (0, 0, -0.38), 3.8, (0.12, 0.12, 0.35), (0.02, sy * 0.02, -0.15))
        j = add("leg_%s_5_joint" % side, j, J_RY, (0, 0, -0.325), 1.3, (0.1, 0.1, 0.1), (-0.01, sy * 0.02, 0.02))
        add("leg_%s_6_joint" % side, j, J_RX, (0, 0, 0), 1.6, (0.21, 0.14, 0.06), (0.0, 0.0, -0.08))
    t1 = add("torso_1_joint", base, J_RZ, (0, 0, 0.0722), 3.0, (0.15, 0.2, 0.1), (0.0, 0.0, 0.03))
    t2 = add("torso_2_joint", t1, J_RY, (0, 0, 0), 17.5, (0.25, 0.35, 0.4), (-0.04, 0.0, 0.2), rpy=(0.0, 0.0, 0.0))
    for side, sy in (("left", 1.0), ("right", -1.0)):
        j = add("arm_%s_1_joint" % side, t2, J_RZ, (0.0, sy * 0.1575, 0.232), 1.4, (0.1, 0.1, 0.1), (-0.01, sy * 0.06, -0.02))
        j = add("arm_%s_2_joint" % side, j, J_RX, (0.005, sy * 0.13, -0.01), 1.7, (0.1, 0.1, 0.12), (0.02, sy * 0.01, -0.05),
                rpy=(0.0, 0.0, 0.0))
        j = add("arm_%s_3_joint" % side, j, J_RZ, (0.02, 0.0, -0.22), 1.7, (0.09, 0.09, 0.2), (0.0, 0.0, -0.1))
        j = add("arm_%s_4_joint" % side, j, J_RY, (-0.02, sy * -0.027, -0.05), 1.5, (0.09, 0.09, 0.18), (-0.01, 0.01, -0.08))
        j = add("arm_%s_5_joint" % side, j, J_RZ, (-0.004, 0.0, -0.19), 1.9, (0.08, 0.08, 0.15), (0.0, 0.0, 0.07),
                rpy=(0.0, 0.0, sy * 0.1))
        j = add("arm_%s_6_joint" % side, j, J_RX, (0, 0, -0.1), 0.4, (0.06, 0.06, 0.06), (0.0, 0.0, 0.0))
        w = add("arm_%s_7_joint" % side, j, J_RY, (0, 0, 0), 0.9, (0.08, 0.08, 0.1), (0.0, 0.0, -0.06))
        g = add("gripper_%s_inner_double_joint" % side, w, J_RY, (0.0, sy * 0.02, -0.12), 0.1, (0.03, 0.03, 0.06), (0, 0, -0.02))
        add("gripper_%s_fingertip_1_joint" % side, g, J_RY, (0.03, sy * 0.01, -0.06), 0.03, (0.02, 0.02, 0.04), (0, 0, -0.01))
        add("gripper_%s_fingertip_2_joint" % side, g, J_RY, (0.03, sy * -0.01, -0.06), 0.03, (0.02, 0.02, 0.04), (0, 0, -0.01))
        g = add("gripper_%s_inner_single_joint" % side, w, J_RY, (0.0, sy * -0.02, -0.12), 0.1, (0.03, 0.03, 0.06), (0, 0, -0.02))
        add("gripper_%s_fingertip_3_joint" % side, g, J_RY, (-0.03, 0.0, -0.06), 0.03, (0.02, 0.02, 0.04), (0, 0, -0.01))
        add("gripper_%s_joint" % side, w, J_RY, (0.0, 0.0, -0.09), 0.15, (0.04, 0.04, 0.06), (0, 0, -0.02))
        add("gripper_%s_motor_single_joint" % side, w, J_RY, (0.0, 0.0, -0.07), 0.1, (0.03, 0.03, 0.03), (0, 0, 0))
    h1 = add("head_1_joint", t2, J_RY, (0.0, 0.0, 0.316), 0.7, (0.1, 0.1, 0.1), (0.0, 0.0, 0.02))
    add("head_2_joint", h1, J_RZ, (0.0, 0.0, 0.0), 1.4, (0.18, 0.16, 0.2), (0.01, 0.0, 0.1))

    nb = len(names)
    # frames: one per joint (pinocchio adds a JOINT frame for each), the links and virtual frames the stack names
    # (etc/talos/frames.yaml)
    fnames = list(names)
    fbody = list(range(nb))
    fplace = [pack_se3(np.eye(3), (0, 0, 0)) for _ in range(nb)]

    def frame(name, body_name, xyz):
        fnames.append(name)
        fbody.append(names.index(body_name))
        fplace.append(pack_se3(np.eye(3), xyz))

    frame("base_link", "root_joint", (0, 0, 0))
    frame("torso_2_link", "torso_2_joint", (0, 0, 0))
    frame("v_leg_right_3", "leg_right_3_joint", (0.0, -0.1, -0.2))
    frame("v_leg_left_3", "leg_left_3_joint", (0.0, 0.1, -0.2))
    frame("v_base_link_left", "root_joint", (0.0, -0.1, 0.0))
    frame("v_base_link_right", "root_joint", (0.0, 0.1, 0.0))
    na = nb - 1
    q_hi = rng.uniform(1.2, 2.6, na)
    q0 = np.array([-0.0374385, -5.73035e-05, 1.051606, 2.2858e-06, -0.0328617, 6.66461e-05, 0.99946,
                   -0.000138691, 7.6436e-05, -0.233079, 0.611222, -0.312407, -8.55702e-05,
                   -0.000138691, 7.64337e-05, -0.233048, 0.611226, -0.312443, -8.55679e-05,
                   -0.000162128, 0.0973496,
                   0.218615, 0.566335, -0.582134, -1.4169, 0.315339, 0.0615923, -0.146577, -0.2, 0.2, 0.2, 0.2, 0.2, -0.2, 0.2,
                   -0.218505, -0.566321, 0.582227, -1.41689, -0.315343, -0.0617214, -0.146588, -0.2, 0.2, 0.2, 0.2, 0.2, -0.2, 0.2,
                   -0.0041334, 1.98607e-05])  # etc/talos/configurations.srdf:4-48 (inria_start)
    q0[3:7] /= np.linalg.norm(q0[3:7])
    m = Model(name="talos_like", floating_base=True, parent=np.array(parent, dtype=np.int32), jtype=np.array(jtype, dtype=np.int32),
              placement=np.stack(place), inertia=np.stack(inert), joint_names=names, frame_names=fnames,
              frame_body=np.array(fbody, dtype=np.int32), frame_placement=np.stack(fplace),
              q_lb=np.minimum(-q_hi, q0[7:] - 0.3), q_ub=np.maximum(q_hi, q0[7:] + 0.3), dq_max=rng.uniform(2.0, 10.0, na),
              tau_max=np.concatenate([np.full(12, 400.0), np.full(2, 200.0), np.full(na - 14, 100.0)]), q0=q0)
    m.validate()
    return m


def franka_like(seed: int = 11) -> Model:
    """Seven revolute joints and two prismatic fingers (etc/franka/configurations.srdf:4-12), fixed base."""
    rng = np.random.default_rng(seed)
    names, parent, jtype, place, inert = [], [], [], [], []
    spec = [("panda_joint1", J_RZ, (0, 0, 0.333), (0, 0, 0)), ("panda_joint2", J_RZ, (0, 0, 0), (-np.pi / 2, 0, 0)),
            ("panda_joint3", J_RZ, (0, -0.316, 0), (np.pi / 2, 0, 0)), ("panda_joint4", J_RZ, (0.0825, 0, 0), (np.pi / 2, 0, 0)),
            ("panda_joint5", J_RZ, (-0.0825, 0.384, 0), (-np.pi / 2, 0, 0)), ("panda_joint6", J_RZ, (0, 0, 0), (np.pi / 2, 0, 0)),
            ("panda_joint7", J_RZ, (0.088, 0, 0), (np.pi / 2, 0, 0))]
    for i, (nm, jt, xyz, rpy) in enumerate(spec):
        names.append(nm)
        parent.append(i - 1)
        jtype.append(jt)
        place.append(pack_se3(_rot(2, rpy[2]) @ _rot(1, rpy[1]) @ _rot(0, rpy[0]), xyz))
        inert.append(_inertia(rng, rng.uniform(1.0, 4.0), (0.1, 0.1, 0.15), 0.03 * rng.standard_normal(3)))
    for k, sy in enumerate((1.0, -1.0)):
        names.append("panda_finger_joint%d" % (k + 1))
        parent.append(6)
        jtype.append(J_PY)
        place.append(pack_se3(_rot(2, 0.0 if k == 0 else np.pi), (0, 0, 0.1654)))
        inert.append(_inertia(rng, 0.015, (0.02, 0.02, 0.05), (0, sy * 0.01, 0.02)))
    nb = len(names)
    q0 = np.array([0.0, np.pi / 4, 0.0, -np.pi / 4, 0.0, np.pi / 2, 0.0, 0.0, 0.0])
    m = Model(name="franka_like", floating_base=False, parent=np.array(parent, dtype=np.int32), jtype=np.array(jtype, dtype=np.int32),
              placement=np.stack(place), inertia=np.stack(inert), joint_names=names, frame_names=list(names),
              frame_body=np.arange(nb, dtype=np.int32), frame_placement=np.stack([pack_se3(np.eye(3), (0, 0, 0))] * nb),
              q_lb=np.full(nb, -2.8), q_ub=np.full(nb, 2.8), dq_max=np.full(nb, 2.2), tau_max=np.full(nb, 87.0), q0=q0)
    m.validate()
    return m


def random_tree(seed: int, nbody: int = 24, floating_base: bool = True, nframe: int = 10) -> Model:
    """A random branching tree with every joint type and arbitrary placement rotations: the parity tests' stress model."""
    rng = np.random.default_rng(seed)
    parent = [-1]
    for i in range(1, nbody):
        parent.append(i - 1 if rng.random() < 0.7 else int(rng.integers(0, i)))
    # renumber depth-first so that subtrees are contiguous
    children: Dict[int, List[int]] = {i: [] for i in range(nbody)}
    for i in range(1, nbody):
        children[parent[i]].append(i)
    order: List[int] = []

    def visit(i):
        order.append(i)
        for c in children[i]:
            visit(c)

    visit(0)
    new = {old: k for k, old in enumerate(order)}
    parent = [(-1 if parent[old] < 0 else new[parent[old]]) for old in order]
    jtype = [J_FREEFLYER if (i == 0 and floating_base) else int(rng.integers(J_RX, J_PZ + 1)) for i in range(nbody)]
    place, inert = [], []
    for i in range(nbody):
        w = rng.standard_normal(3)
        R = _rot(0, w[0]) @ _rot(1, w[1]) @ _rot(2, w[2])
        place.append(pack_se3(np.eye(3) if (i == 0 and floating_base) else R, np.zeros(3) if (i == 0 and floating_base) else 0.3 * rng.standard_normal(3)))
        inert.append(_inertia(rng, rng.uniform(0.2, 8.0), rng.uniform(0.05, 0.4, 3), 0.1 * rng.standard_normal(3)))
    fb = rng.integers(0, nbody, nframe).astype(np.int32)
    fp = []
    for _ in range(nframe):
        w = rng.standard_normal(3)
        fp.append(pack_se3(_rot(0, w[0]) @ _rot(1, w[1]) @ _rot(2, w[2]), 0.2 * rng.standard_normal(3)))
    nq = nbody + (6 if floating_base else 0)
    na = nbody - (1 if floating_base else 0)
    q0 = 0.5 * rng.standard_normal(nq)
    if floating_base:
        q0[3:7] = rng.standard_normal(4)
        q0[3:7] /= np.linalg.norm(q0[3:7])
    m = Model(name="random_tree_%d" % seed, floating_base=floating_base, parent=np.array(parent, dtype=np.int32),
              jtype=np.array(jtype, dtype=np.int32), placement=np.stack(place), inertia=np.stack(inert),
              joint_names=["j%d" % i for i in range(nbody)], frame_names=["f%d" % i for i in range(nframe)], frame_body=fb,
              frame_placement=np.stack(fp), q_lb=np.full(na, -2.5), q_ub=np.full(na, 2.5), dq_max=rng.uniform(2.0, 10.0, na),
              tau_max=np.full(na, 100.0), q0=q0)
    m.validate()
    return m


# ---- synthetic states and references -----------------------------------------------------------------------------

def sample_states(model: Model, tm: TaskMap, batch: int, seed: int, q_noise: float = 0.05, v_noise: float = 0.2,
                  ref_noise: float = 0.01, com_offset: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """`batch` robot states around the reference configuration and the references a freshly constructed controller holds:
    every SE(3) / CoM / contact reference is the placement at q0 (tasks.cpp:64-80,109,361), moved by `ref_noise` (a tracking
    target slightly away), the posture reference is q0's actuated part (tasks.cpp:217).  Instance i depends on seed + i only."""
    nq, nv, na = model.nq, model.nv, model.na
    q = np.zeros((batch, nq))
    v = np.zeros((batch, nv))
    ref = np.zeros((batch, tm.nref))
    Rf0, pf0 = model.frame_placements(model.q0)
    com0 = model.com(model.q0)
    for i in range(batch):
        rng = np.random.default_rng(seed + i)
        qi = model.q0.copy()
        if model.floating_base:
            qi[0:3] += q_noise * 0.2 * rng.standard_normal(3)
            dq = np.concatenate([0.5 * q_noise * 0.2 * rng.standard_normal(3), [1.0]])
            x, y, z, w = qi[3:7]
            a, b, c, d = dq / np.linalg.norm(dq)
            qi[3:7] = [w * a + x * d + y * c - z * b, w * b - x * c + y * d + z * a, w * c + x * b - y * a + z * d,
                       w * d - x * a - y * b - z * c]
            qi[7:] += q_noise * rng.standard_normal(na)
        else:
            qi += q_noise * rng.standard_normal(na)
        q[i] = qi
        v[i] = v_noise * rng.standard_normal(nv)
        r = ref[i]
        for b in tm.blocks:
            if b.kind == T_SE3:
                w3 = ref_noise * rng.standard_normal(3)
                Rr = Rf0[b.frame] @ _rot(0, w3[0]) @ _rot(1, w3[1]) @ _rot(2, w3[2])
                r[b.ref:b.ref + 12] = se3_ref(Rr, pf0[b.frame] + ref_noise * rng.standard_normal(3))
                r[b.ref + 12:b.ref + 24] = ref_noise * rng.standard_normal(12)
            elif b.kind == T_COM:
                r[b.ref:b.ref + 3] = com0 + ref_noise * rng.standard_normal(3) + (0.0 if com_offset is None else com_offset[i, 0:3])
                if com_offset is not None:
                    r[b.ref + 3:b.ref + 9] = com_offset[i, 3:9]
            elif b.kind == T_MOMENTUM:
                r[b.ref:b.ref + 12] = 0.0  # tasks.cpp:144-145: zero reference
        if tm.sel_col.size:
            r[tm.posture_ref:tm.posture_ref + na] = model.q0[nq - na:]
        for c in range(tm.ncontact):
            f = tm.contact_frame[c]
            r[tm.contact_ref[c]:tm.contact_ref[c] + 12] = se3_ref(Rf0[f], pf0[f])
    return dict(q=q, v=v, ref=ref)


def random_stack(model: Model, seed: int, n_contacts: int = 2) -> Tuple[Structure, List[dict]]:
    """A task stack over `model`'s frames that uses every kind of task with random masks and gains (stress input for the
    parity tests; not a shipped inria_wbc stack)."""
    from . import structure as S
    rng = np.random.default_rng(seed)
    fr = lambda: model.frame_names[int(rng.integers(0, model.nframe))]
    mask = lambda n: "".join("1" if (rng.random() < 0.7 or i == k) else "0" for k in [int(rng.integers(0, n))] for i in range(n))
    stack: List[dict] = []
    dense = []
    for t in range(4):
        mk = mask(6)
        stack.append(dict(name="se3_%d" % t, type="se3", tracked=fr(), kp=float(rng.uniform(1, 50)), mask=mk))
        dense.append(("se3_%d" % t, mk.count("1"), float(rng.uniform(1, 100))))
    mk = mask(3)
    stack.append(dict(name="com", type="com", kp=30.0, mask=mk))
    dense.append(("com", mk.count("1"), 1000.0))
    stack.append(dict(name="posture", type="posture", kp=10.0))
    dense.append(("__posture__", "posture", 0.5))
    mk = mask(6)
    stack.append(dict(name="momentum", type="momentum", kp=20.0, mask=mk))
    dense.append(("momentum", mk.count("1"), 10.0))
    stack.append(dict(name="bounds", type="bounds"))
    contacts = []
    if model.floating_base:
        pts = S.contact6d_points(lxn=0.06, lyn=0.045, lxp=0.14, lyp=0.045, lz=0.065)
        for c in range(n_contacts):
            contacts.append(S.Contact("contact_%d" % c, pts, (0.0, 0.0, 1.0), 0.4, 5.0, 1200.0))
            stack.append(dict(name="contact_%d" % c, type="contact", joint=fr(), kp=30.0))
        dense.append(("__contacts__",))
    sc = []
    for t in range(3):
        av = {}
        tracked = fr()
        while len(av) < 1 + 2 * t:
            f = fr()
            if f != tracked:
                av[f] = float(rng.uniform(0.05, 0.3))
        stack.append(dict(name="sc_%d" % t, type="self-collision", tracked=tracked, radius=float(rng.uniform(0.05, 0.2)), avoided=av,
                          kp=50.0, kd=250.0, margin=0.02 + 0.1 * t, m=0.2 + 0.3 * t))
        sc.append(("sc_%d" % t, 500.0))
    level0 = [(S.INEQ_BOUNDS, 0)] + [(S.INEQ_FORCE, c) for c in range(len(contacts))]
    st = S._mk("stack_" + model.name, model.nv, model.na, contacts, dense, None, sc, True, False, level0)
    return st, stack
