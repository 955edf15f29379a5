"""Synthetic whole-body QP batches (the reference's URDFs and pinocchio are unavailable, so the
"step before the path" -- M, h, task rows, contact Jacobians -- is drawn at random with
humanoid-like scales; SURVEY.md 8(d) describes the recipe).

Every QP is feasible by construction: a point (dv*, f*) is drawn that satisfies the base dynamics,
the contact motion constraints, the friction pyramids and all bounds, and the right-hand sides are
derived from it. A controlled share of torque / acceleration limits is made tight around that point
so that the active-set loop has work to do.

QP i of a batch depends only on (seed_base + i): permuting or sharding a batch leaves each QP unchanged.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .structure import Structure, contact6d_friction, cop_rows
from .trajs import move_com_stream

FIELDS = ("M", "h", "A", "b1", "Ac", "bc", "blb", "bub", "tlb", "tub", "w", "Acop")

SEED_BASE = {"franka": 1_000_000, "talos": 2_000_000, "icub": 3_000_000, "talos_squat": 4_000_000, "ragged": 5_000_000,
             "tiago": 6_000_000, "talos_single_support": 7_000_000, "three_contact": 8_000_000,
             "talos_torque": 9_000_000, "talos_cop": 10_000_000, "talos_torque_cop": 11_000_000, "icub_torque": 12_000_000}


def _spd_mass(rng, nv: int, nu: int) -> np.ndarray:
    G = rng.standard_normal((nv, nv))
    C0 = G @ G.T / nv + 0.3 * np.eye(nv)
    dg = np.sqrt(np.diag(C0))
    Cm = C0 / np.outer(dg, dg)
    scale = np.exp(rng.uniform(np.log(0.1), np.log(50.0), nv))
    if nu:
        scale[:3] = rng.uniform(60.0, 100.0)  # total mass on the base translation
        scale[3:nu] = rng.uniform(5.0, 25.0, nu - 3)  # base rotational inertia
    D = np.sqrt(scale)
    M = Cm * np.outer(D, D)
    return 0.5 * (M + M.T)


def _jac_rows(rng, rows: int, nv: int, nu: int, sparsity: float = 0.6) -> np.ndarray:
    A = 0.5 * rng.standard_normal((rows, nv))
    if nv > nu:
        mask = rng.random((rows, nv - nu)) >= sparsity
        A[:, nu:] *= mask
    return A


def _skew(p):
    x, y, z = p
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def _foot_jacobian(rng, c: int, nv: int, nu: int) -> np.ndarray:
    """Local-frame 6 x nv Jacobian of a foot frame of a floating-base biped: the base block is the
    rigid-body adjoint [R' , -R' skew(p); 0, R'] of a foot ~0.9 m below the base, the six joints of
    that leg contribute [axis x r; axis] columns, every other joint column is zero."""
    side = 1.0 if c % 2 == 0 else -1.0
    p = np.array([rng.uniform(-0.03, 0.03), side * rng.uniform(0.07, 0.10), -rng.uniform(0.80, 0.95)])
    w = 0.05 * rng.standard_normal(3)
    R, _ = np.linalg.qr(np.eye(3) + _skew(w))
    R = R * np.sign(np.diag(R))  # proper rotation close to identity
    J = np.zeros((6, nv))
    J[0:3, 0:3] = R.T
    J[0:3, 3:6] = -R.T @ _skew(p)
    J[3:6, 3:6] = R.T
    for j in range(6):
        axis = rng.standard_normal(3)
        axis /= np.linalg.norm(axis)
        r = rng.uniform(0.05, 0.9) * np.array([0.1, 0.1, 1.0]) * rng.standard_normal(3)
        J[0:3, nu + 6 * c + j] = R.T @ np.cross(axis, r)
        J[3:6, nu + 6 * c + j] = R.T @ axis
    return J


def generate_one(st: Structure, seed: int, p_act: float = 0.10, p_bnd: float = 0.05, task_noise: float = 0.5,
                 com_ref: Optional[np.ndarray] = None, weight_jitter: float = 0.0, torque_ref_noise: float = 0.0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    nv, na, nu, nc = st.nv, st.na, st.nu, st.nc
    L = st.field_lengths()

    M = _spd_mass(rng, nv, nu)
    A = _jac_rows(rng, st.n_dense, nv, nu)
    Ac = np.zeros((nc, 6, nv))
    for c in range(nc):
        if nu == 6 and na >= 6 * nc:
            Ac[c] = _foot_jacobian(rng, c, nv, nu)
        else:
            Ac[c] = _jac_rows(rng, 6, nv, nu)
    # lf / rf SE3 tasks track the contact frames (etc/talos/tasks.yaml:37-48 vs :70-95): share the Jacobian
    names = st.task_names
    for c, contact in enumerate(st.contacts):
        short = {"contact_lfoot": "lf", "contact_rfoot": "rf"}.get(contact.name)
        if short in names:
            t = names.index(short)
            rows = np.where(st.dense_row_task == t)[0]
            if rows.size == 6:
                A[rows] = Ac[c]

    # ---- feasible point ------------------------------------------------------------------
    dv = 0.5 * rng.standard_normal(nv)
    f = np.zeros(12 * nc)
    for c, contact in enumerate(st.contacts):
        n = np.asarray(contact.normal)
        t1 = np.cross(n, [1.0, 0.0, 0.0])
        if np.linalg.norm(t1) < 1e-5:
            t1 = np.cross(n, [0.0, 1.0, 0.0])
        t2 = np.cross(n, t1)
        t1 /= np.linalg.norm(t1)
        t2 /= np.linalg.norm(t2)
        for p in range(4):
            fz = rng.uniform(60.0, 160.0)
            a, b = rng.uniform(-0.3, 0.3, 2) * contact.mu * fz
            f[12 * c + 3 * p:12 * c + 3 * p + 3] = fz * n + a * t1 + b * t2
    T = st.force_gen()
    Jc = np.concatenate([T[c].T @ Ac[c] for c in range(nc)], axis=0) if nc else np.zeros((0, nv))  # (12nc, nv)
    h = 5.0 * rng.standard_normal(nv)
    if nu:
        h[:nu] = -(M[:nu] @ dv - Jc[:, :nu].T @ f)  # base dynamics hold at (dv*, f*)
    bc = np.einsum("crj,j->cr", Ac, dv) if nc else np.zeros((0, 6))
    tau = M[nu:] @ dv + h[nu:] - (Jc[:, nu:].T @ f if nc else 0.0)

    # ---- level-0 bounds ------------------------------------------------------------------
    dt = 1e-3
    nb = st.n_bound
    dq_max = rng.uniform(2.0, 10.0, nb)
    blb, bub = -dq_max / dt, dq_max / dt  # tasks.cpp:288-291: ddq_max = dq_max / dt
    if nb:
        tight = rng.random(nb) < p_bnd
        dvb = dv[st.bound_col]
        blb = np.where(tight, dvb - rng.uniform(0.01, 0.3, nb), blb)
        bub = np.where(tight, dvb + rng.uniform(0.01, 0.3, nb), bub)
    if st.act_bounds:
        tight = rng.random(na) < p_act
        tmax = np.where(tight, np.abs(tau) * (1.0 + rng.uniform(0.02, 0.5, na)) + 1e-3, np.abs(tau) + rng.uniform(20.0, 200.0, na))
        tlb, tub = -tmax, tmax  # tasks.cpp:315-316: setBounds(-tau_max, tau_max)
    else:
        tlb = tub = np.zeros(0)

    # ---- level-1 right-hand sides and weights --------------------------------------------
    # near-equilibrium tracking: desired task accelerations = what (dv*) already produces + a correction
    b1 = np.zeros(st.r1)
    b1[:st.n_dense] = A @ dv + task_noise * rng.standard_normal(st.n_dense)
    b1[st.n_dense:st.n_dense + st.n_sel] = dv[st.sel_col] + task_noise * rng.standard_normal(st.n_sel)
    # force-regularisation rhs = diag(w_f) f_ref with f_ref = 0 unless the stabiliser sets one
    if com_ref is not None and "com" in names:
        rows = np.where(st.dense_row_task == names.index("com"))[0]
        b1[rows] += com_ref[:rows.size]
    # torque task (tasks.cpp:227-271): the reference torque is zero there (:262-263); `torque_ref_noise` exercises the general rhs
    # S tau_ref.  cop task (tasks.cpp:156-178): rows from the contact frames' placements in the world, reference point (0, 0, 0), rhs 0.
    # (drawn from a generator of their own so that the rest of the record is the plain stack's)
    Acop = np.zeros(0)
    if st.n_acteq or st.cop_task >= 0:
        rng2 = np.random.default_rng(seed + 0x5eed)
        o = st.n_dense + st.n_sel + 6 * nc
        if st.n_acteq:
            b1[o:o + st.n_acteq] = st.acteq_scale * torque_ref_noise * rng2.standard_normal(st.n_acteq)
        if st.cop_task >= 0:
            placements = []
            for c in range(nc):
                side = 1.0 if c % 2 == 0 else -1.0
                wv = 0.05 * rng2.standard_normal(3)
                Rq, _ = np.linalg.qr(np.eye(3) + _skew(wv))
                Rq = Rq * np.sign(np.diag(Rq))
                placements.append((Rq, np.array([rng2.uniform(-0.05, 0.05), side * rng2.uniform(0.07, 0.10), rng2.uniform(-0.01, 0.01)])))
            Acop = cop_rows(st, placements).reshape(-1)
    w = st.default_weights.copy()
    if weight_jitter > 0.0:
        w = w * np.exp(rng.uniform(-weight_jitter, weight_jitter, w.size))

    iu = np.tril_indices(nv)
    out = dict(M=M[iu], h=h, A=A.reshape(-1), b1=b1, Ac=Ac.reshape(-1), bc=bc.reshape(-1),
               blb=blb, bub=bub, tlb=tlb, tub=tub, w=w, Acop=Acop)
    for k_ in FIELDS:
        assert out[k_].size == L[k_], (k_, out[k_].size, L[k_])
    out["_dv_star"] = dv
    out["_f_star"] = f
    return out


def squat_com_rhs(st: Structure, tick: int, kp: float = 30.0, dt: float = 1e-3) -> np.ndarray:
    """CoM task rhs when the CoM sits at its start while the reference follows etc/talos/squat.yaml
    (move_com.cpp:8-61): b = Kp (x_ref - x) + Kd (v_ref - v) + a_ref, Kd = 2 sqrt(Kp) (tasks.cpp:106-107)."""
    pos, vel, acc = _squat_tables(dt)
    i = tick % pos.shape[0]
    kd = 2.0 * np.sqrt(kp)
    return kp * pos[i] + kd * vel[i] + acc[i]


_SQUAT_CACHE = {}


def _squat_tables(dt: float):
    if dt not in _SQUAT_CACHE:
        _SQUAT_CACHE[dt] = move_com_stream([0.0, 0.0, 0.0], [[0.0, 0.0, -0.2]], "001", dt, 2.0, loop=True, absolute=False)
    return _SQUAT_CACHE[dt]


def generate(st: Structure, batch: int, seed_base: int, first: int = 0, squat: bool = False, dtype=np.float64,
             **kw) -> Dict[str, np.ndarray]:
    """[batch, len] arrays for QPs first .. first+batch-1 of the stream `seed_base`."""
    L = st.field_lengths()
    out = {k: np.zeros((batch, L[k]), dtype=np.float64) for k in FIELDS}
    for i in range(batch):
        gi = first + i
        com_ref = squat_com_rhs(st, gi % 4000, st.kp.get("com", 30.0)) if squat else None
        one = generate_one(st, seed_base + gi, com_ref=com_ref, **kw)
        for k in FIELDS:
            out[k][i] = one[k]
    if dtype != np.float64:
        out = {k: v.astype(dtype) for k, v in out.items()}
    return out
