"""The rigid-body / task-law oracle (oracle/rbd_oracle.c) against identities that do not depend on it being a faithful
restatement: a second recursion, finite differences of the forward kinematics, energy conservation, closed-loop decay.
CPU only."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from inria_wbc_amd import model as mdl
from inria_wbc_amd import structure


@pytest.fixture(scope="module")
def rbd():
    from oracle import rbd as r
    return r


@pytest.fixture(scope="module")
def oracle_mod():
    from oracle import oracle as o
    return o


MODELS = {"talos": lambda: mdl.talos_like(), "icub": lambda: mdl.icub_like(), "franka": lambda: mdl.franka_like(), "tree_fb": lambda: mdl.random_tree(3, 24, True),
          "tree_fixed": lambda: mdl.random_tree(4, 17, False)}


def _state(m, seed):
    rng = np.random.default_rng(seed)
    q = m.q0.copy()
    q[(7 if m.floating_base else 0):] += 0.3 * rng.standard_normal(m.na)
    if m.floating_base:
        q[0:3] += rng.standard_normal(3)
        q[3:7] += 0.3 * rng.standard_normal(4)
        q[3:7] /= np.linalg.norm(q[3:7])
    return q, 0.5 * rng.standard_normal(m.nv)


def _advance(oracle_mod, m, q, v, eps):
    """q (+) eps v with pinocchio's integrate (the rank-2 oracle)."""
    return oracle_mod.integrate(m.floating_base, eps, q[None], v[None], np.zeros((1, m.nv)))["q_next"][0]


@pytest.mark.parametrize("name", list(MODELS))
def test_crba_nle_against_rnea(rbd, name):
    m = MODELS[name]()
    q, v = _state(m, 1)
    t = rbd.rbd_terms(m, q, v)
    assert np.abs(t["M"] - t["M"].T).max() == 0.0
    assert np.linalg.eigvalsh(t["M"]).min() > 0.0
    for s in range(3):
        a = np.random.default_rng(10 + s).standard_normal(m.nv)
        tau = rbd.rnea(m, q, v, a)
        assert np.abs(t["M"] @ a + t["nle"] - tau).max() < 1e-10 * max(1.0, np.abs(tau).max())


@pytest.mark.parametrize("name", list(MODELS))
def test_placements_against_numpy_fk(rbd, name):
    m = MODELS[name]()
    q, v = _state(m, 2)
    t = rbd.rbd_terms(m, q, v)
    Rf, pf = m.frame_placements(q)
    assert np.abs(t["oMf"][:, 9:] - pf).max() < 1e-13
    assert np.abs(t["oMf"][:, :9].reshape(-1, 3, 3) - Rf).max() < 1e-13
    assert np.abs(t["com"] - m.com(q)).max() < 1e-13


@pytest.mark.parametrize("name", list(MODELS))
def test_jacobians_and_drifts_by_finite_differences(rbd, oracle_mod, name):
    m = MODELS[name]()
    q, v = _state(m, 3)
    t = rbd.rbd_terms(m, q, v)
    eps = 1e-6
    qp, qm = _advance(oracle_mod, m, q, v, eps), _advance(oracle_mod, m, q, v, -eps)
    tp, tm_ = rbd.rbd_terms(m, qp, v), rbd.rbd_terms(m, qm, v)
    R = t["oMf"][:, :9].reshape(-1, 3, 3)
    p = t["oMf"][:, 9:]
    # velocities: J v, and the derivative of the placement along v
    assert np.abs(np.einsum("fij,j->fi", t["Jl"], v) - t["vf"]).max() < 1e-12
    dp = (tp["oMf"][:, 9:] - tm_["oMf"][:, 9:]) / (2 * eps)
    assert np.abs(dp - np.einsum("fij,fj->fi", R, t["vf"][:, :3])).max() < 1e-7
    dR = (tp["oMf"][:, :9] - tm_["oMf"][:, :9]).reshape(-1, 3, 3) / (2 * eps)
    for f in range(m.nframe):
        W = R[f].T @ dR[f]  # = skew(w_local)
        assert np.abs(np.array([W[2, 1], W[0, 2], W[1, 0]]) - t["vf"][f, 3:]).max() < 1e-7
    # WORLD Jacobian: velocity of the point of the body that sits at the world origin
    ww = np.einsum("fij,fj->fi", R, t["vf"][:, 3:])
    lin0 = np.einsum("fij,fj->fi", R, t["vf"][:, :3]) - np.cross(ww, p)
    Jwv = np.einsum("fij,j->fi", t["Jw"], v)
    assert np.abs(Jwv[:, :3] - lin0).max() < 1e-12 and np.abs(Jwv[:, 3:] - ww).max() < 1e-12
    # classical acceleration at ddq = 0: derivative of the world velocity of the frame origin, in local axes
    Rp, Rm = tp["oMf"][:, :9].reshape(-1, 3, 3), tm_["oMf"][:, :9].reshape(-1, 3, 3)
    dvw = (np.einsum("fij,fj->fi", Rp, tp["vf"][:, :3]) - np.einsum("fij,fj->fi", Rm, tm_["vf"][:, :3])) / (2 * eps)
    dww = (np.einsum("fij,fj->fi", Rp, tp["vf"][:, 3:]) - np.einsum("fij,fj->fi", Rm, tm_["vf"][:, 3:])) / (2 * eps)
    scale = max(1.0, np.abs(t["af"]).max())
    assert np.abs(np.einsum("fji,fj->fi", R, dvw) - t["af"][:, :3]).max() < 1e-6 * scale
    assert np.abs(np.einsum("fji,fj->fi", R, dww) - t["af"][:, 3:]).max() < 1e-6 * scale
    # centre of mass and centroidal momentum
    mass = m.inertia[:, 0].sum()
    assert np.abs(t["Jcom"] @ v - t["vcom"]).max() < 1e-12
    assert np.abs((tp["com"] - tm_["com"]) / (2 * eps) - t["vcom"]).max() < 1e-7
    assert np.abs((tp["vcom"] - tm_["vcom"]) / (2 * eps) - t["acom"]).max() < 1e-6 * max(1.0, np.abs(t["acom"]).max())
    assert np.abs(t["Ag"][:3] / mass - t["Jcom"]).max() < 1e-12
    dh = (tp["Ag"] @ v - tm_["Ag"] @ v) / (2 * eps)
    assert np.abs(dh - t["dAgv"]).max() < 1e-6 * max(1.0, np.abs(t["dAgv"]).max())
    assert np.abs(t["dAgv"][:3] - mass * t["acom"]).max() < 1e-9 * max(1.0, mass * np.abs(t["acom"]).max())


@pytest.mark.parametrize("name", list(MODELS))
def test_free_dynamics_conserve_energy(rbd, oracle_mod, name):
    """a = -M^-1 nle is the unforced motion: d/dt (kinetic + potential) = v' (Mdot/2 - C) v = 0."""
    m = MODELS[name]()
    q, v = _state(m, 4)
    t = rbd.rbd_terms(m, q, v)
    a = -np.linalg.solve(t["M"], t["nle"])
    eps = 1e-6
    ep = rbd.energy(m, _advance(oracle_mod, m, q, v, eps), v + eps * a)
    em = rbd.energy(m, _advance(oracle_mod, m, q, v, -eps), v - eps * a)
    power_scale = np.abs(v * t["nle"]).sum()
    assert abs(ep - em) / (2 * eps) < 1e-5 * power_scale
    assert abs(rbd.energy(m, q, v) - (0.5 * v @ t["M"] @ v - sum(
        m.inertia[i, 0] * np.dot(m.gravity, (m.body_placements(q)[0][i] @ m.inertia[i, 1:4] + m.body_placements(q)[1][i])) for i in range(m.nbody)))) < 1e-9


def test_log3_against_scipy(rbd):
    rng = np.random.default_rng(5)
    for k in range(200):
        w = rng.standard_normal(3)
        w *= rng.uniform(0, np.pi - 1e-3) / np.linalg.norm(w)
        if k % 10 == 0:
            w *= 1e-6  # Taylor branch
        if k % 10 == 1:
            w *= (np.pi - 1e-4) / np.linalg.norm(w)  # near-pi branch
        R = Rotation.from_rotvec(w).as_matrix()
        assert np.abs(rbd.log3(R) - w).max() < 2e-7 if k % 10 == 1 else np.abs(rbd.log3(R) - w).max() < 1e-10, (k, w, rbd.log3(R))


def _talos_setup():
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    return m, st, tm


def test_taskmap_matches_structure():
    m, st, tm = _talos_setup()
    assert tm.n_dense == st.n_dense == 41 and tm.ncontact == 2 and tm.n_bound == 44
    assert tm.nref == 8 * 24 + 9 + 12 + 44 + 2 * 24
    m2 = mdl.franka_like()
    st2 = structure.franka_structure()
    tm2 = mdl.build_taskmap(m2, st2, mdl.franka_stack())
    assert tm2.n_dense == 6 and tm2.nref == 24 + 9
    with pytest.raises(KeyError):
        mdl.build_taskmap(m2, st2, [dict(name="ee", type="se3", tracked="nope", kp=1.0, mask="111111")])


def test_icub_stack_stands_still_at_its_reference_posture(rbd, oracle_mod):
    """Static equilibrium through the whole oracle pipeline: at rest in the reference posture, with every reference where the
    robot is, the QP must ask for (almost) no acceleration and for contact forces that carry the weight -- along -z of the
    ankle frames, the normal etc/icub/tasks.yaml:55-80 gives."""
    m = mdl.icub_like()
    st = structure.icub_structure()
    tm = mdl.build_taskmap(m, st, mdl.icub_stack())
    s = mdl.sample_states(m, tm, 1, 0, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    rows = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"])
    out = oracle_mod.tick_batch(st, dict(rows, tlb=np.zeros((1, 0)), tub=np.zeros((1, 0)), w=st.default_weights[None]))
    assert out["status"][0] == 0
    assert np.abs(out["x"][0, :m.nv]).max() < 1e-2
    fz = out["x"][0, m.nv:].reshape(8, 3)[:, 2]
    assert np.all(fz < 0.0) and abs(fz.sum() + m.inertia[:, 0].sum() * 9.81) < 1e-3 * m.inertia[:, 0].sum() * 9.81


def test_task_rows_reuse_the_terms(rbd):
    m, st, tm = _talos_setup()
    s = mdl.sample_states(m, tm, 3, 100)
    rows = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"])
    t = rbd.rbd_terms(m, s["q"][1], s["v"][1])
    iu = np.tril_indices(m.nv)
    assert np.array_equal(rows["M"][1], t["M"][iu]) and np.array_equal(rows["h"][1], t["nle"])
    A = rows["A"][1].reshape(st.n_dense, m.nv)
    lh = m.frame("gripper_left_joint")
    assert np.array_equal(A[4:10], t["Jl"][lh])  # head 2 + head_pitch 1 + head_yaw 1 rows come first
    assert np.array_equal(A[0:2], t["Jl"][m.frame("head_1_joint")][0:2])
    assert np.array_equal(rows["Ac"][1].reshape(2, 6, m.nv)[0], t["Jl"][m.frame("leg_left_6_joint")])
    com_row = 2 + 1 + 1 + 6 + 6 + 2 + 6 + 6
    assert np.array_equal(A[com_row:com_row + 3], t["Jcom"])
    assert np.array_equal(A[com_row + 3:com_row + 5], t["Ag"][3:5])  # momentum mask 000110
    assert np.all(rows["blb"] <= rows["bub"])
    assert np.all(rows["b1"][:, st.n_dense + st.n_sel:] == 0.0)


def test_se3_and_com_laws_close_the_loop(rbd, oracle_mod):
    """Integrate dv = pinv(A) b for the hand task + CoM task only: each error must decay like the critically damped
    second-order system the gains describe (Kd = 2 sqrt(Kp)): e(t) = e0 (1 + sqrt(Kp) t) exp(-sqrt(Kp) t) from rest."""
    m = mdl.talos_like()
    frame = m.frame("gripper_left_joint")
    blocks = [mdl.TaskBlock("lh", mdl.T_SE3, frame, 63, 30.0, 2 * np.sqrt(30.0), 0), mdl.TaskBlock("com", mdl.T_COM, 0, 7, 30.0, 2 * np.sqrt(30.0), 24)]
    tm = mdl.TaskMap(blocks=blocks, sel_col=np.zeros(0, np.int32), posture_kp=0.0, posture_kd=0.0, posture_ref=33,
                     contact_frame=np.zeros(0, np.int32), contact_kp=np.zeros(0), contact_kd=np.zeros(0), contact_ref=np.zeros(0, np.int32),
                     n_bound=0, dt=1e-3, nref=33)

    class St:  # the few sizes task_rows needs
        n_dense = 9

        @staticmethod
        def field_lengths():
            return dict(M=m.nv * (m.nv + 1) // 2, h=m.nv, A=9 * m.nv, b1=9, Ac=0, bc=0, blb=0, bub=0)

    Rf0, pf0 = m.frame_placements(m.q0)
    com0 = m.com(m.q0)
    dpos = np.array([0.05, -0.03, 0.04])
    Rref = Rf0[frame] @ Rotation.from_rotvec([0.2, -0.1, 0.15]).as_matrix()
    ref = np.zeros((1, 33))
    ref[0, 0:12] = mdl.se3_ref(Rref, pf0[frame] + dpos)
    ref[0, 24:27] = com0 + np.array([0.01, 0.02, -0.03])
    q, v = m.q0.copy()[None], np.zeros((1, m.nv))
    dt, kp = 1e-3, 30.0
    e0p = e0c = None
    for k in range(601):
        rows = rbd.task_rows(m, tm, St, q, v, ref)
        A = rows["A"][0].reshape(9, m.nv)
        Rf, pf = m.frame_placements(q[0])
        ep = np.linalg.norm(pf[frame] - (pf0[frame] + dpos))
        er = np.linalg.norm(Rotation.from_matrix(Rf[frame].T @ Rref).as_rotvec())
        ec = np.linalg.norm(m.com(q[0]) - ref[0, 24:27])
        if k == 0:
            e0p, e0r, e0c = ep, er, ec
        if k in (100, 300, 600):
            tt = k * dt
            env = (1 + np.sqrt(kp) * tt) * np.exp(-np.sqrt(kp) * tt)
            assert abs(ep - e0p * env) < 0.03 * e0p, (k, ep, e0p * env)
            assert abs(er - e0r * env) < 0.03 * e0r, (k, er, e0r * env)
            assert abs(ec - e0c * env) < 0.03 * e0c, (k, ec, e0c * env)
        dv = np.linalg.lstsq(A, rows["b1"][0], rcond=None)[0]
        nxt = oracle_mod.integrate(True, dt, q, v, dv[None])
        q, v = nxt["q_next"], nxt["v_next"]


def test_bounds_keep_a_joint_inside_its_limits(rbd):
    """Push a joint towards its upper limit as hard as the bounds allow every tick (constant acceleration over the tick, as the bounds assume):
    it must never cross the limit, and must respect the velocity limit until it gets there (with the reference's
    ddq_max = dq_max / dt, tasks.cpp:288, the position bound then takes precedence over the velocity bound)."""
    m, st, tm = _talos_setup()
    j = 10
    q, v = m.q0.copy(), np.zeros(m.nv)
    s = mdl.sample_states(m, tm, 1, 5)
    q[7 + j] = m.q_ub[j] - 0.2
    v[6 + j] = 0.9 * m.dq_max[j]
    worst, hit = -1e9, False
    for k in range(400):
        rows = rbd.task_rows(m, tm, st, q[None], v[None], s["ref"])
        lb, ub = rows["blb"][0, j], rows["bub"][0, j]
        assert lb <= ub
        a = ub  # push as hard as allowed towards the upper limit
        q[7 + j] += tm.dt * v[6 + j] + 0.5 * tm.dt ** 2 * a
        v[6 + j] += tm.dt * a
        worst = max(worst, q[7 + j] - m.q_ub[j])
        if not hit:
            assert abs(v[6 + j]) <= m.dq_max[j] * (1 + 1e-9)
        hit = hit or (m.q_ub[j] - q[7 + j]) < 1e-3
    assert worst < 1e-6, worst


def _golden_case(tag):
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "before_path", "rows_%s.npz" % tag))
    if tag == "talos":
        m, st, stack = mdl.talos_like(), structure.talos_structure(), mdl.talos_stack()
    else:
        m, st, stack = mdl.franka_like(), structure.franka_structure(), mdl.franka_stack()
    return z, m, st, mdl.build_taskmap(m, st, stack)


@pytest.mark.parametrize("tag", ["talos", "franka"])
def test_rows_golden(rbd, tag):
    """The committed fixtures pin the oracle AND the seeded model builders (tests/golden/make_golden.py::main_before_path)."""
    z, m, st, tm = _golden_case(tag)
    rows = rbd.task_rows(m, tm, st, z["q"], z["v"], z["ref"])
    for k, a in rows.items():
        if a.size:
            assert np.abs(a - z[k]).max() <= 1e-12 * max(1.0, np.abs(z[k]).max()), k


def test_sample_states_do_not_depend_on_the_sharding():
    """Instance i of a stream depends on seed + i only: a rank that owns [lo, hi) draws exactly its slice."""
    m, st, tm = _talos_setup()
    full = mdl.sample_states(m, tm, 12, 42_000)
    part = mdl.sample_states(m, tm, 5, 42_000 + 4)
    for k in ("q", "v", "ref"):
        assert np.array_equal(full[k][4:9], part[k])


def test_cop_rows_of_a_yawed_robot_are_the_physics_not_the_formula(oracle_mod):
    """ADVICE (round 4): the device code, this oracle and structure.cop_rows all carry the same recalled expression (d n' - (n . d) I) R for tsid's
    TaskCopEquality -- a shared misreading would pass every parity test.  This check shares nothing with them: the contact frames' placements
    come from the numpy forward kinematics (inria_wbc_amd/model.py), the robot is turned by 40 degrees about the vertical (plus the usual state
    noise), and the three rows applied to random contact-FRAME forces must be n x sum_i (p_i - c) x (R f_i), the tangential moment of the world
    forces about the reference point, computed with np.cross.  (Without the factor R -- the other reading of upstream -- the identity fails as
    soon as a foot is not world-aligned: that variant is evaluated below and must NOT pass.)"""
    from inria_wbc_amd import model as mdl, structure
    from oracle import rbd
    m = mdl.talos_like()
    st = structure.STRUCTURES["talos_torque_cop"]()
    stack = mdl.talos_stack() + [dict(name="torque", type="torque", weight=1e-2), dict(name="cop", type="cop", weight=10.0)]
    tm = mdl.build_taskmap(m, st, stack)
    s = mdl.sample_states(m, tm, 4, 77, q_noise=0.05)
    yaw = np.deg2rad(40.0)
    for i in range(4):  # turn the floating base about the world's z axis: q[3:7] = (x, y, z, w)
        x, y, z, w = s["q"][i, 3:7]
        c, sn = np.cos(yaw / 2), np.sin(yaw / 2)
        s["q"][i, 3:7] = [c * x - sn * y, c * y + sn * x, c * z + sn * w, c * w - sn * z]  # (0, 0, sn, c) * q
    rows = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"])
    n = np.array([0.0, 0.0, 1.0])
    rng = np.random.default_rng(3)
    for i in range(4):
        A = rows["Acop"][i].reshape(3, st.k)
        Rf, pf = m.frame_placements(s["q"][i])
        f = rng.standard_normal(st.k)
        mom, no_R = np.zeros(3), np.zeros(3)
        for c, contact in enumerate(st.contacts):
            R, p = Rf[tm.contact_frame[c]], pf[tm.contact_frame[c]]
            assert abs(np.arctan2(R[1, 0], R[0, 0])) > np.deg2rad(25.0)  # the foot really is yawed
            for k in range(4):
                d = R @ contact.points[:, k] + p  # cop_ref = 0
                fk = f[12 * c + 3 * k:12 * c + 3 * k + 3]
                mom += np.cross(d, R @ fk)
                no_R += np.cross(d, fk)
        assert np.abs(A @ f - np.cross(n, mom)).max() <= 1e-10 * max(1.0, np.abs(mom).max())
        assert np.abs(A @ f - np.cross(n, no_R)).max() > 1e-3  # the reading without R is a different constraint on a yawed foot
