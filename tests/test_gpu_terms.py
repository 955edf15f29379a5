"""GPU parity of the step before the path (wbcqp_problem_data: rigid-body terms + task laws)
against oracle/rbd_oracle.c, through the C ABI.

Tolerance: floating point, two different formulations (world-frame prefix sums on the device, pinocchio-style local
recursions in the oracle): every array must agree to TOL_ROWS relative to its own largest entry (at least 1)."""
import numpy as np
import pytest

from inria_wbc_amd import capi, structure
from inria_wbc_amd import model as mdl
from tests.util import assert_parity, device_outputs, host_outputs

pytestmark = pytest.mark.gpu

TOL_ROWS = 1e-10


def _cases():
    def talos():
        m = mdl.talos_like()
        st = structure.talos_structure()
        return m, st, mdl.build_taskmap(m, st, mdl.talos_stack())

    def franka():
        m = mdl.franka_like()
        st = structure.franka_structure()
        return m, st, mdl.build_taskmap(m, st, mdl.franka_stack())

    def talos_ss():  # single support: contact_lfoot removed as WalkOnSpot does (walk_on_spot.cpp:165-184)
        m = mdl.talos_like()
        st = structure.talos_structure(single_support=True)
        return m, st, mdl.build_taskmap(m, st, [n for n in mdl.talos_stack() if n["name"] != "contact_lfoot"])

    def icub():
        m = mdl.icub_like()
        st = structure.icub_structure()
        return m, st, mdl.build_taskmap(m, st, mdl.icub_stack())

    def talos_torque_cop():  # the two task types no shipped stack uses (tasks.cpp:227-271, :156-178) at the end of Talos' stack
        m = mdl.talos_like()
        st = structure.STRUCTURES["talos_torque_cop"]()
        stack = mdl.talos_stack() + [dict(name="torque", type="torque", weight=1e-2), dict(name="cop", type="cop", weight=10.0)]
        return m, st, mdl.build_taskmap(m, st, stack)

    def talos_posture_mask():  # `mask:` on the posture task (tasks.cpp:205-214): the grippers' joints left out
        m = mdl.talos_like()
        mask = np.ones(44, int)
        mask[[21, 22, 23, 24, 25, 26, 27, 35, 36, 37, 38, 39, 40, 41]] = 0
        st = structure.with_posture_mask(structure.talos_structure(), mask)
        stack = [dict(n, mask="".join(str(int(b)) for b in mask)) if n["type"] == "posture" else n for n in mdl.talos_stack()]
        return m, st, mdl.build_taskmap(m, st, stack)

    def tree(seed, nb, fb, n_contacts=2):
        def f():
            m = mdl.random_tree(seed, nb, fb, nframe=12)
            st, stack = mdl.random_stack(m, seed + 100, n_contacts)
            return m, st, mdl.build_taskmap(m, st, stack, dt=2e-3)
        return f

    def three_limbs():
        """A floating base with three chains of seven joints and a contact at the end of each: the contact Jacobians are
        independent (six joints and more between any two contact frames), so the 24 equalities have full rank."""
        m = mdl.random_tree(24, 22, True, nframe=12)
        m.parent = np.array([-1] + [0 if k % 7 == 0 else 1 + k - 1 for k in range(21)], dtype=np.int32)
        m.frame_body[0:3] = [7, 14, 21]
        m.validate()
        st, stack = mdl.random_stack(m, 124, 3)
        for node in stack:
            if node["type"] == "contact":
                node["joint"] = "f%d" % int(node["name"][-1])
        return m, st, mdl.build_taskmap(m, st, stack, dt=2e-3)

    return {"talos": talos, "talos_torque_cop": talos_torque_cop, "talos_posture_mask": talos_posture_mask, "talos_single_support": talos_ss, "icub": icub, "franka": franka, "tree_fb": tree(21, 30, True), "tree_fixed": tree(22, 19, False), "tree_big": tree(23, 62, False), "tree_three_contacts": three_limbs}


CASES = _cases()


@pytest.fixture(scope="module")
def rbd():
    from oracle import rbd as r
    return r


@pytest.fixture(scope="module")
def handle():
    h = capi.Handle(0, capi.F64)
    yield h
    h.close()


def _compare(dev, ora, tol=TOL_ROWS):
    worst = {}
    for k in capi.ROW_FIELDS:
        if ora[k].size == 0:
            continue
        scale = max(1.0, np.abs(ora[k]).max())
        worst[k] = np.abs(dev[k] - ora[k]).max() / scale
        assert worst[k] <= tol, (k, worst)
    return worst


@pytest.mark.parametrize("name", list(CASES))
def test_problem_data_parity(handle, rbd, name):
    m, st, tm = CASES[name]()
    handle.set_structure(3, st)
    handle.set_model(3, m, tm)
    big = dict(q_noise=0.3, v_noise=0.5, ref_noise=0.2) if name.startswith("tree") else {}
    s = mdl.sample_states(m, tm, 48, 31_000, **big)
    dev = handle.problem_data_host(3, s["q"], s["v"], s["ref"])
    ora = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"], n_threads=8)
    _compare(dev, ora)


@pytest.mark.parametrize("name", ["talos", "icub"])
def test_momentum_output_is_Ag_v(handle, rbd, name):
    """wbcqp_state.momentum = the centroidal momentum Ag(q) v of the tick's state (linear, then angular about the CoM): its last
    three entries are what the reference keeps as momentum_ (controller.cpp:245, momentumJacobian(data).bottomRows(3) * dq)."""
    m, st, tm = CASES[name]()
    handle.set_structure(3, st)
    handle.set_model(3, m, tm)
    s = mdl.sample_states(m, tm, 24, 35_000, q_noise=0.2, v_noise=0.5)
    dev = handle.problem_data_host(3, s["q"], s["v"], s["ref"])
    for i in range(24):
        t = rbd.rbd_terms(m, s["q"][i], s["v"][i])
        want = t["Ag"] @ s["v"][i]
        assert np.abs(dev["momentum"][i] - want).max() <= 1e-10 * max(1.0, np.abs(want).max()), (i, dev["momentum"][i], want)


def test_problem_data_far_from_the_origin(handle, rbd):
    """The device works about the floating base: a robot 1 km away must give the same rows as the oracle's absolute arithmetic
    (up to what the oracle itself loses there)."""
    m, st, tm = CASES["talos"]()
    handle.set_structure(3, st)
    handle.set_model(3, m, tm)
    s = mdl.sample_states(m, tm, 8, 32_000)
    shift = np.array([1000.0, -500.0, 20.0])
    s["q"][:, 0:3] += shift
    for b in tm.blocks:
        if b.kind in (mdl.T_SE3, mdl.T_COM):
            s["ref"][:, b.ref:b.ref + 3] += shift
    for c in range(tm.ncontact):
        s["ref"][:, tm.contact_ref[c]:tm.contact_ref[c] + 3] += shift
    dev = handle.problem_data_host(3, s["q"], s["v"], s["ref"])
    ora = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"])
    _compare(dev, ora, tol=1e-7)  # the oracle's own cancellation at |p| = 1e3 (M, h through 1e6-sized intermediate moments)


def test_problem_data_wound_up_joint_angles(handle, rbd):
    """The kernel's own sin / cos of a joint angle (Cody-Waite reduction by pi/2 + the fdlibm kernels, wbcqp_terms.hpp sincos_joint)
    against the oracle's libm: angles many turns from zero, exact multiples of pi/2, and beyond 1e5 where the kernel hands the
    argument to the library's sincos().  Only the kinematics matter here: the joint-limit rows see the same q on both sides."""
    m, st, tm = CASES["franka"]()
    handle.set_structure(3, st)
    handle.set_model(3, m, tm)
    s = mdl.sample_states(m, tm, 32, 36_000)
    rng = np.random.default_rng(7)
    rev = np.where(np.asarray(m.jtype) <= 3)[0]  # the revolute joints (J_RX .. J_RZ); fixed base: idx_q = joint index
    nj = rev.size
    s["q"][0:8, rev] += rng.uniform(-60.0, 60.0, (8, nj))                        # tens of turns
    s["q"][8:16, rev] = np.round(s["q"][8:16, rev] / (np.pi / 2)) * (np.pi / 2)  # on the quadrant boundaries
    s["q"][16:24, rev] += rng.uniform(-9.0e4, 9.0e4, (8, nj))                    # the top of the polynomial path
    s["q"][24:32, rev] += rng.choice([-1.0, 1.0], (8, nj)) * rng.uniform(1.0e5, 1.0e7, (8, nj))  # the library path
    dev = handle.problem_data_host(3, s["q"], s["v"], s["ref"])
    ora = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"])
    for k in ("M", "h", "A", "b1"):
        scale = max(1.0, np.abs(ora[k]).max())
        # an angle of 1e7 carries an absolute error of 1e-9 in double: the two sides see the same q, so only the reduction differs
        assert np.abs(dev[k] - ora[k]).max() <= 1e-9 * scale, k


def test_problem_data_then_solve_matches_oracle_pipeline(handle, rbd):
    """State -> rows -> QP -> torques, all on the device, against oracle rows -> oracle tick."""
    import torch
    from oracle import oracle as orc
    m, st, tm = CASES["talos"]()
    handle.set_structure(2, st)
    handle.set_model(2, m, tm)
    B = 96
    s = mdl.sample_states(m, tm, B, 33_000, q_noise=0.01, v_noise=0.05, ref_noise=0.01)
    dev = torch.device("cuda", 0)
    L = st.field_lengths()
    state = {k: torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref")}
    rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
    rows["tlb"] = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
    rows["tub"] = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
    rows["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
    out = device_outputs(B, st, dev)
    stream = torch.cuda.current_stream().cuda_stream
    handle.problem_data(2, B, state, rows, stream=stream)
    handle.solve_batch(2, B, rows, out, stream=stream)
    torch.cuda.synchronize()
    ora_rows = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"], n_threads=8)
    ora_in = dict(ora_rows, tlb=np.tile(-m.tau_max, (B, 1)), tub=np.tile(m.tau_max, (B, 1)), w=np.tile(st.default_weights, (B, 1)))
    ref = orc.tick_batch(st, ora_in, nthreads=8)
    got = host_outputs(out, st)
    assert (ref["status"] == 0).all()
    assert_parity(st, got, ref, what="rows on the device -> QP (talos-like)")


@pytest.mark.parametrize("name", ["talos_torque_cop", "talos_posture_mask"])
def test_whole_tick_with_the_unshipped_task_types(rbd, name):
    """State -> rows (cop rows from the contact frames, zero right-hand sides for the torque rows) -> QP with H as one matrix ->
    torques -> integrated state through wbcqp_tick_host, against the three oracles chained the same way."""
    from oracle import oracle as orc
    m, st, tm = CASES[name]()
    h = capi.Handle(0, capi.F64)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    B = 40
    s = mdl.sample_states(m, tm, B, 36_000, q_noise=0.01, v_noise=0.05, ref_noise=0.01)
    tlb, tub, w = np.tile(-m.tau_max, (B, 1)), np.tile(m.tau_max, (B, 1)), np.tile(st.default_weights, (B, 1))
    got = h.tick_host(0, s["q"], s["v"], s["ref"], tlb, tub, w, tm.dt, want_rows=True)
    ora_rows = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"], n_threads=8)
    _compare(got["rows"], ora_rows)
    if st.cop_task >= 0:
        assert np.abs(ora_rows["Acop"]).max() > 0.01
    ref = orc.tick_batch(st, dict(ora_rows, tlb=tlb, tub=tub, w=w), nthreads=8)
    assert (ref["status"] == 0).all()
    assert_parity(st, got, ref, what=name)
    nxt = orc.integrate(True, tm.dt, s["q"], s["v"], ref["x"][:, :st.nv])
    assert np.abs(got["q_next"] - nxt["q_next"]).max() <= 1e-9 and np.abs(got["v_next"] - nxt["v_next"]).max() <= 1e-8
    h.close()


def test_set_model_rejects_mismatches(handle):
    m, st, tm = CASES["talos"]()
    handle.set_structure(4, structure.icub_structure())
    with pytest.raises(capi.WbcqpError):
        handle.set_model(4, m, tm)  # nv / na of another robot
    handle.set_structure(4, st)
    bad = mdl.talos_like()
    bad.parent = bad.parent.copy()
    bad.parent[20] = 3  # breaks the depth-first numbering
    with pytest.raises(capi.WbcqpError):
        handle.set_model(4, bad, tm)
    tm2 = mdl.build_taskmap(m, st, mdl.talos_stack())
    tm2.blocks[0].ref = tm2.nref  # reference outside the vector
    with pytest.raises(capi.WbcqpError):
        handle.set_model(4, m, tm2)
    h2 = capi.Handle(0, capi.F64)
    h2.set_structure(0, st)
    with pytest.raises(capi.WbcqpError):
        h2.problem_data_host(0, np.zeros((1, m.nq)), np.zeros((1, m.nv)), np.zeros((1, tm.nref)))  # no model bound
    h2.close()


def test_problem_data_f32_boundary(rbd):
    m, st, tm = CASES["talos"]()
    h = capi.Handle(0, capi.F32)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    s = mdl.sample_states(m, tm, 16, 34_000)
    s32 = {k: v.astype(np.float32) for k, v in s.items()}
    dev = h.problem_data_host(0, s32["q"], s32["v"], s32["ref"])
    ora = rbd.task_rows(m, tm, st, s32["q"].astype(np.float64), s32["v"].astype(np.float64), s32["ref"].astype(np.float64))
    for k in capi.ROW_FIELDS:
        if ora[k].size == 0:
            continue
        scale = max(1.0, np.abs(ora[k]).max())
        assert np.abs(dev[k].astype(np.float64) - ora[k]).max() / scale < 1e-6, k
    h.close()


@pytest.mark.parametrize("tag", ["talos", "franka"])
def test_problem_data_golden_gpu(handle, tag):
    """The committed fixtures (tests/golden/before_path) through the C ABI: the oracle does not run here."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "before_path", "rows_%s.npz" % tag))
    m, st, tm = CASES[tag]()
    handle.set_structure(5, st)
    handle.set_model(5, m, tm)
    dev = handle.problem_data_host(5, z["q"], z["v"], z["ref"])
    _compare(dev, {k: (z[k] if k in z.files else np.zeros((z["q"].shape[0], 0))) for k in capi.ROW_FIELDS})  # (Acop: no cop task in the fixtures)


def test_mixed_robots_rows_then_one_ragged_solve(rbd):
    """BASELINE config 5 with rows from the models: Talos-like, iCub-like and Franka-like instances, one wbcqp_problem_data
    launch per robot type, then ONE wbcqp_solve_ragged launch over the three groups, against the oracle pipeline."""
    import torch
    from oracle import oracle as orc
    dev = torch.device("cuda", 0)
    h = capi.Handle(0, capi.F64)
    stream = torch.cuda.current_stream().cuda_stream
    groups, checks = [], []
    for slot, (name, B) in enumerate((("talos", 20), ("icub", 33), ("franka", 50))):
        m, st, tm = CASES[name]()
        h.set_structure(slot, st)
        h.set_model(slot, m, tm)
        s = mdl.sample_states(m, tm, B, 61_000 + slot)
        L = st.field_lengths()
        rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
        tmax = m.tau_max if st.act_bounds else np.zeros(0)
        rows["tlb"] = torch.from_numpy(np.tile(-tmax, (B, 1))).to(dev)
        rows["tub"] = torch.from_numpy(np.tile(tmax, (B, 1))).to(dev)
        rows["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
        h.problem_data(slot, B, {k: torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref")}, rows, stream=stream)
        out = device_outputs(B, st, dev)
        groups.append((slot, B, rows, out))
        ora = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"], n_threads=4)
        checks.append((st, out, dict(ora, tlb=np.tile(-tmax, (B, 1)), tub=np.tile(tmax, (B, 1)), w=np.tile(st.default_weights, (B, 1)))))
    h.solve_ragged(groups, stream=stream)
    torch.cuda.synchronize()
    for st, out, ora_in in checks:
        ref = orc.tick_batch(st, ora_in, nthreads=4)
        got = host_outputs(out, st)
        assert (ref["status"] == 0).all()
        assert_parity(st, got, ref, what="rows on the device -> ragged QP launch: " + st.name)
    h.close()


def test_problem_data_random_trees_and_stacks():
    """A short run of tests/stress/stress_rows.py: random branching trees and pure chains up to 61 bodies, every joint type, random
    masks, contacts, self-collision tasks, large states."""
    from tests.stress import stress_rows
    worst = stress_rows.run(24, tol=TOL_ROWS)
    assert worst and max(worst.values()) < TOL_ROWS


def test_problem_data_minimal_stack(handle, rbd):
    """A stack with nothing optional: two SE(3) tasks on a fixed-base tree -- no posture, no CoM, no contacts, no bounds, no
    self-collision (every optional table of the kernel is empty)."""
    from inria_wbc_amd import structure as S
    m = mdl.random_tree(77, 11, False, nframe=9)
    stack = [dict(name="a", type="se3", tracked="f2", kp=20.0, mask="111000"), dict(name="b", type="se3", tracked="f5", kp=5.0, mask="000111")]
    st = S._mk("minimal", m.nv, m.na, [], [("a", 3, 1.0), ("b", 3, 2.0)], None, [], False, False, [])
    tm = mdl.build_taskmap(m, st, stack)
    assert tm.n_bound == 0 and tm.ncontact == 0 and tm.sel_col.size == 0 and tm.nref == 48
    handle.set_structure(6, st)
    handle.set_model(6, m, tm)
    s = mdl.sample_states(m, tm, 5, 35_000, q_noise=0.4, v_noise=0.8, ref_noise=0.2)
    dev = handle.problem_data_host(6, s["q"], s["v"], s["ref"])
    ora = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"])
    _compare(dev, ora)
    # and the solve on those rows: an unconstrained QP, x = -H^-1 g
    from oracle import oracle as orc
    B = 5
    got = handle.solve_batch_host(6, dict(dev, tlb=np.zeros((B, 0)), tub=np.zeros((B, 0)), w=np.tile(st.default_weights, (B, 1))))
    ref = orc.tick_batch(st, dict(ora, tlb=np.zeros((B, 0)), tub=np.zeros((B, 0)), w=np.tile(st.default_weights, (B, 1))))
    assert (got["status"] == 0).all() and np.abs(got["x"] - ref["x"]).max() <= 1e-8 * max(1.0, np.abs(ref["x"]).max())


def test_set_model_rejects_bad_tables(handle):
    """Every table that indexes another one is checked before anything goes to the device: an error code, not a fault."""
    import copy
    m, st, tm = CASES["talos"]()
    handle.set_structure(7, st)

    def expect_refusal(mutate_model=None, mutate_map=None):
        mm, tt = copy.deepcopy(m), copy.deepcopy(tm)
        if mutate_model:
            mutate_model(mm)
        if mutate_map:
            mutate_map(tt)
        with pytest.raises(capi.WbcqpError):
            handle.set_model(7, mm, tt)

    expect_refusal(mutate_map=lambda t: setattr(t.blocks[3], "frame", m.nframe + 5))          # tracked frame out of range
    expect_refusal(mutate_map=lambda t: t.blocks[-1].avoided.__setitem__(0, (-1, 0.1)))       # avoided frame out of range
    expect_refusal(mutate_map=lambda t: setattr(t, "dt", 0.0))                                # no time step
    expect_refusal(mutate_map=lambda t: t.contact_ref.__setitem__(1, t.nref - 3))             # contact sample past the end
    expect_refusal(mutate_map=lambda t: setattr(t, "posture_ref", t.nref))                    # posture reference past the end
    expect_refusal(mutate_map=lambda t: setattr(t.blocks[-1], "m", 0.0))                      # 5PL exponent must be positive
    expect_refusal(mutate_map=lambda t: t.blocks.pop(0))                                      # rows no longer add up to n_dense
    expect_refusal(mutate_model=lambda x: x.jtype.__setitem__(5, 9))                          # unknown joint type
    expect_refusal(mutate_model=lambda x: x.frame_body.__setitem__(2, 99))                    # frame on a body that does not exist
    handle.set_model(7, m, tm)  # and the untouched pair is accepted


def test_three_contacts_rows_then_solve(handle, rbd):
    """An odd number of contacts (24 equalities: the solver's sequential equality path) with rows from a random tree: rows kernel,
    then the QP, against the oracle pipeline."""
    from oracle import oracle as orc
    m, st, tm = CASES["tree_three_contacts"]()
    assert st.nc == 3 and st.neq == 24
    handle.set_structure(8, st)
    handle.set_model(8, m, tm)
    B = 24
    s = mdl.sample_states(m, tm, B, 36_000, q_noise=0.02, v_noise=0.05, ref_noise=0.01)
    dev = handle.problem_data_host(8, s["q"], s["v"], s["ref"])
    ora = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"], n_threads=4)
    _compare(dev, ora)
    extra = dict(tlb=np.zeros((B, 0)), tub=np.zeros((B, 0)), w=np.tile(st.default_weights, (B, 1)))
    got = handle.solve_batch_host(8, dict(dev, **extra))
    ref = orc.tick_batch(st, dict(ora, **extra), nthreads=4)
    assert np.array_equal(got["status"], ref["status"])
    ok = ref["status"] == 0
    assert ok.sum() >= B // 2  # random contact frames on a random tree: some instances may be infeasible, the statuses must agree
    assert np.abs(got["x"][ok][:, :m.nv] - ref["x"][ok][:, :m.nv]).max() <= 1e-7 * max(1.0, np.abs(ref["x"][ok]).max())
