"""The two task types of the reference's factory that no shipped stack uses -- `torque` (src/controllers/tasks.cpp:227-271, tsid
TaskActuationEquality) and `cop` (tasks.cpp:156-178, tsid TaskCopEquality) -- and the posture task's `mask:` (tasks.cpp:205-214):
CPU side.  The oracle's assembly of the new level-1 rows against an independent numpy statement of the same tsid rules (SURVEY A.1
steps 5-6), KKT conditions of its solutions, the sizes (Appendix-B style) and the C ABI's validation of the new structure fields."""
import numpy as np
import pytest

from inria_wbc_amd import structure, synth


def _dense_level1(st, inp, i):
    """(rows [r1, n], weights [r1], rhs [r1]) of every level-1 constraint of QP i, the way computeProblemData fills them"""
    nv, na, nu, nc, n, k = st.nv, st.na, st.nu, st.nc, st.n, st.k
    M = np.zeros((nv, nv)); M[np.tril_indices(nv)] = inp["M"][i]; M = M + M.T - np.diag(np.diag(M))
    h = inp["h"][i]; A = inp["A"][i].reshape(st.n_dense, nv); b1 = inp["b1"][i]; w = inp["w"][i]
    Ac = inp["Ac"][i].reshape(nc, 6, nv); T = st.force_gen(); F = st.forcereg_mat()
    Jc = np.concatenate([T[c].T @ Ac[c] for c in range(nc)], 0) if nc else np.zeros((0, nv))
    rows, wts, rhs = [], [], []
    for r in range(st.n_dense):
        rows.append(np.concatenate([A[r], np.zeros(k)])); wts.append(w[st.dense_row_task[r]]); rhs.append(b1[r])
    for s_ in range(st.n_sel):
        e = np.zeros(n); e[st.sel_col[s_]] = 1.0
        rows.append(e); wts.append(w[st.sel_task[s_]]); rhs.append(b1[st.n_dense + s_])
    for c in range(nc):
        for q in range(6):
            e = np.zeros(n); e[nv + 12 * c:nv + 12 * c + 12] = F[c][q]
            rows.append(e); wts.append(w[st.forcereg_task[c]]); rhs.append(b1[st.n_dense + st.n_sel + 6 * c + q])
    o = st.n_dense + st.n_sel + 6 * nc
    # actuation task: constraint S tau = S tau_ref, S(j, joint_j) = scale_j; tau = M_a dv + h_a - J_a' f
    S = np.zeros((st.n_acteq, na))
    for j in range(st.n_acteq):
        S[j, st.acteq_joint[j]] = st.acteq_scale[j]
    At = S @ np.hstack([M[nu:], -Jc[:, nu:].T])
    bt = b1[o:o + st.n_acteq] - S @ h[nu:]
    for j in range(st.n_acteq):
        rows.append(At[j]); wts.append(w[st.acteq_task]); rhs.append(bt[j])
    if st.cop_task >= 0:
        Acop = inp["Acop"][i].reshape(3, k)
        for r in range(3):
            rows.append(np.concatenate([np.zeros(nv), Acop[r]])); wts.append(w[st.cop_task]); rhs.append(b1[o + st.n_acteq + r])
    return np.array(rows), np.array(wts), np.array(rhs)


@pytest.mark.parametrize("name", ["talos_torque", "talos_cop", "talos_torque_cop", "icub_torque"])
def test_oracle_assembly_of_torque_and_cop_rows(oracle_mod, name):
    st = structure.STRUCTURES[name]()
    inp = synth.generate(st, 3, synth.SEED_BASE[name] + 5, torque_ref_noise=2.0)
    plain = structure.STRUCTURES[name.split("_")[0]]()
    for i in range(3):
        H, g, CE, ce0, CI, ci0 = oracle_mod.assemble(st, inp, i)
        Aall, W, ball = _dense_level1(st, inp, i)
        assert Aall.shape[0] == st.r1
        H_np = Aall.T @ (W[:, None] * Aall) + st.hessian_reg * np.eye(st.n)
        g_np = -Aall.T @ (W * ball)
        sc = np.abs(H_np).max()
        assert np.abs(H - H_np).max() <= 1e-12 * sc and np.abs(g - g_np).max() <= 1e-12 * max(1.0, np.abs(g_np).max())
        # level-1 tasks leave CE and CI alone: the plain stack's, on the same record
        pin = {k: v for k, v in inp.items() if k != "Acop"}
        pin["b1"] = inp["b1"][:, :plain.r1]
        pin["w"] = inp["w"][:, :plain.n_tasks]
        Hp, gp, CEp, ce0p, CIp, ci0p = oracle_mod.assemble(plain, pin, i)
        assert np.array_equal(CE, CEp) and np.array_equal(CI, CIp) and np.array_equal(ce0, ce0p) and np.array_equal(ci0, ci0p)
        # and they do couple what the other tasks keep apart
        nv = st.nv
        if st.n_acteq:
            assert np.abs(H[:nv, nv:]).max() > 0 and np.abs(Hp[:nv, nv:]).max() == 0
        if st.cop_task >= 0:  # (a leg joint moves one foot only: the torque rows alone leave the two feet's force blocks apart)
            assert np.abs(H[nv:nv + 12, nv + 12:]).max() > 0 and np.abs(Hp[nv:nv + 12, nv + 12:]).max() == 0


@pytest.mark.parametrize("name,noise", [("talos_torque", 0.5), ("talos_cop", 2.0), ("talos_torque_cop", 2.0), ("icub_torque", 5.0)])
def test_kkt_conditions_with_the_new_tasks(oracle_mod, name, noise):
    st = structure.STRUCTURES[name]()
    inp = synth.generate(st, 4, synth.SEED_BASE[name] + 31, task_noise=noise)
    out = oracle_mod.tick_batch(st, inp)
    assert (out["status"] == 0).all()
    for i in range(4):
        H, g, CE, ce0, CI, ci0 = oracle_mod.assemble(st, inp, i)
        o = oracle_mod.tick_single(st, inp, i)
        k = oracle_mod.kkt_residuals(H, g, CE, ce0, CI, ci0, o["x"], o["active"], o["lam"])
        assert k["stationarity"] < 1e-10 and k["eq"] < 1e-6 and k["min_mu"] > -1e-9, k


def test_the_torque_task_does_what_it_says(oracle_mod):
    """A heavy torque task pulls |tau| down (its reference is zero, tasks.cpp:263-265); masked joints are left alone."""
    base = structure.talos_structure()
    mask = np.zeros(base.na, int); mask[:12] = 1  # the legs
    heavy = structure.with_torque_task(base, 1.0, mask=mask)
    inp = synth.generate(heavy, 6, synth.SEED_BASE["talos"] + 3, p_act=0.0)
    lite = {k: v.copy() for k, v in inp.items()}
    lite["w"][:, heavy.acteq_task] = 1e-9
    a, b = oracle_mod.tick_batch(heavy, inp), oracle_mod.tick_batch(heavy, lite)
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    assert (np.abs(a["tau"][:, :12]).sum(axis=1) < np.abs(b["tau"][:, :12]).sum(axis=1)).all()


def test_cop_rows_are_the_tangential_moment_about_the_reference_point():
    """[UPSTREAM-RECALL] of TaskCopEquality::compute: A f = n x sum_i (p_i - c) x f_i for forces given in the contact frames."""
    st = structure.STRUCTURES["talos_cop"]()
    rng = np.random.default_rng(5)
    placements = []
    for c in range(2):
        Rq, _ = np.linalg.qr(np.eye(3) + 0.3 * rng.standard_normal((3, 3)))
        Rq *= np.sign(np.linalg.det(Rq))
        placements.append((Rq, rng.standard_normal(3)))
    cref = np.array([0.02, -0.01, 0.0])
    A = structure.cop_rows(st, placements, cop_ref=cref)
    f = rng.standard_normal(st.k)
    mom = np.zeros(3)
    for c, contact in enumerate(st.contacts):
        R, p = placements[c]
        for i in range(4):
            mom += np.cross(R @ contact.points[:, i] + p - cref, R @ f[12 * c + 3 * i:12 * c + 3 * i + 3])
    n = np.array([0.0, 0.0, 1.0])
    assert np.allclose(A @ f, np.cross(n, mom)) and abs(n @ (A @ f)) < 1e-12  # rank 2: nothing along the normal


@pytest.mark.parametrize("name,n,neq,nin,r1", [("talos_torque", 74, 18, 122, 141), ("talos_cop", 74, 18, 122, 100),
                                                ("talos_torque_cop", 74, 18, 122, 144), ("icub_torque", 62, 18, 66, 115)])
def test_sizes_and_layout(built_lib, name, n, neq, nin, r1):
    """tsid's nVar / nEq / nIn do not move (level-1 tasks); the level-1 rows grow by the mask's ones (+ 3); the stack takes the full layout."""
    from inria_wbc_amd import capi
    st = structure.STRUCTURES[name]()
    assert (st.n, st.neq, st.nin, st.r1) == (n, neq, nin, r1)
    L = capi.layout_of(st)
    assert (L["n"], L["neq"], L["nin"], L["r1"]) == (n, neq, nin, r1)
    assert L["dense_h"] == 1 and L["waves_per_cu"] == 1 and L["wave_per_qp"] == 0
    assert L["len_Acop"] == (3 * st.k if st.cop_task >= 0 else 0) and L["len_b1"] == r1
    plain = capi.layout_of(structure.STRUCTURES[name.split("_")[0]]())
    assert plain["dense_h"] == 0 and plain["waves_per_cu"] == (3 if name.startswith("icub") else 2) and plain["len_Acop"] == 0  # (the compact layout)


def test_posture_mask_keeps_the_ones(built_lib, oracle_mod):
    """tasks.cpp:205-214: `mask:` on the posture task -- one character per actuated joint, the selection rows are the ones."""
    from inria_wbc_amd import capi
    base = structure.talos_structure()
    mask = np.ones(base.na, int); mask[[0, 5, 20, 43]] = 0
    st = structure.with_posture_mask(base, mask)
    assert st.n_sel == base.na - 4 and st.r1 == base.r1 - 4 and capi.layout_of(st)["r1"] == st.r1
    assert not set(st.sel_col.tolist()) & {base.nu + j for j in (0, 5, 20, 43)}
    inp = synth.generate(st, 3, synth.SEED_BASE["talos"] + 17)
    H, g, *_ = oracle_mod.assemble(st, inp, 0)
    full = synth.generate(base, 3, synth.SEED_BASE["talos"] + 17)
    Hf, gf, *_ = oracle_mod.assemble(base, full, 0)
    d = np.diag(Hf - H)
    w = base.default_weights[base.task_names.index("posture")]
    assert np.allclose(d[[base.nu + j for j in (0, 5, 20, 43)]], w) and np.count_nonzero(np.abs(d) > 1e-12) == 4
    with pytest.raises(ValueError, match="wrong size in posture mask"):
        structure.with_posture_mask(base, [1, 0, 1])


def test_invalid_torque_and_cop_structures_are_rejected(built_lib):
    from inria_wbc_amd import capi
    import dataclasses
    base = structure.STRUCTURES["talos_torque"]()
    bad = dataclasses.replace(base, acteq_joint=base.acteq_joint[::-1].copy())
    with pytest.raises(capi.WbcqpError, match="ascending") as e:
        capi.layout_of(bad)
    assert e.value.code == 1
    bad = dataclasses.replace(base, acteq_task=99)
    with pytest.raises(capi.WbcqpError, match="acteq_task"):
        capi.layout_of(bad)
    with pytest.raises(ValueError, match="needs a contact"):
        structure.with_cop_task(structure.franka_structure())
    # three contacts: n = 86 > 80 -- H as one matrix does not fit the register tile; the dense seam is the way (wbcqp_solve_dense)
    with pytest.raises(capi.WbcqpError, match="n <= 80") as e:
        capi.layout_of(structure.with_torque_task(structure.three_contact_structure(nv=50, na=44), 1e-2))
    assert e.value.code == 3
    with pytest.raises(ValueError, match="wrong size in torque mask"):
        structure.with_torque_task(structure.talos_structure(), 1.0, mask=[1, 1])


def test_a_zero_initialised_cop_task_is_refused(built_lib):
    """ADVICE (round 4): wbcqp_structure.cop_task uses -1 for "none", so the usual C idiom -- memset the structure, fill what you know -- declares a
    cop task on task 0.  Task 0 carries other rows in every stack (Talos: the head task's), and a cop task is a task of its own: the library says so
    instead of sizing b1 / Acop for a task nobody asked for."""
    from inria_wbc_amd import capi
    import dataclasses
    plain = structure.talos_structure()
    assert plain.cop_task == -1 and capi.layout_of(plain)["len_Acop"] == 0
    with pytest.raises(capi.WbcqpError, match="cop_task = -1") as e:
        capi.layout_of(dataclasses.replace(plain, cop_task=0))
    assert e.value.code == 1
    good = structure.STRUCTURES["talos_cop"]()  # its cop task is the last one, with a weight of its own
    assert capi.layout_of(good)["len_Acop"] == 3 * good.k
