"""The C++ host facade (inria_wbc_amd/csrc/host): the reference's controller / behavior plugin surface over the C ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_build(built_lib):
    from inria_wbc_amd import build
    return build.build_host()


def test_facade_cpu_checks(host_build, tmp_path):
    """factories, YAML subset, task stacks (SURVEY App. B sizes), solver-switch errors, min-jerk, trajectory files -- no GPU
    needed."""
    r = subprocess.run([host_build["test_facade"], os.path.join(ROOT, "configs"), str(tmp_path)], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK (0 failures)" in r.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/etc"), reason="the reference tree is not on this machine")
def test_facade_reads_the_references_own_task_files(host_build, tmp_path):
    """the YAML-subset reader and TaskStack on the reference's OWN etc/<robot>/tasks.yaml files (comments,
    indentation and key order as shipped) give the sizes of SURVEY Appendix B.  CPU only, and only where /root/reference exists."""
    r = subprocess.run([host_build["test_facade"], "/root/reference/etc", str(tmp_path), "stacks-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK (0 failures)" in r.stdout


def test_hip_batched_solver_refuses_without_gpu(host_build, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    sys.path.insert(0, ROOT)
    from tools import dump_batch
    from inria_wbc_amd import structure, synth
    st = structure.talos_structure()
    path = str(tmp_path / "b.bin")
    dump_batch.dump(path, st, synth.generate(st, 2, synth.SEED_BASE["talos"]))
    r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/talos/pos_tracker.yaml"),
                        os.path.join(ROOT, "configs/talos/squat.yaml"), path, "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr, r.stderr


@pytest.mark.gpu
def test_qp_timer_test_squat_matches_oracle(host_build, oracle_mod, tmp_path):
    """The reference's qp_timer_test loop (qp_timer_test.cpp:55-63) with PosTracker + humanoid::move_com on 48 Talos
    instances; torques of the last tick against the oracle fed the same CoM reference (BASELINE config 4 stream)."""
    from tools import dump_batch
    from inria_wbc_amd import structure, synth
    st = structure.talos_structure()
    B, first_tick, n_ticks = 48, 700, 3
    inputs = synth.generate(st, B, synth.SEED_BASE["talos_squat"])
    path, tau_path = str(tmp_path / "b.bin"), str(tmp_path / "tau.bin")
    dump_batch.dump(path, st, inputs)
    r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/talos/pos_tracker.yaml"),
                        os.path.join(ROOT, "configs/talos/squat.yaml"), path, str(n_ticks), tau_path, str(first_tick)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "solver:" in r.stdout and "instances per tick: 48" in r.stdout
    tau = np.fromfile(tau_path, dtype=np.float64).reshape(B, st.na)
    rows = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    ref_in = {k: v.copy() for k, v in inputs.items()}
    ref_in["b1"][:, rows] += synth.squat_com_rhs(st, first_tick + n_ticks - 1, 30.0)
    ref = oracle_mod.tick_batch(st, ref_in, nthreads=4)
    assert (ref["status"] == 0).all()
    assert np.abs(tau - ref["tau"]).max() <= 1e-8 * max(1.0, np.abs(ref["tau"]).max())


@pytest.mark.gpu
def test_qp_timer_test_closed_loop_on_the_model(host_build, oracle_mod, tmp_path):
    """The same harness with the step before the path on the device too (CONTROLLER.model / frames / ref_config as the
    reference's urdf / frames / ref_config keys): PosTracker + humanoid::move_com close the loop through the integrated
    state for 40 ticks of the squat.  Every instance starts at the reference configuration, so all rows must stay
    bitwise equal (the reference's test_determinism.cpp:44-57 bar), and they must match the loop built from the oracles."""
    from inria_wbc_amd import model as mdl, structure, trajs
    from oracle import rbd
    n_ticks = 40
    tau_path, q_path = str(tmp_path / "tau.bin"), str(tmp_path / "q.bin")
    r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/talos/pos_tracker_model.yaml"),
                        os.path.join(ROOT, "configs/talos/squat.yaml"), "-", str(n_ticks), tau_path, "0", q_path],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "instances per tick: 8" in r.stdout
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    tau = np.fromfile(tau_path, dtype=np.float64).reshape(8, st.na)
    q = np.fromfile(q_path, dtype=np.float64).reshape(8, m.nq)
    assert all(np.array_equal(tau[0], tau[i]) and np.array_equal(q[0], q[i]) for i in range(1, 8))
    # the oracle loop: references of a fresh controller (placements at q0), CoM reference from the squat stream
    s = mdl.sample_states(m, tm, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    oq, ov, oref = s["q"], s["v"], s["ref"]
    com_blk = next(b for b in tm.blocks if b.kind == mdl.T_COM)
    pos, vel, acc = trajs.move_com_stream(m.com(m.q0), [[0.0, 0.0, -0.2]], "001", tm.dt, 2.0, loop=True, absolute=False)
    tl, tu, w = -m.tau_max[None], m.tau_max[None], st.default_weights[None]
    for k in range(n_ticks):
        oref[:, com_blk.ref:com_blk.ref + 9] = np.concatenate([pos[k], vel[k], acc[k]])
        rows = rbd.task_rows(m, tm, st, oq, ov, oref)
        oo = oracle_mod.tick_batch(st, dict(rows, tlb=tl, tub=tu, w=w))
        assert oo["status"][0] == 0
        nxt = oracle_mod.integrate(True, tm.dt, oq, ov, oo["x"][:, :st.nv])
        oq, ov = nxt["q_next"], nxt["v_next"]
    assert np.abs(q[0] - oq[0]).max() < 1e-8, np.abs(q[0] - oq[0]).max()
    assert np.abs(tau[0] - oo["tau"][0]).max() < 1e-6 * max(1.0, np.abs(oo["tau"]).max())
    # Controller::cost("com") = |A ddq - b| over the CoM rows of the last tick: the rows stayed on the device during the loop and
    # are fetched when cost() is asked for
    cr = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    A = rows["A"][0].reshape(st.n_dense, st.nv)[cr]
    want = np.linalg.norm(A @ oo["x"][0, :st.nv] - rows["b1"][0, cr])
    got = float([ln for ln in r.stdout.splitlines() if ln.startswith("cost com:")][0].split(":")[1])
    assert abs(got - want) < 1e-6 * max(1.0, want), (got, want)


@pytest.mark.gpu
def test_closed_loop_with_a_torque_task_a_cop_task_and_a_posture_mask(host_build, oracle_mod, tmp_path):
    """A tasks.yaml that uses what no shipped stack does -- `type: torque` with mask and scaling (tasks.cpp:227-271), `type: cop`
    (tasks.cpp:156-178) and `mask:` on the posture task (tasks.cpp:205-214) -- through PosTracker + humanoid::move_com for 25 ticks
    of the squat on the model: the facade accepts the file, the library runs the stack on the full layout with H as one matrix, and
    the final state equals the loop built from the oracles."""
    from inria_wbc_amd import model as mdl, structure, trajs
    from oracle import rbd
    m = mdl.talos_like()
    na = m.na
    tmask = np.zeros(na, int); tmask[:12] = 1                       # the legs' torques
    scaling = np.ones(na); scaling[:12] = np.linspace(0.5, 1.5, 12)
    pmask = np.ones(na, int); pmask[[21, 22, 30, 43]] = 0
    tasks = open(os.path.join(ROOT, "configs/talos/tasks.yaml")).read()
    assert "posture:" in tasks
    lines, out, in_posture = tasks.splitlines(), [], False
    for ln in lines:
        out.append(ln)
        if ln.startswith("posture:"):
            in_posture = True
        elif in_posture and ln.strip().startswith("type:"):
            out.append("  mask: %s" % "".join(str(b) for b in pmask))
            in_posture = False
    out += ["torque:", "  type: torque", "  weight: 0.02", "  mask: %s" % "".join(str(b) for b in tmask),
            "  scaling: [%s]" % ", ".join("%.17g" % x for x in scaling), "cop:", "  type: cop", "  weight: 5.0"]
    (tmp_path / "tasks_extra.yaml").write_text("\n".join(out) + "\n")
    base = open(os.path.join(ROOT, "configs/talos/pos_tracker_model.yaml")).read()
    for key in ("model", "frames"):
        base = base.replace("  %s: " % key, "  %s: %s/" % (key, os.path.join(ROOT, "configs/talos")))
    base = base.replace("  tasks: tasks.yaml", "  tasks: %s" % (tmp_path / "tasks_extra.yaml")).replace("  verbose: false", "  verbose: true")
    cfg = tmp_path / "pos_tracker_extra.yaml"
    cfg.write_text(base)
    n_ticks = 25
    tau_path, q_path = str(tmp_path / "tau.bin"), str(tmp_path / "q.bin")
    r = subprocess.run([host_build["qp_timer_test"], str(cfg), os.path.join(ROOT, "configs/talos/squat.yaml"), "-", str(n_ticks), tau_path, "0", q_path],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    st = structure.with_cop_task(structure.with_torque_task(structure.with_posture_mask(structure.talos_structure(), pmask), 0.02, mask=tmask,
                                                            scaling=scaling), 5.0)
    stack = [dict(n, mask="".join(str(b) for b in pmask)) if n["type"] == "posture" else n for n in mdl.talos_stack()]
    stack += [dict(name="torque", type="torque", weight=0.02, mask="".join(str(b) for b in tmask)), dict(name="cop", type="cop", weight=5.0)]
    tm = mdl.build_taskmap(m, st, stack)
    tau = np.fromfile(tau_path, dtype=np.float64).reshape(8, st.na)
    q = np.fromfile(q_path, dtype=np.float64).reshape(8, m.nq)
    assert all(np.array_equal(tau[0], tau[i]) and np.array_equal(q[0], q[i]) for i in range(1, 8))
    s = mdl.sample_states(m, tm, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    oq, ov, oref = s["q"], s["v"], s["ref"]
    com_blk = next(b for b in tm.blocks if b.kind == mdl.T_COM)
    pos, vel, acc = trajs.move_com_stream(m.com(m.q0), [[0.0, 0.0, -0.2]], "001", tm.dt, 2.0, loop=True, absolute=False)
    tl, tu, w = -m.tau_max[None], m.tau_max[None], st.default_weights[None]
    for k in range(n_ticks):
        oref[:, com_blk.ref:com_blk.ref + 9] = np.concatenate([pos[k], vel[k], acc[k]])
        rows = rbd.task_rows(m, tm, st, oq, ov, oref)
        oo = oracle_mod.tick_batch(st, dict(rows, tlb=tl, tub=tu, w=w))
        assert oo["status"][0] == 0
        nxt = oracle_mod.integrate(True, tm.dt, oq, ov, oo["x"][:, :st.nv])
        oq, ov = nxt["q_next"], nxt["v_next"]
    assert np.abs(q[0] - oq[0]).max() < 1e-8, np.abs(q[0] - oq[0]).max()
    assert np.abs(tau[0] - oo["tau"][0]).max() < 1e-6 * max(1.0, np.abs(oo["tau"]).max())
    # and the extra tasks did change the motion: the plain stack ends elsewhere
    plain = structure.talos_structure()
    tmp = mdl.build_taskmap(m, plain, mdl.talos_stack())
    sp = mdl.sample_states(m, tmp, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    rows = rbd.task_rows(m, tmp, plain, sp["q"], sp["v"], sp["ref"])
    o2 = oracle_mod.tick_batch(plain, dict(rows, tlb=tl, tub=tu, w=plain.default_weights[None]))
    rows1 = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"])
    o1 = oracle_mod.tick_batch(st, dict(rows1, tlb=tl, tub=tu, w=w))
    assert np.abs(o1["tau"] - o2["tau"]).max() > 1e-3


@pytest.mark.gpu
def test_mimic_filter_momentum_and_step_back(host_build, oracle_mod, tmp_path):
    """What the reference's robot-side consumers read after a tick (controller.cpp:208-229,245,369-397,445-450): tau() / q() with the
    mimic joints of CONTROLLER.mimic_dof_names filtered out (the real Talos has twelve: etc/talos/talos_pos_tracker.yaml:20-31),
    momentum() = the angular momentum about the CoM of the state the tick was solved at, and qp_step_back()."""
    from inria_wbc_amd import model as mdl, structure, trajs
    from oracle import rbd
    mimics = ["gripper_left_inner_double_joint", "gripper_left_fingertip_1_joint", "gripper_left_fingertip_2_joint",
              "gripper_left_inner_single_joint", "gripper_left_fingertip_3_joint", "gripper_left_motor_single_joint",
              "gripper_right_inner_double_joint", "gripper_right_fingertip_1_joint", "gripper_right_fingertip_2_joint",
              "gripper_right_inner_single_joint", "gripper_right_fingertip_3_joint", "gripper_right_motor_single_joint"]
    base = open(os.path.join(ROOT, "configs/talos/pos_tracker_model.yaml")).read()
    for key in ("model", "frames", "tasks"):  # the harness sets base_path to the configuration file's directory: absolute paths
        base = base.replace("  %s: " % key, "  %s: %s/" % (key, os.path.join(ROOT, "configs/talos")))
    cfg = tmp_path / "pos_tracker_mimic.yaml"
    cfg.write_text(base + "  floating_base_joint_name: root_joint\n  mimic_dof_names: [%s]\n" % ", ".join('"%s"' % m for m in mimics))
    n_ticks = 6
    tau_path, q_path, cmd_path = str(tmp_path / "tau.bin"), str(tmp_path / "q.bin"), str(tmp_path / "cmd.bin")
    env = dict(os.environ, IWBC_DUMP_COMMAND=cmd_path, IWBC_STEP_BACK="1")
    r = subprocess.run([host_build["qp_timer_test"], str(cfg), os.path.join(ROOT, "configs/talos/squat.yaml"), "-", str(n_ticks), tau_path, "0", q_path],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    assert "command columns: %d of %d dofs (12 mimic joints filtered)" % (st.nv - 12, st.nv) in r.stdout, r.stdout
    # the filtered command = the non-mimic columns of [0 x 6, tau_tsid] (the dump of tau_tsid is taken after the extra tick of the
    # step-back check, so the command is compared with the oracle loop instead)
    names = ["rootJoint_pos_x", "rootJoint_pos_y", "rootJoint_pos_z", "rootJoint_rot_x", "rootJoint_rot_y", "rootJoint_rot_z"] + list(m.joint_names[1:])
    assert len(names) == st.nv
    keep = [i for i, nme in enumerate(names) if nme not in mimics]
    cmd = np.fromfile(cmd_path, dtype=np.float64)
    ncol = st.nv - 12
    tau_cmd, q_cmd = cmd[:8 * ncol].reshape(8, ncol), cmd[8 * ncol:].reshape(8, ncol)
    s = mdl.sample_states(m, tm, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    oq, ov, oref = s["q"], s["v"], s["ref"]
    com_blk = next(b for b in tm.blocks if b.kind == mdl.T_COM)
    pos, vel, acc = trajs.move_com_stream(m.com(m.q0), [[0.0, 0.0, -0.2]], "001", tm.dt, 2.0, loop=True, absolute=False)
    tl, tu, w = -m.tau_max[None], m.tau_max[None], st.default_weights[None]
    for k in range(n_ticks):
        oref[:, com_blk.ref:com_blk.ref + 9] = np.concatenate([pos[k], vel[k], acc[k]])
        rows = rbd.task_rows(m, tm, st, oq, ov, oref)
        oo = oracle_mod.tick_batch(st, dict(rows, tlb=tl, tub=tu, w=w))
        q_start, v_start = oq.copy(), ov.copy()
        nxt = oracle_mod.integrate(True, tm.dt, oq, ov, oo["x"][:, :st.nv])
        oq, ov = nxt["q_next"], nxt["v_next"]
    tau_full = np.concatenate([np.zeros(6), oo["tau"][0]])
    assert np.abs(tau_cmd[0] - tau_full[keep]).max() < 1e-6 * max(1.0, np.abs(tau_full).max())
    assert np.abs(q_cmd[0] - nxt["q_solver"][0][keep]).max() < 1e-8
    # momentum(): Ag(q) v at the state the last tick was solved at, angular part
    want = (rbd.rbd_terms(m, q_start[0], v_start[0])["Ag"] @ v_start[0])[3:]
    got = np.array([float(x) for x in [ln for ln in r.stdout.splitlines() if ln.startswith("momentum[0]:")][0].split(":")[1].split()])
    assert np.abs(got - want).max() < 1e-9 * max(1.0, np.abs(want).max()), (got, want)
    # qp_step_back(): q returns to where the tick started and the redone tick lands where the first one did
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("step back moved q by")][0]
    moved, diff = float(line.split("by")[1].split(",")[0]), float(line.rsplit("by", 1)[1])
    assert moved > 0.0 and diff < 1e-12, line


@pytest.mark.gpu
def test_closed_loop_takes_reference_shaped_sensor_data(host_build, tmp_path):
    """Controller::update in closed loop (controller.cpp:161-205): the sensor keys the reference requires are required here,
    a floating base arrives split into floating_base_position / _velocity + joints, and feeding the controller's own
    integrated state back through them gives bitwise the open-loop run.  Also: the same file loads as "talos-pos-tracker"
    (talos_pos_tracker.cpp:35), and a configuration that switches the stabiliser on is refused with a message that says why."""
    cfg = os.path.join(ROOT, "configs/talos/pos_tracker_model.yaml")
    beh = os.path.join(ROOT, "configs/talos/squat.yaml")
    outs = {}
    for mode, env in (("open", {}), ("sensors", {"IWBC_SENSOR_LOOP": "1"}), ("talos", {"IWBC_CONTROLLER_NAME": "talos-pos-tracker"})):
        tau_path, q_path = str(tmp_path / ("tau_%s.bin" % mode)), str(tmp_path / ("q_%s.bin" % mode))
        r = subprocess.run([host_build["qp_timer_test"], cfg, beh, "-", "25", tau_path, "0", q_path], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        outs[mode] = (np.fromfile(tau_path, dtype=np.float64), np.fromfile(q_path, dtype=np.float64))
    for mode in ("sensors", "talos"):
        assert np.array_equal(outs["open"][0], outs[mode][0]) and np.array_equal(outs["open"][1], outs[mode][1]), mode
    r = subprocess.run([host_build["qp_timer_test"], cfg, beh, "-", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, IWBC_SENSOR_LOOP="missing"))
    assert r.returncode != 0 and "we need the joint positions in closed loop mode" in (r.stdout + r.stderr)
    text = open(cfg).read()
    assert "CONTROLLER:" in text
    stab = str(tmp_path / "stab.yaml")
    with open(stab, "w") as fh:
        fh.write(text.rstrip("\n") + "\n  stabilizer:\n    activated: true\n")
    for name in ("base_path",):
        pass
    # the harness derives base_path from the file's directory: keep the relative files reachable
    for f in os.listdir(os.path.dirname(cfg)):
        src = os.path.join(os.path.dirname(cfg), f)
        if os.path.isfile(src) and not os.path.exists(str(tmp_path / f)):
            os.symlink(src, str(tmp_path / f))
    r = subprocess.run([host_build["qp_timer_test"], stab, beh, "-", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, IWBC_CONTROLLER_NAME="humanoid-pos-tracker"))
    assert r.returncode != 0 and "stabilizer" in (r.stdout + r.stderr) and "not part of this build" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_qp_timer_test_franka_cartesian_line(host_build, oracle_mod, tmp_path):
    """BASELINE config 1 on the model: PosTracker + generic::cartesian (etc/franka/cartesian_line.yaml: ee 0.4 m along -x in
    2 s, min-jerk) on the Franka-like arm, closed loop for 600 ticks, against the oracle loop fed the same SE(3) stream, and
    against what the behaviour is for: the end effector follows the line."""
    from inria_wbc_amd import model as mdl, structure, trajs
    from oracle import rbd
    n_ticks = 600
    tau_path, q_path = str(tmp_path / "tau.bin"), str(tmp_path / "q.bin")
    r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/franka/pos_tracker_model.yaml"),
                        os.path.join(ROOT, "configs/franka/cartesian_line.yaml"), "-", str(n_ticks), tau_path, "0", q_path],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = mdl.franka_like()
    st = structure.franka_structure()
    tm = mdl.build_taskmap(m, st, mdl.franka_stack())
    q = np.fromfile(q_path, dtype=np.float64).reshape(4, m.nq)
    tau = np.fromfile(tau_path, dtype=np.float64).reshape(4, st.na)
    assert all(np.array_equal(q[0], q[i]) for i in range(1, 4))
    s = mdl.sample_states(m, tm, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    oq, ov, oref = s["q"], s["v"], s["ref"]
    ee = m.frame("panda_joint7")
    Rf0, pf0 = m.frame_placements(m.q0)
    Rs, ps, d1, d2 = trajs.cartesian_stream(Rf0[ee], pf0[ee], [-0.4, 0.0, 0.0], tm.dt, 2.0, loop=True)
    blk = tm.blocks[0]
    w = st.default_weights[None]
    empty = np.zeros((1, 0))
    for k in range(n_ticks):
        oref[0, blk.ref:blk.ref + 12] = mdl.se3_ref(Rs[k], ps[k])
        oref[0, blk.ref + 12:blk.ref + 18] = d1[k]
        oref[0, blk.ref + 18:blk.ref + 24] = d2[k]
        rows = rbd.task_rows(m, tm, st, oq, ov, oref)
        oo = oracle_mod.tick_batch(st, dict(rows, tlb=empty, tub=empty, w=w))
        assert oo["status"][0] == 0
        nxt = oracle_mod.integrate(False, tm.dt, oq, ov, oo["x"][:, :st.nv])
        oq, ov = nxt["q_next"], nxt["v_next"]
    assert np.abs(q[0] - oq[0]).max() < 1e-8, np.abs(q[0] - oq[0]).max()
    assert np.abs(tau[0] - oo["tau"][0]).max() < 1e-6 * max(1.0, np.abs(oo["tau"]).max())
    # the end effector follows the line; the posture task pulls the other way, so it lags by centimetres (the reference's own
    # Franka test accepts a mean tracking error of 0.09 m: tests/ref_test_franka.yaml:13-15)
    p_now = m.frame_placements(q[0])[1][ee]
    assert np.abs(p_now - ps[n_ticks - 1]).max() < 0.06, (p_now, ps[n_ticks - 1])
    assert pf0[ee][0] - p_now[0] > 0.02


@pytest.mark.gpu
def test_qp_timer_test_franka_cartesian_traj_files(host_build, oracle_mod, tmp_path):
    """generic::cartesian_traj: SE(3) references replayed from trajectory files in the reference's wire format
    (src/trajs/loader.cpp:11-53: 3 + 9 numbers per sample, rotation column-major), forward then backward, against the oracle
    loop stepping through the same samples."""
    from scipy.spatial.transform import Rotation as Rot
    from inria_wbc_amd import model as mdl, structure
    from oracle import rbd
    m = mdl.franka_like()
    st = structure.franka_structure()
    tm = mdl.build_taskmap(m, st, mdl.franka_stack())
    ee = m.frame("panda_joint7")
    Rf0, pf0 = m.frame_placements(m.q0)
    n_samples, n_ticks, scale = 60, 150, 1.0
    poses = []
    for k in range(n_samples):
        a = k / (n_samples - 1.0)
        R = Rf0[ee] @ Rot.from_rotvec([0.3 * a, -0.2 * a, 0.1 * a]).as_matrix()
        poses.append(mdl.se3_ref(R, pf0[ee] + np.array([-0.1 * a, 0.05 * a, 0.02 * np.sin(3 * a)])))
    np.savetxt(tmp_path / "ee.csv", np.array(poses))
    (tmp_path / "refs.yaml").write_text("refs:\n  ee: ee.csv\n")
    (tmp_path / "traj.yaml").write_text("BEHAVIOR:\n  name: generic::cartesian_traj\n  trajectories: %s\n  scale: %g\n  loop: true\n"
                                        % (tmp_path / "refs.yaml", scale))
    tau_path, q_path = str(tmp_path / "tau.bin"), str(tmp_path / "q.bin")
    r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/franka/pos_tracker_model.yaml"), str(tmp_path / "traj.yaml"),
                        "-", str(n_ticks), tau_path, "0", q_path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    q = np.fromfile(q_path, dtype=np.float64).reshape(4, m.nq)
    s = mdl.sample_states(m, tm, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    oq, ov, oref = s["q"], s["v"], s["ref"]
    blk = tm.blocks[0]
    empty, w = np.zeros((1, 0)), st.default_weights[None]
    t, step, visited = 0, 1, []
    for k in range(n_ticks):
        visited.append(t)
        oref[0, blk.ref:blk.ref + 24] = 0.0
        oref[0, blk.ref:blk.ref + 12] = poses[t]
        rows = rbd.task_rows(m, tm, st, oq, ov, oref)
        oo = oracle_mod.tick_batch(st, dict(rows, tlb=empty, tub=empty, w=w))
        nxt = oracle_mod.integrate(False, tm.dt, oq, ov, oo["x"][:, :st.nv])
        oq, ov = nxt["q_next"], nxt["v_next"]
        t += step
        if t >= n_samples - 1:
            step = -1
        elif t <= 0:
            step = 1
    assert max(visited) == n_samples - 1 and visited[-1] < n_samples - 1  # went to the end and is on its way back
    assert np.abs(q[0] - oq[0]).max() < 1e-8, np.abs(q[0] - oq[0]).max()


@pytest.mark.gpu
def test_qp_timer_test_walk_on_spot_changes_the_qp_mid_run(host_build, oracle_mod, tmp_path):
    """humanoid::walk-on-spot (etc/talos/walk_on_spot.yaml) on the Talos-like model: the CoM goes over the right foot, the left
    contact is removed (n 74 -> 62), the foot is lifted 0.1 m and put down, the contact comes back (SURVEY 3.4:
    walk_on_spot.cpp:165-184, pos_tracker.cpp:246-263).  The state after 2000 and 3100 ticks is checked against the loop of the
    three oracles driven by the same state machine, and against the physics: the foot is up, then down again, the support foot
    never moves."""
    from inria_wbc_amd import model as mdl, structure, trajs
    from oracle import rbd
    m = mdl.talos_like()
    full = (structure.talos_structure(), mdl.talos_stack())
    ss = (structure.talos_structure(single_support=True), [n for n in mdl.talos_stack() if n["name"] != "contact_lfoot"])
    maps = {True: (full[0], mdl.build_taskmap(m, *full)), False: (ss[0], mdl.build_taskmap(m, *ss))}
    lf, rf = m.frame("leg_left_6_joint"), m.frame("leg_right_6_joint")
    Rf0, pf0 = m.frame_placements(m.q0)
    com0 = m.com(m.q0)
    dt, T = 1e-3, 1.0
    n = int(np.floor(T / dt))
    lf_low, rf_low = (Rf0[lf], pf0[lf]), (Rf0[rf], pf0[rf])
    up = np.array([0.0, 0.0, 0.1])
    com_rf, com_lf = np.array([pf0[rf][0], pf0[rf][1], com0[2]]), np.array([pf0[lf][0], pf0[lf][1], com0[2]])
    const = lambda Rp: [mdl.se3_ref(*Rp)] * n
    move = lambda a, b: [mdl.se3_ref(R, p) for R, p in zip(*trajs.min_jerk_se3(a[0], a[1], b[0], b[1], dt, T)[:2])]
    mj = lambda a, b: list(trajs.min_jerk_trajectory(a, b, dt, T, 0))
    lf_high = (lf_low[0], lf_low[1] + up)
    # phases INIT, LIFT_UP_LF, LIFT_DOWN_LF, MOVE_COM_LEFT: (lf, rf, com, both feet in contact at the START of the tick)
    phases = [(const(lf_low), const(rf_low), mj(com0, com_rf)), (move(lf_low, lf_high), const(rf_low), [com_rf] * n),
              (move(lf_high, lf_low), const(rf_low), [com_rf] * n), (const(lf_low), const(rf_low), mj(com_rf, com_lf))]
    w = structure.talos_structure().default_weights.copy()
    w[structure.talos_structure().task_names.index("momentum")] = 0.0  # customize_task_weights

    def oracle_loop(n_ticks):
        both = True
        s = mdl.sample_states(m, maps[True][1], 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
        oq, ov = s["q"], s["v"]
        named = {}
        for k in range(n_ticks):
            ph, t = divmod(k, n)
            if ph == 1 and t == 0:
                both = False  # remove_contact("contact_lfoot")
            if ph == 2 and t == n - 1:
                both = True   # add_contact("contact_lfoot")
            st, tm = maps[both]
            named.update(lf=phases[ph][0][t], rf=phases[ph][1][t], com=phases[ph][2][t], contact_lfoot=phases[ph][0][t], contact_rfoot=phases[ph][1][t])
            ref = mdl.sample_states(m, tm, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)["ref"]
            for b in tm.blocks:
                if b.name in ("lf", "rf"):
                    ref[0, b.ref:b.ref + 24] = 0.0
                    ref[0, b.ref:b.ref + 12] = named[b.name]
                elif b.kind == mdl.T_COM:
                    ref[0, b.ref:b.ref + 9] = 0.0
                    ref[0, b.ref:b.ref + 3] = named["com"]
            cnames = ["contact_lfoot", "contact_rfoot"] if both else ["contact_rfoot"]
            for c, nm in enumerate(cnames):
                ref[0, tm.contact_ref[c]:tm.contact_ref[c] + 12] = named[nm]
            rows = rbd.task_rows(m, tm, st, oq, ov, ref)
            ww = w if both else w[[i for i, nm in enumerate(maps[True][0].task_names) if nm in st.task_names]]
            oo = oracle_mod.tick_batch(st, dict(rows, tlb=-m.tau_max[None], tub=m.tau_max[None], w=ww[None]))
            assert oo["status"][0] == 0, (k, oo["status"])
            nxt = oracle_mod.integrate(True, dt, oq, ov, oo["x"][:, :st.nv])
            oq, ov = nxt["q_next"], nxt["v_next"]
        return oq[0]

    results = {}
    for n_ticks in (2000, 3100):
        q_path = str(tmp_path / ("q%d.bin" % n_ticks))
        r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/talos/pos_tracker_model.yaml"),
                            os.path.join(ROOT, "configs/talos/walk_on_spot.yaml"), "-", str(n_ticks), str(tmp_path / "tau.bin"), "0", q_path],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr
        q = np.fromfile(q_path, dtype=np.float64).reshape(8, m.nq)
        assert all(np.array_equal(q[0], q[i]) for i in range(1, 8))
        results[n_ticks] = q[0]
        oq = oracle_loop(n_ticks)
        assert np.abs(q[0] - oq).max() < 1e-6, (n_ticks, np.abs(q[0] - oq).max())
    p_up = m.frame_placements(results[2000])[1]
    p_down = m.frame_placements(results[3100])[1]
    assert p_up[lf][2] - pf0[lf][2] > 0.05, p_up[lf][2] - pf0[lf][2]       # the left foot is up (no feed-forward in these references: it lags)
    assert abs(p_down[lf][2] - pf0[lf][2]) < 0.03                          # and (nearly: 0.1 s after the contact came back) down again
    assert np.abs(p_up[rf] - pf0[rf]).max() < 2e-3 and np.abs(p_down[rf] - pf0[rf]).max() < 2e-3  # the support foot stays
    assert abs(m.com(results[2000])[1] - pf0[rf][1]) < 0.02                # the CoM is over the support foot


def _talos_model_loop(oracle_mod, n_ticks, fill_ref):
    """The oracle loop of the Talos-like model with the full stack: fill_ref(k, ref, tm) writes the references of tick k."""
    from inria_wbc_amd import model as mdl, structure
    from oracle import rbd
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    s = mdl.sample_states(m, tm, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    oq, ov, ref = s["q"], s["v"], s["ref"]
    for k in range(n_ticks):
        fill_ref(k, ref, tm)
        rows = rbd.task_rows(m, tm, st, oq, ov, ref)
        oo = oracle_mod.tick_batch(st, dict(rows, tlb=-m.tau_max[None], tub=m.tau_max[None], w=st.default_weights[None]))
        assert oo["status"][0] == 0, (k, oo["status"])
        nxt = oracle_mod.integrate(True, tm.dt, oq, ov, oo["x"][:, :st.nv])
        oq, ov = nxt["q_next"], nxt["v_next"]
    return m, tm, oq[0]


def _run_model_harness(host_build, tmp_path, behavior_yaml, n_ticks):
    q_path = str(tmp_path / "q.bin")
    r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/talos/pos_tracker_model.yaml"), behavior_yaml, "-", str(n_ticks),
                        str(tmp_path / "tau.bin"), "0", q_path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    q = np.fromfile(q_path, dtype=np.float64).reshape(8, -1)
    assert all(np.array_equal(q[0], q[i]) for i in range(1, 8))
    return q[0]


@pytest.mark.gpu
def test_qp_timer_test_move_feet_moves_the_contacts(host_build, oracle_mod, tmp_path):
    """humanoid::move-feet (etc/talos/move_feet.yaml): the feet tasks AND the contacts' motion tasks follow the same min-jerk
    sample (pose, velocity, acceleration): PosTracker::set_contact_se3_ref(sample, name), pos_tracker.cpp:240-244."""
    from inria_wbc_amd import model as mdl, trajs
    n_ticks = 700
    q = _run_model_harness(host_build, tmp_path, os.path.join(ROOT, "configs/talos/move_feet.yaml"), n_ticks)
    m0 = mdl.talos_like()
    Rf0, pf0 = m0.frame_placements(m0.q0)
    streams = {nm: trajs.cartesian_stream(Rf0[m0.frame(fr)], pf0[m0.frame(fr)], [0.05, 0.05, 0.05], 1e-3, 4.0, loop=False, rel_rpy=[0.0, 0.0, 0.0])
               for nm, fr in (("lf", "leg_left_6_joint"), ("rf", "leg_right_6_joint"))}

    def fill(k, ref, tm):
        for b in tm.blocks:
            if b.name in streams:
                Rs, ps, d1, d2 = streams[b.name]
                sample = np.concatenate([mdl.se3_ref(Rs[k], ps[k]), d1[k], d2[k]])
                ref[0, b.ref:b.ref + 24] = sample
                c = 0 if b.name == "lf" else 1
                ref[0, tm.contact_ref[c]:tm.contact_ref[c] + 24] = sample

    m, tm, oq = _talos_model_loop(oracle_mod, n_ticks, fill)
    assert np.abs(q - oq).max() < 1e-7, np.abs(q - oq).max()
    lf = m.frame("leg_left_6_joint")
    moved = m.frame_placements(q)[1][lf] - pf0[lf]
    want = streams["lf"][1][n_ticks - 1] - pf0[lf]
    assert np.abs(moved - want).max() < 2e-3 and want[0] > 1e-3, (moved, want)  # the contact constraint drags the foot along its sample


@pytest.mark.gpu
def test_qp_timer_test_clapping(host_build, oracle_mod, tmp_path):
    """humanoid::clapping (etc/talos/clapping.yaml): both hands move along y towards each other, pose-only references."""
    from inria_wbc_amd import model as mdl, trajs
    n_ticks = 500
    q = _run_model_harness(host_build, tmp_path, os.path.join(ROOT, "configs/talos/clapping.yaml"), n_ticks)
    m0 = mdl.talos_like()
    Rf0, pf0 = m0.frame_placements(m0.q0)
    streams = {}
    for nm, fr, dy in (("lh", "gripper_left_joint", -0.1), ("rh", "gripper_right_joint", 0.1)):
        f = m0.frame(fr)
        streams[nm] = trajs.cartesian_stream(Rf0[f], pf0[f], [0.0, dy, 0.0], 1e-3, 1.0, loop=True)

    def fill(k, ref, tm):
        for b in tm.blocks:
            if b.name in streams:
                Rs, ps, _, _ = streams[b.name]
                ref[0, b.ref:b.ref + 24] = 0.0
                ref[0, b.ref:b.ref + 12] = mdl.se3_ref(Rs[k], ps[k])

    m, tm, oq = _talos_model_loop(oracle_mod, n_ticks, fill)
    assert np.abs(q - oq).max() < 1e-7, np.abs(q - oq).max()
    lh, rh = m.frame("gripper_left_joint"), m.frame("gripper_right_joint")
    p = m.frame_placements(q)[1]
    # the hands came closer: weight 10 against posture, self-collision and 1000-weight tasks, and no feed-forward -- by millimetres
    assert (pf0[lh][1] - pf0[rh][1]) - (p[lh][1] - p[rh][1]) > 0.005


@pytest.mark.gpu
def test_qp_timer_test_walk_takes_steps(host_build, oracle_mod, tmp_path):
    """humanoid::walk (walk.cpp:8-246) on the Talos-like model: one cycle = two steps forward plus the closing half step, eight
    contact switches; the final state against the loop of the three oracles driven by a Python restatement of the same state
    machine, and against the point of it: both feet and the CoM end 2 x step_length further."""
    from inria_wbc_amd import model as mdl, structure, trajs
    from oracle import rbd
    Tc = Tf = 0.5
    H, Lstep, dt = 0.05, 0.1, 1e-3
    beh = tmp_path / "walk.yaml"
    beh.write_text("BEHAVIOR:\n  name: humanoid::walk\n  traj_com_duration: %g\n  traj_foot_duration: %g\n  step_height: %g\n  step_length: %g\n"
                   "  num_of_cycles: 1\n  customize_task_weights:\n    momentum: 0.0\n" % (Tc, Tf, H, Lstep))
    m = mdl.talos_like()
    full_st = structure.talos_structure()
    stacks = {"both": mdl.talos_stack(), "no_l": [n for n in mdl.talos_stack() if n["name"] != "contact_lfoot"],
              "no_r": [n for n in mdl.talos_stack() if n["name"] != "contact_rfoot"]}
    sts = {"both": full_st, "no_l": structure.talos_structure(single_support=True), "no_r": structure.talos_structure(single_support=True)}
    # the single-support structure of structure.py keeps the RIGHT contact; the stack without the right one has the same shape
    maps = {k: mdl.build_taskmap(m, sts[k], stacks[k]) for k in stacks}
    fid = {nm: m.frame(fr) for nm, fr in (("lf", "leg_left_6_joint"), ("rf", "leg_right_6_joint"), ("lh", "gripper_left_joint"), ("rh", "gripper_right_joint"))}
    Rf0, pf0 = m.frame_placements(m.q0)
    R = {k: Rf0[f] for k, f in fid.items()}
    n = int(np.floor(Tc / dt))
    const = lambda nm, p: [mdl.se3_ref(R[nm], p)] * n
    move = lambda nm, a, b: [mdl.se3_ref(R[nm], p) for p in trajs.min_jerk_trajectory(a, b, dt, Tc, 0)]  # no rotation in this walk
    mj = lambda a, b: list(trajs.min_jerk_trajectory(a, b, dt, Tc, 0))
    up = np.array([0.0, 0.0, H])
    lf_low, rf_low = pf0[fid["lf"]].copy(), pf0[fid["rf"]].copy()
    lf_high, rf_high = lf_low + up, rf_low + up
    com_init = m.com(m.q0).copy()
    com_lf, com_rf = np.array([lf_low[0], lf_low[1], com_init[2]]), np.array([rf_low[0], rf_low[1], com_init[2]])
    lh_init, rh_init = pf0[fid["lh"]].copy(), pf0[fid["rh"]].copy()
    lh_fwd, rh_fwd = lh_init.copy(), rh_init.copy()
    ex = np.array([1.0, 0.0, 0.0])
    phases = []  # (name, lf, rf, com, lh, rh)
    cycle = ["INIT", "LF_INIT", "LIFT_DOWN_LF", "MOVE_COM_LEFT", "LIFT_UP_RF", "LIFT_DOWN_RF", "MOVE_COM_RIGHT", "LIFT_UP_LF", "LIFT_DOWN_LF_FINAL",
             "MOVE_COM_CENTER_FINAL"]
    for c in cycle:
        if c == "INIT":
            phases.append((c, const("lf", lf_low), const("rf", rf_low), mj(com_init, com_rf), const("lh", lh_init), const("rh", rh_init)))
        elif c == "LF_INIT":
            phases.append((c, move("lf", lf_low, lf_high), const("rf", rf_low), [com_rf.copy()] * n, const("lh", lh_init), const("rh", rh_init)))
        elif c == "LIFT_DOWN_LF":
            diff = rf_low[0] - lf_low[0]
            lf_low = lf_low + (diff + Lstep) * ex
            lh_fwd = lh_init + (diff + Lstep) * ex
            phases.append((c, move("lf", lf_high, lf_low), const("rf", rf_low), [com_rf.copy()] * n, move("lh", lh_init, lh_fwd), const("rh", rh_init)))
            lh_init = lh_fwd
        elif c == "MOVE_COM_LEFT":
            com_lf = com_lf.copy(); com_lf[0] = lf_low[0]
            phases.append((c, const("lf", lf_low), const("rf", rf_low), mj(com_rf, com_lf), const("lh", lh_init), const("rh", rh_init)))
        elif c == "LIFT_UP_RF":
            rf_high = rf_high.copy(); rf_high[0] = lf_low[0]
            phases.append((c, const("lf", lf_low), move("rf", rf_low, rf_high), [com_lf.copy()] * n, const("lh", lh_init), const("rh", rh_init)))
        elif c == "LIFT_DOWN_RF":
            rf_low = rf_low + 2 * Lstep * ex
            rh_fwd = rh_init + 2 * Lstep * ex
            phases.append((c, const("lf", lf_low), move("rf", rf_high, rf_low), [com_lf.copy()] * n, const("lh", lh_init), move("rh", rh_init, rh_fwd)))
            rh_init = rh_fwd
        elif c == "MOVE_COM_RIGHT":
            com_rf = com_rf.copy(); com_rf[0] = rf_low[0]
            phases.append((c, const("lf", lf_low), const("rf", rf_low), mj(com_lf, com_rf), const("lh", lh_init), const("rh", rh_init)))
        elif c == "LIFT_UP_LF":
            lf_high = lf_high.copy(); lf_high[0] = rf_low[0]
            phases.append((c, move("lf", lf_low, lf_high), const("rf", rf_low), [com_rf.copy()] * n, const("lh", lh_init), const("rh", rh_init)))
        elif c == "LIFT_DOWN_LF_FINAL":
            lf_low = lf_low + Lstep * ex
            lh_fwd = lh_init + Lstep * ex
            phases.append((c, move("lf", lf_high, lf_low), const("rf", rf_low), [com_rf.copy()] * n, move("lh", lh_init, lh_fwd), const("rh", rh_init)))
        else:
            com_c = com_init.copy(); com_c[:2] = (rf_low[:2] + lf_low[:2]) / 2.0
            phases.append((c, const("lf", lf_low), const("rf", rf_low), mj(com_rf, com_c), const("lh", lh_fwd), const("rh", rh_fwd)))
    n_ticks = len(phases) * n
    q = _run_model_harness(host_build, tmp_path, str(beh), n_ticks)
    w_full = full_st.default_weights.copy()
    w_full[full_st.task_names.index("momentum")] = 0.0
    s = mdl.sample_states(m, maps["both"], 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    oq, ov = s["q"], s["v"]
    mode, switches = "both", 0
    for k in range(n_ticks):
        ph, t = divmod(k, n)
        name, lfp, rfp, comp, lhp, rhp = phases[ph]
        before = mode
        if t == 0 and name in ("LIFT_UP_LF", "LF_INIT"):
            mode = "no_l"
        if t == 0 and name == "LIFT_UP_RF":
            mode = "no_r"
        if t == n - 1 and name in ("LIFT_DOWN_LF", "LIFT_DOWN_LF_FINAL", "LIFT_DOWN_RF"):
            mode = "both"
        switches += mode != before
        st, tm = sts[mode], maps[mode]
        ref = mdl.sample_states(m, tm, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)["ref"]
        poses = dict(lf=lfp[t], rf=rfp[t], lh=lhp[t], rh=rhp[t])
        for b in tm.blocks:
            if b.name in poses:
                ref[0, b.ref:b.ref + 24] = 0.0
                ref[0, b.ref:b.ref + 12] = poses[b.name]
            elif b.kind == mdl.T_COM:
                ref[0, b.ref:b.ref + 9] = 0.0
                ref[0, b.ref:b.ref + 3] = comp[t]
        cnames = {"both": ["contact_lfoot", "contact_rfoot"], "no_l": ["contact_rfoot"], "no_r": ["contact_lfoot"]}[mode]
        for c, nm in enumerate(cnames):
            ref[0, tm.contact_ref[c]:tm.contact_ref[c] + 24] = 0.0
            ref[0, tm.contact_ref[c]:tm.contact_ref[c] + 12] = poses["lf" if nm == "contact_lfoot" else "rf"]
        ww = w_full if mode == "both" else np.array([w_full[full_st.task_names.index(nm)] if nm in full_st.task_names else structure.W_FORCE_FEET
                                                      for nm in st.task_names])
        rows = rbd.task_rows(m, tm, st, oq, ov, ref)
        oo = oracle_mod.tick_batch(st, dict(rows, tlb=-m.tau_max[None], tub=m.tau_max[None], w=ww[None]))
        assert oo["status"][0] == 0, (k, name, oo["status"])
        nxt = oracle_mod.integrate(True, dt, oq, ov, oo["x"][:, :st.nv])
        oq, ov = nxt["q_next"], nxt["v_next"]
    assert switches == 6  # lf off/on, rf off/on, lf off/on
    assert np.abs(q - oq[0]).max() < 1e-5, np.abs(q - oq[0]).max()
    pf = m.frame_placements(q)[1]
    assert abs(pf[fid["lf"]][0] - pf0[fid["lf"]][0] - 2 * Lstep) < 5e-3 and abs(pf[fid["rf"]][0] - pf0[fid["rf"]][0] - 2 * Lstep) < 5e-3
    assert abs(m.com(q)[0] - m.com(m.q0)[0] - 2 * Lstep) < 0.02


@pytest.mark.gpu
def test_qp_timer_test_icub_squat(host_build, oracle_mod, tmp_path):
    """The iCub stack (etc/icub/tasks.yaml: posture in mid-stack, no actuation bounds, contact normal -z on z-down ankle frames,
    virtual frames of etc/icub/frames.yaml) through the facade on the iCub-like model: a 5 cm squat, closed loop, against the
    oracle loop."""
    from inria_wbc_amd import model as mdl, structure, trajs
    from oracle import rbd
    n_ticks = 400
    q_path = str(tmp_path / "q.bin")
    r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/icub/pos_tracker_model.yaml"),
                        os.path.join(ROOT, "configs/icub/squat.yaml"), "-", str(n_ticks), str(tmp_path / "tau.bin"), "0", q_path],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    m = mdl.icub_like()
    st = structure.icub_structure()
    tm = mdl.build_taskmap(m, st, mdl.icub_stack())
    q = np.fromfile(q_path, dtype=np.float64).reshape(4, m.nq)
    assert all(np.array_equal(q[0], q[i]) for i in range(1, 4))
    s = mdl.sample_states(m, tm, 1, 1, q_noise=0.0, v_noise=0.0, ref_noise=0.0)
    oq, ov, oref = s["q"], s["v"], s["ref"]
    blk = next(b for b in tm.blocks if b.kind == mdl.T_COM)
    pos, vel, acc = trajs.move_com_stream(m.com(m.q0), [[0.0, 0.0, -0.05]], "001", tm.dt, 1.0, loop=True, absolute=False)
    empty, w = np.zeros((1, 0)), st.default_weights[None]
    for k in range(n_ticks):
        oref[0, blk.ref:blk.ref + 9] = np.concatenate([pos[k], vel[k], acc[k]])
        rows = rbd.task_rows(m, tm, st, oq, ov, oref)
        oo = oracle_mod.tick_batch(st, dict(rows, tlb=empty, tub=empty, w=w))
        assert oo["status"][0] == 0
        nxt = oracle_mod.integrate(True, tm.dt, oq, ov, oo["x"][:, :st.nv])
        oq, ov = nxt["q_next"], nxt["v_next"]
    assert np.abs(q[0] - oq[0]).max() < 1e-8, np.abs(q[0] - oq[0]).max()
    assert abs(m.com(q[0])[2] - pos[n_ticks - 1][2]) < 5e-3 and m.com(m.q0)[2] - m.com(q[0])[2] > 0.005


@pytest.mark.gpu
def test_two_runs_are_bitwise_equal(host_build, tmp_path):
    """The reference's determinism bar (tests/test_determinism.cpp:44-57: two controllers fed the same inputs agree below 1e-8)
    -- here two separate processes running 300 closed-loop ticks of the walk on the spot (with a contact switch inside) must
    agree bit for bit, in the joint state and in the torques."""
    outs = []
    for run in range(2):
        q_path, tau_path = str(tmp_path / ("q%d.bin" % run)), str(tmp_path / ("tau%d.bin" % run))
        beh = tmp_path / "wos.yaml"
        beh.write_text("BEHAVIOR:\n  name: humanoid::walk-on-spot\n  traj_com_duration: 0.2\n  traj_foot_duration: 0.2\n  step_height: 0.03\n")
        r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/talos/pos_tracker_model.yaml"), str(beh), "-", "300", tau_path,
                            "0", q_path], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr
        outs.append((open(q_path, "rb").read(), open(tau_path, "rb").read()))
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_task_file_key_order_does_not_change_the_trajectory(host_build, tmp_path):
    """The other half of the reference's determinism test (tests/test_determinism.cpp:118-138): tasks.yaml re-emitted with its keys in a
    random order must give the same joint trajectory below 1e-8 (there over 50 000 ticks of two behaviours; here 300 closed-loop ticks
    of the squat through the facade, model -> rows kernel -> solve -> integration on the device, three shuffles)."""
    import random
    import shutil
    src = os.path.join(ROOT, "configs", "talos")
    text = open(os.path.join(src, "tasks.yaml")).read().splitlines()
    head = [ln for ln in text[:1] if ln.startswith("#")]
    blocks, cur = [], []
    for ln in text[len(head):]:
        if ln and not ln.startswith((" ", "#")) and cur:
            blocks.append(cur)
            cur = []
        cur.append(ln)
    if cur:
        blocks.append(cur)
    assert len(blocks) >= 8
    runs = []
    for k in range(4):
        d = tmp_path / ("talos%d" % k)
        shutil.copytree(src, d)
        order = list(blocks)
        if k:
            random.Random(100 + k).shuffle(order)
            assert [b[0] for b in order] != [b[0] for b in blocks]
        (d / "tasks.yaml").write_text("\n".join(head + [ln for b in order for ln in b]) + "\n")
        q_path, tau_path = str(tmp_path / ("q%d.bin" % k)), str(tmp_path / ("tau%d.bin" % k))
        r = subprocess.run([host_build["qp_timer_test"], str(d / "pos_tracker_model.yaml"), str(d / "squat.yaml"), "-", "300", tau_path, "0", q_path],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr
        runs.append(np.frombuffer(open(q_path, "rb").read(), dtype=np.float64))
    assert runs[0].size > 0 and np.isfinite(runs[0]).all() and np.abs(runs[0]).max() > 0.1
    for k in range(1, 4):
        assert runs[k].shape == runs[0].shape
        assert np.abs(runs[k] - runs[0]).max() < 1e-8, (k, float(np.abs(runs[k] - runs[0]).max()))


@pytest.mark.gpu
def test_a_bounded_solver_fails_with_the_references_text(host_build, oracle_mod, tmp_path):
    """controller.cpp:285-307: a QP that does not end OPTIMAL throws "Controller failed, can't solve problem. Status : N => ...".  With
    CONTROLLER.solver_max_iter (eiquadprog-fast's maxIter, 1000 when absent) below what a tick needs, the first instance whose QP is
    stopped must come back as status 3, "Max iter reached" -- and the same batch must run clean under the default bound."""
    from tools import dump_batch
    from inria_wbc_amd import structure, synth
    st = structure.talos_structure()
    B = 24
    inputs = synth.generate(st, B, synth.SEED_BASE["talos_squat"] + 5, task_noise=2.0)
    first_tick = 700
    rows = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    ref_in = {k: v.copy() for k, v in inputs.items()}
    ref_in["b1"][:, rows] += synth.squat_com_rhs(st, first_tick, 30.0)  # what move_com adds at that tick (as in the squat test above)
    ref = oracle_mod.tick_batch(st, ref_in, nthreads=4)
    assert (ref["status"] == 0).all() and ref["iters"].max() > 6
    first = int(np.argmax(ref["iters"] >= 4))  # eiquadprog stops when its counter REACHES maxIter
    path = str(tmp_path / "b.bin")
    dump_batch.dump(path, st, inputs)
    base = open(os.path.join(ROOT, "configs/talos/pos_tracker.yaml")).read()
    bounded = str(tmp_path / "pos_tracker_bounded.yaml")
    with open(bounded, "w") as f:
        f.write(base.rstrip("\n") + "\n  solver_max_iter: 4\n")
    for name in ("tasks.yaml",):
        with open(str(tmp_path / name), "w") as f:
            f.write(open(os.path.join(ROOT, "configs/talos", name)).read())
    squat = os.path.join(ROOT, "configs/talos/squat.yaml")
    tau_path = str(tmp_path / "tau.bin")
    r = subprocess.run([host_build["qp_timer_test"], bounded, squat, path, "1", tau_path, str(first_tick)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, r.stdout + r.stderr
    assert "Controller failed, can't solve problem. Status : 3 => Max iter reached" in r.stderr, r.stderr
    assert "[instance %d]" % first in r.stderr, (first, r.stderr)
    r = subprocess.run([host_build["qp_timer_test"], os.path.join(ROOT, "configs/talos/pos_tracker.yaml"), squat, path, "1", tau_path, str(first_tick)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_dump_tool_compiles_against_declaration_stubs():
    """tools/dump_reference_vectors.cpp -- the program that will PIN parity on a machine with tsid -- is written against the reference's
    real API and cannot be built here.  It is at least syntax- and type-checked against declaration-only stubs of the names it uses (tools/stubs/README.md:
    no bodies, no object file, pins nothing), so that the day such a machine appears the one command does not start with a compile error.  Contract: the
    calls of /root/reference/src/controllers/controller.cpp:244-251 and pos_tracker.cpp:83-106."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "tools", "stubs"),
                           os.path.join(root, "tools", "dump_reference_vectors.cpp")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    # the stubs stay declarations: nothing under tools/stubs defines a function body or is named in any build recipe
    for dirpath, _, files in os.walk(os.path.join(root, "tools", "stubs")):
        for f in files:
            if f == "README.md":
                continue
            text = open(os.path.join(dirpath, f)).read()
            assert "COMPILE-CHECK STUB" in text, f
    for recipe in ("inria_wbc_amd/build.py", "oracle/Makefile", "__graft_entry__.py"):
        assert "tools/stubs" not in open(os.path.join(root, recipe)).read(), recipe
