"""Closed loop on the device: state -> rows (wbcqp_problem_data) -> QP (wbcqp_solve_batch) -> state (wbcqp_integrate), tick
after tick, for a batch of Talos-like robots following the squat reference of etc/talos/squat.yaml (move_com.cpp:8-61).
Checked against the same loop made of the three oracles, and against the physics it is meant to produce (the CoM follows
its reference, the feet stay where the contacts hold them)."""
import numpy as np
import pytest

from inria_wbc_amd import capi, structure, trajs
from inria_wbc_amd import model as mdl

pytestmark = pytest.mark.gpu

TICKS = 120
TOL_STATE = 1e-7  # on q after TICKS closed-loop ticks (per-tick agreement is ~1e-12; the loop amplifies it)


def test_squat_rollout_matches_oracle_loop_and_tracks_the_com():
    import torch
    from oracle import oracle as orc
    from oracle import rbd
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    B, dt = 6, tm.dt
    s = mdl.sample_states(m, tm, B, 77_000, q_noise=0.002, v_noise=0.01, ref_noise=0.0)
    com_blk = next(b for b in tm.blocks if b.kind == mdl.T_COM)
    com0 = m.com(m.q0)
    pos, vel, acc = trajs.move_com_stream(com0, [[0.0, 0.0, -0.2]], "001", dt, 2.0, loop=True, absolute=False)
    dev = torch.device("cuda", 0)
    h = capi.Handle(0, capi.F64)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    L = st.field_lengths()
    q, v, ref = (torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref"))
    qn, vn = torch.zeros_like(q), torch.zeros_like(v)
    rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
    rows["tlb"] = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
    rows["tub"] = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
    rows["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
    out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
               status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    stream = torch.cuda.current_stream().cuda_stream
    # the oracle loop on the host
    oq, ov, oref = s["q"].copy(), s["v"].copy(), s["ref"].copy()
    tl, tu, w = np.tile(-m.tau_max, (B, 1)), np.tile(m.tau_max, (B, 1)), np.tile(st.default_weights, (B, 1))
    lf = m.frame("leg_left_6_joint")
    foot0 = m.frame_placements(s["q"][0])[1][lf]
    for k in range(TICKS):
        r9 = np.concatenate([pos[k], vel[k], acc[k]])
        ref[:, com_blk.ref:com_blk.ref + 9] = torch.from_numpy(r9).to(dev)
        h.problem_data(0, B, dict(q=q, v=v, ref=ref), rows, stream=stream)
        h.solve_batch(0, B, rows, out, stream=stream)
        h.integrate(B, st.nv, True, dt, q, v, out["x"], st.n, out["status"], qn, vn, None, stream=stream)
        q, qn = qn, q
        v, vn = vn, v
        oref[:, com_blk.ref:com_blk.ref + 9] = r9
        orow = rbd.task_rows(m, tm, st, oq, ov, oref, n_threads=4)
        oo = orc.tick_batch(st, dict(orow, tlb=tl, tub=tu, w=w), nthreads=4)
        assert (oo["status"] == 0).all(), (k, oo["status"])
        nxt = orc.integrate(True, dt, oq, ov, oo["x"][:, :st.nv])
        oq, ov = nxt["q_next"], nxt["v_next"]
    torch.cuda.synchronize()
    assert (out["status"] == 0).all().item()
    gq, gv = q.cpu().numpy(), v.cpu().numpy()
    assert np.abs(gq - oq).max() < TOL_STATE, np.abs(gq - oq).max()
    assert np.abs(gv - ov).max() < 1e-5, np.abs(gv - ov).max()
    # physics: the CoM went down with its reference (critically damped tracking lags by ~2/sqrt(Kp) s), the feet did not move
    for i in range(B):
        c = m.com(gq[i])
        assert abs(c[2] - pos[TICKS - 1][2]) < 5e-3, (c, pos[TICKS - 1])
        assert abs(c[2] - com0[2]) > 1e-4
        assert np.abs(m.frame_placements(gq[i])[1][lf] - m.frame_placements(s["q"][i])[1][lf]).max() < 2e-3
    assert np.isfinite(foot0).all()
    h.close()


def _tick_buffers(m, st, tm, B, seed, dev, torch):
    s = mdl.sample_states(m, tm, B, seed, q_noise=0.005, v_noise=0.02, ref_noise=0.005)
    L = st.field_lengths()
    state = {k: torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref")}
    rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
    rows["tlb"] = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
    rows["tub"] = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
    rows["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
    out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
               status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    return state, rows, out, torch.zeros_like(state["q"]), torch.zeros_like(state["v"])


@pytest.mark.parametrize("B,robot", [(1, "talos"), (37, "talos"), (37, "icub"), (1000, "icub")])
def test_tick_and_graph_replay_are_the_three_calls(B, robot):
    """wbcqp_tick = problem_data + solve_batch + integrate; the captured graph replays it bit for bit, tick after tick, with
    the state fed back between launches.  (iCub: the solve of the captured tick is the three-per-CU kernel, whose lanes keep part of their
    registers in scratch -- a captured launch must bring that with it.)"""
    import torch
    m = mdl.talos_like() if robot == "talos" else mdl.icub_like()
    st = structure.talos_structure() if robot == "talos" else structure.icub_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack() if robot == "talos" else mdl.icub_stack())
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    results = []
    for mode in ("calls", "tick", "graph"):
        h = capi.Handle(0, capi.F64)
        h.set_structure(0, st)
        h.set_model(0, m, tm)
        state, rows, out, qn, vn = _tick_buffers(m, st, tm, B, 88_000, dev, torch)
        q_init, v_init = state["q"].clone(), state["v"].clone()
        g = h.tick_graph(0, B, state, rows, out, qn, vn, tm.dt) if mode == "graph" else None
        if g is not None:  # creation ran warm-up ticks on these buffers: start every mode from the same state
            state["q"].copy_(q_init); state["v"].copy_(v_init)
        for k in range(12):
            if mode == "calls":
                h.problem_data(0, B, state, rows, stream=stream)
                h.solve_batch(0, B, rows, out, stream=stream)
                h.integrate(B, st.nv, True, tm.dt, state["q"], state["v"], out["x"], st.n, out["status"], qn, vn, None, stream=stream)
            elif mode == "tick":
                h.tick(0, B, state, rows, out, qn, vn, tm.dt, stream=stream)
            else:
                h.tick_graph_launch(g, stream=stream)
            state["q"].copy_(qn)
            state["v"].copy_(vn)
        torch.cuda.synchronize()
        results.append((state["q"].cpu().numpy().copy(), out["tau"].cpu().numpy().copy(), out["status"].cpu().numpy().copy()))
        if g is not None:
            h.tick_graph_destroy(g)
        h.close()
    for r in results[1:]:
        assert np.array_equal(r[0], results[0][0]) and np.array_equal(r[1], results[0][1]) and np.array_equal(r[2], results[0][2])
    assert (results[0][2] == 0).all()


def test_graph_replay_survives_other_launches_on_the_same_handle():
    """A captured tick owns its launch-order buffer and queue counter: launches of other shapes on the same handle (which
    resize and overwrite the handle's own order buffer) between replays change nothing -- bitwise equal to an index-order run."""
    import torch
    from inria_wbc_amd import synth
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    B = 64
    results = []
    for mode in ("plain", "graph"):
        h = capi.Handle(0, capi.F64, flags=capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH if mode == "plain" else capi.FLAG_QUEUE)
        h.set_structure(0, st)
        h.set_model(0, m, tm)
        state, rows, out, qn, vn = _tick_buffers(m, st, tm, B, 88_500, dev, torch)
        q_init, v_init = state["q"].clone(), state["v"].clone()
        g = h.tick_graph(0, B, state, rows, out, qn, vn, tm.dt) if mode == "graph" else None
        state["q"].copy_(q_init); state["v"].copy_(v_init)
        others = []
        for ob in (256, 32, 1500):
            oi = synth.generate(st, min(ob, 256), synth.SEED_BASE["talos"] + ob)
            reps = (ob + 255) // 256
            d_in = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (reps, 1))[:ob])).to(dev) for k, v in oi.items() if v.size}
            d_out = dict(x=torch.zeros(ob, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(ob, st.na, dtype=torch.float64, device=dev),
                         status=torch.zeros(ob, dtype=torch.int32, device=dev), iters=torch.zeros(ob, dtype=torch.int32, device=dev))
            others.append((ob, d_in, d_out))
        for k in range(9):
            if mode == "graph":
                ob, d_in, d_out = others[k % 3]
                for _ in range(2):  # the second launch of a shape uses (and the first one resizes) the handle's order buffer
                    h.solve_batch(0, ob, d_in, d_out, stream=stream)
                h.tick_graph_launch(g, stream=stream)
            else:
                h.tick(0, B, state, rows, out, qn, vn, tm.dt, stream=stream)
            state["q"].copy_(qn)
            state["v"].copy_(vn)
        torch.cuda.synchronize()
        results.append((state["q"].cpu().numpy().copy(), out["tau"].cpu().numpy().copy(), out["status"].cpu().numpy().copy(),
                        out["iters"].cpu().numpy().copy()))
        if g is not None:
            h.tick_graph_destroy(g)
        h.close()
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)
    assert (results[0][2] == 0).all()


def test_tick_host_equals_the_device_tick():
    """wbcqp_tick_host stages the state up and the solution down around the same three launches."""
    import torch
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    B = 9
    dev = torch.device("cuda", 0)
    h = capi.Handle(0, capi.F64)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    state, rows, out, qn, vn = _tick_buffers(m, st, tm, B, 91_000, dev, torch)
    h.tick(0, B, state, rows, out, qn, vn, tm.dt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = h.tick_host(0, state["q"].cpu().numpy(), state["v"].cpu().numpy(), state["ref"].cpu().numpy(), rows["tlb"].cpu().numpy(),
                      rows["tub"].cpu().numpy(), rows["w"].cpu().numpy(), tm.dt, want_rows=True)
    assert np.array_equal(got["x"], out["x"].cpu().numpy()) and np.array_equal(got["tau"], out["tau"].cpu().numpy())
    assert np.array_equal(got["status"], out["status"].cpu().numpy()) and np.array_equal(got["q_next"], qn.cpu().numpy())
    assert np.array_equal(got["v_next"], vn.cpu().numpy())
    for k in capi.ROW_FIELDS:
        assert np.array_equal(got["rows"][k], rows[k].cpu().numpy()), k
    assert np.abs(got["q_solver"][:, 6:] - got["q_next"][:, 7:]).max() == 0.0
    h.close()


def test_tick_f32_boundary_stays_close_to_f64():
    """BASELINE config 3's boundary (f32 arrays, f64 arithmetic inside) through the whole tick on the iCub-like robot: the f32
    record between the rows kernel and the solve costs precision, not correctness (1e-3 relative on ddq / tau)."""
    import torch
    m = mdl.icub_like()
    st = structure.icub_structure()
    tm = mdl.build_taskmap(m, st, mdl.icub_stack())
    B = 64
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    s = mdl.sample_states(m, tm, B, 93_000, q_noise=0.005, v_noise=0.02, ref_noise=0.005)
    L = st.field_lengths()
    res = {}
    for dtype, tdt in ((capi.F64, torch.float64), (capi.F32, torch.float32)):
        h = capi.Handle(0, dtype)
        h.set_structure(0, st)
        h.set_model(0, m, tm)
        state = {k: torch.from_numpy(s[k]).to(dev).to(tdt) for k in ("q", "v", "ref")}
        rows = {k: torch.zeros(B, L[k], dtype=tdt, device=dev) for k in capi.ROW_FIELDS}
        rows["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev).to(tdt)
        out = dict(x=torch.zeros(B, st.n, dtype=tdt, device=dev), tau=torch.zeros(B, st.na, dtype=tdt, device=dev),
                   status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
        qn, vn = torch.zeros_like(state["q"]), torch.zeros_like(state["v"])
        h.tick(0, B, state, rows, out, qn, vn, tm.dt, stream=stream)
        torch.cuda.synchronize()
        res[dtype] = {k: v.double().cpu().numpy() for k, v in out.items()} | {"q": qn.double().cpu().numpy()}
        h.close()
    a, b = res[capi.F64], res[capi.F32]
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    nv = m.nv
    scale_dv, scale_tau = max(1.0, np.abs(a["x"][:, :nv]).max()), max(1.0, np.abs(a["tau"]).max())
    assert np.abs(a["x"][:, :nv] - b["x"][:, :nv]).max() / scale_dv < 1e-3
    assert np.abs(a["tau"] - b["tau"]).max() / scale_tau < 1e-3
    assert np.abs(a["q"] - b["q"]).max() < 1e-5


def test_tick_keeps_the_state_of_an_instance_whose_qp_fails():
    """One robot of the batch gets torque limits that contradict each other (lower above upper): its QP is infeasible, the tick
    reports that status for it
    (the reference throws there, controller.cpp:284-307) and leaves ITS state where it was; the others go on."""
    import torch
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    B, bad = 8, 5
    dev = torch.device("cuda", 0)
    h = capi.Handle(0, capi.F64)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    state, rows, out, qn, vn = _tick_buffers(m, st, tm, B, 95_000, dev, torch)
    rows["tlb"][bad] = 1.0
    rows["tub"][bad] = -1.0
    h.tick(0, B, state, rows, out, qn, vn, tm.dt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    status = out["status"].cpu().numpy()
    assert status[bad] != 0 and (np.delete(status, bad) == 0).all(), status
    assert torch.equal(qn[bad], state["q"][bad]) and torch.equal(vn[bad], state["v"][bad])
    others = [i for i in range(B) if i != bad]
    assert not torch.equal(qn[others], state["q"][others])
    h.close()


@pytest.mark.parametrize("B,K,robot", [(5, 1, "talos"), (7, 6, "talos"), (600, 4, "talos"), (900, 4, "icub")])
def test_rollout_equals_k_ticks_bit_for_bit(B, K, robot):
    """wbcqp_rollout (K ticks of every instance in ONE launch, persistent workgroups, the record in a per-workgroup slot) against K
    calls of wbcqp_tick with the state fed back: the same device functions run the three phases, so every output is the same bits --
    the final state, q_solver, the last tick's x / tau / status / iters, the momentum of the last tick's state, and the iteration
    total.  600 instances: more than the 512 resident workgroups, so some workgroups run two instances one after the other (iCub, 900:
    more than its 768 -- three per CU)."""
    import torch
    m = mdl.talos_like() if robot == "talos" else mdl.icub_like()
    st = structure.talos_structure() if robot == "talos" else structure.icub_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack() if robot == "talos" else mdl.icub_stack())
    dev = torch.device("cuda", 0)
    s = mdl.sample_states(m, tm, B, 91_000, q_noise=0.01, v_noise=0.05, ref_noise=0.01)
    com_blk = next(b for b in tm.blocks if b.kind == mdl.T_COM)
    pos, vel, acc = trajs.move_com_stream(m.com(m.q0), [[0.0, 0.0, -0.2]], "001", tm.dt, 2.0, loop=True, absolute=False)
    refs = np.repeat(s["ref"][None], K, axis=0).copy()  # [K, B, nref]: the CoM reference advances, instance i is 37 i ticks ahead
    for t in range(K):
        for i in range(B):
            k = (t + 37 * i) % len(pos)
            refs[t, i, com_blk.ref:com_blk.ref + 9] = np.concatenate([pos[k], vel[k], acc[k]])
    L = st.field_lengths()
    tlb = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
    tub = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
    w = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def outs():
        return dict(x=torch.full((B, st.n), float("nan"), dtype=torch.float64, device=dev), tau=torch.full((B, st.na), float("nan"), dtype=torch.float64, device=dev),
                    status=torch.full((B,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))

    h = capi.Handle(0, capi.F64)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    # ---- K ticks, state fed back
    q, v = torch.from_numpy(s["q"]).to(dev), torch.from_numpy(s["v"]).to(dev)
    qn, vn, qs = torch.zeros_like(q), torch.zeros_like(v), torch.zeros_like(v)
    rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
    rows.update(tlb=tlb, tub=tub, w=w)
    o1 = outs()
    mom1 = torch.zeros(B, 6, dtype=torch.float64, device=dev)
    it_sum = np.zeros(B, np.int64)
    for t in range(K):
        state = dict(q=q, v=v, ref=torch.from_numpy(refs[t]).to(dev), momentum=mom1)
        h.tick(0, B, state, rows, o1, qn, vn, tm.dt, q_solver=qs, stream=stream)
        torch.cuda.synchronize()
        it_sum += o1["iters"].cpu().numpy()
        q, qn = qn, q
        v, vn = vn, v
    # ---- one rollout
    o2 = outs()
    q2, v2, qs2 = torch.zeros_like(q), torch.zeros_like(v), torch.zeros_like(v)
    mom2 = torch.zeros(B, 6, dtype=torch.float64, device=dev)
    isum = torch.zeros(B, dtype=torch.int32, device=dev)
    nok = torch.zeros(B, dtype=torch.int32, device=dev)
    st0 = dict(q=torch.from_numpy(s["q"]).to(dev), v=torch.from_numpy(s["v"]).to(dev), ref=torch.from_numpy(np.ascontiguousarray(refs)).to(dev), momentum=mom2)
    h.rollout(0, B, K, st0, dict(tlb=tlb, tub=tub, w=w), o2, q2, v2, tm.dt, q_solver=qs2, iters_sum=isum, ticks_ok=nok, stream=stream)
    torch.cuda.synchronize()
    assert (o1["status"] == 0).all().item() and (nok == K).all().item()
    for name, a, b in (("q", q, q2), ("v", v, v2), ("q_solver", qs, qs2), ("x", o1["x"], o2["x"]), ("tau", o1["tau"], o2["tau"]),
                       ("status", o1["status"], o2["status"]), ("iters", o1["iters"], o2["iters"]), ("momentum", mom1, mom2)):
        assert torch.equal(a, b), name
    assert np.array_equal(isum.cpu().numpy(), it_sum)
    assert np.array_equal(st0["q"].cpu().numpy(), s["q"])  # the caller's state arrays are inputs only
    h.close()
