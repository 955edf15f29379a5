"""One wavefront per QP for the small structures (csrc/wbcqp_small.hpp: Franka etc/franka/tasks.yaml:1-10, Tiago etc/tiago/tasks.yaml)
against the C oracle and against the four-wave kernels (WBCQP_FLAG_WORKGROUP_PER_QP) on the same inputs: batches that do not
fill the last workgroup, bounds that bind (Tiago at high task noise: drops and re-picks), both boundary dtypes, and a ragged
launch where the small groups go out as a launch of their own beside the humanoids."""
import numpy as np
import pytest

from tests.util import TOL_F64, assert_parity, device_outputs, host_outputs

pytestmark = pytest.mark.gpu


def _solve(st, inputs, flags=0, dtype=None):
    import torch
    from inria_wbc_amd import capi
    dtype = capi.F64 if dtype is None else dtype
    tdt = torch.float64 if dtype == capi.F64 else torch.float32
    ndt = np.float64 if dtype == capi.F64 else np.float32
    B = next(iter(inputs.values())).shape[0]
    dev = torch.device("cuda", 0)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v.astype(ndt))).to(dev) for k, v in inputs.items() if v.size}
    h = capi.Handle(0, dtype, flags=flags)
    h.set_structure(0, st)
    d_out = device_outputs(B, st, dev, dtype=tdt)
    d_out["x"].fill_(float("nan")); d_out["tau"].fill_(float("nan")); d_out["iters"].fill_(-1); d_out["active_mask"].fill_(-1)
    h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h.close()
    return host_outputs(d_out, st)


@pytest.mark.parametrize("robot,batch,noise", [("franka", 1, 0.5), ("franka", 7, 0.5), ("franka", 1025, 2.0), ("tiago", 5, 0.5), ("tiago", 258, 2.0),
                                               ("tiago", 1024, 8.0), ("tiago", 333, 30.0)])
def test_wave_per_qp_matches_oracle_and_the_four_wave_kernel(oracle_mod, robot, batch, noise):
    from inria_wbc_amd import capi, structure, synth
    st = structure.STRUCTURES[robot]()
    inputs = synth.generate(st, batch, synth.SEED_BASE[robot] + 777, task_noise=noise, p_bnd=0.3)
    got = _solve(st, inputs)
    wg = _solve(st, inputs, flags=capi.FLAG_WORKGROUP_PER_QP)
    ref = oracle_mod.tick_batch(st, inputs)
    assert np.array_equal(got["status"], ref["status"]) and np.array_equal(wg["status"], ref["status"])
    ok = ref["status"] == 0
    scale = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
    assert (np.abs(got["x"] - ref["x"]).max(axis=1)[ok] <= TOL_F64 * scale[ok]).all()
    assert (np.abs(got["tau"] - ref["tau"]).max(axis=1)[ok] <= TOL_F64 * np.maximum(1.0, np.abs(ref["tau"]).max(axis=1))[ok]).all()
    assert (np.abs(got["x"] - wg["x"]).max(axis=1)[ok] <= 1e-9 * scale[ok]).all()
    assert (got["iters"] == ref["iters"]).mean() >= 0.9, (got["iters"][:20], ref["iters"][:20])
    # the active set (bit r = one-sided CI row r), n_active and the objective of both kernels against eiquadprog's A, iq, f
    assert_parity(st, got, ref, what="one wavefront per QP: %s B %d noise %g" % (robot, batch, noise))
    assert_parity(st, wg, ref, what="four waves per QP: %s B %d noise %g" % (robot, batch, noise))
    if robot == "tiago" and noise >= 8.0:
        assert ref["iters"].max() >= 4  # bounds bind: the active-set loop and its drop path are exercised


def test_wave_per_qp_f32_boundary(oracle_mod):
    from inria_wbc_amd import capi, structure, synth
    st = structure.STRUCTURES["tiago"]()
    inputs = synth.generate(st, 130, synth.SEED_BASE["tiago"] + 5, task_noise=4.0, p_bnd=0.3)
    inputs = {k: v.astype(np.float32).astype(np.float64) for k, v in inputs.items()}
    got = _solve(st, inputs, dtype=capi.F32)
    ref = oracle_mod.tick_batch(st, inputs)
    assert np.array_equal(got["status"], ref["status"])
    assert np.abs(got["x"] - ref["x"]).max() <= 1e-5 * max(1.0, np.abs(ref["x"]).max())


def test_ragged_launch_sends_small_groups_to_their_own_kernel(oracle_mod):
    import torch
    from inria_wbc_amd import capi, structure, synth
    dev = torch.device("cuda", 0)
    names = ["franka", "talos", "tiago", "icub"]
    counts = [37, 9, 50, 11]
    h = capi.Handle(0, capi.F64)
    groups, refs = [], []
    for slot, (name, cnt) in enumerate(zip(names, counts)):
        st = structure.STRUCTURES[name]()
        inp = synth.generate(st, cnt, synth.SEED_BASE["ragged"] + 31 * slot, task_noise=2.0, p_bnd=0.3)
        d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
        d_out = device_outputs(cnt, st, dev)
        d_out["x"].fill_(float("nan")); d_out["tau"].fill_(float("nan")); d_out["iters"].fill_(-1)
        h.set_structure(slot, st)
        groups.append((slot, cnt, d_in, d_out))
        refs.append((st, oracle_mod.tick_batch(st, inp)))
    h.solve_ragged(groups, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for (slot, cnt, _, d_out), (st, ref) in zip(groups, refs):
        assert np.array_equal(d_out["status"].cpu().numpy(), ref["status"]), st.name
        x = d_out["x"].cpu().numpy()
        assert np.abs(x[:, :st.nv] - ref["x"][:, :st.nv]).max() <= TOL_F64 * max(1.0, np.abs(ref["x"]).max()), st.name
        assert_parity(st, host_outputs(d_out, st), ref, what="ragged, small groups on their own kernel: " + st.name)
    h.close()


@pytest.mark.parametrize("nv", [11, 13, 15])
def test_odd_sizes_on_the_four_wave_compact_kernel(oracle_mod, nv):
    """The compact loop walks the rows of J in 16-byte pairs from an even column on, with a zero pad pair behind column n: an odd n puts the pad
    INSIDE the last pair of a row (round 5).  Tiago-like stacks of odd size through the four-wave kernel (WBCQP_FLAG_WORKGROUP_PER_QP), bounds
    binding (adds, drops), against the oracle and against the one-wavefront kernel."""
    from inria_wbc_amd import capi, structure, synth
    st = structure.tiago_structure(nv=nv)
    assert st.n % 2 == 1
    inputs = synth.generate(st, 200, synth.SEED_BASE["tiago"] + 100 + nv, task_noise=30.0, p_bnd=0.3)
    ref = oracle_mod.tick_batch(st, inputs)
    wg = _solve(st, inputs, flags=capi.FLAG_WORKGROUP_PER_QP)
    one = _solve(st, inputs)
    assert ref["iters"].max() >= 4
    for got in (wg, one):
        assert np.array_equal(got["status"], ref["status"])
        ok = ref["status"] == 0
        scale = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
        assert (np.abs(got["x"] - ref["x"]).max(axis=1)[ok] <= TOL_F64 * scale[ok]).all()
        assert (got["iters"] == ref["iters"]).mean() >= 0.9
        assert_parity(st, got, ref, what="odd n = %d" % st.n)
