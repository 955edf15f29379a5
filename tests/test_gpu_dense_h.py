"""GPU parity for stacks with a `torque` task (src/controllers/tasks.cpp:227-271), a `cop` task (tasks.cpp:156-178) or a posture
`mask:` (tasks.cpp:205-214): the HIP path through the C ABI against the dense CPU oracle on the same seeded records.  A torque or a
cop task makes H one n x n matrix (wbcqp_layout.dense_h): the full LDS layout with assemble_eliminate_dense (csrc/wbcqp_device.hpp)."""
import dataclasses

import numpy as np
import pytest

from inria_wbc_amd import structure, synth
from tests.util import assert_parity, device_outputs, host_outputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle(built_lib):
    import torch  # noqa: F401
    from inria_wbc_amd import capi
    h = capi.Handle(device=0, dtype=capi.F64)
    yield h
    h.close()


def _with_weight(st, task, w):
    dw = st.default_weights.copy()
    dw[st.task_names.index(task)] = w
    return dataclasses.replace(st, default_weights=dw)


@pytest.mark.parametrize("name,task,weight,noise", [
    ("talos_torque", "torque", 1e-3, 0.5), ("talos_torque", "torque", 1e-2, 2.0), ("talos_torque", "torque", 1.0, 2.0),
    ("talos_torque", "torque", 10.0, 5.0),
    ("talos_cop", "cop", 1e-3, 0.5), ("talos_cop", "cop", 1.0, 2.0), ("talos_cop", "cop", 10.0, 5.0),
    ("talos_torque_cop", "torque", 1e-1, 2.0), ("icub_torque", "torque", 1e-3, 0.5), ("icub_torque", "torque", 10.0, 5.0)])
def test_parity_vs_oracle(handle, oracle_mod, name, task, weight, noise):
    st = _with_weight(structure.STRUCTURES[name](), task, weight)
    B = 32
    inputs = synth.generate(st, B, synth.SEED_BASE[name] + 100, task_noise=noise, torque_ref_noise=1.0)
    ref = oracle_mod.tick_batch(st, inputs, nthreads=4)
    assert (ref["status"] == 0).all()
    handle.set_structure(0, st)
    got = handle.solve_batch_host(0, inputs)
    info = assert_parity(st, got, ref, what="%s w=%g" % (name, weight))
    assert info["iters_equal"] >= 0.8, (got["iters"], ref["iters"])
    # the task is not a no-op: the plain stack's answer differs
    plain = structure.STRUCTURES[name.split("_")[0]]()
    pin = {k: v for k, v in inputs.items() if k != "Acop"}
    pin["b1"] = inputs["b1"][:, :plain.r1]
    pin["w"] = inputs["w"][:, :plain.n_tasks]
    pref = oracle_mod.tick_batch(plain, pin, nthreads=4)
    assert np.abs(pref["tau"] - ref["tau"]).max() > 1e-6


def test_masked_and_scaled_torque_task(handle, oracle_mod):
    """`mask:` and `scaling:` of the torque task (tasks.cpp:245-261): rows for the mask's ones only, each with its own weight-vector entry."""
    base = structure.talos_structure()
    rng = np.random.default_rng(2)
    mask = (rng.random(base.na) < 0.5).astype(int)
    scaling = rng.uniform(0.2, 3.0, base.na)
    st = structure.with_torque_task(base, 0.05, mask=mask, scaling=scaling)
    assert st.n_acteq == int(mask.sum()) and np.allclose(st.acteq_scale, scaling[mask != 0])
    inputs = synth.generate(st, 24, synth.SEED_BASE["talos"] + 400, task_noise=2.0, torque_ref_noise=2.0)
    ref = oracle_mod.tick_batch(st, inputs, nthreads=4)
    handle.set_structure(1, st)
    got = handle.solve_batch_host(1, inputs)
    assert_parity(st, got, ref, what="masked torque task")


def test_fixed_base_stack_with_a_torque_task(handle, oracle_mod):
    """Franka with a torque task: no contacts, n = 9 -- leaves the one-wavefront-per-QP kernel for the full layout."""
    from inria_wbc_amd import capi
    st = structure.with_torque_task(structure.franka_structure(), 0.5)
    L = capi.layout_of(st)
    assert L["dense_h"] == 1 and L["wave_per_qp"] == 0
    inputs = synth.generate(st, 40, synth.SEED_BASE["franka"] + 9, torque_ref_noise=1.0)
    ref = oracle_mod.tick_batch(st, inputs, nthreads=4)
    handle.set_structure(2, st)
    got = handle.solve_batch_host(2, inputs)
    assert_parity(st, got, ref, what="franka + torque")
    # unconstrained: x = -H^-1 g exactly
    for i in range(3):
        H, g, *_ = oracle_mod.assemble(st, inputs, i)
        assert np.allclose(got["x"][i], -np.linalg.solve(H, g), rtol=1e-9, atol=1e-9)


def test_posture_mask(handle, oracle_mod):
    base = structure.talos_structure()
    mask = np.ones(base.na, int); mask[[1, 7, 30, 43]] = 0
    st = structure.with_posture_mask(base, mask)
    inputs = synth.generate(st, 32, synth.SEED_BASE["talos"] + 600, task_noise=2.0)
    ref = oracle_mod.tick_batch(st, inputs, nthreads=4)
    handle.set_structure(3, st)
    got = handle.solve_batch_host(3, inputs)
    info = assert_parity(st, got, ref, what="posture mask")
    assert info["iters_equal"] >= 0.8
    from inria_wbc_amd import capi
    assert capi.layout_of(st)["waves_per_cu"] == 2  # still the compact layout


def test_device_pointers_ragged_launch_with_a_dense_h_group(handle, oracle_mod):
    """A ragged launch whose groups are a plain Talos stack and a Talos + torque + cop stack: the launch runs the full layout for both."""
    import torch
    from inria_wbc_amd import capi
    dev = torch.device("cuda", 0)
    sts = [structure.talos_structure(), structure.STRUCTURES["talos_torque_cop"]()]
    groups, refs, outs = [], [], []
    h = capi.Handle(0, capi.F64)
    for gi, st in enumerate(sts):
        B = 40 + 8 * gi
        inp = synth.generate(st, B, synth.SEED_BASE[st.name.split("+")[0]] + 50 + gi, task_noise=2.0)
        refs.append(oracle_mod.tick_batch(st, inp, nthreads=4))
        h.set_structure(gi, st)
        d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
        d_out = device_outputs(B, st, dev)
        groups.append((gi, B, d_in, d_out))
        outs.append(d_out)
    for _ in range(3):  # the second and third launch run in the order the first left
        h.solve_ragged(groups, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for st, ref, o in zip(sts, refs, outs):
        assert_parity(st, host_outputs(o, st), ref, what=st.name)
    h.close()
