"""The compact kernel exists once for any structure and once per shipped stack with every size and LDS offset as a literal
(csrc/wbcqp_types.hpp: kSpecDims; wbcqp_layout.specialised).  Same source, same arithmetic order: a launch through a stack's own
instantiation must give the generic kernel's bits (WBCQP_FLAG_GENERIC_KERNEL), for both boundary dtypes, on the hardware's dispatcher and
through the queue; ragged launches and structures that are no shipped stack run the generic kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(st, inputs, flags, dtype, launches=2):
    import torch
    from inria_wbc_amd import capi
    B = inputs["h"].shape[0]
    dev = torch.device("cuda", 0)
    tdt = torch.float64 if dtype == capi.F64 else torch.float32
    ndt = np.float64 if dtype == capi.F64 else np.float32
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v.astype(ndt))).to(dev) for k, v in inputs.items() if v.size}
    h = capi.Handle(0, dtype, flags=flags)
    h.set_structure(0, st)
    out = None
    for _ in range(launches):  # the second launch runs in the order the first one left
        out = dict(x=torch.full((B, st.n), float("nan"), dtype=tdt, device=dev), tau=torch.full((B, st.na), float("nan"), dtype=tdt, device=dev),
                   status=torch.full((B,), -99, dtype=torch.int32, device=dev), iters=torch.full((B,), -1, dtype=torch.int32, device=dev),
                   n_active=torch.zeros(B, dtype=torch.int32, device=dev), active_mask=torch.zeros(B, 8, dtype=torch.int32, device=dev),
                   objective=torch.zeros(B, dtype=tdt, device=dev))
        h.solve_batch(0, B, d_in, out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    h.close()
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("name,spec", [("talos", 1), ("icub", 2), ("talos_single_support", 3)])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_a_shipped_stacks_own_instantiation_gives_the_generic_kernels_bits(name, spec, dtype):
    from inria_wbc_amd import capi, structure, synth
    st = structure.STRUCTURES[name]()
    assert capi.layout_of(st)["specialised"] == spec
    B = 600
    inputs = synth.generate(st, B, synth.SEED_BASE[name] + 4242, task_noise=2.0)
    dt = capi.F64 if dtype == "f64" else capi.F32
    G, HW = capi.FLAG_GENERIC_KERNEL, capi.FLAG_HW_DISPATCH
    ref = _run(st, inputs, G, dt)
    assert (ref["status"] == 0).all() and ref["iters"].max() >= 10
    for flags in (0, HW, capi.FLAG_INDEX_ORDER):
        got = _run(st, inputs, flags, dt)
        for k in ("x", "tau", "status", "iters", "n_active", "active_mask", "objective"):
            assert np.array_equal(got[k], ref[k], equal_nan=True), (name, dtype, flags, k)
    gen_hw = _run(st, inputs, G | HW, dt)
    for k in ("x", "tau", "status", "iters"):
        assert np.array_equal(gen_hw[k], ref[k], equal_nan=True), k


def test_other_structures_and_ragged_launches_take_the_generic_kernel(oracle_mod):
    """A posture mask changes the layout (fewer level-1 rows): no instantiation matches, the generic kernel runs, parity holds; a ragged launch
    of two shipped stacks runs the generic kernel for both groups."""
    import torch
    from inria_wbc_amd import capi, structure, synth
    from tests.util import assert_parity
    base = structure.talos_structure()
    mask = np.ones(base.na, int); mask[[2, 9]] = 0
    st = structure.with_posture_mask(base, mask)
    L = capi.layout_of(st)
    assert L["specialised"] == 0 and L["waves_per_cu"] == 2
    inputs = synth.generate(st, 64, synth.SEED_BASE["talos"] + 7, task_noise=2.0)
    got = _run(st, inputs, 0, capi.F64, launches=1)
    assert_parity(st, got, oracle_mod.tick_batch(st, inputs, nthreads=4), what="posture mask, generic kernel")
    for name in ("franka", "tiago", "three_contact", "talos_torque"):
        assert capi.layout_of(structure.STRUCTURES[name]())["specialised"] == 0
    # ragged: Talos + iCub in one launch against each alone (bitwise: the launch shape never changes a result)
    dev = torch.device("cuda", 0)
    h = capi.Handle(0, capi.F64)
    groups, alone = [], []
    for slot, name in enumerate(("talos", "icub")):
        s2 = structure.STRUCTURES[name]()
        inp = synth.generate(s2, 200, synth.SEED_BASE[name] + 99, task_noise=2.0)
        alone.append(_run(s2, inp, 0, capi.F64, launches=1))
        h.set_structure(slot, s2)
        d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
        d_out = dict(x=torch.zeros(200, s2.n, dtype=torch.float64, device=dev), tau=torch.zeros(200, s2.na, dtype=torch.float64, device=dev),
                     status=torch.full((200,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(200, dtype=torch.int32, device=dev))
        groups.append((slot, 200, d_in, d_out))
    h.solve_ragged(groups, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for g, a in zip(groups, alone):
        for k in ("x", "tau", "status", "iters"):
            assert np.array_equal(g[3][k].cpu().numpy(), a[k]), k
    h.close()
