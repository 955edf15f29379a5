"""The launch order (queue, longest-first, bin-packed) never changes a result, and the device packer agrees with its host
model (inria_wbc_amd/launch_order.py) class by class.  Nothing here has a counterpart in the reference: the reference solves
one QP per process; how a batch is dealt out to the CUs is this library's own business, so the checks are invariance,
permutation validity and agreement with the model."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(st, inputs, flags, launches, want_order=False):
    import torch
    from inria_wbc_amd import capi
    B = next(iter(inputs.values())).shape[0]
    dev = torch.device("cuda", 0)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    h = capi.Handle(0, capi.F64, flags=flags)
    h.set_structure(0, st)
    outs, orders = [], []
    for _ in range(launches):
        d_out = dict(x=torch.full((B, st.n), float("nan"), dtype=torch.float64, device=dev),
                     tau=torch.full((B, st.na), float("nan"), dtype=torch.float64, device=dev),
                     status=torch.full((B,), -99, dtype=torch.int32, device=dev),
                     iters=torch.full((B,), -1, dtype=torch.int32, device=dev))
        h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append({k: v.cpu().numpy() for k, v in d_out.items()})
        if want_order:
            orders.append(h.launch_order())
    h.close()
    return (outs, orders) if want_order else outs


@pytest.mark.parametrize("batch", [1, 17, 256, 300, 512, 1024, 1040, 2048, 2100])
def test_every_dispatch_gives_the_same_bits(batch):
    """default, queue + packed order, queue + longest-first, queue + index order, hardware dispatch: bitwise the same."""
    from inria_wbc_amd import capi, structure, synth
    st = structure.talos_structure()
    inputs = synth.generate(st, batch, synth.SEED_BASE["talos"] + 31 * batch)
    plain = _run(st, inputs, capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH, 1)[0]
    assert (plain["status"] != -99).all()
    Q = capi.FLAG_QUEUE
    for flags in (0, Q, Q | capi.FLAG_NO_PACKING, Q | capi.FLAG_INDEX_ORDER, capi.FLAG_HW_DISPATCH):
        for o in _run(st, inputs, flags, 3):
            for k in ("x", "tau", "status", "iters"):
                assert np.array_equal(o[k], plain[k], equal_nan=True), (flags, k)


@pytest.mark.parametrize("batch,noise", [(1024, 1.0), (1024, 2.0), (512, 1.0), (2048, 1.0), (4096, 1.0), (1536, 0.3), (544, 1.0)])
def test_packed_order_matches_the_host_model(batch, noise):
    from inria_wbc_amd import capi, launch_order, structure, synth
    import torch
    st = structure.talos_structure()
    inputs = synth.generate(st, batch, synth.SEED_BASE["talos"] + 5, task_noise=noise)
    outs, orders = _run(st, inputs, capi.FLAG_QUEUE, 2, want_order=True)
    iters = outs[0]["iters"]
    order, packed = orders[0]
    # resident workgroups: QPs per CU (compact layout: two Talos QPs) x CUs
    resident = capi.layout_of(st)["waves_per_cu"] * torch.cuda.get_device_properties(0).multi_processor_count
    assert packed == launch_order.packs(batch, resident)
    assert sorted(order.tolist()) == list(range(batch))
    cls = np.clip(iters, 0, launch_order.MAX_CLASS)
    if not packed:
        assert np.all(np.diff(cls[order]) <= 0)
        return
    lpt = np.argsort(-cls, kind="stable")
    model = launch_order.pack_order(lpt, iters, resident)
    assert np.array_equal(cls[order], cls[model])
    # and it is worth it on the model's own terms: list scheduling of the predicted costs
    cost = launch_order.SETUP_ITERS + cls.astype(float)
    assert launch_order.makespan(cost[order], resident) <= launch_order.makespan(cost[lpt], resident) * 1.03


def test_small_structures_share_a_cu():
    """Franka QPs are small: several workgroups are resident per CU; forced through the queue, its grid follows the occupancy."""
    from inria_wbc_amd import capi, structure, synth
    st = structure.franka_structure()
    inputs = synth.generate(st, 5000, synth.SEED_BASE["franka"] + 3)
    plain = _run(st, inputs, capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH, 1)[0]
    for flags in (0, capi.FLAG_QUEUE, capi.FLAG_QUEUE | capi.FLAG_NO_PACKING):  # default there: hardware dispatch
        for o in _run(st, inputs, flags, 3):
            for k in ("x", "tau", "status", "iters"):
                assert np.array_equal(o[k], plain[k], equal_nan=True), (flags, k)


@pytest.mark.parametrize("robot,batch", [("talos", 700), ("icub", 900), ("talos_single_support", 300), ("tiago", 1000)])
def test_compact_and_full_lds_layouts_agree(robot, batch):
    """The compact LDS layout (two QPs per CU, wbcqp_compact.hpp) and the full one (WBCQP_FLAG_FULL_LDS) are the same
    algorithm with different summation orders in a few places: same status and iteration counts, x / tau to rounding."""
    from inria_wbc_amd import capi, structure, synth
    st = structure.STRUCTURES[robot]()
    inputs = synth.generate(st, batch, synth.SEED_BASE.get(robot, 77) + 11)
    a = _run(st, inputs, capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH, 1)[0]
    b = _run(st, inputs, capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH | capi.FLAG_FULL_LDS, 1)[0]
    assert np.array_equal(a["status"], b["status"])
    assert (a["iters"] == b["iters"]).mean() >= 0.99
    ok = a["status"] == 0
    scale = np.maximum(1.0, np.abs(b["x"]).max(axis=1, keepdims=True))
    assert (np.abs(a["x"] - b["x"]) / scale)[ok].max() < 1e-9
    if st.na:
        assert np.abs(a["tau"] - b["tau"])[ok].max() < 1e-6 * max(1.0, np.abs(b["tau"][ok]).max())


def test_a_launch_with_one_ineligible_group_runs_every_group_on_the_full_layout():
    """A ragged launch whose groups are all eligible runs the compact kernel; one group that is not (three contacts: 24
    equalities, n = 72) puts the whole launch on the full layout.  Either way every group's results are what it gets alone."""
    import torch
    from inria_wbc_amd import capi, structure, synth
    names = ["talos", "three_contact", "icub"]
    sts = [structure.STRUCTURES[n]() for n in names]
    assert capi.layout_of(sts[0])["waves_per_cu"] == 2 and capi.layout_of(sts[1])["waves_per_cu"] == 1
    dev = torch.device("cuda", 0)
    h = capi.Handle(0, capi.F64, flags=capi.FLAG_INDEX_ORDER)
    groups, alone = [], []
    for slot, st in enumerate(sts):
        B = 40 + 7 * slot
        inp = synth.generate(st, B, synth.SEED_BASE[st.name] + 4321)
        h.set_structure(slot, st)
        alone.append(h.solve_batch_host(slot, inp))
        d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
        d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
                     status=torch.full((B,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
        groups.append((slot, B, d_in, d_out))
    h.solve_ragged(groups, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for (slot, B, _, d_out), ref, st in zip(groups, alone, sts):
        assert np.array_equal(d_out["status"].cpu().numpy(), ref["status"]), st.name
        assert (ref["status"] == 0).all()
        scale = np.maximum(1.0, np.abs(ref["x"]).max(axis=1, keepdims=True))
        # talos and icub ran the compact kernel alone and the full one in the mixed launch: rounding apart, not bits
        assert (np.abs(d_out["x"].cpu().numpy() - ref["x"]) / scale).max() < 1e-9, st.name
        assert (d_out["iters"].cpu().numpy() == ref["iters"]).mean() >= 0.95, st.name
    h.close()
