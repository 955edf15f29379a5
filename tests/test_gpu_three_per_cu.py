"""Three workgroups per CU for a stack whose LDS leaves room for them (csrc/wbcqp_device.hpp: solve_queue3_kernel).  The
compact queue kernel has a twin compiled for three waves per SIMD (168 VGPRs, the rest in scratch); the launch takes it when the runtime says
three workgroups fit a CU: iCub -- BASELINE config 3's stack, 52.8 KB since its rows of J are exactly n long and its friction table lives in J's
dead equality columns -- and iCub on one foot (46.8 KB).  Talos on one foot fits as well since its friction entries moved to two rows of J each (54 480 B)
but stays at two per CU: with actuation bounds the three-per-CU kernel measured slower (8.03 against 9.32 M QP/s: the actuation rows' 38 registers go
to scratch); Talos on two feet (71.6 KB) does not fit.  What must hold: the layout reports
it (tests/test_host.py), and the results are the bits of the two-per-CU kernels (hardware dispatch: solve_kernel, up to 256 VGPRs, no scratch)
and within the parity bar of the oracle -- other registers, same arithmetic (controller.cpp:244-251 is the contract)."""
import numpy as np
import pytest

from tests.util import TOL_F64, assert_parity, device_outputs, host_outputs

pytestmark = pytest.mark.gpu


def _solve(st, inputs, flags=0):
    import torch
    from inria_wbc_amd import capi
    B = next(iter(inputs.values())).shape[0]
    dev = torch.device("cuda", 0)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    h = capi.Handle(0, capi.F64, flags=flags)
    h.set_structure(0, st)
    d_out = device_outputs(B, st, dev)
    d_out["x"].fill_(float("nan")); d_out["tau"].fill_(float("nan")); d_out["iters"].fill_(-1)
    for _ in range(2):  # (the second launch runs in the longest-first order the first one left)
        h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h.close()
    return host_outputs(d_out, st)


@pytest.mark.parametrize("stack", ["icub", "icub_single_support", "talos_single_support"])
@pytest.mark.parametrize("noise", [0.5, 2.0])
def test_three_per_cu_gives_the_bits_of_two_per_cu(oracle_mod, noise, stack):
    from inria_wbc_amd import capi, structure, synth
    st = structure.STRUCTURES[stack]()
    # (Talos on one foot fits three by LDS but runs two per CU -- actuation bounds: measured slower at three --; it rides along as the control)
    assert capi.layout_of(st)["waves_per_cu"] == (2 if stack.startswith("talos") else 3)
    B = 2304  # three rounds of 768 resident workgroups
    inputs = synth.generate(st, B, synth.SEED_BASE["icub"] + 77, task_noise=noise, p_act=0.2 if stack.startswith("talos") else 0.1)
    three = _solve(st, inputs)                                # the queue: solve_queue3_kernel (the stack's own instantiation for iCub on two feet)
    two = _solve(st, inputs, capi.FLAG_HW_DISPATCH)           # solve_kernel, two per CU
    two_idx = _solve(st, inputs, capi.FLAG_HW_DISPATCH | capi.FLAG_INDEX_ORDER)
    gen = _solve(st, inputs, capi.FLAG_GENERIC_KERNEL)        # the generic twin
    assert (three["status"] == 0).all() and three["iters"].max() >= 6
    for k in ("x", "tau", "status", "iters", "objective", "n_active", "active_mask"):
        assert np.array_equal(three[k], two[k]) and np.array_equal(three[k], two_idx[k]) and np.array_equal(three[k], gen[k]), k
    sample = {k: v[:48] for k, v in inputs.items()}
    ref = oracle_mod.tick_batch(st, sample)
    # x / tau / status AND the active set, n_active, objective of the three-per-CU kernel against the oracle's
    info = assert_parity(st, {k: v[:48] for k, v in three.items()}, ref, what="three per CU: %s noise %g" % (stack, noise))
    assert info["iters_equal"] >= 0.9


def test_f32_boundary_takes_the_same_kernel_family():
    """The f32 boundary (BASELINE config 3's dtype) has its own instantiations of the twin: it must run and agree with its two-per-CU form."""
    import torch
    from inria_wbc_amd import capi, structure, synth
    st = structure.icub_structure()
    B = 1024
    inputs = synth.generate(st, B, synth.SEED_BASE["icub"] + 78, dtype=np.float32)
    dev = torch.device("cuda", 0)
    res = []
    for flags in (0, capi.FLAG_HW_DISPATCH):
        d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
        h = capi.Handle(0, capi.F32, flags=flags)
        h.set_structure(0, st)
        o = dict(x=torch.zeros(B, st.n, dtype=torch.float32, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float32, device=dev),
                 status=torch.full((B,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
        h.solve_batch(0, B, d_in, o, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        h.close()
        res.append({k: v.cpu().numpy() for k, v in o.items()})
    assert (res[0]["status"] == 0).all()
    for k in ("x", "tau", "status", "iters"):
        assert np.array_equal(res[0][k], res[1][k]), k
