"""tsid status 3, "Max iter reached" (src/controllers/controller.cpp:297-299): eiquadprog-fast leaves its loop when the iteration counter
reaches maxIter and tsid's SolverHQuadProgFast maps MAX_ITER_REACHED to HQP_STATUS_MAX_ITER_REACHED (A.2).  Every kernel
that runs the loop must stop on the same iteration as the oracle: the compact layout through a shipped stack's own instantiation and through
the generic kernel, the full layout (WBCQP_FLAG_FULL_LDS), one wavefront per QP (Tiago) and the dense seam.  QPs that finish below the bound
must be untouched by it: their x are the bits of the unbounded run."""
import dataclasses

import numpy as np
import pytest

from tests.util import TOL_F64

pytestmark = pytest.mark.gpu


def _solve(st, inputs, flags=0):
    import torch
    from inria_wbc_amd import capi
    B = next(iter(inputs.values())).shape[0]
    dev = torch.device("cuda", 0)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    h = capi.Handle(0, capi.F64, flags=flags)
    h.set_structure(0, st)
    d_out = dict(x=torch.full((B, st.n), float("nan"), dtype=torch.float64, device=dev), tau=torch.full((B, st.na), float("nan"), dtype=torch.float64, device=dev),
                 status=torch.full((B,), -99, dtype=torch.int32, device=dev), iters=torch.full((B,), -1, dtype=torch.int32, device=dev))
    h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h.close()
    return {k: v.cpu().numpy() for k, v in d_out.items()}


def _check(oracle_mod, st0, inputs, flags, max_iter, free, what):
    """free: the same kernel's run with the default bound (1000)."""
    st = dataclasses.replace(st0, max_iter=max_iter)
    ref = oracle_mod.tick_batch(st, inputs, nthreads=4)
    got = _solve(st, inputs, flags)
    assert np.array_equal(got["status"], ref["status"]), (what, max_iter, got["status"], ref["status"])
    hit = ref["status"] == 3
    assert hit.any() and (ref["status"][~hit] == 0).all(), (what, max_iter, np.unique(ref["status"]))
    # a stopped QP reports the iteration it stopped on; the others their own count
    assert (got["iters"][hit] == max_iter).all() and (ref["iters"][hit] == max_iter).all(), (what, max_iter)
    # ... and the QPs that finished are the unbounded run's, bit for bit (the bound is a comparison, not arithmetic) and the oracle's to 1e-8
    done = ~hit
    if done.any():
        assert (np.abs(got["iters"][done] - ref["iters"][done]) <= 1).all() and (got["iters"][done] == ref["iters"][done]).mean() >= 0.9
        assert np.array_equal(got["x"][done], free["x"][done]) and np.array_equal(got["tau"][done], free["tau"][done]), (what, max_iter)
        scale = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
        assert (np.abs(got["x"] - ref["x"]).max(axis=1)[done] <= TOL_F64 * scale[done]).all(), (what, max_iter)
    return int(hit.sum())


@pytest.mark.parametrize("kernel", ["specialised", "generic", "full_lds"])
def test_max_iter_on_the_four_wave_kernels(oracle_mod, kernel):
    from inria_wbc_amd import capi, structure, synth
    st = structure.talos_structure()
    flags = {"specialised": 0, "generic": capi.FLAG_GENERIC_KERNEL, "full_lds": capi.FLAG_FULL_LDS}[kernel]
    inputs = synth.generate(st, 96, synth.SEED_BASE["talos"] + 31, task_noise=1.0)
    free = _solve(st, inputs, flags)
    assert (free["status"] == 0).all() and free["iters"].min() <= 2 and free["iters"].max() >= 12
    n_hit = [_check(oracle_mod, st, inputs, flags, m, free, kernel) for m in (1, 3, 8)]
    assert n_hit[0] == 96 and n_hit[0] > n_hit[1] > n_hit[2] > 0  # max_iter 1 stops every QP before its first pick


def test_a_bound_other_than_1000_keeps_the_stacks_own_kernel(oracle_mod):
    """The per-stack instantiations carry every size as a literal (csrc/wbcqp_types.hpp: kSpecDims) except the iteration bound, which they
    read from the structure: a caller that bounds its tick time (max_iter = 200) keeps its stack's instantiation, and that kernel, the same
    one with the default bound and the generic kernel give the same bits wherever no QP is stopped."""
    from inria_wbc_amd import capi, structure, synth
    st = structure.talos_structure()
    assert capi.layout_of(st)["specialised"] == 1
    st200 = dataclasses.replace(st, max_iter=200)
    assert capi.layout_of(st200)["specialised"] == 1 and capi.layout_of(st200)["waves_per_cu"] == 2
    inputs = synth.generate(st, 200, synth.SEED_BASE["talos"] + 32, task_noise=2.0)
    a, b, g = _solve(st, inputs), _solve(st200, inputs), _solve(st200, inputs, capi.FLAG_GENERIC_KERNEL)
    assert (a["status"] == 0).all() and a["iters"].max() < 200
    for k in ("x", "tau", "status", "iters"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], g[k]), k


def test_max_iter_on_one_wavefront_per_qp(oracle_mod):
    from inria_wbc_amd import capi, structure, synth
    st = structure.STRUCTURES["tiago"]()
    assert capi.layout_of(st)["wave_per_qp"] == 1
    inputs = synth.generate(st, 300, synth.SEED_BASE["tiago"] + 9, task_noise=30.0, p_bnd=0.3)
    for flags, what in ((0, "one wave per QP"), (capi.FLAG_WORKGROUP_PER_QP, "four waves, compact layout")):
        free = _solve(st, inputs, flags)
        assert (free["status"] == 0).all() and free["iters"].max() >= 5
        for m in (1, 3):
            _check(oracle_mod, st, inputs, flags, m, free, what)


def test_max_iter_on_the_dense_seam(oracle_mod):
    from inria_wbc_amd import capi, structure, synth
    st = structure.talos_structure()
    B = 8
    inputs = synth.generate(st, B, synth.SEED_BASE["talos"] + 33, task_noise=2.0)
    qps = [[np.ascontiguousarray(a) for a in oracle_mod.assemble(st, inputs, i)] for i in range(B)]
    stack = lambda j: np.stack([q[j] for q in qps])
    h = capi.Handle(0, capi.F64)
    for m in (0, 1, 3, 8):  # 0: the default bound
        got = h.solve_dense_host(stack(0), stack(1), stack(2), stack(3), stack(4), stack(5), max_iter=m)
        for i, (H, g, CE, ce0, CI, ci0) in enumerate(qps):
            ref = oracle_mod.eiquadprog(H, g, CE, ce0, CI, ci0, max_iter=m if m else 1000)
            assert got["status"][i] == {0: 0, 3: 3}[ref["status"]], (m, i, got["status"][i], ref["status"])
            if ref["status"] == 3:
                assert got["iters"][i] == m == ref["iters"]
            else:
                assert np.abs(got["x"][i] - ref["x"]).max() <= 1e-8 * max(1.0, np.abs(ref["x"]).max())
        if m in (1, 3):
            assert (got["status"] == 3).any()
    h.close()
