"""The compact layout's branches that no shipped stack takes together (csrc/wbcqp_api.hip: derive_compact; csrc/wbcqp_compact.hpp: cp::vec_map,
cp::fric_in_j): 64-entry vector slots WITH the actuation-bound slots, an odd n (rows of J padded, not aliased), n = 2 mod 4 with fewer than fourteen
equalities (rows aliased on the next row's dead columns, the friction table over two rows of J per entry), n = 0 mod 4 (padded rows, friction table in J),
and a small floating-base stack on one contact (n = 30 < 34: cp::fric_in_j == 0, the friction table BEHIND THE ROTATION TABLE IN THE R REGION -- the
branch whose first two coefficients once shared two doubles with the loop's zeroing pass), with and without actuation bounds.  Each
stack runs the generic compact kernel -- through the queue, three per CU where it fits -- against the oracle and against round 1's full layout
(WBCQP_FLAG_FULL_LDS), which shares none of these branches.  The reference contract is the same one as everywhere: controller.cpp:244-251."""
import numpy as np
import pytest

from tests.util import assert_parity, device_outputs, host_outputs

pytestmark = pytest.mark.gpu


def _stack(name, nv, na, n_contacts, act):
    from inria_wbc_amd import structure as S
    pts = S.contact6d_points(lxn=0.06, lyn=0.045, lxp=0.14, lyp=0.045, lz=0.065)
    contacts = [S.Contact("contact_%d" % c, pts, (0.0, 0.0, 1.0), 0.3, 5.0, 1500.0) for c in range(n_contacts)]
    dense = [("lh", 6, 1.0), ("rh", 6, 1.0), ("lf", 6, 100.0), ("com", 3, 1000.0), ("momentum", 2, 100.0), ("__posture__", "posture", 0.5), ("torso", 3, 1.0),
             ("__contacts__",)]
    level0 = [(S.INEQ_BOUNDS, 0)] + ([(S.INEQ_ACTUATION, 0)] if act else []) + [(S.INEQ_FORCE, c) for c in range(n_contacts)]
    return S._mk(name, nv, na, contacts, dense, None, [("self_collision-a", 500.0)], True, act, level0, {"com": 30.0, "posture": 10.0})


CASES = [("slots64_with_actuation", 38, 32, 2, True),      # n 62 (aliased rows, table in J), VS 64, TL / TU present
         ("odd_n", 37, 31, 2, False),                      # n 61: padded rows
         ("one_contact_aliased", 38, 32, 1, True),          # n 50 = 2 mod 4, nEq 12: aliased rows, table in columns 2..7 of two rows of J per entry
         ("table_in_r_region", 18, 12, 1, False),           # n 30 = 2 mod 4, nEq 12, n < 34 nc: cp::fric_in_j == 0, the table behind the rotation table
         ("table_in_r_region_act", 18, 12, 1, True),        # ... with actuation bounds
         ("n_multiple_of_four", 40, 34, 2, False),          # n 64: padded rows, 80-entry slots, table in J
         ("one_contact_padded", 40, 34, 1, False)]          # n 52 = 0 mod 4, nEq 12


def _solve(st, inputs, flags):
    import torch
    from inria_wbc_amd import capi
    B = next(iter(inputs.values())).shape[0]
    dev = torch.device("cuda", 0)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    h = capi.Handle(0, capi.F64, flags=flags)
    h.set_structure(0, st)
    o = device_outputs(B, st, dev)
    o["x"].fill_(float("nan")); o["tau"].fill_(float("nan")); o["iters"].fill_(-1); o["active_mask"].fill_(-1); o["n_active"].fill_(-1)
    h.solve_batch(0, B, d_in, o, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h.close()
    return host_outputs(o, st)


@pytest.mark.parametrize("name,nv,na,nc,act", CASES)
def test_layout_branches_against_the_oracle_and_the_full_layout(oracle_mod, name, nv, na, nc, act):
    from inria_wbc_amd import capi, synth
    st = _stack(name, nv, na, nc, act)
    L = capi.layout_of(st)
    assert L["dense_h"] == 0 and L["specialised"] == 0 and L["wave_per_qp"] == 0 and L["waves_per_cu"] in (2, 3), L  # the compact layout, generic kernel
    B = 192
    inputs = synth.generate(st, B, 77_000 + 31 * nv + nc, task_noise=1.5, p_act=0.3, p_bnd=0.2)
    got = _solve(st, inputs, 0)
    two = _solve(st, inputs, capi.FLAG_HW_DISPATCH)
    full = _solve(st, inputs, capi.FLAG_FULL_LDS)
    ref = oracle_mod.tick_batch(st, inputs)
    assert (ref["status"] == 0).mean() > 0.9 and ref["iters"].max() >= 8, (name, ref["iters"].max())
    assert_parity(st, got, ref, what=name)
    assert_parity(st, full, ref, what=name + " (full layout)")  # round 1's kernel: its own active_mask / objective against the oracle's too
    for k in ("x", "tau", "status", "iters", "objective", "n_active", "active_mask"):
        assert np.array_equal(got[k], two[k], equal_nan=True), (name, k)  # the queue (two or three per CU) and the dispatcher: same bits
    assert np.array_equal(got["status"], full["status"])
    ok = got["status"] == 0
    scale = np.maximum(1.0, np.abs(full["x"]).max(axis=1))
    assert (np.abs(got["x"][:, :nv] - full["x"][:, :nv]).max(axis=1)[ok] <= 1e-8 * scale[ok]).all(), name
    # iteration counts: equal on most QPs; on draws this hard (tens of picks per QP) two near-tied picks may swap -- the measured floor of such batches is
    # 0.82 (tests/stress/stress_parity.py, profiles/r05/v33_stress_parity.txt), 0.885 on the hardest case here
    assert (got["iters"] == ref["iters"]).mean() >= 0.8


def test_poisoned_lds_build_gives_the_same_bits(tmp_path):
    """The compact kernel reads no LDS word it did not write itself behind a barrier: the build that fills the QP's whole LDS block with a NaN pattern
    first (-DWBCQP_POISON_LDS, inria_wbc_amd/build.py:build_poison) must give the product library's bits on every layout branch above -- the R-region
    friction table included, whose first two coefficients the loop's zeroing pass once overwrote from another wave -- and on the shipped
    humanoid stacks, through the queue, the hardware dispatcher and the generic kernel.  One process per library."""
    import os
    import subprocess
    import sys
    from inria_wbc_amd import build
    build.build()
    build.build_poison()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name in ("libwbcqp.so", "libwbcqp_poison.so"):
        path = str(tmp_path / (name + ".npz"))
        subprocess.check_call([sys.executable, os.path.join(root, "tools", "chk_variants.py"), name, path], cwd=root)
        outs[name] = np.load(path)
    prod, pois = outs["libwbcqp.so"], outs["libwbcqp_poison.so"]
    assert sorted(prod.files) == sorted(pois.files) and len(prod.files) >= 10 * 3 * 7
    for k in prod.files:
        assert np.array_equal(prod[k], pois[k], equal_nan=True), k
        if k.endswith("/status"):
            assert (prod[k] == 0).mean() > 0.9, k
