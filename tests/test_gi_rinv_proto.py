"""The arithmetic of the compact kernel's active-set loop (round 3: R^-1 carried instead of R, drop rotations from one prefix sum, rank-one
updates of d / z / r after a drop; round 5: the rotations in closed form -- one running sum per row -- and a constraint's reflector kept
pending and folded into the next pick's d: tools/gi_rinv_proto.py) against the C oracle's eiquadprog-fast restatement
(oracle/wbc_oracle.c).  CPU only: pins the MATH the HIP kernel implements before any GPU is involved."""
import numpy as np
import pytest

from inria_wbc_amd import structure, synth
from oracle import oracle
from tools import gi_rinv_proto as proto


@pytest.mark.parametrize("name,first,count,kw", [
    ("talos", 0, 6, dict(task_noise=5.0)),          # 20-50 iterations, up to 20 drops each
    ("talos", 725, 8, dict(squat=True)),            # bench.py's stream around its heavy ticks
    ("icub", 0, 6, dict(task_noise=2.0)),
    ("talos_single_support", 0, 4, dict(task_noise=2.0)),
])
@pytest.mark.parametrize("projector,round5", [(False, False), (True, False), (False, True)])
def test_proto_matches_oracle(name, first, count, kw, projector, round5):
    st = structure.STRUCTURES[name]()
    seed = synth.SEED_BASE["talos_squat" if kw.get("squat") else name]
    inp = synth.generate(st, count, seed, first=first, **kw)
    ref = oracle.tick_batch(st, inp)
    drops = 0
    for i in range(count):
        H, g, CE, ce0, CI, ci0 = oracle.assemble(st, inp, i)
        tr = {}
        out = proto.solve(H, g, CE, ce0, CI, ci0, trace=tr, projector=projector, round5=round5)
        assert out["status"] == ref["status"][i]
        scale = max(1.0, float(np.abs(ref["x"][i]).max()))
        assert np.abs(out["x"] - ref["x"][i]).max() <= 1e-8 * scale
        assert abs(out["iters"] - ref["iters"][i]) <= 2
        drops += tr.get("partial_steps", 0)
    if kw.get("task_noise", 0) >= 5.0:
        assert drops > 10  # the drop path is what this test is about


def test_proto_reports_infeasible():
    # x >= 1 and -x >= 0 cannot both hold
    H = np.eye(2); g = np.zeros(2)
    CI = np.array([[1.0, 0.0], [-1.0, 0.0]]); ci0 = np.array([-1.0, 0.0])
    out = proto.solve(H, g, np.zeros((0, 2)), np.zeros(0), CI, ci0)
    ref = oracle.eiquadprog(H, g, np.zeros((0, 2)), np.zeros(0), CI, ci0)
    assert out["status"] == proto.INFEASIBLE and ref["status"] != 0


def test_f32_storage_with_f64_accumulation_stays_within_the_fp32_tolerance(oracle_mod):
    """SURVEY section 7's option for BASELINE config 3, measured on the numpy model of the kernel's loop (tools/f32_storage_probe.py,
    profiles/r04/f32_storage_probe.txt): J and R^-1 rounded to f32 after every update, sums and products in f64.  Same statuses, same
    iteration counts, dv and the contact wrench within 1e-4 of the f64 run -- inside SURVEY 8(d)'s fp32 bar of 1e-3.  (What the product
    does for WBCQP_F32 handles is stricter: f32 arrays at the boundary, f64 inside -- DESIGN section 3 says why the narrower storage buys nothing.)"""
    from inria_wbc_amd import structure, synth
    from tools import gi_rinv_proto as gp
    st = structure.icub_structure()
    inp = synth.generate(st, 12, synth.SEED_BASE["icub"])
    inp = {k: v.astype(np.float32).astype(np.float64) for k, v in inp.items()}
    for i in range(12):
        H, g, CE, ce0, CI, ci0 = oracle_mod.assemble(st, inp, i)
        a = gp.solve(H, g, CE, ce0, CI, ci0)
        b = gp.solve(H, g, CE, ce0, CI, ci0, store=np.float32)
        assert a["status"] == b["status"] == 0 and a["iters"] == b["iters"]
        sc = max(1.0, np.abs(a["x"]).max())
        assert np.abs(a["x"][:st.nv] - b["x"][:st.nv]).max() / sc <= 1e-4
        assert np.abs(a["x"][:st.nv] - b["x"][:st.nv]).max() / sc >= 1e-9  # and it IS a different arithmetic
