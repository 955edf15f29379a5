"""Generates the committed golden fixtures tests/golden/*.npz.

The reference (/root/reference) holds no golden vector for this path and cannot be built here
(SURVEY.md 8c), so these fixtures are NOT reference outputs: they are synthetic QPs (inria_wbc_amd.synth)
solved by the fp64 CPU oracle (oracle/wbc_oracle.c) and accepted only after the independent numpy KKT
checker passes. They pin the oracle (and through it the HIP path) against silent drift.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from inria_wbc_amd import structure, synth  # noqa: E402
from oracle import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("franka", 8, 0.5), ("tiago", 8, 2.0), ("icub", 8, 0.5), ("talos", 8, 0.5), ("talos", 6, 5.0),
         ("talos_single_support", 6, 2.0)]


def main():
    for name, batch, noise in CASES:
        st = structure.STRUCTURES[name]()
        inp = synth.generate(st, batch, synth.SEED_BASE[name] + 900_000, task_noise=noise)
        out = oracle.tick_batch(st, inp)
        assert (out["status"] == 0).all(), (name, out["status"])
        for i in range(batch):
            H, g, CE, ce0, CI, ci0 = oracle.assemble(st, inp, i)
            o = oracle.tick_single(st, inp, i)
            k = oracle.kkt_residuals(H, g, CE, ce0, CI, ci0, o["x"], o["active"], o["lam"])
            assert k["stationarity"] < 1e-9 and k["eq"] < 1e-6 and k["min_mu"] > -1e-9, (name, i, k)
        tag = "%s_n%g" % (name, noise)
        path = os.path.join(HERE, tag + ".npz")
        np.savez_compressed(path, **{"in_" + k: v for k, v in inp.items()}, x=out["x"], tau=out["tau"],
                            status=out["status"], iters=out["iters"], task_noise=noise,
                            # HQPOutput's other members ("identical active set"): eiquadprog's A (equality i tagged -i-1, else the
                            # one-sided CI row; padded with INT32_MIN beyond n_active), the same set as the C ABI's 256-bit mask, the objective
                            active=out["active"], n_active=out["n_active"], active_mask=out["active_mask"], fval=out["fval"])
        print("wrote", path, "iters", out["iters"].tolist())


def main_after_path():
    """State integration after the path: q, dq, dv -> q_next, v_next, q_solver by the oracle,
    accepted after the independent scipy check of tests/test_oracle.py holds for the same draw."""
    from scipy.spatial.transform import Rotation as Rot
    os.makedirs(os.path.join(HERE, "after_path"), exist_ok=True)
    for tag, floating, nv in (("talos", True, 50), ("franka", False, 9)):
        rng = np.random.default_rng(77 + nv)
        B, dt = 16, 1e-3
        nq = nv + 1 if floating else nv
        q = rng.normal(size=(B, nq))
        if floating:
            q[:, 3:7] = Rot.random(B, random_state=3).as_quat()
            q[12:, 3:7] = np.array([0.0, 1.0, 0.0, 0.0])  # half turns: trace <= 0 branch
        dq = rng.normal(size=(B, nv))
        dv = 5.0 * rng.normal(size=(B, nv))
        if floating:
            dq[:4, 3:6] = 0.0
            dv[:4, 3:6] = 0.0  # small-angle branch of exp6
        o = oracle.integrate(floating, dt, q, dq, dv)
        if floating:
            rv = Rot.from_quat(o["q_next"][:, 3:7]).as_rotvec()
            assert np.abs(o["q_solver"][:, 3:6] - rv).max() < 1e-13
        path = os.path.join(HERE, "after_path", "integrate_%s.npz" % tag)
        np.savez_compressed(path, floating_base=floating, dt=dt, q=q, dq=dq, dv=dv, **o)
        print("wrote", path)


def main_before_path():
    """The step before the path: q, v, references -> QP record by oracle/rbd_oracle.c for the
    Talos-like and Franka-like models (inria_wbc_amd/model.py; seeded, so the fixture also pins the model builders),
    accepted after the identities of tests/test_oracle_rbd.py hold for the same models."""
    from inria_wbc_amd import model as mdl, structure
    from oracle import rbd
    os.makedirs(os.path.join(HERE, "before_path"), exist_ok=True)
    for tag, m, st, stack in (("talos", mdl.talos_like(), structure.talos_structure(), mdl.talos_stack()),
                              ("franka", mdl.franka_like(), structure.franka_structure(), mdl.franka_stack())):
        tm = mdl.build_taskmap(m, st, stack)
        s = mdl.sample_states(m, tm, 3, 555_000, q_noise=0.05, v_noise=0.2, ref_noise=0.02)
        rows = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"])
        t = rbd.rbd_terms(m, s["q"][0], s["v"][0])
        a = np.random.default_rng(5).standard_normal(m.nv)
        assert np.abs(t["M"] @ a + t["nle"] - rbd.rnea(m, s["q"][0], s["v"][0], a)).max() < 1e-9
        path = os.path.join(HERE, "before_path", "rows_%s.npz" % tag)
        np.savez_compressed(path, q=s["q"], v=s["v"], ref=s["ref"], **rows)
        print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
    main_after_path()
    main_before_path()
