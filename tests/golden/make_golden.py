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
                            status=out["status"], iters=out["iters"], task_noise=noise)
        print("wrote", path, "iters", out["iters"].tolist())


if __name__ == "__main__":
    main()
