#!/usr/bin/env python3
"""The floor of `iters_equal` (share of QPs whose active-set iteration count equals the oracle's) for every case of
tests/test_gpu_parity.py: what the tests' thresholds are set from (the measured value minus one QP).  One line per case."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    from inria_wbc_amd import capi, structure, synth
    from oracle import oracle
    from tests import test_gpu_parity as tp
    from tests.util import assert_parity, load_golden
    h = capi.Handle(0, capi.F64)
    for fname, st, inputs, z in load_golden():
        h.set_structure(0, st)
        got = h.solve_batch_host(0, inputs)
        info = assert_parity(st, got, {k: z[k] for k in ("x", "tau", "status", "iters", "active_mask", "n_active", "fval")}, what=fname)
        n = len(z["iters"])
        print("golden %-32s n %4d iters_equal %.4f (%d differ) max_rel_x %.2e active sets equal %.4f raw force %.2e objective %.2e" % (fname, n, info["iters_equal"], round(n * (1 - info["iters_equal"])), info["max_rel_x"], info["active_set_equal_frac"], info.get("max_rel_raw_force", 0.0), info["max_rel_objective"]))
    for name, batch, noise in tp.PARITY_CASES:
        st = structure.STRUCTURES[name]()
        inputs = synth.generate(st, batch, synth.SEED_BASE[name] + 100, task_noise=noise)
        ref = oracle.tick_batch(st, inputs, nthreads=8)
        h.set_structure(1, st)
        got = h.solve_batch_host(1, inputs)
        info = assert_parity(st, got, ref, what=name)
        print("parity %-24s noise %.1f n %4d iters_equal %.4f (%d differ) max_rel_x %.2e active sets equal %.4f raw force %.2e objective %.2e" % (name, noise, batch, info["iters_equal"], round(batch * (1 - info["iters_equal"])), info["max_rel_x"], info["active_set_equal_frac"], info.get("max_rel_raw_force", 0.0), info["max_rel_objective"]))
    h.close()


if __name__ == "__main__":
    main()
