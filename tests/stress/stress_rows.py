#!/usr/bin/env python3
"""Fuzz of the rows kernel (wbcqp_problem_data) against oracle/rbd_oracle.c: random trees (every joint type, random
branching, depth up to a chain), random task stacks, large states.  Usage (GPU box): python tests/stress/stress_rows.py [--n 60]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(n, tol=1e-9):
    from inria_wbc_amd import capi
    from inria_wbc_amd import model as mdl
    from oracle import rbd
    h = capi.Handle(0, capi.F64)
    worst = {}
    for k in range(n):
        rng = np.random.default_rng(1000 + k)
        fb = bool(rng.integers(0, 2))
        nb = int(rng.integers(3, 46 if fb else 62))
        m = mdl.random_tree(2000 + k, nb, fb, nframe=int(rng.integers(8, 20)))
        if k % 7 == 0:  # a pure chain: the deepest tree the lane count allows
            m.parent = np.arange(-1, nb - 1, dtype=np.int32)
            m.validate()
        st, stack = mdl.random_stack(m, 3000 + k)
        try:
            h.set_structure(0, st)
        except capi.WbcqpError as e:
            print("case %d skipped (solver limit): %s" % (k, e))
            continue
        tm = mdl.build_taskmap(m, st, stack, dt=float(rng.choice([1e-3, 2e-3, 5e-3])))
        h.set_model(0, m, tm)
        s = mdl.sample_states(m, tm, 16, 4000 + 16 * k, q_noise=0.5, v_noise=1.0, ref_noise=0.3)
        dev = h.problem_data_host(0, s["q"], s["v"], s["ref"])
        ora = rbd.task_rows(m, tm, st, s["q"], s["v"], s["ref"], n_threads=8)
        for f in capi.ROW_FIELDS:
            if ora[f].size:
                e = float(np.abs(dev[f] - ora[f]).max() / max(1.0, np.abs(ora[f]).max()))
                worst[f] = max(worst.get(f, 0.0), e)
                assert np.isfinite(dev[f]).all() and e < tol, (k, f, e, nb, fb)
    h.close()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    args = ap.parse_args()
    worst = run(args.n)
    print("cases %d, worst relative differences:" % args.n, {f: "%.1e" % e for f, e in worst.items()})


if __name__ == "__main__":
    main()
