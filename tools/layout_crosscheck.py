#!/usr/bin/env python3
"""Compact LDS layout (two QPs per CU) against the full one and against the oracle on hard draws: many tight torque limits and
bounds, loud task right-hand sides -- long active-set histories with partial steps, dual steps and drops, i.e. the paths the
bench batch rarely takes.  Prints one JSON line per case; exits non-zero on a status mismatch or a solution more than 1e-6 (relative) apart: after
hundreds of active-set changes at degenerate vertices the three implementations sit ~1e-7 from each other (1e-8 is the bar for the
reference-like QPs of tests/ and bench.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(h, st, inputs):
    return h.solve_batch_host(0, inputs)


def main():
    from inria_wbc_amd import capi, structure, synth
    from oracle import oracle
    bad = 0
    cases = [("talos", 2048, 5.0, 0.5, 0.3), ("talos", 2048, 2.0, 0.9, 0.6), ("icub", 2048, 5.0, 0.1, 0.5),
             ("talos_single_support", 1024, 5.0, 0.5, 0.3), ("tiago", 1024, 5.0, 0.1, 0.8), ("talos", 1024, 0.5, 0.1, 0.05)]
    for name, B, noise, p_act, p_bnd in cases:
        st = structure.STRUCTURES[name]()
        inputs = synth.generate(st, B, synth.SEED_BASE[name] + 777_000, task_noise=noise, p_act=p_act, p_bnd=p_bnd)
        outs = {}
        for tag, flags in (("compact", 0), ("full", capi.FLAG_FULL_LDS)):
            h = capi.Handle(0, capi.F64, flags=flags)
            h.set_structure(0, st)
            outs[tag] = run(h, st, inputs)
            h.close()
        ns = min(B, 512)
        ref = oracle.tick_batch(st, {k: v[:ns] for k, v in inputs.items()}, nthreads=os.cpu_count() or 1)
        a, b = outs["compact"], outs["full"]
        ok = (a["status"] == 0) & (b["status"] == 0)
        scale = np.maximum(1.0, np.abs(b["x"]).max(axis=1))
        dx = float((np.abs(a["x"] - b["x"]).max(axis=1) / scale)[ok].max()) if ok.any() else 0.0
        oko = ref["status"] == 0
        so = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
        dxo = float((np.abs(a["x"][:ns] - ref["x"]).max(axis=1) / so)[oko].max()) if oko.any() else 0.0
        row = {"structure": name, "batch": B, "noise": noise, "p_act": p_act, "p_bnd": p_bnd,
               "iters_mean": float(a["iters"].mean()), "iters_max": int(a["iters"].max()),
               "status_hist": np.bincount(a["status"] + 1, minlength=6).tolist(),
               "status_equal_full": bool(np.array_equal(a["status"], b["status"])), "iters_equal_full": float((a["iters"] == b["iters"]).mean()),
               "max_rel_dx_vs_full": dx, "status_equal_oracle": bool(np.array_equal(a["status"][:ns], ref["status"])),
               "iters_equal_oracle": float((a["iters"][:ns] == ref["iters"]).mean()), "max_rel_dx_vs_oracle": dxo}
        print(json.dumps(row), flush=True)
        if not row["status_equal_full"] or not row["status_equal_oracle"] or dx > 1e-6 or dxo > 1e-6:
            bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
