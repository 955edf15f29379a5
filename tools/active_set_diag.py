#!/usr/bin/env python3
"""Where the GPU's final active set differs from the oracle's although the iteration counts agree -- and that every such difference sits on the
friction-pyramid facets of the contact POINTS (tests/util.py:friction_facet_rows), never on a row the solution determines (acceleration bounds,
torque limits, normal-force sums).  One line per case: QPs compared, iteration counts equal, sets equal, differing QPs by kind of row, the raw
point-force deviation next to the dv / wrench deviation.  The `facets_differ=` bars of the parity tests are this tool's counts plus one.

    python tools/active_set_diag.py            # GPU (through the C ABI) against the oracle
    python tools/active_set_diag.py --ulp      # CPU only: the oracle against itself with every input moved by one ulp (what "decided by the last
                                               # bits" means: same iteration count, same dv / wrench, other facets)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compare(st, got, ref, what):
    from tests.util import friction_facet_rows, mask_bits, mask_of
    ok = (ref["status"] == 0) & (got["status"] == 0)
    same_it = got["iters"] == ref["iters"]
    gb, rb = mask_bits(mask_of(got["active_mask"]), st.nin2), mask_bits(mask_of(ref["active_mask"]), st.nin2)
    facet = friction_facet_rows(st)[:gb.shape[1]]
    diff = gb != rb
    any_diff = diff.any(axis=1)
    det_diff = diff[:, ~facet].any(axis=1)
    must = ok & same_it
    nv = st.nv
    xs = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
    edv = (np.abs(got["x"][:, :nv] - ref["x"][:, :nv]).max(axis=1) / xs)
    ef = (np.abs(got["x"][:, nv:] - ref["x"][:, nv:]).max(axis=1, initial=0.0) / xs)
    fv = np.abs(np.asarray(got["objective"], np.float64) - ref["fval"]) / np.maximum(1.0, np.abs(ref["fval"]))
    print("%-44s QPs %5d optimal %5d | iters equal %5d | of those: sets equal %5d, facets only %4d, DETERMINED ROWS %3d | iters differ: sets equal %4d of %4d | "
          "dv %.1e raw f %.1e (on facet-differing QPs %.1e) objective %.1e" % (
              what, len(ok), int(ok.sum()), int(must.sum()), int((must & ~any_diff).sum()), int((must & any_diff & ~det_diff).sum()), int((must & det_diff).sum()),
              int((ok & ~same_it & ~any_diff).sum()), int((ok & ~same_it).sum()), edv[must].max(initial=0.0), ef[must].max(initial=0.0),
              ef[must & any_diff].max(initial=0.0), fv[must].max(initial=0.0)), flush=True)
    return int((must & any_diff & ~det_diff).sum()), int((must & det_diff).sum())


def cases():
    from inria_wbc_amd import structure, synth
    from tests import test_gpu_parity as tp
    from tests.test_gpu_layout_variants import CASES, _stack
    for name, batch, noise in tp.PARITY_CASES:
        st = structure.STRUCTURES[name]()
        yield "parity %s B %d noise %g" % (name, batch, noise), st, synth.generate(st, batch, synth.SEED_BASE[name] + 100, task_noise=noise)
    for name, nv, na, nc, act in CASES:
        st = _stack(name, nv, na, nc, act)
        yield "layout " + name, st, synth.generate(st, 192, 77_000 + 31 * nv + nc, task_noise=1.5, p_act=0.3, p_bnd=0.2)
    for name, noise in (("talos", 2.0), ("talos", 5.0), ("icub", 2.0), ("icub", 5.0), ("talos_single_support", 2.0), ("icub_single_support", 2.0)):
        st = structure.STRUCTURES[name]()
        yield "large %s B 1024 noise %g" % (name, noise), st, synth.generate(st, 1024, synth.SEED_BASE.get(name, 7) + 31337, task_noise=noise)


def main():
    from oracle import oracle
    oracle.build()
    ulp = "--ulp" in sys.argv
    h = None
    if not ulp:
        import torch  # noqa: F401
        from inria_wbc_amd import capi
        h = capi.Handle(0, capi.F64)
    tot_f = tot_d = 0
    for what, st, inputs in cases():
        ref = oracle.tick_batch(st, inputs, nthreads=8)
        if ulp:
            rng = np.random.default_rng(1)
            moved = {k: (np.where(rng.random(v.shape) < 0.5, np.nextafter(v, np.inf), np.nextafter(v, -np.inf)) if v.size else v) for k, v in inputs.items()}
            got = oracle.tick_batch(st, moved, nthreads=8)
            got["objective"] = got["fval"]
        else:
            h.set_structure(0, st)
            got = h.solve_batch_host(0, inputs)
        f, d = compare(st, got, ref, what)
        tot_f += f
        tot_d += d
    print("total: facet-only differences on %d QPs, differences on rows the solution determines on %d QPs" % (tot_f, tot_d))
    if h is not None:
        h.close()


if __name__ == "__main__":
    main()
