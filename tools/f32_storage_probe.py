#!/usr/bin/env python3
"""SURVEY section 7 lists "fp32 storage with fp64 accumulate" as an option for BASELINE config 3 (iCub, fp32).  DESIGN section 3 argues
against it (cond(H) ~ 1e12); this MEASURES it, on the CPU, with the numpy model of the kernel's loop (tools/gi_rinv_proto.py): J and R^-1
-- the two arrays that stay in LDS for the whole active-set loop -- rounded to f32 after every update, every product and sum in f64, on
config 3's own QPs (inputs rounded to f32 like the boundary).  Prints, against the f64 run of the same model: status changes, iteration
count changes, the error of dv / the contact wrench / tau.   python tools/f32_storage_probe.py [--n 48]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--robot", default="icub")
    args = ap.parse_args()
    from inria_wbc_amd import structure, synth
    from oracle import oracle
    from tools import gi_rinv_proto as gp
    st = structure.STRUCTURES[args.robot]()
    inp = synth.generate(st, args.n, synth.SEED_BASE[args.robot])
    inp = {k: v.astype(np.float32).astype(np.float64) for k, v in inp.items()}
    T = np.asarray(st.force_gen()).reshape(st.nc, 6, 12) if st.nc else None
    rows = []
    for i in range(args.n):
        H, g, CE, ce0, CI, ci0 = oracle.assemble(st, inp, i)
        a = gp.solve(H, g, CE, ce0, CI, ci0)
        b = gp.solve(H, g, CE, ce0, CI, ci0, store=np.float32)
        nv = st.nv
        sc = max(1.0, np.abs(a["x"]).max())
        edv = np.abs(a["x"][:nv] - b["x"][:nv]).max() / sc
        ew = 0.0
        if st.nc:
            fa, fb = a["x"][nv:].reshape(st.nc, 12), b["x"][nv:].reshape(st.nc, 12)
            wa, wb = np.einsum("cij,cj->ci", T, fa), np.einsum("cij,cj->ci", T, fb)
            ew = np.abs(wa - wb).max() / max(1.0, np.abs(wa).max())
        # equality residual of the f32-storage iterate: how well J's null space still is one
        eq = np.abs(CE @ b["x"] + ce0).max()
        rows.append((a["status"], b["status"], a["iters"], b["iters"], edv, ew, eq))
    r = np.array(rows, dtype=np.float64)
    print("%s, %d QPs of config 3's generator (inputs rounded to f32); f64 model vs J, R^-1 STORED in f32 (accumulation f64):" % (args.robot, args.n))
    print("  status changed: %d of %d   (f32-storage statuses: %s)" % (int((r[:, 0] != r[:, 1]).sum()), args.n, np.bincount(r[:, 1].astype(int)).tolist()))
    ok = (r[:, 0] == 0) & (r[:, 1] == 0)
    print("  iteration count changed: %d of %d solved by both" % (int((r[ok, 2] != r[ok, 3]).sum()), int(ok.sum())))
    if ok.any():
        print("  rel. error dv:     median %.2e  max %.2e" % (np.median(r[ok, 4]), r[ok, 4].max()))
        print("  rel. error wrench: median %.2e  max %.2e" % (np.median(r[ok, 5]), r[ok, 5].max()))
        print("  equality residual |CE x + ce0|: median %.2e  max %.2e  (f64: ~1e-10)" % (np.median(r[ok, 6]), r[ok, 6].max()))
    print("  SURVEY's fp32 tolerance: 1e-3 relative on ddq / tau")


if __name__ == "__main__":
    main()
