"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

from inria_wbc_amd import structure, synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp64 parity bar: |x_gpu - x_cpu|_inf <= 1e-8 * max(1, |x|_inf), same status.
TOL_F64 = 1e-8


def load_golden():
    cases = []
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        z = np.load(path)
        name = os.path.basename(path).rsplit("_n", 1)[0]
        inputs = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
        cases.append((os.path.basename(path), structure.STRUCTURES[name](), inputs, z))
    return cases


def assert_parity(st, got, ref, tol=TOL_F64, what=""):
    """got/ref: dicts with x, tau, status, iters ([B, ...])."""
    assert np.array_equal(got["status"], ref["status"]), (what, got["status"], ref["status"])
    ok = ref["status"] == 0
    xs = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
    ex = np.abs(got["x"] - ref["x"]).max(axis=1) / xs
    assert (ex[ok] <= tol).all(), (what, "x", ex.max(), int(ex.argmax()))
    if st.na:
        ts = np.maximum(1.0, np.abs(ref["tau"]).max(axis=1))
        et = np.abs(got["tau"] - ref["tau"]).max(axis=1) / ts
        assert (et[ok] <= tol).all(), (what, "tau", et.max(), int(et.argmax()))
    # active-set iteration counts: identical path expected up to ties broken by rounding
    same = float((got["iters"] == ref["iters"]).mean())
    return dict(max_rel_x=float(ex[ok].max(initial=0.0)), iters_equal=same)
