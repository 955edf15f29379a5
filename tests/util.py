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


TOL_RAW_FORCE = 1e-4  # contact-point forces: their internal-force null space is held by the 1e-8 regulariser only


def assert_parity(st, got, ref, tol=TOL_F64, what=""):
    """got/ref: dicts with x, tau, status, iters ([B, ...]).  dv, the contact wrenches T f and tau must agree to `tol`
    (relative to max(1, |.|inf)); the raw contact-point forces f only to TOL_RAW_FORCE: H_ff = w F'F + 1e-8 I has rank-6
    F'F, so six directions of f per contact are conditioned like 1e12 (seen: |df| 1.6e-3 on |f| 215 with dv, T f and tau
    equal to 1e-9, tests/stress/stress_parity.py)."""
    assert np.array_equal(got["status"], ref["status"]), (what, got["status"], ref["status"])
    ok = ref["status"] == 0
    nv = st.nv
    gx, rx = np.asarray(got["x"]), np.asarray(ref["x"])
    xs = np.maximum(1.0, np.abs(rx).max(axis=1))
    ev = np.abs(gx[:, :nv] - rx[:, :nv]).max(axis=1) / xs
    assert (ev[ok] <= tol).all(), (what, "dv", ev.max(), int(ev.argmax()))
    ex = ev.copy()
    if st.nc:
        T = np.asarray(st.force_gen()).reshape(st.nc, 6, 12)
        gf = gx[:, nv:].reshape(-1, st.nc, 12)
        rf = rx[:, nv:].reshape(-1, st.nc, 12)
        gw, rw = np.einsum("cij,bcj->bci", T, gf), np.einsum("cij,bcj->bci", T, rf)
        ws = np.maximum(1.0, np.abs(rw).reshape(len(rx), -1).max(axis=1))
        ew = np.abs(gw - rw).reshape(len(rx), -1).max(axis=1) / ws
        assert (ew[ok] <= tol).all(), (what, "contact wrench", ew.max(), int(ew.argmax()))
        ef = np.abs(gf - rf).reshape(len(rx), -1).max(axis=1) / xs
        assert (ef[ok] <= TOL_RAW_FORCE).all(), (what, "raw contact forces", ef.max(), int(ef.argmax()))
        ex = np.maximum(ev, ew)
    if st.na:
        ts = np.maximum(1.0, np.abs(ref["tau"]).max(axis=1))
        et = np.abs(got["tau"] - ref["tau"]).max(axis=1) / ts
        assert (et[ok] <= tol).all(), (what, "tau", et.max(), int(et.argmax()))
    # active-set iteration counts: identical path expected up to ties broken by rounding
    same = float((got["iters"] == ref["iters"]).mean())
    return dict(max_rel_x=float(ex[ok].max(initial=0.0)), iters_equal=same)
