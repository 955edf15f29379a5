#!/usr/bin/env python3
"""Turns a directory written by tools/dump_reference_vectors (the reference's own tsid / eiquadprog run) into a golden file
in the schema of tests/golden/*.npz:  in_M, in_h, in_A, in_b1, in_Ac, in_bc, in_blb, in_bub, in_tlb, in_tub, in_w, x, tau,
status, iters (+ q, v, source = "reference").  tests/test_reference_vectors.py picks up every tests/golden/reference/*.npz.

    python tools/pack_reference_vectors.py <dump_dir> <structure name> <out.npz>

The mapping is checked while it is made: every level-0 / level-1 constraint of tsid's HQPData must be the block this
repository's compact record implies; a constraint that is not -- wrong order, wrong shape,
a selection matrix that is not one, a friction block that differs from the structure's -- stops the packer with its name.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from inria_wbc_amd import structure as S  # noqa: E402


class PackError(RuntimeError):
    pass


def read_tick(d):
    idx = []
    with open(os.path.join(d, "hqp_index.txt")) as fh:
        for line in fh:
            stem, kind, rows, cols, weight = line.split()
            c = {"stem": stem, "level": int(stem[1:stem.index("_")]), "kind": kind, "rows": int(rows), "cols": int(cols), "w": float(weight)}
            for suffix in ("A", "b", "lb", "ub"):
                p = os.path.join(d, "%s_%s.npy" % (stem, suffix))
                if os.path.exists(p):
                    c[suffix] = np.load(p)
            idx.append(c)
    out = {k: np.load(os.path.join(d, k + ".npy")) for k in ("q", "v", "x", "tau", "M", "h")}
    si = np.load(os.path.join(d, "status_iters.npy"))
    out["status"], out["iters"] = int(si[0]), int(si[1])
    out["hqp"] = idx
    # HQPOutput's other members, when the dump holds them (written since round 6): kept raw -- how tsid slices eiquadprog's active list into
    # HQPOutput::activeSet is [UPSTREAM-RECALL]; tests/test_reference_vectors.py compares them as SETS of one-sided rows with the oracle's
    for k in ("active_set", "lambda", "objective"):
        p = os.path.join(d, k + ".npy")
        if os.path.exists(p):
            out[k] = np.load(p)
    return out


def _need(cond, what):
    if not cond:
        raise PackError(what)


def pack_tick(st, tick, tol=1e-12):
    """one dumped tick -> one row of every record array"""
    nv, na, nu, n, nc = st.nv, st.na, st.nu, st.n, st.nc
    M, h = tick["M"], tick["h"]
    _need(M.shape == (nv, nv) and h.shape == (nv,), "M / h have the wrong size for structure %s" % st.name)
    rec = {"M": M[np.tril_indices(nv)], "h": h}
    T = st.force_gen()
    Bf, lbf, ubf = st.friction()
    l0 = [c for c in tick["hqp"] if c["level"] == 0]
    l1 = [c for c in tick["hqp"] if c["level"] == 1]
    # ---- level 0, in tsid's order: base dynamics (formulation ctor), then the task stack's constraints in the order they were added
    pos = 0
    Ac = np.zeros((nc, 6, nv))
    bc = np.zeros((nc, 6))
    if nu:
        c = l0[pos]; pos += 1
        _need(c["kind"] == "eq" and c["rows"] == nu, "level 0 does not start with the base dynamics (%s)" % c["stem"])
        _need(np.abs(c["A"][:, :nv] - M[:nu]).max() <= tol * max(1.0, np.abs(M).max()), "base dynamics rows differ from M_u")
        _need(np.abs(c["b"] + h[:nu]).max() <= tol * max(1.0, np.abs(h).max()), "base dynamics rhs differs from -h_u")
        base_JuT = -c["A"][:, nv:]
    blb = bub = np.zeros(0)
    tlb = tub = np.zeros(0)
    seen_contacts = 0
    for kind, arg in st.ineq_blocks:
        c = l0[pos]; pos += 1
        if kind == S.INEQ_BOUNDS:
            _need(c["rows"] == st.n_bound, "bounds task has %d rows, structure says %d (%s)" % (c["rows"], st.n_bound, c["stem"]))
            if "A" in c:
                sel = np.zeros((st.n_bound, n)); sel[np.arange(st.n_bound), st.bound_col] = 1.0
                _need(np.array_equal(c["A"], sel), "bounds matrix is not the selection of the actuated joints")
            blb, bub = c["lb"], c["ub"]
        elif kind == S.INEQ_ACTUATION:
            _need(c["rows"] == na, "actuation bounds have %d rows (%s)" % (c["rows"], c["stem"]))
            _need(np.abs(c["A"][:, :nv] - M[nu:]).max() <= tol * max(1.0, np.abs(M).max()), "actuation rows differ from M_a")
            tlb, tub = c["lb"] + h[nu:], c["ub"] + h[nu:]  # the constraint stores lb - h_a, ub - h_a (A.1 step 6)
        else:
            # addRigidContact pushes the force inequality, then the motion equality
            _need(c["rows"] == 17, "contact %d: force constraint has %d rows (%s)" % (arg, c["rows"], c["stem"]))
            blk = c["A"][:, nv + 12 * arg:nv + 12 * arg + 12]
            _need(np.abs(blk - Bf[arg]).max() <= 1e-9, "contact %d: friction block differs from the structure's" % arg)
            _need(np.abs(c["lb"] - lbf[arg]).max() <= 1e-6 and np.abs(c["ub"] - ubf[arg]).max() <= 1e-6, "contact %d: force bounds differ" % arg)
            m = l0[pos]; pos += 1
            _need(m["kind"] == "eq" and m["rows"] == 6, "contact %d: no 6-row motion equality after the force constraint (%s)" % (arg, m["stem"]))
            Ac[arg], bc[arg] = m["A"][:, :nv], m["b"]
            seen_contacts += 1
    _need(pos == len(l0), "level 0 holds %d constraints, the structure explains %d" % (len(l0), pos))
    _need(seen_contacts == nc, "contacts found %d, structure has %d" % (seen_contacts, nc))
    if nu and nc:
        Jc = np.concatenate([T[c_].T @ Ac[c_] for c_ in range(nc)], axis=0)
        _need(np.abs(base_JuT - Jc[:, :nu].T).max() <= 1e-9 * max(1.0, np.abs(Jc).max()), "base dynamics force columns differ from -J_u'")
    rec.update(Ac=Ac.reshape(-1), bc=bc.reshape(-1), blb=blb, bub=bub, tlb=tlb, tub=tub)
    # ---- level 1, in the order of the structure's weight vector: dense motion rows, the posture selection, force regularisation
    A = np.zeros((st.n_dense, nv))
    b1 = np.zeros(st.r1)
    w = np.zeros(st.n_tasks)
    _need(len(l1) == st.n_tasks, "level 1 holds %d tasks, the structure has %d" % (len(l1), st.n_tasks))
    F = st.forcereg_mat()
    for t, c in enumerate(l1):
        w[t] = c["w"]
        rows_d = np.where(st.dense_row_task == t)[0]
        rows_s = np.where(st.sel_task == t)[0]
        if rows_d.size:
            _need(c["rows"] == rows_d.size, "task %d (%s): %d rows, structure says %d" % (t, c["stem"], c["rows"], rows_d.size))
            _need(not np.any(c["A"][:, nv:]), "task %d (%s) touches force columns" % (t, c["stem"]))
            A[rows_d] = c["A"][:, :nv]
            b1[rows_d] = c["b"]
        elif rows_s.size:
            sel = np.zeros((rows_s.size, n)); sel[np.arange(rows_s.size), st.sel_col[rows_s]] = 1.0
            _need(c["A"].shape == sel.shape and np.array_equal(c["A"], sel), "task %d (%s) is not the posture selection" % (t, c["stem"]))
            b1[st.n_dense + rows_s] = c["b"]
        else:
            ct = int(np.where(st.forcereg_task == t)[0][0])
            blk = c["A"][:, nv + 12 * ct:nv + 12 * ct + 12]
            _need(np.abs(blk - F[ct]).max() <= 1e-12, "task %d (%s): force-regularisation block differs from diag(w_f) T" % (t, c["stem"]))
            b1[st.n_dense + st.n_sel + 6 * ct:st.n_dense + st.n_sel + 6 * ct + 6] = c["b"]
    rec.update(A=A.reshape(-1), b1=b1, w=w)
    return rec


def pack(dump_dir, st):
    ticks = sorted(d for d in os.listdir(dump_dir) if d.startswith("tick"))
    if not ticks:
        raise PackError("no tick directories in " + dump_dir)
    rows, outs, extras = [], {"x": [], "tau": [], "status": [], "iters": [], "q": [], "v": []}, []
    for d in ticks:
        t = read_tick(os.path.join(dump_dir, d))
        extras.append({k: t[k] for k in ("active_set", "lambda", "objective") if k in t})
        rows.append(pack_tick(st, t))
        for k in outs:
            outs[k].append(t[k])
    L = st.field_lengths()
    out = {}
    for k in ("M", "h", "A", "b1", "Ac", "bc", "blb", "bub", "tlb", "tub", "w"):
        out["in_" + k] = np.stack([np.asarray(r[k], np.float64).reshape(L[k]) for r in rows])
    out.update({k: np.asarray(v) for k, v in outs.items()})
    out["status"] = out["status"].astype(np.int32)
    out["iters"] = out["iters"].astype(np.int32)
    if all("objective" in t_ for t_ in extras):
        out["ref_objective"] = np.array([float(t_["objective"][0]) for t_ in extras])
    if all("active_set" in t_ for t_ in extras):
        width = max(t_["active_set"].size for t_ in extras)
        a = np.full((len(extras), max(width, 1)), np.iinfo(np.int32).min, np.int32)
        for i, t_ in enumerate(extras):
            a[i, :t_["active_set"].size] = t_["active_set"]
        out["ref_active_set"] = a
    out["source"] = np.array("reference")
    out["structure"] = np.array(st.name)
    return out


def main():
    if len(sys.argv) != 4:
        raise SystemExit(__doc__)
    st = S.STRUCTURES[sys.argv[2]]()
    out = pack(sys.argv[1], st)
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[3])), exist_ok=True)
    np.savez_compressed(sys.argv[3], **out)
    print("wrote %s: %d ticks of %s, iterations %s" % (sys.argv[3], out["x"].shape[0], st.name, out["iters"].tolist()[:16]))


if __name__ == "__main__":
    main()
