"""ctypes binding of oracle/rbd_oracle.c (TEST INFRASTRUCTURE, NOT PRODUCT CODE; PARITY UNPINNED, see rbd_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np

from .oracle import lib, c_double_p, c_int_p, _dp, _ip


class _Model(C.Structure):
    _fields_ = [("nbody", C.c_int), ("nq", C.c_int), ("nv", C.c_int), ("floating_base", C.c_int),
                ("parent", c_int_p), ("jtype", c_int_p), ("placement", c_double_p), ("inertia", c_double_p),
                ("gravity", C.c_double * 3), ("nframe", C.c_int), ("frame_body", c_int_p), ("frame_placement", c_double_p),
                ("na", C.c_int), ("q_lb", c_double_p), ("q_ub", c_double_p), ("dq_max", c_double_p)]


class _TaskBlock(C.Structure):
    _fields_ = [("kind", C.c_int), ("frame", C.c_int), ("mask", C.c_int), ("kp", C.c_double), ("kd", C.c_double), ("ref", C.c_int),
                ("av_begin", C.c_int), ("av_count", C.c_int), ("radius", C.c_double), ("margin", C.c_double), ("m", C.c_double)]


class _TaskMap(C.Structure):
    _fields_ = [("nblock", C.c_int), ("block", C.POINTER(_TaskBlock)), ("avoided_frame", c_int_p), ("avoided_r0", c_double_p),
                ("n_sel", C.c_int), ("sel_col", c_int_p), ("posture_kp", C.c_double), ("posture_kd", C.c_double),
                ("posture_ref", C.c_int), ("ncontact", C.c_int), ("contact_frame", c_int_p), ("contact_kp", c_double_p),
                ("contact_kd", c_double_p), ("contact_ref", c_int_p), ("n_bound", C.c_int), ("dt", C.c_double), ("nref", C.c_int),
                ("n_acteq", C.c_int), ("cop", C.c_int), ("contact_points", c_double_p), ("cop_ref", C.c_double * 3)]


class _Terms(C.Structure):
    _fields_ = [(k, c_double_p) for k in ("M", "nle", "com", "vcom", "acom", "Jcom", "Ag", "dAgv", "oMf", "vf", "af", "Jl", "Jw")]


class OracleModel:
    def __init__(self, model):
        self.model = model
        self._keep = []

        def keep(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            if a.size == 0:
                a = np.zeros(1, dtype=dtype)
            self._keep.append(a)
            return a

        m = _Model()
        m.nbody, m.nq, m.nv, m.floating_base = model.nbody, model.nq, model.nv, int(model.floating_base)
        m.parent = _ip(keep(model.parent, np.int32))
        m.jtype = _ip(keep(model.jtype, np.int32))
        m.placement = _dp(keep(model.placement, np.float64))
        m.inertia = _dp(keep(model.inertia, np.float64))
        m.gravity[:] = model.gravity
        m.nframe = model.nframe
        m.frame_body = _ip(keep(model.frame_body, np.int32))
        m.frame_placement = _dp(keep(model.frame_placement, np.float64))
        m.na = model.na
        m.q_lb = _dp(keep(model.q_lb, np.float64))
        m.q_ub = _dp(keep(model.q_ub, np.float64))
        m.dq_max = _dp(keep(model.dq_max, np.float64))
        self.c = m


class OracleTaskMap:
    def __init__(self, tm):
        self.tm = tm
        self._keep = []

        def keep(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            if a.size == 0:
                a = np.zeros(1, dtype=dtype)
            self._keep.append(a)
            return a

        blocks = (_TaskBlock * max(1, len(tm.blocks)))()
        av_f, av_r = [], []
        for i, b in enumerate(tm.blocks):
            blocks[i].kind, blocks[i].frame, blocks[i].mask = b.kind, b.frame, b.mask
            blocks[i].kp, blocks[i].kd, blocks[i].ref = b.kp, b.kd, b.ref
            blocks[i].av_begin, blocks[i].av_count = len(av_f), len(b.avoided)
            blocks[i].radius, blocks[i].margin, blocks[i].m = b.radius, b.margin, b.m
            for f, r in b.avoided:
                av_f.append(f)
                av_r.append(r)
        self._blocks = blocks
        t = _TaskMap()
        t.nblock = len(tm.blocks)
        t.block = C.cast(blocks, C.POINTER(_TaskBlock))
        t.avoided_frame = _ip(keep(av_f, np.int32))
        t.avoided_r0 = _dp(keep(av_r, np.float64))
        t.n_sel = int(tm.sel_col.size)
        t.sel_col = _ip(keep(tm.sel_col, np.int32))
        t.posture_kp, t.posture_kd, t.posture_ref = tm.posture_kp, tm.posture_kd, tm.posture_ref
        t.ncontact = tm.ncontact
        t.contact_frame = _ip(keep(tm.contact_frame, np.int32))
        t.contact_kp = _dp(keep(tm.contact_kp, np.float64))
        t.contact_kd = _dp(keep(tm.contact_kd, np.float64))
        t.contact_ref = _ip(keep(tm.contact_ref, np.int32))
        t.n_bound, t.dt, t.nref = tm.n_bound, tm.dt, tm.nref
        t.n_acteq, t.cop = int(getattr(tm, "n_acteq", 0)), int(getattr(tm, "cop", False))
        t.contact_points = _dp(keep(getattr(tm, "contact_points", np.zeros(0)), np.float64))
        t.cop_ref[:] = [0.0, 0.0, 0.0]
        self.c = t


def rbd_terms(model, q: np.ndarray, v: np.ndarray) -> Dict[str, np.ndarray]:
    om = model if isinstance(model, OracleModel) else OracleModel(model)
    md = om.model
    nv, nf = md.nv, md.nframe
    out = dict(M=np.zeros((nv, nv)), nle=np.zeros(nv), com=np.zeros(3), vcom=np.zeros(3), acom=np.zeros(3), Jcom=np.zeros((3, nv)),
               Ag=np.zeros((6, nv)), dAgv=np.zeros(6), oMf=np.zeros((nf, 12)), vf=np.zeros((nf, 6)), af=np.zeros((nf, 6)),
               Jl=np.zeros((nf, 6, nv)), Jw=np.zeros((nf, 6, nv)))
    t = _Terms()
    for k, a in out.items():
        setattr(t, k, _dp(a))
    q = np.ascontiguousarray(q, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    lib().wbco_rbd_terms(C.byref(om.c), _dp(q), _dp(v), C.byref(t))
    return out


def rnea(model, q, v, a) -> np.ndarray:
    om = model if isinstance(model, OracleModel) else OracleModel(model)
    q, v, a = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v, a))
    tau = np.zeros(om.model.nv)
    lib().wbco_rnea(C.byref(om.c), _dp(q), _dp(v), _dp(a), _dp(tau))
    return tau


def energy(model, q, v) -> float:
    om = model if isinstance(model, OracleModel) else OracleModel(model)
    q, v = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v))
    f = lib().wbco_energy
    f.restype = C.c_double
    return float(f(C.byref(om.c), _dp(q), _dp(v)))


def log3(R: np.ndarray) -> np.ndarray:
    R = np.ascontiguousarray(R, dtype=np.float64)
    w = np.zeros(3)
    lib().wbco_log3(_dp(R), _dp(w))
    return w


def task_rows(model, tm, st, q: np.ndarray, v: np.ndarray, ref: np.ndarray, n_threads: int = 1) -> Dict[str, np.ndarray]:
    """QP record fields (M h A b1 Ac bc blb bub Acop), [batch, len] each, for the states q / v and references ref."""
    om, ot = OracleModel(model), OracleTaskMap(tm)
    B = q.shape[0]
    L = dict(st.field_lengths())
    L.setdefault("Acop", 0)  # (a caller's own stand-in for the structure may know nothing of the cop task)
    out = {k: np.zeros((B, max(L[k], 1))) for k in ("M", "h", "A", "b1", "Ac", "bc", "blb", "bub", "Acop")}
    q, v, ref = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v, ref))
    lib().wbco_task_rows_batch(C.byref(om.c), C.byref(ot.c), B, n_threads, st.n_dense, _dp(q), _dp(v), _dp(ref),
                               *[_dp(out[k]) for k in ("M", "h", "A", "b1", "Ac", "bc", "blb", "bub", "Acop")])
    return {k: a[:, :L[k]] for k, a in out.items()}
