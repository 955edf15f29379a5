"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: see oracle/wbc_oracle.h -- the reference holds no golden vector for this path and
its arithmetic lives in un-vendored, un-pinned third-party libraries; the oracle restates their
published algorithms and is validated by the independent numpy KKT checker below and by scipy.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libwbc_oracle.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class _Structure(C.Structure):
    _fields_ = [
        ("nv", C.c_int), ("na", C.c_int), ("nc", C.c_int),
        ("n_dense", C.c_int), ("n_tasks", C.c_int),
        ("dense_row_task", c_int_p),
        ("n_sel", C.c_int), ("sel_col", c_int_p), ("sel_task", c_int_p),
        ("forcereg_mat", c_double_p), ("forcereg_task", c_int_p),
        ("force_gen", c_double_p),
        ("fric_mat", c_double_p), ("fric_lb", c_double_p), ("fric_ub", c_double_p),
        ("n_bound", C.c_int), ("bound_col", c_int_p),
        ("act_bounds", C.c_int),
        ("n_ineq_blocks", C.c_int), ("ineq_kind", c_int_p), ("ineq_arg", c_int_p),
        ("hessian_reg", C.c_double), ("max_iter", C.c_int),
        ("n_acteq", C.c_int), ("acteq_joint", c_int_p), ("acteq_scale", c_double_p), ("acteq_task", C.c_int), ("cop_task", C.c_int),
    ]


class _Inputs(C.Structure):
    _fields_ = [(k, c_double_p) for k in ("M", "h", "A", "b1", "Ac", "bc", "blb", "bub", "tlb", "tub", "w", "Acop")]


class _Outputs(C.Structure):
    _fields_ = [("x", c_double_p), ("tau", c_double_p), ("lam", c_double_p), ("active", c_int_p),
                ("n_active", C.c_int), ("status", C.c_int), ("iters", C.c_int), ("fval", C.c_double)]


class _BatchInputs(C.Structure):
    _fields_ = [(k, c_double_p) for k in ("M", "h", "A", "b1", "Ac", "bc", "blb", "bub", "tlb", "tub", "w", "Acop")]


class _BatchOutputs(C.Structure):
    _fields_ = [("x", c_double_p), ("tau", c_double_p), ("status", c_int_p), ("iters", c_int_p),
                ("active", c_int_p), ("n_active", c_int_p), ("fval", c_double_p)]


def build(force: bool = False) -> str:
    """Compile oracle/wbc_oracle.c with the committed Makefile (gcc)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("wbc_oracle.c", "wbc_oracle.h", "rbd_oracle.c", "rbd_oracle.h", "Makefile")):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


DEFAULT_CFLAGS = "-O3 -march=x86-64-v3 -ffp-contract=off (reproducibility build: bit-identical across hosts; the checker)"
NATIVE_CFLAGS = "-O3 -march=native (SURVEY 8(d)'s timing build; built on the host that times it, never the checker)"
_NATIVE_PATH = os.path.join(_HERE, "_build", "libwbc_oracle_native.so")
_native = None


def native_lib():
    """The same C files built -O3 -march=native ON THIS HOST (the GPU box's CPU differs from the container's), for
    bench.py's cpu_baseline.native_build only."""
    global _native
    if _native is None:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"], stdout=subprocess.DEVNULL)
        _native = C.CDLL(_NATIVE_PATH)
    return _native


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.wbco_ws_size.restype = C.c_long
        _lib.wbco_tick_ws_size.restype = C.c_long
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


def _ip(a: np.ndarray):
    return a.ctypes.data_as(c_int_p)


class OracleStructure:
    """Keeps the numpy buffers alive behind a wbco_structure."""

    def __init__(self, st):
        self.st = st
        self._keep = []

        def keep(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            self._keep.append(a)
            return a

        B, lb, ub = st.friction()
        kinds = keep([k for k, _ in st.ineq_blocks] or [0], np.int32)
        args = keep([a for _, a in st.ineq_blocks] or [0], np.int32)
        s = _Structure()
        s.nv, s.na, s.nc = st.nv, st.na, st.nc
        s.n_dense, s.n_tasks = st.n_dense, st.n_tasks
        s.dense_row_task = _ip(keep(st.dense_row_task if st.n_dense else [0], np.int32))
        s.n_sel = st.n_sel
        s.sel_col = _ip(keep(st.sel_col if st.n_sel else [0], np.int32))
        s.sel_task = _ip(keep(st.sel_task if st.n_sel else [0], np.int32))
        s.forcereg_mat = _dp(keep(st.forcereg_mat() if st.nc else np.zeros(1), np.float64))
        s.forcereg_task = _ip(keep(st.forcereg_task if st.nc else [0], np.int32))
        s.force_gen = _dp(keep(st.force_gen() if st.nc else np.zeros(1), np.float64))
        s.fric_mat = _dp(keep(B if st.nc else np.zeros(1), np.float64))
        s.fric_lb = _dp(keep(lb if st.nc else np.zeros(1), np.float64))
        s.fric_ub = _dp(keep(ub if st.nc else np.zeros(1), np.float64))
        s.n_bound = st.n_bound
        s.bound_col = _ip(keep(st.bound_col if st.n_bound else [0], np.int32))
        s.act_bounds = int(st.act_bounds)
        s.n_ineq_blocks = len(st.ineq_blocks)
        s.ineq_kind = _ip(kinds)
        s.ineq_arg = _ip(args)
        s.hessian_reg = st.hessian_reg
        s.max_iter = st.max_iter
        s.n_acteq = st.n_acteq
        s.acteq_joint = _ip(keep(st.acteq_joint if st.n_acteq else [0], np.int32))
        s.acteq_scale = _dp(keep(st.acteq_scale if st.n_acteq else [1.0], np.float64))
        s.acteq_task = int(st.acteq_task)
        s.cop_task = int(st.cop_task)
        self.c = s


_FIELDS = ("M", "h", "A", "b1", "Ac", "bc", "blb", "bub", "tlb", "tub", "w", "Acop")


def _prep_inputs(st, inputs: Dict[str, np.ndarray], batch: int):
    lens = st.field_lengths()
    arrs = {}
    for k in _FIELDS:
        if lens[k] == 0 and k not in inputs:  # (records written before a field existed: Acop)
            arrs[k] = np.zeros((batch, 1))
            continue
        a = np.ascontiguousarray(inputs[k], dtype=np.float64).reshape(batch, -1)
        assert a.shape[1] == lens[k], (k, a.shape, lens[k])
        if a.size == 0:
            a = np.zeros((batch, 1))
        arrs[k] = a
    return arrs


def assemble(st, inputs: Dict[str, np.ndarray], index: int = 0):
    """Dense H, g, CE, ce0, CI, ci0 of QP `index` exactly as the reference's solver sees them."""
    ost = OracleStructure(st)
    batch = np.asarray(inputs["h"]).reshape(-1, st.nv).shape[0]
    arrs = _prep_inputs(st, inputs, batch)
    cin = _Inputs(*[_dp(arrs[k][index]) for k in _FIELDS])
    n, neq, nin2 = st.n, st.neq, st.nin2
    H = np.zeros((n, n)); g = np.zeros(n)
    CE = np.zeros((max(neq, 1), n)); ce0 = np.zeros(max(neq, 1))
    CI = np.zeros((max(nin2, 1), n)); ci0 = np.zeros(max(nin2, 1))
    lib().wbco_assemble(C.byref(ost.c), C.byref(cin), _dp(H), _dp(g), _dp(CE), _dp(ce0), _dp(CI), _dp(ci0))
    return H, g, CE[:neq], ce0[:neq], CI[:nin2], ci0[:nin2]


def eiquadprog(H, g, CE, ce0, CI, ci0, max_iter: int = 1000):
    """eiquadprog-fast solve_quadprog on a generic dense QP. Returns dict(x,u,A,iq,iters,fval,status)."""
    H = np.ascontiguousarray(H, np.float64); g = np.ascontiguousarray(g, np.float64)
    n = g.size
    CE = np.ascontiguousarray(CE, np.float64).reshape(-1, n); ce0 = np.ascontiguousarray(ce0, np.float64).reshape(-1)
    CI = np.ascontiguousarray(CI, np.float64).reshape(-1, n); ci0 = np.ascontiguousarray(ci0, np.float64).reshape(-1)
    neq, nin2 = CE.shape[0], CI.shape[0]
    x = np.zeros(n); u = np.zeros(neq + nin2 + 2); A = np.zeros(neq + nin2 + 2, np.int32)
    iq = C.c_int(0); it = C.c_int(0); fv = C.c_double(0.0)
    CEp = CE if neq else np.zeros((1, n)); ce0p = ce0 if neq else np.zeros(1)
    CIp = CI if nin2 else np.zeros((1, n)); ci0p = ci0 if nin2 else np.zeros(1)
    status = lib().wbco_eiquadprog_fast(n, neq, nin2, _dp(H), _dp(g), _dp(CEp), _dp(ce0p), _dp(CIp), _dp(ci0p),
                                        _dp(x), _dp(u), _ip(A), C.byref(iq), C.byref(it), C.byref(fv),
                                        int(max_iter), None)
    return dict(x=x, u=u[:iq.value].copy(), A=A[:iq.value].copy(), iq=iq.value, iters=it.value, fval=fv.value, status=status)


def eiquadprog_timed(H, g, CE, ce0, CI, ci0, reps: int = 200, max_iter: int = 1000, native: bool = False):
    """(seconds per solve, status, iterations): `reps` solves of one dense QP inside ONE C loop after a warm-up solve
    (wbco_eiquadprog_timed) -- no Python between two solves."""
    H = np.ascontiguousarray(H, np.float64); g = np.ascontiguousarray(g, np.float64)
    n = g.size
    CE = np.ascontiguousarray(CE, np.float64).reshape(-1, n); ce0 = np.ascontiguousarray(ce0, np.float64).reshape(-1)
    CI = np.ascontiguousarray(CI, np.float64).reshape(-1, n); ci0 = np.ascontiguousarray(ci0, np.float64).reshape(-1)
    neq, nin2 = CE.shape[0], CI.shape[0]
    CEp = CE if neq else np.zeros((1, n)); ce0p = ce0 if neq else np.zeros(1)
    CIp = CI if nin2 else np.zeros((1, n)); ci0p = ci0 if nin2 else np.zeros(1)
    st, it = C.c_int(0), C.c_int(0)
    f = (native_lib() if native else lib()).wbco_eiquadprog_timed
    f.restype = C.c_double
    secs = f(n, neq, nin2, _dp(H), _dp(g), _dp(CEp), _dp(ce0p), _dp(CIp), _dp(ci0p), int(max_iter), int(reps), C.byref(st), C.byref(it))
    if secs < 0.0:
        raise RuntimeError("wbco_eiquadprog_timed failed")
    return float(secs) / max(1, reps), st.value, it.value


ACTIVE_PAD = np.iinfo(np.int32).min  # entries of `active` beyond n_active


def active_to_mask(active: np.ndarray, n_active: np.ndarray, words: int = 8) -> np.ndarray:
    """eiquadprog's active list -> the C ABI's wbcqp_outputs.active_mask ([B, 8] uint32): bit r of the 256-bit mask set iff the
    one-sided CI row r (a tag >= 0) is in A[0 .. n_active); equalities (tags < 0) have no bit, rows beyond the 256th neither."""
    B = active.shape[0]
    m = np.zeros((B, words), np.uint32)
    for i in range(B):
        a = active[i, :n_active[i]]
        a = a[(a >= 0) & (a < 32 * words)]
        np.bitwise_or.at(m[i], a >> 5, (np.uint32(1) << (a & 31).astype(np.uint32)))
    return m


def tick_batch(st, inputs: Dict[str, np.ndarray], nthreads: int = 1):
    """P1..P4 for every QP of a [B, len] input set. Returns dict(x, tau, status, iters, active, n_active, fval, active_mask):
    `active` [B, neq + nin2] is eiquadprog's A (equality i tagged -i-1, else the one-sided CI row), padded with ACTIVE_PAD
    beyond n_active; `active_mask` the same set in the C ABI's form; `fval` the objective."""
    ost = OracleStructure(st)
    batch = np.asarray(inputs["h"]).reshape(-1, st.nv).shape[0]
    arrs = _prep_inputs(st, inputs, batch)
    x = np.zeros((batch, st.n)); tau = np.zeros((batch, max(st.na, 1)))
    status = np.zeros(batch, np.int32); iters = np.zeros(batch, np.int32)
    active = np.full((batch, max(st.neq + st.nin2, 1)), ACTIVE_PAD, np.int32); n_active = np.zeros(batch, np.int32); fval = np.zeros(batch)
    bin_ = _BatchInputs(*[_dp(arrs[k]) for k in _FIELDS])
    bout = _BatchOutputs(_dp(x), _dp(tau), _ip(status), _ip(iters), _ip(active), _ip(n_active), _dp(fval))
    lib().wbco_tick_batch(C.byref(ost.c), int(batch), C.byref(bin_), C.byref(bout), int(nthreads))
    return dict(x=x, tau=tau[:, :st.na], status=status, iters=iters, active=active, n_active=n_active, fval=fval,
                active_mask=active_to_mask(active, n_active))


def tick_batch_timed(st, inputs: Dict[str, np.ndarray], nthreads: int = 1, reps: int = 1, native: bool = False):
    """`reps` passes over the batch on `nthreads` threads; returns (seconds inside the C driver, dict of pass 0's outputs).
    The clock runs from the threads' common start line to the last thread's end (wbco_tick_batch_timed)."""
    ost = OracleStructure(st)
    batch = np.asarray(inputs["h"]).reshape(-1, st.nv).shape[0]
    arrs = _prep_inputs(st, inputs, batch)
    x = np.zeros((batch, st.n)); tau = np.zeros((batch, max(st.na, 1)))
    status = np.zeros(batch, np.int32); iters = np.zeros(batch, np.int32)
    active = np.full((batch, max(st.neq + st.nin2, 1)), ACTIVE_PAD, np.int32); n_active = np.zeros(batch, np.int32); fval = np.zeros(batch)
    bin_ = _BatchInputs(*[_dp(arrs[k]) for k in _FIELDS])
    bout = _BatchOutputs(_dp(x), _dp(tau), _ip(status), _ip(iters), _ip(active), _ip(n_active), _dp(fval))  # (pass 0 writes them: a few stores per QP)
    f = (native_lib() if native else lib()).wbco_tick_batch_timed
    f.restype = C.c_double
    secs = f(C.byref(ost.c), int(batch), C.byref(bin_), C.byref(bout), int(nthreads), int(reps))
    if secs < 0.0:
        raise RuntimeError("wbco_tick_batch_timed failed (threads)")
    return float(secs), dict(x=x, tau=tau[:, :st.na], status=status, iters=iters, active=active, n_active=n_active, fval=fval,
                             active_mask=active_to_mask(active, n_active))


def tick_single(st, inputs: Dict[str, np.ndarray], index: int = 0):
    """One QP with multipliers and active set."""
    ost = OracleStructure(st)
    batch = np.asarray(inputs["h"]).reshape(-1, st.nv).shape[0]
    arrs = _prep_inputs(st, inputs, batch)
    cin = _Inputs(*[_dp(arrs[k][index]) for k in _FIELDS])
    m = st.neq + st.nin2 + 2
    x = np.zeros(st.n); tau = np.zeros(max(st.na, 1)); lam = np.zeros(m); act = np.zeros(m, np.int32)
    out = _Outputs(_dp(x), _dp(tau), _dp(lam), _ip(act), 0, 0, 0, 0.0)
    lib().wbco_tick(C.byref(ost.c), C.byref(cin), C.byref(out), None)
    q = out.n_active
    return dict(x=x, tau=tau[:st.na], lam=lam[:q].copy(), active=act[:q].copy(), status=out.status, iters=out.iters, fval=out.fval)


def integrate(floating_base: bool, dt: float, q: np.ndarray, dq: np.ndarray, dv: np.ndarray):
    """State integration after the path (controller.cpp:250-272) for every row. Returns dict(q_next, v_next, q_solver)."""
    q = np.ascontiguousarray(q, np.float64); dq = np.ascontiguousarray(dq, np.float64); dv = np.ascontiguousarray(dv, np.float64)
    B, nv = dq.shape
    nq = nv + 1 if floating_base else nv
    assert q.shape == (B, nq) and dv.shape == (B, nv)
    qn = np.zeros((B, nq)); vn = np.zeros((B, nv)); qs = np.zeros((B, nq - 1 if floating_base else nq))
    f = lib().wbco_integrate
    f.restype = None
    for i in range(B):
        f(C.c_int(1 if floating_base else 0), C.c_int(nv), C.c_double(dt), _dp(q[i]), _dp(dq[i]), _dp(dv[i]), _dp(qn[i]), _dp(vn[i]), _dp(qs[i]))
    return dict(q_next=qn, v_next=vn, q_solver=qs)


# ---------------------------------------------------------------------------------------------
# Independent checker (numpy only; shares no code with wbc_oracle.c)
# ---------------------------------------------------------------------------------------------
def kkt_residuals(H, g, CE, ce0, CI, ci0, x, active: Optional[np.ndarray] = None, lam: Optional[np.ndarray] = None):
    """KKT residuals of  min 0.5x'Hx+g'x  s.t. CEx+ce0=0, CIx+ci0>=0.
    If (active, lam) in eiquadprog convention are given (equality i tagged -i-1, inequality row index
    otherwise; Lagrangian H x + g - N lam = 0 with N the active normals) they are used; otherwise the
    multipliers are recovered by least squares on the rows active within tolerance."""
    n = x.size
    s = CI @ x + ci0 if CI.size else np.zeros(0)
    eq = CE @ x + ce0 if CE.size else np.zeros(0)
    grad = H @ x + g
    scale = max(1.0, float(np.abs(grad).max(initial=0.0)), float(np.abs(H).max()) * max(1.0, float(np.abs(x).max())))
    if active is not None and lam is not None:
        rows = [CE[-a - 1] if a < 0 else CI[a] for a in active]
        N = np.array(rows).T if rows else np.zeros((n, 0))
        mu = np.asarray(lam)
        ineq_mask = np.array([a >= 0 for a in active], dtype=bool)
        act_slack = np.array([s[a] for a in active if a >= 0])
    else:
        tol = 1e-7 * max(1.0, float(np.abs(ci0).max(initial=1.0)))
        idx = np.where(s <= tol)[0]
        N = np.concatenate([CE.T, CI[idx].T], axis=1) if CE.size or idx.size else np.zeros((n, 0))
        mu = np.linalg.lstsq(N, grad, rcond=None)[0] if N.shape[1] else np.zeros(0)
        ineq_mask = np.concatenate([np.zeros(CE.shape[0], bool), np.ones(idx.size, bool)])
        act_slack = s[idx]
    stat = grad - (N @ mu if N.shape[1] else 0.0)
    return dict(
        stationarity=float(np.abs(stat).max(initial=0.0)) / scale,
        eq=float(np.abs(eq).max(initial=0.0)),
        min_slack=float(s.min(initial=0.0)),
        min_mu=float(mu[ineq_mask].min(initial=0.0)) if mu.size else 0.0,
        compl=float(np.abs(mu[ineq_mask] * act_slack).max(initial=0.0)) if mu.size else 0.0,
        mu=mu,
        scale=scale,
    )
