"""ctypes binding of the C ABI in include/wbcqp.h (libwbcqp.so).

This is plumbing: every call goes to the HIP library. There is no Python or CPU implementation of
the solve behind it -- if the library is missing or no gfx950 device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Dict, Optional, Sequence

import numpy as np

from .structure import Structure

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwbcqp.so")

WBCQP_OK = 0
ERR_NAMES = {0: "OK", 1: "INVALID", 2: "HIP", 3: "UNSUPPORTED", 4: "NO_DEVICE", 5: "RCCL"}
F64, F32 = 0, 1
K_STAMPS = 32  # phase stamps per QP of the -DWBCQP_STAMPS diagnostic build (kStamps, csrc/wbcqp_prims.hpp)
FIELDS = ("M", "h", "A", "b1", "Ac", "bc", "blb", "bub", "tlb", "tub", "w", "Acop")

# every symbol include/wbcqp.h declares
EXPORTS = ("wbcqp_version", "wbcqp_last_error", "wbcqp_create", "wbcqp_destroy", "wbcqp_set_structure",
           "wbcqp_layout_of", "wbcqp_solve_batch", "wbcqp_solve_batch_host", "wbcqp_solve_ragged",
           "wbcqp_allgather_tau", "wbcqp_integrate", "wbcqp_integrate_host", "wbcqp_set_model", "wbcqp_check_model", "wbcqp_problem_data",
           "wbcqp_problem_data_host", "wbcqp_tick", "wbcqp_tick_host", "wbcqp_tick_graph_create", "wbcqp_tick_graph_launch", "wbcqp_tick_graph_destroy",
           "wbcqp_sync", "wbcqp_launch_order", "wbcqp_solve_dense", "wbcqp_solve_dense_host", "wbcqp_rollout")
ROW_FIELDS = ("M", "h", "A", "b1", "Ac", "bc", "blb", "bub", "Acop")  # what wbcqp_problem_data writes (Acop: stacks with a cop task)

c_i32_p = C.POINTER(C.c_int32)
c_f64_p = C.POINTER(C.c_double)


class WbcqpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("wbcqp error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


class CStructure(C.Structure):
    _fields_ = [
        ("nv", C.c_int32), ("na", C.c_int32), ("nc", C.c_int32),
        ("n_dense", C.c_int32), ("n_tasks", C.c_int32), ("dense_row_task", c_i32_p),
        ("n_sel", C.c_int32), ("sel_col", c_i32_p), ("sel_task", c_i32_p),
        ("forcereg_mat", c_f64_p), ("forcereg_task", c_i32_p),
        ("force_gen", c_f64_p), ("fric_mat", c_f64_p), ("fric_lb", c_f64_p), ("fric_ub", c_f64_p),
        ("n_bound", C.c_int32), ("bound_col", c_i32_p), ("act_bounds", C.c_int32),
        ("n_ineq_blocks", C.c_int32), ("ineq_kind", c_i32_p), ("ineq_arg", c_i32_p),
        ("hessian_reg", C.c_double), ("max_iter", C.c_int32),
        ("n_acteq", C.c_int32), ("acteq_joint", c_i32_p), ("acteq_scale", c_f64_p), ("acteq_task", C.c_int32), ("cop_task", C.c_int32),
    ]


class CLayout(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n", "neq", "nin", "nin2", "r1", "len_M", "len_h", "len_A", "len_b1", "len_Ac",
                                         "len_bc", "len_blb", "len_bub", "len_tlb", "len_tub", "len_w", "lds_bytes",
                                         "waves_per_cu")] + [("algorithmic_bytes", C.c_int64), ("wave_per_qp", C.c_int32), ("dense_h", C.c_int32), ("len_Acop", C.c_int32), ("specialised", C.c_int32)]


class CInputs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in FIELDS]


class COutputs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("tau", C.c_void_p), ("status", C.c_void_p), ("iters", C.c_void_p),
                ("objective", C.c_void_p), ("n_active", C.c_void_p), ("active_mask", C.c_void_p)]


class CDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("dtype", C.c_int32), ("flags", C.c_int32)]


class CDenseInputs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("H", "g", "CE", "ce0", "CI", "ci0")]


class CDenseOutput(C.Structure):
    _fields_ = [("batch", C.c_int32), ("n", C.c_int32), ("x", C.POINTER(C.c_double)), ("status", C.POINTER(C.c_int32)),
                ("iters", C.POINTER(C.c_int32)), ("objective", C.POINTER(C.c_double)), ("n_active", C.POINTER(C.c_int32))]


class CGroup(C.Structure):
    _fields_ = [("slot", C.c_int32), ("batch", C.c_int32), ("inp", CInputs), ("out", COutputs)]


class CModel(C.Structure):
    _fields_ = [("nbody", C.c_int32), ("floating_base", C.c_int32), ("parent", c_i32_p), ("jtype", c_i32_p), ("placement", c_f64_p),
                ("inertia", c_f64_p), ("gravity", C.c_double * 3), ("nframe", C.c_int32), ("frame_body", c_i32_p),
                ("frame_placement", c_f64_p), ("q_lb", c_f64_p), ("q_ub", c_f64_p), ("dq_max", c_f64_p)]


class CTask(C.Structure):
    _fields_ = [("kind", C.c_int32), ("frame", C.c_int32), ("mask", C.c_int32), ("kp", C.c_double), ("kd", C.c_double), ("ref", C.c_int32),
                ("n_avoided", C.c_int32), ("avoided_frame", c_i32_p), ("avoided_r0", c_f64_p), ("radius", C.c_double),
                ("margin", C.c_double), ("m", C.c_double)]


class CTaskMap(C.Structure):
    _fields_ = [("n_task", C.c_int32), ("task", C.POINTER(CTask)), ("posture_kp", C.c_double), ("posture_kd", C.c_double),
                ("posture_ref", C.c_int32), ("n_contact", C.c_int32), ("contact_frame", c_i32_p), ("contact_kp", c_f64_p),
                ("contact_kd", c_f64_p), ("contact_ref", c_i32_p), ("bounds", C.c_int32), ("dt", C.c_double), ("nref", C.c_int32)]


class CState(C.Structure):
    _fields_ = [("q", C.c_void_p), ("v", C.c_void_p), ("ref", C.c_void_p), ("momentum", C.c_void_p)]


class CRolloutIO(C.Structure):
    _fields_ = [("state", CState), ("tlb", C.c_void_p), ("tub", C.c_void_p), ("w", C.c_void_p), ("out", COutputs), ("q_next", C.c_void_p),
                ("v_next", C.c_void_p), ("q_solver", C.c_void_p), ("dt", C.c_double), ("iters_sum", C.c_void_p), ("ticks_ok", C.c_void_p)]


class CTickIO(C.Structure):
    pass  # fields set below (needs CInputs / COutputs)


CTickIO._fields_ = [("state", CState), ("rows", CInputs), ("out", COutputs), ("q_next", C.c_void_p), ("v_next", C.c_void_p),
                    ("q_solver", C.c_void_p), ("dt", C.c_double)]

_lib = None


def load_library(path: Optional[str] = None):
    """Loads libwbcqp.so -- after torch, when torch is installed: the library links libamdhip64.so.7, and a process in which it comes FIRST gets
    /opt/rocm's HIP runtime while a later `import torch` brings torch's bundled one.  Two runtimes on one GPU work, but the first one then answers
    hipOccupancyMaxActiveBlocksPerMultiprocessor with 1 for every solve kernel (measured, tools/occ_state_probe.py --torch-after; the library overrules
    such an answer and says so on stderr).  This module moves tensors that torch allocated, so torch's runtime is the one to share."""
    global _lib
    if _lib is not None:
        return _lib
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401  (before the CDLL below: one HIP runtime per process)
        except ImportError:
            pass
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s is missing: build it with `python -m inria_wbc_amd.build` (hipcc, gfx950). "
            "There is no fallback implementation." % path)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib.wbcqp_version.restype = C.c_int
    lib.wbcqp_last_error.restype = C.c_char_p
    lib.wbcqp_last_error.argtypes = [C.c_void_p]
    lib.wbcqp_create.argtypes = [C.POINTER(CDesc), C.POINTER(C.c_void_p)]
    lib.wbcqp_destroy.argtypes = [C.c_void_p]
    lib.wbcqp_set_structure.argtypes = [C.c_void_p, C.c_int, C.POINTER(CStructure)]
    lib.wbcqp_layout_of.argtypes = [C.POINTER(CStructure), C.POINTER(CLayout)]
    lib.wbcqp_solve_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CInputs), C.POINTER(COutputs), C.c_void_p]
    lib.wbcqp_solve_batch_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CInputs), C.POINTER(COutputs)]
    lib.wbcqp_solve_ragged.argtypes = [C.c_void_p, C.c_int, C.POINTER(CGroup), C.c_void_p]
    lib.wbcqp_allgather_tau.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.wbcqp_sync.argtypes = [C.c_void_p, C.c_void_p]
    lib.wbcqp_integrate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.wbcqp_set_model.argtypes = [C.c_void_p, C.c_int, C.POINTER(CModel), C.POINTER(CTaskMap)]
    lib.wbcqp_check_model.argtypes = [C.POINTER(CStructure), C.POINTER(CModel), C.POINTER(CTaskMap), C.POINTER(C.c_int32)]
    lib.wbcqp_problem_data.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CState), C.POINTER(CInputs), C.c_void_p]
    lib.wbcqp_problem_data_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CState), C.POINTER(CInputs)]
    lib.wbcqp_tick.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CTickIO), C.c_void_p]
    lib.wbcqp_tick_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CTickIO)]
    lib.wbcqp_tick_graph_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CTickIO), C.POINTER(C.c_void_p)]
    lib.wbcqp_tick_graph_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.wbcqp_tick_graph_destroy.argtypes = [C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


class ModelBuffers:
    """Host-side wbcqp_model + wbcqp_taskmap built from a `model.Model` and a `model.TaskMap`; keeps the arrays alive."""

    def __init__(self, model, tm):
        self._keep = []

        def keep(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            if a.size == 0:
                a = np.zeros(1, dtype=dtype)
            self._keep.append(a)
            return a

        ip = lambda a: keep(a, np.int32).ctypes.data_as(c_i32_p)
        dp = lambda a: keep(a, np.float64).ctypes.data_as(c_f64_p)
        m = CModel()
        m.nbody, m.floating_base = model.nbody, int(model.floating_base)
        m.parent, m.jtype = ip(model.parent), ip(model.jtype)
        m.placement, m.inertia = dp(model.placement), dp(model.inertia)
        m.gravity[:] = model.gravity
        m.nframe = model.nframe
        m.frame_body, m.frame_placement = ip(model.frame_body), dp(model.frame_placement)
        m.q_lb, m.q_ub, m.dq_max = dp(model.q_lb), dp(model.q_ub), dp(model.dq_max)
        tasks = (CTask * max(1, len(tm.blocks)))()
        for i, b in enumerate(tm.blocks):
            t = tasks[i]
            t.kind, t.frame, t.mask, t.kp, t.kd, t.ref = b.kind, b.frame, b.mask, b.kp, b.kd, b.ref
            t.n_avoided = len(b.avoided)
            t.avoided_frame = ip([f for f, _ in b.avoided])
            t.avoided_r0 = dp([r for _, r in b.avoided])
            t.radius, t.margin, t.m = b.radius, b.margin, b.m
        self._tasks = tasks
        k = CTaskMap()
        k.n_task = len(tm.blocks)
        k.task = C.cast(tasks, C.POINTER(CTask))
        k.posture_kp, k.posture_kd, k.posture_ref = tm.posture_kp, tm.posture_kd, tm.posture_ref
        k.n_contact = tm.ncontact
        k.contact_frame, k.contact_kp, k.contact_kd, k.contact_ref = ip(tm.contact_frame), dp(tm.contact_kp), dp(tm.contact_kd), ip(tm.contact_ref)
        k.bounds, k.dt, k.nref = int(tm.n_bound > 0), tm.dt, tm.nref
        self.model, self.taskmap = m, k


class StructureBuffers:
    """Host-side wbcqp_structure built from a `Structure`; keeps the numpy arrays alive."""

    def __init__(self, st: Structure):
        self._keep = []

        def keep(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            if a.size == 0:
                a = np.zeros(1, dtype=dtype)
            self._keep.append(a)
            return a

        def ip(a):
            return keep(a, np.int32).ctypes.data_as(c_i32_p)

        def dp(a):
            return keep(a, np.float64).ctypes.data_as(c_f64_p)

        B, lb, ub = st.friction()
        s = CStructure()
        s.nv, s.na, s.nc = st.nv, st.na, st.nc
        s.n_dense, s.n_tasks = st.n_dense, st.n_tasks
        s.dense_row_task = ip(st.dense_row_task)
        s.n_sel = st.n_sel
        s.sel_col = ip(st.sel_col)
        s.sel_task = ip(st.sel_task)
        s.forcereg_mat = dp(st.forcereg_mat())
        s.forcereg_task = ip(st.forcereg_task)
        s.force_gen = dp(st.force_gen())
        s.fric_mat, s.fric_lb, s.fric_ub = dp(B), dp(lb), dp(ub)
        s.n_bound = st.n_bound
        s.bound_col = ip(st.bound_col)
        s.act_bounds = int(st.act_bounds)
        s.n_ineq_blocks = len(st.ineq_blocks)
        s.ineq_kind = ip([k for k, _ in st.ineq_blocks])
        s.ineq_arg = ip([a for _, a in st.ineq_blocks])
        s.hessian_reg = st.hessian_reg
        s.max_iter = st.max_iter
        s.n_acteq = st.n_acteq
        s.acteq_joint = ip(st.acteq_joint)
        s.acteq_scale = dp(st.acteq_scale)
        s.acteq_task = int(st.acteq_task)
        s.cop_task = int(st.cop_task)
        self.c = s


def layout_of(st: Structure) -> Dict[str, int]:
    """wbcqp_layout_of: sizes, per-QP array lengths and LDS footprint (pure host; no GPU needed)."""
    lib = load_library()
    sb = StructureBuffers(st)
    L = CLayout()
    rc = lib.wbcqp_layout_of(C.byref(sb.c), C.byref(L))
    if rc != WBCQP_OK:
        raise WbcqpError(rc, (lib.wbcqp_last_error(None) or b"").decode())
    return {k: getattr(L, k) for k, _ in CLayout._fields_}


def check_model(st: Structure, model, tm) -> int:
    """wbcqp_check_model: validates (structure, tree, task bindings) on the host -- no GPU needed -- and returns the LDS bytes one
    instance of the rows kernel needs.  Raises WbcqpError with the library's message otherwise."""
    lib = load_library()
    sb, mb = StructureBuffers(st), ModelBuffers(model, tm)
    lds = C.c_int32(0)
    rc = lib.wbcqp_check_model(C.byref(sb.c), C.byref(mb.model), C.byref(mb.taskmap), C.byref(lds))
    if rc != WBCQP_OK:
        raise WbcqpError(rc, (lib.wbcqp_last_error(None) or b"").decode())
    return int(lds.value)


FLAG_INDEX_ORDER = 1  # wbcqp_desc.flags: launch in index order (default: longest-first, see include/wbcqp.h)
FLAG_NO_PACKING = 4   # wbcqp_desc.flags: plain longest-first order for the queue (default: bin-packed order for small launches)
FLAG_QUEUE = 8        # wbcqp_desc.flags: the queue also when several workgroups share a CU (default there: hardware dispatch)
FLAG_FULL_LDS = 16    # wbcqp_desc.flags: keep the one-QP-per-CU LDS layout (default: compact layout, two QPs per CU, where eligible)
FLAG_WARM_START = 64  # wbcqp_desc.flags: OPT-IN pick priority for the rows of outputs["active_mask"] (not eiquadprog's rule; include/wbcqp.h)
FLAG_WORKGROUP_PER_QP = 32  # wbcqp_desc.flags: four waves per QP also for n <= 16 (default there: one wavefront per QP, wbcqp_small.hpp)
FLAG_GENERIC_KERNEL = 128  # wbcqp_desc.flags: the generic compact kernel also for the shipped stacks (default: their own instantiations)
FLAG_HW_DISPATCH = 2  # wbcqp_desc.flags: one workgroup per QP through the hardware dispatcher (default: resident workgroups + queue)


def flag_refresh(n: int) -> int:
    """wbcqp_desc.flags: renew the launch order every n-th launch (WBCQP_FLAG_REFRESH; 0 = default 4)"""
    return (n & 0xff) << 8


class Handle:
    """wbcqp_handle bound to one HIP device."""

    def __init__(self, device: int = 0, dtype: int = F64, flags: int = 0):
        self.lib = load_library()
        self.dtype = dtype
        self.np_dtype = np.float64 if dtype == F64 else np.float32
        self.device = device
        self._h = C.c_void_p()
        self._structs: Dict[int, Structure] = {}
        desc = CDesc(device, dtype, flags)
        rc = self.lib.wbcqp_create(C.byref(desc), C.byref(self._h))
        if rc != WBCQP_OK:
            raise WbcqpError(rc, (self.lib.wbcqp_last_error(None) or b"").decode())

    def _check(self, rc: int):
        if rc != WBCQP_OK:
            raise WbcqpError(rc, (self.lib.wbcqp_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self.lib.wbcqp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_structure(self, slot: int, st: Structure):
        sb = StructureBuffers(st)
        self._check(self.lib.wbcqp_set_structure(self._h, slot, C.byref(sb.c)))
        self._structs[slot] = st

    # ---- device-pointer path (torch tensors are only carriers of device memory) ----
    def _pack(self, slot: int, batch: int, inputs, outputs):
        st = self._structs[slot]
        L = st.field_lengths()
        cin = CInputs()
        for k in FIELDS:
            t = inputs.get(k)
            if L[k] == 0 or t is None:
                setattr(cin, k, None)
                continue
            assert t.is_cuda and t.is_contiguous() and t.numel() == batch * L[k], (k, tuple(t.shape), batch, L[k])
            setattr(cin, k, t.data_ptr())
        cout = COutputs()
        for k in ("x", "tau", "status", "iters", "objective", "n_active", "active_mask"):  # active_mask: int32 [batch, 8], in/out
            t = outputs.get(k)
            setattr(cout, k, t.data_ptr() if t is not None and t.numel() else None)
        return cin, cout

    def solve_batch(self, slot: int, batch: int, inputs: Dict[str, "object"], outputs: Dict[str, "object"], stream: int = 0):
        cin, cout = self._pack(slot, batch, inputs, outputs)
        self._check(self.lib.wbcqp_solve_batch(self._h, slot, batch, C.byref(cin), C.byref(cout), C.c_void_p(stream)))

    def solve_ragged(self, groups: Sequence[tuple], stream: int = 0):
        """groups: sequence of (slot, batch, inputs, outputs) with device tensors."""
        arr = (CGroup * len(groups))()
        for i, (slot, batch, inputs, outputs) in enumerate(groups):
            cin, cout = self._pack(slot, batch, inputs, outputs)
            arr[i].slot, arr[i].batch, arr[i].inp, arr[i].out = slot, batch, cin, cout
        self._check(self.lib.wbcqp_solve_ragged(self._h, len(groups), arr, C.c_void_p(stream)))

    def integrate(self, batch: int, nv: int, floating_base: bool, dt: float, q, dq, x, ldx: int, status, q_next, v_next,
                  q_solver=None, stream: int = 0):
        """State integration after the path (controller.cpp:250-272) on device tensors (anything with .data_ptr())."""
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._check(self.lib.wbcqp_integrate(self._h, batch, nv, 1 if floating_base else 0, float(dt), ptr(q), ptr(dq), ptr(x), ldx,
                                             ptr(status), ptr(q_next), ptr(v_next), ptr(q_solver), C.c_void_p(stream)))

    def set_model(self, slot: int, model, tm):
        """Binds a kinematic tree and its task bindings to a slot that holds the matching structure (wbcqp_set_model)."""
        mb = ModelBuffers(model, tm)
        self._check(self.lib.wbcqp_set_model(self._h, slot, C.byref(mb.model), C.byref(mb.taskmap)))
        self._models = getattr(self, "_models", {})
        self._models[slot] = (model, tm)

    def problem_data(self, slot: int, batch: int, state: Dict[str, "object"], rows: Dict[str, "object"], stream: int = 0):
        """q, v, ref -> M, h, A, b1, Ac, bc, blb, bub on device tensors (wbcqp_problem_data)."""
        st = self._structs[slot]
        L = st.field_lengths()
        cs = CState(state["q"].data_ptr(), state["v"].data_ptr(), state["ref"].data_ptr() if state.get("ref") is not None else None,
                    state["momentum"].data_ptr() if state.get("momentum") is not None else None)  # momentum: [batch, 6] output, optional
        cin = CInputs()
        for k in FIELDS:
            t = rows.get(k)
            if k not in ROW_FIELDS or L[k] == 0 or t is None:
                setattr(cin, k, None)
                continue
            assert t.is_cuda and t.is_contiguous() and t.numel() == batch * L[k], (k, tuple(t.shape), batch, L[k])
            setattr(cin, k, t.data_ptr())
        self._check(self.lib.wbcqp_problem_data(self._h, slot, batch, C.byref(cs), C.byref(cin), C.c_void_p(stream)))

    def problem_data_host(self, slot: int, q: np.ndarray, v: np.ndarray, ref: np.ndarray) -> Dict[str, np.ndarray]:
        st = self._structs[slot]
        L = st.field_lengths()
        q, v, ref = (np.ascontiguousarray(a, dtype=self.np_dtype) for a in (q, v, ref))
        batch = q.shape[0]
        out = {k: np.zeros((batch, max(L[k], 1)), self.np_dtype) for k in ROW_FIELDS}
        mom = np.zeros((batch, 6), self.np_dtype)
        cs = CState(q.ctypes.data, v.ctypes.data, ref.ctypes.data, mom.ctypes.data)
        cin = CInputs()
        for k in FIELDS:
            setattr(cin, k, out[k].ctypes.data if k in ROW_FIELDS and L[k] else None)
        self._check(self.lib.wbcqp_problem_data_host(self._h, slot, batch, C.byref(cs), C.byref(cin)))
        res = {k: a[:, :L[k]] for k, a in out.items()}
        res["momentum"] = mom  # centroidal momentum Ag v: linear, then angular about the CoM (controller.cpp:245 keeps the last three)
        return res

    def _tick_io(self, slot: int, batch: int, state, rows, out, q_next, v_next, dt: float, q_solver=None) -> CTickIO:
        cin, cout = self._pack(slot, batch, rows, out)
        io = CTickIO()
        io.state = CState(state["q"].data_ptr(), state["v"].data_ptr(), state["ref"].data_ptr(),
                          state["momentum"].data_ptr() if state.get("momentum") is not None else None)
        io.rows, io.out = cin, cout
        io.q_next, io.v_next = q_next.data_ptr(), v_next.data_ptr()
        io.q_solver = q_solver.data_ptr() if q_solver is not None else None
        io.dt = float(dt)
        return io

    def tick(self, slot: int, batch: int, state, rows, out, q_next, v_next, dt: float, q_solver=None, stream: int = 0):
        """rows -> QP -> integration for one control tick (wbcqp_tick), device tensors."""
        io = self._tick_io(slot, batch, state, rows, out, q_next, v_next, dt, q_solver)
        self._check(self.lib.wbcqp_tick(self._h, slot, batch, C.byref(io), C.c_void_p(stream)))

    def rollout(self, slot: int, batch: int, n_ticks: int, state, limits, out, q_next, v_next, dt: float, q_solver=None, iters_sum=None,
                ticks_ok=None, stream: int = 0):
        """n_ticks control ticks of every instance in one launch, no batch barrier (wbcqp_rollout).  state: q [B, nq], v [B, nv],
        ref [n_ticks, B, nref] (+ optional momentum [B, 6]); limits: tlb, tub, w; out: x, tau, status, iters of the LAST tick."""
        st = self._structs[slot]
        io = CRolloutIO()
        io.state = CState(state["q"].data_ptr(), state["v"].data_ptr(), state["ref"].data_ptr(),
                          state["momentum"].data_ptr() if state.get("momentum") is not None else None)
        assert state["ref"].is_contiguous() and state["ref"].shape[0] == n_ticks and state["ref"].shape[1] == batch
        io.tlb = limits["tlb"].data_ptr() if limits.get("tlb") is not None and st.act_bounds else None
        io.tub = limits["tub"].data_ptr() if limits.get("tub") is not None and st.act_bounds else None
        io.w = limits["w"].data_ptr()
        cout = COutputs()
        for k in ("x", "tau", "status", "iters", "objective", "n_active", "active_mask"):
            t = out.get(k)
            setattr(cout, k, t.data_ptr() if t is not None and t.numel() else None)
        io.out = cout
        io.q_next, io.v_next = q_next.data_ptr(), v_next.data_ptr()
        io.q_solver = q_solver.data_ptr() if q_solver is not None else None
        io.dt = float(dt)
        io.iters_sum = iters_sum.data_ptr() if iters_sum is not None else None
        io.ticks_ok = ticks_ok.data_ptr() if ticks_ok is not None else None
        self._check(self.lib.wbcqp_rollout(self._h, slot, batch, n_ticks, C.byref(io), C.c_void_p(stream)))

    def tick_host(self, slot: int, q: np.ndarray, v: np.ndarray, ref: np.ndarray, tlb, tub, w, dt: float, want_rows: bool = False):
        """One whole tick with host arrays (wbcqp_tick_host): returns dict(x, tau, status, iters, objective, n_active, active_mask, q_next, v_next,
        q_solver, momentum[, rows])."""
        st = self._structs[slot]
        L = st.field_lengths()
        f = lambda a: np.ascontiguousarray(a, dtype=self.np_dtype)
        q, v, ref, w = f(q), f(v), f(ref), f(w)
        B = q.shape[0]
        tlb, tub = (f(tlb), f(tub)) if L["tlb"] else (None, None)
        out = dict(x=np.zeros((B, st.n), self.np_dtype), tau=np.zeros((B, max(st.na, 1)), self.np_dtype), status=np.full(B, -99, np.int32),
                   iters=np.zeros(B, np.int32), q_next=np.zeros_like(q), v_next=np.zeros_like(v), q_solver=np.zeros_like(v))
        rows = {k: np.zeros((B, max(L[k], 1)), self.np_dtype) for k in ROW_FIELDS} if want_rows else {}
        io = CTickIO()
        out["momentum"] = np.zeros((B, 6), self.np_dtype)
        io.state = CState(q.ctypes.data, v.ctypes.data, ref.ctypes.data, out["momentum"].ctypes.data)
        cin = CInputs()
        for k in FIELDS:
            setattr(cin, k, None)
        for k, a in rows.items():
            if L[k]:
                setattr(cin, k, a.ctypes.data)
        if L["tlb"]:
            cin.tlb, cin.tub = tlb.ctypes.data, tub.ctypes.data
        cin.w = w.ctypes.data
        io.rows = cin
        out.update(objective=np.zeros(B, self.np_dtype), n_active=np.zeros(B, np.int32), active_mask=np.zeros((B, 8), np.uint32))
        io.out = COutputs(out["x"].ctypes.data, out["tau"].ctypes.data, out["status"].ctypes.data, out["iters"].ctypes.data,
                          out["objective"].ctypes.data, out["n_active"].ctypes.data, out["active_mask"].ctypes.data)
        io.q_next, io.v_next, io.q_solver, io.dt = out["q_next"].ctypes.data, out["v_next"].ctypes.data, out["q_solver"].ctypes.data, float(dt)
        self._check(self.lib.wbcqp_tick_host(self._h, slot, B, C.byref(io)))
        out["tau"] = out["tau"][:, :st.na]
        if want_rows:
            out["rows"] = {k: a[:, :L[k]] for k, a in rows.items()}
        return out

    def tick_graph(self, slot: int, batch: int, state, rows, out, q_next, v_next, dt: float, q_solver=None) -> int:
        """Captures the tick into a HIP graph bound to these buffers; returns the graph handle for tick_graph_launch."""
        io = self._tick_io(slot, batch, state, rows, out, q_next, v_next, dt, q_solver)
        g = C.c_void_p()
        self._check(self.lib.wbcqp_tick_graph_create(self._h, slot, batch, C.byref(io), C.byref(g)))
        return g.value

    def tick_graph_launch(self, graph: int, stream: int = 0):
        self._check(self.lib.wbcqp_tick_graph_launch(self._h, C.c_void_p(graph), C.c_void_p(stream)))

    def tick_graph_destroy(self, graph: int):
        self.lib.wbcqp_tick_graph_destroy(self._h, C.c_void_p(graph))

    def launch_order(self):
        """(order, packed): the launch order the next solve of the last launch's shape will use; order is None before any"""
        buf = np.zeros(1 << 16, dtype=np.int32)
        packed = C.c_int32(0)
        n = self.lib.wbcqp_launch_order(self._h, buf.ctypes.data_as(c_i32_p), C.c_int32(buf.size), C.byref(packed))
        if n < 0:
            self._check(n)
        return (buf[:n].copy() if n else None), bool(packed.value)

    def sync(self, stream: int = 0):
        self._check(self.lib.wbcqp_sync(self._h, C.c_void_p(stream)))

    # ---- host-pointer path ----
    def solve_batch_host(self, slot: int, inputs: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        st = self._structs[slot]
        L = st.field_lengths()
        batch = int(np.asarray(inputs["h"]).reshape(-1, st.nv).shape[0])
        keep = {}
        cin = CInputs()
        for k in FIELDS:
            if L[k] == 0:
                setattr(cin, k, None)
                continue
            a = np.ascontiguousarray(inputs[k], dtype=self.np_dtype).reshape(batch, L[k])
            keep[k] = a
            setattr(cin, k, a.ctypes.data)
        out = dict(x=np.zeros((batch, st.n), self.np_dtype), tau=np.zeros((batch, max(st.na, 1)), self.np_dtype),
                   status=np.full(batch, -99, np.int32), iters=np.zeros(batch, np.int32),
                   objective=np.zeros(batch, self.np_dtype), n_active=np.zeros(batch, np.int32),
                   active_mask=np.zeros((batch, 8), np.uint32))  # (in/out: zeros = no warm-start hint; out = the solution's active rows)
        cout = COutputs(*[out[k].ctypes.data for k in ("x", "tau", "status", "iters", "objective", "n_active", "active_mask")])
        self._check(self.lib.wbcqp_solve_batch_host(self._h, slot, batch, C.byref(cin), C.byref(cout)))
        out["tau"] = out["tau"][:, :st.na]
        return out

    def solve_dense_host(self, H, g, CE, ce0, CI, ci0, max_iter: int = 0):
        """The narrow seam (wbcqp_solve_dense_host): dense QPs in eiquadprog's convention, [B, ...] double arrays.  The
        returned arrays are COPIES of the handle-owned HQPOutput (which is valid until the next call)."""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        H, g = f(H), f(g)
        if H.ndim == 2:
            H, g = H[None], g[None]
        B, n = g.shape
        CE = f(CE).reshape(B, -1, n) if CE is not None and np.size(CE) else np.zeros((B, 0, n))
        CI = f(CI).reshape(B, -1, n) if CI is not None and np.size(CI) else np.zeros((B, 0, n))
        neq, nin = CE.shape[1], CI.shape[1]
        ce0 = f(ce0).reshape(B, neq) if neq else np.zeros((B, 0))
        ci0 = f(ci0).reshape(B, nin) if nin else np.zeros((B, 0))
        ptr = lambda a: a.ctypes.data if a.size else None
        din = CDenseInputs(ptr(H), ptr(g), ptr(CE), ptr(ce0), ptr(CI), ptr(ci0))
        res = C.POINTER(CDenseOutput)()
        self.lib.wbcqp_solve_dense_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(CDenseInputs),
                                                    C.POINTER(C.POINTER(CDenseOutput))]
        self._check(self.lib.wbcqp_solve_dense_host(self._h, B, n, neq, nin, int(max_iter), C.byref(din), C.byref(res)))
        o = res.contents
        return dict(x=np.ctypeslib.as_array(o.x, shape=(B, n)).copy(), status=np.ctypeslib.as_array(o.status, shape=(B,)).copy(),
                    iters=np.ctypeslib.as_array(o.iters, shape=(B,)).copy(), objective=np.ctypeslib.as_array(o.objective, shape=(B,)).copy(),
                    n_active=np.ctypeslib.as_array(o.n_active, shape=(B,)).copy())

    def solve_dense(self, n: int, neq: int, nin: int, inputs: Dict[str, "object"], outputs: Dict[str, "object"], max_iter: int = 0,
                    stream: int = 0):
        """Device-pointer form of the narrow seam (wbcqp_solve_dense): inputs H, g, CE, ce0, CI, ci0 and outputs x, status, iters
        (objective, n_active optional) are device tensors of the handle's dtype, [batch, ...]."""
        batch = inputs["g"].shape[0]
        ptr = lambda t: t.data_ptr() if t is not None and t.numel() else None
        din = CDenseInputs(*[ptr(inputs.get(k)) for k in ("H", "g", "CE", "ce0", "CI", "ci0")])
        cout = COutputs()
        for k in ("x", "tau", "status", "iters", "objective", "n_active"):
            setattr(cout, k, ptr(outputs.get(k)))
        self.lib.wbcqp_solve_dense.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(CDenseInputs),
                                               C.POINTER(COutputs), C.c_void_p]
        self._check(self.lib.wbcqp_solve_dense(self._h, int(batch), int(n), int(neq), int(nin), int(max_iter), C.byref(din), C.byref(cout),
                                               C.c_void_p(stream)))

    def allgather_tau(self, comm: int, send_ptr: int, recv_ptr: int, count: int, stream: int = 0):
        self._check(self.lib.wbcqp_allgather_tau(self._h, C.c_void_p(comm), C.c_void_p(send_ptr), C.c_void_p(recv_ptr),
                                                 count, C.c_void_p(stream)))
