#!/usr/bin/env python3
"""Per-phase cycle shares of the rows kernel (wbcqp::terms_kernel) from the -DWBCQP_STAMPS diagnostic build.
Usage (GPU box): python tools/terms_profile.py [--batch 1024]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["load state", "joint transform", "tree sweep", "inertia+scan", "task laws", "sc pairs", "posture/com/mom rhs", "S,F,h",
         "M rows", "Jacobian rows", "com/mom/sc rows", "bounds+b1 store"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--time", default=None, metavar="LIB", help="no stamps: microseconds per launch of the rows kernel of this library (product build)")
    args = ap.parse_args()
    import torch
    from inria_wbc_amd import capi, structure
    from inria_wbc_amd import model as mdl
    capi.LIB_PATH = os.path.abspath(args.time) if args.time else os.path.join(ROOT, "inria_wbc_amd", "lib", "libwbcqp_stamps.so")
    lib = capi.load_library(capi.LIB_PATH)
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    B = args.batch
    h = capi.Handle(0, capi.F64)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    dev = torch.device("cuda", 0)
    s = mdl.sample_states(m, tm, B, 9_000_000, q_noise=0.01, v_noise=0.05, ref_noise=0.01)
    L = st.field_lengths()
    state = {k: torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref")}
    rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
    if args.time:
        sp = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            h.problem_data(0, B, state, rows, stream=sp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            h.problem_data(0, B, state, rows, stream=sp)
        e1.record()
        torch.cuda.synchronize()
        print("%s: batch %d, %.2f us per launch (200 back to back)" % (os.path.basename(capi.LIB_PATH), B, e0.elapsed_time(e1) / 200 * 1e3))
        h.close()
        return
    dbg = torch.zeros(B, 24, dtype=torch.int64, device=dev)
    lib.wbcqp_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.wbcqp_debug_set_stamp_buffer(h._h, C.c_void_p(dbg.data_ptr())) == 0
    for _ in range(3):
        h.problem_data(0, B, state, rows, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().astype(np.float64)[:, :12]
    mean = t.mean(axis=0)
    print("batch %d: mean cycles per instance %.0f" % (B, mean.sum()))
    for i, nm in enumerate(NAMES):
        print("  %-22s %9.0f  %5.1f %%" % (nm, mean[i], 100 * mean[i] / mean.sum()))
    h.close()


if __name__ == "__main__":
    main()
