#!/usr/bin/env python3
"""What a third workgroup per CU is worth, measured before anything is rebuilt for it.  The solve kernels allocate up to
256 VGPRs (two waves per SIMD = two workgroups per CU); a third needs <= 168 registers AND <= 54.6 KB of LDS.  This probe takes a stack that
already fits the LDS side -- iCub on ONE foot (n 50, nEq 12: 45.6 KB; the generic compact kernel) -- and runs it through a diagnostic build whose
queue kernels are compiled with __launch_bounds__(256, 3) (-DWBCQP_X_OCC3: 168 VGPRs, the rest spilled to scratch; never the product):

    tools/variants.sh occ3 "-DWBCQP_X_OCC3"
    python tools/occ3_probe.py                                        # product build: 2 per CU, no scratch
    WBCQP_DEBUG_LDS_PAD=12000 python tools/occ3_probe.py --lib inria_wbc_amd/lib/libwbcqp_occ3.so   # the spilled build held at 2 per CU: what the spills cost
    python tools/occ3_probe.py --lib inria_wbc_amd/lib/libwbcqp_occ3.so                            # the spilled build at 3 per CU: what residency returns

Since solve_queue3_kernel is in the product library the probe also compares the product's own two forms on any stack (--stack icub --two: hardware
dispatch of solve_kernel, two per CU; without --two: the queue, three per CU where they fit); tools/occ3_pmc.sh collects the SQ counters of both.

One JSON line per run; `x_sha` must agree between the runs (same arithmetic, other registers)."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--stack", default="icub_single_support", help="icub_single_support (the probe's original stack), icub (config 3's), talos_single_support (does not fit: control)")
    ap.add_argument("--two", action="store_true", help="WBCQP_FLAG_HW_DISPATCH: the two-per-CU kernel (solve_kernel) on the same batch")
    args = ap.parse_args()
    import torch
    from inria_wbc_amd import capi, structure, synth
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    st = structure.STRUCTURES[args.stack]()
    lay = capi.layout_of(st)
    dev = torch.device("cuda", 0)
    B = args.batch
    inputs = synth.generate(st, B, 4242)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    o = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
             status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    sp = torch.cuda.current_stream().cuda_stream
    h = capi.Handle(0, capi.F64, flags=capi.FLAG_HW_DISPATCH if args.two else 0)
    h.set_structure(0, st)
    for _ in range(6):
        h.solve_batch(0, B, d_in, o, stream=sp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        h.solve_batch(0, B, d_in, o, stream=sp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    it = o["iters"].cpu().numpy()
    stt = o["status"].cpu().numpy()
    x = o["x"].cpu().numpy()
    h.close()
    print(json.dumps({"lib": os.path.basename(capi.LIB_PATH), "stack": args.stack, "two_per_cu_kernel": bool(args.two), "n": st.n, "neq": st.neq,
                      "lds_bytes": lay["lds_bytes"], "layout_waves_per_cu": lay["waves_per_cu"], "lds_pad": int(os.environ.get("WBCQP_DEBUG_LDS_PAD", "0")),
                      "batch": B, "qps": B / dt, "ms": dt * 1e3, "iters_mean": float(it.mean()), "iters_max": int(it.max()),
                      "optimal": int((stt == 0).sum()), "x_sha": hashlib.sha256(x.tobytes()).hexdigest()[:16]}))


if __name__ == "__main__":
    main()
