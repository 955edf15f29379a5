#!/usr/bin/env python3
"""BASELINE config 3's stack (iCub, f32 boundary) over batch sizes around B = 4096, at three workgroups per CU (the default: solve_queue3_kernel) and at two
(the same queue with WBCQP_DEBUG_LDS_PAD keeping the third workgroup out: solve_queue_kernel): is the 768-slot residency's quantisation (B = 4096 is 5.33
rounds of 768) worth a per-batch choice between the two?    One line per (B, residency): QP/s, ms per launch, rounds of the residency.

    python tools/residency_sweep.py [--stack icub] [--batches 3072,3840,4096,4608]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(st, inputs, B, per_cu, dtype, reps=40):
    import torch
    from inria_wbc_amd import capi
    if per_cu == 2:
        os.environ["WBCQP_DEBUG_LDS_PAD"] = "4096"  # (read at wbcqp_create: a 57 KB workgroup, two per CU; the queue stays)
    else:
        os.environ.pop("WBCQP_DEBUG_LDS_PAD", None)
    dev = torch.device("cuda", 0)
    tdt = torch.float32 if dtype == capi.F32 else torch.float64
    ndt = np.float32 if dtype == capi.F32 else np.float64
    reps_in = (B + inputs["h"].shape[0] - 1) // inputs["h"].shape[0]
    d_in = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (reps_in, 1))[:B].astype(ndt))).to(dev) for k, v in inputs.items() if v.size}
    o = dict(x=torch.zeros(B, st.n, dtype=tdt, device=dev), tau=torch.zeros(B, st.na, dtype=tdt, device=dev),
             status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    sp = torch.cuda.current_stream().cuda_stream
    h = capi.Handle(0, dtype)
    h.set_structure(0, st)
    for _ in range(8):
        h.solve_batch(0, B, d_in, o, stream=sp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        h.solve_batch(0, B, d_in, o, stream=sp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ok = int((o["status"] == 0).sum().item())
    h.close()
    os.environ.pop("WBCQP_DEBUG_LDS_PAD", None)
    return dt, ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stack", default="icub")
    ap.add_argument("--batches", default="2304,3072,3840,4096,4608,6144,8192")
    ap.add_argument("--f64", action="store_true")
    args = ap.parse_args()
    import torch  # noqa: F401
    from inria_wbc_amd import capi, structure, synth
    st = structure.STRUCTURES[args.stack]()
    inputs = synth.generate(st, 1024, synth.SEED_BASE["icub"])
    dtype = capi.F64 if args.f64 else capi.F32
    print(json.dumps({"stack": args.stack, "layout": capi.layout_of(st), "dtype": "f64" if args.f64 else "f32 boundary"}))
    for B in [int(b) for b in args.batches.split(",")]:
        row = {"batch": B}
        for per_cu in (3, 2):
            dt, ok = run(st, inputs, B, per_cu, dtype)
            row["per_cu_%d" % per_cu] = {"qps": B / dt, "ms": dt * 1e3, "rounds": B / (per_cu * 256.0), "optimal": ok}
        row["three_over_two"] = row["per_cu_3"]["qps"] / row["per_cu_2"]["qps"]
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
