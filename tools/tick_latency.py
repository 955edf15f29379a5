#!/usr/bin/env python3
"""Latency of one whole control tick on the device (rows -> QP -> integration) at small batch, as three API calls from
Python, as one wbcqp_tick call, and as one HIP-graph launch.  Usage (GPU box): python tools/tick_latency.py [--out file]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    import torch
    from inria_wbc_amd import capi, structure
    from inria_wbc_amd import model as mdl
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    L = st.field_lengths()
    res = []
    for B in (1, 4, 16, 64, 256, 1024, 4096, 8192):
        h = capi.Handle(0, capi.F64)
        h.set_structure(0, st)
        h.set_model(0, m, tm)
        s = mdl.sample_states(m, tm, B, 9_000_000, q_noise=0.01, v_noise=0.05, ref_noise=0.01)
        state = {k: torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref")}
        rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
        rows["tlb"] = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
        rows["tub"] = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
        rows["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
        out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
                   status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
        qn, vn = torch.zeros_like(state["q"]), torch.zeros_like(state["v"])
        g = h.tick_graph(0, B, state, rows, out, qn, vn, tm.dt)

        def run(fn, sync_each):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
                if sync_each:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.reps * 1e6

        def calls():
            h.problem_data(0, B, state, rows, stream=stream)
            h.solve_batch(0, B, rows, out, stream=stream)
            h.integrate(B, st.nv, True, tm.dt, state["q"], state["v"], out["x"], st.n, out["status"], qn, vn, None, stream=stream)

        row = {"batch": B, "iters_mean": None}
        # latency: launch, wait for the result (what a 1 kHz control loop sees); throughput: back-to-back launches
        row["three_calls_us"] = run(calls, True)
        row["tick_us"] = run(lambda: h.tick(0, B, state, rows, out, qn, vn, tm.dt, stream=stream), True)
        row["graph_us"] = run(lambda: h.tick_graph_launch(g, stream=stream), True)
        row["graph_back_to_back_us"] = run(lambda: h.tick_graph_launch(g, stream=stream), False)
        row["tick_back_to_back_us"] = run(lambda: h.tick(0, B, state, rows, out, qn, vn, tm.dt, stream=stream), False)
        row["ticks_per_s_back_to_back"] = B / (row["tick_back_to_back_us"] * 1e-6)
        row["iters_mean"] = float(out["iters"].float().mean().item())
        row["status_optimal"] = int((out["status"] == 0).sum().item())
        print(json.dumps(row))
        res.append(row)
        h.tick_graph_destroy(g)
        h.close()
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"workload": "talos_like whole tick (rows + solve + schedule + integrate), launch-to-result wall time per tick",
                       "rows": res}, f, indent=1)


if __name__ == "__main__":
    main()
