#!/usr/bin/env python3
"""Throughput of one library build in the three regimes the round's targets are stated in: the tick stream at B = 8192
(set-up bound), SURVEY config 2's batch replayed at B = 1024, the tick stream at B = 1024 (straggler bound); and one median QP alone.
One JSON line; --lib takes a variant built by tools/variants.sh.   python tools/throughput_time.py [--lib inria_wbc_amd/lib/libwbcqp_x.so]"""
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
    ap.add_argument("--lib", default=None)
    ap.add_argument("--reps", type=int, default=60)
    args = ap.parse_args()
    import torch
    from inria_wbc_amd import capi, structure, synth
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    st = structure.talos_structure()
    dev = torch.device("cuda", 0)
    B0 = 1024
    inputs = synth.generate(st, B0, synth.SEED_BASE["talos"])
    com_rows = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    table = np.stack([synth.squat_com_rhs(st, t, st.kp.get("com", 30.0)) for t in range(4000)])
    sp = torch.cuda.current_stream().cuda_stream

    def outs(b):
        return dict(x=torch.zeros(b, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(b, st.na, dtype=torch.float64, device=dev),
                    status=torch.zeros(b, dtype=torch.int32, device=dev), iters=torch.zeros(b, dtype=torch.int32, device=dev))

    def run(bsz, stream_ticks, reps):
        rep_in = (bsz + B0 - 1) // B0
        base = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (rep_in, 1))[:bsz])).to(dev) for k, v in inputs.items() if v.size}
        dicts = []
        for t in range(max(1, stream_ticks)):
            d = dict(base)
            if stream_ticks:
                b1 = np.tile(inputs["b1"], (rep_in, 1))[:bsz].copy()
                b1[:, com_rows] += table[(np.arange(bsz) + 17 * t) % 4000][:, :com_rows.size]
                d["b1"] = torch.from_numpy(b1).to(dev)
            dicts.append(d)
        h = capi.Handle(0, capi.F64)
        h.set_structure(0, st)
        o = outs(bsz)
        for t in range(8):
            h.solve_batch(0, bsz, dicts[t % len(dicts)], o, stream=sp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(reps):
            h.solve_batch(0, bsz, dicts[t % len(dicts)], o, stream=sp)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        it = o["iters"].cpu().numpy()
        h.close()
        return {"qps": bsz / dt, "ms": dt * 1e3, "iters_mean": float(it.mean()), "iters_max": int(it.max())}

    res = {"lib": os.path.basename(capi.LIB_PATH)}
    res["stream_b8192"] = run(8192, 8, max(10, args.reps // 3))
    res["config2_replayed_b1024"] = run(1024, 0, args.reps * 2)
    res["stream_b1024"] = run(1024, 64, args.reps * 2)
    # one median QP alone (index order, hardware dispatch, batch 1)
    h = capi.Handle(0, capi.F64, flags=capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH)
    h.set_structure(0, st)
    full = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    o = outs(B0)
    h.solve_batch(0, B0, full, o, stream=sp)
    torch.cuda.synchronize()
    it = o["iters"].cpu().numpy()
    med = int(np.argsort(it)[len(it) // 2])
    one = {k: v[med:med + 1].contiguous() for k, v in full.items()}
    o1 = outs(1)
    for _ in range(10):
        h.solve_batch(0, 1, one, o1, stream=sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        h.solve_batch(0, 1, one, o1, stream=sp)
    e1.record()
    torch.cuda.synchronize()
    res["median_qp_alone"] = {"us": e0.elapsed_time(e1) / 200 * 1e3, "iters": int(it[med])}
    h.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
