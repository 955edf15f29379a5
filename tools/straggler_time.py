#!/usr/bin/env python3
"""Wall time of single QPs ALONE on the chip, product build (no stamps): the latency chain of the longest QPs of bench.py's
tick stream, which bounds a 1024-QP launch.  Prints, for the QPs with the most active-set iterations of a tick (and a median one),
microseconds per solve at batch 1, and the slope per iteration fitted over a spread of QPs.

    python tools/straggler_time.py [--tick 40] [--reps 30]
"""
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
    ap.add_argument("--tick", type=int, default=40)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--out", default=None)
    ap.add_argument("--stamps", action="store_true", help="also run the longest QP alone through the -DWBCQP_STAMPS build and print its phase cycles\n"
                    "(a second process state is not needed: the stamped library is loaded INSTEAD of the product one)")
    ap.add_argument("--queue", action="store_true", help="the QPs alone through the queue kernel (solve_queue_kernel, one workgroup) instead of solve_kernel")
    ap.add_argument("--lib", default=None, help="a variant library (tools/variants.sh) instead of inria_wbc_amd/lib/libwbcqp.so")
    args = ap.parse_args()
    import torch
    from inria_wbc_amd import capi, structure, synth
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    if args.stamps:
        capi.LIB_PATH = os.path.join(ROOT, "inria_wbc_amd", "lib", "libwbcqp_stamps.so")
    st = structure.talos_structure()
    B = args.batch
    dev = torch.device("cuda", 0)
    inputs = synth.generate(st, B, synth.SEED_BASE["talos"])
    com_rows = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    table = np.stack([synth.squat_com_rhs(st, t, st.kp.get("com", 30.0)) for t in range(4000)])
    inputs["b1"][:, com_rows] += table[(np.arange(B) + args.tick) % 4000][:, :com_rows.size]
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
               status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    h = capi.Handle(0, capi.F64, flags=capi.FLAG_INDEX_ORDER | (capi.FLAG_QUEUE if args.queue else capi.FLAG_HW_DISPATCH))
    h.set_structure(0, st)
    sp = torch.cuda.current_stream().cuda_stream
    h.solve_batch(0, B, d_in, out, stream=sp)
    torch.cuda.synchronize()
    iters = out["iters"].cpu().numpy()
    order = np.argsort(-iters)
    picks = list(order[:4]) + [order[B // 2]] + list(order[np.linspace(8, B - 1, 24).astype(int)])
    o1 = dict(x=torch.zeros(1, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(1, st.na, dtype=torch.float64, device=dev),
              status=torch.zeros(1, dtype=torch.int32, device=dev), iters=torch.zeros(1, dtype=torch.int32, device=dev))
    rows = []
    for i in picks:
        one = {k: v[i:i + 1].contiguous() for k, v in d_in.items()}
        for _ in range(3):
            h.solve_batch(0, 1, one, o1, stream=sp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            h.solve_batch(0, 1, one, o1, stream=sp)
        e1.record()
        torch.cuda.synchronize()
        rows.append((int(i), int(iters[i]), e0.elapsed_time(e1) / args.reps * 1e3))
    its = np.array([r[1] for r in rows], float)
    us = np.array([r[2] for r in rows])
    fit = np.linalg.lstsq(np.stack([np.ones_like(its), its], 1), us, rcond=None)[0]
    res = {"lib": os.path.basename(capi.LIB_PATH), "tick": args.tick, "iters_max": int(iters.max()), "iters_mean": float(iters.mean()),
           "longest": [{"qp": r[0], "iters": r[1], "us_alone": round(r[2], 2)} for r in rows[:4]],
           "median": {"qp": rows[4][0], "iters": rows[4][1], "us_alone": round(rows[4][2], 2)},
           "fit_us": {"setup_plus_launch": round(float(fit[0]), 2), "per_iteration": round(float(fit[1]), 3)},
           "note": "batch-1 launches back to back on one stream: launch overhead included (about 5 us)"}
    if args.stamps:
        import ctypes as C
        from tools.phase_profile import NAMES
        lib = capi.load_library()
        lib.wbcqp_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
        dbg = torch.zeros(1, capi.K_STAMPS, dtype=torch.int64, device=dev)
        assert lib.wbcqp_debug_set_stamp_buffer(h._h, C.c_void_p(dbg.data_ptr())) == 0
        for r in (rows[0], rows[4]):
            one = {k: v[r[0]:r[0] + 1].contiguous() for k, v in d_in.items()}
            for _ in range(3):
                h.solve_batch(0, 1, one, o1, stream=sp)
            torch.cuda.synchronize()
            t = dbg.cpu().numpy()[0]
            print("QP %d alone, %d iterations, %d stamped cycles:" % (r[0], r[1], t.sum()))
            for i, nm in enumerate(NAMES):
                if t[i] > 0:
                    print("   %-46s %9d  %5.1f %%   %7.0f per iteration" % (nm, t[i], 100.0 * t[i] / t.sum(), t[i] / max(1, r[1])))
    print(json.dumps(res))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(res, fh, indent=1)
    h.close()


if __name__ == "__main__":
    main()
