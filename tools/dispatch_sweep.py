#!/usr/bin/env python3
"""Hardware dispatcher or queue of resident workgroups, per batch size and workload, on the compact layout (two QPs per CU).
Workloads: `plain` = SURVEY config 2's generator replayed; `squat` = the tick stream bench.py times (heavier tail).
Prints one JSON line per (workload, batch) with QP/s per dispatch mode."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from inria_wbc_amd import capi, structure, synth
    robot = sys.argv[1] if len(sys.argv) > 1 else "talos"
    st = structure.STRUCTURES[robot]()
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    base = synth.generate(st, 1024, synth.SEED_BASE[robot])
    kp = st.kp.get("com", 30.0)
    table = np.stack([synth.squat_com_rhs(st, t, kp) for t in range(4000)])
    com_rows = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    modes = (("hw", capi.FLAG_HW_DISPATCH), ("queue_packed", capi.FLAG_QUEUE), ("queue_lpt", capi.FLAG_QUEUE | capi.FLAG_NO_PACKING),
             ("hw_index", capi.FLAG_HW_DISPATCH | capi.FLAG_INDEX_ORDER), ("full_lds_queue", capi.FLAG_FULL_LDS))
    for workload in ("plain", "squat"):
        for B in (512, 1024, 2048, 4096, 8192):
            reps = (B + 1023) // 1024
            d_in = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (reps, 1))[:B])).to(dev) for k, v in base.items() if v.size}
            ticks = [d_in]
            if workload == "squat":
                ticks = []
                for t in range(8):
                    b1 = np.tile(base["b1"], (reps, 1))[:B].copy()
                    b1[:, com_rows] += table[(np.arange(B) + t) % 4000][:, :com_rows.size]
                    d = dict(d_in)
                    d["b1"] = torch.from_numpy(b1).to(dev)
                    ticks.append(d)
            d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, max(st.na, 1), dtype=torch.float64, device=dev),
                         status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
            row = {"robot": robot, "workload": workload, "batch": B}
            for name, flags in modes:
                h = capi.Handle(0, capi.F64, flags=flags)
                h.set_structure(0, st)
                for t in range(8):
                    h.solve_batch(0, B, ticks[t % len(ticks)], d_out, stream=sp)
                torch.cuda.synchronize()
                n = 60 if B <= 2048 else 30
                t0 = time.perf_counter()
                for t in range(n):
                    h.solve_batch(0, B, ticks[t % len(ticks)], d_out, stream=sp)
                torch.cuda.synchronize()
                row[name] = round(B * n / (time.perf_counter() - t0))
                h.close()
            it = d_out["iters"].cpu().numpy()
            row["iters_mean"] = round(float(it.mean()), 2)
            row["iters_max"] = int(it.max())
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
