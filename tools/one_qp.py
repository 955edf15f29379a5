#!/usr/bin/env python3
"""One Talos QP of bench.py's tick alone on the chip, a few launches: the workload of tools/phase_insts.sh (SQ counters per phase)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--qp", type=int, default=412)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from inria_wbc_amd import capi, structure, synth
    capi.LIB_PATH = os.path.abspath(args.lib)
    st = structure.talos_structure()
    B = 1024
    dev = torch.device("cuda", 0)
    inputs = synth.generate(st, B, synth.SEED_BASE["talos"])
    com_rows = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    table = np.stack([synth.squat_com_rhs(st, t, st.kp.get("com", 30.0)) for t in range(4000)])
    inputs["b1"][:, com_rows] += table[(np.arange(B) + 40) % 4000][:, :com_rows.size]
    i = args.qp
    one = {k: torch.from_numpy(np.ascontiguousarray(v[i:i + 1])).to(dev) for k, v in inputs.items() if v.size}
    o1 = dict(x=torch.zeros(1, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(1, st.na, dtype=torch.float64, device=dev),
              status=torch.zeros(1, dtype=torch.int32, device=dev), iters=torch.zeros(1, dtype=torch.int32, device=dev))
    h = capi.Handle(0, capi.F64, flags=capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH)
    h.set_structure(0, st)
    sp = torch.cuda.current_stream().cuda_stream
    for _ in range(args.reps):
        h.solve_batch(0, 1, one, o1, stream=sp)
    torch.cuda.synchronize()
    print("iters", int(o1["iters"][0]))
    h.close()


if __name__ == "__main__":
    main()
