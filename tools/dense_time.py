#!/usr/bin/env python3
"""Launch time of wbcqp::solve_dense_kernel alone (the narrow seam's kernel, csrc/wbcqp_dense.hpp), device pointers, no PCIe: one Talos
QP (batch 1) and 256 of them, 200 back-to-back launches each.  The command tools/profile_kernels.sh puts under rocprofv3 --kernel-trace;
the dense QPs come from the oracle's assembly of the headline batch (the checker producing an INPUT, as in bench.py's dense_seam)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="*", default=[1, 256], help="batch sizes to time (one per rocprofv3 trace keeps their rows apart)")
    args = ap.parse_args()
    import torch
    from inria_wbc_amd import capi, structure, synth
    from oracle import oracle
    st = structure.talos_structure()
    nq = 256
    inputs = synth.generate(st, nq, synth.SEED_BASE["talos"])
    mats = [oracle.assemble(st, inputs, i) for i in range(nq)]
    dev = torch.device("cuda", 0)
    names = ("H", "g", "CE", "ce0", "CI", "ci0")
    full = {k: torch.from_numpy(np.ascontiguousarray(np.stack([m[j] for m in mats]))).to(dev) for j, k in enumerate(names)}
    h = capi.Handle(0, capi.F64)
    sp = torch.cuda.current_stream().cuda_stream
    for B in args.batch:
        din = {k: v[:B].contiguous() for k, v in full.items()}
        out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), status=torch.full((B,), -99, dtype=torch.int32, device=dev),
                   iters=torch.zeros(B, dtype=torch.int32, device=dev))
        for _ in range(10):
            h.solve_dense(st.n, st.neq, st.nin2, din, out, stream=sp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            h.solve_dense(st.n, st.neq, st.nin2, din, out, stream=sp)
        e1.record()
        torch.cuda.synchronize()
        it = out["iters"].cpu().numpy()
        print("solve_dense_kernel batch %d: %.2f us per launch (200 back to back), iters mean %.2f max %d, optimal %d"
              % (B, e0.elapsed_time(e1) / 200 * 1e3, it.mean(), it.max(), int((out["status"] == 0).sum().item())))
    h.close()


if __name__ == "__main__":
    main()
