#!/usr/bin/env python3
"""Queue against hardware dispatch on the solve kernel: the same QP replicated B times (what one hand-over costs), and the
bench batch (what the better balance buys).  Wall time per launch, product library."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from inria_wbc_amd import capi, structure, synth
    st = structure.talos_structure()
    dev = torch.device("cuda", 0)
    one = synth.generate(st, 4, synth.SEED_BASE["talos"])
    mixed = synth.generate(st, 1024, synth.SEED_BASE["talos"])  # the bench batch

    def run(inp, B, flags, reps=30):
        d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
        d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
                     status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
        h = capi.Handle(0, capi.F64, flags=flags)
        h.set_structure(0, st)
        for _ in range(3):
            h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        h.close()
        return us, int(d_out["iters"][0])

    for B in (256, 512, 1024, 2048):
        inp = {k: np.repeat(v[0:1], B, axis=0) for k, v in one.items()}
        q, it = run(inp, B, 0)
        hw, _ = run(inp, B, capi.FLAG_HW_DISPATCH)
        print("identical QPs (iters %d) x %4d: queue %.1f us, hardware dispatch %.1f us -> %.2f us per hand-over" %
              (it, B, q, hw, (q - hw) / max(1, B // 256)))
    for B in (512, 1024):
        inp = {k: v[:B] for k, v in mixed.items()}
        for name, fl in (("longest-first", 0), ("index order", capi.FLAG_INDEX_ORDER)):
            q, _ = run(inp, B, fl)
            hw, _ = run(inp, B, fl | capi.FLAG_HW_DISPATCH)
            print("bench batch x %4d, %s: queue %.1f us, hardware dispatch %.1f us" % (B, name, q, hw))


if __name__ == "__main__":
    main()
