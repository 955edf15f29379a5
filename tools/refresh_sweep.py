#!/usr/bin/env python3
"""Bench batch through the default path for several renewal periods of the launch order (WBCQP_FLAG_REFRESH): what the two
order kernels cost per launch, and what is left when they run rarely.  Same inputs every step, so staleness costs nothing here --
the figure bounds the overhead side of the trade only."""
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
    B = 1024
    dev = torch.device("cuda", 0)
    inp = synth.generate(st, B, synth.SEED_BASE["talos"])
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
    d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
                 status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    sp = torch.cuda.current_stream().cuda_stream
    for rnd in range(2):
        for n in (1, 2, 4, 8, 16, 64):
            h = capi.Handle(0, capi.F64, flags=capi.flag_refresh(n))
            h.set_structure(0, st)
            for _ in range(40):
                h.solve_batch(0, B, d_in, d_out, stream=sp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(256):
                h.solve_batch(0, B, d_in, d_out, stream=sp)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 256 * 1e6
            h.close()
            if rnd:
                print("renew every %2d launches: %.1f us per step, %.3f M QP/s" % (n, us, B / us))


if __name__ == "__main__":
    main()
