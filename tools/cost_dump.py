#!/usr/bin/env python3
"""Per-QP cycles of the bench batch (stamped build, index order) next to the iteration counts: how well does
setup + iterations x constant predict a QP's cost?  Writes gpurun_out/cost_dump.npz."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from inria_wbc_amd import capi, structure, synth
    capi.LIB_PATH = os.path.join(ROOT, "inria_wbc_amd", "lib", "libwbcqp_stamps.so")
    lib = capi.load_library(capi.LIB_PATH)
    lib.wbcqp_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    st = structure.talos_structure()
    B = 1024
    inp = synth.generate(st, B, synth.SEED_BASE["talos"])
    dev = torch.device("cuda", 0)
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
    d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
                 status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    dbg = torch.zeros(B, capi.K_STAMPS, dtype=torch.int64, device=dev)
    h = capi.Handle(0, capi.F64, flags=capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH)
    h.set_structure(0, st)
    assert lib.wbcqp_debug_set_stamp_buffer(h._h, C.c_void_p(dbg.data_ptr())) == 0
    runs = []
    for _ in range(3):
        h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        runs.append(dbg.cpu().numpy().sum(axis=1))
    it = d_out["iters"].cpu().numpy()
    tot = np.array(runs, dtype=np.float64)
    A = np.stack([np.ones(B), it], axis=1)
    coef, *_ = np.linalg.lstsq(A, tot[-1], rcond=None)
    res = tot[-1] - A @ coef
    print("fit: %.0f + %.0f x iters cycles; residual std %.0f, max |res| %.0f; run-to-run std %.0f" %
          (coef[0], coef[1], res.std(), np.abs(res).max(), (tot[-1] - tot[-2]).std()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "cost_dump.npz"), iters=it, cycles=tot)
    h.close()


if __name__ == "__main__":
    main()
