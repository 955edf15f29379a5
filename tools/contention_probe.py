#!/usr/bin/env python3
"""Same QP replicated B times: per-QP cycles (stamped build) and wall time against the number of busy CUs.
Separates what a QP costs alone from what the chip adds when every CU runs one (instruction fetch, HBM, clocks)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from inria_wbc_amd import capi, structure, synth
    capi.LIB_PATH = os.path.join(ROOT, "inria_wbc_amd", "lib", "libwbcqp_stamps.so")
    lib = capi.load_library(capi.LIB_PATH)
    st = structure.talos_structure()
    one = synth.generate(st, 4, synth.SEED_BASE["talos"])
    dev = torch.device("cuda", 0)
    lib.wbcqp_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    for which in (0, 2):
        for B in (1, 8, 32, 64, 128, 256, 512):
            inp = {k: np.repeat(v[which:which + 1], B, axis=0) for k, v in one.items()}
            d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items() if v.size}
            d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
                         status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
            dbg = torch.zeros(B, capi.K_STAMPS, dtype=torch.int64, device=dev)
            h = capi.Handle(0, capi.F64, flags=capi.FLAG_INDEX_ORDER)
            h.set_structure(0, st)
            assert lib.wbcqp_debug_set_stamp_buffer(h._h, C.c_void_p(dbg.data_ptr())) == 0
            for _ in range(3):
                h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            t = dbg.cpu().numpy().astype(np.float64)
            tot = t.sum(axis=1)
            print("QP %d x %4d: iters %d  cycles/QP mean %.0f max %.0f  (load %.0f, cholesky %.0f, QR %.0f)  wall %.3f ms -> %.2f GHz-equivalent"
                  % (which, B, int(d_out["iters"][0]), tot.mean(), tot.max(), t[:, 0].mean(), t[:, 2].mean(), t[:, 6].mean(), ms,
                     tot.max() * ((B + 255) // 256) / (ms * 1e-3) / 1e9))
            h.close()


if __name__ == "__main__":
    main()
