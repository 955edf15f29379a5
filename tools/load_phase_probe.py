#!/usr/bin/env python3
"""Is the solve kernel's load phase HBM latency?  Stamped build, Talos records from the rows kernel: (a) 256 records solved twice
(the second pass finds them in L2 / MALL), (b) 8192 distinct records (287 MB) after 2 GiB of unrelated traffic has gone through the
caches, 512 workgroups at a time.  Prints the mean 'load' stamp of both (cycles from the QP's first instruction to the barrier that
ends phase 0).   python tools/load_phase_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from inria_wbc_amd import capi, structure
    from inria_wbc_amd import model as mdl
    capi.LIB_PATH = os.path.join(ROOT, "inria_wbc_amd", "lib", "libwbcqp_stamps.so")
    lib = capi.load_library(capi.LIB_PATH)
    st = structure.talos_structure()
    m = mdl.talos_like()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    dev = torch.device("cuda", 0)
    L = st.field_lengths()
    sp = torch.cuda.current_stream().cuda_stream
    lib.wbcqp_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]

    def records(B):
        s = mdl.sample_states(m, tm, min(B, 512), 9_000_000, q_noise=0.01, v_noise=0.05, ref_noise=0.01)
        rep = (B + 511) // 512
        hh = capi.Handle(0, capi.F64)
        hh.set_structure(0, st)
        hh.set_model(0, m, tm)
        d_in = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
        state = {k: torch.from_numpy(np.tile(s[k], (rep, 1))[:B].copy()).to(dev) for k in ("q", "v", "ref")}
        hh.problem_data(0, B, state, d_in)
        torch.cuda.synchronize()
        hh.close()
        d_in["tlb"] = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
        d_in["tub"] = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
        d_in["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
        return d_in

    def solve(B, d_in, passes, flush):
        d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
                     status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
        dbg = torch.zeros(B, capi.K_STAMPS, dtype=torch.int64, device=dev)
        h = capi.Handle(0, capi.F64)
        h.set_structure(0, st)
        assert lib.wbcqp_debug_set_stamp_buffer(h._h, C.c_void_p(dbg.data_ptr())) == 0
        for _ in range(passes):
            if flush:
                a = torch.empty(256 * 1024 * 1024, dtype=torch.float64, device=dev)  # 2 GiB written, then read
                a.fill_(1.0)
                float(a.sum().item())
                del a
            h.solve_batch(0, B, d_in, d_out, stream=sp)
            torch.cuda.synchronize()
        t = dbg.cpu().numpy().astype(np.float64)
        h.close()
        return float(t[:, 0].mean()), float(np.median(t[:, 0])), float(t.sum(axis=1).mean())

    warm = solve(256, records(256), 3, False)
    cold = solve(8192, records(8192), 1, True)
    print("load stamp, cycles (mean, median) and whole QP mean:")
    print("  256 records, third pass over the same records (caches warm): %.0f %.0f | %.0f" % warm)
    print("  8192 distinct records (287 MB) after a 2 GiB flush, one pass:  %.0f %.0f | %.0f" % cold)


if __name__ == "__main__":
    main()
