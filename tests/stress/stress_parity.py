#!/usr/bin/env python3
"""One-off larger parity sweep on the GPU box: every structure, easy and hard active sets, thousands of QPs against the
CPU oracle (x, tau to TOL, status and iteration counts).      python tests/stress/stress_parity.py [batch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from inria_wbc_amd import capi, structure, synth
    from oracle import oracle
    oracle.build()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = torch.device("cuda", 0)
    worst = 0.0
    for name in ("talos", "icub", "talos_single_support", "three_contact", "tiago", "franka"):
        st = structure.STRUCTURES[name]()
        for noise in (0.5, 5.0):
            inputs = synth.generate(st, B, synth.SEED_BASE[name] + 31337, task_noise=noise)
            t0 = time.time()
            ref = oracle.tick_batch(st, inputs, nthreads=os.cpu_count())
            d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
            d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, max(st.na, 1), dtype=torch.float64, device=dev),
                         status=torch.full((B,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
            h = capi.Handle(0, capi.F64)
            h.set_structure(0, st)
            for _ in range(2):  # second launch runs longest-first
                h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            x, tau = d_out["x"].cpu().numpy(), d_out["tau"].cpu().numpy()[:, :st.na]
            stt, it = d_out["status"].cpu().numpy(), d_out["iters"].cpu().numpy()
            ok = ref["status"] == 0
            xs = np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
            rel_all = np.abs(x - ref["x"]).max(axis=1) / xs
            relx = float(rel_all[ok].max())
            same = ok & (it == ref["iters"])
            rel_same = float(rel_all[same].max()) if same.any() else 0.0
            rel_diff = float(rel_all[ok & ~same].max()) if (ok & ~same).any() else 0.0
            dtau = float(np.abs(tau - ref["tau"])[ok].max()) if st.na else 0.0
            worst = max(worst, rel_same)
            if relx > 1e-8:
                q = int(np.argmax(np.where(ok, rel_all, 0.0)))
                comp = np.abs(x[q] - ref["x"][q])
                wr = np.zeros(0)
                if st.nc:
                    T = np.asarray(st.force_gen()).reshape(st.nc, 6, 12)
                    wr = np.concatenate([T[c] @ (x[q, st.nv + 12 * c: st.nv + 12 * c + 12] - ref["x"][q, st.nv + 12 * c: st.nv + 12 * c + 12]) for c in range(st.nc)])
                print("    worst QP %d: iters gpu %d oracle %d; max |d dv| %.2e, max |d f| %.2e (|f| max %.2e), max |d wrench| %.2e, max |dtau| %.2e" %
                      (q, it[q], ref["iters"][q], comp[:st.nv].max(), comp[st.nv:].max() if st.nc else 0.0,
                       np.abs(ref["x"][q, st.nv:]).max() if st.nc else 0.0, np.abs(wr).max() if wr.size else 0.0,
                       np.abs(tau[q] - ref["tau"][q]).max() if st.na else 0.0))
            print("%-22s noise %.1f: status equal %s, iters equal %.4f (mean %.1f, max %d), max rel dx %.2e (same path %.2e, other path %.2e), max |dtau| %.2e, not optimal %d  [%.1fs]" %
                  (name, noise, bool(np.array_equal(stt, ref["status"])), float((it == ref["iters"]).mean()), it.mean(), it.max(), relx, rel_same, rel_diff, dtau,
                   int((~ok).sum()), time.time() - t0), flush=True)
            h.close()
    print("worst max rel dx among QPs with the oracle's iteration count:", worst)


if __name__ == "__main__":
    main()
