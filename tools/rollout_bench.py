#!/usr/bin/env python3
"""wbcqp_rollout against wbcqp_tick on the same closed loop: B Talos-like robots, K ticks of the squat reference, instance i
`--phase` i ticks ahead on the reference (the heavy ticks of the squat then visit the instances one after the other instead of
all at once).  Prints ticks/s of both and the ratio.  python tools/rollout_bench.py [--batch 1024] [--ticks 64] [--phase 1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(batch=1024, ticks=64, phase=1, reps=5, noise=0.01):
    import torch
    from inria_wbc_amd import capi, structure, trajs
    from inria_wbc_amd import model as mdl
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    B, K = batch, ticks
    dev = torch.device("cuda", 0)
    s = mdl.sample_states(m, tm, B, 9_000_000, q_noise=noise, v_noise=5 * noise, ref_noise=noise)
    com_blk = next(b for b in tm.blocks if b.kind == mdl.T_COM)
    pos, vel, acc = trajs.move_com_stream(m.com(m.q0), [[0.0, 0.0, -0.2]], "001", tm.dt, 2.0, loop=True, absolute=False)
    tab = np.concatenate([pos, vel, acc], axis=1)  # [4000, 9]
    refs = np.repeat(s["ref"][None], K, axis=0).copy()
    idx = (np.arange(K)[:, None] + phase * np.arange(B)[None, :]) % len(tab)
    refs[:, :, com_blk.ref:com_blk.ref + 9] = tab[idx]
    L = st.field_lengths()
    tlb = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
    tub = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
    w = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
    d_refs = torch.from_numpy(np.ascontiguousarray(refs)).to(dev)
    sp = torch.cuda.current_stream().cuda_stream
    h = capi.Handle(0, capi.F64)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
               status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    q0, v0 = torch.from_numpy(s["q"]).to(dev), torch.from_numpy(s["v"]).to(dev)
    rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
    rows.update(tlb=tlb, tub=tub, w=w)

    def by_ticks():
        q, v = q0.clone(), v0.clone()
        qn, vn = torch.zeros_like(q), torch.zeros_like(v)
        its = 0
        for t in range(K):
            h.tick(0, B, dict(q=q, v=v, ref=d_refs[t]), rows, out, qn, vn, tm.dt, stream=sp)
            q, qn = qn, q
            v, vn = vn, v
        return q, v

    isum = torch.zeros(B, dtype=torch.int32, device=dev)
    nok = torch.zeros(B, dtype=torch.int32, device=dev)
    q2, v2 = torch.zeros_like(q0), torch.zeros_like(v0)

    def by_rollout():
        h.rollout(0, B, K, dict(q=q0, v=v0, ref=d_refs), dict(tlb=tlb, tub=tub, w=w), out, q2, v2, tm.dt, iters_sum=isum, ticks_ok=nok, stream=sp)
        return q2, v2

    res = {}
    finals = {}
    for name, fn in (("ticks", by_ticks), ("rollout", by_rollout)):
        # the library measures one stream of ticks, then two sub-batches -- the first sample of either form discarded, each read by a LATER call --
        # and then chooses: eight calls settle it (three did before the cold samples were discarded, round 5)
        for _ in range(8 if name == "rollout" else 1):
            fn()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            qf, vf = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        finals[name] = (qf.clone(), vf.clone())
        res[name] = {"ticks_per_s": B * K / dt, "ms_per_tick_of_the_batch": dt / K * 1e3}
    res["speedup"] = res["rollout"]["ticks_per_s"] / res["ticks"]["ticks_per_s"]
    res["bitwise_equal_final_state"] = bool(torch.equal(finals["ticks"][0], finals["rollout"][0]) and torch.equal(finals["ticks"][1], finals["rollout"][1]))
    it = isum.cpu().numpy() / K
    res.update(batch=B, n_ticks=K, phase=phase, iters_per_tick_mean=float(it.mean()), iters_per_tick_max_instance=float(it.max()),
               ticks_ok_all=bool((nok == K).all().item()))
    h.close()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--ticks", type=int, default=64)
    ap.add_argument("--phase", type=int, default=1)
    ap.add_argument("--noise", type=float, default=0.01)
    a = ap.parse_args()
    print(json.dumps(run(a.batch, a.ticks, a.phase, noise=a.noise)))
