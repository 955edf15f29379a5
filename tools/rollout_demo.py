#!/usr/bin/env python3
"""Closed-loop squat of a batch of Talos-like robots, every tick on the device (rows -> QP -> integration through wbcqp_tick),
the CoM reference following etc/talos/squat.yaml's stream.  Prints the CoM height of robot 0 against its reference and the
tick rate.  Usage (GPU box): python tools/rollout_demo.py [--batch 1024] [--ticks 2000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--ticks", type=int, default=2000)
    ap.add_argument("--flags", type=lambda x: int(x, 0), default=0, help="wbcqp_desc.flags (launch-order variants, include/wbcqp.h)")
    ap.add_argument("--desync", action="store_true", help="every robot at its own phase of the squat (iteration counts spread and drift)")
    args = ap.parse_args()
    import torch
    from inria_wbc_amd import capi, structure, trajs
    from inria_wbc_amd import model as mdl
    m = mdl.talos_like()
    st = structure.talos_structure()
    tm = mdl.build_taskmap(m, st, mdl.talos_stack())
    B, dt = args.batch, tm.dt
    dev = torch.device("cuda", 0)
    h = capi.Handle(0, capi.F64, flags=args.flags)
    h.set_structure(0, st)
    h.set_model(0, m, tm)
    s = mdl.sample_states(m, tm, B, 123_000, q_noise=0.002, v_noise=0.01, ref_noise=0.0)
    L = st.field_lengths()
    q, v, ref = (torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref"))
    qn, vn = torch.zeros_like(q), torch.zeros_like(v)
    rows = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
    rows["tlb"] = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
    rows["tub"] = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
    rows["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
    out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, st.na, dtype=torch.float64, device=dev),
               status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    blk = next(b for b in tm.blocks if b.kind == mdl.T_COM)
    pos, vel, acc = trajs.move_com_stream(m.com(m.q0), [[0.0, 0.0, -0.2]], "001", dt, 2.0, loop=True, absolute=False)
    stream9 = torch.from_numpy(np.concatenate([pos, vel, acc], axis=1)).to(dev)
    sp = torch.cuda.current_stream().cuda_stream
    phase = torch.from_numpy(np.random.default_rng(5).integers(0, stream9.shape[0], B)).to(dev) if args.desync else None
    t0 = time.perf_counter()
    for k in range(args.ticks):
        if phase is None:
            ref[:, blk.ref:blk.ref + 9] = stream9[k % stream9.shape[0]]
        else:
            ref[:, blk.ref:blk.ref + 9] = stream9[(phase + k) % stream9.shape[0]]
        h.tick(0, B, dict(q=q, v=v, ref=ref), rows, out, qn, vn, dt, stream=sp)
        q, qn = qn, q
        v, vn = vn, v
        if (k + 1) % 250 == 0:
            torch.cuda.synchronize()
            bad = int((out["status"] != 0).sum().item())
            print("tick %5d  CoM z of robot 0: %.4f  (reference %.4f)  mean iterations %.2f  non-optimal %d" %
                  (k + 1, m.com(q[0].cpu().numpy())[2], pos[k % pos.shape[0]][2], out["iters"].float().mean().item(), bad))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%d robots x %d ticks in %.2f s: %.2f M ticks/s (a 1 kHz controller for %d robots needs %.2f M)" %
          (B, args.ticks, el, B * args.ticks / el / 1e6, B, B * 1e-3))
    h.close()


if __name__ == "__main__":
    main()
