#!/usr/bin/env python3
"""What does a second QP resident on the same CU buy?  Same launch, same library, once as the LDS footprint allows and once
with the dynamic LDS padded (env WBCQP_DEBUG_LDS_PAD, read at wbcqp_create) so that fewer workgroups fit a CU.
Index order through the hardware's dispatcher: no launch-order effects in the comparison.

    python tools/residency_probe.py [--robot mini|icub|talos|talos_single_support] [--batch 4096]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mini_structure():
    """Floating base, nv 26, na 20, one 6-D contact: the phases of a humanoid QP at a size whose working set leaves room."""
    from inria_wbc_amd import structure as S
    pts = S.contact6d_points(lxn=0.11, lyn=0.069, lxp=0.1, lyp=0.069, lz=0.107)
    cr = S.Contact("contact_rfoot", pts, (0.0, 0.0, 1.0), 0.3, 5.0, 1500.0)
    dense = [("lh", 6, 10.0), ("rf", 6, 1000.0), ("com", 3, 1000.0), ("__posture__", "posture", 1.75), ("__contacts__",)]
    level0 = [(S.INEQ_BOUNDS, 0), (S.INEQ_ACTUATION, 0), (S.INEQ_FORCE, 0)]
    return S._mk("mini", 26, 20, [cr], dense, None, [], True, True, level0, {"com": 30.0})


def run(st, B, pad, steps, flags):
    import torch
    from inria_wbc_amd import capi, synth
    if pad:
        os.environ["WBCQP_DEBUG_LDS_PAD"] = str(pad)
    else:
        os.environ.pop("WBCQP_DEBUG_LDS_PAD", None)
    dev = torch.device("cuda", 0)
    nb = min(B, 1024)
    inp = synth.generate(st, nb, 7_000_000)
    reps = (B + nb - 1) // nb
    d_in = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (reps, 1))[:B])).to(dev) for k, v in inp.items() if v.size}
    d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, max(st.na, 1), dtype=torch.float64, device=dev),
                 status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    h = capi.Handle(0, capi.F64, flags=flags)
    h.set_structure(0, st)
    sp = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        h.solve_batch(0, B, d_in, d_out, stream=sp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.solve_batch(0, B, d_in, d_out, stream=sp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    it = d_out["iters"].cpu().numpy()
    ok = int((d_out["status"] == 0).sum().item())
    h.close()
    return {"pad": pad, "ms": dt * 1e3, "qps": B / dt, "iters_mean": float(it.mean()), "optimal": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robot", default="mini")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--pads", default="0,4000")
    args = ap.parse_args()
    from inria_wbc_amd import capi, structure
    st = mini_structure() if args.robot == "mini" else structure.STRUCTURES[args.robot]()
    lay = capi.layout_of(st)
    out = {"robot": st.name, "n": st.n, "lds_bytes": lay["lds_bytes"], "batch": args.batch, "runs": []}
    for pad in [int(p) for p in args.pads.split(",")]:
        total = min(lay["lds_bytes"] + pad, 160 * 1024)
        r = run(st, args.batch, pad, args.steps, capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH)
        r["lds_total"] = total
        r["resident_per_cu"] = (160 * 1024) // total
        out["runs"].append(r)
        print(json.dumps(r), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
