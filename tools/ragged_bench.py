#!/usr/bin/env python3
"""BASELINE config 5: one launch over a mixed Franka / Tiago / iCub / Talos / Talos-single-support batch (per-QP structure
drawn uniformly, seed base 5e6), `wbcqp_solve_ragged`.  Prints one JSON line: QP/s, the mix, the algorithmic
bytes moved, the fraction of HBM peak, parity of a sample against the oracle, and how `inria_wbc_amd.shard.ragged_shards`
would cut this batch over 8 ranks (by cumulative n^3 cost).

    python tools/ragged_bench.py [--batch 8192] [--steps 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["franka", "tiago", "icub", "talos", "talos_single_support"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    print(json.dumps(run(args)), flush=True)


def run(args):
    """args: .batch, .steps, .no_parity -> the result dict (bench.py calls this for its `other_configs.config5_ragged`)"""
    import torch
    from inria_wbc_amd import capi, shard, structure, synth
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5_000_000)
    kinds = rng.integers(0, len(NAMES), size=args.batch)
    h = capi.Handle(0, capi.F64)
    groups, metas, total_bytes = [], [], 0
    for slot, name in enumerate(NAMES):
        st = structure.STRUCTURES[name]()
        cnt = int((kinds == slot).sum())
        if cnt == 0:
            continue
        nb = min(cnt, 256)
        inp = synth.generate(st, nb, synth.SEED_BASE["ragged"] + 10_000 * slot, task_noise=1.0)
        reps = (cnt + nb - 1) // nb
        d_in = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (reps, 1))[:cnt])).to(dev) for k, v in inp.items() if v.size}
        d_out = dict(x=torch.zeros(cnt, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(cnt, max(st.na, 1), dtype=torch.float64, device=dev),
                     status=torch.full((cnt,), -99, dtype=torch.int32, device=dev), iters=torch.zeros(cnt, dtype=torch.int32, device=dev))
        h.set_structure(slot, st)
        groups.append((slot, cnt, d_in, d_out))
        metas.append((st, inp, cnt))
        total_bytes += cnt * st.algorithmic_bytes()
    sp = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        h.solve_ragged(groups, stream=sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        h.solve_ragged(groups, stream=sp)
    e1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ms_ev = e0.elapsed_time(e1) / args.steps
    res = {"metric": "QP solves/sec (ragged Franka/Tiago/iCub/Talos mix, one launch)", "value": args.batch / dt, "unit": "QP/s", "batch": args.batch,
           "ms_per_step": dt * 1e3, "kernel_ms": ms_ev,
           "mix": {st.name: cnt for st, _, cnt in metas}, "n": {st.name: st.n for st, _, _ in metas},
           "algorithmic_bytes_per_launch": total_bytes, "frac_hbm": total_bytes / (ms_ev * 1e-3) / 1e9 / 8000.0,
           "status_optimal": int(sum(int((g[3]["status"] == 0).sum().item()) for g in groups)),
           "iters_mean": {st.name: float(g[3]["iters"].float().mean().item()) for (st, _, _), g in zip(metas, groups)}}
    plan = shard.ragged_shards([(st.n, cnt) for st, _, cnt in metas], 8)
    cost = shard.shard_costs([(st.n, cnt) for st, _, cnt in metas], plan)
    res["shards_8_ranks"] = {"qps_per_rank": [sum(e - b for _, b, e in pieces) for pieces in plan],
                             "cost_share": [round(c / sum(cost), 4) for c in cost]}
    if not args.no_parity:
        from oracle import oracle
        worst, same = 0.0, []
        for (st, inp, cnt), g in zip(metas, groups):
            ns = min(cnt, inp["h"].shape[0], 64)
            ref = oracle.tick_batch(st, {k: v[:ns] for k, v in inp.items()})
            x = g[3]["x"][:ns].cpu().numpy()
            ok = ref["status"] == 0
            worst = max(worst, float((np.abs(x - ref["x"]).max(axis=1) / np.maximum(1.0, np.abs(ref["x"]).max(axis=1)))[ok].max()))
            same.append(bool(np.array_equal(g[3]["status"][:ns].cpu().numpy(), ref["status"])))
        res["parity"] = {"max_rel_dx": worst, "status_equal": all(same), "sample_per_structure": 64}
    h.close()
    return res


if __name__ == "__main__":
    main()
