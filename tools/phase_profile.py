#!/usr/bin/env python3
"""Per-phase cycle shares of the solve kernel from the -DWBCQP_STAMPS diagnostic build.

Reads SHARES, not run time: the stamped build forbids overlaps the real kernel has
(cdna_hip_programming.md section 7, In-kernel stamps). Usage (on the GPU box):
    python inria_wbc_amd/build.py --stamps && python tools/phase_profile.py [--robot talos] [--batch 256] [--noise 0.5]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# stamp index -> phase.  The inequality loop's names are the compact kernel's (round 5: P / A / B / C / E / D1 / D2 of csrc/wbcqp_compact.hpp, a work
# stamp before each barrier and a wait stamp behind it); the full layout (flags 16) keeps round 1's phases under the same indices 9-15
NAMES = ["load", "assemble_Hg", "cholesky", "J=U^-1", "x0", "eq:N,rhs,B=J0'N", "eq:QR", "actuation rows -> registers", "eq:solve x,u (+Givens path)",
         "in:P decide | full: s+psi+save", "in:P row constants | full: argmin+build", "in:A barrier wait | full: d", "in:B barrier wait | full: z+r",
         "in:D1 barrier wait | full: steplen+step", "in:E read the word, re-arm, x swap | full: add", "in:D2 barrier wait | full: delete",
         "loop-exit", "decode+store", "in:B pending update + z (waves 0-2)", "eq:W<-WT || y,u", "in:C read B's results, decide", "eq:N build (full)", "eq:rhs",
         "in:C w, x, u, Ri column, next s", "in:A d = J'n (work)", "in:B ballot | wave 3: r, t1, |d2|^2, reflector", "in:C bookkeeping", "in:D1 move, coefficients (work)",
         "in:D2 rows of J / Ri (work)", "in:E minimum: atomics, barrier E1", "in:E word: read, atomic, barrier E2", ""]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robot", default="talos")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--noise", type=float, default=0.5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--model", action="store_true", help="rows from the Talos-like model (wbcqp_problem_data) instead of the synthetic generator")
    ap.add_argument("--qnoise", type=float, default=0.01)
    ap.add_argument("--flags", type=int, default=0, help="wbcqp_desc.flags (16: the full LDS layout of round 1)")
    ap.add_argument("--first", type=int, default=0, help="index of the first instance in the seeded stream (with --squat: its tick)")
    ap.add_argument("--squat", action="store_true", help="CoM rows follow the squat stream (bench.py's workload): heavier tail")
    ap.add_argument("--lib", default=None, help="a stamped variant (built with -DWBCQP_STAMPS and other switches) instead of libwbcqp_stamps.so")
    args = ap.parse_args()
    import torch
    from inria_wbc_amd import capi, structure, synth
    capi.LIB_PATH = os.path.abspath(args.lib) if args.lib else os.path.join(ROOT, "inria_wbc_amd", "lib", "libwbcqp_stamps.so")
    lib = capi.load_library(capi.LIB_PATH)
    st = structure.STRUCTURES[args.robot]()
    B = args.batch
    dev = torch.device("cuda", 0)
    if args.model:
        from inria_wbc_amd import model as mdl
        m = mdl.talos_like()
        tm = mdl.build_taskmap(m, st, mdl.talos_stack())
        s = mdl.sample_states(m, tm, B, 9_000_000, q_noise=args.qnoise, v_noise=5 * args.qnoise, ref_noise=args.qnoise)
        hh = capi.Handle(0, capi.F64)
        hh.set_structure(0, st)
        hh.set_model(0, m, tm)
        L = st.field_lengths()
        d_in = {k: torch.zeros(B, L[k], dtype=torch.float64, device=dev) for k in capi.ROW_FIELDS}
        hh.problem_data(0, B, {k: torch.from_numpy(s[k]).to(dev) for k in ("q", "v", "ref")}, d_in)
        torch.cuda.synchronize()
        d_in["tlb"] = torch.from_numpy(np.tile(-m.tau_max, (B, 1))).to(dev)
        d_in["tub"] = torch.from_numpy(np.tile(m.tau_max, (B, 1))).to(dev)
        d_in["w"] = torch.from_numpy(np.tile(st.default_weights, (B, 1))).to(dev)
    else:
        inputs = synth.generate(st, B, synth.SEED_BASE[args.robot], first=args.first, task_noise=args.noise, squat=args.squat)
        d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    d_out = dict(x=torch.zeros(B, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(B, max(st.na, 1), dtype=torch.float64, device=dev),
                 status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    dbg = torch.zeros(B, capi.K_STAMPS, dtype=torch.int64, device=dev)
    h = capi.Handle(0, capi.F64, flags=args.flags | capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH)
    h.set_structure(0, st)
    lib.wbcqp_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    rc = lib.wbcqp_debug_set_stamp_buffer(h._h, C.c_void_p(dbg.data_ptr()))
    assert rc == 0, rc
    for _ in range(2):
        h.solve_batch(0, B, d_in, d_out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().astype(np.float64)
    iters = d_out["iters"].cpu().numpy()
    tot = t.sum(axis=1)
    mean = t.mean(axis=0)
    print("robot %s batch %d noise %g: mean iters %.2f, mean cycles/QP %.0f (s_memtime ticks @100MHz? see note), max %.0f" %
          (args.robot, B, args.noise, iters.mean(), tot.mean(), tot.max()))
    rows = []
    for i, nm in enumerate(NAMES):
        if mean[i] > 0:
            print("  %-18s %12.0f  %5.1f %%" % (nm, mean[i], 100 * mean[i] / mean.sum()))
            rows.append({"phase": nm, "ticks": float(mean[i]), "share": float(mean[i] / mean.sum())})
    # what the spread of per-QP cycles costs a launch of B workgroups on 256 CUs, one QP per CU at a time:
    # in-order greedy dispatch (what the hardware does) against longest-first order and against the perfect split
    import heapq
    def makespan(costs, cus=256):
        heap = [0.0] * cus
        for cst in costs:
            heapq.heappush(heap, heapq.heappop(heap) + cst)
        return max(heap)
    ideal = tot.sum() / 256.0
    print("  schedule on 256 CUs: in-order %.0f, longest-first %.0f, perfect split %.0f cycles (iters max %d)" %
          (makespan(tot), makespan(sorted(tot, reverse=True)), ideal, int(iters.max())))
    hist = np.bincount(np.minimum(iters, 20), minlength=21)
    print("  iterations histogram:", hist.tolist())
    # cost model: cycles = setup + per_iteration * iterations (least squares over the batch)
    cA = np.stack([np.ones(B), iters.astype(np.float64)], axis=1)
    fit = np.linalg.lstsq(cA, tot, rcond=None)[0]
    print("  fit: %.0f + %.0f per iteration" % (fit[0], fit[1]))
    by_it = [float(tot[iters == k].mean()) if (iters == k).any() else 0.0 for k in range(1, 13)]
    print("  mean cycles by iteration count 1..12:", [int(v) for v in by_it])
    worst = int(np.argmax(iters))
    print("  the longest QP: %d iterations, %.0f cycles:" % (iters[worst], tot[worst]), {NAMES[i]: int(t[worst, i]) for i in range(len(NAMES)) if t[worst, i] > 0})
    if args.out:
        with open(args.out, "w") as fh:
            json.dump({"robot": args.robot, "batch": B, "noise": args.noise, "iters_mean": float(iters.mean()),
                       "ticks_per_qp": float(tot.mean()), "phases": rows}, fh, indent=1)
    h.close()


if __name__ == "__main__":
    main()
