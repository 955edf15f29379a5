#!/usr/bin/env python3
"""A drop of the compact kernel's inequality loop (partial / dual step: D1 move + coefficients | barrier | D2 rows of J / Ri | barrier | the step's scalars
again), stamp by stamp and wave by wave, on the stream's straggler alone on the chip: four stamped builds with the drop's finer stamps
(-DWBCQP_STAMPS -DWBCQP_STAMP_DROP -DWBCQP_STAMP_TID=0 / 64 / 128 / 192).  Cycles PER DROP (the loop's other rows per iteration as tools/loop_waves.py).

    tools/variants.sh dp0 "-DWBCQP_STAMPS -DWBCQP_STAMP_DROP -DWBCQP_STAMP_TID=0" dp1 "... =64" dp2 "... =128" dp3 "... =192"
    gpurun -- python tools/drop_profile.py [--qps 943]"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DROP = [(36, "re-entry: t, checks, up to the drop's start"), (37, "D1 x, u moved (all), act flag"), (32, "D1 wave 1: loads of row p of Ri and of d"),
        (33, "D1 wave 1: the two scans"), (34, "D1 wave 1: rsqrt x 2"), (35, "D1 wave 1: coefficients stored"), (27, "D1 rest of the work stamp (wave 2: P_L of every row of Ri)"),
        (13, "D1 barrier wait"), (38, "D2 waves 0-1: row of J through the rotations"), (39, "D2 wave 3: row of Ri through the rotations"),
        (40, "D2 wave 2: Z(i, last) from D1's P_L; r, u, A moved"), (41, "D2 wave 2: t1 elected"), (28, "D2 rest of the work stamp"), (15, "D2 barrier wait"),
        (42, "after the drop: t1, t2, reflector scalars")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tick", type=int, default=40)
    ap.add_argument("--qps", default="943")
    ap.add_argument("--tags", default="dp0,dp1,dp2,dp3")
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    qps = [int(q) for q in args.qps.split(",")]
    if args.child:
        from inria_wbc_amd import capi
        from tools import loop_waves
        capi.K_STAMPS = 48
        for qp, (it, d) in loop_waves.one(args.child, qps, args.tick).items():
            print("R", qp, it, " ".join(str(int(v)) for v in d))
        return
    data = {}
    for w, tag in enumerate(args.tags.split(",")):
        lib = os.path.join(ROOT, "inria_wbc_amd", "lib", "libwbcqp_%s.so" % tag)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, "--tick", str(args.tick), "--qps", args.qps],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        for ln in out.splitlines():
            if ln.startswith("R "):
                f = ln.split()
                data[(int(f[1]), w)] = (int(f[2]), np.array([int(v) for v in f[3:]], dtype=np.int64))
    for qp in qps:
        it = data[(qp, 0)][0]
        drops = int(data[(qp, 0)][1][44])
        print("QP %d alone: %d iterations, %d drops, mean rotations per drop %.1f, mean active inequality rows at a drop %.1f; cycles PER DROP, waves 0..3" %
              (qp, it, drops, data[(qp, 0)][1][45] / max(drops, 1), data[(qp, 0)][1][46] / max(drops, 1)))
        tot = [0.0] * 4
        for idx, nm in DROP:
            vals = [data[(qp, w)][1][idx] / max(drops, 1) for w in range(4)]
            tot = [a + b for a, b in zip(tot, vals)]
            print("   %-52s %7.0f %7.0f %7.0f %7.0f" % (nm, *vals))
        print("   %-52s %7.0f %7.0f %7.0f %7.0f" % ("a drop, total", *tot))
        print("   whole QP, cycles: %s" % " ".join(str(int(data[(qp, w)][1][:44].sum())) for w in range(4)))


if __name__ == "__main__":
    main()
