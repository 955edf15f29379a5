#!/usr/bin/env python3
"""The inequality loop of the compact kernel as each of its four waves sees it: the straggler and the median QP of the bench's tick, alone on
the chip, through four stamped builds (-DWBCQP_STAMPS -DWBCQP_STAMP_TID=0 / 64 / 128 / 192; tools/variants.sh st0 ... st3).  A phase that ends in
a barrier has two stamps: one before the barrier (the wave's own work) and one behind it (its wait for the slowest wave), so the critical wave of
every phase can be read off.  Cycles per pick (columns: wave 0..3).

    tools/variants.sh st0 "-DWBCQP_STAMPS -DWBCQP_STAMP_TID=0" st1 "-DWBCQP_STAMPS -DWBCQP_STAMP_TID=64" st2 "-DWBCQP_STAMPS -DWBCQP_STAMP_TID=128" st3 "-DWBCQP_STAMPS -DWBCQP_STAMP_TID=192"
    gpurun -- python tools/loop_waves.py [--tick 40] [--qps 943,412]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ROWS = [(9, "P  decide (new outer iteration: optimal? psi?)"), (10, "P  row constants"), (24, "A  d = J'n (work)"), (11, "A  barrier wait"),
        (18, "B  pending update + z (waves 0-2)"), (25, "B  ballot | wave 3: r, t1, |d2|^2, reflector"), (12, "B  barrier wait"),
        (20, "C  read B's results, decide"), (23, "C  w, x, u, Ri column, next s"), (26, "C  bookkeeping"),
        (29, "E  minimum: atomics, barrier E1"), (30, "E  word: read, atomic, barrier E2"), (14, "E  read the word, re-arm, x swap"),
        (27, "D1 move, coefficients (work)"), (13, "D1 barrier wait"), (28, "D2 rows of J / Ri (work)"), (15, "D2 barrier wait")]

SETUP = [(0, "load, landing"), (1, "H, g assembly"), (2, "elimination of the dv block"), (3, "force blocks, J = U^-1"), (4, "x0"), (22, "rhs of the equalities"),
         (5, "B = J0'N"), (6, "equality QR"), (19, "y"), (8, "x, u"), (7, "actuation rows -> registers, loop's arrays zeroed"), (16, "loop exit"), (17, "decode, store")]


def one(lib, qp_list, tick):
    import torch
    from inria_wbc_amd import capi, structure, synth
    capi.LIB_PATH = lib
    L = capi.load_library()
    L.wbcqp_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    st = structure.talos_structure()
    B = 1024
    dev = torch.device("cuda", 0)
    inputs = synth.generate(st, B, synth.SEED_BASE["talos"])
    com_rows = np.where(st.dense_row_task == st.task_names.index("com"))[0]
    table = np.stack([synth.squat_com_rhs(st, t, st.kp.get("com", 30.0)) for t in range(4000)])
    inputs["b1"][:, com_rows] += table[(np.arange(B) + tick) % 4000][:, :com_rows.size]
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items() if v.size}
    h = capi.Handle(0, capi.F64, flags=capi.FLAG_INDEX_ORDER | capi.FLAG_HW_DISPATCH)
    h.set_structure(0, st)
    sp = torch.cuda.current_stream().cuda_stream
    res = {}
    for qp in qp_list:
        o1 = dict(x=torch.zeros(1, st.n, dtype=torch.float64, device=dev), tau=torch.zeros(1, st.na, dtype=torch.float64, device=dev),
                  status=torch.zeros(1, dtype=torch.int32, device=dev), iters=torch.zeros(1, dtype=torch.int32, device=dev))
        one_in = {k: v[qp:qp + 1].contiguous() for k, v in d_in.items()}
        dbg = torch.zeros(1, capi.K_STAMPS, dtype=torch.int64, device=dev)
        assert L.wbcqp_debug_set_stamp_buffer(h._h, C.c_void_p(dbg.data_ptr())) == 0
        for _ in range(3):
            h.solve_batch(0, 1, one_in, o1, stream=sp)
        torch.cuda.synchronize()
        res[qp] = (int(o1["iters"].cpu()[0]), dbg.cpu().numpy()[0].copy())
    h.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tick", type=int, default=40)
    ap.add_argument("--qps", default="943,412")
    ap.add_argument("--tags", default="st0,st1,st2,st3")
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    qps = [int(q) for q in args.qps.split(",")]
    if args.child:
        r = one(args.child, qps, args.tick)
        for qp, (it, d) in r.items():
            print("R", qp, it, " ".join(str(int(v)) for v in d))
        return
    data = {}
    for w, tag in enumerate(args.tags.split(",")):  # one process per library: a process keeps the first libwbcqp it loaded
        lib = os.path.join(ROOT, "inria_wbc_amd", "lib", "libwbcqp_%s.so" % tag)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, "--tick", str(args.tick), "--qps", args.qps],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        for ln in out.splitlines():
            if ln.startswith("R "):
                f = ln.split()
                data[(int(f[1]), w)] = (int(f[2]), np.array([int(v) for v in f[3:]], dtype=np.int64))
    for qp in qps:
        it = data[(qp, 0)][0]
        print("QP %d alone, %d iterations; cycles per iteration, waves 0..3 (total cycles of the QP: %s)" %
              (qp, it, " ".join(str(int(data[(qp, w)][1].sum())) for w in range(4))))
        for idx, nm in ROWS:
            vals = [data[(qp, w)][1][idx] / max(it, 1) for w in range(4)]
            print("   %-46s %7.0f %7.0f %7.0f %7.0f" % (nm, *vals))
        loop = [sum(data[(qp, w)][1][idx] for idx, _ in ROWS) / max(it, 1) for w in range(4)]
        print("   %-46s %7.0f %7.0f %7.0f %7.0f" % ("loop, per iteration", *loop))
        print("   set-up, cycles of the QP (a phase's stamp follows its last barrier: a wave that finishes early shows the others' time as its own):")
        for idx, nm in SETUP:
            print("   %-46s %7.0f %7.0f %7.0f %7.0f" % (nm, *[data[(qp, w)][1][idx] for w in range(4)]))
        print("   %-46s %7.0f %7.0f %7.0f %7.0f" % ("set-up, total", *[sum(data[(qp, w)][1][idx] for idx, _ in SETUP) for w in range(4)]))


if __name__ == "__main__":
    main()
