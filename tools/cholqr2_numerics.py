#!/usr/bin/env python3
"""Numerics of the one equality-phase candidate that was never bounded: CholeskyQR2 of B = J0'N = L^-1 CE' instead of the
eighteen-reflector Householder QR.  CPU only (numpy on the oracle's dense assembly -- the oracle as a checker, never the product): for every shipped
humanoid stack, cond(B), cond(H), and the loss of orthogonality ||Q'Q - I||_max after one and after two Cholesky passes.  What it answers: is the
method admissible at the 1e-8 parity bar on these stacks at all (cond(B)^2 against 1/eps)?   python tools/cholqr2_numerics.py [--batch 64]"""
import argparse
import os
import sys

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--noise", type=float, default=0.5)
    args = ap.parse_args()
    from inria_wbc_amd import structure, synth
    from oracle import oracle
    stacks = (("talos", structure.talos_structure(), "talos"), ("talos_single_support", structure.talos_structure(True), "talos"),
              ("icub", structure.icub_structure(), "icub"))
    print("stack                  n neq   cond(B) median / max      cond(H) median / max     ||Q'Q-I|| pass 1   pass 2   first-order pass 2   breakdowns")
    for name, st, seed in stacks:
        inp = synth.generate(st, args.batch, synth.SEED_BASE[seed], task_noise=args.noise)
        cb, ch, e1, e2, e2f, bad = [], [], [], [], [], 0
        for i in range(args.batch):
            H, g, CE, ce0, CI, ci0 = oracle.assemble(st, inp, i)
            L = np.linalg.cholesky(H)
            B = sl.solve_triangular(L, CE.T, lower=True)  # n x neq
            s = np.linalg.svd(B, compute_uv=False)
            cb.append(s[0] / s[-1])
            ch.append(np.linalg.cond(H))
            try:
                R1 = np.linalg.cholesky(B.T @ B).T
            except np.linalg.LinAlgError:
                bad += 1
                continue
            Q1 = sl.solve_triangular(R1, B.T, trans="T", lower=False).T
            E = Q1.T @ Q1 - np.eye(st.neq)
            e1.append(np.abs(E).max())
            R2 = np.linalg.cholesky(Q1.T @ Q1).T
            Q2 = sl.solve_triangular(R2, Q1.T, trans="T", lower=False).T
            e2.append(np.abs(Q2.T @ Q2 - np.eye(st.neq)).max())
            # pass 2 without a second factorisation: chol(I + E) = I + triu(E, 1) + diag(E) / 2 + O(E^2), (I + U)^-1 = I - U + O(E^2)
            U = np.triu(E, 1) + 0.5 * np.diag(np.diag(E))
            Qf = Q1 - Q1 @ U
            e2f.append(np.abs(Qf.T @ Qf - np.eye(st.neq)).max())
        cb, ch = np.array(cb), np.array(ch)
        print("%-20s %3d %3d   %9.3g / %9.3g   %9.3g / %9.3g   %12.3g   %9.3g   %12.3g   %6d" %
              (name, st.n, st.neq, np.median(cb), cb.max(), np.median(ch), ch.max(), max(e1), max(e2), max(e2f), bad))


if __name__ == "__main__":
    main()
