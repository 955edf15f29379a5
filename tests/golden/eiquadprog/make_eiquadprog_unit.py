#!/usr/bin/env python3
"""tests/golden/eiquadprog/eiquadprog_unit.npz: the two-variable unit problems of eiquadprog's own test file for EiquadprogFast
(stack-of-tasks/eiquadprog, tests/eiquadprog-fast.cpp -- not in /root/reference: [UPSTREAM-RECALL] for the problem list, but every optimum,
objective value and outcome below is derivable by hand and is written next to its derivation, so the fixture pins the oracle's
`wbco_eiquadprog_fast` and the library's dense seam independently of either).

Convention (eiquadprog-fast.hpp): min 1/2 x'Qx + C'x  s.t.  Aeq x + Beq = 0,  Aineq x + Bineq >= 0.
Status codes (eiquadprog-fast.hpp): OPTIMAL 0, INFEASIBLE 1, UNBOUNDED 2, MAX_ITER_REACHED 3, REDUNDANT_EQUALITIES 4; an infeasible set of
inequalities leaves solve_quadprog through "t = inf" = UNBOUNDED (the dual is unbounded), which tsid's
SolverHQuadProgFast reports as HQP_STATUS_INFEASIBLE -- `tsid` below is the status the C ABI returns.

    python tests/golden/eiquadprog/make_eiquadprog_unit.py      # rewrites the .npz next to this file
"""
import os

import numpy as np

I2 = np.eye(2)
Z = lambda r: np.zeros((r, 2))
CASES = [
    # name, Q, C, Aeq, Beq, Aineq, Bineq, x*, f*, eiquadprog status, tsid status
    # min |x|^2 / 2: x* = 0, f* = 0
    ("unbiased", I2, [0, 0], Z(0), [], Z(0), [], [0, 0], 0.0, 0, 0),
    # min |x - (1, 1)|^2 / 2 - 1: gradient x - (1, 1) = 0 -> x* = (1, 1), f* = 1 - 2 = -1
    ("biased", I2, [-1, -1], Z(0), [], Z(0), [], [1, 1], -1.0, 0, 0),
    # min |x|^2 / 2 s.t. x0 + x1 = 1: by symmetry x* = (1/2, 1/2), f* = 1/4
    ("equality_constraints", I2, [0, 0], [[1, 1]], [-1], Z(0), [], [0.5, 0.5], 0.25, 0, 0),
    # min |x|^2 / 2 s.t. x_i >= 1: both bounds active, x* = (1, 1), f* = 1
    ("inequality_constraints", I2, [0, 0], Z(0), [], I2, [-1, -1], [1, 1], 1.0, 0, 0),
    # min |x - (1, 1)|^2 / 2 - 1 s.t. x0 + x1 = 5, x1 >= 3: on the line the unconstrained minimiser is (5/2, 5/2), x1 >= 3 binds:
    # x* = (2, 3), f* = (4 + 9) / 2 - 5 = 3/2
    ("full", I2, [-1, -1], [[1, 1]], [-5], [[0, 1]], [-3], [2, 3], 1.5, 0, 0),
    # x0 = 1 and x0 = -1: the second equality's normal is the first one's -> REDUNDANT_EQUALITIES -> tsid ERROR
    ("unfeasible_equalities", I2, [0, 0], [[1, 0], [1, 0]], [-1, 1], Z(0), [], None, None, 4, 4),
    # x0 >= 1 and x0 <= -1: after x0 >= 1 is active the second row has z = 0 and no multiplier to trade -> t = inf -> UNBOUNDED -> tsid INFEASIBLE
    ("unfeasible_inequalities", I2, [0, 0], Z(0), [], [[1, 0], [-1, 0]], [-1, -1], None, None, 2, 1),
    # one more with two active rows (hand-checkable, not from upstream): min |x|^2 / 2 - 3 x0 s.t. x0 <= 1, x0 + x1 <= 1/2, written as rows
    # >= 0: -x0 + 1 >= 0, -x0 - x1 + 1/2 >= 0.  With both active x* = (1, -1/2); stationarity x + C = l1 (-1, 0) + l2 (-1, -1) gives
    # (-2, -1/2) = -(l1 + l2, l2): l2 = 1/2, l1 = 3/2, both positive -> optimal; f* = (1 + 1/4) / 2 - 3 = -19/8
    ("two_active", I2, [-3, 0], Z(0), [], [[-1, 0], [-1, -1]], [1, 0.5], [1, -0.5], -19.0 / 8.0, 0, 0),
]


def main():
    out = {"names": np.array([c[0] for c in CASES])}
    for name, Q, C, Aeq, Beq, Ain, Bin, xs, fs, st_e, st_t in CASES:
        out[name + "_H"] = np.asarray(Q, float)
        out[name + "_g"] = np.asarray(C, float)
        out[name + "_CE"] = np.asarray(Aeq, float).reshape(-1, 2)
        out[name + "_ce0"] = np.asarray(Beq, float)
        out[name + "_CI"] = np.asarray(Ain, float).reshape(-1, 2)
        out[name + "_ci0"] = np.asarray(Bin, float)
        out[name + "_x"] = np.full(2, np.nan) if xs is None else np.asarray(xs, float)
        out[name + "_f"] = np.array(np.nan if fs is None else fs)
        out[name + "_status_eiquadprog"] = np.array(st_e)
        out[name + "_status_tsid"] = np.array(st_t)
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "eiquadprog_unit.npz"), **out)


if __name__ == "__main__":
    main()
