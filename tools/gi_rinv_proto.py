"""numpy model of the round-3 active-set loop (csrc/wbcqp_compact.hpp, phase 4) -- the arithmetic only, no parallelism.

What changed against eiquadprog's bookkeeping, and why each piece is still the same algorithm:
  * the inequality block of R is never stored; its INVERSE Ri is.  r = R^-1 d restricted to the inequality rows is
    Ri d_I (a triangular MATVEC instead of a back substitution: no dependent chain); a new column of R, [d; alpha],
    is the column [-r / alpha; 1 / alpha] of the inverse, and r is at hand when a constraint is added.
  * delete_constraint: R without column p is re-triangularised by rotations G; the inverse takes Ri G' with row p and
    the last column dropped.  Row p of Ri G' must vanish left of its last entry, so rotation j is fixed by the running
    norm of row p of Ri: every (cc, ss) follows from ONE prefix sum of squares -- no sequential division / sqrt chain.
  * after a drop nothing is recomputed from scratch: d <- G d, and with delta = the entry of d that leaves the active
    block,  z += delta J(:, iq),  r_i -= delta Z(i, last),  z'n += delta^2,  |d2|^2 += delta^2,  s(ip) += t z'n.

Used by tests/test_gi_rinv_proto.py against the C oracle.  Test infrastructure, not product code.
"""
from __future__ import annotations

import numpy as np

EPS = np.finfo(float).eps
OPTIMAL, INFEASIBLE, MAX_ITER, ERROR = 0, 1, 3, 4


def solve(H, g, CE, ce0, CI, ci0, max_iter=1000, trace=None, projector=False, store=None, round5=False):
    """round5=True: the loop as round 5 runs it (csrc/wbcqp_compact.hpp) -- a drop's rotations in CLOSED FORM (with rho = row p of Ri from column p on,
    S_l = sum_{i<=l} rho_i^2 and P_l = sum_{i<=l} rho_i x_i, the new element of column l of any row x is a_l P_l + b_l x_{l+1}, a_l = -rho_{l+1} / sqrt(S_l S_{l+1}),
    b_l = sqrt(S_l / S_{l+1}), and -P_L / sqrt(S_L) leaves the active block: one running sum per row instead of a chain of rotations), and a constraint's
    reflector kept PENDING: J is left as it is when a constraint is added, the next pick forms d = J_old'n - v (w'n) and applies J <- J - w v' only then.

    store=np.float32: SURVEY section 7's "fp32 storage, fp64 accumulate" option, modelled: J and Ri -- the two arrays that live in LDS for
    the whole loop -- are rounded to `store` after every update, every product and sum stays in f64 (tools/f32_storage_probe.py)."""
    n = g.size
    rnd = (lambda a: a) if store is None else (lambda a: a.astype(store).astype(np.float64))
    neq, m = CE.shape[0], CI.shape[0]
    U = np.linalg.cholesky(H).T
    J = rnd(np.linalg.inv(U))
    c1, c2 = np.trace(H), np.trace(J)
    x = -J @ (J.T @ g)
    iq = 0
    R_norm = 1.0
    if neq:
        B = J.T @ CE.T
        Q, Rq = np.linalg.qr(B, mode="complete")
        if np.any(np.abs(np.diag(Rq[:neq])) <= EPS * max(1.0, np.abs(np.diag(Rq[:neq])).max())):
            return dict(x=x, status=ERROR, iters=0)
        J = rnd(J @ Q)
        y = np.linalg.solve(Rq[:neq].T, -(CE @ x + ce0))
        x = x + J[:, :neq] @ y
        iq = neq
        R_norm = max(1.0, np.abs(np.diag(Rq[:neq])).max())
    Ri = np.zeros((n, n))  # inverse of the inequality block of R, position space (0 .. iq - neq)
    # projector=True: the null-space block J2 = J(:, iq:) is never touched again; the symmetric G = J2 J2' is carried instead
    # (Goldfarb-Idnani's H* operator): z = G n, adding a constraint is G -= z z' / (z'n) with the new column z / sqrt(z'n) of J1,
    # dropping one returns the leaving column q of J1: G += q q'.
    G = J[:, iq:] @ J[:, iq:].T if projector else None
    pend = None  # (w, v, first column) of the last accepted constraint's reflector, not applied to J yet (round5)
    A = [-(i + 1) for i in range(neq)]
    u = np.zeros(n + 2)
    act = np.zeros(m, bool)
    it = 0
    partial_steps = 0
    psi_tol = m * EPS * c1 * c2 * 100.0
    excl = np.ones(m, bool)
    while True:
        it += 1
        if it >= max_iter:
            return dict(x=x, status=MAX_ITER, iters=it)
        s = CI @ x + ci0
        excl[:] = True
        psi = np.minimum(s, 0.0).sum()
        if abs(psi) <= psi_tol:
            break
        x_old, u_old, A_old = x.copy(), u.copy(), list(A)
        done = False
        while True:  # l2
            cand = np.where(~act & excl & (s < 0.0), s, 0.0)
            ip = int(np.argmin(cand))
            if cand[ip] >= 0.0:
                done = True
                break
            npv = CI[ip]
            sip = s[ip]
            if trace is not None:
                trace.setdefault("events", []).append(("pick", ip, iq - neq))
            u[iq] = 0.0
            A = A[:iq] + [ip]
            if round5 and pend is not None:
                w_, v_, pc_ = pend
                d = J.T @ npv                       # from J as it stands ...
                d[pc_:] -= v_ * float(w_ @ npv)     # ... with the pending update folded in
                J[:, pc_:] = rnd(J[:, pc_:] - np.outer(w_, v_))  # (the kernel does this in the pass that forms z)
                pend = None
            else:
                d = J.T @ npv
            if projector:
                d[iq:] = 0.0
                z = G @ npv
            else:
                z = J[:, iq:] @ d[iq:]
            mi = iq - neq
            r = Ri[:mi, :mi] @ d[neq:iq]
            zz, znp, dn2 = z @ z, z @ npv, d[iq:] @ d[iq:]
            if projector:
                dn2 = znp
            rejected = False
            while True:  # l2a with everything carried across drops
                mi = iq - neq
                t1, lpos = np.inf, -1
                for kk in range(mi):
                    if r[kk] > 0.0 and u[neq + kk] / r[kk] < t1:
                        t1, lpos = u[neq + kk] / r[kk], kk
                t2 = -sip / znp if abs(zz) > EPS else np.inf
                t = min(t1, t2)
                if t == np.inf:
                    return dict(x=x, status=INFEASIBLE, iters=it)
                if t2 == np.inf:  # dual step
                    u[neq:iq] -= t * r
                    u[iq] += t
                else:
                    x = x + t * z
                    u[neq:iq] -= t * r
                    u[iq] += t
                    if t == t2:
                        break
                    sip = sip + t * znp
                partial_steps += 1
                # ---- drop position lpos: rotations from the prefix sum over row p of Ri
                p = lpos
                if trace is not None:
                    trace.setdefault("events", []).append(("drop", p, iq - neq))
                act[A[neq + p]] = False
                rho = Ri[p, p:mi].copy()
                S = np.cumsum(rho * rho)
                Z = Ri[:mi, :mi].copy()
                Lr = mi - 1 - p
                if round5:
                    al = -rho[1:] / np.sqrt(S[:-1] * S[1:])
                    bl = np.sqrt(S[:-1] / S[1:])
                    cl = -1.0 / np.sqrt(S[-1])

                    def rot(X):  # rows x columns p .. mi - 1 -> (new columns p .. mi - 2, what leaves)
                        P = np.cumsum(X * rho, axis=1)
                        return al * P[:, :-1] + bl * X[:, 1:], cl * P[:, -1]
                    Zn, Zl = rot(Z[:, p:mi])
                    Jn, Jl = rot(J[:, neq + p:neq + mi])
                    dn, dl = rot(d[None, neq + p:neq + mi])
                    Z[:, p:mi - 1] = Zn; Z[:, mi - 1] = Zl
                    J[:, neq + p:neq + mi - 1] = rnd(Jn); J[:, neq + mi - 1] = rnd(Jl)
                    d[neq + p:neq + mi - 1] = dn[0]; d[neq + mi - 1] = dl[0]
                for j in range(0 if round5 else Lr):  # (rounds 3-4: the rotations one after the other) columns (p + j, p + j + 1)
                    a = rho[0] if j == 0 else -np.sqrt(S[j])
                    b = rho[j + 1]
                    hh = np.sqrt(S[j + 1])
                    cc, ss = b / hh, -a / hh
                    for Mx, c0 in ((Z, p + j), (J, neq + p + j)):
                        t1c, t2c = Mx[:, c0].copy(), Mx[:, c0 + 1].copy()
                        Mx[:, c0] = rnd(cc * t1c + ss * t2c)
                        Mx[:, c0 + 1] = rnd(ss * t1c - cc * t2c)
                    da, db = d[neq + p + j], d[neq + p + j + 1]
                    d[neq + p + j] = cc * da + ss * db
                    d[neq + p + j + 1] = ss * da - cc * db
                delta = d[iq - 1]
                r_full = r - delta * Z[:, mi - 1]
                keep = [i for i in range(mi) if i != p]
                Ri[:mi, :mi] = 0.0
                Ri[:mi - 1, :mi - 1] = Z[np.ix_(keep, range(mi - 1))]
                r = r_full[keep]
                u[neq + p:iq] = u[neq + p + 1:iq + 1]
                u[iq] = 0.0
                A = A[:neq + p] + A[neq + p + 1:]
                iq -= 1
                if projector:
                    G += np.outer(J[:, iq], J[:, iq])
                z = z + delta * J[:, iq]
                znp += delta * delta
                dn2 += delta * delta
                zz = z @ z
            # ---- full step: add ip with one reflector (v = d[iq:] - alpha e0)
            diq = d[iq]
            alpha = diq
            if projector:
                alpha = np.sqrt(znp) if znp > 0.0 else 0.0
                if abs(alpha) > EPS * R_norm:
                    J[:, iq] = z / alpha
                    G -= np.outer(z, z) / znp
            elif iq + 1 < n and dn2 > 0.0:
                nx = np.sqrt(dn2)
                alpha = -nx if diq >= 0.0 else nx
                v = d[iq:].copy()
                v[0] -= alpha
                tau = 1.0 / (nx * abs(diq) + dn2)
                w = tau * (z - alpha * J[:, iq])
                if round5:
                    pend = (w, v, iq)
                else:
                    J[:, iq:] = rnd(J[:, iq:] - np.outer(w, v))
            if abs(alpha) <= EPS * R_norm:  # dependent: back to the saved iterate, pick another
                excl[ip] = False
                for i in range(min(iq, len(A_old))):
                    pass
                A = list(A_old[:iq])
                act[:] = False
                for a in A:
                    if a >= 0:
                        act[a] = True
                u[:iq] = u_old[:iq]
                x = x_old.copy()
                continue
            mi = iq - neq
            Ri[:mi, mi] = rnd(-r / alpha)
            Ri[mi, mi] = rnd(np.array(1.0 / alpha))
            Ri[mi, :mi] = 0.0
            R_norm = max(R_norm, abs(alpha))
            act[ip] = True
            iq += 1
            break
        if done:
            break
    if trace is not None:
        trace["partial_steps"] = partial_steps
        trace["n_active"] = iq
    return dict(x=x, status=OPTIMAL, iters=it)
