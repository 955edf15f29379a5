// wbcqp_small.hpp -- ONE WAVEFRONT PER QP for the small structures (n <= 16: Franka n = 9, Tiago n = 12): four QPs per
// 256-thread workgroup, one per wave, no workgroup barrier anywhere, H in registers -- the launch shape north_star names.
//
// solve_one / solve_one_compact give a QP four waves and synchronise them a few times per phase; that pays for n >= 60
//  and wastes a workgroup on a 9-variable unconstrained QP (round 2: Franka 45 M QP/s on the GPU against 21 M
// on 16 CPU threads).  Here every phase is wave-local: LDS operations of one wave execute in order, so a value written by one
// lane is read by another with no barrier; reductions are DPP + readlane; what decides control flow comes out of those
// reductions, i.e. out of SGPRs, and every branch is scalar.
//
// Same algorithm and reference contract as solve_one_compact (controller.cpp:244-251): H = sum w A'A + 1e-8 I, g; Cholesky
// by the blocked elimination on an 8 x 8 lane grid (eliminate_block, the force blocks' instance); J = U^-1; x0 = -J J'g; the
// Goldfarb-Idnani loop in eiquadprog's order with the inverse of R carried (ri_matvec / drop_coefficients / rotate_row of
// wbcqp_compact.hpp); tau = h + M dv.  Eligibility (host, small_ok): fixed base without contacts (nu = nc = neq = 0), n = nv <= 16,
// at most 16 dense rows / selection rows / tasks / bounds, inequality rows are bounds only (nin2 <= 32).
#pragma once

#include "wbcqp_prims.hpp"
#include "wbcqp_factor.hpp"
#include "wbcqp_compact.hpp"

namespace wbcqp {
#ifdef __HIPCC__

namespace sm {
constexpr int LDJ = 17;
// per-wave LDS map (doubles)
constexpr int J = 0 /* 16 x 17; the staged task rows (16 x 16) until J exists */, WB = 272 /* (weight, rhs) per dense row */, RB = 304, YB = 368,
              RI = 432 /* packed inverse of R: roff(17) = 170 */, PRM = 604 /* 2 x 16 + over-read of rotate_row */, X = 652, XOLD = 668, Z = 684,
              D0 = 700 /* + over-read */, D1 = 724, G = 748, R = 764, U = 780 /* 18 */, UOLD = 800, DINV = 820, DSEL = 836, GSEL = 852, W = 868,
              B1 = 884 /* 32 */, BLB = 916, BUB = 932, IA = 948 /* ints: A (20), Aold (20) */, COUNT = 968;
} // namespace sm

template <typename TI>
__device__ __forceinline__ void solve_one_wave(const GroupArgs<TI>& ga, const DevStruct& S, const int b, double* L, const int lane)
{
    const int n = S.n, nd = S.n_dense, nsel = S.n_sel, nb = S.n_bound, nin2 = S.nin2, r1 = S.r1, nt = S.n_tasks;
    const size_t qp = (size_t)b;
    double* const Jm = L + sm::J;
    double* const RI = L + sm::RI;
    double* const PRM = L + sm::PRM;
    double* const X = L + sm::X;
    double* const Z = L + sm::Z;
    double* const G = L + sm::G;
    double* const U = L + sm::U;
    int* const A = reinterpret_cast<int*>(L + sm::IA);
    int* const Aold = A + 20;
    Ctx c; // only what the shared helpers read
    c.lane = lane;
    c.tid = lane;
    c.wave = 0;
    c.neq = 0;
    c.n = n;
    c.r = L + sm::R;
    const double eps = 2.220446049250313e-16;
    const double inf = __builtin_huge_val();

    // ---------------- phase 0: the record's loads in flight, then landed ----------------
    const int lenA = nd * n, lenM = n * (n + 1) / 2;
    const int rrow = min(lane >> 2, n - 1), q4 = lane & 3; // decode role: four lanes per row of M
    TI mrow[4];
    {
        const TI* pM = ga.M + qp * lenM;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = min(q4 + 4 * u, n - 1);
            const int hi = max(rrow, cc), lo = min(rrow, cc);
            mrow[u] = pM[hi * (hi + 1) / 2 + lo];
        }
    }
    const TI vh = ga.h[qp * n + rrow];
    {
        TI va[4];
        const TI* pA = ga.A + qp * (size_t)lenA;
        if (lenA > 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) va[u] = pA[min(lane + 64 * u, lenA - 1)];
        }
        const TI vb1 = ga.b1[qp * r1 + min(lane, r1 - 1)];
        const TI vw = ga.w[qp * nt + min(lane, nt - 1)];
        TI vbl = TI(0), vbu = TI(0);
        if (nb > 0) {
            vbl = ga.blb[qp * nb + min(lane, nb - 1)];
            vbu = ga.bub[qp * nb + min(lane, nb - 1)];
        }
        const int drt = (nd > 0) ? S.dense_row_task[min(lane, nd - 1)] : 0;
        int selc = 0, selt = 0;
        if (nsel > 0) {
            selc = S.sel_col[min(lane, nsel - 1)];
            selt = S.sel_task[min(lane, nsel - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) Jm[lane + 64 * u] = 0.0; // the staging area: rows past nd and columns past n stay zero
        if (lane < 16) {
            (L + sm::DSEL)[lane] = 0.0;
            (L + sm::GSEL)[lane] = 0.0;
        }
        if (lenA > 0) {
            const int inv = (65536 + n - 1) / n; // e / n for e < 256, n <= 16 (exact: see the header of tools/..., checked on the host side of the tests)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = lane + 64 * u;
                if (e < lenA) {
                    const int row = (e * inv) >> 16, col = e - row * n;
                    Jm[row * 16 + col] = (double)va[u];
                }
            }
        }
        if (lane < r1) (L + sm::B1)[lane] = (double)vb1;
        if (lane < nt) (L + sm::W)[lane] = (double)vw;
        if (lane < nb) {
            (L + sm::BLB)[lane] = (double)vbl;
            (L + sm::BUB)[lane] = (double)vbu;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < nd) {
            (L + sm::WB)[2 * lane] = (L + sm::W)[drt];
            (L + sm::WB)[2 * lane + 1] = (L + sm::B1)[lane];
        }
        if (lane < nsel) { // selection rows (posture): H(c,c) += w, g(c) -= w b  (distinct columns)
            const double wt = (L + sm::W)[selt];
            (L + sm::DSEL)[selc] = wt;
            (L + sm::GSEL)[selc] = wt * (L + sm::B1)[nd + lane];
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---------------- phases 1-2: H on an 8 x 8 lane grid (2 x 2 positions per lane), elimination, J = U^-1 ----------------
    const int la = lane >> 3, le = lane & 7;
    double c1, c2;
    {
        double h[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, y[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        double g0 = 0.0, g1 = 0.0;
        const double* As = Jm;
        const double* WB = L + sm::WB;
        for (int t = 0; t < nd; ++t) {
            const double ar0 = As[t * 16 + la], ar1 = As[t * 16 + la + 8];
            const double ac0 = As[t * 16 + le], ac1 = As[t * 16 + le + 8];
            const double2v wb = ld2(WB + 2 * t);
            const double w0 = ac0 * wb.x, w1 = ac1 * wb.x;
            h[0][0] = fma(ar0, w0, h[0][0]);
            h[0][1] = fma(ar0, w1, h[0][1]);
            h[1][1] = fma(ar1, w1, h[1][1]);
            g0 = fma(w0, wb.y, g0);
            g1 = fma(w1, wb.y, g1);
        }
        if (la == 0) {
            G[le] = -g0 - (L + sm::GSEL)[le];
            G[le + 8] = -g1 - (L + sm::GSEL)[le + 8];
        }
        double trace = 0.0;
        if (la == le) {
            if (la < n) {
                h[0][0] += (L + sm::DSEL)[la] + S.hessian_reg;
                trace += h[0][0];
            }
            if (la + 8 < n) {
                h[1][1] += (L + sm::DSEL)[la + 8] + S.hessian_reg;
                trace += h[1][1];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int r = la + 8 * u, q = le + 8 * w;
                if (r >= n || q >= n) h[u][w] = (r == q) ? 1.0 : 0.0;
            }
        double* RBp = L + sm::RB;
        double* YBp = L + sm::YB;
        double* dinv = L + sm::DINV;
        publish_panel<3, 2, true, 0>(c, h, y, la, le, 0, RBp, YBp);
        eliminate_block<3, 2, true, 0>(c, h, y, la, le, (n + 3) & ~3, RBp, YBp, dinv, lane < 4, lane & 3);
        __builtin_amdgcn_wave_barrier();
        // the staged rows are dead: the region becomes J (row-major, ld 17)
#pragma unroll
        for (int u = 0; u < 5; ++u)
            if (lane + 64 * u < 16 * sm::LDJ) Jm[lane + 64 * u] = 0.0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int w = u; w < 2; ++w) {
                const int r = la + 8 * u, q = le + 8 * w;
                if (q < n && r < q) Jm[r * sm::LDJ + q] = y[u][w] * dinv[q];
                else if (r == q && r < n) Jm[r * sm::LDJ + r] = dinv[r];
            }
        c1 = wave_sum(trace);
        c2 = wave_sum(lane < n ? dinv[lane] : 0.0);
    }

    // ---------------- x = -J (J' g); f = 0.5 g'x ----------------
    double f_value;
    {
        const int col = lane >> 2; // d_c = sum_{r <= c} J(r,c) g(r): four lanes per column
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = q4 + 4 * u;
            const double jv = Jm[min(r, n - 1) * sm::LDJ + min(col, n - 1)], gv = G[min(r, n - 1)];
            a = fma((r <= col && col < n) ? jv : 0.0, gv, a);
        }
        a = quad_sum(a);
        if (q4 == 0 && col < n) (L + sm::D0)[col] = a;
        double zv = 0.0; // x_r = -sum_{c >= r} J(r,c) d(c): four lanes per row
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = q4 + 4 * u;
            const double jv = Jm[min(col, n - 1) * sm::LDJ + min(cc, n - 1)], dv = (L + sm::D0)[min(cc, n - 1)];
            zv = fma((cc >= col && cc < n && col < n) ? jv : 0.0, dv, zv);
        }
        zv = quad_sum(zv);
        double part = 0.0;
        if (q4 == 0 && col < n) {
            X[col] = -zv;
            part = 0.5 * G[col] * (-zv);
        }
        f_value = wave_sum(part);
    }

    int status = -2, iter = 0, iq = 0;
    bool act_i = false; // this lane's row is in the active set
    // ---------------- inequality loop (bounds only): GI steps 1, 2, 2a-2c, one wave ----------------
    if (nin2 > 0) {
        int ocol = 0;
        double osg = 0.0, oci0 = 0.0;
        const bool own = lane < nin2;
        if (own) {
            const int mt = S.rowmeta[lane];
            const int orr = (mt >> 3) & 255;
            const bool neg = (mt >> 2) & 1;
            ocol = (mt >> 15) & 255;
            osg = neg ? -1.0 : 1.0;
            oci0 = neg ? (L + sm::BUB)[orr] : -(L + sm::BLB)[orr];
        }
        bool excl_i = true;
        double s_i = 0.0, R_norm = 1.0;
        const double psi_tol = (double)nin2 * eps * c1 * c2 * 100.0;
        double* dcur = L + sm::D0;
        double* dalt = L + sm::D1;
        double* const XOLD = L + sm::XOLD;
        double* const UOLD = L + sm::UOLD;
        bool redo = false;
        while (status == -2) {
            if (!redo) {
                ++iter;
                if (iter >= S.max_iter) {
                    status = HQP_MAX_ITER;
                    break;
                }
                if (lane < iq) {
                    UOLD[lane] = U[lane];
                    Aold[lane] = A[lane];
                }
                if (lane < n) XOLD[lane] = X[lane];
                excl_i = true;
                s_i = own ? fma(osg, X[ocol], oci0) : 0.0;
                const double psi = wave_sum(fmin(0.0, s_i));
                if (fabs(psi) <= psi_tol) {
                    status = HQP_OPTIMAL;
                    break;
                }
            }
            redo = false;
            ValIdx best{0.0, 0x7fffffff};
            if (own && s_i < 0.0 && !act_i && excl_i) best = ValIdx{s_i, lane};
            best = wave_argmin(best);
            if (!(best.v < 0.0)) {
                status = HQP_OPTIMAL;
                break;
            }
            const int ip = best.i;
            double sip = best.v;
            const int pcol = __builtin_amdgcn_readlane(ocol, ip);
            const double psg = bcast_lane(osg, ip);
            // d = J'n = +- row pcol of J; z = J2 d2; r = Ri d1; the reductions of step 2b
            const double dv_lane = (lane < n) ? psg * Jm[pcol * sm::LDJ + lane] : 0.0;
            if (lane < n) dcur[lane] = dv_lane;
            const int zr = lane >> 2;
            double zv = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = iq + q4 + 4 * u;
                const double jv = Jm[min(zr, n - 1) * sm::LDJ + min(cc, n - 1)], dd = dcur[min(cc, n - 1)];
                zv = fma((cc < n) ? jv : 0.0, dd, zv);
            }
            zv = quad_sum(zv);
            if (q4 == 0 && zr < n) Z[zr] = zv;
            double zz = wave_sum((q4 == 0 && zr < n) ? zv * zv : 0.0);
            double dn2 = wave_sum((lane >= iq && lane < n) ? dv_lane * dv_lane : 0.0);
            double znp = dn2; // z'n = |d2|^2
            double rl = ri_matvec(c, RI, iq, dcur, 1.0);
            ValIdx bt{inf, 0x7fffffff};
            if (lane < iq && rl > 0.0) bt = ValIdx{ratio_pos(U[lane], rl), lane};
            bt = wave_argmin(bt);
            double t1 = bt.v;
            int lpos = bt.i;
            double uiq = 0.0;
            while (true) {
                const double t2 = (fabs(zz) > eps) ? ratio_pos(-sip, znp) : inf;
                const double t = fmin(t1, t2);
                if (t >= inf) {
                    status = HQP_INFEASIBLE;
                    break;
                }
                if (t2 < inf) f_value += t * znp * (0.5 * t + uiq);
                if (t2 < inf && t == t2) {
                    // full step: add ip with one reflector H = I - tau v v' (v = d[iq:] - alpha e_0)
                    const double diq = dcur[iq];
                    double alpha = diq, v0 = 0.0, tau = 0.0;
                    const bool reflect = (iq + 1 < n && dn2 > 0.0);
                    if (reflect) {
                        const double inx = rsqrt(dn2);
                        const double nx = dn2 * inx;
                        alpha = (diq >= 0.0) ? -nx : nx;
                        v0 = diq - alpha;
                        tau = fast_rcp(fma(nx, fabs(diq), dn2));
                    }
                    if (!(fabs(alpha) > eps * R_norm)) {
                        // numerically dependent: back to the saved iterate, the row is excluded, pick another
                        if (lane == ip) excl_i = false;
                        if (lane < iq) {
                            A[lane] = Aold[lane];
                            U[lane] = UOLD[lane];
                        }
                        if (lane < n) X[lane] = XOLD[lane];
                        act_i = false;
                        for (int j = 0; j < iq; ++j)
                            if (A[j] == lane) act_i = true;
                        redo = true;
                        break;
                    }
                    if (reflect) { // J(k, c) -= w_k v_c, w_k = tau (z_k - alpha J(k,iq)): four lanes per row
                        const int kr = min(zr, n - 1);
                        double* Jk = Jm + kr * sm::LDJ;
                        const double wk = tau * (Z[kr] - alpha * Jk[iq]);
                        double jj[4], dd[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int cc = min(iq + q4 + 4 * u, n - 1);
                            jj[u] = Jk[cc];
                            dd[u] = dcur[cc];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int cc = iq + q4 + 4 * u;
                            if (cc < n && zr < n) Jk[cc] = fma(-wk, (cc == iq) ? v0 : dd[u], jj[u]);
                        }
                    }
                    const double ralpha = fast_rcp(alpha);
                    if (lane < iq) {
                        RI[roff(iq) + lane] = -rl * ralpha; // the new column of the inverse: [-r / alpha; 1 / alpha]
                        U[lane] = fma(-t, rl, U[lane]);
                    }
                    if (lane == iq) {
                        RI[roff(iq) + iq] = ralpha;
                        U[iq] = uiq + t;
                        A[iq] = ip;
                    }
                    if (lane < n) X[lane] = fma(t, Z[lane], X[lane]);
                    if (lane == ip) act_i = true;
                    ++iq;
                    R_norm = fmax(R_norm, fabs(alpha));
                    break;
                }
                // dual / partial step: move, then drop position p (rank-one updates of d, z, r: wbcqp_compact.hpp)
                const bool primal = t2 < inf;
                const int p = lpos, mi = iq, Lr = iq - 1 - p;
                const int ldrop = A[p];
                if (lane < n && primal) X[lane] = fma(t, Z[lane], X[lane]);
                if (lane < iq) U[lane] = fma(-t, rl, U[lane]);
                if (lane == ldrop) act_i = false;
                if (primal) sip = fma(t, znp, sip);
                uiq += t;
                drop_coefficients(c, RI, mi, p, PRM);
                if (lane < n && (lane < p || lane >= iq)) dalt[lane] = dcur[lane];
                // lanes 0..15: rows of J; lanes 16..31: rows of Ri (one wave: lane i + 1 writes what lane i read a step earlier);
                // every lane carries d's entry along, so delta needs no exchange
                const bool jlane = lane < 16, rlane = lane >= 16 && lane < 32;
                const int ri = lane - 16;
                const bool rrow2 = rlane && ri < mi, has = rrow2 && ri != p;
                const int i2 = ri - ((ri > p) ? 1 : 0);
                const int kr = min(lane, n - 1);
                double* Jk = Jm + kr * sm::LDJ + p;
                double tj = jlane ? Jk[0] : ((rrow2 && ri <= p) ? RI[roff(p) + min(ri, p)] : 0.0);
                double dch = dcur[p];
                rotate_row(PRM, dcur, p, Lr, tj, dch,
                           [&](int jj) {
                               const int j = p + jj;
                               const double vj = Jk[jj], vr = RI[roff(j) + min(max(ri, 0), j)];
                               return jlane ? vj : ((rrow2 && ri <= j) ? vr : 0.0);
                           },
                           [&](int jj, double v) {
                               if (jlane && lane < n) Jk[jj] = v;
                               if (has && i2 <= p + jj) RI[roff(p + jj) + i2] = v;
                           },
                           [&](int jj, double v) { if (lane == 0) dalt[p + jj] = v; });
                const double delta = dch;
                double zq = 0.0;
                if (jlane && lane < n) {
                    Jk[Lr] = tj;
                    const double zn = fma(delta, tj, Z[lane]);
                    Z[lane] = zn;
                    zq = zn * zn;
                }
                if (lane == 0) dalt[iq - 1] = delta;
                {
                    const int rs = min(max(ri, 0), mi - 1);
                    const double rn = fma(-delta, tj, (L + sm::R)[rs]);
                    const double uu = U[rs];
                    const int aa = A[rs];
                    if (has) {
                        (L + sm::R)[i2] = rn;
                        U[i2] = uu;
                        A[i2] = aa;
                    }
                }
                zz = wave_sum(zq);
                znp = fma(delta, delta, znp);
                dn2 = fma(delta, delta, dn2);
                --iq;
                rl = (lane < iq) ? (L + sm::R)[lane] : 0.0;
                bt = ValIdx{inf, 0x7fffffff};
                if (lane < iq && rl > 0.0) bt = ValIdx{ratio_pos(U[lane], rl), lane};
                bt = wave_argmin(bt);
                t1 = bt.v;
                lpos = bt.i;
                double* sw = dcur;
                dcur = dalt;
                dalt = sw;
            }
        }
    }
    else {
        iter = 1;
        status = HQP_OPTIMAL;
    }

    // ---------------- decode + write-out: tau = h + M dv (getActuatorForces, fixed base) ----------------
    {
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = q4 + 4 * u;
            acc = fma((cc < n) ? (double)mrow[u] : 0.0, X[min(cc, n - 1)], acc);
        }
        acc = quad_sum(acc);
        if (lane < n) ga.x[qp * n + lane] = (TI)X[lane];
        if (S.na > 0 && q4 == 0 && (lane >> 2) < S.na) ga.tau[qp * S.na + (lane >> 2)] = (TI)((double)vh + acc);
        if (ga.amask) { // nin2 <= 32: the rows active at the solution are one ballot (the hint itself is not taken here: these QPs are
            // a handful of iterations long)
            const unsigned long long m = __ballot(act_i && status == HQP_OPTIMAL);
            if (lane < 8) ga.amask[qp * 8 + lane] = lane == 0 ? (unsigned)(m & 0xffffffffull) : 0u;
        }
        if (lane == 0) {
            ga.status[qp] = status;
            ga.iters[qp] = iter;
            if (ga.objective) ga.objective[qp] = (TI)f_value;
            if (ga.n_active) ga.n_active[qp] = iq;
        }
    }
}

// grid = ceil(total / 4) workgroups of four waves; wave w of workgroup g solves QP 4 g + w of the launch (groups in table order)
template <typename TI>
__global__ __launch_bounds__(kThreads) void solve_small_kernel(const GroupTable<TI> tab, const int total)
{
    extern __shared__ __align__(16) double lds[];
    const int wave = uni((int)threadIdx.x >> 6), lane = threadIdx.x & (kWave - 1);
    int b = (int)blockIdx.x * kWaves + wave, gi = 0;
    if (b >= total) return;
    while (gi + 1 < tab.n && b >= tab.g[gi].count) {
        b -= tab.g[gi].count;
        ++gi;
    }
    const GroupArgs<TI>& ga = tab.g[gi];
    solve_one_wave<TI>(ga, ga.st, b, lds + wave * sm::COUNT, lane);
}

#endif // __HIPCC__
} // namespace wbcqp
