// wbcqp_dense.hpp -- the NARROW seam of the drop-in boundary: a dense QP the way tsid's
// SolverHQuadProgFast hands it to eiquadprog (controller.cpp:247 -> solver_->solve(HQPData); pos_tracker.cpp:102
// solver_->resize(nVar, nEq, nIn)):
//        min 1/2 x'Hx + g'x   s.t.  CE x + ce0 = 0,  CI x + ci0 >= 0          (eiquadprog's convention)
// with H, CE, CI dense and of ANY structure -- no task stack behind it.  One workgroup per QP, J / R / the vectors in LDS,
// H only while it is factorised, CE and CI read row by row from HBM / L2.  This is the compatibility path for a caller that
// owns its own HQPData (a tsid SolverHQPBase subclass); the batched structured path
// (wbcqp_solve_batch) is the fast one and the one bench.py times.  Same Goldfarb-Idnani steps, status map and stopping rule as
// solve_one.  For n <= 80 the set-up is the structured kernels': H -> J = L^-T by the blocked elimination in registers
// (wbcqp_factor.hpp) and, for 1 <= neq <= 22, the equalities in one blocked phase (wbcqp_equality.hpp); beyond those sizes Cholesky
// in LDS, a triangular inverse and the equalities one by one (eiquadprog's own order), add_constraint in its one-reflector form.
#pragma once

#include "wbcqp_prims.hpp"
#include "wbcqp_activeset.hpp"
#include "wbcqp_equality.hpp"
#include "wbcqp_factor.hpp"

namespace wbcqp {
#ifdef __HIPCC__

struct DenseArgs {
    int n, neq, nin;            // nin = rows of CI as eiquadprog sees them (tsid: 2 x nIn two-sided rows)
    int ldj;                    // odd
    int ldb;                    // leading dimension of N = CE' / B = J0'N (blocked equality phase)
    int blocked_eq;             // 1 <= neq <= 22 and n <= 80: the equalities enter in one blocked phase (wbcqp_equality.hpp)
    int o_J, o_R, o_vec, o_int; // LDS layout (doubles)
    int o_eqw, o_eqt;           // blocked equality phase: N (n x ldb), then T / tau / rhs
    int max_iter;
    int count;
    const void *H, *g, *CE, *ce0, *CI, *ci0; // [count][...] row-major, the handle's dtype
    void *x, *objective;
    int *status, *iters, *n_active;
};

constexpr int kDenseMaxVars = 126, kDenseMaxIneq = 512;

template <typename TI>
__global__ __launch_bounds__(kThreads) void solve_dense_kernel(const DenseArgs a)
{
    extern __shared__ __align__(16) double lds[];
    const int tid = threadIdx.x;
    const size_t qp = blockIdx.x;
    const int n = a.n, neq = a.neq, nin = a.nin, ldj = a.ldj;
    Ctx c;
    c.S = nullptr;
    c.tid = tid;
    c.lane = tid & (kWave - 1);
    c.wave = uni(tid >> 6);
    c.rslot = 0;
    c.nblk = n;
    c.nv = n; c.na = 0; c.nc = 0; c.k = 0; c.n = n; c.nu = 0; c.neq = neq; c.nin2 = nin; c.ldj = ldj; c.ldm = 0; c.ldc = 0; c.ldb = 0;
    c.J = lds + a.o_J; c.R = lds + a.o_R;
    c.M = c.Jc = c.Ac = nullptr;
    {
        double* vec = lds + a.o_vec;
        c.h = vec + V_H * kSlot; c.x = vec + V_X * kSlot; c.np = vec + V_NP * kSlot; c.d = vec + V_D * kSlot;
        c.z = vec + V_Z * kSlot; c.xold = vec + V_XOLD * kSlot; c.r = vec + V_R * kSlot; c.u = vec + V_U * kSlot;
        c.uold = vec + V_UOLD * kSlot; c.q = vec + V_Q * kSlot; c.g = vec + V_G * kSlot; c.w = vec + V_W * kSlot;
        c.wrow = vec + V_WROW * kSlot; c.blb = vec + V_BLB * kSlot; c.bub = vec + V_BUB * kSlot; c.tl = vec + V_TL * kSlot;
        c.tu = vec + V_TU * kSlot; c.bc = vec + V_BC * kSlot; c.rdinv = vec + V_RDINV * kSlot; c.dinv = vec + V_DINV * kSlot;
        c.red = vec + V_RED * kSlot; c.prm = vec + V_PRM * kSlot; c.b1 = vec + V_B1 * kSlot; c.s = vec + V_S * kSlot;
        c.stash = vec + V_STASH * kSlot; c.part = vec + V_PART * kSlot;
    }
    c.eqw = a.blocked_eq ? lds + a.o_eqw : nullptr;
    c.eqt = a.blocked_eq ? lds + a.o_eqt : nullptr;
    c.ldb = a.ldb;
    int* ia = reinterpret_cast<int*>(lds + a.o_int);
    c.A = ia + kIntA; c.Aold = ia + kIntAold; c.gskip = ia + kIntGskip; c.iai = ia + kIntIai; c.iaexcl = ia + kIntIaexcl;
    c.meta = ia + kIntMeta;
    c.iq = 0;
    c.R_norm = 1.0;
    const TI* H = static_cast<const TI*>(a.H) + qp * (size_t)n * n;
    const TI* g = static_cast<const TI*>(a.g) + qp * (size_t)n;
    const TI* CE = static_cast<const TI*>(a.CE) + qp * (size_t)neq * n;
    const TI* ce0 = static_cast<const TI*>(a.ce0) + qp * (size_t)neq;
    const TI* CI = static_cast<const TI*>(a.CI) + qp * (size_t)nin * n;
    const TI* ci0 = static_cast<const TI*>(a.ci0) + qp * (size_t)nin;
    const double eps = 2.220446049250313e-16;
    const double inf = __builtin_huge_val();

    double c1, tr2 = 0.0;
    if (n <= 80) {
        // ---- H -> J = L^-T by the structured kernels' blocked elimination (wbcqp_factor.hpp): a 16 x 16 thread grid keeps the
        //      positions (ta + 16 u, te + 16 w), u <= w < 5, of H and of Y = U^-1 in registers, four pivots per barrier, only the pivot
        //      rows and columns travel through LDS (the J region, idle until J is written).  (Cholesky in LDS, then the triangular
        //      inverse, were 110 us of a Talos-sized QP.)  The lower triangle of H is what is read, as Eigen's LLT does.
        const int ta = tid >> 4, te = tid & 15;
        double h[5][5], y[5][5];
        double tr = 0.0;
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int w = 0; w < 5; ++w) {
                const int r = ta + 16 * u, q = te + 16 * w;
                const int hi = min(max(r, q), n - 1), lo = min(min(r, q), n - 1);
                const double v = (w >= u) ? (double)H[(size_t)hi * n + lo] : 0.0;
                h[u][w] = (r < n && q < n) ? v : ((r == q) ? 1.0 : 0.0);
                y[u][w] = 0.0;
                if (w >= u && r == q && r < n) tr += v;
            }
        for (int i = tid; i < n; i += kThreads) c.g[i] = (double)g[i];
        c1 = block_sum(c, tr);
        double* RB = c.J;
        double* YB = c.J + 2 * 5 * 16 * 4;
        publish_panel<4, 5, false, 0>(c, h, y, ta, te, 0, RB, YB);
        eliminate_block<4, 5, false, 0>(c, h, y, ta, te, (n + 3) & ~3, RB, YB, c.dinv, tid >= 128 && tid < 132, tid & 3);
        bsync(); // the panels are dead: the region becomes J
        for (int e = tid; e < n * ldj; e += kThreads) c.J[e] = 0.0;
        bsync(); // J is zero, every 1 / sqrt(pivot) is published
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int w = u; w < 5; ++w) {
                const int r = ta + 16 * u, q = te + 16 * w;
                if (q < n && r < q) c.J[r * ldj + q] = y[u][w] * c.dinv[q];
                else if (r == q && r < n) c.J[r * ldj + r] = c.dinv[r];
            }
        for (int i = tid; i < n; i += kThreads) tr2 += c.dinv[i];
    }
    else {
    // ---- H (lower triangle is read) into the J region, g; c1 = tr H
    double tr = 0.0;
    {
        int i = tid / n, j = tid - i * n; // (one division per thread; the running (row, column) follows the flat index from there)
        for (int e = tid; e < n * n; e += kThreads) {
            const double v = (double)H[e];
            if (j <= i) c.J[i * ldj + j] = v;
            if (i == j) tr += v;
            j += kThreads;
            while (j >= n) {
                j -= n;
                ++i;
            }
        }
    }
    for (int i = tid; i < n; i += kThreads) c.g[i] = (double)g[i];
    c1 = block_sum(c, tr);
    // ---- Cholesky H = L L' in place, right-looking, ONE barrier per column: the trailing update uses the unscaled pivot column,
    //      A(i, k) -= A(i, j) A(k, j) / A(j, j), on a 16 x 16 thread grid (no index divisions), and the columns are scaled by
    //      1 / sqrt(pivot) in one pass at the end (the first form scaled the column first: three barriers per column and an integer
    //      division per element).  A non-positive pivot ends in a NaN, which propagates like Eigen's LLT on a matrix that is not SPD
    {
        const int ta = tid >> 4, te = tid & 15;
        bsync();
        // a thread's tile of the trailing matrix, rows j + 1 + ta + 16 u, columns j + 1 + te + 16 w: every operand of the step in flight
        // before the first product (an element at a time is three dependent LDS round trips per element)
        auto step = [&](auto NUc, int j) __attribute__((always_inline)) {
            constexpr int NU = decltype(NUc)::value;
            const double inv = 1.0 / c.J[j * ldj + j];
            double li[NU], lk[NU], av[NU][NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = min(j + 1 + ta + 16 * u, n - 1), k = min(j + 1 + te + 16 * u, n - 1);
                li[u] = c.J[i * ldj + j];
                lk[u] = c.J[k * ldj + j];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int w = 0; w < NU; ++w) {
                    const int i = min(j + 1 + ta + 16 * u, n - 1), k = min(j + 1 + te + 16 * w, n - 1);
                    av[u][w] = c.J[i * ldj + min(k, i)];
                }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = j + 1 + ta + 16 * u;
                const double lij = li[u] * inv;
#pragma unroll
                for (int w = 0; w < NU; ++w) {
                    const int k = j + 1 + te + 16 * w;
                    if (i < n && k <= i) c.J[i * ldj + k] = fma(-lij, lk[w], av[u][w]);
                }
            }
        };
        for (int j = 0; j < n; ++j) {
            const int m = n - j - 1;
            if (m > 80) step(std::integral_constant<int, 8>{}, j);
            else if (m > 48) step(std::integral_constant<int, 5>{}, j);
            else if (m > 16) step(std::integral_constant<int, 3>{}, j);
            else step(std::integral_constant<int, 1>{}, j);
            bsync();
        }
        for (int j = c.wave; j < n; j += kWaves) { // a wave per column
            const double ljj = sqrt(c.J[j * ldj + j]);
            const double il = 1.0 / ljj;
            __builtin_amdgcn_wave_barrier();
            for (int i = j + c.lane; i < n; i += kWave) c.J[i * ldj + j] = (i == j) ? ljj : c.J[i * ldj + j] * il;
        }
        bsync();
    }
    // ---- J = L^-T (upper triangular): column q of J solves L' J(:, q) = e_q; thread per column, rows q .. 0.
    //      L is read from the lower triangle, J lands in a second array (the R region is too small: use part of it plus ...)
    //      -> done in place column by column is impossible (J overwrites L): stage L^-T through the vector area is too small
    //      too, so the inverse is formed into the UPPER triangle while L stays in the lower one; the diagonal of L moves to dinv.
    if (tid < n) c.dinv[tid] = 1.0 / c.J[tid * ldj + tid];
    bsync();
    // U = L' is upper triangular; X = U^-1 is upper triangular with X(q,q) = 1 / L(q,q) and, for i < q,
    // X(i,q) = -(sum_{p=i+1..q} U(i,p) X(p,q)) / U(i,i) = -(sum_p L(p,i) X(p,q)) dinv[i].  Column q belongs to a quad of lanes (the
    // sum over p in four strided parts, met by DPP), columns q and q + 64 in two rounds: L is read from the lower triangle (never
    // written here) and a quad writes only its own column above the diagonal -- read back by itself alone.
    // (the 64 LONGEST columns first, the n - 64 shortest in a second round: the rounds are as long as their longest column)
    const int qoff = max(n - kThreads / 4, 0);
    for (int round = 0; round < (qoff > 0 ? 2 : 1); ++round) {
        const int qb = round == 0 ? qoff : 0, qe = round == 0 ? n : qoff; // columns [qb, qe)
        const int q = qb + (tid >> 2), kc = tid & 3;
        const int qs = min(q, qe - 1);
        if (uni(qb + ((tid >> 6) << 4)) >= qe) continue;                      // (a wave without a column in this round)
        const int imax = uni(min(qb + (((tid >> 6) + 1) << 4) - 1, qe - 1)); // the wave's longest column
        for (int i = imax - 1; i >= 0; --i) {
            double acc = 0.0;
            if (i < qs) {
                for (int p = i + 1 + kc; p < qs; p += 4) acc = fma(c.J[p * ldj + i], c.J[p * ldj + qs], acc);
                if (kc == 0) acc = fma(c.J[qs * ldj + i], c.dinv[qs], acc); // p = q: L(q,i) X(q,q)
            }
            acc = quad_sum(acc);
            if (kc == 0 && i < q && q < qe) c.J[i * ldj + q] = -acc * c.dinv[i];
        }
    }
    bsync();
    // lower triangle := 0, diagonal := 1 / L(q,q): J = L^-T complete
    for (int i = tid >> 4; i < n; i += 16) {
        for (int j = tid & 15; j < i; j += 16) c.J[i * ldj + j] = 0.0;
        if ((tid & 15) == 0) {
            c.J[i * ldj + i] = c.dinv[i];
            tr2 += c.dinv[i];
        }
    }
    }
    const double c2 = block_sum(c, tr2);
    for (int i = tid; i < n + 2; i += kThreads) {
        c.u[i] = 0.0;
        c.A[i] = 0;
    }
    // ---- x = -H^-1 g = -J (J' g): two triangular matvecs, a thread per column, then per row (the first form solved L y = g, L' x = y
    //      on one wave: 2 n dependent steps with a division each -- 32 us of a Talos-sized QP)
    if (tid < n) {
        double acc = 0.0;
        for (int i = 0; i <= tid; ++i) acc = fma(c.J[i * ldj + tid], c.g[i], acc);
        c.d[tid] = acc;
    }
    bsync();
    double part = 0.0;
    if (tid < n) {
        const double* Jr = c.J + tid * ldj;
        double acc = 0.0;
        for (int j = tid; j < n; ++j) acc = fma(Jr[j], c.d[j], acc);
        c.x[tid] = -acc;
        part = 0.5 * c.g[tid] * (-acc);
    }
    double f_value = block_sum(c, part);

    int status = -2, iter = 0;
    // ---- equalities: in one blocked phase where it applies (N = CE', B = J0'N, Householder QR with J <- J Q in its shadow: the same
    //      (J, R, x, f) as neq add_constraint calls, wbcqp_equality.hpp; 18 equalities one by one were 92 us of a Talos-sized QP) ...
    if (a.blocked_eq) {
        const bool ok = equality_phase_blocked_t(
            c, f_value,
            [&](double* Nm) __attribute__((always_inline)) {
                const int e = tid & 31, k8 = tid >> 5;
                if (e < neq) {
                    TI v[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) v[i] = CE[(size_t)e * n + min(k8 + 8 * i, n - 1)];
#pragma unroll
                    for (int i = 0; i < 10; ++i)
                        if (k8 + 8 * i < n) Nm[(k8 + 8 * i) * a.ldb + e] = (double)v[i];
                }
            },
            [&](int e) __attribute__((always_inline)) { return (double)ce0[e]; });
        if (!ok) status = HQP_ERROR; // redundant equalities
    }
    // ---- ... else one by one (eiquadprog's order)
    for (int i = 0; i < neq && status == -2 && !a.blocked_eq; ++i) {
        if (tid < n) c.np[tid] = (double)CE[(size_t)i * n + tid];
        const double ce = (double)ce0[i];
        bsync();
        compute_d(c, 0, n);
        update_z_r(c, 0);
        double zz = 0.0, znp = 0.0, npx = 0.0, dn2 = 0.0;
        if (tid < n) {
            const double zv = c.z[tid], nv_ = c.np[tid];
            zz = zv * zv;
            if (tid >= c.iq) dn2 = c.d[tid] * c.d[tid];
            znp = zv * nv_;
            npx = nv_ * c.x[tid];
        }
        block_sum4(c, zz, znp, npx, dn2);
        double t2 = 0.0;
        if (fabs(zz) > eps) t2 = (-npx - ce) / znp;
        const int iq = c.iq;
        if (tid < n) c.x[tid] = fma(t2, c.z[tid], c.x[tid]);
        if (tid >= 128 && tid - 128 < iq) c.u[tid - 128] = fma(-t2, c.r[tid - 128], c.u[tid - 128]);
        if (tid == kThreads - 1) {
            c.u[iq] = t2;
            c.A[i] = -i - 1;
        }
        f_value += 0.5 * (t2 * t2) * znp;
        if (!add_constraint_hh(c, dn2)) status = HQP_ERROR; // redundant equalities
    }
    // ---- inequalities
    if (status == -2) {
        for (int i = tid; i < nin; i += kThreads) c.iai[i] = i;
        bsync();
        const double psi_tol = (double)nin * eps * c1 * c2 * 100.0;
        bool redo_l2 = false;
        while (status == -2) {
            ValIdx best{0.0, 0x7fffffff};
            if (!redo_l2) {
                ++iter;
                if (iter >= a.max_iter) {
                    status = HQP_MAX_ITER;
                    break;
                }
                for (int i = tid; i < c.iq; i += kThreads) {
                    c.uold[i] = c.u[i];
                    c.Aold[i] = c.A[i];
                }
                for (int i = tid; i < n; i += kThreads) c.xold[i] = c.x[i];
                double psi = 0.0;
                // eight lanes per row, consecutive columns across them: a load instruction touches one cache line per row (a lane per
                // row walked its own row: 64 lines per instruction); four rows per lane group in flight at once (a pass of 32 rows is one
                // round trip to L2: 244 rows one pass at a time were eight of them per evaluation)
                if (n <= 80) {
                    for (int i0 = 0; i0 < nin; i0 += kThreads / 2) {
                        const int l8 = tid & 7;
                        TI rv[4][10], cv[4];
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const int ir = min(i0 + 32 * p + (tid >> 3), nin - 1);
                            const TI* row = CI + (size_t)ir * n;
#pragma unroll
                            for (int q = 0; q < 10; ++q) rv[p][q] = row[min(l8 + 8 * q, n - 1)];
                            cv[p] = ci0[ir]; // (with the row, not behind its reduction)
                        }
                        double xv[10];
#pragma unroll
                        for (int q = 0; q < 10; ++q) xv[q] = (l8 + 8 * q < n) ? c.x[min(l8 + 8 * q, n - 1)] : 0.0;
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const int i = i0 + 32 * p + (tid >> 3);
                            double s0 = 0.0, s1 = 0.0;
#pragma unroll
                            for (int q = 0; q < 10; q += 2) {
                                s0 = fma((double)rv[p][q], xv[q], s0);
                                s1 = fma((double)rv[p][q + 1], xv[q + 1], s1);
                            }
                            const double sum = grp8_sum(s0 + s1);
                            if (l8 == 0 && i < nin) {
                                const double v = sum + (double)cv[p];
                                c.s[i] = v;
                                c.iaexcl[i] = 1;
                                psi += fmin(0.0, v);
                                if (v < 0.0 && c.iai[i] != -1) best = vi_min(best, ValIdx{v, i});
                            }
                        }
                    }
                }
                else
                for (int i0 = 0; i0 < nin; i0 += kThreads / 8) {
                    const int i = i0 + (tid >> 3), l8 = tid & 7;
                    const TI* row = CI + (size_t)min(i, nin - 1) * n;
                    double s0 = 0.0, s1 = 0.0;
                    int j = l8;
                    for (; j + 8 < n; j += 16) {
                        s0 = fma((double)row[j], c.x[j], s0);
                        s1 = fma((double)row[j + 8], c.x[j + 8], s1);
                    }
                    if (j < n) s0 = fma((double)row[j], c.x[j], s0);
                    const double sum = grp8_sum(s0 + s1);
                    if (l8 == 0 && i < nin) {
                        const double v = sum + (double)ci0[i];
                        c.s[i] = v;
                        c.iaexcl[i] = 1;
                        psi += fmin(0.0, v);
                        if (v < 0.0 && c.iai[i] != -1) best = vi_min(best, ValIdx{v, i});
                    }
                }
                psi = block_sum(c, psi);
                best = block_argmin(c, best);
                if (fabs(psi) <= psi_tol) {
                    status = HQP_OPTIMAL;
                    break;
                }
            }
            else {
                for (int i = tid; i < nin; i += kThreads) {
                    const double sv = c.s[i];
                    if (sv < 0.0 && c.iai[i] != -1 && c.iaexcl[i]) best = vi_min(best, ValIdx{sv, i});
                }
                best = block_argmin(c, best);
                redo_l2 = false;
            }
            if (best.v >= 0.0) {
                status = HQP_OPTIMAL;
                break;
            }
            const int ip = best.i;
            if (tid < n) c.np[tid] = (double)CI[(size_t)ip * n + tid];
            if (tid == kThreads - 1) {
                c.u[c.iq] = 0.0;
                c.A[c.iq] = ip;
            }
            bsync();
            while (true) {
                const int iq = c.iq;
                compute_d(c, 0, n);
                update_z_r(c, neq);
                double zz = 0.0, znp = 0.0, dn2 = 0.0, dummy = 0.0;
                if (tid < n) {
                    const double zv = c.z[tid];
                    zz = zv * zv;
                    znp = zv * c.np[tid];
                    if (tid >= iq) dn2 = c.d[tid] * c.d[tid];
                }
                block_sum4(c, zz, znp, dn2, dummy);
                ValIdx bt{inf, 0x7fffffff};
                for (int kk = neq + tid; kk < iq; kk += kThreads) {
                    const double rk = c.r[kk];
                    if (rk > 0.0) bt = vi_min(bt, ValIdx{c.u[kk] / rk, kk});
                }
                bt = block_argmin(c, bt);
                const double t1 = bt.v;
                const int l = (t1 < inf) ? c.A[bt.i] : 0;
                const double sip = c.s[ip], uiq = c.u[iq];
                const double t2 = (fabs(zz) > eps) ? (-sip / znp) : inf;
                const double t = fmin(t1, t2);
                if (t >= inf) {
                    status = HQP_INFEASIBLE; // eiquadprog UNBOUNDED (dual) -> tsid INFEASIBLE
                    break;
                }
                bsync(); // everyone has read s[ip], u[iq], A[.] before they change
                if (t2 >= inf) {
                    for (int j = neq + tid; j < iq; j += kThreads) c.u[j] = fma(-t, c.r[j], c.u[j]);
                    if (tid == kThreads - 1) {
                        c.u[iq] = uiq + t;
                        c.iai[l] = l;
                    }
                    bsync();
                    delete_constraint(c, l);
                    continue;
                }
                f_value += t * znp * (0.5 * t + uiq);
                if (tid < n) c.x[tid] = fma(t, c.z[tid], c.x[tid]);
                if (tid >= 128 + neq && tid - 128 < iq) c.u[tid - 128] = fma(-t, c.r[tid - 128], c.u[tid - 128]);
                if (tid == kThreads - 1) c.u[iq] = uiq + t;
                if (t == t2) {
                    bsync();
                    if (add_constraint_hh(c, dn2)) {
                        if (tid == 0) c.iai[ip] = -1;
                        bsync();
                    }
                    else {
                        if (tid == 0) c.iaexcl[ip] = 0;
                        bsync();
                        delete_constraint(c, ip);
                        for (int i = tid; i < nin; i += kThreads) c.iai[i] = i;
                        bsync();
                        for (int i = tid; i < c.iq; i += kThreads) {
                            const int av = c.Aold[i];
                            c.A[i] = av;
                            if (av >= 0) c.iai[av] = -1;
                            c.u[i] = c.uold[i];
                        }
                        for (int i = tid; i < n; i += kThreads) c.x[i] = c.xold[i];
                        bsync();
                        redo_l2 = true;
                    }
                    break;
                }
                if (tid == 0) c.iai[l] = l;
                bsync();
                delete_constraint(c, l);
                {
                    double p2 = 0.0;
                    if (tid < n) p2 = c.np[tid] * c.x[tid];
                    p2 = block_sum(c, p2);
                    if (tid == 0) c.s[ip] = p2 + (double)ci0[ip];
                    bsync();
                }
            }
        }
    }
    bsync();
    TI* xo = static_cast<TI*>(a.x) + qp * (size_t)n;
    for (int i = tid; i < n; i += kThreads) xo[i] = (TI)c.x[i];
    if (tid == 0) {
        a.status[qp] = status;
        a.iters[qp] = iter;
        if (a.objective) static_cast<TI*>(a.objective)[qp] = (TI)f_value;
        if (a.n_active) a.n_active[qp] = c.iq;
    }
}

#endif // __HIPCC__
} // namespace wbcqp
