// wbcqp_equality.hpp -- the blocked equality phase: B = J0'N, Householder QR with the columns in registers and
// J <- J Q in its shadow, y = R'^-1 rhs, x = x0 + J1 y.
#pragma once

#include "wbcqp_prims.hpp"

namespace wbcqp {
#ifdef __HIPCC__

// ------------------------------------------------------------------------------------------------
// Householder QR of B (n x m) with J <- J Q in its shadow; every vector on a quad of its own.  A column of B and a row of J
// take the same operation per reflector, x <- x - tau (x . v) v: m columns + n rows = up to 102 vectors, four lanes each, lane
// kc of a vector keeps the row pairs (2 kc + 8 t, + 1), t < NT (ten; eight in the instantiation of a stack with n <= 64: the pairs left out hold
// rows past n, zeros that add nothing to any sum -- same bits), in registers for the whole factorisation.  Per step only the
// reflector travels: the quad of column j leaves v_j (zeros above row j, v0 on it) and (tau_j, alpha_j) in LDS, every other
// vector reads it ONCE (10 x 16 bytes per lane), reduces its dot product over its quad by DPP and updates its registers; the
// quad of column j + 1 goes on to the next reflector.  One barrier per column, no reloads or stores of the trailing matrix, and
// J <- J Q costs no phase of its own (the compact-WY route -- W = J V, W T, J - W T V' -- is three LDS GEMM phases, 20 k cycles).
// Round 1: vectors 0..63 on the 64 quads (the columns of B first); round 2: the remaining rows of J on the LAST quads (waves 2
// and 3), so the waves that carry the columns' critical path have one round only.  (Round 2's form, qr_resident, kept a lane
// pair per row of J and the columns on the first lanes: wave 1 ran the column path and the row path one after the other in 16
// of 18 steps and the J lanes read the reflector twice; this form returned 36 registers, not time: a step is as long as the
// chain of the next reflector on ONE wave -- measured with stamps inside a step: 1.5 k of its 2.1 k cycles, the rest is the
// drain of its stores and the barrier; the other waves need 0.7 k.  Measured and not kept: the barrier replaced by a counter
// of published reflectors with a slot per reflector, so that the rows of J trail the columns -- same time, the chain does not
// wait for the other waves.)  Requires n <= 80, m <= 22, m + n <= 102.
// On return: J = J0 Q in LDS, the packed R and 1/R(j,j).  Returns false when a column is (numerically) dependent.
// ------------------------------------------------------------------------------------------------
template <int NT = 10>
__device__ __forceinline__ bool qr_unified(Ctx& c, const double* Bm, double* vbuf, double* sc)
{
    const int n = c.n, m = c.neq, ldb = c.ldb, ldj = c.ldj, tid = c.tid;
    const int e = tid >> 2, kc = tid & 3;
    const int nvec = m + n, r2 = max(nvec - 64, 0);
    const bool col1 = e < m;                 // round 1: a column of B ...
    const int jr1 = e - m;                   // ... or row jr1 of J
    const bool row1 = !col1 && jr1 < n;
    const bool has2 = e >= 64 - r2;          // round 2: row jr2 of J
    const int jr2 = 64 - m + (e - (64 - r2));
    static_assert(NT == 8 || NT == 10, "row pairs per lane: ten cover n <= 80, eight n <= 64");
    double b[NT][2], b2[NT][2];
    {
        const int es = col1 ? e : 0;
        const double* Jr = c.J + (row1 ? jr1 : 0) * ldj;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = 2 * kc + 8 * t + i, kk = min(k, n - 1);
                const double vb = Bm[kk * ldb + es], vj = Jr[kk];
                b[t][i] = (k < n && (col1 || row1)) ? (col1 ? vb : vj) : 0.0;
            }
        const double* Jr2 = c.J + (has2 ? jr2 : 0) * ldj;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = 2 * kc + 8 * t + i;
                const double vj = Jr2[min(k, n - 1)];
                b2[t][i] = (k < n && has2) ? vj : 0.0;
            }
    }
    double my_alpha = 1.0;
    // reflector of column jn from the registers of its quad (call under e == jn)
    auto prepare_t = [&](auto T0c, int jn) __attribute__((always_inline)) {
        constexpr int T0 = decltype(T0c)::value;
        const int row0 = 2 * kc + 8 * T0;
        const double e0 = (row0 >= jn) ? b[T0][0] : 0.0, e1 = (row0 + 1 >= jn) ? b[T0][1] : 0.0;
        double sq0 = e0 * e0, sq1 = e1 * e1, sq2 = 0.0, sq3 = 0.0;
#pragma unroll
        for (int t = T0 + 1; t + 1 < NT; t += 2) {
            sq0 = fma(b[t][0], b[t][0], sq0);
            sq1 = fma(b[t][1], b[t][1], sq1);
            sq2 = fma(b[t + 1][0], b[t + 1][0], sq2);
            sq3 = fma(b[t + 1][1], b[t + 1][1], sq3);
        }
        if constexpr (((NT - (T0 + 1)) & 1) != 0) {
            sq0 = fma(b[NT - 1][0], b[NT - 1][0], sq0);
            sq1 = fma(b[NT - 1][1], b[NT - 1][1], sq1);
        }
        double x0 = (row0 == jn) ? b[T0][0] : ((row0 + 1 == jn) ? b[T0][1] : 0.0);
        const double nrm = quad_sum((sq0 + sq1) + (sq2 + sq3));
        x0 = quad_sum(x0);
        const double inx = rsqrt(nrm);
        const double nx = (nrm > 0.0) ? nrm * inx : 0.0; // exactly dependent column: alpha = 0 -> reported as redundant
        const double alpha = (x0 >= 0.0) ? -nx : nx;
        const double v0 = x0 - alpha;
        const double tj = fast_rcp(fma(nx, fabs(x0), nrm)); // 2 / v'v
        my_alpha = alpha;
        double* vb = vbuf + (jn & 1) * 80 + 2 * kc;
        if (row0 == jn) b[T0][0] = v0;
        if (row0 + 1 == jn) b[T0][1] = v0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double2v o;
            if (t < T0) {
                o.x = 0.0;
                o.y = 0.0;
            }
            else if (t == T0) {
                o.x = (row0 >= jn) ? b[T0][0] : 0.0;
                o.y = (row0 + 1 >= jn) ? b[T0][1] : 0.0;
            }
            else {
                o.x = b[t][0];
                o.y = b[t][1];
            }
            *reinterpret_cast<double2v*>(__builtin_assume_aligned(vb + 8 * t, 16)) = o;
        }
        if (kc == 0) {
            sc[(jn & 1) * 2] = tj;
            sc[(jn & 1) * 2 + 1] = alpha;
        }
    };
    auto prepare = [&](int jn) __attribute__((always_inline)) {
        switch (jn >> 3) { // jn < 32
        case 0: prepare_t(std::integral_constant<int, 0>{}, jn); break;
        case 1: prepare_t(std::integral_constant<int, 1>{}, jn); break;
        case 2: prepare_t(std::integral_constant<int, 2>{}, jn); break;
        default: prepare_t(std::integral_constant<int, 3>{}, jn); break;
        }
    };
    auto apply = [&](double (&x)[NT][2], const double2v (&v)[NT], double tj) __attribute__((always_inline)) {
        double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
            d0 = fma(v[t].x, x[t][0], d0);
            d1 = fma(v[t].y, x[t][1], d1);
            d2 = fma(v[t + 1].x, x[t + 1][0], d2);
            d3 = fma(v[t + 1].y, x[t + 1][1], d3);
        }
        const double coef = quad_sum((d0 + d1) + (d2 + d3)) * tj;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            x[t][0] = fma(-coef, v[t].x, x[t][0]);
            x[t][1] = fma(-coef, v[t].y, x[t][1]);
        }
    };
    if (e == 0) prepare(0);
    // A (numerically) dependent column does not leave the loop: it is remembered and reported at the end.  With an early return in the
    // loop every step began with an LDS round trip of its own for (tau, alpha) before the reflector's ten reads were issued; without it
    // alpha is off the critical path altogether and tau travels with the reflector.  What the remaining steps compute after a bad pivot
    // (infinities, NaNs) is never used -- the caller gives up on the QP -- and no address depends on it.
    bool bad = false;
    for (int j = 0; j < m; ++j) {
        bsync();
        const double* vb = vbuf + (j & 1) * 80 + 2 * kc;
        double2v v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = ld2(vb + 8 * t);
        const double tj = sc[(j & 1) * 2], alpha = sc[(j & 1) * 2 + 1];
        bad = bad || !(fabs(alpha) > 2.220446049250313e-16 * c.R_norm); // also catches a NaN pivot
        c.R_norm = fmax(c.R_norm, fabs(alpha));
        if ((col1 && e > j) || row1) {
            apply(b, v, tj);
            if (e == j + 1 && j + 1 < m) prepare(j + 1);
        }
        if (has2) apply(b2, v, tj);
    }
    if (bad) return false;
    // R packed, 1/R(j,j); J rows back to LDS
    if (col1) {
        double* Rc = c.R + roff(e);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 2 * kc + 8 * t + i;
                if (row < e) Rc[row] = b[t][i];
            }
        if (kc == 0) {
            Rc[e] = my_alpha;
            c.rdinv[e] = 1.0 / my_alpha;
        }
    }
    if (row1) {
        double* Jr = c.J + jr1 * ldj;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = 2 * kc + 8 * t + i;
                if (k < n) Jr[k] = b[t][i];
            }
    }
    if (has2) {
        double* Jr = c.J + jr2 * ldj;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = 2 * kc + 8 * t + i;
                if (k < n) Jr[k] = b2[t][i];
            }
    }
    return true;
}

// y = R'^-1 rhs (forward substitution) on one wave: lane = index.  Column `lane` of the packed R sits in registers
// (clamped loads, all in flight at once); lanes past m carry zeros, so the loop runs to the compile-time bound MM >= m
// without guards.  The multipliers u = R^-1 y of the equality rows are not formed: no later decision reads them (the
// step-length test runs over the inequality rows only) and they are not an output.
template <int MM>
__device__ __forceinline__ void solve_y(Ctx& c, double* rhs)
{
    const int m = c.neq, lane = c.lane;
    const bool live = lane < m;
    const int ls = live ? lane : 0;
    double rc[MM];
#pragma unroll
    for (int i = 0; i < MM; ++i) rc[i] = c.R[roff(ls) + min(i, ls)]; // R(i, lane), used for i < lane
    const double rinv = live ? c.rdinv[ls] : 0.0;
    double yv = live ? rhs[ls] : 0.0;
#pragma unroll
    for (int i = 0; i < MM; ++i) {
        const double yi = bcast_lane(yv * rinv, i);
        if (lane == i) yv = yi;
        if (lane > i) yv = fma(-yi, rc[i], yv);
    }
    if (live) {
        rhs[lane] = yv; // y
        c.u[lane] = 0.0;
        c.A[lane] = -lane - 1;
    }
}

// ------------------------------------------------------------------------------------------------
// Equality phase, blocked.  eiquadprog adds the neq equalities one by one (d = J'n, Givens sweep over J, ...): 18
// full passes over J for Talos.  The same state (J, R, x, u, f) is reached in one go: with N = CE' (n x m) and
// B = J0' N, a Householder QR  Q' B = [R; 0]  gives J = J0 Q (applied as one rank-m update through the compact WY
// form Q = I - V T V'), and the equality-constrained minimiser follows from R' y = -(CE x0 + ce0):
// x = x0 + J[:, :m] y,  u = R^-1 y,  f = f0 + y'y / 2.  J' H J = I and J' N = [R; 0] hold exactly as after m
// add_constraint calls (R's diagonal signs and the null-space basis differ, which the later steps never see).
// Returns false on (numerically) redundant equalities -- upstream's REDUNDANT_EQUALITIES.
// Requires n <= 80, 1 <= m <= 22.
// ------------------------------------------------------------------------------------------------
// build_n(Nm) writes N = CE' (n x m, leading dimension c.ldb) -- every thread calls it, a barrier follows; ce0_of(e) is ce0 of equality e
template <typename BuildN, typename Ce0>
__device__ __forceinline__ bool equality_phase_blocked_t(Ctx& c, double& f_value, BuildN build_n, Ce0 ce0_of)
{
    const int n = c.n, m = c.neq, ldj = c.ldj, ldb = c.ldb, tid = c.tid;
    double* Nm = c.eqw;        // N = CE' (n x m), later W = J0 V
    double* Bm = c.R + 256;    // B -> V (lower trapezoid) / R (strict upper), in the unused tail of the R region
    double* Tm = c.eqt;        // T (m x (m+1))
    double* tau = Tm + m * (m + 1);
    double* rhs = tau + 2 * m;  // later y

    build_n(Nm);
    bsync();
    STAMP(21)
    // ---- rhs_e = -(N(:,e)'x0 + ce0_e): 8 lanes per equality, ten terms each in flight
    {
        const int e = tid >> 3, kc = tid & 7;
        const int es = min(e, m - 1);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < 10; i += 2) {
            const int k0 = min(kc + 8 * i, n - 1), k1 = min(kc + 8 * i + 8, n - 1);
            const double x0 = (kc + 8 * i < n) ? c.x[k0] : 0.0, x1 = (kc + 8 * i + 8 < n) ? c.x[k1] : 0.0;
            a0 = fma(Nm[k0 * ldb + es], x0, a0);
            a1 = fma(Nm[k1 * ldb + es], x1, a1);
        }
        const double acc = grp8_sum(a0 + a1);
        if (e < m && kc == 0) {
            rhs[e] = -(acc + ce0_of(e));
        }
    }
    STAMP(22)
    // ---- B = J0' N: item (pair of columns of J0, 4 equalities).  The k range is the same for the whole wave (J0 is upper
    //      triangular and block diagonal: whatever lies outside a lane's own range is an exact zero), so every J0 read is
    //      a stride-1 row segment and every N read a broadcast.
    {
        const int ncg = (m + 3) >> 2;
        const int cp = tid / ncg, cg = tid - cp * ncg;
        const int c0 = 2 * cp, c1 = min(c0 + 1, n - 1);
        const bool act = c0 < n;
        int kmin = act ? blk_begin(c0, c.nblk) : n, kmax = act ? c1 + 1 : 0;
        kmin = wave_min_int(kmin);
        kmax = wave_max_int(kmax);
        double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        const int c0s = act ? c0 : 0, c1s = act ? c1 : 0;
        tile2x4(c.J, c0s, c1s, ldj, Nm + 4 * cg, ldb, kmin, kmax, acc);
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * cg + q < m) {
                    Bm[c0 * ldb + 4 * cg + q] = acc[0][q];
                    if (c0 + 1 < n) Bm[(c0 + 1) * ldb + 4 * cg + q] = acc[1][q];
                }
        }
    }
    bsync();
    STAMP(5)
    // ---- Householder QR of B (columns in registers, one barrier per column) and J <- J Q in its shadow (rows in registers)
    if (!qr_unified(c, Bm, c.s, c.s + 160)) return false; // redundant equalities
    bsync();
    STAMP(6)
    // ---- y = R'^-1 rhs on one wave
    if (c.wave == 0) {
        if (m <= 12) solve_y<12>(c, rhs);
        else if (m <= 20) solve_y<20>(c, rhs);
        else solve_y<24>(c, rhs);
    }
    bsync();
    STAMP(19)
    // ---- x = x0 + J[:, :m] y ; f += y'y / 2
    {
        double yy = 0.0;
        if (tid < m) yy = rhs[tid] * rhs[tid];
        if (tid >= 128 && tid - 128 < n) {
            const int kk = tid - 128;
            const double* Jr = c.J + kk * ldj;
            double acc = 0.0;
            for (int e = 0; e < m; ++e) acc = fma(Jr[e], rhs[e], acc);
            c.x[kk] += acc;
        }
        yy = block_sum(c, yy);
        f_value += 0.5 * yy;
    }
    c.iq = m;
    bsync();
    return true;
}

// The structured layout's form: N from the record's blocks in LDS (M, Jc = T'A_c, A_c), ce0 = h_u | -bc.
__device__ __forceinline__ bool equality_phase_blocked(Ctx& c, double& f_value)
{
    return equality_phase_blocked_t(
        c, f_value,
        [&](double* Nm) __attribute__((always_inline)) {
            const int n = c.n, m = c.neq, nv = c.nv, nu = c.nu, ldb = c.ldb, tid = c.tid;
    // ---- N = CE': base dynamics rows [M_u | -J_u'], then the contact motion rows [A_c | 0].  Thread = (equality e,
    //      every 8th row): no index division, the ten loads of a thread are in flight together (m <= 22, n <= 80)
    {
        const int e = tid & 31, k8 = tid >> 5;
        if (e < m) {
            // both candidate sources are read unconditionally (clamped addresses) and selected: a load behind a per-lane
            // branch waits for its own round trip
            double v[10];
            if (e < nu) {
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int kk = min(k8 + 8 * i, n - 1);
                    const double mv = c.M[min(kk, nv - 1) * c.ldm + e];
                    const double jv = (n > nv) ? c.Jc[max(kk - nv, 0) * c.ldc + e] : 0.0;
                    v[i] = (kk < nv) ? mv : -jv;
                }
            }
            else {
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int kk = min(k8 + 8 * i, n - 1);
                    const double av = c.Ac[(e - nu) * nv + min(kk, nv - 1)];
                    v[i] = (kk < nv) ? av : 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < 10; ++i)
                if (k8 + 8 * i < n) Nm[(k8 + 8 * i) * ldb + e] = v[i];
        }
    }
        },
        [&](int e) __attribute__((always_inline)) { return (e < c.nu) ? c.h[e] : -c.bc[e - c.nu]; });
}

#endif // __HIPCC__
} // namespace wbcqp
