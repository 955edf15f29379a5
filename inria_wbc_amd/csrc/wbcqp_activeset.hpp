// wbcqp_activeset.hpp -- pieces of the Goldfarb-Idnani active-set iteration (eiquadprog-fast): d = J'n, z and r,
// add_constraint as one Householder reflector, delete_constraint, the row-owned evaluation of s = CI x + ci0.
#pragma once

#include "wbcqp_prims.hpp"

namespace wbcqp {
#ifdef __HIPCC__

// d = J' np over the support [k0, k1) of np (eiquadprog compute_d).
// threads 0..127 own column idx for the first half of the support, threads 128..255 for the second half;
// the two partial sums meet in LDS.  Ends with a barrier: d is visible to every thread on return.
__device__ __forceinline__ void compute_d(Ctx& c, int k0, int k1)
{
    const int n = c.n, ldj = c.ldj;
    const int idx = c.tid & 127, grp = c.tid >> 7;
    const int mid = k0 + ((k1 - k0 + 1) >> 1);
    const int ka = grp ? mid : k0, kb = grp ? k1 : mid;
    if (idx < n) {
        const double* Jc0 = c.J + idx;
        const double* np = c.np;
        double a0 = 0.0, b0 = 0.0;
        int kk = ka;
        for (; kk + 4 <= kb; kk += 4) {
            const double v0 = np[kk], v1 = np[kk + 1], v2 = np[kk + 2], v3 = np[kk + 3];
            a0 = fma(Jc0[kk * ldj], v0, a0);
            b0 = fma(Jc0[(kk + 1) * ldj], v1, b0);
            a0 = fma(Jc0[(kk + 2) * ldj], v2, a0);
            b0 = fma(Jc0[(kk + 3) * ldj], v3, b0);
        }
        for (; kk < kb; ++kk) a0 = fma(Jc0[kk * ldj], np[kk], a0);
        c.part[grp * 128 + idx] = a0 + b0;
    }
    bsync();
    if (c.tid < n) c.d[c.tid] = c.part[c.tid] + c.part[128 + c.tid];
    bsync();
}
// r = R[:iq,:iq]^-1 d[:iq] for the rows rlo..iq-1 on ONE wave (update_r): column-oriented back substitution, the pivot
// travels by readlane, 1/R(j,j) and the column entries of four steps are fetched ahead of the dependent chain.
__device__ __forceinline__ void update_r_wave(Ctx& c, int rlo)
{
    const int lane = c.lane, iq = c.iq;
    if (iq <= rlo) return;
        double v0 = (lane < iq) ? c.d[lane] : 0.0;
        double v1 = (lane + kWave < iq) ? c.d[lane + kWave] : 0.0;
        auto step = [&](int j, double rd, double ra, double rb) {
            const double dj = (j < kWave) ? bcast_lane(v0, j) : bcast_lane(v1, j - kWave);
            const double rj = dj * rd;
            if (lane == (j & (kWave - 1))) c.r[j] = rj;
            if (lane < j) v0 = fma(-rj, ra, v0);
            if (lane + kWave < j) v1 = fma(-rj, rb, v1);
        };
        int j = iq - 1;
        for (; j >= rlo + 3; j -= 4) {
            double rd[4], ra[4], rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j - u;
                const double* Rc = c.R + roff(jj);
                rd[u] = c.rdinv[jj];
                ra[u] = Rc[min(lane, jj)];
                rb[u] = Rc[min(lane + kWave, jj)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) step(j - u, rd[u], ra[u], rb[u]);
        }
        for (; j >= rlo; --j) {
            const double* Rc = c.R + roff(j);
            step(j, c.rdinv[j], Rc[min(lane, j)], Rc[min(lane + kWave, j)]);
        }
}

// z = J[:, iq:] d[iq:] (update_z) on waves 0..2 (each a third of the columns, both row sets), and
// r = R[:iq,:iq]^-1 d[:iq] (update_r) on wave 3: column-oriented back substitution, the pivot travels by
// readlane, 1/R(j,j) and the column entries of four steps are fetched ahead of the dependent chain.  Only r[rlo:iq] is
// formed: the inequality loop passes rlo = neq, because r of the equality rows only feeds the equality multipliers,
// which are neither an output nor an input of any decision.  Ends with barriers: z and r are visible on return.
__device__ __forceinline__ void update_z_r(Ctx& c, int rlo)
{
    const int n = c.n, ldj = c.ldj, lane = c.lane, iq = c.iq;
    if (c.wave < 3) {
        const int span = n - iq;
        const int chunk = (span + 2) / 3;
        const int ca = iq + c.wave * chunk, cb = min(n, ca + chunk);
        const int k0 = lane, k1 = lane + kWave;
        const bool has1 = k1 < n;
        if (k0 < n) {
            const double* J0 = c.J + (size_t)k0 * ldj;
            const double* J1 = c.J + (size_t)(has1 ? k1 : k0) * ldj;
            const double* d = c.d;
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
            int cc = ca;
            for (; cc + 4 <= cb; cc += 4) {
                const double v0 = d[cc], v1 = d[cc + 1], v2 = d[cc + 2], v3 = d[cc + 3];
                a0 = fma(J0[cc], v0, a0);
                a1 = fma(J1[cc], v0, a1);
                b0 = fma(J0[cc + 1], v1, b0);
                b1 = fma(J1[cc + 1], v1, b1);
                a0 = fma(J0[cc + 2], v2, a0);
                a1 = fma(J1[cc + 2], v2, a1);
                b0 = fma(J0[cc + 3], v3, b0);
                b1 = fma(J1[cc + 3], v3, b1);
            }
            for (; cc < cb; ++cc) {
                const double v0 = d[cc];
                a0 = fma(J0[cc], v0, a0);
                a1 = fma(J1[cc], v0, a1);
            }
            c.part[c.wave * 128 + k0] = a0 + b0;
            if (has1) c.part[c.wave * 128 + k1] = a1 + b1;
        }
    }
    else update_r_wave(c, rlo);
    bsync();
    if (c.tid < n) c.z[c.tid] = (c.part[c.tid] + c.part[128 + c.tid]) + c.part[256 + c.tid];
    bsync();
}

// add_constraint, Householder form.  eiquadprog zeroes d[iq+1:] with a chain of n-iq-1 Givens rotations of J's columns (a
// sequential sweep); one reflector H = I - tau v v' (v = d[iq:] - alpha e_0) spans the same subspaces, and its product
// with J needs no new matvec: J[:, iq:] v = z - alpha J[:, iq] with z = J[:, iq:] d[iq:] from update_z.  dn2 = |d[iq:]|^2.
// The new column of R is [d[:iq]; alpha].  Returns false when the constraint is (numerically) dependent.
__device__ __forceinline__ bool add_constraint_hh(Ctx& c, double dn2)
{
    const int n = c.n, ldj = c.ldj, iq = c.iq, tid = c.tid;
    const double diq = c.d[iq];
    double alpha = diq;
    if (iq + 1 < n && dn2 > 0.0) {
        const double inx = rsqrt(dn2);
        const double nx = dn2 * inx;
        alpha = (diq >= 0.0) ? -nx : nx;
        const double v0 = diq - alpha;
        const double tau = inx / (nx + fabs(diq));
        // w_k = tau (z_k - alpha J(k,iq)) for every row, published before anybody touches column iq
        if (tid < n) c.part[tid] = tau * (c.z[tid] - alpha * c.J[tid * ldj + iq]);
        bsync();
        // J(k,c) -= w_k v_c: thread (row k = tid & 127, half of the columns)
        const int k = tid & 127, half = tid >> 7;
        if (k < n) {
            const int span = n - iq;
            const int ca = iq + half * ((span + 1) >> 1), cb = half ? n : iq + ((span + 1) >> 1);
            double* Jk = c.J + k * ldj;
            const double wk = c.part[k];
            int cc = ca;
            if (cc == iq && cc < cb) {
                Jk[cc] = fma(-wk, v0, Jk[cc]);
                ++cc;
            }
            for (; cc + 4 <= cb; cc += 4) {
                const double d0 = c.d[cc], d1 = c.d[cc + 1], d2 = c.d[cc + 2], d3 = c.d[cc + 3];
                const double j0 = Jk[cc], j1 = Jk[cc + 1], j2 = Jk[cc + 2], j3 = Jk[cc + 3];
                Jk[cc] = fma(-wk, d0, j0);
                Jk[cc + 1] = fma(-wk, d1, j1);
                Jk[cc + 2] = fma(-wk, d2, j2);
                Jk[cc + 3] = fma(-wk, d3, j3);
            }
            for (; cc < cb; ++cc) Jk[cc] = fma(-wk, c.d[cc], Jk[cc]);
        }
    }
    double* Rc = c.R + roff(iq);
    for (int i = tid; i < iq; i += kThreads) Rc[i] = c.d[i];
    if (tid == kThreads - 1) {
        Rc[iq] = alpha;
        c.rdinv[iq] = 1.0 / alpha;
    }
    c.iq = iq + 1;
    bsync();
    if (fabs(alpha) <= 2.220446049250313e-16 * c.R_norm) return false; // degenerate
    c.R_norm = fmax(c.R_norm, fabs(alpha));
    return true;
}

// delete_constraint (eiquadprog): drop active constraint l; the Givens chain that restores R's triangle is
// sequential (short: only inequality columns move) and runs on wave 0; the matching J update is a lane-per-row
// sweep on waves 0 and 1.
__device__ __forceinline__ void delete_constraint(Ctx& c, int l)
{
    const int n = c.n, ldj = c.ldj, lane = c.lane, neq = c.neq;
    const int iq_old = c.iq;
    int found = -1;
    for (int i = neq + c.tid; i < iq_old; i += kThreads)
        if (c.A[i] == l) found = i;
    found = block_max_int(c, found);
    const int qq = found < 0 ? 0 : found;

    // remove the constraint from the active set and the duals: positions qq..iq_old-1 take their right
    // neighbour (position iq_old holds the candidate constraint), position iq_old is cleared
    {
        int a0 = 0;
        double u0 = 0.0;
        const int i0 = qq + c.tid;
        if (i0 < iq_old) {
            a0 = c.A[i0 + 1];
            u0 = c.u[i0 + 1];
        }
        bsync();
        if (i0 < iq_old) {
            c.A[i0] = a0;
            c.u[i0] = u0;
        }
        if (c.tid == kThreads - 1) {
            c.A[iq_old] = 0;
            c.u[iq_old] = 0.0;
        }
    }
    // R columns qq+1..iq_old-1 move one place left (each thread moves its own row)
    if (c.tid < 128)
        for (int col = qq; col < iq_old - 1; ++col) {
            const double* src = c.R + roff(col + 1);
            double* dst = c.R + roff(col);
            if (c.tid <= col + 1) dst[c.tid] = src[c.tid];
        }
    const int iq = iq_old - 1;
    c.iq = iq;
    bsync();
    if (iq == 0) return;

    if (c.wave == 0) {
        for (int j = qq; j < iq; ++j) {
            double* Rj = c.R + roff(j);
            double cc = Rj[j], ss = Rj[j + 1];
            double h = gi_distance(cc, ss);
            if (h == 0.0) {
                if (lane == 0) c.gskip[j] = 1;
                continue;
            }
            const double rh = 1.0 / h;
            cc = cc * rh;
            ss = ss * rh;
            double rjj;
            if (cc < 0.0) {
                rjj = -h;
                cc = -cc;
                ss = -ss;
            }
            else
                rjj = h;
            if (lane == 0) {
                Rj[j + 1] = 0.0;
                Rj[j] = rjj;
                c.rdinv[j] = 1.0 / rjj;
                c.prm[2 * j] = cc;
                c.prm[2 * j + 1] = ss;
                c.gskip[j] = 0;
            }
            for (int kc = j + 1 + lane; kc < iq; kc += kWave) {
                double* Rk = c.R + roff(kc);
                const double t1 = Rk[j], t2 = Rk[j + 1];
                Rk[j] = fma(t2, ss, t1 * cc);
                Rk[j + 1] = fma(t1, ss, -(t2 * cc));
            }
            // the next step reads what other lanes of this wave just wrote
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    bsync();
    // J columns qq..iq: ascending sweep, running element in a register
    if (c.wave < 2) {
        const int k = lane + c.wave * kWave;
        if (k < n) {
            double* Jk = c.J + (size_t)k * ldj;
            double t1 = Jk[qq];
            for (int j = qq; j < iq; ++j) {
                const double t2 = Jk[j + 1];
                if (c.gskip[j]) {
                    Jk[j] = t1; // columns j, j+1 untouched by this step
                    t1 = t2;
                    continue;
                }
                const double cc = c.prm[2 * j], ss = c.prm[2 * j + 1];
                Jk[j] = fma(t2, ss, t1 * cc);
                t1 = fma(t1, ss, -(t2 * cc));
            }
            Jk[iq] = t1;
        }
    }
    bsync();
}

// Builds the normal np of equality row i (CE.row(i)) in LDS; returns its support and ce0(i). No barrier.
__device__ __forceinline__ void build_eq_row(Ctx& c, int i, int& k0, int& k1, double& ce0)
{
    const int nv = c.nv, k = c.k, nu = c.nu, tid = c.tid;
    if (i < nu) {
        // base dynamics [M_u | -J_u'] x = -h_u
        if (tid < nv) c.np[tid] = c.M[i * c.ldm + tid];
        else if (tid < nv + k) c.np[tid] = -c.Jc[(tid - nv) * c.ldc + i];
        k0 = 0;
        k1 = c.n;
        ce0 = c.h[i];
    }
    else {
        const int rr = i - nu; // contact*6 + row
        if (tid < nv) c.np[tid] = c.Ac[rr * nv + tid];
        k0 = 0;
        k1 = nv;
        ce0 = -c.bc[rr];
    }
}

// What one thread keeps about the (at most two) rows of s it owns: rows tid and tid + 256
struct OwnRows {
    int meta[2];
    double ci0[2];
    double coef[2][12]; // friction rows only
};
__device__ __forceinline__ void own_rows_init(Ctx& c, OwnRows& o, const double* fmat, const double* flb, const double* fub)
{
#pragma unroll
    for (int z2 = 0; z2 < 2; ++z2) {
        const int i = c.tid + z2 * kThreads;
        o.meta[z2] = -1;
        o.ci0[z2] = 0.0;
#pragma unroll
        for (int m = 0; m < 12; ++m) o.coef[z2][m] = 0.0;
        if (i < c.nin2) {
            const int mt = c.meta[i];
            const int kind = mt & 3, rr = (mt >> 3) & 255, ct = (mt >> 11) & 15;
            const bool neg = (mt >> 2) & 1;
            o.meta[z2] = mt;
            if (kind == INEQ_BOUNDS) o.ci0[z2] = neg ? c.bub[rr] : -c.blb[rr];
            else if (kind == INEQ_ACTUATION) o.ci0[z2] = neg ? c.tu[rr] : -c.tl[rr];
            else {
                o.ci0[z2] = neg ? fub[ct * 17 + rr] : -flb[ct * 17 + rr];
                const double* B = fmat + (ct * 17 + rr) * 12;
#pragma unroll
                for (int m = 0; m < 12; ++m) o.coef[z2][m] = neg ? -B[m] : B[m];
            }
        }
    }
}

// tau' = M_a xn - J_a' fn with xn = x + t z formed on the fly (t = 0: xn = x exactly), four lanes per actuated row,
// partial sums meet by DPP inside the quad.  Every call sums in the same order, so the value for x + t z here is bitwise
// the value a later call on the stored x would give.  No barrier inside; out[rr] is written by the quad's first lane.
__device__ __forceinline__ void act_rows(Ctx& c, double* out, double t)
{
    const int nv = c.nv, nu = c.nu, k = c.k, na = c.na;
    const int rr = c.tid >> 2, q4 = c.tid & 3;
    const int row = nu + min(rr, na - 1);
    const double* Mr = c.M + row * c.ldm;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // the lane's terms j = q4 + 4 i, eight in flight; past the end: index clamped, weight zero
    for (int j = q4; j < nv; j += 32) {
        double mv[8], zv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = min(j + 4 * u, nv - 1);
            mv[u] = (j + 4 * u < nv) ? Mr[jj] : 0.0;
            zv[u] = c.z[jj];
            xv[u] = c.x[jj];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
            a0 = fma(mv[u], fma(t, zv[u], xv[u]), a0);
            a1 = fma(mv[u + 1], fma(t, zv[u + 1], xv[u + 1]), a1);
            a2 = fma(mv[u + 2], fma(t, zv[u + 2], xv[u + 2]), a2);
            a3 = fma(mv[u + 3], fma(t, zv[u + 3], xv[u + 3]), a3);
        }
    }
    for (int m = q4; m < k; m += 32) {
        double jv[8], zv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int mm = min(m + 4 * u, k - 1);
            jv[u] = (m + 4 * u < k) ? c.Jc[mm * c.ldc + row] : 0.0;
            zv[u] = c.z[nv + mm];
            xv[u] = c.x[nv + mm];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
            a0 = fma(-jv[u], fma(t, zv[u], xv[u]), a0);
            a1 = fma(-jv[u + 1], fma(t, zv[u + 1], xv[u + 1]), a1);
            a2 = fma(-jv[u + 2], fma(t, zv[u + 2], xv[u + 2]), a2);
            a3 = fma(-jv[u + 3], fma(t, zv[u + 3], xv[u + 3]), a3);
        }
    }
    double acc = (a0 + a1) + (a2 + a3);
    acc += dpp_get<0xB1>(acc);
    acc += dpp_get<0x4E>(acc);
    if (q4 == 0 && rr < na) out[rr] = acc;
}

// s = CI x + ci0 for the (at most two) rows this thread owns; tact = tau' of act_rows.  Stores s and returns
// sum min(s, 0) and the most violated eligible row (first index on ties).
__device__ __forceinline__ void own_rows_eval(Ctx& c, const OwnRows& o, const double* tact, double& psi, ValIdx& best)
{
    psi = 0.0;
    best = ValIdx{0.0, 0x7fffffff};
#pragma unroll
    for (int z2 = 0; z2 < 2; ++z2) {
        const int mt = o.meta[z2];
        if (mt >= 0) {
            const int kind = mt & 3, rr = (mt >> 3) & 255, ct = (mt >> 11) & 15, col = (mt >> 15) & 255;
            const bool neg = (mt >> 2) & 1;
            double v;
            if (kind == INEQ_BOUNDS) v = neg ? -c.x[col] : c.x[col];
            else if (kind == INEQ_ACTUATION) v = neg ? -tact[rr] : tact[rr];
            else {
                const double* f = c.x + c.nv + 12 * ct;
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < 12; ++m) a = fma(o.coef[z2][m], f[m], a);
                v = a;
            }
            v += o.ci0[z2];
            const int i = c.tid + z2 * kThreads;
            c.s[i] = v;
            c.iaexcl[i] = 1;
            psi += fmin(0.0, v);
            if (v < 0.0 && c.iai[i] != -1) best = vi_min(best, ValIdx{v, i});
        }
    }
}

#endif // __HIPCC__
} // namespace wbcqp
