// wbcqp_device.hpp -- device-side data model and the fused one-wavefront-per-QP kernel (gfx950).
//
// One 64-lane wavefront owns one QP for its whole life: assemble H,g (P2 cost part), Cholesky +
// J = L^-T (P3 preprocessing), Goldfarb-Idnani equality phase and inequality loop (P3), torque
// decode (P4).  Everything that is touched more than once lives in LDS; HBM is read once per QP
// (the compact per-QP record) and written once (x, tau, status, iters).
//
// What each phase stands behind in the reference (/root/reference):
//   assemble / stack : tsid computeProblemData + SolverHQuadProgFast::solve  controller.cpp:244,247
//   GI active set    : eiquadprog-fast solve_quadprog                        controller.cpp:247
//   decode           : getActuatorForces / getAccelerations                  controller.cpp:250-251
// The dense CE / CI matrices of the reference are never formed: rows are regenerated from their
// structure (+-e_col bounds rows, +-[M_a | -J_a'] actuation rows, 17x12 friction blocks).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wbcqp {

constexpr int kWave = 64;
constexpr int kMaxBlocks = 16;
constexpr int kMaxGroups = 8;

enum { INEQ_BOUNDS = 0, INEQ_ACTUATION = 1, INEQ_FORCE = 2 };
enum { HQP_UNKNOWN = -1, HQP_OPTIMAL = 0, HQP_INFEASIBLE = 1, HQP_UNBOUNDED = 2, HQP_MAX_ITER = 3, HQP_ERROR = 4 };

// Constant structure of a task stack, resident in device memory (one per slot).
struct DevStruct {
    int nv, na, nc, k, n, nu;
    int n_dense, n_tasks, n_sel, n_bound, act_bounds;
    int neq, nin2, r1;
    int n_blocks;
    int blk_kind[kMaxBlocks], blk_arg[kMaxBlocks], blk_off[kMaxBlocks], blk_rows[kMaxBlocks];
    int max_iter;
    double hessian_reg;
    const int *dense_row_task, *sel_col, *sel_task, *forcereg_task, *bound_col;
    const double *force_gen; // [nc][6][12]
    const double *ftf;       // [nc][12][12]  F'F,  F = diag(w_f) T
    const double *ft;        // [nc][12][6]   F'
    const double *fric_mat, *fric_lb, *fric_ub;
    // LDS layout: leading dimensions and element offsets (in doubles)
    int ldj, ldm, ldc;
    int o_J, o_R, o_M, o_Jc, o_Ac, o_h, o_x, o_np, o_d, o_z, o_xold, o_r, o_u, o_uold, o_s;
    int o_blb, o_bub, o_tl, o_tu, o_bc, o_prm, o_rdinv, o_dinv, o_g, o_w, o_b1, o_q, o_wrow;
    int o_int; // int area: A[n+2], Aold[n+2], iai[nin2], iaexcl[nin2], gskip[n+2]
    int lds_doubles;
};

template <typename TI>
struct GroupArgs {
    const DevStruct* st;
    const TI *M, *h, *A, *b1, *Ac, *bc, *blb, *bub, *tlb, *tub, *w;
    TI *x, *tau, *objective;
    int *status, *iters, *n_active;
    long long* dbg; // per-QP phase cycle counters, only written by the WBCQP_STAMPS diagnostic build
    int count;
};

template <typename TI>
struct GroupTable {
    int n;
    GroupArgs<TI> g[kMaxGroups];
};

#ifdef __HIPCC__

// ------------------------------------------------------------------------------------------------
// wave64 primitives
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wsync() { __syncthreads(); } // block == one wave: LDS fence, barrier elided

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_movi(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ double bcast_lane(double v, int src)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// DPP controls: quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E, row_half_mirror=0x141, row_mirror=0x140
#define WBCQP_ROW_REDUCE(v, OP)              \
    v = OP(v, dpp_mov<0xB1>(v));             \
    v = OP(v, dpp_mov<0x4E>(v));             \
    v = OP(v, dpp_mov<0x141>(v));            \
    v = OP(v, dpp_mov<0x140>(v));

__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
__device__ __forceinline__ double op_min(double a, double b) { return fmin(a, b); }

// all-lanes sum (every lane returns the bitwise-identical total)
__device__ __forceinline__ double wave_sum(double v)
{
    WBCQP_ROW_REDUCE(v, op_add)
    double r0 = bcast_lane(v, 0), r1 = bcast_lane(v, 16), r2 = bcast_lane(v, 32), r3 = bcast_lane(v, 48);
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ void wave_sum2(double& a, double& b)
{
    WBCQP_ROW_REDUCE(a, op_add)
    WBCQP_ROW_REDUCE(b, op_add)
    a = (bcast_lane(a, 0) + bcast_lane(a, 16)) + (bcast_lane(a, 32) + bcast_lane(a, 48));
    b = (bcast_lane(b, 0) + bcast_lane(b, 16)) + (bcast_lane(b, 32) + bcast_lane(b, 48));
}

// lexicographic (value, index) minimum: smallest value, ties -> smallest index
struct ValIdx {
    double v;
    int i;
};
__device__ __forceinline__ ValIdx vi_min(ValIdx a, ValIdx b)
{
    bool take_b = (b.v < a.v) || (b.v == a.v && b.i < a.i);
    return take_b ? b : a;
}
template <int CTRL>
__device__ __forceinline__ ValIdx vi_dpp(ValIdx a)
{
    ValIdx o;
    o.v = dpp_mov<CTRL>(a.v);
    o.i = dpp_movi<CTRL>(a.i);
    return o;
}
__device__ __forceinline__ ValIdx wave_argmin(ValIdx a)
{
    a = vi_min(a, vi_dpp<0xB1>(a));
    a = vi_min(a, vi_dpp<0x4E>(a));
    a = vi_min(a, vi_dpp<0x141>(a));
    a = vi_min(a, vi_dpp<0x140>(a));
    ValIdx r0{bcast_lane(a.v, 0), __builtin_amdgcn_readlane(a.i, 0)};
    ValIdx r1{bcast_lane(a.v, 16), __builtin_amdgcn_readlane(a.i, 16)};
    ValIdx r2{bcast_lane(a.v, 32), __builtin_amdgcn_readlane(a.i, 32)};
    ValIdx r3{bcast_lane(a.v, 48), __builtin_amdgcn_readlane(a.i, 48)};
    return vi_min(vi_min(r0, r1), vi_min(r2, r3));
}
__device__ __forceinline__ int wave_max_int(int v)
{
    v = max(v, dpp_movi<0xB1>(v));
    v = max(v, dpp_movi<0x4E>(v));
    v = max(v, dpp_movi<0x141>(v));
    v = max(v, dpp_movi<0x140>(v));
    int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// overflow-safe hypot exactly as eiquadprog utils::distance
__device__ __forceinline__ double gi_distance(double a, double b)
{
    double a1 = fabs(a), b1 = fabs(b);
    if (a1 > b1) {
        double t = b1 / a1;
        return a1 * sqrt(1.0 + t * t);
    }
    else if (b1 > a1) {
        double t = a1 / b1;
        return b1 * sqrt(1.0 + t * t);
    }
    return a1 * sqrt(2.0);
}

// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts; identity 0.0)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_shift0(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_prefix_sum(double v)
{
    v += dpp_shift0<0x111, 0xf>(v); // row_shr:1
    v += dpp_shift0<0x112, 0xf>(v); // row_shr:2
    v += dpp_shift0<0x114, 0xf>(v); // row_shr:4
    v += dpp_shift0<0x118, 0xf>(v); // row_shr:8
    v += dpp_shift0<0x142, 0xa>(v); // row_bcast:15 into rows 1,3
    v += dpp_shift0<0x143, 0xc>(v); // row_bcast:31 into rows 2,3
    return v;
}

// loop helpers: body(i) for i in [begin, end), unrolled by U so that the LDS reads of U steps are in flight together
template <int U, typename F>
__device__ __forceinline__ void for_up(int begin, int end, F body)
{
    int i = begin;
    for (; i + U <= end; i += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) body(i + u);
    }
    for (; i < end; ++i) body(i);
}
// body(i) for i = hi, hi-1, ..., lo  (inclusive)
template <int U, typename F>
__device__ __forceinline__ void for_down(int hi, int lo, F body)
{
    int i = hi;
    for (; i - U + 1 >= lo; i -= U) {
#pragma unroll
        for (int u = 0; u < U; ++u) body(i - u);
    }
    for (; i >= lo; --i) body(i);
}

// ------------------------------------------------------------------------------------------------
// per-wave context: LDS pointers + sizes (all wave-uniform)
// ------------------------------------------------------------------------------------------------
struct Ctx {
    const DevStruct* S;
    int lane;
    int nv, na, nc, k, n, nu, neq, nin2, ldj, ldm, ldc;
    double *J, *R, *M, *Jc, *Ac, *h, *x, *np, *d, *z, *xold, *r, *u, *uold, *s;
    double *blb, *bub, *tl, *tu, *bc, *prm, *rdinv, *dinv, *g, *w, *b1, *q, *wrow;
    int *A, *Aold, *iai, *iaexcl, *gskip;
    int iq;
    double R_norm;
};

// packed upper-triangular R with one spare slot per column (column j holds rows 0..j+1):
__device__ __forceinline__ int roff(int j) { return (j * (j + 3)) >> 1; }

// d = J' np over the support [k0, k1) of np  (eiquadprog compute_d); lanes over columns
template <bool TWO>
__device__ __forceinline__ void compute_d(Ctx& c, int k0, int k1)
{
    const int n = c.n, ldj = c.ldj, lane = c.lane;
    const int c0 = lane, c1 = lane + kWave;
    const bool has1 = TWO && (c1 < n);
    if (c0 < n) {
        const double* J0 = c.J + c0;
        const double* J1 = c.J + (has1 ? c1 : c0);
        const double* np = c.np;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        int kk = k0;
        for (; kk + 8 <= k1; kk += 8) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const double v0 = np[kk + u], v1 = np[kk + u + 1];
                a0 = fma(J0[(kk + u) * ldj], v0, a0);
                b0 = fma(J0[(kk + u + 1) * ldj], v1, b0);
                if (TWO) {
                    a1 = fma(J1[(kk + u) * ldj], v0, a1);
                    b1 = fma(J1[(kk + u + 1) * ldj], v1, b1);
                }
            }
        }
        for (; kk < k1; ++kk) {
            const double v0 = np[kk];
            a0 = fma(J0[kk * ldj], v0, a0);
            if (TWO) a1 = fma(J1[kk * ldj], v0, a1);
        }
        c.d[c0] = a0 + b0;
        if (has1) c.d[c1] = a1 + b1;
    }
}
// d = sign * J[row, :]   (np = sign * e_row)
__device__ __forceinline__ void compute_d_unit(Ctx& c, int row, double sign)
{
    for (int cc = c.lane; cc < c.n; cc += kWave) c.d[cc] = sign * c.J[(size_t)row * c.ldj + cc];
}

// z = J[:, iq:] d[iq:]  (update_z); lanes over rows
template <bool TWO>
__device__ __forceinline__ void update_z(Ctx& c)
{
    const int n = c.n, ldj = c.ldj, lane = c.lane, iq = c.iq;
    const int k0 = lane, k1 = lane + kWave;
    const bool has1 = TWO && (k1 < n);
    if (k0 < n) {
        const double* J0 = c.J + (size_t)k0 * ldj;
        const double* J1 = c.J + (size_t)(has1 ? k1 : k0) * ldj;
        const double* d = c.d;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        int cc = iq;
        for (; cc + 8 <= n; cc += 8) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const double v0 = d[cc + u], v1 = d[cc + u + 1];
                a0 = fma(J0[cc + u], v0, a0);
                b0 = fma(J0[cc + u + 1], v1, b0);
                if (TWO) {
                    a1 = fma(J1[cc + u], v0, a1);
                    b1 = fma(J1[cc + u + 1], v1, b1);
                }
            }
        }
        for (; cc < n; ++cc) {
            const double v0 = d[cc];
            a0 = fma(J0[cc], v0, a0);
            if (TWO) a1 = fma(J1[cc], v0, a1);
        }
        c.z[k0] = a0 + b0;
        if (has1) c.z[k1] = a1 + b1;
    }
}

// r = R[:iq,:iq]^-1 d[:iq]  (update_r): column-oriented back substitution, lanes over rows; the pivot
// travels by readlane; 1/R(j,j) and the column entries of four steps are fetched ahead of the chain.
template <bool TWO>
__device__ __forceinline__ void update_r(Ctx& c)
{
    const int iq = c.iq, lane = c.lane;
    if (iq == 0) return;
    double v0 = (lane < iq) ? c.d[lane] : 0.0;
    double v1 = (TWO && lane + kWave < iq) ? c.d[lane + kWave] : 0.0;
    auto step = [&](int j, double rd, double ra, double rb) {
        double dj;
        if (TWO)
            dj = (j < kWave) ? bcast_lane(v0, j) : bcast_lane(v1, j - kWave);
        else
            dj = bcast_lane(v0, j);
        const double rj = dj * rd;
        if (lane == (j & (kWave - 1))) c.r[j] = rj;
        if (lane < j) v0 = fma(-rj, ra, v0);
        if (TWO && lane + kWave < j) v1 = fma(-rj, rb, v1);
    };
    int j = iq - 1;
    for (; j >= 3; j -= 4) {
        double rd[4], ra[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int jj = j - u;
            const double* Rc = c.R + roff(jj);
            rd[u] = c.rdinv[jj];
            ra[u] = Rc[min(lane, jj)];
            rb[u] = TWO ? Rc[min(lane + kWave, jj)] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) step(j - u, rd[u], ra[u], rb[u]);
    }
    for (; j >= 0; --j) {
        const double* Rc = c.R + roff(j);
        step(j, c.rdinv[j], Rc[min(lane, j)], TWO ? Rc[min(lane + kWave, j)] : 0.0);
    }
}

// add_constraint (eiquadprog): Givens sweep that zeroes d[iq+1:], updates J, appends a column to R.
// The rotation parameters come from suffix sums of d^2 (closed form of the upstream hypot chain, one DPP
// scan); the sweep runs lane-per-row with the running element in a register and the loads of 4 steps in flight.
template <bool TWO>
__device__ __forceinline__ bool add_constraint(Ctx& c)
{
    const int n = c.n, ldj = c.ldj, lane = c.lane, iq = c.iq;
    // suffix sums Q_i = sum_{m>=i} d_m^2 : lane l owns element 63-l (and 127-l), so a prefix scan over lanes is a suffix scan over i
    const int e0 = kWave - 1 - lane, e1 = 2 * kWave - 1 - lane;
    double q0, q1 = 0.0, hi_total = 0.0;
    if (TWO) {
        const double v1 = (e1 < n) ? c.d[e1] : 0.0;
        q1 = wave_prefix_sum(v1 * v1);
        hi_total = bcast_lane(q1, kWave - 1);
    }
    {
        const double v0 = (e0 < n) ? c.d[e0] : 0.0;
        q0 = wave_prefix_sum(v0 * v0) + hi_total;
    }
    int last_nz = -1;
    if (e0 < n) {
        c.q[e0] = q0;
        if (q0 > 0.0) last_nz = e0;
    }
    if (TWO && e1 < n) {
        c.q[e1] = q1;
        if (q1 > 0.0) last_nz = e1;
    }
    last_nz = wave_max_int(last_nz);
    wsync();
    const int jstart = min(n - 1, last_nz + 1); // steps j > jstart have h == 0 and are skipped upstream
    const bool any = jstart >= iq + 1;
    if (any) {
        for (int j = iq + 1 + lane; j <= jstart; j += kWave) {
            const int i = j - 1;
            const double qi = c.q[i], qj = c.q[j], di = c.d[i], dj = c.d[j];
            const double rh = 1.0 / sqrt(qi);
            const double ej = (dj < 0.0 ? -1.0 : 1.0) * sqrt(qj);
            double cc = di * rh, ss = ej * rh;
            if (cc < 0.0) {
                cc = -cc;
                ss = -ss;
            }
            double* pr = c.prm + 4 * j;
            pr[0] = cc;
            pr[1] = ss;
            pr[2] = ss / (1.0 + cc);
        }
        wsync();
        const int k0 = lane, k1 = lane + kWave;
        const bool has1 = TWO && (k1 < n);
        if (k0 < n) {
            double* J0 = c.J + (size_t)k0 * ldj;
            double* J1 = c.J + (size_t)(has1 ? k1 : k0) * ldj;
            const double* prm = c.prm;
            double t2a = J0[jstart], t2b = TWO ? J1[jstart] : 0.0;
            int j = jstart;
            for (; j - 3 > iq; j -= 4) {
                double t1a[4], t1b[4], pc[4], ps[4], px[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    t1a[u] = J0[j - u - 1];
                    if (TWO) t1b[u] = J1[j - u - 1];
                    pc[u] = prm[4 * (j - u)];
                    ps[u] = prm[4 * (j - u) + 1];
                    px[u] = prm[4 * (j - u) + 2];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double na_ = fma(t2a, ps[u], t1a[u] * pc[u]);
                    J0[j - u] = fma(px[u], t1a[u] + na_, -t2a);
                    t2a = na_;
                    if (TWO) {
                        const double nb_ = fma(t2b, ps[u], t1b[u] * pc[u]);
                        if (has1) J1[j - u] = fma(px[u], t1b[u] + nb_, -t2b);
                        t2b = nb_;
                    }
                }
            }
            for (; j > iq; --j) {
                const double pc = prm[4 * j], ps = prm[4 * j + 1], px = prm[4 * j + 2];
                const double t1a = J0[j - 1];
                const double na_ = fma(t2a, ps, t1a * pc);
                J0[j] = fma(px, t1a + na_, -t2a);
                t2a = na_;
                if (TWO) {
                    const double t1b = J1[j - 1];
                    const double nb_ = fma(t2b, ps, t1b * pc);
                    if (has1) J1[j] = fma(px, t1b + nb_, -t2b);
                    t2b = nb_;
                }
            }
            J0[iq] = t2a;
            if (has1) J1[iq] = t2b;
        }
    }
    // new column of R = d[0..iq] with d[iq] replaced by the accumulated norm
    double diq;
    if (any)
        diq = (c.d[iq] < 0.0 ? -1.0 : 1.0) * sqrt(c.q[iq]);
    else
        diq = c.d[iq];
    double* Rc = c.R + roff(iq);
    for (int i = lane; i < iq; i += kWave) Rc[i] = c.d[i];
    if (lane == 0) {
        Rc[iq] = diq;
        c.rdinv[iq] = 1.0 / diq;
    }
    c.iq = iq + 1;
    wsync();
    if (fabs(diq) <= 2.220446049250313e-16 * c.R_norm) return false; // degenerate
    c.R_norm = fmax(c.R_norm, fabs(diq));
    return true;
}

// delete_constraint (eiquadprog): drop active constraint l; the Givens chain that restores R's
// triangle is sequential (short: only inequality columns move), the matching J update is a
// lane-per-row sweep like add_constraint's.
template <bool TWO>
__device__ __forceinline__ void delete_constraint(Ctx& c, int l)
{
    const int n = c.n, ldj = c.ldj, lane = c.lane, neq = c.neq;
    const int iq_old = c.iq;
    int found = -1;
    for (int i = neq + lane; i < iq_old; i += kWave)
        if (c.A[i] == l) found = i;
    found = wave_max_int(found);
    const int qq = found < 0 ? 0 : found;

    // remove the constraint from the active set and the duals: positions qq..iq_old-1 take their
    // right neighbour (position iq_old holds the candidate constraint), position iq_old is cleared
    {
        int a0 = 0, a1 = 0;
        double u0 = 0.0, u1 = 0.0;
        const int i0 = qq + lane, i1 = qq + lane + kWave;
        if (i0 < iq_old) { a0 = c.A[i0 + 1]; u0 = c.u[i0 + 1]; }
        if (i1 < iq_old) { a1 = c.A[i1 + 1]; u1 = c.u[i1 + 1]; }
        wsync();
        if (i0 < iq_old) { c.A[i0] = a0; c.u[i0] = u0; }
        if (i1 < iq_old) { c.A[i1] = a1; c.u[i1] = u1; }
        if (lane == 0) { c.A[iq_old] = 0; c.u[iq_old] = 0.0; }
    }
    // R columns qq+1..iq_old-1 move one place left (each lane moves its own row)
    for (int col = qq; col < iq_old - 1; ++col) {
        const double* src = c.R + roff(col + 1);
        double* dst = c.R + roff(col);
        for (int i = lane; i <= col + 1; i += kWave) dst[i] = src[i];
    }
    const int iq = iq_old - 1;
    c.iq = iq;
    wsync();
    if (iq == 0) return;

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
        const double xny = ss / (1.0 + cc);
        if (lane == 0) {
            Rj[j + 1] = 0.0;
            Rj[j] = rjj;
            c.rdinv[j] = (cc < 0.0 ? -rh : rh) * 1.0;
            c.rdinv[j] = 1.0 / rjj;
            double* pr = c.prm + 4 * j;
            pr[0] = cc;
            pr[1] = ss;
            pr[2] = xny;
            c.gskip[j] = 0;
        }
        for (int kc = j + 1 + lane; kc < iq; kc += kWave) {
            double* Rk = c.R + roff(kc);
            double t1 = Rk[j], t2 = Rk[j + 1];
            double nj = fma(t2, ss, t1 * cc);
            Rk[j] = nj;
            Rk[j + 1] = fma(xny, t1 + nj, -t2);
        }
        wsync();
    }
    wsync();
    // J columns qq..iq: ascending sweep, running element in a register
    {
        const int k0 = lane, k1 = lane + kWave;
        const bool has1 = TWO && (k1 < n);
        if (k0 < n) {
            double* J0 = c.J + (size_t)k0 * ldj;
            double* J1 = c.J + (size_t)(has1 ? k1 : k0) * ldj;
            double t1a = J0[qq], t1b = TWO ? J1[qq] : 0.0;
            for (int j = qq; j < iq; ++j) {
                const double t2a = J0[j + 1];
                const double t2b = TWO ? J1[j + 1] : 0.0;
                if (c.gskip[j]) {
                    // columns j, j+1 untouched by this step
                    J0[j] = t1a;
                    if (has1) J1[j] = t1b;
                    t1a = t2a;
                    t1b = t2b;
                    continue;
                }
                const double* pr = c.prm + 4 * j;
                const double cc = pr[0], ss = pr[1], xny = pr[2];
                const double na_ = fma(t2a, ss, t1a * cc);
                J0[j] = na_;
                t1a = fma(xny, na_ + t1a, -t2a);
                if (TWO) {
                    const double nb_ = fma(t2b, ss, t1b * cc);
                    if (has1) J1[j] = nb_;
                    t1b = fma(xny, nb_ + t1b, -t2b);
                }
            }
            J0[iq] = t1a;
            if (has1) J1[iq] = t1b;
        }
    }
    wsync();
}

// Builds the normal np of equality row i (CE.row(i)) in LDS; returns its support and ce0(i).
__device__ __forceinline__ void build_eq_row(Ctx& c, int i, int& k0, int& k1, double& ce0)
{
    const int nv = c.nv, k = c.k, nu = c.nu, lane = c.lane;
    if (i < nu) {
        // base dynamics [M_u | -J_u'] x = -h_u
        for (int j = lane; j < nv; j += kWave) c.np[j] = c.M[i * c.ldm + j];
        for (int m = lane; m < k; m += kWave) c.np[nv + m] = -c.Jc[m * c.ldc + i];
        k0 = 0;
        k1 = c.n;
        ce0 = c.h[i];
    }
    else {
        const int rr = i - nu; // contact*6 + row
        for (int j = lane; j < nv; j += kWave) c.np[j] = c.Ac[rr * nv + j];
        k0 = 0;
        k1 = nv;
        ce0 = -c.bc[rr];
    }
}

// Decodes CI row ip: builds np in LDS, returns support, ci0, and for bound rows the column (unit_col < 0 if not a unit row)
__device__ __forceinline__ void build_ineq_row(Ctx& c, int ip, int& k0, int& k1, double& ci0, int& unit_col, double& unit_sign)
{
    const DevStruct& S = *c.S;
    const int nv = c.nv, k = c.k, nu = c.nu, lane = c.lane;
    int b = 0;
    while (b + 1 < S.n_blocks && ip >= S.blk_off[b] + 2 * S.blk_rows[b]) ++b;
    const int rows = S.blk_rows[b], local = ip - S.blk_off[b];
    const bool neg = local >= rows;
    const int rr = neg ? local - rows : local;
    const double sg = neg ? -1.0 : 1.0;
    const int kind = S.blk_kind[b];
    unit_col = -1;
    unit_sign = sg;
    if (kind == INEQ_BOUNDS) {
        const int col = S.bound_col[rr];
        if (lane == 0) c.np[col] = sg;
        k0 = col;
        k1 = col + 1;
        ci0 = neg ? c.bub[rr] : -c.blb[rr];
        unit_col = col;
    }
    else if (kind == INEQ_ACTUATION) {
        const int row = nu + rr;
        for (int j = lane; j < nv; j += kWave) c.np[j] = sg * c.M[row * c.ldm + j];
        for (int m = lane; m < k; m += kWave) c.np[nv + m] = -sg * c.Jc[m * c.ldc + row];
        k0 = 0;
        k1 = c.n;
        ci0 = neg ? c.tu[rr] : -c.tl[rr];
    }
    else {
        const int ct = S.blk_arg[b];
        const double* B = S.fric_mat + ((size_t)ct * 17 + rr) * 12;
        if (lane < 12) c.np[nv + 12 * ct + lane] = sg * B[lane];
        k0 = nv + 12 * ct;
        k1 = k0 + 12;
        ci0 = neg ? S.fric_ub[ct * 17 + rr] : -S.fric_lb[ct * 17 + rr];
    }
}

// s = CI x + ci0 for every one-sided row, from the structure of each block
__device__ __forceinline__ void compute_s(Ctx& c)
{
    const DevStruct& S = *c.S;
    const int nv = c.nv, k = c.k, nu = c.nu, lane = c.lane;
    for (int b = 0; b < S.n_blocks; ++b) {
        const int kind = S.blk_kind[b], rows = S.blk_rows[b], off = S.blk_off[b];
        if (kind == INEQ_BOUNDS) {
            for (int rr = lane; rr < rows; rr += kWave) {
                double xv = c.x[S.bound_col[rr]];
                c.s[off + rr] = xv - c.blb[rr];
                c.s[off + rows + rr] = -xv + c.bub[rr];
            }
        }
        else if (kind == INEQ_ACTUATION) {
            for (int rr = lane; rr < rows; rr += kWave) {
                const int row = nu + rr;
                const double* Mr = c.M + row * c.ldm;
                const double* Jcr = c.Jc + row;
                const double* x = c.x;
                const int ldc = c.ldc;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                int j = 0;
                for (; j + 8 <= nv; j += 8) {
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        a0 = fma(Mr[j + u], x[j + u], a0);
                        a1 = fma(Mr[j + u + 1], x[j + u + 1], a1);
                    }
                }
                for (; j < nv; ++j) a0 = fma(Mr[j], x[j], a0);
                int m = 0;
                for (; m + 8 <= k; m += 8) {
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        a2 = fma(Jcr[(m + u) * ldc], x[nv + m + u], a2);
                        a3 = fma(Jcr[(m + u + 1) * ldc], x[nv + m + u + 1], a3);
                    }
                }
                for (; m < k; ++m) a2 = fma(Jcr[m * ldc], x[nv + m], a2);
                const double t = (a0 + a1) - (a2 + a3);
                c.s[off + rr] = t - c.tl[rr];
                c.s[off + rows + rr] = -t + c.tu[rr];
            }
        }
        else {
            const int ct = S.blk_arg[b];
            if (lane < 17) {
                const double* B = S.fric_mat + ((size_t)ct * 17 + lane) * 12;
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < 12; ++m) a = fma(B[m], c.x[nv + 12 * ct + m], a);
                c.s[off + lane] = a - S.fric_lb[ct * 17 + lane];
                c.s[off + 17 + lane] = -a + S.fric_ub[ct * 17 + lane];
            }
        }
    }
}

// global -> LDS copy with 8 loads in flight per lane
template <typename TI>
__device__ __forceinline__ void copy_in(const TI* __restrict__ src, double* dst, int len, int lane)
{
    int e = lane;
    for (; e + 7 * kWave < len; e += 8 * kWave) {
        TI v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[e + u * kWave];
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[e + u * kWave] = (double)v[u];
    }
    for (; e < len; e += kWave) dst[e] = (double)src[e];
}

// In-kernel phase stamps (diagnostic build only: -DWBCQP_STAMPS). Never compiled into the product library.
#ifdef WBCQP_STAMPS
constexpr int kStamps = 20;
#define STAMP_DECL long long st_prev_ = clock64(); long long st_acc_[kStamps] = {};
#define STAMP(i) { long long now_ = clock64(); st_acc_[i] += now_ - st_prev_; st_prev_ = now_; }
#else
#define STAMP_DECL
#define STAMP(i)
#endif

// ------------------------------------------------------------------------------------------------
// one QP on one wavefront.  TWO = (n > 64): every lane also owns index lane+64
// ------------------------------------------------------------------------------------------------
template <typename TI, bool TWO>
__device__ __forceinline__ void solve_one(const GroupArgs<TI>& ga, const DevStruct& S, const int b, double* lds)
{
    const int lane = threadIdx.x;
    Ctx c;
    c.S = &S;
    c.lane = lane;
    c.nv = S.nv; c.na = S.na; c.nc = S.nc; c.k = S.k; c.n = S.n; c.nu = S.nu;
    c.neq = S.neq; c.nin2 = S.nin2; c.ldj = S.ldj; c.ldm = S.ldm; c.ldc = S.ldc;
    c.J = lds + S.o_J; c.R = lds + S.o_R; c.M = lds + S.o_M; c.Jc = lds + S.o_Jc; c.Ac = lds + S.o_Ac;
    c.h = lds + S.o_h; c.x = lds + S.o_x; c.np = lds + S.o_np; c.d = lds + S.o_d; c.z = lds + S.o_z;
    c.xold = lds + S.o_xold; c.r = lds + S.o_r; c.u = lds + S.o_u; c.uold = lds + S.o_uold; c.s = lds + S.o_s;
    c.blb = lds + S.o_blb; c.bub = lds + S.o_bub; c.tl = lds + S.o_tl; c.tu = lds + S.o_tu; c.bc = lds + S.o_bc;
    c.prm = lds + S.o_prm; c.rdinv = lds + S.o_rdinv;
    c.dinv = lds + S.o_dinv; c.g = lds + S.o_g; c.w = lds + S.o_w; c.b1 = lds + S.o_b1; c.q = lds + S.o_q;
    c.wrow = lds + S.o_wrow;
    int* ia = reinterpret_cast<int*>(lds + S.o_int);
    const int n = c.n, nv = c.nv, na = c.na, nc = c.nc, k = c.k, nu = c.nu, neq = c.neq, nin2 = c.nin2;
    const int ldj = c.ldj, ldm = c.ldm, ldc = c.ldc;
    c.A = ia; c.Aold = ia + (n + 2); c.iai = ia + 2 * (n + 2); c.iaexcl = c.iai + nin2; c.gskip = c.iaexcl + nin2;
    c.iq = 0;
    c.R_norm = 1.0;

    const int n_dense = S.n_dense, n_sel = S.n_sel, n_bound = S.n_bound, r1 = S.r1, n_tasks = S.n_tasks;
    const size_t qp = (size_t)b;
    double* As = c.R; // dense task rows are staged in the (not yet used) R region
    double* Mst = c.J; // packed M is staged in the J region before H is assembled there

    STAMP_DECL
    // ---------------- phase 0: one pass over the QP's HBM record ----------------
    const int lenM = nv * (nv + 1) / 2;
    copy_in(ga.M + qp * lenM, Mst, lenM, lane);
    copy_in(ga.A + qp * (size_t)(n_dense * nv), As, n_dense * nv, lane);
    copy_in(ga.h + qp * nv, c.h, nv, lane);
    copy_in(ga.b1 + qp * r1, c.b1, r1, lane);
    copy_in(ga.w + qp * n_tasks, c.w, n_tasks, lane);
    if (nc > 0) {
        copy_in(ga.Ac + qp * (size_t)(nc * 6 * nv), c.Ac, nc * 6 * nv, lane);
        copy_in(ga.bc + qp * (nc * 6), c.bc, nc * 6, lane);
    }
    if (n_bound > 0) {
        copy_in(ga.blb + qp * n_bound, c.blb, n_bound, lane);
        copy_in(ga.bub + qp * n_bound, c.bub, n_bound, lane);
    }
    if (S.act_bounds) {
        copy_in(ga.tlb + qp * na, c.tl, na, lane);
        copy_in(ga.tub + qp * na, c.tu, na, lane);
    }
    wsync();
    // expand packed M into the full symmetric matrix; lb - h_a, ub - h_a (computeProblemData, actuation tasks)
    for (int i = 0; i < nv; ++i) {
        const int base = i * (i + 1) / 2;
        for (int j = lane; j <= i; j += kWave) {
            const double v = Mst[base + j];
            c.M[i * ldm + j] = v;
            c.M[j * ldm + i] = v;
        }
    }
    if (S.act_bounds) {
        for (int e = lane; e < na; e += kWave) {
            const double ha = c.h[nu + e];
            c.tl[e] -= ha;
            c.tu[e] -= ha;
        }
    }
    for (int r = lane; r < n_dense; r += kWave) c.wrow[r] = c.w[S.dense_row_task[r]];
    // Jc = T' A_c  (12 x nv per contact)
    for (int ct = 0; ct < nc; ++ct) {
        const double* T = S.force_gen + ct * 72;
        const double* Acc = c.Ac + ct * 6 * nv;
        for (int j = lane; j < nv; j += kWave) {
            double a[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) a[r] = Acc[r * nv + j];
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                double sacc = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) sacc = fma(T[r * 12 + m], a[r], sacc);
                c.Jc[(12 * ct + m) * ldc + j] = sacc;
            }
        }
    }
    wsync();
    for (int e = lane; e < n * ldj; e += kWave) c.J[e] = 0.0;
    wsync();
    STAMP(0)

    // ---------------- phase 1: H = sum_r w_r a_r a_r' (+ selection, force-reg, reg), g ----------------
    // lanes over columns j, eight rows of H per pass; the lane's own operand carries the row weight
    for (int j = lane; j < nv; j += kWave) {
        const double* Aj = As + j;
        for (int ib = 0; ib < nv; ib += 8) {
            double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            const double* Ai = As + ib;
            for_up<2>(0, n_dense, [&](int r) {
                const double ajw = Aj[r * nv] * c.wrow[r];
#pragma unroll
                for (int qd = 0; qd < 8; ++qd) acc[qd] = fma(Ai[r * nv + qd], ajw, acc[qd]);
            });
#pragma unroll
            for (int qd = 0; qd < 8; ++qd)
                if (ib + qd < nv) c.J[(ib + qd) * ldj + j] = acc[qd];
        }
        double g0 = 0.0, g1 = 0.0;
        int r = 0;
        for (; r + 2 <= n_dense; r += 2) {
            g0 = fma(Aj[r * nv] * c.wrow[r], c.b1[r], g0);
            g1 = fma(Aj[(r + 1) * nv] * c.wrow[r + 1], c.b1[r + 1], g1);
        }
        for (; r < n_dense; ++r) g0 = fma(Aj[r * nv] * c.wrow[r], c.b1[r], g0);
        c.g[j] = -(g0 + g1);
    }
    for (int m = lane; m < k; m += kWave) c.g[nv + m] = 0.0;
    wsync();
    // selection rows (posture): H(c,c) += w, g(c) -= w b
    for (int sidx = lane; sidx < n_sel; sidx += kWave) {
        const int col = S.sel_col[sidx];
        const double wt = c.w[S.sel_task[sidx]];
        c.J[col * ldj + col] += wt;
        c.g[col] -= wt * c.b1[n_dense + sidx];
    }
    // force regularisation blocks: H_ff += w F'F, g_f -= w F' b
    for (int ct = 0; ct < nc; ++ct) {
        const double wt = c.w[S.forcereg_task[ct]];
        const double* FtF = S.ftf + ct * 144;
        const double* Ft = S.ft + ct * 72;
        const double* bb = c.b1 + n_dense + n_sel + 6 * ct;
        for (int e = lane; e < 144; e += kWave) {
            const int a = e / 12, bcol = e % 12;
            c.J[(nv + 12 * ct + a) * ldj + nv + 12 * ct + bcol] = wt * FtF[e];
        }
        if (lane < 12) {
            double sacc = 0.0;
#pragma unroll
            for (int qd = 0; qd < 6; ++qd) sacc = fma(Ft[lane * 6 + qd], bb[qd], sacc);
            c.g[nv + 12 * ct + lane] = -wt * sacc;
        }
    }
    wsync();
    double c1;
    {
        double tr = 0.0;
        for (int i = lane; i < n; i += kWave) {
            double v = c.J[i * ldj + i] + S.hessian_reg;
            c.J[i * ldj + i] = v;
            tr += v;
        }
        c1 = wave_sum(tr);
    }
    wsync();
    STAMP(1)

    // ---------------- phase 2: Cholesky H = L L' in place (lower), skyline = block structure ----------------
    // first structurally non-zero column of row i: 0 for dv rows, start of the contact block for force rows
    for (int j = 0; j < n; ++j) {
        const int fj = (j < nv) ? 0 : nv + 12 * ((j - nv) / 12);
        const int i0 = lane, i1 = lane + kWave;
        const bool act0 = (i0 >= j) && (i0 < n), act1 = TWO && (i1 >= j) && (i1 < n);
        const double* Lj = c.J + j * ldj;
        const double* L0 = c.J + (act0 ? i0 : j) * ldj;
        const double* L1 = c.J + (act1 ? i1 : j) * ldj;
        double a0 = L0[j], a1 = TWO ? L1[j] : 0.0, e0 = 0.0, e1 = 0.0;
        int p = fj;
        for (; p + 8 <= j; p += 8) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const double l0 = Lj[p + u], l1 = Lj[p + u + 1];
                a0 = fma(-L0[p + u], l0, a0);
                e0 = fma(-L0[p + u + 1], l1, e0);
                if (TWO) {
                    a1 = fma(-L1[p + u], l0, a1);
                    e1 = fma(-L1[p + u + 1], l1, e1);
                }
            }
        }
        for (; p < j; ++p) {
            const double l0 = Lj[p];
            a0 = fma(-L0[p], l0, a0);
            if (TWO) a1 = fma(-L1[p], l0, a1);
        }
        a0 += e0;
        a1 += e1;
        double piv;
        if (TWO)
            piv = (j < kWave) ? bcast_lane(a0, j) : bcast_lane(a1, j - kWave);
        else
            piv = bcast_lane(a0, j);
        // L(j,j) itself is never read again: only 1/L(j,j) is (column scaling here, diagonal of J = L^-T later)
        const double inv = rsqrt(piv);
        if (act0) c.J[i0 * ldj + j] = (i0 == j) ? inv : a0 * inv;
        if (act1) c.J[i1 * ldj + j] = (i1 == j) ? inv : a1 * inv;
        if (lane == 0) c.dinv[j] = inv;
        wsync();
    }
    STAMP(2)

    // ---------------- phase 2b: J = L^-T. Row i of X = L^-1 is built from rows < i; X' is written into the
    // upper triangle of the same buffer (X(i,c) -> J[c][i]). Once row i of L has been consumed it is dead, so its
    // lower part is zeroed and its diagonal already holds 1/L(i,i): later steps then read, for every lane c,
    // J[c][p] = 0 (p < c), X(c,c) (p = c), X(p,c) (p > c) without any masking. ----------------
    for (int i = 1; i < n; ++i) {
        const int fi = (i < nv) ? 0 : nv + 12 * ((i - nv) / 12);
        double* Li = c.J + i * ldj;
        const double di = c.dinv[i];
        const int c0 = lane, c1 = lane + kWave;
        const bool act0 = (c0 >= fi) && (c0 < i), act1 = TWO && (c1 >= fi) && (c1 < i);
        const double* X0 = c.J + (act0 ? c0 : 0) * ldj; // X(p, c0) at J[c0][p]
        const double* X1 = c.J + (act1 ? c1 : 0) * ldj;
        double a0 = 0.0, a1 = 0.0, e0 = 0.0, e1 = 0.0;
        int p = fi;
        for (; p + 8 <= i; p += 8) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const double l0 = Li[p + u], l1 = Li[p + u + 1];
                a0 = fma(l0, X0[p + u], a0);
                e0 = fma(l1, X0[p + u + 1], e0);
                if (TWO) {
                    a1 = fma(l0, X1[p + u], a1);
                    e1 = fma(l1, X1[p + u + 1], e1);
                }
            }
        }
        for (; p < i; ++p) {
            const double l0 = Li[p];
            a0 = fma(l0, X0[p], a0);
            if (TWO) a1 = fma(l0, X1[p], a1);
        }
        if (act0) c.J[c0 * ldj + i] = -(a0 + e0) * di;
        if (act1) c.J[c1 * ldj + i] = -(a1 + e1) * di;
        for (int pz = lane; pz < i; pz += kWave) Li[pz] = 0.0;
        wsync();
    }
    double c2;
    {
        double tr = 0.0;
        for (int i = lane; i < n; i += kWave) tr += c.dinv[i];
        c2 = wave_sum(tr);
    }
    wsync();
    STAMP(3)

    // ---------------- x = -H^-1 g = -J (J' g); f = 0.5 g'x ----------------
    for (int i = lane; i < n; i += kWave) c.np[i] = c.g[i];
    wsync();
    compute_d<TWO>(c, 0, n); // d = J' g
    wsync();
    c.iq = 0;
    update_z<TWO>(c); // z = J d
    wsync();
    double f_value;
    {
        double part = 0.0;
        for (int i = lane; i < n; i += kWave) {
            const double xv = -c.z[i];
            c.x[i] = xv;
            part = fma(0.5 * c.g[i], xv, part);
        }
        f_value = wave_sum(part);
    }
    for (int i = lane; i < n + 2; i += kWave) {
        c.u[i] = 0.0;
        c.A[i] = 0;
    }
    wsync();
    STAMP(4)

    const double eps = 2.220446049250313e-16;
    const double inf = __builtin_huge_val();
    int status = -2; // running
    int iter = 0;

    // ---------------- phase 3: equality constraints ----------------
    for (int i = 0; i < neq && status == -2; ++i) {
        int k0, k1;
        double ce0;
        build_eq_row(c, i, k0, k1, ce0);
        wsync();
        compute_d<TWO>(c, k0, k1);
        wsync();
        STAMP(5)
        update_z<TWO>(c);
        update_r<TWO>(c);
        wsync();
        STAMP(6)
        double zz = 0.0, znp = 0.0, npx = 0.0;
        for (int j = lane; j < n; j += kWave) {
            const double zv = c.z[j];
            zz = fma(zv, zv, zz);
            if (j >= k0 && j < k1) {
                const double nv_ = c.np[j];
                znp = fma(zv, nv_, znp);
                npx = fma(nv_, c.x[j], npx);
            }
        }
        wave_sum2(zz, znp);
        npx = wave_sum(npx);
        double t2 = 0.0;
        if (fabs(zz) > eps) t2 = (-npx - ce0) / znp;
        for (int j = lane; j < n; j += kWave) c.x[j] = fma(t2, c.z[j], c.x[j]);
        const int iq = c.iq;
        for (int j = lane; j < iq; j += kWave) c.u[j] = fma(-t2, c.r[j], c.u[j]);
        if (lane == 0) {
            c.u[iq] = t2;
            c.A[i] = -i - 1;
        }
        f_value += 0.5 * (t2 * t2) * znp;
        wsync();
        STAMP(7)
        if (!add_constraint<TWO>(c)) status = HQP_ERROR; // redundant equalities
        STAMP(8)
    }

    // ---------------- phase 4: inequality loop (GI steps 1, 2, 2a-2c) ----------------
    if (status == -2) {
        for (int i = lane; i < nin2; i += kWave) c.iai[i] = i;
        wsync();
        const double psi_tol = (double)nin2 * eps * c1 * c2 * 100.0;
        while (status == -2) {
            // l1
            ++iter;
            if (iter >= S.max_iter) {
                status = HQP_MAX_ITER;
                break;
            }
            for (int i = neq + lane; i < c.iq; i += kWave) c.iai[c.A[i]] = -1;
            compute_s(c);
            wsync();
            double psi = 0.0;
            for (int i = lane; i < nin2; i += kWave) {
                c.iaexcl[i] = 1;
                psi += fmin(0.0, c.s[i]);
            }
            psi = wave_sum(psi);
            if (fabs(psi) <= psi_tol) {
                status = HQP_OPTIMAL;
                break;
            }
            for (int i = lane; i < c.iq; i += kWave) {
                c.uold[i] = c.u[i];
                c.Aold[i] = c.A[i];
            }
            for (int i = lane; i < n; i += kWave) c.xold[i] = c.x[i];
            wsync();
            STAMP(9)

            bool again_l2 = true;
            while (again_l2 && status == -2) {
                again_l2 = false;
                // l2: most violated non-active, non-excluded constraint (first index on ties)
                ValIdx best{0.0, 0x7fffffff};
                for (int i = lane; i < nin2; i += kWave) {
                    const double sv = c.s[i];
                    if (sv < 0.0 && c.iai[i] != -1 && c.iaexcl[i]) best = vi_min(best, ValIdx{sv, i});
                }
                best = wave_argmin(best);
                if (best.v >= 0.0) {
                    status = HQP_OPTIMAL;
                    break;
                }
                const int ip = uni(best.i);
                int k0, k1, ucol;
                double ci0, usign;
                build_ineq_row(c, ip, k0, k1, ci0, ucol, usign);
                if (lane == 0) {
                    c.u[c.iq] = 0.0;
                    c.A[c.iq] = ip;
                }
                wsync();
                STAMP(10)

                // l2a
                while (true) {
                    if (ucol >= 0)
                        compute_d_unit(c, ucol, usign);
                    else
                        compute_d<TWO>(c, k0, k1);
                    wsync();
                    STAMP(11)
                    update_z<TWO>(c);
                    update_r<TWO>(c);
                    wsync();
                    STAMP(12)
                    const int iq = c.iq;
                    // step 2b: partial step length t1 (dual feasibility) and full step length t2
                    ValIdx bt{inf, 0x7fffffff};
                    for (int kk = neq + lane; kk < iq; kk += kWave) {
                        const double rk = c.r[kk];
                        if (rk > 0.0) bt = vi_min(bt, ValIdx{c.u[kk] / rk, kk});
                    }
                    bt = wave_argmin(bt);
                    const double t1 = bt.v;
                    const int lpos = uni(bt.i);
                    const int l = (t1 < inf) ? c.A[lpos] : 0;
                    double zz = 0.0, znp = 0.0;
                    for (int j = lane; j < n; j += kWave) {
                        const double zv = c.z[j];
                        zz = fma(zv, zv, zz);
                        if (j >= k0 && j < k1) znp = fma(zv, c.np[j], znp);
                    }
                    wave_sum2(zz, znp);
                    const double sip = c.s[ip];
                    const double t2 = (fabs(zz) > eps) ? (-sip / znp) : inf;
                    const double t = fmin(t1, t2);
                    if (t >= inf) {
                        status = HQP_INFEASIBLE; // eiquadprog UNBOUNDED (dual) -> tsid INFEASIBLE
                        break;
                    }
                    if (t2 >= inf) {
                        // (ii) dual step only, drop l
                        for (int j = lane; j < iq; j += kWave) c.u[j] = fma(-t, c.r[j], c.u[j]);
                        if (lane == 0) {
                            c.u[iq] += t;
                            c.iai[l] = l;
                        }
                        wsync();
                        STAMP(13)
                        delete_constraint<TWO>(c, l);
                        STAMP(15)
                        continue;
                    }
                    // (iii) primal + dual step
                    for (int j = lane; j < n; j += kWave) c.x[j] = fma(t, c.z[j], c.x[j]);
                    const double uiq = c.u[iq];
                    f_value += t * znp * (0.5 * t + uiq);
                    for (int j = lane; j < iq; j += kWave) c.u[j] = fma(-t, c.r[j], c.u[j]);
                    wsync();
                    if (lane == 0) c.u[iq] = uiq + t;
                    wsync();
                    STAMP(13)
                    if (t == t2) {
                        // full step: add ip to the active set
                        const bool added_ = add_constraint<TWO>(c);
                        STAMP(14)
                        if (!added_) {
                            if (lane == 0) c.iaexcl[ip] = 0;
                            wsync();
                            delete_constraint<TWO>(c, ip);
                            for (int i = lane; i < nin2; i += kWave) c.iai[i] = i;
                            wsync();
                            for (int i = lane; i < c.iq; i += kWave) {
                                const int av = c.Aold[i];
                                c.A[i] = av;
                                if (av >= 0) c.iai[av] = -1;
                                c.u[i] = c.uold[i];
                            }
                            for (int i = lane; i < n; i += kWave) c.x[i] = c.xold[i];
                            wsync();
                            again_l2 = true;
                        }
                        else {
                            if (lane == 0) c.iai[ip] = -1;
                        }
                        break; // -> l1 (or l2 again)
                    }
                    // partial step: drop l, refresh s(ip)
                    if (lane == 0) c.iai[l] = l;
                    wsync();
                    delete_constraint<TWO>(c, l);
                    STAMP(15)
                    double part = 0.0;
                    for (int j = k0 + lane; j < k1; j += kWave) part = fma(c.np[j], c.x[j], part);
                    part = wave_sum(part);
                    if (lane == 0) c.s[ip] = part + ci0;
                    wsync();
                }
            }
        }
    }

    STAMP(16)
    // ---------------- phase 5: decode + write-out ----------------
    // tau = h_a + M_a dv - J_a' f   (getActuatorForces)
    TI* xo = ga.x + qp * n;
    for (int i = lane; i < n; i += kWave) xo[i] = (TI)c.x[i];
    if (na > 0) {
        TI* to = ga.tau + qp * na;
        for (int rr = lane; rr < na; rr += kWave) {
            const int row = nu + rr;
            const double* Mr = c.M + row * ldm;
            double a0 = c.h[row], a1 = 0.0;
            for_up<8>(0, nv, [&](int j) { a0 = fma(Mr[j], c.x[j], a0); });
            for_up<8>(0, k, [&](int m) { a1 = fma(c.Jc[m * ldc + row], c.x[nv + m], a1); });
            to[rr] = (TI)(a0 - a1);
        }
    }
    if (lane == 0) {
        ga.status[qp] = status;
        ga.iters[qp] = iter;
        if (ga.objective) ga.objective[qp] = (TI)f_value;
        if (ga.n_active) ga.n_active[qp] = c.iq;
    }
#ifdef WBCQP_STAMPS
    STAMP(17)
    if (lane == 0 && ga.dbg)
        for (int i = 0; i < kStamps; ++i) ga.dbg[qp * kStamps + i] = st_acc_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// the kernel: grid = total QPs, block = 64 threads = one wavefront = one QP
// ------------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(kWave) void solve_kernel(const GroupTable<TI> tab)
{
    extern __shared__ __align__(16) double lds[];
    int b = blockIdx.x, gi = 0;
    while (gi + 1 < tab.n && b >= tab.g[gi].count) {
        b -= tab.g[gi].count;
        ++gi;
    }
    const GroupArgs<TI>& ga = tab.g[gi];
    const DevStruct& S = *ga.st;
    if (S.n > kWave)
        solve_one<TI, true>(ga, S, b, lds);
    else
        solve_one<TI, false>(ga, S, b, lds);
}

#endif // __HIPCC__
} // namespace wbcqp
