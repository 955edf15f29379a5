// wbcqp_integrate.hpp -- state integration after the path.
#pragma once

#include "wbcqp_prims.hpp"

namespace wbcqp {
#ifdef __HIPCC__

// ------------------------------------------------------------------------------------------------
// After the path: Controller::_solve's use of an optimal solution, controller.cpp:250-272:
// v = dq + dt dv, q = pinocchio::integrate(model, q, dt v) for a free-flyer root + revolute joints (or revolute joints
// only), base orientation repacked from quaternion to angle * axis.  One wavefront per instance: lane j integrates
// joint j, lane 0 the SE(3) part (exp6, M0 * exp6, rotation -> quaternion, sign continuity, first-order normalisation:
// pinocchio's free-flyer integrate; Eigen's AngleAxis(quaternion)).  HBM-bound: (3 nq + 3 nv) words per instance.
// ------------------------------------------------------------------------------------------------
// one instance on one wavefront: (qi, dqi) the state, dvi the accelerations of the solution, ok = the QP was solved
template <typename TI>
__device__ __forceinline__ void integrate_one(const int nv, const int floating_base, const double dt, const TI* qi, const TI* dqi, const TI* dvi,
                                              const bool ok, TI* qo, TI* vo, TI* so, const int lane)
{
    const int nq = floating_base ? nv + 1 : nv;
    if (!ok) { // the reference throws here; the state stays where it was
        for (int j = lane; j < nq; j += 64) qo[j] = qi[j];
        for (int j = lane; j < nv; j += 64) vo[j] = dqi[j];
        if (so) {
            if (!floating_base) {
                for (int j = lane; j < nv; j += 64) so[j] = qi[j];
            }
            else if (lane == 0) {
                const double u0 = (double)qi[3], u1 = (double)qi[4], u2 = (double)qi[5], u3 = (double)qi[6];
                double nn = sqrt(u0 * u0 + u1 * u1 + u2 * u2), angle = 0.0, a0 = 1.0, a1 = 0.0, a2 = 0.0;
                if (nn != 0.0) {
                    angle = 2.0 * atan2(nn, fabs(u3));
                    if (u3 < 0.0) nn = -nn;
                    a0 = u0 / nn; a1 = u1 / nn; a2 = u2 / nn;
                }
                so[0] = qi[0]; so[1] = qi[1]; so[2] = qi[2];
                so[3] = (TI)(angle * a0); so[4] = (TI)(angle * a1); so[5] = (TI)(angle * a2);
            }
            if (floating_base)
                for (int j = 6 + lane; j < nv; j += 64) so[j] = qi[j + 1];
        }
        return;
    }
    const int j0 = floating_base ? 6 : 0;
    for (int j = lane; j < nv; j += 64) {
        const double vj = __dadd_rn((double)dqi[j], __dmul_rn(dt, (double)dvi[j])); // unfused, as the reference's Eigen expression
        vo[j] = (TI)vj;
        if (j >= j0) {
            const double qj = __dadd_rn((double)qi[j + (floating_base ? 1 : 0)], __dmul_rn(dt, vj));
            qo[j + (floating_base ? 1 : 0)] = (TI)qj;
            if (so) so[j] = (TI)qj;
        }
    }
    if (floating_base && lane == 0) {
        double vv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) vv[i] = __dmul_rn(dt, __dadd_rn((double)dqi[i], __dmul_rn(dt, (double)dvi[i])));
        const double* v = vv;
        const double* w = vv + 3;
        const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
        const double t = sqrt(t2);
        const double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
        double ct, alpha_v, alpha_wxv, alpha_w;
        if (t > 1e-4) {
            ct = cos(t);
            const double st = sin(t), inv_t2 = 1.0 / t2;
            alpha_wxv = (1.0 - ct) * inv_t2;
            alpha_v = st / t;
            alpha_w = (1.0 - alpha_v) * inv_t2 * wv;
        }
        else {
            alpha_wxv = 0.5 - t2 / 24.0;
            alpha_v = 1.0 - t2 / 6.0;
            alpha_w = (1.0 / 6.0 - t2 / 120.0) * wv;
            ct = 1.0 - t2 / 2.0;
        }
        const double cx = w[1] * v[2] - w[2] * v[1], cy = w[2] * v[0] - w[0] * v[2], cz = w[0] * v[1] - w[1] * v[0];
        const double tr0 = alpha_v * v[0] + alpha_w * w[0] + alpha_wxv * cx;
        const double tr1 = alpha_v * v[1] + alpha_w * w[1] + alpha_wxv * cy;
        const double tr2 = alpha_v * v[2] + alpha_w * w[2] + alpha_wxv * cz;
        double E[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) E[3 * i + jj] = alpha_wxv * w[i] * w[jj];
        E[0] += ct; E[4] += ct; E[8] += ct;
        E[1] -= alpha_v * w[2]; E[3] += alpha_v * w[2];
        E[2] += alpha_v * w[1]; E[6] -= alpha_v * w[1];
        E[5] -= alpha_v * w[0]; E[7] += alpha_v * w[0];
        const double qx = (double)qi[3], qy = (double)qi[4], qz = (double)qi[5], qw = (double)qi[6];
        double R0[9];
        {
            const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
            const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
            const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
            const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
            R0[0] = 1.0 - (tyy + tzz); R0[1] = txy - twz;         R0[2] = txz + twy;
            R0[3] = txy + twz;         R0[4] = 1.0 - (txx + tzz); R0[5] = tyz - twx;
            R0[6] = txz - twy;         R0[7] = tyz + twx;         R0[8] = 1.0 - (txx + tyy);
        }
        double pn[3], R1[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            pn[i] = (double)qi[i] + R0[3 * i] * tr0 + R0[3 * i + 1] * tr1 + R0[3 * i + 2] * tr2;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) R1[3 * i + jj] = R0[3 * i] * E[jj] + R0[3 * i + 1] * E[3 + jj] + R0[3 * i + 2] * E[6 + jj];
        }
        double qt[4];
        double tt = R1[0] + R1[4] + R1[8];
        if (tt > 0.0) {
            tt = sqrt(tt + 1.0);
            qt[3] = 0.5 * tt;
            tt = 0.5 / tt;
            qt[0] = (R1[7] - R1[5]) * tt;
            qt[1] = (R1[2] - R1[6]) * tt;
            qt[2] = (R1[3] - R1[1]) * tt;
        }
        else {
            // Eigen's branch for a rotation by more than 120 degrees: the largest diagonal element picks the axis.  Three explicit cases with
            // compile-time indices: an index computed at run time (R1[4 * i], qv[i]) sent both arrays to scratch memory (80 B per lane)
            auto from_axis = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value, jx = (i + 1) % 3, kx = (jx + 1) % 3;
                double t3 = sqrt(R1[4 * i] - R1[4 * jx] - R1[4 * kx] + 1.0);
                qt[i] = 0.5 * t3;
                t3 = 0.5 / t3;
                qt[3] = (R1[3 * kx + jx] - R1[3 * jx + kx]) * t3;
                qt[jx] = (R1[3 * jx + i] + R1[3 * i + jx]) * t3;
                qt[kx] = (R1[3 * kx + i] + R1[3 * i + kx]) * t3;
            };
            const bool one = R1[4] > R1[0];
            const bool two = R1[8] > (one ? R1[4] : R1[0]);
            if (two) from_axis(std::integral_constant<int, 2>{});
            else if (one) from_axis(std::integral_constant<int, 1>{});
            else from_axis(std::integral_constant<int, 0>{});
        }
        const double dotp = qt[0] * qx + qt[1] * qy + qt[2] * qz + qt[3] * qw;
        const double sgn = (dotp < 0.0) ? -1.0 : 1.0;
        const double N2 = qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2] + qt[3] * qt[3];
        const double al = sgn * (3.0 - N2) / 2.0;
        const double u0 = qt[0] * al, u1 = qt[1] * al, u2 = qt[2] * al, u3 = qt[3] * al;
        qo[0] = (TI)pn[0]; qo[1] = (TI)pn[1]; qo[2] = (TI)pn[2];
        qo[3] = (TI)u0; qo[4] = (TI)u1; qo[5] = (TI)u2; qo[6] = (TI)u3;
        if (so) {
            double nn = sqrt(u0 * u0 + u1 * u1 + u2 * u2), angle = 0.0, a0 = 1.0, a1 = 0.0, a2 = 0.0;
            if (nn != 0.0) {
                angle = 2.0 * atan2(nn, fabs(u3));
                if (u3 < 0.0) nn = -nn;
                a0 = u0 / nn; a1 = u1 / nn; a2 = u2 / nn;
            }
            so[0] = (TI)pn[0]; so[1] = (TI)pn[1]; so[2] = (TI)pn[2];
            so[3] = (TI)(angle * a0); so[4] = (TI)(angle * a1); so[5] = (TI)(angle * a2);
        }
    }
}

// wbcqp_rollout's per-instance totals over the ticks of a roll-out (active-set iterations, ticks whose QP was solved) ride along with
// the integration of each tick: a kernel of their own was one more launch on every tick's critical path (2.4 % of a tick)
struct RollAcc {
    const int* iters;
    int* iters_sum;
    int* ticks_ok;
    int first; // first tick of the roll-out: the totals start from zero
};

template <typename TI>
__global__ __launch_bounds__(256) void integrate_kernel(int batch, int nv, int floating_base, double dt, const TI* __restrict__ q,
                                                        const TI* __restrict__ dq, const TI* __restrict__ x, int ldx,
                                                        const int* __restrict__ status, TI* __restrict__ q_next,
                                                        TI* __restrict__ v_next, TI* __restrict__ q_solver, const RollAcc acc)
{
    const int inst = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (inst >= batch) return;
    if (lane == 0) {
        if (acc.iters_sum) acc.iters_sum[inst] = (acc.first ? 0 : acc.iters_sum[inst]) + acc.iters[inst];
        if (acc.ticks_ok) acc.ticks_ok[inst] = (acc.first ? 0 : acc.ticks_ok[inst]) + ((!status || status[inst] == HQP_OPTIMAL) ? 1 : 0);
    }
    const int nq = floating_base ? nv + 1 : nv;
    integrate_one<TI>(nv, floating_base, dt, q + (size_t)inst * nq, dq + (size_t)inst * nv, x + (size_t)inst * ldx, !status || status[inst] == HQP_OPTIMAL,
                      q_next + (size_t)inst * nq, v_next + (size_t)inst * nv, q_solver ? q_solver + (size_t)inst * nv : nullptr, lane);
}

#endif // __HIPCC__
} // namespace wbcqp
