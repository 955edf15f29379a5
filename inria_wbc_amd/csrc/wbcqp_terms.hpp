// wbcqp_terms.hpp -- the step BEFORE the QP: from (q, v, task references) to the rows of the
// QP record, one wavefront per robot instance.
//
// What it replaces: the upstream half of tsid_->computeProblemData(t, q, dq) (/root/reference/src/controllers/controller.cpp:244):
// pinocchio's computeAllTerms / centerOfMass / ccrba / frame Jacobians (call set listed by RobotModel::update,
// /root/reference/src/utils/robot_model.cpp:83-113) and every task's compute(): the SE(3) law
// (/root/reference/example_project/src/tsid/ex_task.cpp:175-247), TaskMEquality (/root/reference/src/tsid/
// task-momentum-equality.cpp:144-173), TaskSelfCollision (/root/reference/src/tsid/task-self-collision.cpp:84-203), tsid's CoM,
// posture, joint-bounds and Contact6d motion tasks with the gains of /root/reference/src/controllers/tasks.cpp.
//
// Formulation (not pinocchio's): everything is expressed in ONE frame, world-aligned with its origin at the floating
// base (the dynamics do not depend on where the world origin is; references are shifted by the base position once).
// Then a spatial quantity of a subtree is a plain sum over its bodies, bodies are numbered depth-first so a subtree is a
// contiguous lane range, and every composite (inertia for CRBA, bias force for the non-linear effects, momentum) is one
// wave-wide prefix sum and a difference of two entries:
//   lanes = bodies:   joint transform, placement / velocity / bias acceleration down the tree (one step per depth level),
//                     world inertia, momentum, bias force, prefix sums
//   lanes = tasks:    frame placement, velocity, classical acceleration, SE(3) error (log3), right-hand sides
//   lanes = pairs:    self-collision repulsors (one lane per tracked / avoided pair)
//   lanes = columns:  S_j, F_j = Y_subtree S_j, M (row by row, straight into the packed triangle), h, the Jacobian rows of
//                     every task (local frame), CoM and centroidal-momentum rows, self-collision rows
// HBM traffic per instance = state + references in, QP record out; nothing else leaves the CU.
#pragma once

#include "wbcqp_prims.hpp"

namespace wbcqp {

enum { J_FREEFLYER = 0, J_RX = 1, J_RY = 2, J_RZ = 3, J_PX = 4, J_PY = 5, J_PZ = 6 };
enum { T_SE3 = 0, T_COM = 1, T_MOMENTUM = 2, T_SELFCOLLISION = 3 };

constexpr int kKinStride = 25; // per body: R (9) p (3) v (6) a (6), odd stride
constexpr int kScanStride = 17; // per body: m, m c (3), inertia about the origin (6), bias force (6)
constexpr int kFStride = 7;
constexpr int kLawStride = 13;  // per task frame: R (9) p (3)
constexpr int kPairStride = 7;  // per self-collision pair: grad (3), rhs share, tracked body, avoided body

// Constant tables of one (model, task map), resident in device memory; offsets index the two pools.
struct TermsDev {
    int nb, nq, nv, na, floating_base, maxdepth;
    int nlaw, npair, nblock, nc, n_dense, n_sel, n_bound, r1, nref;
    int posture_ref;
    double posture_kp, posture_kd, dt;
    double g[3];
    const int* ipool;
    const double* dpool;
    // int pool offsets
    int i_parent, i_jtype, i_depth, i_last, i_idxq, i_idxv; // [nb]
    int i_bodyof, i_kof;                                    // [nv]
    int i_law_body, i_law_mask, i_law_row, i_law_ref, i_law_va, i_law_contact; // [nlaw]
    int i_pair_block, i_pair_bt, i_pair_ba;                 // [npair]
    int i_blk_kind, i_blk_mask, i_blk_row, i_blk_ref, i_blk_law, i_blk_pair0, i_blk_npair; // [nblock]
    int i_sel_col;                                          // [n_sel]
    // double pool offsets
    int d_place, d_inertia;                                  // [nb][12], [nb][10]
    int d_law_place, d_law_kp, d_law_kd;                     // [nlaw][12], [nlaw], [nlaw]
    int d_pair_pt, d_pair_pa, d_pair_par;                    // [npair][12], [npair][12], [npair][6]: aa, k, s_p, m, kp, kd
    int d_blk_kp, d_blk_kd;                                  // [nblock]
    int d_qlb, d_qub, d_dqmax;                               // [na]
    // LDS layout (doubles)
    int o_state, o_kin, o_scan, o_F, o_law, o_pair, o_b1, o_bc;
    int lds_doubles;
};

template <typename TI>
struct TermsArgs {
    TermsDev T;
    const TI *q, *v, *ref;
    TI *M, *h, *A, *b1, *Ac, *bc, *blb, *bub;
    int batch;
};

#ifdef __HIPCC__

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const double* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(double* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
// R row-major
__device__ __forceinline__ V3 mv(const double* R, V3 a)
{
    return {R[0] * a.x + R[1] * a.y + R[2] * a.z, R[3] * a.x + R[4] * a.y + R[5] * a.z, R[6] * a.x + R[7] * a.y + R[8] * a.z};
}
__device__ __forceinline__ V3 mtv(const double* R, V3 a)
{
    return {R[0] * a.x + R[3] * a.y + R[6] * a.z, R[1] * a.x + R[4] * a.y + R[7] * a.z, R[2] * a.x + R[5] * a.y + R[8] * a.z};
}
__device__ __forceinline__ void mm(const double* A, const double* B, double* O)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) O[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ V3 col(const double* R, int k) { return {R[k], R[3 + k], R[6 + k]}; }
// symmetric 3x3 stored xx xy xz yy yz zz
__device__ __forceinline__ V3 symv(const double* s, V3 a)
{
    return {s[0] * a.x + s[1] * a.y + s[2] * a.z, s[1] * a.x + s[3] * a.y + s[4] * a.z, s[2] * a.x + s[4] * a.y + s[5] * a.z};
}
__device__ __forceinline__ double scan_incl(double v, int lane)
{
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const double t = __shfl_up(v, d, kWave);
        if (lane >= d) v += t;
    }
    return v;
}

// pinocchio log3 [UPSTREAM-RECALL, as oracle/rbd_oracle.c wbco_log3]
__device__ __forceinline__ V3 log3(const double* R)
{
    const double tr = R[0] + R[4] + R[8];
    const double PI = 3.14159265358979323846;
    double theta;
    if (tr > 3.0) theta = 0.0;
    else if (tr < -1.0) theta = PI;
    else theta = acos((tr - 1.0) / 2.0);
    if (theta >= PI - 1e-2) {
        const double cphi = cos(theta - PI), beta = theta * theta / (1.0 + cphi);
        const double t0 = (R[0] + cphi) * beta, t1 = (R[4] + cphi) * beta, t2 = (R[8] + cphi) * beta;
        return {(R[7] > R[5] ? 1.0 : -1.0) * (t0 > 0.0 ? sqrt(t0) : 0.0), (R[2] > R[6] ? 1.0 : -1.0) * (t1 > 0.0 ? sqrt(t1) : 0.0),
                (R[3] > R[1] ? 1.0 : -1.0) * (t2 > 0.0 ? sqrt(t2) : 0.0)};
    }
    const double t = ((theta > 1.220703125e-4) ? theta / sin(theta) : 1.0) / 2.0;
    return {t * (R[7] - R[5]), t * (R[2] - R[6]), t * (R[3] - R[1])};
}

// Frame placement, velocity and classical acceleration in the frame's own axes (tsid RobotWrapper::framePosition /
// frameVelocity / frameClassicAcceleration), from the body's world-aligned kinematics.
struct FrameKin { double R[9]; V3 p, v, w, a, al; };
__device__ __forceinline__ void frame_kin(const double* kin_b, const double* place, FrameKin& f)
{
    mm(kin_b, place, f.R);
    f.p = mv(kin_b, ld3(place + 9)) + ld3(kin_b + 9);
    const V3 ov = ld3(kin_b + 12), ow = ld3(kin_b + 15), oa = ld3(kin_b + 18), oal = ld3(kin_b + 21);
    f.v = mtv(f.R, ov + cross(ow, f.p));
    f.w = mtv(f.R, ow);
    f.al = mtv(f.R, oal);
    f.a = mtv(f.R, oa + cross(oal, f.p)) + cross(f.w, f.v);
}

template <typename TI>
__global__ __launch_bounds__(kWave) void terms_kernel(const TermsArgs<TI> args)
{
    extern __shared__ double lds[];
    const TermsDev& T = args.T;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= args.batch) return;
    const int nb = T.nb, nq = T.nq, nv = T.nv, na = T.na;
    const int* ip = T.ipool;
    const double* dp = T.dpool;
    double* q = lds + T.o_state;
    double* v = q + nq;
    double* ref = v + nv;
    double* kin = lds + T.o_kin;
    double* scan = lds + T.o_scan;
    double* Fl = lds + T.o_F;
    double* law = lds + T.o_law;
    double* pair = lds + T.o_pair;
    double* b1s = lds + T.o_b1;
    double* bcs = lds + T.o_bc;

    // ---- state and references into LDS ----------------------------------------------------------------------------
    {
        const TI* gq = args.q + (size_t)inst * nq;
        const TI* gv = args.v + (size_t)inst * nv;
        const TI* gr = args.ref + (size_t)inst * T.nref;
        for (int i = lane; i < nq; i += kWave) q[i] = (double)gq[i];
        for (int i = lane; i < nv; i += kWave) v[i] = (double)gv[i];
        for (int i = lane; i < T.nref; i += kWave) ref[i] = (double)gr[i];
        for (int i = lane; i < T.r1; i += kWave) b1s[i] = 0.0;
    }
    __syncthreads();
    const V3 p0 = T.floating_base ? ld3(q) : V3{0.0, 0.0, 0.0}; // the origin everything below is expressed about

    // ---- lanes = bodies: joint transform ---------------------------------------------------------------------------
    const bool body = lane < nb;
    const int bi = body ? lane : 0;
    const int jt = ip[T.i_jtype + bi], par = ip[T.i_parent + bi], dep = body ? ip[T.i_depth + bi] : -1;
    const int iq = ip[T.i_idxq + bi], iv = ip[T.i_idxv + bi];
    double Rl[9];
    V3 pl;
    {
        const double* P = dp + T.d_place + 12 * bi;
        if (jt == J_FREEFLYER) {
            const double x = q[3], y = q[4], z = q[5], w = q[6];
            const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
            const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
                         tzz = tz * z;
            Rl[0] = 1 - (tyy + tzz); Rl[1] = txy - twz; Rl[2] = txz + twy;
            Rl[3] = txy + twz; Rl[4] = 1 - (txx + tzz); Rl[5] = tyz - twx;
            Rl[6] = txz - twy; Rl[7] = tyz + twx; Rl[8] = 1 - (txx + tyy);
            pl = {0.0, 0.0, 0.0};
        }
        else if (jt <= J_RZ) {
            double s, c;
            sincos(q[iq], &s, &c);
            // P.R * Rot(axis): the axis column stays, the other two mix
            const int a = jt - J_RX, b = (a + 1) % 3, d = (a + 2) % 3;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double pa = P[3 * r + a], pb = P[3 * r + b], pd = P[3 * r + d];
                Rl[3 * r + a] = pa;
                Rl[3 * r + b] = c * pb + s * pd;
                Rl[3 * r + d] = c * pd - s * pb;
            }
            pl = ld3(P + 9);
        }
        else {
#pragma unroll
            for (int r = 0; r < 9; ++r) Rl[r] = P[r];
            pl = ld3(P + 9) + q[iq] * col(P, jt - J_PX);
        }
    }
    // ---- down the tree, one depth level per step: placement, velocity, bias acceleration (world-aligned, about p0) ----
    double R[9];
    V3 p = {0, 0, 0}, ov = {0, 0, 0}, ow = {0, 0, 0}, oa = {0, 0, 0}, oal = {0, 0, 0};
    for (int d = 0; d <= T.maxdepth; ++d) {
        if (dep == d) {
            V3 pv = {0, 0, 0}, pw = {0, 0, 0}, pa = {0, 0, 0}, pal = {0, 0, 0};
            if (par >= 0) {
                const double* K = kin + kKinStride * par;
                mm(K, Rl, R);
                p = mv(K, pl) + ld3(K + 9);
                pv = ld3(K + 12); pw = ld3(K + 15); pa = ld3(K + 18); pal = ld3(K + 21);
            }
            else {
#pragma unroll
                for (int r = 0; r < 9; ++r) R[r] = Rl[r];
                p = pl;
            }
            V3 jv, jw; // the joint's own velocity, world-aligned
            if (jt == J_FREEFLYER) {
                jw = mv(R, ld3(v + 3));
                jv = mv(R, ld3(v)) + cross(p, jw);
            }
            else if (jt <= J_RZ) {
                jw = v[iv] * col(R, jt - J_RX);
                jv = cross(p, jw);
            }
            else {
                jw = {0, 0, 0};
                jv = v[iv] * col(R, jt - J_PX);
            }
            ov = pv + jv;
            ow = pw + jw;
            // a = a_parent + v x vJ (motion cross product)
            oa = pa + cross(ow, jv) + cross(ov, jw);
            oal = pal + cross(ow, jw);
            double* K = kin + kKinStride * lane;
#pragma unroll
            for (int r = 0; r < 9; ++r) K[r] = R[r];
            st3(K + 9, p); st3(K + 12, ov); st3(K + 15, ow); st3(K + 18, oa); st3(K + 21, oal);
        }
        __syncthreads();
    }
    // ---- world inertia about the origin, momentum, bias force; prefix sums over the depth-first order ----------------
    double sc[16];
    double hm[6];
    {
        const double* Y = dp + T.d_inertia + 10 * bi;
        const double m = body ? Y[0] : 0.0;
        const V3 cw = mv(R, ld3(Y + 1)) + p;
        // R I_c R'
        const double Ic[9] = {Y[4], Y[5], Y[6], Y[5], Y[7], Y[8], Y[6], Y[8], Y[9]};
        double RI[9], Iw[9], Rt[9];
        mm(R, Ic, RI);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i];
        mm(RI, Rt, Iw);
        const double c2 = dot(cw, cw);
        const V3 hc = m * cw;
        double Io[6] = {Iw[0] + m * (c2 - cw.x * cw.x), Iw[1] - m * cw.x * cw.y, Iw[2] - m * cw.x * cw.z,
                        Iw[4] + m * (c2 - cw.y * cw.y), Iw[5] - m * cw.y * cw.z, Iw[8] + m * (c2 - cw.z * cw.z)};
        if (!body) {
#pragma unroll
            for (int r = 0; r < 6; ++r) Io[r] = 0.0;
        }
        // momentum h = Y v, bias force f = Y a + v x* h
        const V3 hl = m * ov + cross(ow, hc);
        const V3 ha = symv(Io, ow) + cross(hc, ov);
        const V3 fl = m * oa + cross(oal, hc) + cross(ow, hl);
        const V3 fa = symv(Io, oal) + cross(hc, oa) + cross(ow, ha) + cross(ov, hl);
        sc[0] = m; sc[1] = hc.x; sc[2] = hc.y; sc[3] = hc.z;
#pragma unroll
        for (int r = 0; r < 6; ++r) sc[4 + r] = Io[r];
        sc[10] = fl.x; sc[11] = fl.y; sc[12] = fl.z; sc[13] = fa.x; sc[14] = fa.y; sc[15] = fa.z;
        hm[0] = hl.x; hm[1] = hl.y; hm[2] = hl.z; hm[3] = ha.x; hm[4] = ha.y; hm[5] = ha.z;
        if (!body) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.0;
#pragma unroll
            for (int r = 0; r < 6; ++r) hm[r] = 0.0;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = scan_incl(sc[r], lane);
#pragma unroll
    for (int r = 0; r < 6; ++r) hm[r] = wave_sum(hm[r]);
    // entry 0 of the table is the empty prefix, entry i + 1 the sum over bodies 0..i
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scan[r] = 0.0;
    }
    if (body) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scan[kScanStride * (lane + 1) + r] = sc[r];
    }
    __syncthreads();
    // totals: mass, centre of mass (about p0), its velocity and bias acceleration, centroidal momentum and its bias rate
    const double* tot = scan + kScanStride * nb;
    const double mass = tot[0], imass = 1.0 / mass;
    const V3 com = imass * ld3(tot + 1);
    const V3 htl = {hm[0], hm[1], hm[2]}, hta = {hm[3], hm[4], hm[5]};
    const V3 vcom = imass * htl, acom = imass * ld3(tot + 10);
    const V3 Lang = hta - cross(com, htl);                    // angular momentum about the com
    const V3 dLang = ld3(tot + 13) - cross(com, ld3(tot + 10)); // its rate at ddq = 0

    // ---- lanes = tasks with a frame (SE(3) blocks, then contacts): the law of ex_task.cpp:175-247, local frame ---------
    if (lane < T.nlaw) {
        FrameKin f;
        frame_kin(kin + kKinStride * ip[T.i_law_body + lane], dp + T.d_law_place + 12 * lane, f);
        double* Lw = law + kLawStride * lane;
#pragma unroll
        for (int r = 0; r < 9; ++r) Lw[r] = f.R[r];
        st3(Lw + 9, f.p);
        const double* rf = ref + ip[T.i_law_ref + lane];
        // errorInSE3: M_err = oMf^-1 M_ref -> (translation, log3(rotation)); the reference rotation is column-major
        const V3 pe = mtv(f.R, (ld3(rf) - p0) - f.p);
        double Rr[9], Re[9], Rft[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { Rr[3 * i + j] = rf[3 + 3 * j + i]; Rft[3 * i + j] = f.R[3 * j + i]; }
        mm(Rft, Rr, Re);
        const V3 we = log3(Re);
        V3 vr = {0, 0, 0}, wr = {0, 0, 0}, ar = {0, 0, 0}, alr = {0, 0, 0};
        if (ip[T.i_law_va + lane]) { // wMl^-1 v_ref, wMl^-1 a_ref (:201,208)
            vr = mtv(f.R, ld3(rf + 12)); wr = mtv(f.R, ld3(rf + 15));
            ar = mtv(f.R, ld3(rf + 18)); alr = mtv(f.R, ld3(rf + 21));
        }
        const double kp = dp[T.d_law_kp + lane], kd = dp[T.d_law_kd + lane];
        const V3 rl = (kp * pe + kd * (vr - f.v) + ar) - f.a;
        const V3 ra = (kp * we + kd * (wr - f.w) + alr) - f.al;
        const double rhs[6] = {rl.x, rl.y, rl.z, ra.x, ra.y, ra.z};
        const int mask = ip[T.i_law_mask + lane], ct = ip[T.i_law_contact + lane];
        double* out = (ct >= 0) ? bcs + 6 * ct : b1s + ip[T.i_law_row + lane];
        int o = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if ((mask >> i) & 1) out[o++] = rhs[i];
    }
    // ---- lanes = self-collision pairs (task-self-collision.cpp:84-203, 5PL repulsor :147-156) -------------------------
    for (int s0 = 0; s0 < T.npair; s0 += kWave) {
        const int s = s0 + lane;
        if (s < T.npair) {
            const int bt = ip[T.i_pair_bt + s], ba = ip[T.i_pair_ba + s];
            FrameKin ft, fa;
            frame_kin(kin + kKinStride * bt, dp + T.d_pair_pt + 12 * s, ft);
            frame_kin(kin + kKinStride * ba, dp + T.d_pair_pa + 12 * s, fa);
            const double* par_ = dp + T.d_pair_par + 6 * s;
            const double aa = par_[0], k5 = par_[1], s_p = par_[2], mm_ = par_[3], kp = par_[4], kd = par_[5];
            const V3 diff = ft.p - fa.p;
            const V3 drift = ft.a - fa.a; // each in its own frame's axes, as the reference subtracts them (:92,131-133)
            // J v for the WORLD Jacobians = difference of the bodies' spatial velocities at the true world origin
            const double* Kt = kin + kKinStride * bt;
            const double* Ka = kin + kKinStride * ba;
            const V3 Jv = (ld3(Kt + 12) + cross(p0, ld3(Kt + 15))) - (ld3(Ka + 12) + cross(p0, ld3(Ka + 15)));
            const double sn = dot(diff, diff), norm = sqrt(sn);
            const double x = k5 * (norm - aa + s_p);
            const double e_p = exp(-x);
            const double pw1 = pow(e_p + 1.0, -mm_ - 1.0);
            const double C = 1.0 - pow(1.0 + e_p, -mm_);
            const double gscale = -1.0 / norm * k5 * mm_ * e_p * pw1;
            const double hh = 1.0 / sn * k5 * k5 * (-mm_ - 1.0) * mm_ * exp(-2.0 * x) * pow(e_p + 1.0, -mm_ - 2.0)
                + 1.0 / sn * k5 * k5 * mm_ * e_p * pw1 + 1.0 / pow(norm, 1.5) * k5 * mm_ * e_p * pw1;
            const double dJv = dot(diff, Jv);
            const double quad = hh * dJv * dJv + gscale * dot(Jv, Jv); // Hess = hh diff diff' + gscale I (:156)
            const V3 gd = gscale * diff;
            const double g2 = dot(gd, kd * Jv - drift);
            double* Pw = pair + kPairStride * s;
            st3(Pw, gd);
            Pw[3] = -(quad + g2 + kp * C);
            Pw[4] = (double)bt;
            Pw[5] = (double)ba;
        }
    }
    // posture: a_des = -Kp (q_a - ref) - Kd v_a (tsid TaskJointPosture; tasks.cpp:203-217)
    for (int r = lane; r < T.n_sel; r += kWave) {
        const int c = ip[T.i_sel_col + r], ja = c - (nv - na);
        b1s[T.n_dense + r] = -T.posture_kp * (q[nq - na + ja] - ref[T.posture_ref + ja]) - T.posture_kd * v[c];
    }
    // CoM and momentum right-hand sides (tsid TaskComEquality; task-momentum-equality.cpp:151-165)
    if (lane == 0) {
        for (int t = 0; t < T.nblock; ++t) {
            const int kind = ip[T.i_blk_kind + t];
            if (kind != T_COM && kind != T_MOMENTUM) continue;
            const int mask = ip[T.i_blk_mask + t];
            const double* rf = ref + ip[T.i_blk_ref + t];
            const double kp = dp[T.d_blk_kp + t], kd = dp[T.d_blk_kd + t];
            double* out = b1s + ip[T.i_blk_row + t];
            int o = 0;
            if (kind == T_COM) {
                const V3 e = com - (ld3(rf) - p0);
                const V3 r = (-kp * e - kd * (vcom - ld3(rf + 3)) + ld3(rf + 6)) - acom;
                const double rr[3] = {r.x, r.y, r.z};
                for (int i = 0; i < 3; ++i)
                    if ((mask >> i) & 1) out[o++] = rr[i];
            }
            else {
                const double L[6] = {htl.x, htl.y, htl.z, Lang.x, Lang.y, Lang.z};
                const double dL[6] = {tot[10], tot[11], tot[12], dLang.x, dLang.y, dLang.z};
                for (int i = 0; i < 6; ++i)
                    if ((mask >> i) & 1) out[o++] = (-kp * (L[i] - rf[i]) + rf[6 + i]) - dL[i];
            }
        }
    }
    __syncthreads();
    // self-collision right-hand side: the sum over the pairs of a block
    if (lane < T.nblock && ip[T.i_blk_kind + lane] == T_SELFCOLLISION) {
        double B = 0.0;
        const int s0 = ip[T.i_blk_pair0 + lane], ns = ip[T.i_blk_npair + lane];
        for (int s = s0; s < s0 + ns; ++s) B += pair[kPairStride * s + 3];
        b1s[ip[T.i_blk_row + lane]] = B;
    }

    // ---- lanes = velocity coordinates --------------------------------------------------------------------------------
    const bool colv = lane < nv;
    const int cj = colv ? lane : 0;
    const int bj = ip[T.i_bodyof + cj], kj = ip[T.i_kof + cj];
    const int lastj = ip[T.i_last + bj];
    V3 Sv, Sw;
    {
        const double* K = kin + kKinStride * bj;
        const int jtj = ip[T.i_jtype + bj];
        const V3 pj = ld3(K + 9);
        if (jtj == J_FREEFLYER) {
            if (kj < 3) { Sv = col(K, kj); Sw = {0, 0, 0}; }
            else { Sw = col(K, kj - 3); Sv = cross(pj, Sw); }
        }
        else if (jtj <= J_RZ) { Sw = col(K, jtj - J_RX); Sv = cross(pj, Sw); }
        else { Sv = col(K, jtj - J_PX); Sw = {0, 0, 0}; }
    }
    V3 Fv, Fw;
    {
        // composite of the subtree of body bj: prefix(last + 1) - prefix(bj)
        const double* hi = scan + kScanStride * (lastj + 1);
        const double* lo = scan + kScanStride * bj;
        double Y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[r] = hi[r] - lo[r];
        const V3 hc = {Y[1], Y[2], Y[3]};
        Fv = Y[0] * Sv + cross(Sw, hc);
        Fw = symv(Y + 4, Sw) + cross(hc, Sv);
        if (colv) { st3(Fl + kFStride * lane, Fv); st3(Fl + kFStride * lane + 3, Fw); }
        // non-linear effects: bias force of the subtree + its weight, projected on the joint axis
        const V3 gvec = {T.g[0], T.g[1], T.g[2]};
        const V3 gl = ld3(Y + 10) - Y[0] * gvec;
        const V3 ga = ld3(Y + 13) - cross(hc, gvec);
        if (colv) args.h[(size_t)inst * nv + lane] = (TI)(dot(Sv, gl) + dot(Sw, ga));
    }
    __syncthreads();
    // M, row by row into the packed lower triangle: M(i, j) = S_j . F_i for j an ancestor dof of i (crba)
    {
        TI* Mo = args.M + (size_t)inst * (nv * (nv + 1) / 2);
        for (int i = 0; i < nv; ++i) {
            const int b_i = ip[T.i_bodyof + i];
            const double* F = Fl + kFStride * i;
            const double val = dot(Sv, ld3(F)) + dot(Sw, ld3(F + 3));
            if (lane <= i) Mo[i * (i + 1) / 2 + lane] = (TI)((bj <= b_i && b_i <= lastj) ? val : 0.0);
        }
    }
    // Jacobian rows of the tasks with a frame: local frame, rows picked by the mask (ex_task.cpp:233-236)
    if (colv) {
        TI* Ao = args.A + (size_t)inst * T.n_dense * nv;
        TI* Aco = args.Ac + (size_t)inst * T.nc * 6 * nv;
        for (int l = 0; l < T.nlaw; ++l) {
            const double* Lw = law + kLawStride * l;
            const int bl = ip[T.i_law_body + l], mask = ip[T.i_law_mask + l], ct = ip[T.i_law_contact + l];
            const bool sup = bj <= bl && bl <= lastj;
            const V3 pf = ld3(Lw + 9);
            const V3 jl = mtv(Lw, Sv + cross(Sw, pf)), ja = mtv(Lw, Sw);
            const double e[6] = {jl.x, jl.y, jl.z, ja.x, ja.y, ja.z};
            TI* out = (ct >= 0) ? Aco + (size_t)ct * 6 * nv : Ao + (size_t)ip[T.i_law_row + l] * nv;
            int o = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if ((mask >> i) & 1) { out[(size_t)o * nv + lane] = (TI)(sup ? e[i] : 0.0); ++o; }
        }
        // CoM, momentum and self-collision rows
        const V3 lt = Sv + cross(p0, Sw); // WORLD Jacobian column (linear part): about the true world origin
        for (int t = 0; t < T.nblock; ++t) {
            const int kind = ip[T.i_blk_kind + t];
            if (kind == T_SE3) continue;
            TI* out = Ao + (size_t)ip[T.i_blk_row + t] * nv;
            const int mask = ip[T.i_blk_mask + t];
            if (kind == T_COM) {
                const double e[3] = {imass * Fv.x, imass * Fv.y, imass * Fv.z};
                int o = 0;
                for (int i = 0; i < 3; ++i)
                    if ((mask >> i) & 1) { out[(size_t)o * nv + lane] = (TI)e[i]; ++o; }
            }
            else if (kind == T_MOMENTUM) {
                const V3 an = Fw - cross(com, Fv);
                const double e[6] = {Fv.x, Fv.y, Fv.z, an.x, an.y, an.z};
                int o = 0;
                for (int i = 0; i < 6; ++i)
                    if ((mask >> i) & 1) { out[(size_t)o * nv + lane] = (TI)e[i]; ++o; }
            }
            else {
                // A = sum grad_C' (J_tracked - J_avoided): the WORLD columns coincide wherever both frames hang on dof j
                V3 acc = {0, 0, 0};
                const int s0 = ip[T.i_blk_pair0 + t], ns = ip[T.i_blk_npair + t];
                for (int s = s0; s < s0 + ns; ++s) {
                    const double* Pw = pair + kPairStride * s;
                    const int bt = (int)Pw[4], ba = (int)Pw[5];
                    const double sg = (double)((bj <= bt && bt <= lastj) ? 1 : 0) - (double)((bj <= ba && ba <= lastj) ? 1 : 0);
                    acc = acc + sg * ld3(Pw);
                }
                out[lane] = (TI)dot(acc, lt);
            }
        }
    }
    // joint bounds: tsid TaskJointPosVelAccBounds::computeAccLimits [UPSTREAM-RECALL, as oracle/rbd_oracle.c]
    for (int j = lane; j < T.n_bound; j += kWave) {
        const double dt = T.dt, qj = q[nq - na + j], dq = v[nv - na + j];
        const double qmin = dp[T.d_qlb + j], qmax = dp[T.d_qub + j], dqmax = dp[T.d_dqmax + j], ddqmax = dqmax / dt;
        const double two_dt_sq = 2.0 / (dt * dt), mdq_dt = -dq / dt;
        const double max_q3 = two_dt_sq * (qmax - qj - dt * dq), min_q3 = two_dt_sq * (qmin - qj - dt * dq);
        double lb_pos, ub_pos;
        if (dq <= 0.0) {
            ub_pos = max_q3;
            if (min_q3 < mdq_dt) lb_pos = min_q3;
            else if (qj != qmin) lb_pos = fmax(dq * dq / (2.0 * (qj - qmin)), mdq_dt);
            else lb_pos = 1e6;
        }
        else {
            lb_pos = min_q3;
            if (max_q3 > mdq_dt) ub_pos = max_q3;
            else if (qj != qmax) ub_pos = fmin(-dq * dq / (2.0 * (qmax - qj)), mdq_dt);
            else ub_pos = -1e6;
        }
        const double lb_vel = (-dqmax - dq) / dt, ub_vel = (dqmax - dq) / dt;
        const double dt_dq = dt * dq, two_a = 2.0 * dt * dt, dt_ddq_dt = ddqmax * dt * dt;
        const double b_1 = 2.0 * dt_dq + dt_ddq_dt, b_2 = 2.0 * dt_dq - dt_ddq_dt;
        const double c_1 = dq * dq - 2.0 * ddqmax * (qmax - (qj + dt_dq)), c_2 = dq * dq - 2.0 * ddqmax * ((qj + dt_dq) - qmin);
        const double delta_1 = b_1 * b_1 - 2.0 * two_a * c_1, delta_2 = b_2 * b_2 - 2.0 * two_a * c_2;
        const double ub_via = delta_1 >= 0.0 ? (-b_1 + sqrt(delta_1)) / two_a : mdq_dt;
        const double lb_via = delta_2 >= 0.0 ? (-b_2 - sqrt(delta_2)) / two_a : mdq_dt;
        double lb = fmax(fmax(lb_pos, lb_via), fmax(lb_vel, -ddqmax));
        double ub = fmin(fmin(ub_pos, ub_via), fmin(ub_vel, ddqmax));
        if (ub < lb) {
            if (ub == ub_pos) lb = ub;
            else ub = lb;
        }
        args.blb[(size_t)inst * T.n_bound + j] = (TI)lb;
        args.bub[(size_t)inst * T.n_bound + j] = (TI)ub;
    }
    __syncthreads();
    for (int i = lane; i < T.r1; i += kWave) args.b1[(size_t)inst * T.r1 + i] = (TI)b1s[i];
    for (int i = lane; i < 6 * T.nc; i += kWave) args.bc[(size_t)inst * 6 * T.nc + i] = (TI)bcs[i];
}

#endif // __HIPCC__
} // namespace wbcqp
