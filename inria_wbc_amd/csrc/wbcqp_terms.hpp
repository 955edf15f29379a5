// wbcqp_terms.hpp -- the step BEFORE the QP: from (q, v, task references) to the rows of the
// QP record, one workgroup of four wavefronts per robot instance.
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
//   lanes = bodies  (wave 0):    joint transform; placement, velocity and bias acceleration down the tree by ancestor doubling
//                                (ceil(log2(depth + 1)) rounds of ds_bpermute); world inertia, momentum, bias force, prefix sums
//   lanes = tasks   (wave 1):    frame placement, velocity, classical acceleration, SE(3) error (log3), right-hand sides
//   lanes = frames, then pairs (wave 2): the distinct self-collision frames, then one repulsor per tracked / avoided pair
//   lanes = joints  (wave 3):    posture right-hand side, joint bounds
//   lanes = columns (every wave, a quarter of the rows each): S_j, F_j = Y_subtree S_j, M (row by row, straight into the packed
//                                triangle), h, the Jacobian rows of every task (local frame), CoM / momentum / self-collision rows
// HBM traffic per instance = state + references in, QP record out; nothing else leaves the CU.
#pragma once

#include "wbcqp_prims.hpp"

namespace wbcqp {

enum { J_FREEFLYER = 0, J_RX = 1, J_RY = 2, J_RZ = 3, J_PX = 4, J_PY = 5, J_PZ = 6 };
enum { T_SE3 = 0, T_COM = 1, T_MOMENTUM = 2, T_SELFCOLLISION = 3 };

constexpr int kKinStride = 25; // per body: R (9) p (3) v (6) a (6), odd stride
constexpr int kScanStride = 17; // per body: m, m c (3), inertia about the origin (6), bias force (6)
constexpr int kSFStride = 13;   // per column: S (6), F (6)
constexpr int kLawStride = 13;  // per task frame: R (9) p (3)
constexpr int kPairStride = 7;  // per self-collision pair: grad (3), rhs share, tracked body, avoided body
constexpr int kScfStride = 7;   // per self-collision frame: position (3), classical linear acceleration in its own axes (3)

// Constant tables of one (model, task map), resident in device memory; offsets index the two pools.
struct TermsDev {
    int nb, nq, nv, na, floating_base, nrounds;
    int nlaw, npair, nscf, nblock, nc, n_dense, n_sel, n_bound, r1, nref;
    int posture_ref;
    int cop;                                                // 1: the structure has a cop task -- the contact lanes also write its rows
    double posture_kp, posture_kd, dt;
    double g[3];
    const int* ipool;
    const double* dpool;
    // int pool offsets
    int i_jtype, i_last, i_idxq, i_idxv;                    // [nb]
    int i_anc;                                              // [nrounds][nb] 2^r-th ancestor, -1 beyond the root
    int i_bodyof, i_kof;                                    // [nv]
    int i_law_body, i_law_mask, i_law_row, i_law_ref, i_law_va, i_law_contact; // [nlaw]
    int i_pair_bt, i_pair_ba, i_pair_ft, i_pair_fa;         // [npair] bodies and self-collision-frame indices
    int i_scf_body;                                         // [nscf] distinct frames the self-collision tasks touch
    int i_blk_kind, i_blk_mask, i_blk_row, i_blk_ref, i_blk_pair0, i_blk_npair; // [nblock]
    int i_sel_col;                                          // [n_sel]
    // double pool offsets
    int d_place, d_inertia;                                  // [nb][12], [nb][10]
    int d_law_place, d_law_kp, d_law_kd;                     // [nlaw][12], [nlaw], [nlaw]
    int d_scf_place, d_pair_par;                             // [nscf][12], [npair][6]: aa, k, s_p, m, kp, kd
    int d_blk_kp, d_blk_kd;                                  // [nblock]
    int d_qlb, d_qub, d_dqmax;                               // [na]
    int d_cop_pts;                                           // [nc][4][3] contact points in the contact frame (cop task)
    // LDS layout (doubles)
    int o_state, o_kin, o_scan, o_tot, o_sf, o_law, o_pair, o_scf, o_b1, o_bc;
    int lds_doubles;
};

template <typename TI>
struct TermsArgs {
    TermsDev T;
    const TI *q, *v, *ref;
    TI *M, *h, *A, *b1, *Ac, *bc, *blb, *bub;
    TI* Acop;     // [batch][3][12 nc] rows of the cop task (T.cop), or null
    TI* momentum; // [batch][6] centroidal momentum (linear, angular about the CoM), or null
    int batch;
    long long* dbg; // per-instance phase cycle counters, only written by the WBCQP_STAMPS diagnostic build
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

// sin and cos of a joint angle: Cody-Waite reduction by pi/2 in two fused steps and the fdlibm kernel polynomials on
// [-pi/4, pi/4] -- some 35 instructions where the library's sincos() is over a hundred (it carries the reduction for
// arguments of any size).  Under one ulp up to |x| = 1e5, which no joint angle reaches; beyond that the library's.
__device__ __forceinline__ void sincos_joint(double x, double* sn, double* cs)
{
    if (!(fabs(x) < 1.0e5)) {
        sincos(x, sn, cs);
        return;
    }
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double S = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double C = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double s0 = (q & 1) ? C : S, c0 = (q & 1) ? S : C;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
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
    // theta / (2 sin theta): 2 sin(theta) is the length of the antisymmetric part itself (one square root where sin() is ~150
    // instructions; the two agree to rounding on (1.2e-4, pi - 1e-2))
    const V3 w = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double t = (theta > 1.220703125e-4) ? theta / sqrt(dot(w, w)) : 0.5;
    return {t * w.x, t * w.y, t * w.z};
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

#ifdef WBCQP_STAMPS
#define TSTAMP(i) { const long long now_ = clock64(); tacc_[i] += now_ - tprev_; tprev_ = now_; }
#else
#define TSTAMP(i)
#endif

__device__ __forceinline__ int rl(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// Column j of the world-aligned joint Jacobian: the joint's motion subspace column in its own axes, moved by oMi.act(.)
struct ColumnAxis { int body, last; V3 Sv, Sw; };
// (body, last, kj, jtj: the column's constants -- its body, the last body of that body's subtree, the index of the dof inside its
// joint, the joint type -- fetched once per thread at the top of the kernel)
__device__ __forceinline__ ColumnAxis column_axis(const double* kin, int body, int last, int kj, int jtj)
{
    ColumnAxis c;
    c.body = body;
    c.last = last;
    const double* K = kin + kKinStride * c.body;
    const int a = (jtj == J_FREEFLYER) ? (kj % 3) : (jtj <= J_RZ) ? jtj - J_RX : jtj - J_PX;
    const bool ang = (jtj == J_FREEFLYER) ? (kj >= 3) : (jtj <= J_RZ);
    const V3 ax = col(K, 0), ay = col(K, 1), az = col(K, 2);
    const V3 wa = (a == 0) ? ax : (a == 1) ? ay : az; // world direction of the axis
    const V3 pj = ld3(K + 9);
    if (ang) { c.Sw = wa; c.Sv = cross(pj, wa); }
    else { c.Sv = wa; c.Sw = {0, 0, 0}; }
    return c;
}

// Jacobian rows of the framed tasks l = first, first + step, ...: local frame, rows picked by the mask (ex_task.cpp:233-236).
// Lane = column; lane l of the calling wave holds task l's table entries, the loop reads them back as uniform values.
template <typename TI>
__device__ __forceinline__ void jacobian_rows(const TermsDev& T, const double* law, const ColumnAxis& c, bool colv, int lane, int first, int step,
                                              TI* Ao, TI* Aco, int nv)
{
    const int* ip = T.ipool;
    const int ll = min(lane, max(T.nlaw - 1, 0));
    const int t_body = ip[T.i_law_body + ll], t_mask = ip[T.i_law_mask + ll], t_ct = ip[T.i_law_contact + ll], t_row = ip[T.i_law_row + ll];
    for (int l = first; l < T.nlaw; l += step) {
        const double* Lw = law + kLawStride * l;
        const int bl = rl(t_body, l), mask = rl(t_mask, l), ct = rl(t_ct, l), row = rl(t_row, l);
        const bool sup = c.body <= bl && bl <= c.last;
        const V3 pf = ld3(Lw + 9);
        const V3 jl = mtv(Lw, c.Sv + cross(c.Sw, pf)), ja = mtv(Lw, c.Sw);
        const double e[6] = {jl.x, jl.y, jl.z, ja.x, ja.y, ja.z};
        TI* out = (ct >= 0) ? Aco + (size_t)ct * 6 * nv : Ao + (size_t)row * nv;
        int o = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if ((mask >> i) & 1) {
                if (colv) out[(size_t)o * nv + lane] = (TI)(sup ? e[i] : 0.0);
                ++o;
            }
    }
}

constexpr int kTermsThreads = 256; // four wavefronts per instance: one runs the tree, all four share the rows

// One workgroup of four wavefronts per instance.
//   phase 1  wave 0: joint transforms and the sweep down the tree;  wave 3 meanwhile: posture right-hand side, joint bounds
//   phase 2  wave 0: world inertias, bias forces, prefix sums;  wave 1: task frames (published at once, with a flag), their laws, a
//            third of their Jacobian rows;  wave 2: self-collision frames and pairs;  wave 3: the other two thirds of the Jacobian rows
//   phase 3  every wave: S_j and F_j of its lanes' columns (from wave 0, through LDS), then a quarter of the remaining rows each -- rows of M,
//            CoM / momentum / self-collision rows
// inst: which instance's state and references are read; rinst: where the record goes in the row arrays (the same index for the
// kernel below; wbcqp_rollout's persistent workgroups keep ONE record slot each and pass their own index)
template <typename TI>
__device__ __forceinline__ void terms_one(const TermsArgs<TI>& args, const TI* gq, const TI* gv, const TI* gr, TI* gmom, const int inst,
                                          const int rinst, double* lds, const int tid)
{
    const TermsDev& T = args.T;
    const int lane = tid & (kWave - 1);
    // no rotation of the roles: the hardware itself starts co-resident workgroups on different SIMDs (measured with
    // tools/ubench/wave_placement.hip: wave 0 of the four workgroups of a CU lands on SIMD 2, 1, 3, 0), so their tree waves
    // already sit on four different SIMDs
    const int wave = uni(tid >> 6); // told to the compiler as wave-uniform: scalar branches on the role, scalar row counters
#ifdef WBCQP_STAMPS
    long long tacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = clock64();
#endif
    const int nb = T.nb, nq = T.nq, nv = T.nv, na = T.na;
    const int* ip = T.ipool;
    const double* dp = T.dpool;
    double* q = lds + T.o_state;
    double* v = q + nq;
    double* ref = v + nv;
    double* kin = lds + T.o_kin;
    double* scan = lds + T.o_scan;
    double* law = lds + T.o_law;
    double* pair = lds + T.o_pair;
    double* b1s = lds + T.o_b1;
    double* bcs = lds + T.o_bc;
    double* tots = lds + T.o_tot; // total momentum of the robot (6 doubles)
    int* frames_ready = reinterpret_cast<int*>(tots + 6); // wave 1 has published the task frames (wave 3 waits for it, nobody else)

    // the constants of this lane's column (two dependent global reads): asked for here, they arrive behind the state; phases 2 and 3
    // used to fetch them again on every wave, the second time right after a barrier
    const int cj = min(lane, nv - 1);
    const int bj = ip[T.i_bodyof + cj], kofj = ip[T.i_kof + cj];
    const int lastj = ip[T.i_last + bj], jtypej = ip[T.i_jtype + bj];
    // ---- state and references into LDS ----------------------------------------------------------------------------
    {
        // one pass: every thread fetches its elements of the three arrays before any of them is stored
        const int n1 = nq, n2 = nq + nv, n3 = nq + nv + T.nref;
        TI buf[2];
        int idx[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * kTermsThreads;
            idx[u] = e;
            const int ec = min(e, n3 - 1);
            buf[u] = (ec < n1) ? gq[ec] : (ec < n2) ? gv[ec - n1] : gr[ec - n2];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (idx[u] < n3) q[idx[u]] = (double)buf[u];
        for (int e = tid + 2 * kTermsThreads; e < n3; e += kTermsThreads)
            q[e] = (double)((e < n1) ? gq[e] : (e < n2) ? gv[e - n1] : gr[e - n2]);
        for (int i = tid; i < T.r1; i += kTermsThreads) b1s[i] = 0.0;
        if (tid == 0) __hip_atomic_store(frames_ready, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    bsync();
    TSTAMP(0)
    const V3 p0 = T.floating_base ? ld3(q) : V3{0.0, 0.0, 0.0}; // the origin everything below is expressed about

    if (wave == 0) {
        // ---- lanes = bodies: joint transform ----------------------------------------------------------------------
        const bool body = lane < nb;
        const int bi = body ? lane : 0;
        const int jt = ip[T.i_jtype + bi];
        const int iq = ip[T.i_idxq + bi], iv = ip[T.i_idxv + bi];
        double Yb[10]; // this body's inertia: fetched now, used after the sweep
#pragma unroll
        for (int r = 0; r < 10; ++r) Yb[r] = dp[T.d_inertia + 10 * bi + r];
        double Rl[9];
        V3 pl, vJ, wJ; // joint placement in the parent, joint velocity in the joint's own axes
        {
            const double* P = dp + T.d_place + 12 * bi;
            if (jt == J_FREEFLYER) {
                const double x = q[3], y = q[4], z = q[5], w = q[6];
                const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
                const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                             tyz = tz * y, tzz = tz * z;
                Rl[0] = 1 - (tyy + tzz); Rl[1] = txy - twz; Rl[2] = txz + twy;
                Rl[3] = txy + twz; Rl[4] = 1 - (txx + tzz); Rl[5] = tyz - twx;
                Rl[6] = txz - twy; Rl[7] = tyz + twx; Rl[8] = 1 - (txx + tyy);
                pl = {0.0, 0.0, 0.0};
                vJ = ld3(v); wJ = ld3(v + 3);
            }
            else {
                const int a = (jt <= J_RZ) ? jt - J_RX : jt - J_PX;
                const double qd = v[iv];
                const V3 e = {a == 0 ? qd : 0.0, a == 1 ? qd : 0.0, a == 2 ? qd : 0.0};
                if (jt <= J_RZ) {
                    double sn, cs;
                    sincos_joint(q[iq], &sn, &cs);
                    // P.R * Rot(axis): the axis column stays, the other two mix
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const double c0 = P[3 * r], c1 = P[3 * r + 1], c2 = P[3 * r + 2];
                        const double pa = (a == 0) ? c0 : (a == 1) ? c1 : c2;
                        const double pb = (a == 0) ? c1 : (a == 1) ? c2 : c0;
                        const double pd = (a == 0) ? c2 : (a == 1) ? c0 : c1;
                        const double nb_ = cs * pb + sn * pd, nd_ = cs * pd - sn * pb;
                        Rl[3 * r] = (a == 0) ? pa : (a == 1) ? nd_ : nb_;
                        Rl[3 * r + 1] = (a == 0) ? nb_ : (a == 1) ? pa : nd_;
                        Rl[3 * r + 2] = (a == 0) ? nd_ : (a == 1) ? nb_ : pa;
                    }
                    pl = ld3(P + 9);
                    vJ = {0.0, 0.0, 0.0}; wJ = e;
                }
                else {
#pragma unroll
                    for (int r = 0; r < 9; ++r) Rl[r] = P[r];
                    const V3 ax = {a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0};
                    pl = ld3(P + 9) + q[iq] * mv(P, ax);
                    vJ = e; wJ = {0.0, 0.0, 0.0};
                }
            }
        }
        TSTAMP(1)
        // ---- down the tree by ancestor doubling: after round r every body holds the composition over its 2^(r+1) nearest
        //      ancestors-and-self, so ceil(log2(depth + 1)) rounds give placements, then velocities, then bias accelerations
        //      (a path sum each) -- rigid transforms compose associatively, so the order of the products is free ------------
        int anc[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) anc[r] = (r < T.nrounds) ? ip[T.i_anc + r * nb + bi] : -1;
        double R[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) R[r] = Rl[r];
        V3 p = pl;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (r < T.nrounds) {
                const int src = anc[r] >= 0 ? anc[r] : lane;
                double Ra[9], Rn[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) Ra[k] = __shfl(R[k], src, kWave);
                const V3 pa = {__shfl(p.x, src, kWave), __shfl(p.y, src, kWave), __shfl(p.z, src, kWave)};
                if (anc[r] >= 0) {
                    mm(Ra, R, Rn);
                    p = mv(Ra, p) + pa;
#pragma unroll
                    for (int k = 0; k < 9; ++k) R[k] = Rn[k];
                }
            }
        }
        const V3 jw = mv(R, wJ);               // the joint's own velocity, world-aligned
        const V3 jv = mv(R, vJ) + cross(p, jw);
        V3 ov = jv, ow = jw;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (r < T.nrounds) {
                const int src = anc[r] >= 0 ? anc[r] : lane;
                const V3 a = {__shfl(ov.x, src, kWave), __shfl(ov.y, src, kWave), __shfl(ov.z, src, kWave)};
                const V3 b = {__shfl(ow.x, src, kWave), __shfl(ow.y, src, kWave), __shfl(ow.z, src, kWave)};
                if (anc[r] >= 0) { ov = ov + a; ow = ow + b; }
            }
        }
        // a = a_parent + v x vJ (motion cross product): a path sum of the bias terms
        V3 oa = cross(ow, jv) + cross(ov, jw), oal = cross(ow, jw);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (r < T.nrounds) {
                const int src = anc[r] >= 0 ? anc[r] : lane;
                const V3 a = {__shfl(oa.x, src, kWave), __shfl(oa.y, src, kWave), __shfl(oa.z, src, kWave)};
                const V3 b = {__shfl(oal.x, src, kWave), __shfl(oal.y, src, kWave), __shfl(oal.z, src, kWave)};
                if (anc[r] >= 0) { oa = oa + a; oal = oal + b; }
            }
        }
        if (body) {
            double* K = kin + kKinStride * lane;
#pragma unroll
            for (int r = 0; r < 9; ++r) K[r] = R[r];
            st3(K + 9, p); st3(K + 12, ov); st3(K + 15, ow); st3(K + 18, oa); st3(K + 21, oal);
        }
        TSTAMP(2)
        // ---- phase 2 on this wave: world inertia about the origin, momentum, bias force; prefix sums ---------------
        bsync(); // barrier 1: kin is complete
        double sc[16];
        double hm[6];
        {
            const double* Y = Yb;
            const double m = body ? Y[0] : 0.0;
            const V3 cw = mv(R, ld3(Y + 1)) + p;
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
            const double Io[6] = {Iw[0] + m * (c2 - cw.x * cw.x), Iw[1] - m * cw.x * cw.y, Iw[2] - m * cw.x * cw.z,
                                  Iw[4] + m * (c2 - cw.y * cw.y), Iw[5] - m * cw.y * cw.z, Iw[8] + m * (c2 - cw.z * cw.z)};
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
        for (int r = 0; r < 6; ++r) hm[r] = wave_sum(hm[r]);
        // Prefix sums over the bodies through the table itself: entry 0 is the empty prefix, entry i + 1 the sum over bodies
        // 0..i.  The raw values go in first; then lane (r = lane & 15, chunk = lane >> 4) runs the sum of value r over the 16
        // bodies of its chunk, the chunk totals are exchanged, and the offsets of the earlier chunks are added on the way back.
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scan[r] = 0.0;
#pragma unroll
            for (int r = 0; r < 6; ++r) tots[r] = hm[r];
        }
        if (body) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scan[kScanStride * (lane + 1) + r] = sc[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const int r = lane & 15, chunk = lane >> 4;
            double vals[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int b = 16 * chunk + k;
                vals[k] = scan[kScanStride * (min(b, nb - 1) + 1) + r];
            }
            double run = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                run += (16 * chunk + k < nb) ? vals[k] : 0.0;
                vals[k] = run;
            }
            const double t0 = __shfl(run, r, kWave), t1 = __shfl(run, 16 + r, kWave), t2 = __shfl(run, 32 + r, kWave);
            const double off = (chunk == 0) ? 0.0 : (chunk == 1) ? t0 : (chunk == 2) ? (t0 + t1) : ((t0 + t1) + t2);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int b = 16 * chunk + k;
                if (b < nb) scan[kScanStride * (b + 1) + r] = vals[k] + off;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lanes = columns: S_j, the composite of the subtree under it, F_j = Y S_j, and the non-linear effects h_j; into LDS for
        // the four waves of phase 3 (this wave has time to spare here: the task-law wave is the long one in phase 2)
        {
            const bool colv = lane < nv;
            const ColumnAxis c = column_axis(kin, bj, lastj, kofj, jtypej);
            const double* hi = scan + kScanStride * (c.last + 1); // prefix(last + 1) - prefix(body)
            const double* lo = scan + kScanStride * c.body;
            double Y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[r] = hi[r] - lo[r];
            const V3 hc = {Y[1], Y[2], Y[3]};
            const V3 Fv = Y[0] * c.Sv + cross(c.Sw, hc);
            const V3 Fw = symv(Y + 4, c.Sw) + cross(hc, c.Sv);
            if (colv) {
                double* SF = lds + T.o_sf + kSFStride * lane;
                st3(SF, c.Sv); st3(SF + 3, c.Sw); st3(SF + 6, Fv); st3(SF + 9, Fw);
                // bias force of the subtree + its weight, projected on the joint axis
                const V3 gvec = {T.g[0], T.g[1], T.g[2]};
                const V3 gl = ld3(Y + 10) - Y[0] * gvec;
                const V3 ga = ld3(Y + 13) - cross(hc, gvec);
                args.h[(size_t)rinst * nv + lane] = (TI)(dot(c.Sv, gl) + dot(c.Sw, ga));
            }
        }
        TSTAMP(3)
    }
    else if (wave == 3) {
        // ---- phase 1 beside the tree sweep: what needs the state only -------------------------------------------------
        // posture: a_des = -Kp (q_a - ref) - Kd v_a (tsid TaskJointPosture; tasks.cpp:203-217)
        for (int r = lane; r < T.n_sel; r += kWave) {
            const int c = ip[T.i_sel_col + r], ja = c - (nv - na);
            b1s[T.n_dense + r] = -T.posture_kp * (q[nq - na + ja] - ref[T.posture_ref + ja]) - T.posture_kd * v[c];
        }
        // joint bounds: tsid TaskJointPosVelAccBounds::computeAccLimits [UPSTREAM-RECALL, as oracle/rbd_oracle.c]
        for (int j = lane; j < T.n_bound; j += kWave) {
            // (the divisions by dt and by 2 dt^2 are multiplications by reciprocals formed once: a double division is ~30 instructions)
            const double dt = T.dt, idt = 1.0 / dt, qj = q[nq - na + j], dq = v[nv - na + j];
            const double qmin = dp[T.d_qlb + j], qmax = dp[T.d_qub + j], dqmax = dp[T.d_dqmax + j], ddqmax = dqmax * idt;
            const double two_dt_sq = 2.0 * idt * idt, mdq_dt = -dq * idt;
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
            const double lb_vel = (-dqmax - dq) * idt, ub_vel = (dqmax - dq) * idt;
            const double dt_dq = dt * dq, two_a = 2.0 * dt * dt, i_two_a = 0.5 * idt * idt, dt_ddq_dt = ddqmax * dt * dt;
            const double b_1 = 2.0 * dt_dq + dt_ddq_dt, b_2 = 2.0 * dt_dq - dt_ddq_dt;
            const double c_1 = dq * dq - 2.0 * ddqmax * (qmax - (qj + dt_dq)), c_2 = dq * dq - 2.0 * ddqmax * ((qj + dt_dq) - qmin);
            const double delta_1 = b_1 * b_1 - 2.0 * two_a * c_1, delta_2 = b_2 * b_2 - 2.0 * two_a * c_2;
            const double ub_via = delta_1 >= 0.0 ? (-b_1 + sqrt(delta_1)) * i_two_a : mdq_dt;
            const double lb_via = delta_2 >= 0.0 ? (-b_2 - sqrt(delta_2)) * i_two_a : mdq_dt;
            double lb = fmax(fmax(lb_pos, lb_via), fmax(lb_vel, -ddqmax));
            double ub = fmin(fmin(ub_pos, ub_via), fmin(ub_vel, ddqmax));
            if (ub < lb) {
                if (ub == ub_pos) lb = ub;
                else ub = lb;
            }
            args.blb[(size_t)rinst * T.n_bound + j] = (TI)lb;
            args.bub[(size_t)rinst * T.n_bound + j] = (TI)ub;
        }
        // ---- phase 2 on this wave: the Jacobian rows of two framed tasks in three (wave 1 takes the others).  The frames come
        //      from wave 1 (round 2 recomputed them here, ~200 instructions on the vector port this kernel saturates): wave 1
        //      publishes them first thing after the barrier and raises a flag; this wave sleeps on the flag, which costs no issue slot
        bsync(); // barrier 1: kin is complete
        {
            const bool colv = lane < nv;
            const ColumnAxis c = column_axis(kin, bj, lastj, kofj, jtypej);
            while (__hip_atomic_load(frames_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // two tasks in three here, one in three on wave 1, which also evaluates the laws
            jacobian_rows<TI>(T, law, c, colv, lane, 1, 3, args.A + (size_t)rinst * T.n_dense * nv, args.Ac + (size_t)rinst * T.nc * 6 * nv, nv);
            jacobian_rows<TI>(T, law, c, colv, lane, 2, 3, args.A + (size_t)rinst * T.n_dense * nv, args.Ac + (size_t)rinst * T.nc * 6 * nv, nv);
        }
    }
    else if (wave == 1) {
        // ---- lanes = tasks with a frame (SE(3) blocks, then contacts): the law of ex_task.cpp:175-247, local frame ---------
        // constant tables first: they travel while the tree wave works
        const int ll = min(lane, max(T.nlaw - 1, 0));
        const int l_body = ip[T.i_law_body + ll], l_ref = ip[T.i_law_ref + ll], l_va = ip[T.i_law_va + ll];
        const int l_mask = ip[T.i_law_mask + ll], l_ct = ip[T.i_law_contact + ll], l_row = ip[T.i_law_row + ll];
        const double l_kp = dp[T.d_law_kp + ll], l_kd = dp[T.d_law_kd + ll];
        double l_place[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) l_place[r] = dp[T.d_law_place + 12 * ll + r];
        bsync(); // barrier 1: kin is complete
        {
            FrameKin f;
            if (lane < T.nlaw) {
                frame_kin(kin + kKinStride * l_body, l_place, f);
                double* Lw = law + kLawStride * lane;
#pragma unroll
                for (int r = 0; r < 9; ++r) Lw[r] = f.R[r];
                st3(Lw + 9, f.p);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(frames_ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane < T.nlaw) {
                const double* rf = ref + l_ref;
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
                if (l_va) { // wMl^-1 v_ref, wMl^-1 a_ref (:201,208)
                    vr = mtv(f.R, ld3(rf + 12)); wr = mtv(f.R, ld3(rf + 15));
                    ar = mtv(f.R, ld3(rf + 18)); alr = mtv(f.R, ld3(rf + 21));
                }
                const double kp = l_kp, kd = l_kd;
                const V3 rl_ = (kp * pe + kd * (vr - f.v) + ar) - f.a;
                const V3 ra = (kp * we + kd * (wr - f.w) + alr) - f.al;
                const double rhs[6] = {rl_.x, rl_.y, rl_.z, ra.x, ra.y, ra.z};
                const int mask = l_mask, ct = l_ct;
                double* out = (ct >= 0) ? bcs + 6 * ct : b1s + l_row;
                int o = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if ((mask >> i) & 1) out[o++] = rhs[i];
                // cop task (tasks.cpp:156-178; tsid TaskCopEquality::compute [UPSTREAM-RECALL]): a contact's lane writes its 3 x 12
                // block, per contact point (d n' - (n.d) I) R with d = oMf.act(p_i) - cop_ref, cop_ref = the world's origin
                // (tasks.cpp:171), n = e_z: entry (a, b) = d_a R(2, b) - d_z R(a, b)
                if (T.cop && ct >= 0 && args.Acop) {
                    const int kk = 12 * T.nc;
                    TI* oc = args.Acop + (size_t)rinst * 3 * kk + 12 * ct;
                    for (int i = 0; i < 4; ++i) {
                        const V3 d = (mv(f.R, ld3(dp + T.d_cop_pts + 12 * ct + 3 * i)) + f.p) + p0;
                        const double da[3] = {d.x, d.y, d.z};
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int b = 0; b < 3; ++b) oc[a * kk + 3 * i + b] = (TI)(da[a] * f.R[6 + b] - d.z * f.R[3 * a + b]);
                    }
                }
            }
            // the same wave, lanes = columns now: the Jacobian rows of those tasks.  They need the frames just written and the
            // bodies' placements, not the prefix sums, so two thirds of the record's bytes leave here, while wave 0 is still
            // scanning, instead of in one burst at the end of the kernel
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const bool colv = lane < nv;
                const ColumnAxis c = column_axis(kin, bj, lastj, kofj, jtypej);
                jacobian_rows<TI>(T, law, c, colv, lane, 0, 3, args.A + (size_t)rinst * T.n_dense * nv, args.Ac + (size_t)rinst * T.nc * 6 * nv, nv);
            }
            TSTAMP(4)
        }
    }
    else {
        {
            // ---- lanes = self-collision pairs (task-self-collision.cpp:84-203, 5PL repulsor :147-156) ---------------------
            // two passes on this wave: lanes = the distinct frames the tasks touch (placement, classical acceleration), then
            // lanes = pairs.  The constants of both are fetched before the barrier.
            double* scf = lds + T.o_scf;
            const int fl = min(lane, max(T.nscf - 1, 0));
            const int f_body = ip[T.i_scf_body + fl];
            double f_place[12];
#pragma unroll
            for (int r = 0; r < 12; ++r) f_place[r] = dp[T.d_scf_place + 12 * fl + r];
            int bt, ba, it, ia;
            double ppar[6];
            auto fetch = [&](int s) {
                const int sc_ = min(s, max(T.npair - 1, 0));
                bt = ip[T.i_pair_bt + sc_]; ba = ip[T.i_pair_ba + sc_];
                it = ip[T.i_pair_ft + sc_]; ia = ip[T.i_pair_fa + sc_];
#pragma unroll
                for (int r = 0; r < 6; ++r) ppar[r] = dp[T.d_pair_par + 6 * sc_ + r];
            };
            fetch(lane);
            bsync(); // barrier 1: kin is complete
            if (lane < T.nscf) {
                FrameKin f;
                frame_kin(kin + kKinStride * f_body, f_place, f);
                st3(scf + kScfStride * lane, f.p);
                st3(scf + kScfStride * lane + 3, f.a);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int s0 = 0; s0 < T.npair; s0 += kWave) {
                const int s = s0 + lane;
                if (s0 > 0) fetch(s);
                if (s < T.npair) {
                    struct { V3 p, a; } ft, fa;
                    ft.p = ld3(scf + kScfStride * it); ft.a = ld3(scf + kScfStride * it + 3);
                    fa.p = ld3(scf + kScfStride * ia); fa.a = ld3(scf + kScfStride * ia + 3);
                    const double aa = ppar[0], k5 = ppar[1], s_p = ppar[2], mm_ = ppar[3], kp = ppar[4], kd = ppar[5];
                    const V3 diff = ft.p - fa.p;
                    const V3 drift = ft.a - fa.a; // each in its own frame's axes, as the reference subtracts them (:92,131-133)
                    // J v for the WORLD Jacobians = difference of the bodies' spatial velocities at the true world origin
                    const double* Kt = kin + kKinStride * bt;
                    const double* Ka = kin + kKinStride * ba;
                    const V3 Jv = (ld3(Kt + 12) + cross(p0, ld3(Kt + 15))) - (ld3(Ka + 12) + cross(p0, ld3(Ka + 15)));
                    const double sn = dot(diff, diff), norm = sqrt(sn);
                    const double x = k5 * (norm - aa + s_p);
                    const double e_p = exp(-x), e1 = e_p + 1.0;
                    const double pw1 = exp((-mm_ - 1.0) * log(e1)); // (1 + e)^(-m-1), e1 >= 1; the two neighbouring powers follow from it
                    const double C = 1.0 - pw1 * e1;
                    // (1 / norm once; 1 / sn and 1 / (norm sqrt(norm)) follow from it -- the reference's expression has five divisions)
                    const double inorm = 1.0 / norm, isn = inorm * inorm, ie1 = 1.0 / e1;
                    const double kme = k5 * mm_ * e_p * pw1;
                    const double gscale = -inorm * kme;
                    const double hh = isn * k5 * (-mm_ - 1.0) * e_p * ie1 * kme + isn * k5 * kme + inorm * sqrt(inorm) * kme;
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // self-collision right-hand side: the sum over the pairs of a block
            if (lane < T.nblock && ip[T.i_blk_kind + lane] == T_SELFCOLLISION) {
                double B = 0.0;
                const int s0 = ip[T.i_blk_pair0 + lane], ns = ip[T.i_blk_npair + lane];
                for (int s = s0; s < s0 + ns; ++s) B += pair[kPairStride * s + 3];
                b1s[ip[T.i_blk_row + lane]] = B;
            }
            TSTAMP(5)
        }
    }
    bsync(); // barrier 2: scan table, task frames, pairs, right-hand sides of the framed tasks
    TSTAMP(6)

    // ---- phase 3: lanes = velocity coordinates, on every wave -----------------------------------------------------------
    const double* tot = scan + kScanStride * nb;
    const double mass = tot[0], imass = 1.0 / mass;
    const V3 com = imass * ld3(tot + 1);
    const V3 htl = ld3(tots), hta = ld3(tots + 3);
    if (gmom && tid == 0) { // Ag v: total momentum, the angular part taken about the CoM (controller.cpp:245 reads its last three)
        const V3 La = hta - cross(com, htl);
        TI* mo = gmom;
        mo[0] = (TI)htl.x; mo[1] = (TI)htl.y; mo[2] = (TI)htl.z;
        mo[3] = (TI)La.x; mo[4] = (TI)La.y; mo[5] = (TI)La.z;
    }
    const bool colv = lane < nv;
    // S_j and F_j = Y_subtree(j) S_j of this lane's column: formed once, by wave 0 at the end of phase 2
    const double* SF = lds + T.o_sf + kSFStride * cj;
    const V3 Sv = ld3(SF), Sw = ld3(SF + 3), Fv = ld3(SF + 6), Fw = ld3(SF + 9);
    TSTAMP(7)
    // M, row by row into the packed lower triangle: M(i, j) = S_j . F_i for j an ancestor dof of i (crba)
    {
        TI* Mo = args.M + (size_t)rinst * (nv * (nv + 1) / 2);
        // four rows in flight: the reads of F_i (LDS) and the stores of one row do not wait for the row before
        for (int i0 = wave; i0 < nv; i0 += 4 * kWaves) {
            V3 fv[4], fw[4];
            int b_i[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * kWaves, nv - 1);
                // F_i from the table (one address for the whole wave: a broadcast read on the LDS port; six readlane pairs would
                // sit on the vector port, which is the one this kernel saturates)
                const double* SFi = lds + T.o_sf + kSFStride * i + 6;
                fv[u] = ld3(SFi);
                fw[u] = ld3(SFi + 3);
                b_i[u] = rl(bj, i);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kWaves;
                const double val = dot(Sv, fv[u]) + dot(Sw, fw[u]);
                if (i < nv && lane <= i) Mo[i * (i + 1) / 2 + lane] = (TI)((bj <= b_i[u] && b_i[u] <= lastj) ? val : 0.0);
            }
        }
    }
    TSTAMP(8)
    TI* Ao = args.A + (size_t)rinst * T.n_dense * nv; // (the Jacobian rows of the framed tasks left in phase 2, from wave 1)
    TSTAMP(9)
    // CoM, momentum and self-collision rows and the CoM / momentum right-hand sides, one block per wave in turn
    {
        const int tt = min(lane, max(T.nblock - 1, 0));
        const int k_kind = ip[T.i_blk_kind + tt], k_mask = ip[T.i_blk_mask + tt], k_row = ip[T.i_blk_row + tt];
        const int k_p0 = ip[T.i_blk_pair0 + tt], k_np = ip[T.i_blk_npair + tt], k_ref = ip[T.i_blk_ref + tt];
        const double k_kp = dp[T.d_blk_kp + tt], k_kd = dp[T.d_blk_kd + tt];
        const V3 lt = Sv + cross(p0, Sw); // WORLD Jacobian column (linear part): about the true world origin
        for (int t = wave; t < T.nblock; t += kWaves) {
            const int kind = rl(k_kind, t);
            if (kind == T_SE3) continue;
            const int mask = rl(k_mask, t), row = rl(k_row, t);
            TI* out = Ao + (size_t)row * nv;
            if (kind == T_COM) {
                // tsid TaskComEquality: rows of Jcom; a_des = -Kp (com - ref) - Kd (vcom - vref) + aref, minus the drift
                const double e[3] = {imass * Fv.x, imass * Fv.y, imass * Fv.z};
                const double* rf = ref + rl(k_ref, t);
                const double kp = bcast_lane(k_kp, t), kd = bcast_lane(k_kd, t);
                const V3 vcom = imass * htl, acom = imass * ld3(tot + 10);
                const V3 er = com - (ld3(rf) - p0);
                const V3 r = (-kp * er - kd * (vcom - ld3(rf + 3)) + ld3(rf + 6)) - acom;
                const double rr[3] = {r.x, r.y, r.z};
                int o = 0;
                for (int i = 0; i < 3; ++i)
                    if ((mask >> i) & 1) {
                        if (colv) out[(size_t)o * nv + lane] = (TI)e[i];
                        if (lane == 0) b1s[row + o] = rr[i];
                        ++o;
                    }
            }
            else if (kind == T_MOMENTUM) {
                // task-momentum-equality.cpp:151-165: rows of Ag; dL_des = -Kp (L - ref') + ref'', minus the drift
                const V3 an = Fw - cross(com, Fv);
                const double e[6] = {Fv.x, Fv.y, Fv.z, an.x, an.y, an.z};
                const double* rf = ref + rl(k_ref, t);
                const double kp = bcast_lane(k_kp, t);
                const V3 Lang = hta - cross(com, htl);                      // angular momentum about the com
                const V3 dLang = ld3(tot + 13) - cross(com, ld3(tot + 10)); // its rate at ddq = 0
                const double L[6] = {htl.x, htl.y, htl.z, Lang.x, Lang.y, Lang.z};
                const double dL[6] = {tot[10], tot[11], tot[12], dLang.x, dLang.y, dLang.z};
                int o = 0;
                for (int i = 0; i < 6; ++i)
                    if ((mask >> i) & 1) {
                        if (colv) out[(size_t)o * nv + lane] = (TI)e[i];
                        if (lane == 0) b1s[row + o] = (-kp * (L[i] - rf[i]) + rf[6 + i]) - dL[i];
                        ++o;
                    }
            }
            else {
                // A = sum grad_C' (J_tracked - J_avoided): the WORLD columns coincide wherever both frames hang on dof j
                V3 acc = {0, 0, 0};
                const int s0 = rl(k_p0, t), ns = rl(k_np, t);
#pragma unroll 4
                for (int s = s0; s < s0 + ns; ++s) {
                    const double* Pw = pair + kPairStride * s;
                    const int bt = (int)Pw[4], ba = (int)Pw[5];
                    const double sg = (double)((bj <= bt && bt <= lastj) ? 1 : 0) - (double)((bj <= ba && ba <= lastj) ? 1 : 0);
                    acc = acc + sg * ld3(Pw);
                }
                if (colv) out[lane] = (TI)dot(acc, lt);
            }
        }
    }
    TSTAMP(10)
    bsync();
    for (int i = tid; i < T.r1; i += kTermsThreads) args.b1[(size_t)rinst * T.r1 + i] = (TI)b1s[i];
    for (int i = tid; i < 6 * T.nc; i += kTermsThreads) args.bc[(size_t)rinst * 6 * T.nc + i] = (TI)bcs[i];
#ifdef WBCQP_STAMPS
    TSTAMP(11)
    // every wave accumulates its own phases; wave w writes the slots it owns (0: 0-3,6-11; 1: 4; 2: 5)
    if (lane == 0 && args.dbg) {
        long long* D = args.dbg + (size_t)inst * 24;
        if (wave == 0) { for (int i = 0; i < 12; ++i) if (i != 4 && i != 5) D[i] = tacc_[i]; }
        if (wave == 1) D[4] = tacc_[4];
        if (wave == 2) D[5] = tacc_[5];
    }
#endif
}

template <typename TI>
__global__ __launch_bounds__(kTermsThreads, 4) void terms_kernel(const TermsArgs<TI> args)
{
    extern __shared__ double lds[];
    if ((int)blockIdx.x >= args.batch) return;
    const int inst = (int)blockIdx.x;
    terms_one<TI>(args, args.q + (size_t)inst * args.T.nq, args.v + (size_t)inst * args.T.nv, args.ref + (size_t)inst * args.T.nref,
                  args.momentum ? args.momentum + (size_t)inst * 6 : nullptr, inst, inst, lds, (int)threadIdx.x);
}

#endif // __HIPCC__
} // namespace wbcqp
