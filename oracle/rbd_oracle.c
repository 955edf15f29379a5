/*
 * rbd_oracle.c -- CPU restatement of the step before the hot path: rigid-body terms and task laws.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see rbd_oracle.h for what that means here).
 *
 * Which quantities: RobotModel::update (/root/reference/src/utils/robot_model.cpp:83-113) lists the pinocchio calls the
 * reference makes around a tick -- forwardKinematics, centerOfMass, computeJointJacobians, jacobianCenterOfMass, crba
 * (+ symmetrisation :101), nonLinearEffects, updateFramePlacements; tsid's computeProblemData (call site
 * /root/reference/src/controllers/controller.cpp:244) makes the same set plus ccrba through RobotWrapper::computeAllTerms
 * [UPSTREAM-RECALL].  The recursions below follow pinocchio 2.x's local-frame formulations on purpose: the device code
 * uses a world-frame formulation, so the two are independent derivations of the same numbers.
 */
#include "rbd_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define MAXB 64  /* bodies */
#define MAXV 72  /* velocity coordinates */

typedef struct { double R[9], p[3]; } se3;
typedef struct { double v[3], w[3]; } motion; /* also used for forces: (linear, angular) */
typedef struct { double m, c[3], I[9]; } inertia;

/* ------------------------------------------------------------------ small algebra ---- */
static void cross(const double* a, const double* b, double* o)
{
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void mv(const double* R, const double* x, double* o)
{
    const double a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2], b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2],
                 c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
    o[0] = a; o[1] = b; o[2] = c;
}
static void mtv(const double* R, const double* x, double* o)
{
    const double a = R[0] * x[0] + R[3] * x[1] + R[6] * x[2], b = R[1] * x[0] + R[4] * x[1] + R[7] * x[2],
                 c = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
    o[0] = a; o[1] = b; o[2] = c;
}
static void mm(const double* A, const double* B, double* O)
{
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(O, t, sizeof t);
}
static void se3_identity(se3* a)
{
    memset(a, 0, sizeof *a);
    a->R[0] = a->R[4] = a->R[8] = 1.0;
}
/* c = a * b */
static void se3_mul(const se3* a, const se3* b, se3* c)
{
    se3 t;
    mm(a->R, b->R, t.R);
    mv(a->R, b->p, t.p);
    for (int i = 0; i < 3; ++i) t.p[i] += a->p[i];
    *c = t;
}
/* c = a^-1 * b */
static void se3_inv_mul(const se3* a, const se3* b, se3* c)
{
    se3 t;
    double Rt[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt[3 * i + j] = a->R[3 * j + i];
    mm(Rt, b->R, t.R);
    double d[3] = {b->p[0] - a->p[0], b->p[1] - a->p[1], b->p[2] - a->p[2]};
    mv(Rt, d, t.p);
    *c = t;
}
/* SE3::act on a motion: (R v + p x R w, R w) */
static void act_motion(const se3* M, const motion* a, motion* o)
{
    motion t;
    double pxw[3];
    mv(M->R, a->w, t.w);
    mv(M->R, a->v, t.v);
    cross(M->p, t.w, pxw);
    for (int i = 0; i < 3; ++i) t.v[i] += pxw[i];
    *o = t;
}
/* SE3::actInv on a motion: (R'(v - p x w), R' w) */
static void actinv_motion(const se3* M, const motion* a, motion* o)
{
    motion t;
    double pxw[3], d[3];
    cross(M->p, a->w, pxw);
    for (int i = 0; i < 3; ++i) d[i] = a->v[i] - pxw[i];
    mtv(M->R, d, t.v);
    mtv(M->R, a->w, t.w);
    *o = t;
}
/* SE3::act on a force: (R f, R n + p x R f) */
static void act_force(const se3* M, const motion* a, motion* o)
{
    motion t;
    double pxf[3];
    mv(M->R, a->v, t.v);
    mv(M->R, a->w, t.w);
    cross(M->p, t.v, pxf);
    for (int i = 0; i < 3; ++i) t.w[i] += pxf[i];
    *o = t;
}
/* motion x motion */
static void mxm(const motion* a, const motion* b, motion* o)
{
    motion t;
    double u[3], s[3];
    cross(a->w, b->v, u);
    cross(a->v, b->w, s);
    for (int i = 0; i < 3; ++i) t.v[i] = u[i] + s[i];
    cross(a->w, b->w, t.w);
    *o = t;
}
/* motion x* force */
static void mxf(const motion* a, const motion* f, motion* o)
{
    motion t;
    double u[3], s[3];
    cross(a->w, f->v, t.v);
    cross(a->w, f->w, u);
    cross(a->v, f->v, s);
    for (int i = 0; i < 3; ++i) t.w[i] = u[i] + s[i];
    *o = t;
}
/* Y * motion -> force (pinocchio InertiaTpl::__mult__) */
static void imul(const inertia* Y, const motion* a, motion* o)
{
    motion t;
    double cxw[3], u[3];
    cross(Y->c, a->w, cxw);
    for (int i = 0; i < 3; ++i) t.v[i] = Y->m * (a->v[i] - cxw[i]);
    mv(Y->I, a->w, t.w);
    cross(Y->c, t.v, u);
    for (int i = 0; i < 3; ++i) t.w[i] += u[i];
    *o = t;
}
/* M.act(Y): same body, expressed in the parent frame */
static void iact(const se3* M, const inertia* Y, inertia* o)
{
    inertia t;
    double RI[9], Rt[9];
    t.m = Y->m;
    mv(M->R, Y->c, t.c);
    for (int i = 0; i < 3; ++i) t.c[i] += M->p[i];
    mm(M->R, Y->I, RI);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt[3 * i + j] = M->R[3 * j + i];
    mm(RI, Rt, t.I);
    *o = t;
}
/* Ya + Yb (pinocchio InertiaTpl::__plus__): I = Ia + Ib - (ma mb / m) skew(ca - cb)^2 */
static void iadd(const inertia* a, const inertia* b, inertia* o)
{
    inertia t;
    t.m = a->m + b->m;
    const double inv = 1.0 / t.m;
    double ab[3];
    for (int i = 0; i < 3; ++i) {
        t.c[i] = (a->m * a->c[i] + b->m * b->c[i]) * inv;
        ab[i] = a->c[i] - b->c[i];
    }
    const double k = a->m * b->m * inv, n2 = dot3(ab, ab);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double sk2 = ab[i] * ab[j] - (i == j ? n2 : 0.0); /* skew(ab)^2 */
            t.I[3 * i + j] = a->I[3 * i + j] + b->I[3 * i + j] - k * sk2;
        }
    *o = t;
}

/* ------------------------------------------------------------------ model access ---- */
static int idx_q(const wbco_model* m, int i) { return m->floating_base ? (i == 0 ? 0 : 6 + i) : i; }
static int idx_v(const wbco_model* m, int i) { return m->floating_base ? (i == 0 ? 0 : 5 + i) : i; }
static int nv_of(const wbco_model* m, int i) { return m->jtype[i] == WBCO_J_FREEFLYER ? 6 : 1; }
static void load_se3(const double* a, se3* o)
{
    memcpy(o->R, a, 9 * sizeof(double));
    memcpy(o->p, a + 9, 3 * sizeof(double));
}
static void load_inertia(const double* a, inertia* Y)
{
    Y->m = a[0];
    Y->c[0] = a[1]; Y->c[1] = a[2]; Y->c[2] = a[3];
    Y->I[0] = a[4]; Y->I[1] = a[5]; Y->I[2] = a[6];
    Y->I[3] = a[5]; Y->I[4] = a[7]; Y->I[5] = a[8];
    Y->I[6] = a[6]; Y->I[7] = a[8]; Y->I[8] = a[9];
}
/* joint transform (pinocchio JointModel*::calc): free-flyer = (R(quat), p); revolute about a coordinate axis;
 * prismatic along one */
static void joint_transform(int jt, const double* q, se3* M)
{
    se3_identity(M);
    if (jt == WBCO_J_FREEFLYER) {
        const double x = q[3], y = q[4], z = q[5], w = q[6];
        /* Eigen::Quaternion::toRotationMatrix */
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
                     tzz = tz * z;
        M->R[0] = 1 - (tyy + tzz); M->R[1] = txy - twz; M->R[2] = txz + twy;
        M->R[3] = txy + twz; M->R[4] = 1 - (txx + tzz); M->R[5] = tyz - twx;
        M->R[6] = txz - twy; M->R[7] = tyz + twx; M->R[8] = 1 - (txx + tyy);
        M->p[0] = q[0]; M->p[1] = q[1]; M->p[2] = q[2];
    }
    else if (jt <= WBCO_J_RZ) {
        const double c = cos(q[0]), s = sin(q[0]);
        if (jt == WBCO_J_RX) { M->R[4] = c; M->R[5] = -s; M->R[7] = s; M->R[8] = c; }
        if (jt == WBCO_J_RY) { M->R[0] = c; M->R[2] = s; M->R[6] = -s; M->R[8] = c; }
        if (jt == WBCO_J_RZ) { M->R[0] = c; M->R[1] = -s; M->R[3] = s; M->R[4] = c; }
    }
    else M->p[jt - WBCO_J_PX] = q[0];
}
/* column k (0 <= k < nv_of) of the joint's motion subspace, in the joint frame */
static void joint_S(int jt, int k, motion* S)
{
    memset(S, 0, sizeof *S);
    if (jt == WBCO_J_FREEFLYER) { if (k < 3) S->v[k] = 1.0; else S->w[k - 3] = 1.0; }
    else if (jt <= WBCO_J_RZ) S->w[jt - WBCO_J_RX] = 1.0;
    else S->v[jt - WBCO_J_PX] = 1.0;
}
static void joint_velocity(int jt, const double* vq, motion* o)
{
    memset(o, 0, sizeof *o);
    for (int k = 0; k < (jt == WBCO_J_FREEFLYER ? 6 : 1); ++k) {
        motion S;
        joint_S(jt, k, &S);
        for (int i = 0; i < 3; ++i) { o->v[i] += S.v[i] * vq[k]; o->w[i] += S.w[i] * vq[k]; }
    }
}

typedef struct {
    se3 liMi[MAXB], oMi[MAXB];
    motion v[MAXB], a[MAXB];
} kin;

/* pinocchio forwardKinematics(q, v, a) (second order): v_i = vJ + liMi^-1 v_parent,
 * a_i = S ddq + v_i x vJ + liMi^-1 a_parent; the world's acceleration is a0 (0, or -gravity for a_gf of rnea). */
static void forward(const wbco_model* m, const double* q, const double* v, const double* ddq, const double* a0_lin, kin* K)
{
    for (int i = 0; i < m->nbody; ++i) {
        se3 P, Mj;
        load_se3(m->placement + 12 * i, &P);
        joint_transform(m->jtype[i], q + idx_q(m, i), &Mj);
        se3_mul(&P, &Mj, &K->liMi[i]);
        motion vJ, aJ, par_v, par_a, c;
        joint_velocity(m->jtype[i], v + idx_v(m, i), &vJ);
        memset(&aJ, 0, sizeof aJ);
        if (ddq) joint_velocity(m->jtype[i], ddq + idx_v(m, i), &aJ);
        const int p = m->parent[i];
        if (p >= 0) {
            se3_mul(&K->oMi[p], &K->liMi[i], &K->oMi[i]);
            actinv_motion(&K->liMi[i], &K->v[p], &par_v);
            actinv_motion(&K->liMi[i], &K->a[p], &par_a);
        }
        else {
            K->oMi[i] = K->liMi[i];
            memset(&par_v, 0, sizeof par_v);
            motion w0;
            memset(&w0, 0, sizeof w0);
            if (a0_lin) memcpy(w0.v, a0_lin, sizeof w0.v);
            actinv_motion(&K->liMi[i], &w0, &par_a);
        }
        for (int k = 0; k < 3; ++k) { K->v[i].v[k] = vJ.v[k] + par_v.v[k]; K->v[i].w[k] = vJ.w[k] + par_v.w[k]; }
        mxm(&K->v[i], &vJ, &c);
        for (int k = 0; k < 3; ++k) { K->a[i].v[k] = aJ.v[k] + c.v[k] + par_a.v[k]; K->a[i].w[k] = aJ.w[k] + c.w[k] + par_a.w[k]; }
    }
}

/* pinocchio rnea: f_i = Y a_gf + v x* (Y v); backward tau_i = S' f_i, f_parent += liMi f_i */
static void rnea_core(const wbco_model* m, const double* q, const double* v, const double* ddq, double* tau)
{
    static __thread kin K;
    motion f[MAXB];
    const double a0[3] = {-m->gravity[0], -m->gravity[1], -m->gravity[2]};
    forward(m, q, v, ddq, a0, &K);
    for (int i = 0; i < m->nbody; ++i) {
        inertia Y;
        motion h, ya, vxh;
        load_inertia(m->inertia + 10 * i, &Y);
        imul(&Y, &K.v[i], &h);
        imul(&Y, &K.a[i], &ya);
        mxf(&K.v[i], &h, &vxh);
        for (int k = 0; k < 3; ++k) { f[i].v[k] = ya.v[k] + vxh.v[k]; f[i].w[k] = ya.w[k] + vxh.w[k]; }
    }
    for (int i = m->nbody - 1; i >= 0; --i) {
        for (int k = 0; k < nv_of(m, i); ++k) {
            motion S;
            joint_S(m->jtype[i], k, &S);
            tau[idx_v(m, i) + k] = dot3(S.v, f[i].v) + dot3(S.w, f[i].w);
        }
        const int p = m->parent[i];
        if (p >= 0) {
            motion t;
            act_force(&K.liMi[i], &f[i], &t);
            for (int k = 0; k < 3; ++k) { f[p].v[k] += t.v[k]; f[p].w[k] += t.w[k]; }
        }
    }
}

void wbco_rnea(const wbco_model* m, const double* q, const double* v, const double* a, double* tau) { rnea_core(m, q, v, a, tau); }

/* pinocchio crba (local frame): Ycrb_i = Y_i; backward: F_i = Ycrb_i S_i, M(i, subtree i) = S_i' F, Ycrb_parent += liMi Ycrb_i,
 * F columns of the subtree move to the parent frame.  robot_model.cpp:101 then mirrors the upper triangle. */
static void crba(const wbco_model* m, const kin* K, double* M)
{
    const int nv = m->nv, nb = m->nbody;
    inertia* Y = (inertia*)malloc(sizeof(inertia) * nb);
    motion* F = (motion*)calloc((size_t)nv, sizeof(motion)); /* column j, expressed in the frame of the body being visited */
    int* last = (int*)malloc(sizeof(int) * nb);              /* last velocity index of the subtree */
    for (int i = 0; i < nb; ++i) { load_inertia(m->inertia + 10 * i, &Y[i]); last[i] = idx_v(m, i) + nv_of(m, i) - 1; }
    for (int i = nb - 1; i > 0; --i)
        if (last[i] > last[m->parent[i]]) last[m->parent[i]] = last[i];
    memset(M, 0, sizeof(double) * nv * nv);
    for (int i = nb - 1; i >= 0; --i) {
        const int iv = idx_v(m, i), ni = nv_of(m, i);
        for (int k = 0; k < ni; ++k) {
            motion S;
            joint_S(m->jtype[i], k, &S);
            imul(&Y[i], &S, &F[iv + k]);
        }
        for (int k = 0; k < ni; ++k) {
            motion S;
            joint_S(m->jtype[i], k, &S);
            for (int j = iv; j <= last[i]; ++j) M[(iv + k) * nv + j] = dot3(S.v, F[j].v) + dot3(S.w, F[j].w);
        }
        const int p = m->parent[i];
        if (p >= 0) {
            inertia t;
            iact(&K->liMi[i], &Y[i], &t);
            iadd(&Y[p], &t, &Y[p]);
            for (int j = iv; j <= last[i]; ++j) act_force(&K->liMi[i], &F[j], &F[j]);
        }
    }
    for (int i = 0; i < nv; ++i)
        for (int j = 0; j < i; ++j) M[i * nv + j] = M[j * nv + i];
    free(Y); free(F); free(last);
}

static int supports(const wbco_model* m, int body, int of_body)
{
    /* is `body` of_body itself or one of its ancestors */
    for (int b = of_body; b >= 0; b = m->parent[b])
        if (b == body) return 1;
    return 0;
}

double wbco_energy(const wbco_model* m, const double* q, const double* v)
{
    static __thread kin K;
    forward(m, q, v, NULL, NULL, &K);
    double e = 0.0;
    for (int i = 0; i < m->nbody; ++i) {
        inertia Y;
        motion h;
        double cw[3];
        load_inertia(m->inertia + 10 * i, &Y);
        imul(&Y, &K.v[i], &h);
        e += 0.5 * (dot3(K.v[i].v, h.v) + dot3(K.v[i].w, h.w));
        mv(K.oMi[i].R, Y.c, cw);
        for (int k = 0; k < 3; ++k) e -= Y.m * m->gravity[k] * (cw[k] + K.oMi[i].p[k]);
    }
    return e;
}

void wbco_rbd_terms(const wbco_model* m, const double* q, const double* v, wbco_terms* out)
{
    static __thread kin K;
    const int nv = m->nv, nb = m->nbody;
    /* data.a as tsid leaves it: centerOfMass(q, v, 0) ran forwardKinematics with zero joint acceleration and no gravity */
    forward(m, q, v, NULL, NULL, &K);
    if (out->M) crba(m, &K, out->M);
    if (out->nle) {
        double* zero = (double*)calloc((size_t)nv, sizeof(double));
        rnea_core(m, q, v, zero, out->nle);
        free(zero);
    }
    /* world-frame joint Jacobian (pinocchio data.J): column = oMi.act(S) */
    motion* Jw = (motion*)calloc((size_t)nv, sizeof(motion));
    int* body_of = (int*)malloc(sizeof(int) * nv);
    for (int i = 0; i < nb; ++i)
        for (int k = 0; k < nv_of(m, i); ++k) {
            motion S;
            joint_S(m->jtype[i], k, &S);
            act_motion(&K.oMi[i], &S, &Jw[idx_v(m, i) + k]);
            body_of[idx_v(m, i) + k] = i;
        }
    /* centre of mass, its velocity and bias acceleration (pinocchio centerOfMass): point quantities of each body's com */
    double mass = 0.0, com[3] = {0, 0, 0}, vcom[3] = {0, 0, 0}, acom[3] = {0, 0, 0};
    double cw[MAXB][3], vc[MAXB][3], ac[MAXB][3];
    inertia Y[MAXB];
    for (int i = 0; i < nb; ++i) {
        load_inertia(m->inertia + 10 * i, &Y[i]);
        double t[3], u[3], vl[3], al[3];
        mv(K.oMi[i].R, Y[i].c, t);
        for (int k = 0; k < 3; ++k) cw[i][k] = t[k] + K.oMi[i].p[k];
        cross(K.v[i].w, Y[i].c, u);
        for (int k = 0; k < 3; ++k) vl[k] = K.v[i].v[k] + u[k];
        mv(K.oMi[i].R, vl, vc[i]);
        cross(K.a[i].w, Y[i].c, u);
        cross(K.v[i].w, vl, t);
        for (int k = 0; k < 3; ++k) al[k] = K.a[i].v[k] + u[k] + t[k];
        mv(K.oMi[i].R, al, ac[i]);
        mass += Y[i].m;
        for (int k = 0; k < 3; ++k) { com[k] += Y[i].m * cw[i][k]; vcom[k] += Y[i].m * vc[i][k]; acom[k] += Y[i].m * ac[i][k]; }
    }
    for (int k = 0; k < 3; ++k) { com[k] /= mass; vcom[k] /= mass; acom[k] /= mass; }
    if (out->com) memcpy(out->com, com, sizeof com);
    if (out->vcom) memcpy(out->vcom, vcom, sizeof vcom);
    if (out->acom) memcpy(out->acom, acom, sizeof acom);
    /* Jcom and the centroidal momentum matrix Ag (pinocchio jacobianCenterOfMass / ccrba): column j = momentum of the
     * bodies joint j carries when it alone moves at unit speed, linear part over the total mass / about the com */
    if (out->Jcom || out->Ag) {
        for (int j = 0; j < nv; ++j) {
            double lin[3] = {0, 0, 0}, ang[3] = {0, 0, 0};
            for (int i = 0; i < nb; ++i) {
                if (!supports(m, body_of[j], i)) continue;
                double u[3], pv[3], Iw[9], Rt[9], RI[9], Lw[3], r[3], rxp[3];
                cross(Jw[j].w, cw[i], u);
                for (int k = 0; k < 3; ++k) pv[k] = Y[i].m * (Jw[j].v[k] + u[k]);
                mm(K.oMi[i].R, Y[i].I, RI);
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) Rt[3 * a + b] = K.oMi[i].R[3 * b + a];
                mm(RI, Rt, Iw);
                mv(Iw, Jw[j].w, Lw);
                for (int k = 0; k < 3; ++k) r[k] = cw[i][k] - com[k];
                cross(r, pv, rxp);
                for (int k = 0; k < 3; ++k) { lin[k] += pv[k]; ang[k] += Lw[k] + rxp[k]; }
            }
            for (int k = 0; k < 3; ++k) {
                if (out->Jcom) out->Jcom[k * nv + j] = lin[k] / mass;
                if (out->Ag) { out->Ag[k * nv + j] = lin[k]; out->Ag[(3 + k) * nv + j] = ang[k]; }
            }
        }
    }
    /* d/dt of the centroidal momentum with ddq = 0 (pinocchio computeCentroidalMomentumTimeVariation on data.a as above;
     * task-momentum-equality.cpp:158-159 takes its linear and angular halves): Newton-Euler of every body about the com */
    if (out->dAgv) {
        double lin[3] = {0, 0, 0}, ang[3] = {0, 0, 0};
        for (int i = 0; i < nb; ++i) {
            double Ia[3], Iv[3], wxIv[3], nl[3], nw[3], r[3], f[3], rxf[3];
            mv(Y[i].I, K.a[i].w, Ia);
            mv(Y[i].I, K.v[i].w, Iv);
            cross(K.v[i].w, Iv, wxIv);
            for (int k = 0; k < 3; ++k) nl[k] = Ia[k] + wxIv[k];
            mv(K.oMi[i].R, nl, nw);
            for (int k = 0; k < 3; ++k) { f[k] = Y[i].m * ac[i][k]; r[k] = cw[i][k] - com[k]; }
            cross(r, f, rxf);
            for (int k = 0; k < 3; ++k) { lin[k] += f[k]; ang[k] += nw[k] + rxf[k]; }
        }
        memcpy(out->dAgv, lin, sizeof lin);
        memcpy(out->dAgv + 3, ang, sizeof ang);
    }
    /* frames: placement (updateFramePlacements), velocity, classical acceleration (tsid RobotWrapper::frameVelocity /
     * frameClassicAcceleration: a.linear += w x v), local and WORLD Jacobians (pinocchio getFrameJacobian) */
    for (int f = 0; f < m->nframe; ++f) {
        const int b = m->frame_body[f];
        se3 P, oMf;
        motion vf, af;
        load_se3(m->frame_placement + 12 * f, &P);
        se3_mul(&K.oMi[b], &P, &oMf);
        actinv_motion(&P, &K.v[b], &vf);
        actinv_motion(&P, &K.a[b], &af);
        double wxv[3];
        cross(vf.w, vf.v, wxv);
        for (int k = 0; k < 3; ++k) af.v[k] += wxv[k];
        if (out->oMf) { memcpy(out->oMf + 12 * f, oMf.R, 9 * sizeof(double)); memcpy(out->oMf + 12 * f + 9, oMf.p, 3 * sizeof(double)); }
        if (out->vf) { memcpy(out->vf + 6 * f, vf.v, 3 * sizeof(double)); memcpy(out->vf + 6 * f + 3, vf.w, 3 * sizeof(double)); }
        if (out->af) { memcpy(out->af + 6 * f, af.v, 3 * sizeof(double)); memcpy(out->af + 6 * f + 3, af.w, 3 * sizeof(double)); }
        for (int j = 0; j < nv; ++j) {
            motion loc, wor;
            memset(&loc, 0, sizeof loc);
            memset(&wor, 0, sizeof wor);
            if (supports(m, body_of[j], b)) { wor = Jw[j]; actinv_motion(&oMf, &Jw[j], &loc); }
            for (int k = 0; k < 3; ++k) {
                if (out->Jl) { out->Jl[((size_t)f * 6 + k) * nv + j] = loc.v[k]; out->Jl[((size_t)f * 6 + 3 + k) * nv + j] = loc.w[k]; }
                if (out->Jw) { out->Jw[((size_t)f * 6 + k) * nv + j] = wor.v[k]; out->Jw[((size_t)f * 6 + 3 + k) * nv + j] = wor.w[k]; }
            }
        }
    }
    free(Jw); free(body_of);
}

/* pinocchio log3 [UPSTREAM-RECALL]: angle from the trace; the antisymmetric part away from pi, the diagonal near it */
void wbco_log3(const double* R, double* w)
{
    const double tr = R[0] + R[4] + R[8];
    double theta;
    if (tr > 3.0) theta = 0.0;
    else if (tr < -1.0) theta = M_PI;
    else theta = acos((tr - 1.0) / 2.0);
    if (theta >= M_PI - 1e-2) {
        const double cphi = cos(theta - M_PI), beta = theta * theta / (1.0 + cphi);
        const double t0 = (R[0] + cphi) * beta, t1 = (R[4] + cphi) * beta, t2 = (R[8] + cphi) * beta;
        w[0] = (R[7] > R[5] ? 1.0 : -1.0) * (t0 > 0.0 ? sqrt(t0) : 0.0);
        w[1] = (R[2] > R[6] ? 1.0 : -1.0) * (t1 > 0.0 ? sqrt(t1) : 0.0);
        w[2] = (R[3] > R[1] ? 1.0 : -1.0) * (t2 > 0.0 ? sqrt(t2) : 0.0);
    }
    else {
        const double t = ((theta > 1.220703125e-4 /* eps^(1/4) */) ? theta / sin(theta) : 1.0) / 2.0;
        w[0] = t * (R[7] - R[5]);
        w[1] = t * (R[2] - R[6]);
        w[2] = t * (R[3] - R[1]);
    }
}

/* reference placement: translation (3) then rotation COLUMN-major (9) -> se3 (row-major R) */
static void ref_se3(const double* r, se3* M)
{
    memcpy(M->p, r, 3 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M->R[3 * i + j] = r[3 + 3 * j + i];
}

/* The SE(3) task law in the local frame: /root/reference/example_project/src/tsid/ex_task.cpp:175-247 (the in-tree copy of
 * tsid TaskSE3Equality::compute, m_local_frame = true which is tsid's default):
 *   errorInSE3: M_err = oMf^-1 M_ref, p_error = (translation, log3(rotation))  [UPSTREAM-RECALL tsid math/utils.cpp]
 *   v_error = wMl^-1 v_ref - v_frame;  a_des = Kp p_error + Kd v_error + wMl^-1 a_ref  (:201-208)
 *   row i (mask) = J_local row i, rhs = (a_des - drift)_i  (:233-237) */
static void se3_law(const double* oMf12, const double* vf, const double* af, const double* ref, int has_vel_acc, double kp, double kd,
                    double* rhs6)
{
    se3 oMf, Mref, Merr;
    memcpy(oMf.R, oMf12, 9 * sizeof(double));
    memcpy(oMf.p, oMf12 + 9, 3 * sizeof(double));
    ref_se3(ref, &Mref);
    se3_inv_mul(&oMf, &Mref, &Merr);
    double perr[6], vref[6] = {0, 0, 0, 0, 0, 0}, aref[6] = {0, 0, 0, 0, 0, 0};
    memcpy(perr, Merr.p, 3 * sizeof(double));
    wbco_log3(Merr.R, perr + 3);
    if (has_vel_acc) {
        mtv(oMf.R, ref + 12, vref);
        mtv(oMf.R, ref + 15, vref + 3);
        mtv(oMf.R, ref + 18, aref);
        mtv(oMf.R, ref + 21, aref + 3);
    }
    for (int i = 0; i < 6; ++i) rhs6[i] = (kp * perr[i] + kd * (vref[i] - vf[i]) + aref[i]) - af[i];
}

static double sqr(double x) { return x * x; }

void wbco_task_rows(const wbco_model* m, const wbco_taskmap* map, const double* q, const double* v, const double* ref,
                    double* Mp, double* h, double* A, double* b1, double* Ac, double* bc, double* blb, double* bub, double* Acop)
{
    const int nv = m->nv, nf = m->nframe;
    wbco_terms T;
    T.M = (double*)malloc(sizeof(double) * nv * nv);
    T.nle = h;
    double com[3], vcom[3], acom[3], dAgv[6];
    T.com = com; T.vcom = vcom; T.acom = acom; T.dAgv = dAgv;
    T.Jcom = (double*)malloc(sizeof(double) * 3 * nv);
    T.Ag = (double*)malloc(sizeof(double) * 6 * nv);
    T.oMf = (double*)malloc(sizeof(double) * 12 * (nf + 1));
    T.vf = (double*)malloc(sizeof(double) * 6 * (nf + 1));
    T.af = (double*)malloc(sizeof(double) * 6 * (nf + 1));
    T.Jl = (double*)malloc(sizeof(double) * 6 * nv * (nf + 1));
    T.Jw = (double*)malloc(sizeof(double) * 6 * nv * (nf + 1));
    wbco_rbd_terms(m, q, v, &T);
    for (int i = 0; i < nv; ++i)
        for (int j = 0; j <= i; ++j) Mp[i * (i + 1) / 2 + j] = T.M[i * nv + j];
    int row = 0;
    for (int t = 0; t < map->nblock; ++t) {
        const wbco_taskblock* B = &map->block[t];
        if (B->kind == WBCO_T_SE3) {
            double rhs[6];
            se3_law(T.oMf + 12 * B->frame, T.vf + 6 * B->frame, T.af + 6 * B->frame, ref + B->ref, 1, B->kp, B->kd, rhs);
            for (int i = 0; i < 6; ++i) {
                if (!((B->mask >> i) & 1)) continue;
                memcpy(A + (size_t)row * nv, T.Jl + ((size_t)B->frame * 6 + i) * nv, sizeof(double) * nv);
                b1[row++] = rhs[i];
            }
        }
        else if (B->kind == WBCO_T_COM) {
            /* tsid TaskComEquality::compute [UPSTREAM-RECALL]: a_des = -Kp (com - ref) - Kd (vcom - vref) + aref,
             * rows of Jcom, rhs = a_des - drift (drift = com acceleration at ddq = 0); gains tasks.cpp:106-107 */
            const double* r = ref + B->ref;
            for (int i = 0; i < 3; ++i) {
                if (!((B->mask >> i) & 1)) continue;
                memcpy(A + (size_t)row * nv, T.Jcom + (size_t)i * nv, sizeof(double) * nv);
                b1[row++] = (-B->kp * (com[i] - r[i]) - B->kd * (vcom[i] - r[3 + i]) + r[6 + i]) - acom[i];
            }
        }
        else if (B->kind == WBCO_T_MOMENTUM) {
            /* /root/reference/src/tsid/task-momentum-equality.cpp:144-173: L = Ag v, dL_des = -Kp (L - ref') + ref'',
             * rows of Ag, rhs = dL_des - drift */
            const double* r = ref + B->ref;
            for (int i = 0; i < 6; ++i) {
                if (!((B->mask >> i) & 1)) continue;
                double L = 0.0;
                for (int j = 0; j < nv; ++j) L += T.Ag[(size_t)i * nv + j] * v[j];
                memcpy(A + (size_t)row * nv, T.Ag + (size_t)i * nv, sizeof(double) * nv);
                b1[row++] = (-B->kp * (L - r[i]) + r[6 + i]) - dAgv[i];
            }
        }
        else {
            /* /root/reference/src/tsid/task-self-collision.cpp:84-203, 5PL repulsor (:147-156), one row:
             * A = sum grad_C' J, B = -sum [ (Hess_C J v)' J v + grad_C' (-drift + Kd J v) + Kp C ]  (:193-197)
             * with J = linear rows of (WORLD Jacobian of the tracked frame - of the avoided frame) (:100,129,132) and
             * drift = difference of the frames' classical linear accelerations (:92,99,131,133) */
            double* Arow = A + (size_t)row * nv;
            double Bs = 0.0;
            memset(Arow, 0, sizeof(double) * nv);
            const double* pos = T.oMf + 12 * B->frame + 9;
            for (int a = 0; a < B->av_count; ++a) {
                const int fa = map->avoided_frame[B->av_begin + a];
                const double r0 = map->avoided_r0[B->av_begin + a];
                const double* pos2 = T.oMf + 12 * fa + 9;
                double diff[3], drift[3], Jv[3] = {0, 0, 0};
                for (int k = 0; k < 3; ++k) { diff[k] = pos[k] - pos2[k]; drift[k] = T.af[6 * B->frame + k] - T.af[6 * fa + k]; }
                const double sn = dot3(diff, diff), norm = sqrt(sn), aa = r0 + B->radius;
                const double mm_ = B->m;
                const double k5 = -log(pow(-1e-5 + 1., -1. / mm_) - 1.) / B->margin;
                const double s_p = -1. / k5 * log(-1 + pow(2, 1. / mm_));
                const double x = k5 * (norm - aa + s_p);
                const double e_p = exp(-x);
                const double C = 1. - pow(1 + e_p, -mm_);
                const double gscale = -1. / norm * k5 * mm_ * e_p * pow(e_p + 1, -mm_ - 1);
                const double hh = 1. / sn * sqr(k5) * (-mm_ - 1.) * mm_ * exp(-2 * x) * pow(e_p + 1, -mm_ - 2)
                    + 1. / sn * sqr(k5) * mm_ * e_p * pow(e_p + 1, -mm_ - 1)
                    + 1. / pow(norm, 1.5) * k5 * mm_ * e_p * pow(e_p + 1, -mm_ - 1);
                const double iso = -1. / norm * k5 * mm_ * e_p * pow(e_p + 1, -mm_ - 1);
                for (int j = 0; j < nv; ++j) {
                    double Jc[3];
                    for (int k = 0; k < 3; ++k) {
                        Jc[k] = T.Jw[((size_t)B->frame * 6 + k) * nv + j] - T.Jw[((size_t)fa * 6 + k) * nv + j];
                        Jv[k] += Jc[k] * v[j];
                    }
                    Arow[j] += gscale * dot3(diff, Jc);
                }
                /* (Hess J v)' J v with Hess = hh diff diff' + iso I */
                const double dJv = dot3(diff, Jv);
                const double quad = hh * dJv * dJv + iso * dot3(Jv, Jv);
                double gd = 0.0;
                for (int k = 0; k < 3; ++k) gd += gscale * diff[k] * (-drift[k] + B->kd * Jv[k]);
                Bs += -(quad + gd + B->kp * C);
            }
            b1[row++] = Bs;
        }
    }
    /* posture: tsid TaskJointPosture::compute [UPSTREAM-RECALL]: a_des = -Kp (q_a - ref) - Kd v_a (reference velocity and
     * acceleration are zero, tasks.cpp:217); the selection matrix itself is constant structure */
    for (int r = 0; r < map->n_sel; ++r) {
        const int col = map->sel_col[r];
        const int ja = col - (nv - m->na);
        const double qa = q[m->nq - m->na + ja];
        b1[row++] = -map->posture_kp * (qa - ref[map->posture_ref + ja]) - map->posture_kd * v[col];
    }
    /* force regularisation rows: tsid Contact6d force regularisation task has a zero reference force */
    for (int c = 0; c < 6 * map->ncontact; ++c) b1[row++] = 0.0;
    /* torque task: S tau_ref with tau_ref = 0 (tasks.cpp:263-265); cop task: TaskCopEquality's constraint vector is zero */
    for (int j = 0; j < map->n_acteq + (map->cop ? 3 : 0); ++j) b1[row++] = 0.0;
    if (map->cop && Acop) {
        /* tsid TaskCopEquality::compute [UPSTREAM-RECALL]: with the forces of a contact given in its frame (oMf = (R, p)) the
         * tangential moment about cop_ref, n x sum_i (d_i x R f_i), is sum_i (d_i n' - (n . d_i) I) R f_i; n = (0, 0, 1) (the ctor) */
        const int k = 12 * map->ncontact;
        const double nrm[3] = {0.0, 0.0, 1.0};
        for (int c = 0; c < map->ncontact; ++c) {
            const double* oM = T.oMf + 12 * map->contact_frame[c]; /* rotation row-major (9), translation (3) */
            for (int i = 0; i < 4; ++i) {
                const double* pl = map->contact_points + 12 * c + 3 * i;
                double d[3];
                for (int a = 0; a < 3; ++a)
                    d[a] = oM[3 * a] * pl[0] + oM[3 * a + 1] * pl[1] + oM[3 * a + 2] * pl[2] + oM[9 + a] - map->cop_ref[a];
                const double nd = dot3(nrm, d);
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        double s = 0.0; /* ((d n' - nd I) R)(a, b) */
                        for (int mm = 0; mm < 3; ++mm) s += (d[a] * nrm[mm] - (a == mm ? nd : 0.0)) * oM[3 * mm + b];
                        Acop[(size_t)a * k + 12 * c + 3 * i + b] = s;
                    }
            }
        }
    }
    /* contacts: Contact6d::computeMotionTask = TaskSE3Equality in the local frame, all six rows; the reference is a full sample
     * (Contact6dExt::setReference, contact-6d-ext.hpp:18-21), velocity and acceleration zero unless a behaviour sets them
     * (tasks.cpp:359-362) */
    for (int c = 0; c < map->ncontact; ++c) {
        const int f = map->contact_frame[c];
        double rhs[6];
        se3_law(T.oMf + 12 * f, T.vf + 6 * f, T.af + 6 * f, ref + map->contact_ref[c], 1, map->contact_kp[c], map->contact_kd[c], rhs);
        memcpy(Ac + (size_t)c * 6 * nv, T.Jl + (size_t)f * 6 * nv, sizeof(double) * 6 * nv);
        memcpy(bc + 6 * c, rhs, sizeof rhs);
    }
    /* bounds: tsid TaskJointPosVelAccBounds::computeAccLimits [UPSTREAM-RECALL] with the limits of tasks.cpp:284-292
     * (ddq_max = dq_max / dt): the tightest of position, viability, velocity and acceleration limits per joint */
    for (int j = 0; j < map->n_bound; ++j) {
        const double dt = map->dt, qj = q[m->nq - m->na + j], dq = v[nv - m->na + j];
        const double qmin = m->q_lb[j], qmax = m->q_ub[j], dqmax = m->dq_max[j], ddqmax = dqmax / dt;
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
        blb[j] = lb;
        bub[j] = ub;
    }
    free(T.M); free(T.Jcom); free(T.Ag); free(T.oMf); free(T.vf); free(T.af); free(T.Jl); free(T.Jw);
}

typedef struct {
    const wbco_model* m; const wbco_taskmap* map;
    int lo, hi, n_dense;
    const double *q, *v, *ref;
    double *M, *h, *A, *b1, *Ac, *bc, *blb, *bub, *Acop;
} rows_job;

static void* rows_worker(void* arg)
{
    rows_job* J = (rows_job*)arg;
    const wbco_model* m = J->m;
    const wbco_taskmap* map = J->map;
    const int nv = m->nv, lM = nv * (nv + 1) / 2, r1 = J->n_dense + map->n_sel + 6 * map->ncontact + map->n_acteq + (map->cop ? 3 : 0);
    for (int i = J->lo; i < J->hi; ++i)
        wbco_task_rows(m, map, J->q + (size_t)i * m->nq, J->v + (size_t)i * nv, J->ref + (size_t)i * map->nref,
                       J->M + (size_t)i * lM, J->h + (size_t)i * nv, J->A + (size_t)i * J->n_dense * nv, J->b1 + (size_t)i * r1,
                       J->Ac + (size_t)i * map->ncontact * 6 * nv, J->bc + (size_t)i * map->ncontact * 6,
                       J->blb + (size_t)i * map->n_bound, J->bub + (size_t)i * map->n_bound,
                       (map->cop && J->Acop) ? J->Acop + (size_t)i * 36 * map->ncontact : NULL);
    return NULL;
}

void wbco_task_rows_batch(const wbco_model* m, const wbco_taskmap* map, int batch, int n_threads, int n_dense,
                          const double* q, const double* v, const double* ref,
                          double* M, double* h, double* A, double* b1, double* Ac, double* bc, double* blb, double* bub, double* Acop)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > batch) n_threads = batch > 0 ? batch : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
    rows_job* jobs = (rows_job*)malloc(sizeof(rows_job) * n_threads);
    for (int t = 0; t < n_threads; ++t) {
        rows_job j = {m, map, (int)((long long)batch * t / n_threads), (int)((long long)batch * (t + 1) / n_threads), n_dense,
                      q, v, ref, M, h, A, b1, Ac, bc, blb, bub, Acop};
        jobs[t] = j;
        if (n_threads == 1) rows_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, rows_worker, &jobs[t]);
    }
    if (n_threads > 1)
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
}
