/*
 * wbc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See wbc_oracle.h.
 *
 * PARITY UNPINNED (see header): restates published algorithms of un-vendored third-party
 * libraries; every function names the inria_wbc call site it stands behind and the upstream
 * routine it restates.
 *
 * Dense, scalar, row-major, one QP at a time -- on purpose the way the reference does it
 * (Eigen dense blocks, no structure exploitation), so that it is an independent check of the
 * structure-exploiting HIP path.
 */
#include "wbc_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define IDX(i, j, ld) ((size_t)(i) * (size_t)(ld) + (size_t)(j))

void wbco_sizes(const wbco_structure* st, int* n, int* neq, int* nin2, int* r1)
{
    int k = 12 * st->nc;
    int nu = st->nv - st->na; /* un-actuated (floating base) dofs; base-dynamics rows (A.1 ctor) */
    int nin = 0;
    for (int b = 0; b < st->n_ineq_blocks; ++b) {
        if (st->ineq_kind[b] == WBCO_INEQ_BOUNDS) nin += st->n_bound;
        else if (st->ineq_kind[b] == WBCO_INEQ_ACTUATION) nin += st->na;
        else nin += 17;
    }
    if (n) *n = st->nv + k;
    if (neq) *neq = nu + 6 * st->nc;
    if (nin2) *nin2 = 2 * nin;
    if (r1) *r1 = st->n_dense + st->n_sel + 6 * st->nc + st->n_acteq + (st->cop_task >= 0 ? 3 : 0);
}

/* M(i,j) from the packed lower triangle */
static inline double Msym(const double* Mp, int i, int j)
{
    return (i >= j) ? Mp[(size_t)i * (i + 1) / 2 + j] : Mp[(size_t)j * (j + 1) / 2 + i];
}

/*
 * Assembly: tsid InverseDynamicsFormulationAccForce::computeProblemData (call site
 * controller.cpp:244; init call pos_tracker.cpp:106) step 2-6 of SURVEY A.1, followed by
 * tsid SolverHQuadProgFast::solve stacking (call site controller.cpp:247):
 *   level 0: equality  -> CE rows = A, ce0 = -b
 *            inequality-> CI rows = A, ci0 = -lb ; then CI rows = -A, ci0 = ub   (per constraint)
 *   level 1: H += w A'A ; g -= w A'b ; H.diagonal() += hessian_reg
 * Level-0 order: base dynamics (formulation ctor), then constraints in task-stack order;
 * addRigidContact pushes the force inequality, then the motion equality (tasks.cpp:365).
 */
/* doubles of scratch wbco_assemble_ws needs: Jc (k x nv), Arow (n), Ht (n x n), gt (n) */
long wbco_assemble_ws_size(const wbco_structure* st)
{
    int n, neq, nin2, r1;
    wbco_sizes(st, &n, &neq, &nin2, &r1);
    const int k = 12 * st->nc;
    return (long)(k > 0 ? k : 1) * st->nv + n + (long)n * n + n;
}

static void wbco_assemble_ws(const wbco_structure* st, const wbco_inputs* in,
                             double* H, double* g, double* CE, double* ce0, double* CI, double* ci0, double* scratch)
{
    const int nv = st->nv, na = st->na, nc = st->nc, k = 12 * nc, nu = nv - na;
    int n, neq, nin2, r1;
    wbco_sizes(st, &n, &neq, &nin2, &r1);

    /* Jc = T' * A_c  (12 x nv per contact): m_Jc.middleRows(idx,12) = T.transpose()*mc.matrix() */
    double* Jc = scratch;
    memset(Jc, 0, sizeof(double) * (size_t)(k > 0 ? k : 1) * nv);
    for (int c = 0; c < nc; ++c) {
        const double* T = st->force_gen + (size_t)c * 72;
        const double* Ac = in->Ac + (size_t)c * 6 * nv;
        for (int m = 0; m < 12; ++m)
            for (int j = 0; j < nv; ++j) {
                double s = 0.0;
                for (int r = 0; r < 6; ++r) s += T[IDX(r, m, 12)] * Ac[IDX(r, j, nv)];
                Jc[IDX(12 * c + m, j, nv)] = s;
            }
    }

    /* ---- level 0 equalities -> CE, ce0 ---- */
    memset(CE, 0, sizeof(double) * (size_t)neq * n);
    int ie = 0;
    /* base dynamics: [M_u | -J_u'] x = -h_u  (A.1 step 3) */
    for (int i = 0; i < nu; ++i, ++ie) {
        for (int j = 0; j < nv; ++j) CE[IDX(ie, j, n)] = Msym(in->M, i, j);
        for (int m = 0; m < k; ++m) CE[IDX(ie, nv + m, n)] = -Jc[IDX(m, i, nv)];
        ce0[ie] = -(-in->h[i]);
    }
    /* contact motion constraints: leftCols(nv) = mc.A, vector = mc.b (A.1 step 2) */
    for (int c = 0; c < nc; ++c)
        for (int r = 0; r < 6; ++r, ++ie) {
            for (int j = 0; j < nv; ++j) CE[IDX(ie, j, n)] = in->Ac[IDX(c * 6 + r, j, nv)];
            ce0[ie] = -in->bc[c * 6 + r];
        }

    /* ---- level 0 inequalities -> CI, ci0 ---- */
    memset(CI, 0, sizeof(double) * (size_t)nin2 * n);
    int ii = 0;
    for (int b = 0; b < st->n_ineq_blocks; ++b) {
        const int kind = st->ineq_kind[b];
        if (kind == WBCO_INEQ_BOUNDS) {
            /* TaskJointPosVelAccBounds as an inequality with A = selection (A.1 addTask/step 4) */
            const int rows = st->n_bound;
            for (int r = 0; r < rows; ++r) {
                CI[IDX(ii + r, st->bound_col[r], n)] = 1.0;
                ci0[ii + r] = -in->blb[r];
                CI[IDX(ii + rows + r, st->bound_col[r], n)] = -1.0;
                ci0[ii + rows + r] = in->bub[r];
            }
            ii += 2 * rows;
        }
        else if (kind == WBCO_INEQ_ACTUATION) {
            /* TaskActuationBounds: [M_a | -J_a'] with lb - h_a, ub - h_a (A.1 step 6) */
            for (int r = 0; r < na; ++r) {
                for (int j = 0; j < nv; ++j) {
                    double v = Msym(in->M, nu + r, j);
                    CI[IDX(ii + r, j, n)] = v;
                    CI[IDX(ii + na + r, j, n)] = -v;
                }
                for (int m = 0; m < k; ++m) {
                    double v = -Jc[IDX(m, nu + r, nv)];
                    CI[IDX(ii + r, nv + m, n)] = v;
                    CI[IDX(ii + na + r, nv + m, n)] = -v;
                }
                ci0[ii + r] = -(in->tlb[r] - in->h[nu + r]);
                ci0[ii + na + r] = in->tub[r] - in->h[nu + r];
            }
            ii += 2 * na;
        }
        else {
            /* Contact6d force inequality, 17 x 12 in the contact's force columns (A.1 step 2) */
            const int c = st->ineq_arg[b];
            const double* B = st->fric_mat + (size_t)c * 17 * 12;
            for (int r = 0; r < 17; ++r) {
                for (int m = 0; m < 12; ++m) {
                    CI[IDX(ii + r, nv + 12 * c + m, n)] = B[IDX(r, m, 12)];
                    CI[IDX(ii + 17 + r, nv + 12 * c + m, n)] = -B[IDX(r, m, 12)];
                }
                ci0[ii + r] = -st->fric_lb[c * 17 + r];
                ci0[ii + 17 + r] = st->fric_ub[c * 17 + r];
            }
            ii += 34;
        }
    }

    /* ---- level 1 -> H, g (dense, the way Eigen does: H += w * A' * A per task) ---- */
    memset(H, 0, sizeof(double) * (size_t)n * n);
    memset(g, 0, sizeof(double) * (size_t)n);
    double* Arow = Jc + (size_t)(k > 0 ? k : 1) * nv;
    double* Ht = Arow + n; /* per-task A'A */
    double* gt = Ht + (size_t)n * n;
    /* dense motion rows, grouped by task (consecutive rows of one task share dense_row_task) */
    int r = 0;
    while (r < st->n_dense) {
        const int t = st->dense_row_task[r];
        memset(Ht, 0, sizeof(double) * (size_t)n * n);
        memset(gt, 0, sizeof(double) * (size_t)n);
        while (r < st->n_dense && st->dense_row_task[r] == t) {
            memset(Arow, 0, sizeof(double) * (size_t)n);
            for (int j = 0; j < nv; ++j) Arow[j] = in->A[IDX(r, j, nv)];
            for (int i = 0; i < n; ++i) {
                if (Arow[i] == 0.0) continue;
                for (int j = 0; j < n; ++j) Ht[IDX(i, j, n)] += Arow[i] * Arow[j];
                gt[i] += Arow[i] * in->b1[r];
            }
            ++r;
        }
        for (int i = 0; i < n * n; ++i) H[i] += in->w[t] * Ht[i];
        for (int i = 0; i < n; ++i) g[i] -= in->w[t] * gt[i];
    }
    /* selection rows (TaskJointPosture): A = [0 | S] */
    for (int s = 0; s < st->n_sel; ++s) {
        const int c = st->sel_col[s];
        const double w = in->w[st->sel_task[s]];
        H[IDX(c, c, n)] += w * 1.0;
        g[c] -= w * in->b1[st->n_dense + s];
    }
    /* contact force regularisation: matrix = diag(w_f) T in the contact's 12 force columns */
    for (int c = 0; c < nc; ++c) {
        const double* F = st->forcereg_mat + (size_t)c * 72;
        const double w = in->w[st->forcereg_task[c]];
        const double* b = in->b1 + st->n_dense + st->n_sel + 6 * c;
        for (int a = 0; a < 12; ++a) {
            double gb = 0.0;
            for (int q = 0; q < 6; ++q) gb += F[IDX(q, a, 12)] * b[q];
            g[nv + 12 * c + a] -= w * gb;
            for (int bb = 0; bb < 12; ++bb) {
                double s = 0.0;
                for (int q = 0; q < 6; ++q) s += F[IDX(q, a, 12)] * F[IDX(q, bb, 12)];
                H[IDX(nv + 12 * c + a, nv + 12 * c + bb, n)] += w * s;
            }
        }
    }
    /* actuation task ("torque", tasks.cpp:227-271; tsid TaskActuationEquality through computeProblemData's actuation-task
     * branch): A = S [M_a | -J_a'], b = S tau_ref - S h_a, S(j, joint_j) = scale_j */
    if (st->n_acteq > 0) {
        const double w = in->w[st->acteq_task];
        const double* bt = in->b1 + st->n_dense + st->n_sel + 6 * nc;
        memset(Ht, 0, sizeof(double) * (size_t)n * n);
        memset(gt, 0, sizeof(double) * (size_t)n);
        for (int j = 0; j < st->n_acteq; ++j) {
            const int a = st->acteq_joint[j];
            const double sc = st->acteq_scale[j];
            for (int c = 0; c < nv; ++c) Arow[c] = sc * Msym(in->M, nu + a, c);
            for (int m = 0; m < k; ++m) Arow[nv + m] = -sc * Jc[IDX(m, nu + a, nv)];
            const double b = bt[j] - sc * in->h[nu + a];
            for (int i = 0; i < n; ++i) {
                for (int c = 0; c < n; ++c) Ht[IDX(i, c, n)] += Arow[i] * Arow[c];
                gt[i] += Arow[i] * b;
            }
        }
        for (int i = 0; i < n * n; ++i) H[i] += w * Ht[i];
        for (int i = 0; i < n; ++i) g[i] -= w * gt[i];
    }
    /* force task ("cop", tasks.cpp:156-178; tsid TaskCopEquality is associated with no single contact, so its 3 x k matrix
     * lands in columns nv .. nv + k: computeProblemData's m_taskContactForces loop) */
    if (st->cop_task >= 0 && k > 0) {
        const double w = in->w[st->cop_task];
        const double* bcop = in->b1 + st->n_dense + st->n_sel + 6 * nc + st->n_acteq;
        for (int r3 = 0; r3 < 3; ++r3) {
            const double* a = in->Acop + (size_t)r3 * k;
            for (int i = 0; i < k; ++i) {
                for (int c = 0; c < k; ++c) H[IDX(nv + i, nv + c, n)] += w * a[i] * a[c];
                g[nv + i] -= w * a[i] * bcop[r3];
            }
        }
    }
    for (int i = 0; i < n; ++i) H[IDX(i, i, n)] += st->hessian_reg;
}

void wbco_assemble(const wbco_structure* st, const wbco_inputs* in,
                   double* H, double* g, double* CE, double* ce0, double* CI, double* ci0)
{
    double* scratch = (double*)malloc(sizeof(double) * (size_t)wbco_assemble_ws_size(st));
    wbco_assemble_ws(st, in, H, g, CE, ce0, CI, ci0, scratch);
    free(scratch);
}

/* ------------------------------------------------------------------------------------------
 * eiquadprog-fast: Goldfarb-Idnani dual active set, Cholesky + Givens-updated (J,R)
 * ------------------------------------------------------------------------------------------ */

/* eiquadprog utils::distance: overflow-safe hypot */
static inline double gi_distance(double a, double b)
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

/* d = J' np */
static void gi_compute_d(int n, double* d, const double* J, const double* np)
{
    for (int c = 0; c < n; ++c) d[c] = 0.0;
    for (int k = 0; k < n; ++k) {
        const double v = np[k];
        if (v == 0.0) continue;
        const double* Jk = J + IDX(k, 0, n);
        for (int c = 0; c < n; ++c) d[c] += Jk[c] * v;
    }
}
/* z = J[:, iq:] d[iq:] */
static void gi_update_z(int n, double* z, const double* J, const double* d, int iq)
{
    for (int k = 0; k < n; ++k) {
        double s = 0.0;
        const double* Jk = J + IDX(k, 0, n);
        for (int c = iq; c < n; ++c) s += Jk[c] * d[c];
        z[k] = s;
    }
}
/* r = R[:iq,:iq]^-1 d[:iq], R upper triangular */
static void gi_update_r(int n, const double* R, double* r, const double* d, int iq)
{
    for (int i = iq - 1; i >= 0; --i) {
        double s = d[i];
        for (int j = i + 1; j < iq; ++j) s -= R[IDX(i, j, n)] * r[j];
        r[i] = s / R[IDX(i, i, n)];
    }
}

static int gi_add_constraint(int n, double* R, double* J, double* d, int* iq, double* R_norm)
{
    for (int j = n - 1; j >= *iq + 1; --j) {
        /* Givens "rotation" with the matrix (cc ss; ss -cc) reducing d(j) to zero */
        double cc = d[j - 1], ss = d[j];
        double h = gi_distance(cc, ss);
        if (h == 0.0) continue;
        d[j] = 0.0;
        ss = ss / h;
        cc = cc / h;
        if (cc < 0.0) {
            cc = -cc;
            ss = -ss;
            d[j - 1] = -h;
        }
        else
            d[j - 1] = h;
        double xny = ss / (1.0 + cc);
        for (int k = 0; k < n; ++k) {
            double t1 = J[IDX(k, j - 1, n)];
            double t2 = J[IDX(k, j, n)];
            J[IDX(k, j - 1, n)] = t1 * cc + t2 * ss;
            J[IDX(k, j, n)] = xny * (t1 + J[IDX(k, j - 1, n)]) - t2;
        }
    }
    (*iq)++;
    /* put the iq components of d into column iq-1 of R */
    for (int i = 0; i < *iq; ++i) R[IDX(i, *iq - 1, n)] = d[i];
    if (fabs(d[*iq - 1]) <= DBL_EPSILON * (*R_norm)) return 0; /* degenerate */
    *R_norm = fmax(*R_norm, fabs(d[*iq - 1]));
    return 1;
}

static void gi_delete_constraint(int n, double* R, double* J, int* A, double* u, int neq, int* iq, int l)
{
    int qq = 0;
    for (int i = neq; i < *iq; ++i)
        if (A[i] == l) {
            qq = i;
            break;
        }
    /* remove the constraint from the active set and the duals */
    for (int i = qq; i < *iq - 1; ++i) {
        A[i] = A[i + 1];
        u[i] = u[i + 1];
        for (int j = 0; j < n; ++j) R[IDX(j, i, n)] = R[IDX(j, i + 1, n)];
    }
    A[*iq - 1] = A[*iq];
    u[*iq - 1] = u[*iq];
    A[*iq] = 0;
    u[*iq] = 0.0;
    for (int j = 0; j < *iq; ++j) R[IDX(j, *iq - 1, n)] = 0.0;
    (*iq)--;
    if (*iq == 0) return;

    for (int j = qq; j < *iq; ++j) {
        double cc = R[IDX(j, j, n)];
        double ss = R[IDX(j + 1, j, n)];
        double h = gi_distance(cc, ss);
        if (h == 0.0) continue;
        cc = cc / h;
        ss = ss / h;
        R[IDX(j + 1, j, n)] = 0.0;
        if (cc < 0.0) {
            R[IDX(j, j, n)] = -h;
            cc = -cc;
            ss = -ss;
        }
        else
            R[IDX(j, j, n)] = h;
        double xny = ss / (1.0 + cc);
        for (int k = j + 1; k < *iq; ++k) {
            double t1 = R[IDX(j, k, n)];
            double t2 = R[IDX(j + 1, k, n)];
            R[IDX(j, k, n)] = t1 * cc + t2 * ss;
            R[IDX(j + 1, k, n)] = xny * (t1 + R[IDX(j, k, n)]) - t2;
        }
        for (int k = 0; k < n; ++k) {
            double t1 = J[IDX(k, j, n)];
            double t2 = J[IDX(k, j + 1, n)];
            J[IDX(k, j, n)] = t1 * cc + t2 * ss;
            J[IDX(k, j + 1, n)] = xny * (J[IDX(k, j, n)] + t1) - t2;
        }
    }
}

long wbco_ws_size(int n, int neq, int nin2)
{
    /* J, R (n*n each), L (n*n), d z r np x_old (5n), u u_old (2(n+1)... sized neq+nin2+1), s (nin2),
       ints packed into doubles: A A_old (2*(neq+nin2+1)), iai iaexcl (2*nin2) */
    long m = neq + nin2 + 2;
    return 3L * n * n + 5L * n + 2L * m + nin2 + 2L * m + 2L * nin2 + 64;
}

int wbco_eiquadprog_fast(int n, int neq, int nin2,
                         const double* H, const double* g,
                         const double* CE, const double* ce0,
                         const double* CI, const double* ci0,
                         double* x, double* u_out, int* A_out, int* iq_out, int* iter_out, double* fval,
                         int max_iter, double* ws_in)
{
    double* ws = ws_in ? ws_in : (double*)malloc(sizeof(double) * (size_t)wbco_ws_size(n, neq, nin2));
    const long m = neq + nin2 + 2;
    double* J = ws;
    double* R = J + (size_t)n * n;
    double* L = R + (size_t)n * n;
    double* d = L + (size_t)n * n;
    double* z = d + n;
    double* r = z + n;
    double* np = r + n;
    double* x_old = np + n;
    double* u = x_old + n;
    double* u_old = u + m;
    double* s = u_old + m;
    int* A = (int*)(s + nin2);
    int* A_old = A + m;
    int* iai = A_old + m;
    int* iaexcl = iai + nin2 + 1;

    const double inf = INFINITY;
    int status = WBCO_EIQ_OPTIMAL;
    int iter = 0, iq = 0, ip = 0, l = 0;
    double f_value = 0.0, psi, c1, c2, ss, R_norm, t, t1, t2;

    /* ---- preprocessing ---- */
    c1 = 0.0;
    for (int i = 0; i < n; ++i) c1 += H[IDX(i, i, n)];
    /* chol_.compute(H): H = L L' (Eigen LLT, lower) */
    for (int j = 0; j < n; ++j) {
        double sum = H[IDX(j, j, n)];
        for (int p = 0; p < j; ++p) sum -= L[IDX(j, p, n)] * L[IDX(j, p, n)];
        double ljj = sqrt(sum); /* NaN propagates like Eigen's LLT on a non-SPD matrix */
        L[IDX(j, j, n)] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double v = H[IDX(i, j, n)];
            for (int p = 0; p < j; ++p) v -= L[IDX(i, p, n)] * L[IDX(j, p, n)];
            L[IDX(i, j, n)] = v / ljj;
        }
        for (int i = 0; i < j; ++i) L[IDX(i, j, n)] = 0.0;
    }
    memset(d, 0, sizeof(double) * n);
    memset(R, 0, sizeof(double) * (size_t)n * n);
    R_norm = 1.0;
    /* J = L^-T : solve U J = I with U = L' (back substitution per column) */
    for (int c = 0; c < n; ++c) {
        for (int i = n - 1; i >= 0; --i) {
            double v = (i == c) ? 1.0 : 0.0;
            if (i > c) {
                J[IDX(i, c, n)] = 0.0;
                continue;
            }
            for (int p = i + 1; p <= c; ++p) v -= L[IDX(p, i, n)] * J[IDX(p, c, n)];
            J[IDX(i, c, n)] = v / L[IDX(i, i, n)];
        }
    }
    c2 = 0.0;
    for (int i = 0; i < n; ++i) c2 += J[IDX(i, i, n)];

    /* x = -H^-1 g via chol_.solve(g): L y = g ; L' x = y */
    for (int i = 0; i < n; ++i) {
        double v = g[i];
        for (int p = 0; p < i; ++p) v -= L[IDX(i, p, n)] * z[p];
        z[i] = v / L[IDX(i, i, n)];
    }
    for (int i = n - 1; i >= 0; --i) {
        double v = z[i];
        for (int p = i + 1; p < n; ++p) v -= L[IDX(p, i, n)] * x[p];
        x[i] = v / L[IDX(i, i, n)];
    }
    for (int i = 0; i < n; ++i) x[i] = -x[i];
    f_value = 0.0;
    for (int i = 0; i < n; ++i) f_value += 0.5 * g[i] * x[i];

    for (long i = 0; i < m; ++i) {
        u[i] = 0.0;
        A[i] = 0;
    }

    /* ---- add equality constraints to the working set ---- */
    for (int i = 0; i < neq; ++i) {
        for (int j = 0; j < n; ++j) np[j] = CE[IDX(i, j, n)];
        gi_compute_d(n, d, J, np);
        gi_update_z(n, z, J, d, iq);
        gi_update_r(n, R, r, d, iq);
        double zz = 0.0, znp = 0.0, npx = 0.0;
        for (int j = 0; j < n; ++j) {
            zz += z[j] * z[j];
            znp += z[j] * np[j];
            npx += np[j] * x[j];
        }
        t2 = 0.0;
        if (fabs(zz) > DBL_EPSILON) t2 = (-npx - ce0[i]) / znp;
        for (int j = 0; j < n; ++j) x[j] += t2 * z[j];
        u[iq] = t2;
        for (int j = 0; j < iq; ++j) u[j] -= t2 * r[j];
        f_value += 0.5 * (t2 * t2) * znp;
        A[i] = -i - 1;
        if (!gi_add_constraint(n, R, J, d, &iq, &R_norm)) {
            status = WBCO_EIQ_REDUNDANT_EQUALITIES;
            goto done;
        }
    }

    for (int i = 0; i < nin2; ++i) iai[i] = i;

l1:
    iter++;
    if (iter >= max_iter) {
        status = WBCO_EIQ_MAX_ITER_REACHED;
        goto done;
    }
    /* step 1: choose a violated constraint */
    for (int i = neq; i < iq; ++i) {
        ip = A[i];
        iai[ip] = -1;
    }
    ss = 0.0;
    ip = 0;
    psi = 0.0;
    for (int i = 0; i < nin2; ++i) {
        double v = ci0[i];
        const double* Ci = CI + IDX(i, 0, n);
        for (int j = 0; j < n; ++j) v += Ci[j] * x[j];
        s[i] = v;
        iaexcl[i] = 1;
        psi += fmin(0.0, v);
    }
    if (fabs(psi) <= nin2 * DBL_EPSILON * c1 * c2 * 100.0) {
        status = WBCO_EIQ_OPTIMAL; /* numerically no infeasibility left */
        goto done;
    }
    for (int i = 0; i < iq; ++i) {
        u_old[i] = u[i];
        A_old[i] = A[i];
    }
    for (int i = 0; i < n; ++i) x_old[i] = x[i];

l2:
    /* step 2: most violated constraint among K \ A not excluded (first index wins ties).
     * Restated with `ss` re-armed on every entry: upstream leaves the previous (negative) value
     * in place when it comes back here after a failed add_constraint, which re-selects the
     * just-excluded constraint; the evident intent (GI step 2 over the non-excluded set) is kept. */
    ss = 0.0;
    for (int i = 0; i < nin2; ++i) {
        if (s[i] < ss && iai[i] != -1 && iaexcl[i]) {
            ss = s[i];
            ip = i;
        }
    }
    if (ss >= 0.0) {
        status = WBCO_EIQ_OPTIMAL;
        goto done;
    }
    for (int j = 0; j < n; ++j) np[j] = CI[IDX(ip, j, n)];
    u[iq] = 0.0;
    A[iq] = ip;

l2a:
    /* step 2a: step direction in primal (z) and dual (r) space */
    gi_compute_d(n, d, J, np);
    gi_update_z(n, z, J, d, iq);
    gi_update_r(n, R, r, d, iq);

    /* step 2b: step lengths */
    l = 0;
    t1 = inf;
    for (int k = neq; k < iq; ++k) {
        double tmp;
        if (r[k] > 0.0 && ((tmp = u[k] / r[k]) < t1)) {
            t1 = tmp;
            l = A[k];
        }
    }
    {
        double zz = 0.0, znp = 0.0;
        for (int j = 0; j < n; ++j) {
            zz += z[j] * z[j];
            znp += z[j] * np[j];
        }
        if (fabs(zz) > DBL_EPSILON)
            t2 = -s[ip] / znp;
        else
            t2 = inf;
        t = fmin(t1, t2);

        /* step 2c */
        if (t >= inf) {
            status = WBCO_EIQ_UNBOUNDED; /* dual unbounded = primal infeasible */
            goto done;
        }
        if (t2 >= inf) {
            /* (ii) step in dual space only: drop constraint l */
            for (int k = 0; k < iq; ++k) u[k] -= t * r[k];
            u[iq] += t;
            iai[l] = l;
            gi_delete_constraint(n, R, J, A, u, neq, &iq, l);
            goto l2a;
        }
        /* (iii) step in primal and dual space */
        for (int j = 0; j < n; ++j) x[j] += t * z[j];
        f_value += t * znp * (0.5 * t + u[iq]);
        for (int k = 0; k < iq; ++k) u[k] -= t * r[k];
        u[iq] += t;
    }
    if (t == t2) {
        /* full step: add constraint ip to the active set */
        if (!gi_add_constraint(n, R, J, d, &iq, &R_norm)) {
            iaexcl[ip] = 0;
            gi_delete_constraint(n, R, J, A, u, neq, &iq, ip);
            for (int i = 0; i < nin2; ++i) iai[i] = i;
            for (int i = 0; i < iq; ++i) {
                A[i] = A_old[i];
                if (A[i] >= 0) iai[A[i]] = -1; /* upstream indexes iai with the negative equality tags too (UB); guarded */
                u[i] = u_old[i];
            }
            for (int i = 0; i < n; ++i) x[i] = x_old[i];
            goto l2;
        }
        else
            iai[ip] = -1;
        goto l1;
    }
    /* partial step: drop constraint l */
    iai[l] = l;
    gi_delete_constraint(n, R, J, A, u, neq, &iq, l);
    {
        double v = ci0[ip];
        for (int j = 0; j < n; ++j) v += CI[IDX(ip, j, n)] * x[j];
        s[ip] = v;
    }
    goto l2a;

done:
    if (iq_out) *iq_out = iq;
    if (iter_out) *iter_out = iter;
    if (fval) *fval = f_value;
    if (u_out)
        for (int i = 0; i < iq; ++i) u_out[i] = u[i];
    if (A_out)
        for (int i = 0; i < iq; ++i) A_out[i] = A[i];
    if (!ws_in) free(ws);
    return status;
}

/* ------------------------------------------------------------------------------------------ */
static double now_s(void);

double wbco_eiquadprog_timed(int n, int neq, int nin2, const double* H, const double* g, const double* CE, const double* ce0,
                             const double* CI, const double* ci0, int max_iter, int reps, int* status_out, int* iter_out)
{
    if (reps < 1) reps = 1;
    double* ws = (double*)malloc(sizeof(double) * (size_t)wbco_ws_size(n, neq, nin2));
    double* x = (double*)malloc(sizeof(double) * (size_t)n);
    double* u = (double*)malloc(sizeof(double) * (size_t)(neq + nin2 + 2));
    int* A = (int*)malloc(sizeof(int) * (size_t)(neq + nin2 + 2));
    if (!ws || !x || !u || !A) return -1.0;
    int iq = 0, iter = 0, st = 0;
    double fval = 0.0;
    st = wbco_eiquadprog_fast(n, neq, nin2, H, g, CE, ce0, CI, ci0, x, u, A, &iq, &iter, &fval, max_iter, ws);
    const double t0 = now_s();
    for (int r = 0; r < reps; ++r) st = wbco_eiquadprog_fast(n, neq, nin2, H, g, CE, ce0, CI, ci0, x, u, A, &iq, &iter, &fval, max_iter, ws);
    const double dt = now_s() - t0;
    if (status_out) *status_out = st;
    if (iter_out) *iter_out = iter;
    free(ws); free(x); free(u); free(A);
    return dt;
}

long wbco_tick_ws_size(const wbco_structure* st)
{
    int n, neq, nin2, r1;
    wbco_sizes(st, &n, &neq, &nin2, &r1);
    return (long)n * n + n + (long)neq * n + neq + (long)nin2 * n + nin2 + wbco_ws_size(n, neq, nin2) + wbco_assemble_ws_size(st) + 16;
}

/* One control tick of the path: Controller::_solve lines 244-251 (controller.cpp). */
int wbco_tick(const wbco_structure* st, const wbco_inputs* in, wbco_outputs* out, double* ws_in)
{
    int n, neq, nin2, r1;
    wbco_sizes(st, &n, &neq, &nin2, &r1);
    double* ws = ws_in ? ws_in : (double*)malloc(sizeof(double) * (size_t)wbco_tick_ws_size(st));
    double* H = ws;
    double* g = H + (size_t)n * n;
    double* CE = g + n;
    double* ce0 = CE + (size_t)neq * n;
    double* CI = ce0 + neq;
    double* ci0 = CI + (size_t)nin2 * n;
    double* qws = ci0 + nin2;
    double* aws = qws + wbco_ws_size(n, neq, nin2);

    wbco_assemble_ws(st, in, H, g, CE, ce0, CI, ci0, aws);
    int iq = 0, iter = 0;
    double fval = 0.0;
    int est = wbco_eiquadprog_fast(n, neq, nin2, H, g, CE, ce0, CI, ci0, out->x, out->lambda, out->active,
                                   &iq, &iter, &fval, st->max_iter, qws);
    /* SolverHQuadProgFast::solve status map */
    int status;
    switch (est) {
    case WBCO_EIQ_OPTIMAL: status = WBCO_HQP_OPTIMAL; break;
    case WBCO_EIQ_UNBOUNDED: status = WBCO_HQP_INFEASIBLE; break;
    case WBCO_EIQ_MAX_ITER_REACHED: status = WBCO_HQP_MAX_ITER_REACHED; break;
    case WBCO_EIQ_REDUNDANT_EQUALITIES: status = WBCO_HQP_ERROR; break;
    default: status = WBCO_HQP_UNKNOWN; break;
    }
    out->status = status;
    out->iters = iter;
    out->n_active = iq;
    out->fval = fval;

    /* P4 decode: tau = h_a + M_a dv - J_a' f  (getActuatorForces, controller.cpp:250) */
    const int nv = st->nv, na = st->na, nc = st->nc, nu = nv - na;
    for (int i = 0; i < na; ++i) {
        double tau = in->h[nu + i];
        for (int j = 0; j < nv; ++j) tau += Msym(in->M, nu + i, j) * out->x[j];
        for (int c = 0; c < nc; ++c) {
            const double* T = st->force_gen + (size_t)c * 72;
            const double* Ac = in->Ac + (size_t)c * 6 * nv;
            for (int m = 0; m < 12; ++m) {
                double jc = 0.0; /* Jc(12c+m, nu+i) */
                for (int r = 0; r < 6; ++r) jc += T[IDX(r, m, 12)] * Ac[IDX(r, nu + i, nv)];
                tau -= jc * out->x[nv + 12 * c + m];
            }
        }
        out->tau[i] = tau;
    }
    if (!ws_in) free(ws);
    return status;
}

/* ---- batched driver (pthreads): work items = reps x batch QPs handed out from one atomic counter in chunks, so that the
 * threads finish together whatever the iteration counts; per-thread workspace allocated once; nothing is allocated per QP.
 * Pass 0 writes the caller's outputs; later passes (reps > 1: the CPU baseline's timed sample) solve the same QPs again into
 * per-thread scratch.  The clock starts when every thread stands at the start line and stops when the last one is done:
 * thread creation is not part of what is timed. ---- */
#include <stdatomic.h>
#include <time.h>
typedef struct {
    const wbco_structure* st;
    const wbco_batch_inputs* in;
    const wbco_batch_outputs* out;
    int batch;
    long total;
    atomic_long* next;
    pthread_barrier_t* start;
} batch_job;

enum { WBCO_CHUNK = 4 };

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* batch_worker(void* arg)
{
    batch_job* job = (batch_job*)arg;
    const wbco_structure* st = job->st;
    int n, neq, nin2, r1;
    wbco_sizes(st, &n, &neq, &nin2, &r1);
    const int nv = st->nv, na = st->na, nc = st->nc;
    const size_t mlen = (size_t)nv * (nv + 1) / 2;
    double* ws = (double*)malloc(sizeof(double) * ((size_t)wbco_tick_ws_size(st) + n + (na > 0 ? na : 1)));
    double* sx = ws + wbco_tick_ws_size(st);
    double* stau = sx + n;
    if (job->start) pthread_barrier_wait(job->start);
    for (;;) {
        const long t0 = atomic_fetch_add(job->next, (long)WBCO_CHUNK);
        if (t0 >= job->total) break;
        const long t1 = t0 + WBCO_CHUNK < job->total ? t0 + WBCO_CHUNK : job->total;
        for (long t = t0; t < t1; ++t) {
            const int i = (int)(t % job->batch);
            const int first = t < job->batch;
            wbco_inputs in;
            in.M = job->in->M + (size_t)i * mlen;
            in.h = job->in->h + (size_t)i * nv;
            in.A = job->in->A + (size_t)i * st->n_dense * nv;
            in.b1 = job->in->b1 + (size_t)i * r1;
            in.Ac = job->in->Ac + (size_t)i * nc * 6 * nv;
            in.bc = job->in->bc + (size_t)i * nc * 6;
            in.blb = job->in->blb + (size_t)i * st->n_bound;
            in.bub = job->in->bub + (size_t)i * st->n_bound;
            in.tlb = job->in->tlb + (size_t)i * na;
            in.tub = job->in->tub + (size_t)i * na;
            in.w = job->in->w + (size_t)i * st->n_tasks;
            in.Acop = (st->cop_task >= 0 && job->in->Acop) ? job->in->Acop + (size_t)i * 36 * nc : NULL;
            wbco_outputs out;
            out.x = first ? job->out->x + (size_t)i * n : sx;
            out.tau = first ? job->out->tau + (size_t)i * na : stau;
            out.lambda = NULL;
            out.active = (first && job->out->active) ? job->out->active + (size_t)i * (neq + nin2) : NULL;
            wbco_tick(st, &in, &out, ws);
            if (first) {
                job->out->status[i] = out.status;
                job->out->iters[i] = out.iters;
                if (job->out->n_active) job->out->n_active[i] = out.n_active;
                if (job->out->fval) job->out->fval[i] = out.fval;
            }
        }
    }
    free(ws);
    return NULL;
}

/* reps passes over the batch on nthreads threads; returns the seconds between the start line and the last thread's end
 * (negative on error).  reps <= 1: one pass. */
double wbco_tick_batch_timed(const wbco_structure* st, int batch, const wbco_batch_inputs* in,
                             const wbco_batch_outputs* out, int nthreads, int reps)
{
    if (batch <= 0) return 0.0;
    if (reps < 1) reps = 1;
    if (nthreads < 1) nthreads = 1;
    const long total = (long)batch * reps;
    if ((long)nthreads * WBCO_CHUNK > total) nthreads = (int)((total + WBCO_CHUNK - 1) / WBCO_CHUNK);
    atomic_long next;
    atomic_init(&next, 0);
    batch_job job = {st, in, out, batch, total, &next, NULL};
    if (nthreads == 1) {
        const double t0 = now_s();
        batch_worker(&job);
        return now_s() - t0;
    }
    pthread_barrier_t start;
    if (pthread_barrier_init(&start, NULL, (unsigned)nthreads + 1) != 0) return -1.0;
    job.start = &start;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    int made = 0;
    for (; made < nthreads; ++made)
        if (pthread_create(&th[made], NULL, batch_worker, &job) != 0) break;
    if (made < nthreads) { /* cannot reach the start line: release what exists through the counter and give up */
        atomic_store(&next, total);
        for (int t = made; t < nthreads; ++t) pthread_create(&th[t], NULL, batch_worker, &job);
    }
    pthread_barrier_wait(&start);
    const double t0 = now_s();
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    const double dt = now_s() - t0;
    pthread_barrier_destroy(&start);
    free(th);
    return made < nthreads ? -1.0 : dt;
}

int wbco_tick_batch(const wbco_structure* st, int batch, const wbco_batch_inputs* in,
                    const wbco_batch_outputs* out, int nthreads)
{
    return wbco_tick_batch_timed(st, batch, in, out, nthreads, 1) < 0.0 ? -1 : 0;
}


/* ------------------------------------------------------------------------------------------------
 * After the path: state integration (controller.cpp:250-272), see wbc_oracle.h
 * ------------------------------------------------------------------------------------------------ */
static void quat_to_rot(const double* qt /* x y z w */, double R[9])
{
    /* Eigen::Quaternion::toRotationMatrix */
    const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

static void rot_to_quat(const double R[9], double* qt /* x y z w */)
{
    /* Eigen quaternionbase_assign_impl<Matrix3> (Ken Shoemake) */
    double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        qt[3] = 0.5 * t;
        t = 0.5 / t;
        qt[0] = (R[7] - R[5]) * t;
        qt[1] = (R[2] - R[6]) * t;
        qt[2] = (R[3] - R[1]) * t;
    }
    else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        qt[i] = 0.5 * t;
        t = 0.5 / t;
        qt[3] = (R[3 * k + j] - R[3 * j + k]) * t;
        qt[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        qt[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}

void wbco_integrate(int floating_base, int nv, double dt, const double* q, const double* dq, const double* dv,
                    double* q_next, double* v_next, double* q_solver)
{
    for (int j = 0; j < nv; ++j) v_next[j] = dq[j] + dt * dv[j]; /* controller.cpp:254 */
    if (!floating_base) {
        for (int j = 0; j < nv; ++j) q_next[j] = q[j] + dt * v_next[j]; /* revolute joints: q + v */
        if (q_solver)
            for (int j = 0; j < nv; ++j) q_solver[j] = q_next[j]; /* controller.cpp:273 */
        return;
    }
    /* free flyer: M1 = M0 * exp6(dt v) */
    const double v[3] = {dt * v_next[0], dt * v_next[1], dt * v_next[2]};
    const double w[3] = {dt * v_next[3], dt * v_next[4], dt * v_next[5]};
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double t = sqrt(t2);
    double ct, alpha_v, alpha_wxv, alpha_w;
    const double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
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
    const double c[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
    double tr[3], E[9];
    for (int i = 0; i < 3; ++i) tr[i] = alpha_v * v[i] + alpha_w * w[i] + alpha_wxv * c[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) E[3 * i + j] = alpha_wxv * w[i] * w[j];
    E[0] += ct; E[4] += ct; E[8] += ct;
    E[1] -= alpha_v * w[2]; E[3] += alpha_v * w[2];
    E[2] += alpha_v * w[1]; E[6] -= alpha_v * w[1];
    E[5] -= alpha_v * w[0]; E[7] += alpha_v * w[0];
    double R0[9], R1[9];
    quat_to_rot(q + 3, R0);
    for (int i = 0; i < 3; ++i) {
        q_next[i] = q[i] + R0[3 * i] * tr[0] + R0[3 * i + 1] * tr[1] + R0[3 * i + 2] * tr[2];
        for (int j = 0; j < 3; ++j) R1[3 * i + j] = R0[3 * i] * E[j] + R0[3 * i + 1] * E[3 + j] + R0[3 * i + 2] * E[6 + j];
    }
    double qt[4];
    rot_to_quat(R1, qt);
    const double dotp = qt[0] * q[3] + qt[1] * q[4] + qt[2] * q[5] + qt[3] * q[6];
    if (dotp < 0.0)
        for (int i = 0; i < 4; ++i) qt[i] = -qt[i]; /* continuity with the previous quaternion */
    const double N2 = qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2] + qt[3] * qt[3];
    const double alpha = (3.0 - N2) / 2.0; /* firstOrderNormalize */
    for (int i = 0; i < 4; ++i) q_next[3 + i] = qt[i] * alpha;
    for (int j = 6; j < nv; ++j) q_next[j + 1] = q[j + 1] + dt * v_next[j];
    if (q_solver) {
        /* Eigen::AngleAxisd(quaternion): controller.cpp:263-265 */
        const double* u = q_next + 3;
        double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        double angle = 0.0, ax[3] = {1.0, 0.0, 0.0};
        if (n != 0.0) {
            angle = 2.0 * atan2(n, fabs(u[3]));
            if (u[3] < 0.0) n = -n;
            ax[0] = u[0] / n; ax[1] = u[1] / n; ax[2] = u[2] / n;
        }
        for (int i = 0; i < 3; ++i) q_solver[i] = q_next[i];
        for (int i = 0; i < 3; ++i) q_solver[3 + i] = angle * ax[i];
        for (int j = 6; j < nv; ++j) q_solver[j] = q_next[j + 1];
    }
}
