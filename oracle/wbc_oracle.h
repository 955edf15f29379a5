/*
 * wbc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain-C restatement of the one hot path of resibots/inria_wbc:
 *   Controller::_solve            /root/reference/src/controllers/controller.cpp:231-313
 *     tsid_->computeProblemData   controller.cpp:244   (assembly half; tsid is un-vendored)
 *     solver_->solve              controller.cpp:247   (tsid SolverHQuadProgFast + eiquadprog-fast)
 *     getActuatorForces/...       controller.cpp:250-251,260
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in third-party libraries that are
 * absent from /root/reference and un-pinned by it (docs/installation.md:37-76 clones the
 * default branches of stack-of-tasks/tsid, stack-of-tasks/eiquadprog, stack-of-tasks/pinocchio;
 * era of the only fixture, tests/ref_test_franka.yaml:1, is 2021-07: tsid ~1.6, eiquadprog ~1.2).
 * The reference holds no golden vector for x / ddq / tau / H / g, and cannot be
 * built here (no Eigen/pinocchio/tsid/eiquadprog/yaml-cpp/Boost).  This file restates the
 * PUBLISHED algorithms of those libraries (Goldfarb-Idnani dual active set as implemented in
 * eiquadprog-fast; tsid's two-level HQP stacking) and is validated by independent checks
 * (KKT residuals in numpy, scipy cross-solves), not by reference outputs.
 */
#ifndef WBC_ORACLE_H
#define WBC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* kinds of level-0 inequality blocks, in the order the task stack added them
 * (pos_tracker.cpp:161-189 iterates tasks.yaml in file order) */
enum { WBCO_INEQ_BOUNDS = 0, WBCO_INEQ_ACTUATION = 1, WBCO_INEQ_FORCE = 2 };

/* eiquadprog-fast status codes (eiquadprog-fast.hpp) */
enum {
    WBCO_EIQ_OPTIMAL = 0,
    WBCO_EIQ_INFEASIBLE = 1,
    WBCO_EIQ_UNBOUNDED = 2,
    WBCO_EIQ_MAX_ITER_REACHED = 3,
    WBCO_EIQ_REDUNDANT_EQUALITIES = 4
};
/* tsid HQP status codes consumed at controller.cpp:249,284-307 */
enum {
    WBCO_HQP_UNKNOWN = -1,
    WBCO_HQP_OPTIMAL = 0,
    WBCO_HQP_INFEASIBLE = 1,
    WBCO_HQP_UNBOUNDED = 2,
    WBCO_HQP_MAX_ITER_REACHED = 3,
    WBCO_HQP_ERROR = 4
};

/* Constant structure of one task stack (what tasks.cpp:38-404 + tasks.yaml fix at init). */
typedef struct {
    int nv, na, nc;            /* velocity dofs, actuated dofs, 6-D contacts (12 force vars each) */
    int n_dense;               /* level-1 dense motion rows (se3/com/momentum/self-collision), nv wide */
    int n_tasks;               /* number of level-1 weights carried per QP */
    const int* dense_row_task; /* [n_dense] row -> index into w */
    int n_sel;                 /* level-1 selection rows (posture): A = e_col^T */
    const int* sel_col;        /* [n_sel] column in [0,nv) */
    const int* sel_task;       /* [n_sel] row -> index into w */
    const double* forcereg_mat;/* [nc][6][12] diag(w_f)*T  (Contact6d::updateForceRegularizationTask) */
    const int* forcereg_task;  /* [nc] -> index into w (tasks.hpp:23 w_force_feet) */
    const double* force_gen;   /* [nc][6][12] T = [I3;skew(p_i)] (Contact6d::updateForceGeneratorMatrix) */
    const double* fric_mat;    /* [nc][17][12] B (Contact6d::updateForceInequalityConstraints) */
    const double* fric_lb;     /* [nc][17] */
    const double* fric_ub;     /* [nc][17] */
    int n_bound;               /* rows of the joint bounds task (tasks.cpp:274-300), A = e_col^T */
    const int* bound_col;      /* [n_bound] */
    int act_bounds;            /* 1 if an actuation-bounds task is present (tasks.cpp:303-324) */
    int n_ineq_blocks;         /* order of level-0 inequality blocks */
    const int* ineq_kind;      /* [n_ineq_blocks] WBCO_INEQ_* */
    const int* ineq_arg;       /* [n_ineq_blocks] contact index for FORCE, else 0 */
    double hessian_reg;        /* tsid DEFAULT_HESSIAN_REGULARIZATION = 1e-8 */
    int max_iter;              /* eiquadprog-fast DEFAULT_MAX_ITER = 1000 */
    /* level-1 tasks that couple the acceleration and force blocks of H (unused in the shipped stacks, registered by the
     * factory all the same):
     *  "torque" (tasks.cpp:227-271): tsid TaskActuationEquality [UPSTREAM-RECALL] -- constraint S tau = S tau_ref with
     *      S(j, joint_j) = weight(joint_j) over the mask's ones (task-actuation-equality.cpp: mask(), compute());
     *      the formulation turns it into rows S [M_a | -J_a'] with vector S tau_ref - S h_a;
     *  "cop" (tasks.cpp:156-178): tsid TaskCopEquality [UPSTREAM-RECALL] -- 3 rows over ALL contact forces (its
     *      getAssociatedContactName() is empty: computeProblemData maps it to columns nv .. nv + k), per contact point
     *      (d n' - (n.d) I) R with d = p_world - cop_ref, vector 0: n x (sum_i (p_i - c) x f_i) = 0.  The rows depend on
     *      the contact frames' placements: a per-QP input (Acop). */
    int n_acteq;               /* rows of the torque task (ones of its mask), 0: none */
    const int* acteq_joint;    /* [n_acteq] actuated joint in [0, na) */
    const double* acteq_scale; /* [n_acteq] weight-vector entry of that joint (`scaling:`), 1 without */
    int acteq_task;            /* -> index into w */
    int cop_task;              /* -> index into w, -1: no cop task */
} wbco_structure;

/* Per-QP inputs: outputs of "the step before the path" (pinocchio + task.compute()). */
typedef struct {
    const double* M;      /* [nv(nv+1)/2] joint-space inertia, packed lower triangle, row-major */
    const double* h;      /* [nv] non-linear effects */
    const double* A;      /* [n_dense][nv] dense level-1 rows */
    const double* b1;     /* [n_dense + n_sel + 6*nc + n_acteq + 3 (cop)] all level-1 rhs
                             (dense | selection | force-reg | torque: S tau_ref | cop: 0) */
    const double* Ac;     /* [nc][6][nv] contact motion-task matrices (local frame) */
    const double* bc;     /* [nc][6] contact motion-task rhs */
    const double* blb;    /* [n_bound] acceleration lower bounds */
    const double* bub;    /* [n_bound] */
    const double* tlb;    /* [na] torque lower bounds (before the -h_a shift) */
    const double* tub;    /* [na] */
    const double* w;      /* [n_tasks] level-1 task weights */
    const double* Acop;   /* [3][12 nc] rows of the cop task over the force variables (cop_task >= 0), else unused */
} wbco_inputs;

typedef struct {
    double* x;       /* [n] = [dv; f] */
    double* tau;     /* [na] */
    double* lambda;  /* [neq + nin2] multipliers in active-set order (may be NULL) */
    int* active;     /* [neq + nin2] active set, eiquadprog convention (may be NULL) */
    int n_active;
    int status;      /* WBCO_HQP_* */
    int iters;       /* active-set iterations (eiquadprog `iter`) */
    double fval;     /* objective 0.5 x'Hx + g'x */
} wbco_outputs;

/* sizes: n variables, neq equalities, nin2 ONE-SIDED inequality rows (2 x two-sided), r1 level-1 rows */
void wbco_sizes(const wbco_structure* st, int* n, int* neq, int* nin2, int* r1);

/* P1 (assembly half) + P2: dense H,g,CE,ce0,CI,ci0 exactly as SolverHQuadProgFast::solve builds them.
 * All matrices row-major: H[n*n], CE[neq*n], CI[nin2*n]. */
void wbco_assemble(const wbco_structure* st, const wbco_inputs* in,
                   double* H, double* g, double* CE, double* ce0, double* CI, double* ci0);

/* P3: eiquadprog-fast solve_quadprog.  min 0.5 x'Hx + g'x  s.t. CE x + ce0 = 0, CI x + ci0 >= 0.
 * H is overwritten?  No: inputs are const; workspace is allocated internally unless ws != NULL
 * (ws from wbco_ws_size doubles).  Returns WBCO_EIQ_*. */
long wbco_ws_size(int n, int neq, int nin2);
int wbco_eiquadprog_fast(int n, int neq, int nin2,
                         const double* H, const double* g,
                         const double* CE, const double* ce0,
                         const double* CI, const double* ci0,
                         double* x, double* u, int* A, int* iq_out, int* iter_out, double* fval,
                         int max_iter, double* ws);

/* `reps` solves of the same dense QP in one C loop (one warm-up solve first, workspace allocated once): seconds inside the loop,
 * < 0 on error.  The CPU side of the single-robot comparison (bench.py dense_seam): no Python between two solves. */
double wbco_eiquadprog_timed(int n, int neq, int nin2, const double* H, const double* g, const double* CE, const double* ce0,
                             const double* CI, const double* ci0, int max_iter, int reps, int* status_out, int* iter_out);

/* P1+P2+P3+P4 for one QP. ws may be NULL (malloc) or wbco_tick_ws_size(st) doubles. */
long wbco_tick_ws_size(const wbco_structure* st);
int wbco_tick(const wbco_structure* st, const wbco_inputs* in, wbco_outputs* out, double* ws);

/* Batched driver over contiguous [B][len] arrays, nthreads pthreads (one QP per work item,
 * in the style of qp_timer_test.cpp:55-63 / utest.hpp:62-96). Returns 0. */
typedef struct {
    const double *M, *h, *A, *b1, *Ac, *bc, *blb, *bub, *tlb, *tub, *w, *Acop;
} wbco_batch_inputs;
typedef struct {
    double *x, *tau;
    int *status, *iters;
    /* the rest of HQPOutput (may each be NULL): the final active set in eiquadprog's convention -- A[0 .. n_active): equality i
     * tagged -i-1, otherwise the index of the one-sided CI row (SolverHQuadProgFast stacks a two-sided block of m rows as
     * rows [k, k+m) = A (lower side), [k+m, k+2m) = -A (upper side)); entries beyond n_active are left untouched --
     * HQPOutput::activeSet, and getObjectiveValue (pos_tracker.hpp:44). */
    int* active;   /* [batch][neq + nin2] */
    int* n_active; /* [batch] */
    double* fval;  /* [batch] 0.5 x'Hx + g'x */
} wbco_batch_outputs;
/* reps passes over the batch on nthreads threads (work handed out from one counter, per-thread workspace, no allocation
 * per QP); returns the seconds from the moment every thread stands at the start line to the last thread's end (< 0: error).
 * Pass 0 writes `out`.  The timing loop of the CPU baseline (shape of qp_timer_test.cpp:55-63, many ticks per timer). */
double wbco_tick_batch_timed(const wbco_structure* st, int batch, const wbco_batch_inputs* in,
                             const wbco_batch_outputs* out, int nthreads, int reps);
long wbco_assemble_ws_size(const wbco_structure* st);
int wbco_tick_batch(const wbco_structure* st, int batch, const wbco_batch_inputs* in,
                    const wbco_batch_outputs* out, int nthreads);

/* After the path: what Controller::_solve does with an optimal solution,
 * src/controllers/controller.cpp:250-272:  v = dq + dt dv;  q = pinocchio::integrate(model, q, dt v);  and, for a
 * floating base, the repack of the base orientation from quaternion to angle * axis.  The model is a free-flyer root
 * (q = [p, quat(x,y,z,w)], v = [v_lin, w] in the body frame) followed by revolute joints, or revolute joints only.
 * pinocchio is not in the reference tree: its SE(3) integration (exp6, M0 * exp6, rotation -> quaternion, sign
 * continuity, first-order normalisation) and Eigen's quaternion -> angle-axis are restated from their published
 * algorithms [UPSTREAM-RECALL]; parity unpinned like the rest of this file.
 * nq = nv + 1 for a floating base, else nq = nv.  q_solver has nq - 1 (floating base) or nq entries; may be NULL. */
void wbco_integrate(int floating_base, int nv, double dt, const double* q, const double* dq, const double* dv,
                    double* q_next, double* v_next, double* q_solver);

#ifdef __cplusplus
}
#endif
#endif
