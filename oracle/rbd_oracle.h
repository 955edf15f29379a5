/*
 * rbd_oracle.h -- CPU restatement of the step BEFORE the hot path: the rigid-body terms
 * tsid's computeProblemData asks pinocchio for, and the task laws that turn them into the rows of the QP.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED.  The reference holds no golden vector for these quantities, and the rigid-body algorithms live in
 * pinocchio / tsid, which are absent from /root/reference and from this image.  What is restated:
 *   - in-tree reference code, cited by file:line in rbd_oracle.c (RobotModel::update's call set, the SE(3) task law,
 *     TaskMEquality::compute, TaskSelfCollision::compute, the task factories of tasks.cpp);
 *   - the published recursive algorithms of pinocchio 2.x (forwardKinematics, crba, nonLinearEffects, centerOfMass,
 *     jacobianCenterOfMass, ccrba, getFrameJacobian, frame classical acceleration) and tsid's TaskComEquality,
 *     TaskJointPosture, TaskJointPosVelAccBounds, Contact6d motion task  [UPSTREAM-RECALL].
 * The restatement is validated by identities that do not depend on it being a faithful copy (tests/test_oracle_rbd.py):
 * M a + nle == RNEA(q, v, a) (two different recursions), Jacobians and drifts against finite differences of the
 * forward kinematics, energy conservation of the free dynamics, Ag v == sum of body momenta, scipy rotations for log3.
 *
 * Conventions (pinocchio's): a spatial motion is (linear, angular), a force (linear, angular); the free-flyer has
 * q = [p, quat(x,y,z,w)] and v = [linear, angular] in the BODY frame; rotations are stored row-major.
 */
#ifndef RBD_ORACLE_H
#define RBD_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { WBCO_J_FREEFLYER = 0, WBCO_J_RX = 1, WBCO_J_RY = 2, WBCO_J_RZ = 3, WBCO_J_PX = 4, WBCO_J_PY = 5, WBCO_J_PZ = 6 };
enum { WBCO_T_SE3 = 0, WBCO_T_COM = 1, WBCO_T_MOMENTUM = 2, WBCO_T_SELFCOLLISION = 3 };

/* A kinematic tree as data (what pinocchio::Model holds after the URDF is parsed).  Bodies are numbered so that
 * parent[i] < i; body 0 is attached to the world (parent -1) by a free-flyer (floating_base) or by its own joint. */
typedef struct {
    int nbody, nq, nv, floating_base;
    const int* parent;             /* [nbody] */
    const int* jtype;              /* [nbody] WBCO_J_* */
    const double* placement;       /* [nbody][12] joint frame in the parent's joint frame: R row-major (9), p (3) */
    const double* inertia;         /* [nbody][10] mass, centre of mass (3), I_c: xx xy xz yy yz zz */
    double gravity[3];             /* pinocchio's default: (0, 0, -9.81) */
    int nframe;
    const int* frame_body;         /* [nframe] */
    const double* frame_placement; /* [nframe][12] */
    int na;                        /* actuated joints = the last na velocity coordinates */
    const double* q_lb;            /* [na] model().lowerPositionLimit.tail(na)  (tasks.cpp:291-292) */
    const double* q_ub;            /* [na] */
    const double* dq_max;          /* [na] model().velocityLimit.tail(na)       (tasks.cpp:287) */
} wbco_model;

/* One level-1 task that produces dense rows (order = order of addMotionTask calls = order in tasks.yaml). */
typedef struct {
    int kind;         /* WBCO_T_* */
    int frame;        /* tracked frame (SE3, self-collision) */
    int mask;         /* bit i set = row i kept (SE3: 6 bits, CoM: 3, momentum: 6); self-collision: 1 row */
    double kp, kd;    /* gains (every axis the same: tasks.cpp:56-57,106-107,139-140) */
    int ref;          /* offset of this task's reference in the per-instance reference vector */
    int av_begin, av_count; /* self-collision: avoided frames */
    double radius, margin, m;
} wbco_taskblock;

typedef struct {
    int nblock;
    const wbco_taskblock* block;
    const int* avoided_frame;  /* frame ids */
    const double* avoided_r0;
    int n_sel;                 /* posture rows: column sel_col[r] of the velocity vector */
    const int* sel_col;
    double posture_kp, posture_kd;
    int posture_ref;           /* offset of the na reference positions */
    int ncontact;
    const int* contact_frame;  /* [ncontact] */
    const double* contact_kp;  /* [ncontact] */
    const double* contact_kd;
    const int* contact_ref;    /* [ncontact] offset of the 24-number reference sample */
    int n_bound;               /* 0 or na */
    double dt;                 /* CONTROLLER.dt (tasks.cpp:283) */
    int nref;                  /* reference doubles per instance */
    /* level-1 tasks whose rows the solver side forms or that need the contact frames (wbc_oracle.h: n_acteq, cop_task) */
    int n_acteq;               /* "torque" task: n_acteq right-hand sides scale_j tau_ref_j, zero (tasks.cpp:263-265) */
    int cop;                   /* 1: a "cop" task -- 3 rows over all contact forces from the contact frames' placements, rhs 0 */
    const double* contact_points; /* [ncontact][4][3] contact points in the contact frame (tasks.cpp:353-358), used by the cop rows */
    double cop_ref[3];         /* tasks.cpp:171: (0, 0, 0) */
} wbco_taskmap;

/* Reference layouts inside the per-instance reference vector (tsid TrajectorySample):
 *   SE3:       pos 12 = translation (3) + rotation column-major (9)  (tsid SE3ToVector; src/trajs/loader.cpp:11-53),
 *              vel 6, acc 6  (world-oriented)                                              -> 24 doubles
 *   CoM:       pos 3, vel 3, acc 3                                                          ->  9
 *   momentum:  reference momentum 6, its derivative 6 (TrajectorySample vel / acc)          -> 12
 *   posture:   na reference positions (vel = acc = 0: tasks.cpp:217 sets the value only)    -> na
 *   contact:   like SE3: placement 12 (tasks.cpp:361-362), vel 6, acc 6 (zero unless a behaviour
 *              moves the contact, pos_tracker.cpp:240-244)                                  -> 24 */

typedef struct {
    double* M;     /* [nv][nv] */
    double* nle;   /* [nv] */
    double* com;   /* [3] */
    double* vcom;  /* [3] */
    double* acom;  /* [3] com acceleration with ddq = 0, no gravity */
    double* Jcom;  /* [3][nv] */
    double* Ag;    /* [6][nv] centroidal momentum matrix */
    double* dAgv;  /* [6] */
    double* oMf;   /* [nframe][12] */
    double* vf;    /* [nframe][6] local */
    double* af;    /* [nframe][6] classical acceleration, local, ddq = 0 */
    double* Jl;    /* [nframe][6][nv] local */
    double* Jw;    /* [nframe][6][nv] pinocchio WORLD */
} wbco_terms;

/* Any pointer of `out` may be NULL. */
void wbco_rbd_terms(const wbco_model* model, const double* q, const double* v, wbco_terms* out);
/* Recursive Newton-Euler, tau = M(q) a + nle(q, v): a second recursion to check crba + nonLinearEffects against. */
void wbco_rnea(const wbco_model* model, const double* q, const double* v, const double* a, double* tau);
/* kinetic + potential energy (for the conservation check) */
double wbco_energy(const wbco_model* model, const double* q, const double* v);
void wbco_log3(const double* R_rowmajor, double* w);

/* The rows of one instance in the record layout of include/wbcqp.h (wbcqp_inputs): M packed lower triangle, h,
 * A [n_dense][nv], b1 [n_dense + n_sel + 6 ncontact + n_acteq + 3 cop] (force-regularisation, torque and cop entries zero),
 * Ac [ncontact][6][nv], bc [ncontact][6], blb / bub [n_bound], Acop [3][12 ncontact] (map->cop; may be NULL otherwise):
 * tsid TaskCopEquality::compute [UPSTREAM-RECALL], per contact point (d n' - (n.d) I) R with d = oMf.act(p_i) - cop_ref, n = e_z. */
void wbco_task_rows(const wbco_model* model, const wbco_taskmap* map, const double* q, const double* v, const double* ref,
                    double* M, double* h, double* A, double* b1, double* Ac, double* bc, double* blb, double* bub, double* Acop);
void wbco_task_rows_batch(const wbco_model* model, const wbco_taskmap* map, int batch, int n_threads, int n_dense,
                          const double* q, const double* v, const double* ref,
                          double* M, double* h, double* A, double* b1, double* Ac, double* bc, double* blb, double* bub, double* Acop);

#ifdef __cplusplus
}
#endif
#endif
