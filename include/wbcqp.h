/*
 * wbcqp.h -- C ABI of the MI355X batched whole-body QP ("one control tick of inria_wbc, B robots at once").
 *
 * This is the drop-in boundary for the single hot path of resibots/inria_wbc
 * (citations are into /root/reference/):
 *
 *   Controller::_solve                         src/controllers/controller.cpp:231-313
 *     tsid_->computeProblemData(t, q, dq)      src/controllers/controller.cpp:244   (assembly half: HQPData)
 *     solver_->solve(HQPData)                  src/controllers/controller.cpp:247   (tsid SolverHQuadProgFast
 *                                                                                    + eiquadprog-fast GI solve)
 *     getActuatorForces / getAccelerations     src/controllers/controller.cpp:250-251
 *     getContactForces                         src/controllers/controller.cpp:260
 *
 * One "QP" below = that sequence for one robot instance.  Inputs are what the step before the path
 * produces (pinocchio terms + each task's compute(): M, h, task rows, contact Jacobians, bounds);
 * outputs are x = [dv; f], tau, status, active-set iterations.
 *
 * Plain pointers and sizes only; no C++/torch types.  Nothing throws across this boundary: every
 * entry point returns a wbcqp_status and wbcqp_last_error() gives the text.  A handle is bound to one
 * HIP device and is not thread-safe: one host thread at a time calls into it; distinct handles may be
 * used concurrently (the reference's Controller is single-threaded and non-copyable,
 * controller.hpp:50-51).  That one thread may launch on SEVERAL HIP streams: the handle keeps its
 * launch-order buffer and its queue counter per stream (up to 16 distinct streams; a launch on a 17th runs
 * in index order on the hardware's dispatcher and keeps no state), so launches in flight on two streams
 * share nothing -- tests/test_gpu_streams.py interleaves two streams on one handle and compares bit for
 * bit.  A stream's state is keyed by the stream handle's value: destroy a stream only after its launches
 * have completed.  wbcqp_rollout waits for the previous roll-out of the same handle (on whatever stream it
 * ran) before it reuses the handle's roll-out buffers.  The device-pointer solve entry points allocate
 * nothing in the steady state (upstream's solver runs under EIGEN_MALLOC_NOT_ALLOWED); the FIRST launch
 * on a stream of a larger batch than any before it on that stream (launch-order buffer, 8 bytes per QP,
 * with one synchronisation of that stream) and the first launch on a new stream (an 8-byte queue counter)
 * do allocate.  A captured tick (wbcqp_tick_graph_create) owns its buffers and never does.  The first solve of a slot after
 * wbcqp_set_structure also synchronises its stream once (the force blocks' factor cache, interface history 151).
 */
#ifndef WBCQP_H
#define WBCQP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Interface history (WBCQP_VERSION = 100 major + 10 minor + patch; round 5 changed no declaration of this header -- wbcqp_layout.waves_per_cu may now be 3,
 * and wbcqp_structure.max_iter no longer decides whether a shipped stack runs its own instantiation):
 *   151  no declaration changed.  wbcqp_tick_host fills wbcqp_outputs.active_mask when given; the numbering of active_mask's bits is documented (below);
 *        a slot's FIRST solve on the compact layout makes the force blocks' factor for the force-regularisation weights of its first QP (one small kernel
 *        and one synchronisation of the launch's stream, once per wbcqp_set_structure; never inside a stream capture) -- later QPs that carry those weights
 *        take the factor from there, others compute it as before: the same bits either way (env WBCQP_DEBUG_NO_FFCACHE=1, read at wbcqp_create, turns it off)
 *   150  launch-order state per (handle, stream), active_mask written by every kernel, torque / cop task rows
 *        (wbcqp_structure.n_acteq, cop_*), posture mask
 *   140  wbcqp_rollout, wbcqp_outputs.active_mask (WBCQP_FLAG_WARM_START), wbcqp_state.momentum, wbcqp_layout.wave_per_qp
 *        (WBCQP_FLAG_WORKGROUP_PER_QP)
 *   130  wbcqp_integrate, wbcqp_set_model / wbcqp_problem_data / wbcqp_tick and companions
 *   121  queue + packed launch order, wbcqp_launch_order */
#define WBCQP_VERSION 151
#define WBCQP_MAX_STRUCTURES 16
#define WBCQP_MAX_INEQ_BLOCKS 16
#define WBCQP_MAX_VARS 126 /* n = nv + 12*nc: every per-QP vector fits one 128-entry LDS slot, n + 2 <= 128 */

/* ---- return codes of the API itself ---- */
typedef enum {
    WBCQP_OK = 0,
    WBCQP_ERR_INVALID = 1,     /* bad argument / inconsistent structure */
    WBCQP_ERR_HIP = 2,         /* HIP runtime error (text in wbcqp_last_error) */
    WBCQP_ERR_UNSUPPORTED = 3, /* size outside what the kernels support */
    WBCQP_ERR_NO_DEVICE = 4,   /* no gfx950 device: there is NO CPU fallback */
    WBCQP_ERR_RCCL = 5
} wbcqp_status;

/* ---- per-QP solver status: the tsid HQP status values the reference switches on
 *      (controller.cpp:249, 284-307) ---- */
typedef enum {
    WBCQP_HQP_UNKNOWN = -1,
    WBCQP_HQP_OPTIMAL = 0,
    WBCQP_HQP_INFEASIBLE = 1,
    WBCQP_HQP_UNBOUNDED = 2,
    WBCQP_HQP_MAX_ITER_REACHED = 3,
    WBCQP_HQP_ERROR = 4 /* redundant equalities */
} wbcqp_hqp_status;

/* kinds of level-0 inequality blocks, in the order the task stack added them
 * (pos_tracker.cpp:161-189 walks tasks.yaml in file order) */
typedef enum {
    WBCQP_INEQ_BOUNDS = 0,    /* "bounds"           tasks.cpp:274-300  rows = +-e_col          */
    WBCQP_INEQ_ACTUATION = 1, /* "actuation-bounds" tasks.cpp:303-324  rows = +-[M_a | -J_a']  */
    WBCQP_INEQ_FORCE = 2      /* contact friction   tasks.cpp:329-368  17 x 12 per contact     */
} wbcqp_ineq_kind;

typedef enum {
    WBCQP_F64 = 0, /* inputs/outputs double (the reference's Eigen double) */
    WBCQP_F32 = 1  /* inputs/outputs float at the boundary; the solve still runs in f64 */
} wbcqp_dtype;

/*
 * Constant structure of one task stack = what PosTracker::parse_tasks + tasks.cpp factories fix at
 * construction time.  Variables x = [dv (nv); f (12 per contact)], n = nv + 12 nc.
 * All arrays are HOST pointers, copied by wbcqp_set_structure.
 */
typedef struct {
    int32_t nv, na, nc;
    /* level 1 (cost): H = sum_t w_t A_t' A_t + hessian_reg I, g = -sum_t w_t A_t' b_t */
    int32_t n_dense;               /* dense motion rows (se3 / com / momentum / self-collision), nv wide   */
    int32_t n_tasks;               /* length of the per-QP weight vector w                                 */
    const int32_t* dense_row_task; /* [n_dense] row -> index into w; rows of one task are consecutive      */
    int32_t n_sel;                 /* selection rows (posture task, tasks.cpp:181-224): A = e_col'          */
    const int32_t* sel_col;        /* [n_sel] column in [0, nv)                                            */
    const int32_t* sel_task;       /* [n_sel] row -> index into w                                          */
    const double* forcereg_mat;    /* [nc][6][12] diag(w_f) T   (Contact6d force regularisation task)      */
    const int32_t* forcereg_task;  /* [nc] -> index into w      (weight w_force_feet, tasks.hpp:23)        */
    /* level 0 (hard constraints) */
    const double* force_gen;       /* [nc][6][12] T: contact-point forces -> 6-D wrench; Jc = T' A_c       */
    const double* fric_mat;        /* [nc][17][12] friction pyramid + normal force rows                    */
    const double* fric_lb;         /* [nc][17] */
    const double* fric_ub;         /* [nc][17] */
    int32_t n_bound;               /* rows of the joint bounds task                                        */
    const int32_t* bound_col;      /* [n_bound] column in [0, nv)                                          */
    int32_t act_bounds;            /* 1 if an actuation-bounds task exists (na rows)                       */
    int32_t n_ineq_blocks;
    const int32_t* ineq_kind;      /* [n_ineq_blocks] wbcqp_ineq_kind                                      */
    const int32_t* ineq_arg;       /* [n_ineq_blocks] contact index for WBCQP_INEQ_FORCE                   */
    double hessian_reg;            /* tsid default 1e-8                                                    */
    int32_t max_iter;              /* eiquadprog-fast default 1000                                         */
    /* Two level-1 task types of the reference's factory that no shipped stack uses.  Each couples blocks of H that every other
     * task keeps apart (dv with f, one contact's forces with another's), so a stack that has one runs the full LDS layout with H
     * assembled and factored as ONE n x n matrix (wbcqp_layout.dense_h; n <= 80) instead of a dv block and a 12 x 12 block per
     * contact.  CE, CI and the decode are what they are without them: level-1 tasks only enter H and g.
     *  "torque" (tasks.cpp:227-271, tsid TaskActuationEquality): n_acteq rows, one per 1 of the task's mask,
     *      scale_j [M_a(joint_j, :) | -J_a(:, joint_j)'] x = scale_j tau_ref_j - scale_j h_a(joint_j)   (SURVEY A.1 step 6);
     *      scale = the task's weight vector (`scaling:`, tasks.cpp:256-261), tau_ref = 0 in the reference (tasks.cpp:263-265); the
     *      products scale_j tau_ref_j are the task's entries of wbcqp_inputs.b1.
     *  "cop" (tasks.cpp:156-178, tsid TaskCopEquality): 3 rows over ALL 12 nc force variables (the task is associated with no
     *      single contact), given per QP in wbcqp_inputs.Acop -- they depend on the contact frames' placements; right-hand side in b1. */
    int32_t n_acteq;               /* rows of the torque task; 0: none                                     */
    const int32_t* acteq_joint;    /* [n_acteq] actuated joint in [0, na), ascending                       */
    const double* acteq_scale;     /* [n_acteq]                                                            */
    int32_t acteq_task;            /* -> index into w                                                      */
    int32_t cop_task;              /* -> index into w, a task of its own (shared with no other row); -1: no cop task.  A zero-initialised structure must set it
                                      to -1: 0 declares a cop task on task 0 -- wbcqp_set_structure refuses that when task 0 has other rows, as it has in every stack */
} wbcqp_structure;

/* Sizes derived from a structure (what PosTracker prints under `verbose`, pos_tracker.cpp:150-158). */
typedef struct {
    int32_t n;    /* tsid nVar */
    int32_t neq;  /* tsid nEq  = (nv - na) + 6 nc */
    int32_t nin;  /* tsid nIn  (two-sided rows) */
    int32_t nin2; /* one-sided rows of eiquadprog's CI = 2 nin */
    int32_t r1;   /* level-1 rows = n_dense + n_sel + 6 nc + n_acteq + (3 if a cop task) */
    /* element counts of the per-QP input arrays below */
    int32_t len_M, len_h, len_A, len_b1, len_Ac, len_bc, len_blb, len_bub, len_tlb, len_tub, len_w;
    int32_t lds_bytes;         /* dynamic LDS one QP (one 256-thread workgroup) needs */
    int32_t waves_per_cu;      /* resident QPs (workgroups) per CU a launch of THIS structure alone runs with by default: 1 (full layout), 2 (compact
                                  layout: the kernels' registers admit two), or 3 -- compact layout, NO actuation bounds, 40 KB <= lds_bytes <= 54592:
                                  such launches take the queue kernel compiled for three waves per SIMD (iCub on one or two feet).  A launch falls
                                  back to two per CU under WBCQP_FLAG_WARM_START, WBCQP_FLAG_HW_DISPATCH, env WBCQP_DEBUG_LDS_PAD, or when it is ragged
                                  and any of its groups has actuation bounds; env WBCQP_DEBUG_LAUNCH=1 prints the residency of every launch */
    int64_t algorithmic_bytes; /* compact in+out bytes per QP at WBCQP_F64 */
    int32_t wave_per_qp;       /* 1: the structure runs one WAVEFRONT per QP (n <= 16, fixed base, no contacts, bounds only: Franka, Tiago),
                                  four QPs per workgroup, 7.7 KB of LDS per QP; lds_bytes / waves_per_cu then describe the four-wave
                                  kernel that WBCQP_FLAG_WORKGROUP_PER_QP selects */
    int32_t dense_h;           /* 1: the stack has a torque or a cop task -- full layout, H factored as one n x n matrix (never compact) */
    int32_t len_Acop;          /* element count of wbcqp_inputs.Acop: 3 * 12 nc with a cop task, else 0 */
    int32_t specialised;       /* > 0: the library holds an instantiation of the compact kernel with THIS layout's sizes and offsets as literals
                                  (the shipped stacks: 1 Talos, 2 iCub, 3 Talos in single support); a launch of one such group runs it unless
                                  WBCQP_FLAG_GENERIC_KERNEL.  Bit for bit the generic kernel's results, about 10 % less time per QP */
} wbcqp_layout;

/*
 * Per-QP inputs, each a contiguous [batch][len] row-major array of the handle's dtype.
 * One workgroup reads one QP's rows, so consecutive lanes read consecutive addresses.
 */
typedef struct {
    const void* M;   /* [batch][nv(nv+1)/2] inertia matrix, packed lower triangle (i>=j at i(i+1)/2+j)      */
    const void* h;   /* [batch][nv]  non-linear effects                                                     */
    const void* A;   /* [batch][n_dense][nv] dense level-1 task rows (TaskSE3Equality & co: ex_task.cpp:175-247,
                        task-momentum-equality.cpp:144-173, task-self-collision.cpp:84-203)                   */
    const void* b1;  /* [batch][r1] level-1 right-hand sides: dense | selection | force-regularisation
                        | torque task (scale_j tau_ref_j, n_acteq) | cop task (3)                             */
    const void* Ac;  /* [batch][nc][6][nv] contact motion-task matrices (local frame)                       */
    const void* bc;  /* [batch][nc][6]     contact motion-task right-hand sides                             */
    const void* blb; /* [batch][n_bound]   acceleration bounds                                              */
    const void* bub; /* [batch][n_bound]                                                                    */
    const void* tlb; /* [batch][na]        torque bounds, before the -h_a shift (NULL if !act_bounds)       */
    const void* tub; /* [batch][na]                                                                         */
    const void* w;   /* [batch][n_tasks]   level-1 task weights (PosTracker::update_task_weights)           */
    const void* Acop;/* [batch][3][12 nc]  rows of the cop task over the force variables (NULL without one)      */
} wbcqp_inputs;

typedef struct {
    void* x;          /* [batch][n]  = [dv; f]        (getAccelerations / getContactForces)  */
    void* tau;        /* [batch][na] actuator torques (getActuatorForces)                    */
    int32_t* status;  /* [batch] wbcqp_hqp_status                                            */
    int32_t* iters;   /* [batch] active-set iterations (eiquadprog `iter`)                   */
    void* objective;  /* [batch] 0.5x'Hx + g'x (SolverHQPBase::getObjectiveValue), may be NULL */
    int32_t* n_active;/* [batch] size of the final active set incl. equalities, may be NULL  */
    uint32_t* active_mask; /* [batch][8], may be NULL.  OUT: bit r of the 256-bit mask (bit r % 32 of word r / 32) = one-sided inequality row r is
                         active at the solution (HQPOutput::activeSet's inequality entries).  r IS eiquadprog's CI row index, i.e. the row
                         SolverHQuadProgFast::solve stacks: the level-0 inequality blocks in the order of wbcqp_structure.ineq_kind / ineq_arg
                         (= the order tasks.yaml added them, pos_tracker.cpp:161-189), each two-sided block of m rows lb <= A x <= ub as
                         rows [k, k + m) = +A with ci0 = -lb (the LOWER side) followed by rows [k + m, k + 2m) = -A with ci0 = ub (the UPPER
                         side); m = n_bound for the bounds block (row j: joint bound_col[j]), na for the actuation bounds (row j: actuated
                         joint j, limits shifted by -h_a), 17 for a contact's force block (rows 0-15 the friction pyramid of the four points,
                         row 16 the normal-force sum).  Talos: bounds 0-43 / 44-87, torque 88-131 / 132-175, first contact 176-192 / 193-209,
                         second contact 210-226 / 227-243.  Equalities (always active; eiquadprog tags them -i-1) have no bit: popcount =
                         n_active - nEq.  tests/util.py:assert_parity compares the mask with the oracle's active list bit for bit on every
                         parity case.  With WBCQP_FLAG_WARM_START also IN: the mask a previous tick left for the same instance (zeros: no
                         hint).  Device pointer on the device entry points.  Written by every kernel (rows beyond the 256th have no bit;
                         all zero where the status is not OPTIMAL); READ as the hint by the compact-layout kernel only -- structures on
                         the one-wavefront-per-QP kernel (wbcqp_layout.wave_per_qp) and on the full layout ignore the hint. */
} wbcqp_outputs;

/* wbcqp_desc.flags */
#define WBCQP_FLAG_INDEX_ORDER 1 /* launch the QPs of a batch in index order.  Default (0): longest-first -- a launch is
                                    ordered by the active-set iteration counts of the previous launch of the same shape on
                                    the same stream (control ticks change little); results do not depend on the order */
#define WBCQP_FLAG_HW_DISPATCH 2 /* one workgroup per QP, handed to the CUs by the hardware's dispatcher (the default for
                                    structures whose workgroup is under 48 KB of LDS: Franka, Tiago).  Default (0) for the
                                    humanoid stacks: as many workgroups as the chip holds (two per CU on the compact layout)
                                    take QPs from a queue in the launch order (the dispatcher binds workgroup i to one shader
                                    engine of XCD i % 8 and waits for it; the queue does not); results do not depend on it */
#define WBCQP_FLAG_QUEUE 8       /* the queue also for small structures, and the bin-packed order also where two workgroups share
                                    a CU (measured no better than longest-first there: tools/dispatch_sweep.py) */
#define WBCQP_FLAG_NO_PACKING 4  /* keep the plain longest-first order for the queue.  Default (0): where one workgroup fills a
                                    CU and a launch holds between one and eight QPs per resident workgroup, the order is
                                    bin-packed from the predicted costs (setup + iterations of the previous launch) */
#define WBCQP_FLAG_FULL_LDS 16    /* keep every structure on the layout that holds M, Jc and A_c in LDS for the QP's whole life
                                    (Talos: one QP per CU).  Default (0): structures within n <= 80, nEq <= 22, nv <= 52, two
                                    contacts (every stack the reference ships) use the compact layout -- half the LDS, two
                                    QPs resident per CU; same algorithm, results agree to rounding */

#define WBCQP_FLAG_WORKGROUP_PER_QP 32 /* keep the small structures (n <= 16 without contacts: Franka, Tiago) on the four-wave kernels too.
                                     Default (0): one WAVEFRONT per QP for them, four QPs per 256-thread workgroup and no workgroup
                                     barrier (csrc/wbcqp_small.hpp); in a ragged launch they go out as a launch of their own */
#define WBCQP_FLAG_WARM_START 64 /* OPT-IN, not what the reference does: among the violated constraints the active-set loop first picks
                                    those that were active at the previous tick's solution (wbcqp_outputs.active_mask, in/out), the most
                                    violated of them first; when none of them is violated, eiquadprog's rule (the most violated row).
                                    Still Goldfarb-Idnani -- any violated constraint is a legal pick, the QP is strictly convex, so the
                                    EXACT solution is the cold start's -- but eiquadprog's loop ends on |sum min(s, 0)| <= nIneq eps
                                    tr(H) tr(J) 100 (about 0.3 for the humanoid stacks), so x and tau equal the cold start's only up to
                                    that termination tolerance: the two pick orders may end on different iterates (measured on the
                                    squat stream: 2-4 of 1024 QPs per tick, |dx| up to 4e-4 relative; bench.py's warm_start.parity
                                    counts them, tests/test_gpu_warm.py bounds them).  What it buys: the add / drop churn of a cold
                                    start is avoided, iteration counts fall to about the number of constraints active at the solution.
                                    eiquadprog-fast has no such thing: bench.py reports it BESIDE the headline, never as it */
#define WBCQP_FLAG_GENERIC_KERNEL 128 /* never the shipped stacks' own instantiations of the compact kernel (wbcqp_layout.specialised): the generic
                                     kernel for every structure (what the specialised ones are tested against, bit for bit) */
#define WBCQP_FLAG_REFRESH_SHIFT 8
#define WBCQP_FLAG_REFRESH(n) (((n) & 0xff) << WBCQP_FLAG_REFRESH_SHIFT) /* renew the launch order every n-th launch of a shape
                                    (1: after every launch; 0: the default, 4).  Between renewals the same order is used:
                                    iteration counts drift slowly between control ticks and the queue absorbs the drift.
                                    A captured tick (wbcqp_tick_graph_create) renews it on every replay */

typedef struct {
    int32_t device;   /* HIP device ordinal */
    int32_t dtype;    /* wbcqp_dtype of every input/output array */
    int32_t flags;    /* WBCQP_FLAG_* */
} wbcqp_desc;

/* one homogeneous group of a ragged (mixed-robot) batch */
typedef struct {
    int32_t slot;     /* structure slot */
    int32_t batch;
    wbcqp_inputs in;  /* DEVICE pointers */
    wbcqp_outputs out;
} wbcqp_group;

typedef struct wbcqp_handle wbcqp_handle;

int wbcqp_version(void);
/* text of the last error on this handle (or of the last failed wbcqp_create if handle == NULL) */
const char* wbcqp_last_error(const wbcqp_handle* handle);

/* Binds to a gfx950 device. Fails with WBCQP_ERR_NO_DEVICE when none is present (no CPU path). */
int wbcqp_create(const wbcqp_desc* desc, wbcqp_handle** out);
int wbcqp_destroy(wbcqp_handle* handle);

/* Uploads a task stack into `slot` (0..WBCQP_MAX_STRUCTURES-1): the analogue of
 * solver_->resize(nVar, nEq, nIn) at pos_tracker.cpp:102 plus the constant blocks. */
int wbcqp_set_structure(wbcqp_handle* handle, int slot, const wbcqp_structure* st);
/* Pure host computation; handle may be NULL (usable without a GPU). */
int wbcqp_layout_of(const wbcqp_structure* st, wbcqp_layout* out);

/* Solve `batch` QPs of one structure. All pointers are DEVICE pointers on the handle's device;
 * the launch is asynchronous on `stream` (a hipStream_t, NULL = default stream). */
int wbcqp_solve_batch(wbcqp_handle* handle, int slot, int batch,
                      const wbcqp_inputs* in, const wbcqp_outputs* out, void* stream);
/* Same with HOST pointers: stages through device buffers owned by the handle, blocks until done. */
int wbcqp_solve_batch_host(wbcqp_handle* handle, int slot, int batch,
                           const wbcqp_inputs* in, const wbcqp_outputs* out);
/* Mixed-robot batch: one launch over several homogeneous groups (per-QP n differs between groups). */
int wbcqp_solve_ragged(wbcqp_handle* handle, int n_groups, const wbcqp_group* groups, void* stream);

/* ---- The narrow seam: one dense QP the way tsid's solver hands it to eiquadprog --------------------------
 * What stands behind `solver_->resize(nVar, nEq, nIn)` (pos_tracker.cpp:102) and `const HQPOutput& solver_->solve(const
 * HQPData&)` (controller.cpp:247) once SolverHQuadProgFast has stacked the HQPData:
 *      min 1/2 x'Hx + g'x   s.t.  CE x + ce0 = 0,  CI x + ci0 >= 0
 * H [n][n] (symmetric, the lower triangle is read), g [n], CE [neq][n], ce0 [neq], CI [nin][n], ci0 [nin], all row-major,
 * nin = the rows eiquadprog sees (tsid stacks every two-sided row twice: [A; -A], ci0 = [-lb; ub]).  No structure is
 * assumed and none is needed: this is the compatibility path for a caller that owns its HQPData (INTEGRATION.md section
 * 3 shows the tsid SolverHQPBase subclass); wbcqp_solve_batch on a task stack is the fast path.  n <= 96 (J, R and the vectors of one QP must fit one CU's 160 KiB of LDS), nin <= 512.
 * Status values are tsid's (wbcqp_hqp_status: eiquadprog UNBOUNDED -> INFEASIBLE, REDUNDANT_EQUALITIES -> ERROR). */
typedef struct {
    const void *H, *g, *CE, *ce0, *CI, *ci0; /* [batch][...] of the handle's dtype; CE / ce0 may be NULL when neq == 0, CI / ci0 when nin == 0 */
} wbcqp_dense_inputs;

/* DEVICE pointers, asynchronous on `stream`; out->tau is not written (a dense QP has no actuation model), out->x is [batch][n]. */
int wbcqp_solve_dense(wbcqp_handle* handle, int batch, int n, int neq, int nin, int max_iter,
                      const wbcqp_dense_inputs* in, const wbcqp_outputs* out, void* stream);

/* HQPOutput of the reference: owned by the solver, valid until the next call on the same handle (the reference returns a
 * reference to solver-owned storage, controller.cpp:247). */
typedef struct {
    int32_t batch, n;
    const double* x;         /* [batch][n] */
    const int32_t* status;   /* [batch] wbcqp_hqp_status */
    const int32_t* iters;    /* [batch] */
    const double* objective; /* [batch] SolverHQPBase::getObjectiveValue (pos_tracker.hpp:44) */
    const int32_t* n_active; /* [batch] */
} wbcqp_dense_output;

/* HOST pointers to double arrays (whatever the handle's dtype: the reference's Eigen matrices are double), blocks until
 * done; *result points into storage owned by the handle. */
int wbcqp_solve_dense_host(wbcqp_handle* handle, int batch, int n, int neq, int nin, int max_iter,
                           const wbcqp_dense_inputs* in, const wbcqp_dense_output** result);

/* Optional exchange step: all-gather joint torques of a batch sharded over ranks.
 * `comm` is an ncclComm_t created by the caller (RCCL). send: [count] elements, recv: [nranks*count]. */
int wbcqp_allgather_tau(wbcqp_handle* handle, void* comm, const void* send, void* recv,
                        size_t count, void* stream);

/* After the path -- what Controller::_solve does with an optimal solution, controller.cpp:250-272:
 *   v_next = dq + dt dv;  q_next = pinocchio::integrate(model, q, dt v_next);  q_solver = q_next with the base orientation
 *   repacked from quaternion to angle * axis (floating base) or q_next itself.
 * Model: free-flyer root (q = [p, quat(x,y,z,w)], v = [v_lin, w] in the body frame, nq = nv + 1) followed by revolute
 * joints when floating_base != 0, revolute joints only otherwise (nq = nv).  x is the solver output [batch][ldx], dv its
 * first nv entries.  status may be NULL; an instance whose status is not WBCQP_HQP_OPTIMAL keeps its state (the
 * reference throws there, controller.cpp:284-307).  q_solver ([batch][nv]) may be NULL.  DEVICE pointers of the handle's
 * dtype, asynchronous on `stream`. */
int wbcqp_integrate(wbcqp_handle* handle, int batch, int nv, int floating_base, double dt,
                    const void* q, const void* dq, const void* x, int ldx, const int32_t* status,
                    void* q_next, void* v_next, void* q_solver, void* stream);

/* Same with HOST pointers: stages through device buffers owned by the handle, blocks until done. */
int wbcqp_integrate_host(wbcqp_handle* handle, int batch, int nv, int floating_base, double dt,
                         const void* q, const void* dq, const void* x, int ldx, const int32_t* status,
                         void* q_next, void* v_next, void* q_solver);

/* ---- Before the path: from the robot state to the rows of the QP -------------------------
 * What the upstream half of tsid_->computeProblemData (controller.cpp:244) produces: pinocchio's rigid-body terms (call
 * set of RobotModel::update, src/utils/robot_model.cpp:83-113) and every task's compute() (example_project/src/tsid/
 * ex_task.cpp:175-247, src/tsid/task-momentum-equality.cpp:144-173, src/tsid/task-self-collision.cpp:84-203, gains and
 * masks of src/controllers/tasks.cpp:38-404).  The kinematic tree and the task bindings are DATA (no URDF / YAML parser
 * behind this boundary). */
typedef enum { WBCQP_J_FREEFLYER = 0, WBCQP_J_RX = 1, WBCQP_J_RY = 2, WBCQP_J_RZ = 3, WBCQP_J_PX = 4, WBCQP_J_PY = 5, WBCQP_J_PZ = 6 } wbcqp_joint_type;
typedef enum { WBCQP_T_SE3 = 0, WBCQP_T_COM = 1, WBCQP_T_MOMENTUM = 2, WBCQP_T_SELFCOLLISION = 3 } wbcqp_task_kind;

/* A kinematic tree (what pinocchio::Model holds).  Bodies are numbered depth-first (parent[i] < i and every subtree is a
 * contiguous index range, as pinocchio numbers joints); body 0 hangs on the world by a free-flyer (floating_base: q =
 * [p, quat(x,y,z,w)], v = [linear, angular] in the body frame) or by its own joint.  nq = nbody + 6 / nbody,
 * nv = nbody + 5 / nbody.  Placements are 12 doubles: rotation row-major (9), translation (3).  HOST pointers. */
typedef struct {
    int32_t nbody, floating_base;
    const int32_t* parent;         /* [nbody], -1 for body 0 */
    const int32_t* jtype;          /* [nbody] wbcqp_joint_type */
    const double* placement;       /* [nbody][12] joint frame in the parent's joint frame */
    const double* inertia;         /* [nbody][10] mass, centre of mass (3), inertia at the com: xx xy xz yy yz zz */
    double gravity[3];             /* pinocchio's default (0, 0, -9.81) */
    int32_t nframe;
    const int32_t* frame_body;     /* [nframe] */
    const double* frame_placement; /* [nframe][12] frame in its body's joint frame */
    const double* q_lb;            /* [na] position limits of the actuated joints (tasks.cpp:291-292) */
    const double* q_ub;            /* [na] */
    const double* dq_max;          /* [na] velocity limits (tasks.cpp:287); acceleration limit = dq_max / dt (:288) */
} wbcqp_model;

/* One level-1 task that yields dense rows, in the order of the addMotionTask calls (= file order of tasks.yaml, with the
 * self-collision tasks last as in the shipped stacks): its rows are consecutive in wbcqp_inputs.A / b1. */
typedef struct {
    int32_t kind;    /* wbcqp_task_kind */
    int32_t frame;   /* tracked frame (SE3, self-collision) */
    int32_t mask;    /* bit i = row i kept (tasks.cpp:27-35 convert_mask: character i of the yaml string) */
    double kp, kd;
    int32_t ref;     /* offset of the task's reference inside one instance's reference vector:
                        SE3: placement 12 (translation, rotation COLUMN-major: tsid SE3ToVector, src/trajs/loader.cpp:11-53),
                             velocity 6, acceleration 6 (world-oriented);  CoM: pos 3, vel 3, acc 3;
                             momentum: reference 6, its derivative 6;  self-collision: none */
    int32_t n_avoided;              /* self-collision */
    const int32_t* avoided_frame;   /* [n_avoided] */
    const double* avoided_r0;       /* [n_avoided] */
    double radius, margin, m;       /* self-collision (tasks.cpp:380-383) */
} wbcqp_task;

typedef struct {
    int32_t n_task;
    const wbcqp_task* task;
    double posture_kp, posture_kd;  /* the structure's selection rows (tasks.cpp:181-224) */
    int32_t posture_ref;            /* offset of the na reference positions */
    int32_t n_contact;              /* must equal the structure's nc */
    const int32_t* contact_frame;   /* [n_contact] */
    const double* contact_kp;       /* [n_contact] (tasks.cpp:359-360) */
    const double* contact_kd;
    const int32_t* contact_ref;     /* [n_contact] offset of the contact motion task's reference, 24 numbers like an SE3 task:
                                       placement 12 (tasks.cpp:361-362 sets it), velocity 6, acceleration 6 (zero unless a
                                       behaviour moves the contact: PosTracker::set_contact_se3_ref(sample), pos_tracker.cpp:240-244) */
    int32_t bounds;                 /* 1: the structure's n_bound = na joint-bounds rows are computed (tasks.cpp:274-300) */
    double dt;                      /* CONTROLLER.dt */
    int32_t nref;                   /* reference doubles per instance */
} wbcqp_taskmap;

typedef struct {
    const void* q;    /* [batch][nq] */
    const void* v;    /* [batch][nv] */
    const void* ref;  /* [batch][nref] */
    void* momentum;   /* OUT, may be NULL: [batch][6] centroidal momentum Ag(q) v of this state -- linear (3), then angular about the CoM
                         (3): what Controller::_solve keeps as momentum_ = momentumJacobian(data).bottomRows(3) * dq
                         (controller.cpp:245) is entries 3..5.  The rows kernel forms it anyway (momentum task, CoM velocity). */
} wbcqp_state;

/* Binds a tree and its task bindings to a slot that already holds the matching structure (same nv, na, nc, n_dense,
 * n_sel, n_bound).  A later wbcqp_set_structure on the slot drops the model with the old structure: bind it again (a
 * contact added or removed changes both, pos_tracker.cpp:246-263). */
int wbcqp_set_model(wbcqp_handle* handle, int slot, const wbcqp_model* model, const wbcqp_taskmap* map);
/* The checks of wbcqp_set_model without a device (pure host computation, like wbcqp_layout_of): is this tree + these task
 * bindings a valid companion of that structure, and how much LDS does one instance of the rows kernel need.  Error text through
 * wbcqp_last_error(NULL). */
int wbcqp_check_model(const wbcqp_structure* st, const wbcqp_model* model, const wbcqp_taskmap* map, int32_t* lds_bytes);
/* Writes the M, h, A, b1, Ac, bc, blb, bub arrays of `rows` (DEVICE pointers, the layout wbcqp_solve_batch reads; tlb, tub
 * and w are not touched: constant limits and weights) for `batch` instances.  Asynchronous on `stream`. */
int wbcqp_problem_data(wbcqp_handle* handle, int slot, int batch, const wbcqp_state* state, const wbcqp_inputs* rows,
                       void* stream);
/* Same with HOST pointers; blocks until done. */
int wbcqp_problem_data_host(wbcqp_handle* handle, int slot, int batch, const wbcqp_state* state, const wbcqp_inputs* rows);

/* ---- One whole control tick on the device: all of Controller::_solve (controller.cpp:231-313) for `batch` instances ----
 * rows (wbcqp_problem_data) -> QP (wbcqp_solve_batch) -> state integration (wbcqp_integrate), ordered on one stream.
 * All pointers are DEVICE pointers. `rows` is the QP record: M, h, A, b1, Ac, bc, blb, bub are scratch the rows kernel
 * writes and the solve reads; tlb, tub, w are supplied by the caller (constant limits, task weights). */
typedef struct {
    wbcqp_state state;
    wbcqp_inputs rows;
    wbcqp_outputs out;
    void* q_next;    /* [batch][nq] */
    void* v_next;    /* [batch][nv] */
    void* q_solver;  /* [batch][nv] or NULL */
    double dt;
} wbcqp_tick_io;

int wbcqp_tick(wbcqp_handle* handle, int slot, int batch, const wbcqp_tick_io* io, void* stream);
/* Same with HOST pointers; blocks until done.  Only the state, the references and the constant tlb / tub / w cross to the
 * device and only x, tau, status, iters (objective, n_active, active_mask if given) and the integrated state come back: about 4 KB per
 * Talos instance instead of the 34 KB record.  The row arrays M .. bub of io->rows may be NULL; where given they receive the
 * rows of this tick (what Controller::cost() reads). */
int wbcqp_tick_host(wbcqp_handle* handle, int slot, int batch, const wbcqp_tick_io* io);

/* ---- K ticks of every instance in one call, without the batch barrier ----
 * Instance i's tick t + 1 depends on instance i's tick t only (controller.cpp:254-256), never on the batch; a loop of wbcqp_tick calls
 * makes every tick wait for the slowest QP of the WHOLE batch, and on the squat stream that one QP is 0.29 ms long while the rest of a
 * 1024-QP batch is worth 0.13 ms of the chip.  wbcqp_rollout enqueues all K ticks up front (rows -> QP -> integration, the kernels of
 * wbcqp_tick; the host waits for nothing), either as ONE stream of ticks -- what K calls of wbcqp_tick are -- or, for 512 instances and
 * more, as TWO sub-batches on HIP streams of their own (each with its own launch-order state and queue counter), whose launches overlap
 * on the device: the tail of one sub-batch's solve runs beside the bulk of the other's.  Which of the two is faster depends on the
 * workload, measured (tools/rollout_bench.py, profiles/r03/rollout_bench.log, profiles/r04/): two sub-batches 1.04x of the tick loop at
 * B = 1024 where one instance stays the hard one, but 0.76x where every instance is heavy (nothing idles in a tick's tail) and 0.93x at
 * B = 4096 (the launch hides its own tail); three sub-batches 1.02x, four 0.57x.  So the library MEASURES: every roll-out is timed on
 * the device by an event pair that a later call reads without blocking; the first roll-outs of a (slot, batch) run as one stream, then
 * as two -- the first sample of either form is not kept: it pays that form's allocations, stream creation and a device synchronisation
 * (so the first TWO calls of each form block on the device once) --, from then on whichever was faster, the other one tried again every
 * 64th call whatever its last figure.  With both measured it is the tick loop or better, up to what a drifting workload changes between
 * two such retries; what it buys is bounded by the slowest INSTANCE's own chain of K ticks (round 5, B = 1024: 1.06x with two sub-batches,
 * 1.09x with three -- a tick's solve is 0.12 ms of which the hard instance's own chain is 0.10 -- profiles/r05/).  Bit for bit the result of K
 * calls of wbcqp_tick with q_next / v_next fed back, whatever the choice.  (A single persistent kernel running the three phases back to back per instance was built and measured first: 0.69x of the tick
 * loop -- the three phases together do not fit 256 VGPRs, and an instance's rows phase runs at 8 waves per CU instead of 16;
 * profiles/r03/fused_rollout_kernel_not_kept.log, DESIGN.md.)  Loop shape: qp_timer_test.cpp:55-63.  All pointers are DEVICE pointers.
 * Ordered after everything already on `stream`; everything enqueued on `stream` afterwards sees the results. */
typedef struct {
    wbcqp_state state;   /* q, v: the state before the first tick; ref: [n_ticks][batch][nref], one reference sample per tick;
                            momentum: of the LAST tick's state, or NULL */
    const void* tlb;     /* [batch][na] torque bounds (NULL if !act_bounds), constant over the ticks */
    const void* tub;
    const void* w;       /* [batch][n_tasks] */
    wbcqp_outputs out;   /* x, tau, status, iters (objective, n_active) of the LAST tick */
    void* q_next;        /* [batch][nq] the state after n_ticks ticks */
    void* v_next;        /* [batch][nv] */
    void* q_solver;      /* [batch][nv] or NULL */
    double dt;
    int32_t* iters_sum;  /* [batch] active-set iterations over all ticks, may be NULL */
    int32_t* ticks_ok;   /* [batch] ticks whose QP was solved (a tick that is not keeps the state, like wbcqp_integrate), may be NULL */
} wbcqp_rollout_io;
/* The slot needs a model (wbcqp_set_model).  The first call of a larger shape allocates the record arrays and the state ping-pong (and
 * synchronises the device to do it); WBCQP_ROLLOUT_STREAMS in the environment fixes the number of sub-batches (1 .. 8; 1: a plain tick loop)
 * instead of the measured choice. */
int wbcqp_rollout(wbcqp_handle* handle, int slot, int batch, int n_ticks, const wbcqp_rollout_io* io, void* stream);

/* The same sequence captured once into a HIP graph and replayed: one graph launch per tick instead of four kernel
 * launches (what matters when the batch is small -- one robot at 1 kHz is the reference's own use case).  The graph is
 * bound to the pointers of `io`; the caller changes the CONTENT of those buffers between ticks (new references, or
 * q_next / v_next copied back into state.q / state.v), not the pointers.  wbcqp_tick_graph_create runs one ordinary tick
 * first (so nothing is allocated inside the capture) and blocks until it is done. */
typedef struct wbcqp_graph wbcqp_graph;
int wbcqp_tick_graph_create(wbcqp_handle* handle, int slot, int batch, const wbcqp_tick_io* io, wbcqp_graph** out);
int wbcqp_tick_graph_launch(wbcqp_handle* handle, wbcqp_graph* graph, void* stream);
int wbcqp_tick_graph_destroy(wbcqp_handle* handle, wbcqp_graph* graph);

int wbcqp_sync(wbcqp_handle* handle, void* stream);

/* Diagnostics: the launch order the next solve of the last launch's shape will use (what the schedule kernels left),
   copied to host memory after a device synchronisation.  Returns the number of entries written (0: no order yet, the next
   launch runs in index order), or a negative wbcqp_error.  *packed (may be NULL): 1 if this is the bin-packed order,
   0 if plain longest-first.  Nothing in the reference corresponds to it; tests compare it with a host model. */
int wbcqp_launch_order(wbcqp_handle* handle, int32_t* order, int32_t capacity, int32_t* packed);

#ifdef __cplusplus
}
#endif
#endif /* WBCQP_H */
