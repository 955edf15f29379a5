// wbcqp_types.hpp -- constants and the structures shared by the host API and the kernels: LDS slot map, the
// per-structure description that travels by value with every launch, per-group launch arguments.
#pragma once

#include <stdint.h>

namespace wbcqp {
constexpr int kWave = 64;
constexpr int kThreads = 256; // threads per QP
constexpr int kWaves = kThreads / kWave;
constexpr int kMaxBlocks = 16;
constexpr int kMaxGroups = 8;

// All small per-QP vectors live at FIXED offsets (multiples of kSlot doubles) from one LDS base, so that the compiler
// addresses them with immediates instead of keeping ~40 wave-uniform pointers alive in SGPRs (they spilled).
// Limits that make this legal are checked on the host: n <= 126, n_tasks, n_dense, n_bound, na, 6 nc <= 128,
// level-1 rows <= 256, one-sided inequality rows <= 512.
constexpr int kSlot = 128;
enum VecSlot {
    V_H = 0, V_X, V_NP, V_D, V_Z, V_XOLD, V_R, V_U, V_UOLD, V_Q, V_G, V_W, V_WROW, V_BLB, V_BUB, V_TL, V_TU, V_BC,
    V_RDINV, V_DINV, V_RED,
    V_PRM,              // 2 slots
    V_B1 = V_PRM + 2,   // 2 slots
    V_S = V_B1 + 2,     // 4 slots
    V_STASH = V_S + 4,  // 2 slots
    V_PART = V_STASH + 2, // 5 slots
    V_COUNT = V_PART + 5
};
constexpr int kIntA = 0, kIntAold = 128, kIntGskip = 256, kIntIai = 384, kIntIaexcl = 896, kIntMeta = 1408, kIntCount = 1920;
// packed row descriptor: bits 0-1 kind, bit 2 negated copy (-A row), bits 3-10 local row, bits 11-14 contact, bits 15-22 column
__host__ __device__ inline int row_meta_pack(int kind, int neg, int rr, int ct, int col) { return kind | (neg << 2) | (rr << 3) | (ct << 11) | (col << 15); }

enum { INEQ_BOUNDS = 0, INEQ_ACTUATION = 1, INEQ_FORCE = 2 };
enum { HQP_UNKNOWN = -1, HQP_OPTIMAL = 0, HQP_INFEASIBLE = 1, HQP_UNBOUNDED = 2, HQP_MAX_ITER = 3, HQP_ERROR = 4 };

// Constant structure of a task stack, resident in device memory (one per slot).
// inequality blocks in task-stack order (host side only: the device works from the packed row descriptors)
struct HostBlocks {
    int n_blocks;
    int blk_kind[kMaxBlocks], blk_arg[kMaxBlocks], blk_off[kMaxBlocks], blk_rows[kMaxBlocks];
};

struct DevStruct {
    int nv, na, nc, k, n, nu;
    int n_dense, n_tasks, n_sel, n_bound, act_bounds;
    int neq, nin2, r1;
    int max_iter;
    double hessian_reg;
    const int *dense_row_task, *sel_col, *sel_task, *forcereg_task, *bound_col;
    const double *force_gen; // [nc][6][12]
    const double *ftf;       // [nc][12][12]  F'F,  F = diag(w_f) T
    const double *ft;        // [nc][12][6]   F'
    const double *fric_mat, *fric_lb, *fric_ub;
    const int* rowmeta;      // [nin2] packed descriptor of every one-sided inequality row (see row_meta_*)
    const unsigned* mpack;   // [nv(nv+1)/2] packed-M element e=(i,j) -> LDS offsets (i ldm + j) | (j ldm + i) << 16
    const unsigned* apack;   // [n_dense nv] task-row element (r, col) -> offset in the staged rows: with t = col & 15, u = col >> 4 (thread (ta, te) of the
                             // 16 x 16 grid reads columns t + 16 u, u < 4, as two 16-byte pairs) r 64 + (u >> 1) 32 + 2 t + (u & 1): the sixteen
                             // pairs (u = 0, 1) of a row are contiguous, then the sixteen pairs (u = 2, 3) -- a 16-byte read by sixteen lanes with
                             // consecutive t covers 256 contiguous bytes.  (Until round 4: r 64 + 4 t + u, the two pairs of a lane side by side:
                             // lanes t and t + 8 then met in the same banks, and the H loop is bound by the LDS pipe.)
    // LDS layout: leading dimensions and element offsets (in doubles)
    int ldj, ldm, ldc, ldb;
    int o_J, o_R, o_M, o_Jc, o_Ac, o_vec, o_eqw, o_eqt;
    int o_int; // int area (fixed slots, see kInt*)
    int fric_lds; // 1: the friction tables (238 doubles per contact) fit the equality-phase scratch, which is free in the inequality loop
    int lds_doubles;
    // compact layout (wbcqp_compact.hpp: half the LDS, two QPs per CU): 1 when the structure is eligible and not vetoed
    int compact;
    int act_off;             // first one-sided row of the actuation block (+A rows; the -A rows follow na later), -1: none
    int o_pan;               // elimination panels, offset inside the J region (behind the staged task rows)
    const unsigned* acpack;  // [nc 6 nv] contact-Jacobian element (rr, kk) -> offset kk ldb + nu + rr in N = CE'
    // level-1 tasks that make H dense (full layout only): torque rows scale_j [M_a(joint_j,:) | -J_a(:,joint_j)'], cop rows from the record
    int dense_h;             // 1: H is assembled and eliminated as one n x n matrix (n <= 80)
    int n_acteq, acteq_task, cop_task;
    const int* acteq_joint;  // [n_acteq]
    const double* acteq_scale;
    // the force blocks' factor for ONE weight (compact kernels; null: none).  H_ff = w F'F + reg I depends on the record only through the level-1 weight w
    // of the contact's force-regularisation task (tasks.hpp:23 w_force_feet: a constant in every shipped stack), so its 12 x 12 elimination -- 2.9 k cycles of
    // a QP's set-up, on one wave, while two waves idle -- is the same for every QP that carries that weight.  Layout per contact (kFfcStride doubles):
    // [0] the weight the entry was made for (NaN: none yet), [2 .. 14) 1 / sqrt(pivot), [16 + 6 lane ..): the lane's y tile (4) and its two additions to tr(H).
    // Made on the device by the kernels' own elimination code (ffcache_kernel: the same instructions, the same bits) from the weights of the first QP of the
    // slot's first launch; a QP with another weight computes as before.
    const double* ffc;
};
constexpr int kFfcStride = 16 + 6 * 64;

// The integer sizes and LDS offsets of a structure on the compact layout as the kernel sees them.  The compact kernels exist once for ANY
// structure (these read from the DevStruct at run time) and once per SHIPPED stack with all of them as literals: every address then folds
// into an immediate, clamps and loop bounds are constants, and some forty wave-uniform values leave the SGPRs (the generic kernel spills
// 147 of them into VGPR lanes, a specialised one 101).  Measured on Talos (tools/throughput_time.py): median QP alone 61.3 -> 55.2 us, B = 8192
// 6.24 -> 6.56 M QP/s, config 2 replayed 5.72 -> 6.12 M.  Same source, same arithmetic order: bit for bit the generic kernel's results
// (tests/test_gpu_specialised.py).  The host picks an instantiation only when the structure's derived layout EQUALS the literals.
struct Dims {
    int nv, na, nc, k, n, nu, n_dense, n_tasks, n_sel, n_bound, act_bounds, neq, nin2, r1, max_iter, ldj, ldb, o_J, o_R, o_vec, o_int, o_pan, act_off;
};
constexpr int kNumSpecs = 3;
constexpr Dims kSpecDims[kNumSpecs] = {
    // 1: etc/talos/tasks.yaml on talos.urdf (nv 50, two foot contacts, bounds + actuation bounds): n 74, nEq 18, 244 one-sided rows
    {50, 44, 2, 24, 74, 6, 41, 19, 44, 44, 1, 18, 244, 97, 1000, 74, 22, 0, 5478, 7366, 8806, 2714, 88},
    // 2: etc/icub/tasks.yaml (nv 38, two contacts, no actuation bounds): n 62, nEq 18, 132 one-sided rows
    {38, 32, 2, 24, 62, 6, 39, 13, 32, 32, 0, 18, 132, 83, 1000, 62, 22, 0, 3846, 5306, 6458, 2582, -1},
    // 3: Talos in single support (walk / walk-on-spot between a lift-off and a touch-down): n 62, nEq 12, 210 one-sided rows
    {50, 44, 1, 12, 62, 6, 41, 18, 44, 44, 1, 12, 210, 91, 1000, 62, 14, 0, 3846, 5384, 6664, 2714, 88},
};
__host__ __device__ inline Dims dims_from(const struct DevStruct& S);

template <typename TI>
struct GroupArgs {
    DevStruct st; // by value: the sizes, offsets and table pointers arrive with the kernel arguments, not behind a pointer
    const TI *M, *h, *A, *b1, *Ac, *bc, *blb, *bub, *tlb, *tub, *w, *Acop;
    TI *x, *tau, *objective;
    int *status, *iters, *n_active;
    unsigned* amask; // [count][8] active-set mask of the solution (out; in as the pick hint when warm != 0), or null
    int warm;        // WBCQP_FLAG_WARM_START
    long long* dbg; // per-QP phase cycle counters, only written by the WBCQP_STAMPS diagnostic build
    int count;
};

template <typename TI>
struct GroupTable {
    int n;
    const int* order; // launch order -> QP index (longest-first schedule of the previous launch of this shape), or null
    GroupArgs<TI> g[kMaxGroups];
};

__host__ __device__ inline Dims dims_from(const DevStruct& S)
{
    return Dims{S.nv, S.na, S.nc, S.k, S.n, S.nu, S.n_dense, S.n_tasks, S.n_sel, S.n_bound, S.act_bounds, S.neq, S.nin2, S.r1, S.max_iter,
                S.ldj, S.ldb, S.o_J, S.o_R, S.o_vec, S.o_int, S.o_pan, S.act_off};
}
// 1-based index into kSpecDims of the specialisation whose literals equal this compact layout, 0: none (max_iter is not matched: the
// instantiations read it from the structure)
inline int spec_of(const DevStruct& C)
{
    if (!C.compact) return 0;
    const Dims d = dims_from(C);
    for (int i = 0; i < kNumSpecs; ++i) {
        const Dims& s = kSpecDims[i];
        if (d.nv == s.nv && d.na == s.na && d.nc == s.nc && d.k == s.k && d.n == s.n && d.nu == s.nu && d.n_dense == s.n_dense && d.n_tasks == s.n_tasks &&
            d.n_sel == s.n_sel && d.n_bound == s.n_bound && d.act_bounds == s.act_bounds && d.neq == s.neq && d.nin2 == s.nin2 && d.r1 == s.r1 &&
            d.ldj == s.ldj && d.ldb == s.ldb && d.o_J == s.o_J && d.o_R == s.o_R && d.o_vec == s.o_vec && d.o_int == s.o_int &&
            d.o_pan == s.o_pan && d.act_off == s.act_off)
            return i + 1;
    }
    return 0;
}

static_assert(sizeof(GroupTable<double>) <= 4000, "the group table travels as a kernel argument: the kernarg segment is 4 KiB");

// iteration counts of the launch just finished, for the schedule of the next one
struct ScheduleArgs {
    int n;
    const int* iters[kMaxGroups];
    int count[kMaxGroups];
};
} // namespace wbcqp
