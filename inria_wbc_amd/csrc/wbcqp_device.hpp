// wbcqp_device.hpp -- device-side data model and the fused one-workgroup-per-QP kernel (gfx950).
//
// One workgroup of four wavefronts (256 threads, one wave per SIMD of a CU) owns one QP for its whole life:
// assemble H,g (P2 cost part), Cholesky + J = L^-T (P3 preprocessing), Goldfarb-Idnani equality phase and
// inequality loop (P3), torque decode (P4).  Everything that is touched more than once lives in LDS; HBM is
// read once per QP (the compact per-QP record) and written once (x, tau, status, iters).
// Why four waves: the QP's working set (~120 KiB for Talos) admits one QP per CU, and ONE wave alone on a CU
// gets only 32-43 B/clk out of the LDS (measured, tools/ubench/lds_lone_wave.hip) -- a sixth of what the CU has.
//
// What each phase stands behind in the reference (/root/reference):
//   assemble / stack : tsid computeProblemData + SolverHQuadProgFast::solve  controller.cpp:244,247
//   GI active set    : eiquadprog-fast solve_quadprog                        controller.cpp:247
//   decode           : getActuatorForces / getAccelerations                  controller.cpp:250-251
// The dense CE / CI matrices of the reference are never formed: rows are regenerated from their
// structure (+-e_col bounds rows, +-[M_a | -J_a'] actuation rows, 17x12 friction blocks).
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
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
    const unsigned* apack;   // [n_dense nv] task-row element (r, col) -> offset r 64 + (col & 15) 4 + (col >> 4) in the staged rows
    // LDS layout: leading dimensions and element offsets (in doubles)
    int ldj, ldm, ldc, ldb;
    int o_J, o_R, o_M, o_Jc, o_Ac, o_vec, o_eqw, o_eqt;
    int o_int; // int area (fixed slots, see kInt*)
    int fric_lds; // 1: the friction tables (238 doubles per contact) fit the equality-phase scratch, which is free in the inequality loop
    int lds_doubles;
};

template <typename TI>
struct GroupArgs {
    DevStruct st; // by value: the sizes, offsets and table pointers arrive with the kernel arguments, not behind a pointer
    const TI *M, *h, *A, *b1, *Ac, *bc, *blb, *bub, *tlb, *tub, *w;
    TI *x, *tau, *objective;
    int *status, *iters, *n_active;
    long long* dbg; // per-QP phase cycle counters, only written by the WBCQP_STAMPS diagnostic build
    int count;
};

template <typename TI>
struct GroupTable {
    int n;
    const int* order; // launch order -> QP index (longest-first schedule of the previous launch of this shape), or null
    GroupArgs<TI> g[kMaxGroups];
};

// iteration counts of the launch just finished, for the schedule of the next one
struct ScheduleArgs {
    int n;
    const int* iters[kMaxGroups];
    int count[kMaxGroups];
};

#ifdef __HIPCC__

// ------------------------------------------------------------------------------------------------
// wave64 primitives (DPP row operations + readlane)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bsync() { __syncthreads(); }

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
// value of the lane a DPP control selects (all source lanes valid: no old value to preserve, no copy)
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
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

// all-lanes sum within one wave (every lane returns the bitwise-identical total)
__device__ __forceinline__ double wave_sum(double v)
{
    WBCQP_ROW_REDUCE(v, op_add)
    double r0 = bcast_lane(v, 0), r1 = bcast_lane(v, 16), r2 = bcast_lane(v, 32), r3 = bcast_lane(v, 48);
    return (r0 + r1) + (r2 + r3);
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

// In-kernel phase stamps (diagnostic build only: -DWBCQP_STAMPS). Never compiled into the product library.
#ifdef WBCQP_STAMPS
constexpr int kStamps = 24;
#define STAMP_DECL c.st_prev_ = clock64(); for (int i_ = 0; i_ < kStamps; ++i_) c.st_acc_[i_] = 0;
#define STAMP(i) { long long now_ = clock64(); c.st_acc_[i] += now_ - c.st_prev_; c.st_prev_ = now_; }
#else
#define STAMP_DECL
#define STAMP(i)
#endif

// ------------------------------------------------------------------------------------------------
// per-workgroup context: LDS pointers + sizes (all uniform across the 256 threads)
// ------------------------------------------------------------------------------------------------
struct Ctx {
    const DevStruct* S;
    int tid, lane, wave;
    int nv, na, nc, k, n, nu, neq, nin2, ldj, ldm, ldc, ldb;
    double *J, *R, *M, *Jc, *Ac, *h, *x, *np, *d, *z, *xold, *r, *u, *uold, *s;
    double *blb, *bub, *tl, *tu, *bc, *prm, *rdinv, *dinv, *g, *w, *b1, *q, *wrow, *red, *part, *stash, *eqw, *eqt;
    int *A, *Aold, *iai, *iaexcl, *gskip, *meta;
    int iq;
    int rslot; // alternating slot of the block-reduction scratch
    double R_norm;
#ifdef WBCQP_STAMPS
    long long st_prev_, st_acc_[kStamps];
#endif
};

// ---- workgroup-wide reductions: wave-level DPP reduce, four partials through LDS, one barrier ----
__device__ __forceinline__ double block_sum(Ctx& c, double v)
{
    v = wave_sum(v);
    double* slot = c.red + c.rslot * 16;
    if (c.lane == 0) slot[c.wave] = v;
    bsync();
    const double t = (slot[0] + slot[1]) + (slot[2] + slot[3]);
    c.rslot ^= 1;
    return t;
}
__device__ __forceinline__ void block_sum4(Ctx& c, double& a, double& b, double& e, double& f)
{
    a = wave_sum(a);
    b = wave_sum(b);
    e = wave_sum(e);
    f = wave_sum(f);
    double* slot = c.red + c.rslot * 16;
    if (c.lane == 0) {
        slot[c.wave] = a;
        slot[4 + c.wave] = b;
        slot[8 + c.wave] = e;
        slot[12 + c.wave] = f;
    }
    bsync();
    a = (slot[0] + slot[1]) + (slot[2] + slot[3]);
    b = (slot[4] + slot[5]) + (slot[6] + slot[7]);
    e = (slot[8] + slot[9]) + (slot[10] + slot[11]);
    f = (slot[12] + slot[13]) + (slot[14] + slot[15]);
    c.rslot ^= 1;
}
__device__ __forceinline__ ValIdx block_argmin(Ctx& c, ValIdx a)
{
    a = wave_argmin(a);
    double* slot = c.red + c.rslot * 16;
    if (c.lane == 0) {
        slot[c.wave] = a.v;
        slot[4 + c.wave] = __hiloint2double(0, a.i);
    }
    bsync();
    ValIdx r = {slot[0], __double2loint(slot[4])};
#pragma unroll
    for (int w = 1; w < kWaves; ++w) r = vi_min(r, ValIdx{slot[w], __double2loint(slot[4 + w])});
    c.rslot ^= 1;
    return r;
}
__device__ __forceinline__ int block_max_int(Ctx& c, int v)
{
    v = wave_max_int(v);
    double* slot = c.red + c.rslot * 16;
    if (c.lane == 0) slot[c.wave] = __hiloint2double(0, v);
    bsync();
    int r = __double2loint(slot[0]);
#pragma unroll
    for (int w = 1; w < kWaves; ++w) r = max(r, __double2loint(slot[w]));
    c.rslot ^= 1;
    return r;
}

// ds_read2_b64 costs 8 LDS cycles per wave where two ds_read_b64 cost 2 each and one ds_read_b128 4 (MI355X_MICROARCH.md,
// LDS table) -- it matters in the loops that are LDS-bound.  opaque() hides how a pointer was derived, so the load/store
// optimizer cannot pair its accesses with a neighbour's; ld2() is the 16-byte-aligned pair read.
typedef double double2v __attribute__((ext_vector_type(2)));
// 1/x to full precision without the IEEE division's scaling and fix-up: v_rcp_f64 and two Newton steps
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ double2v ld2(const double* p) { return *reinterpret_cast<const double2v*>(__builtin_assume_aligned(p, 16)); }
__device__ __forceinline__ int wave_min_int(int v) { return -wave_max_int(-v); }

// acc[2][4] += sum_k a_i(k) * b(k, 0..3) over the wave-uniform range [k0, k1): element k of operand i is at
// base_a[oa_i + k sa], the four b's at pb[k sb .. + 3] (16-byte aligned).  Four k-steps per trip; the operands of the
// next trip are in flight while this one multiplies (one wave per SIMD: nothing else hides the LDS latency), and every
// operand stream has its own running pointer so that a step costs no index arithmetic.  The prefetch of the last trip
// reads up to four steps past k1 (never used; the rows after any operand here are still inside the LDS allocation).
__device__ __forceinline__ void tile2x4(const double* base_a, int oa0, int oa1, int sa, const double* pb, int sb, int k0, int k1,
                                        double (&acc)[2][4])
{
    if (k0 >= k1) return;
    const double* pa0 = base_a + oa0 + k0 * sa;
    const double* pa1 = base_a + opaque(oa1) + k0 * sa;
    const double* pbk = pb + k0 * sb;
    auto mac1 = [&](double x0, double x1, const double2v& u, const double2v& w) __attribute__((always_inline)) {
        acc[0][0] = fma(x0, u.x, acc[0][0]); acc[0][1] = fma(x0, u.y, acc[0][1]);
        acc[0][2] = fma(x0, w.x, acc[0][2]); acc[0][3] = fma(x0, w.y, acc[0][3]);
        acc[1][0] = fma(x1, u.x, acc[1][0]); acc[1][1] = fma(x1, u.y, acc[1][1]);
        acc[1][2] = fma(x1, w.x, acc[1][2]); acc[1][3] = fma(x1, w.y, acc[1][3]);
    };
    const int sa2 = 2 * sa, sa3 = 3 * sa, sb2 = 2 * sb, sb3 = 3 * sb;
    auto ld4 = [&](double (&a)[4][2], double2v (&b)[4][2]) __attribute__((always_inline)) {
        a[0][0] = pa0[0]; a[0][1] = pa1[0]; a[1][0] = pa0[sa]; a[1][1] = pa1[sa];
        a[2][0] = pa0[sa2]; a[2][1] = pa1[sa2]; a[3][0] = pa0[sa3]; a[3][1] = pa1[sa3];
        b[0][0] = ld2(pbk); b[0][1] = ld2(pbk + 2);
        b[1][0] = ld2(pbk + sb); b[1][1] = ld2(pbk + sb + 2);
        b[2][0] = ld2(pbk + sb2); b[2][1] = ld2(pbk + sb2 + 2);
        b[3][0] = ld2(pbk + sb3); b[3][1] = ld2(pbk + sb3 + 2);
        pa0 += 4 * sa;
        pa1 += 4 * sa;
        pbk += 4 * sb;
    };
    double a0[4][2], a1[4][2];
    double2v b0[4][2], b1[4][2];
    int left = k1 - k0;
    ld4(a0, b0);
    while (left >= 8) {
        ld4(a1, b1);
#pragma unroll
        for (int q = 0; q < 4; ++q) mac1(a0[q][0], a0[q][1], b0[q][0], b0[q][1]);
        ld4(a0, b0);
#pragma unroll
        for (int q = 0; q < 4; ++q) mac1(a1[q][0], a1[q][1], b1[q][0], b1[q][1]);
        left -= 8;
    }
    if (left >= 4) {
        ld4(a1, b1);
#pragma unroll
        for (int q = 0; q < 4; ++q) mac1(a0[q][0], a0[q][1], b0[q][0], b0[q][1]);
        left -= 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < left) mac1(a1[q][0], a1[q][1], b1[q][0], b1[q][1]);
    }
    else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < left) mac1(a0[q][0], a0[q][1], b0[q][0], b0[q][1]);
    }
}

// sum_{k in [k0,k1)} a[k sa] b[k sb] with eight products' operands in flight before the first FMA (one wave per SIMD:
// nothing else hides the LDS latency; a two-term loop body costs a full round trip per two terms)
__device__ __forceinline__ double dot8(const double* a, int sa, const double* b, int sb, int k0, int k1)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[(k + u) * sa];
            y[u] = b[(k + u) * sb];
        }
        s0 = fma(x[0], y[0], s0); s1 = fma(x[1], y[1], s1); s2 = fma(x[2], y[2], s2); s3 = fma(x[3], y[3], s3);
        s0 = fma(x[4], y[4], s0); s1 = fma(x[5], y[5], s1); s2 = fma(x[6], y[6], s2); s3 = fma(x[7], y[7], s3);
    }
    if (k < k1) { // tail: clamp the index, zero the weight
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = min(k + u, k1 - 1);
            x[u] = a[kk * sa];
            y[u] = (k + u < k1) ? b[kk * sb] : 0.0;
        }
        s0 = fma(x[0], y[0], s0); s1 = fma(x[1], y[1], s1); s2 = fma(x[2], y[2], s2); s3 = fma(x[3], y[3], s3);
        s0 = fma(x[4], y[4], s0); s1 = fma(x[5], y[5], s1); s2 = fma(x[6], y[6], s2); s3 = fma(x[7], y[7], s3);
    }
    return (s0 + s1) + (s2 + s3);
}

// packed upper-triangular R with one spare slot per column (column j holds rows 0..j+1):
__device__ __forceinline__ int roff(int j) { return (j * (j + 3)) >> 1; }
// first structurally non-zero column / one past the last of row i: H is block diagonal (dv block, one 12x12 block per contact)
__device__ __forceinline__ int blk_begin(int i, int nv) { return (i < nv) ? 0 : nv + 12 * ((i - nv) / 12); }
__device__ __forceinline__ int blk_end(int i, int nv) { return (i < nv) ? nv : nv + 12 * ((i - nv) / 12) + 12; }

// d = J' np over the support [k0, k1) of np (eiquadprog compute_d).
// threads 0..127 own column idx for the first half of the support, threads 128..255 for the second half;
// the two partial sums meet in LDS.  Ends with a barrier: d is visible to every thread on return.
__device__ __forceinline__ void compute_d(Ctx& c, int k0, int k1)
{
    const int n = c.n, ldj = c.ldj;
    const int idx = c.tid & 127, grp = c.tid >> 7;
    const int mid = k0 + ((k1 - k0 + 1) >> 1);
    const int ka = grp ? mid : k0, kb = grp ? k1 : mid;
    if (idx < n) {
        const double* Jc0 = c.J + idx;
        const double* np = c.np;
        double a0 = 0.0, b0 = 0.0;
        int kk = ka;
        for (; kk + 4 <= kb; kk += 4) {
            const double v0 = np[kk], v1 = np[kk + 1], v2 = np[kk + 2], v3 = np[kk + 3];
            a0 = fma(Jc0[kk * ldj], v0, a0);
            b0 = fma(Jc0[(kk + 1) * ldj], v1, b0);
            a0 = fma(Jc0[(kk + 2) * ldj], v2, a0);
            b0 = fma(Jc0[(kk + 3) * ldj], v3, b0);
        }
        for (; kk < kb; ++kk) a0 = fma(Jc0[kk * ldj], np[kk], a0);
        c.part[grp * 128 + idx] = a0 + b0;
    }
    bsync();
    if (c.tid < n) c.d[c.tid] = c.part[c.tid] + c.part[128 + c.tid];
    bsync();
}
// r = R[:iq,:iq]^-1 d[:iq] for the rows rlo..iq-1 on ONE wave (update_r): column-oriented back substitution, the pivot
// travels by readlane, 1/R(j,j) and the column entries of four steps are fetched ahead of the dependent chain.
__device__ __forceinline__ void update_r_wave(Ctx& c, int rlo)
{
    const int lane = c.lane, iq = c.iq;
    if (iq <= rlo) return;
        double v0 = (lane < iq) ? c.d[lane] : 0.0;
        double v1 = (lane + kWave < iq) ? c.d[lane + kWave] : 0.0;
        auto step = [&](int j, double rd, double ra, double rb) {
            const double dj = (j < kWave) ? bcast_lane(v0, j) : bcast_lane(v1, j - kWave);
            const double rj = dj * rd;
            if (lane == (j & (kWave - 1))) c.r[j] = rj;
            if (lane < j) v0 = fma(-rj, ra, v0);
            if (lane + kWave < j) v1 = fma(-rj, rb, v1);
        };
        int j = iq - 1;
        for (; j >= rlo + 3; j -= 4) {
            double rd[4], ra[4], rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j - u;
                const double* Rc = c.R + roff(jj);
                rd[u] = c.rdinv[jj];
                ra[u] = Rc[min(lane, jj)];
                rb[u] = Rc[min(lane + kWave, jj)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) step(j - u, rd[u], ra[u], rb[u]);
        }
        for (; j >= rlo; --j) {
            const double* Rc = c.R + roff(j);
            step(j, c.rdinv[j], Rc[min(lane, j)], Rc[min(lane + kWave, j)]);
        }
}

// z = J[:, iq:] d[iq:] (update_z) on waves 0..2 (each a third of the columns, both row sets), and
// r = R[:iq,:iq]^-1 d[:iq] (update_r) on wave 3: column-oriented back substitution, the pivot travels by
// readlane, 1/R(j,j) and the column entries of four steps are fetched ahead of the dependent chain.  Only r[rlo:iq] is
// formed: the inequality loop passes rlo = neq, because r of the equality rows only feeds the equality multipliers,
// which are neither an output nor an input of any decision.  Ends with barriers: z and r are visible on return.
__device__ __forceinline__ void update_z_r(Ctx& c, int rlo)
{
    const int n = c.n, ldj = c.ldj, lane = c.lane, iq = c.iq;
    if (c.wave < 3) {
        const int span = n - iq;
        const int chunk = (span + 2) / 3;
        const int ca = iq + c.wave * chunk, cb = min(n, ca + chunk);
        const int k0 = lane, k1 = lane + kWave;
        const bool has1 = k1 < n;
        if (k0 < n) {
            const double* J0 = c.J + (size_t)k0 * ldj;
            const double* J1 = c.J + (size_t)(has1 ? k1 : k0) * ldj;
            const double* d = c.d;
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
            int cc = ca;
            for (; cc + 4 <= cb; cc += 4) {
                const double v0 = d[cc], v1 = d[cc + 1], v2 = d[cc + 2], v3 = d[cc + 3];
                a0 = fma(J0[cc], v0, a0);
                a1 = fma(J1[cc], v0, a1);
                b0 = fma(J0[cc + 1], v1, b0);
                b1 = fma(J1[cc + 1], v1, b1);
                a0 = fma(J0[cc + 2], v2, a0);
                a1 = fma(J1[cc + 2], v2, a1);
                b0 = fma(J0[cc + 3], v3, b0);
                b1 = fma(J1[cc + 3], v3, b1);
            }
            for (; cc < cb; ++cc) {
                const double v0 = d[cc];
                a0 = fma(J0[cc], v0, a0);
                a1 = fma(J1[cc], v0, a1);
            }
            c.part[c.wave * 128 + k0] = a0 + b0;
            if (has1) c.part[c.wave * 128 + k1] = a1 + b1;
        }
    }
    else update_r_wave(c, rlo);
    bsync();
    if (c.tid < n) c.z[c.tid] = (c.part[c.tid] + c.part[128 + c.tid]) + c.part[256 + c.tid];
    bsync();
}

// add_constraint, Householder form.  eiquadprog zeroes d[iq+1:] with a chain of n-iq-1 Givens rotations of J's columns (a
// sequential sweep); one reflector H = I - tau v v' (v = d[iq:] - alpha e_0) spans the same subspaces, and its product
// with J needs no new matvec: J[:, iq:] v = z - alpha J[:, iq] with z = J[:, iq:] d[iq:] from update_z.  dn2 = |d[iq:]|^2.
// The new column of R is [d[:iq]; alpha].  Returns false when the constraint is (numerically) dependent.
__device__ __forceinline__ bool add_constraint_hh(Ctx& c, double dn2)
{
    const int n = c.n, ldj = c.ldj, iq = c.iq, tid = c.tid;
    const double diq = c.d[iq];
    double alpha = diq;
    if (iq + 1 < n && dn2 > 0.0) {
        const double inx = rsqrt(dn2);
        const double nx = dn2 * inx;
        alpha = (diq >= 0.0) ? -nx : nx;
        const double v0 = diq - alpha;
        const double tau = inx / (nx + fabs(diq));
        // w_k = tau (z_k - alpha J(k,iq)) for every row, published before anybody touches column iq
        if (tid < n) c.part[tid] = tau * (c.z[tid] - alpha * c.J[tid * ldj + iq]);
        bsync();
        // J(k,c) -= w_k v_c: thread (row k = tid & 127, half of the columns)
        const int k = tid & 127, half = tid >> 7;
        if (k < n) {
            const int span = n - iq;
            const int ca = iq + half * ((span + 1) >> 1), cb = half ? n : iq + ((span + 1) >> 1);
            double* Jk = c.J + k * ldj;
            const double wk = c.part[k];
            int cc = ca;
            if (cc == iq && cc < cb) {
                Jk[cc] = fma(-wk, v0, Jk[cc]);
                ++cc;
            }
            for (; cc + 4 <= cb; cc += 4) {
                const double d0 = c.d[cc], d1 = c.d[cc + 1], d2 = c.d[cc + 2], d3 = c.d[cc + 3];
                const double j0 = Jk[cc], j1 = Jk[cc + 1], j2 = Jk[cc + 2], j3 = Jk[cc + 3];
                Jk[cc] = fma(-wk, d0, j0);
                Jk[cc + 1] = fma(-wk, d1, j1);
                Jk[cc + 2] = fma(-wk, d2, j2);
                Jk[cc + 3] = fma(-wk, d3, j3);
            }
            for (; cc < cb; ++cc) Jk[cc] = fma(-wk, c.d[cc], Jk[cc]);
        }
    }
    double* Rc = c.R + roff(iq);
    for (int i = tid; i < iq; i += kThreads) Rc[i] = c.d[i];
    if (tid == kThreads - 1) {
        Rc[iq] = alpha;
        c.rdinv[iq] = 1.0 / alpha;
    }
    c.iq = iq + 1;
    bsync();
    if (fabs(alpha) <= 2.220446049250313e-16 * c.R_norm) return false; // degenerate
    c.R_norm = fmax(c.R_norm, fabs(alpha));
    return true;
}

// delete_constraint (eiquadprog): drop active constraint l; the Givens chain that restores R's triangle is
// sequential (short: only inequality columns move) and runs on wave 0; the matching J update is a lane-per-row
// sweep on waves 0 and 1.
__device__ __forceinline__ void delete_constraint(Ctx& c, int l)
{
    const int n = c.n, ldj = c.ldj, lane = c.lane, neq = c.neq;
    const int iq_old = c.iq;
    int found = -1;
    for (int i = neq + c.tid; i < iq_old; i += kThreads)
        if (c.A[i] == l) found = i;
    found = block_max_int(c, found);
    const int qq = found < 0 ? 0 : found;

    // remove the constraint from the active set and the duals: positions qq..iq_old-1 take their right
    // neighbour (position iq_old holds the candidate constraint), position iq_old is cleared
    {
        int a0 = 0;
        double u0 = 0.0;
        const int i0 = qq + c.tid;
        if (i0 < iq_old) {
            a0 = c.A[i0 + 1];
            u0 = c.u[i0 + 1];
        }
        bsync();
        if (i0 < iq_old) {
            c.A[i0] = a0;
            c.u[i0] = u0;
        }
        if (c.tid == kThreads - 1) {
            c.A[iq_old] = 0;
            c.u[iq_old] = 0.0;
        }
    }
    // R columns qq+1..iq_old-1 move one place left (each thread moves its own row)
    if (c.tid < 128)
        for (int col = qq; col < iq_old - 1; ++col) {
            const double* src = c.R + roff(col + 1);
            double* dst = c.R + roff(col);
            if (c.tid <= col + 1) dst[c.tid] = src[c.tid];
        }
    const int iq = iq_old - 1;
    c.iq = iq;
    bsync();
    if (iq == 0) return;

    if (c.wave == 0) {
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
            if (lane == 0) {
                Rj[j + 1] = 0.0;
                Rj[j] = rjj;
                c.rdinv[j] = 1.0 / rjj;
                c.prm[2 * j] = cc;
                c.prm[2 * j + 1] = ss;
                c.gskip[j] = 0;
            }
            for (int kc = j + 1 + lane; kc < iq; kc += kWave) {
                double* Rk = c.R + roff(kc);
                const double t1 = Rk[j], t2 = Rk[j + 1];
                Rk[j] = fma(t2, ss, t1 * cc);
                Rk[j + 1] = fma(t1, ss, -(t2 * cc));
            }
            // the next step reads what other lanes of this wave just wrote
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    bsync();
    // J columns qq..iq: ascending sweep, running element in a register
    if (c.wave < 2) {
        const int k = lane + c.wave * kWave;
        if (k < n) {
            double* Jk = c.J + (size_t)k * ldj;
            double t1 = Jk[qq];
            for (int j = qq; j < iq; ++j) {
                const double t2 = Jk[j + 1];
                if (c.gskip[j]) {
                    Jk[j] = t1; // columns j, j+1 untouched by this step
                    t1 = t2;
                    continue;
                }
                const double cc = c.prm[2 * j], ss = c.prm[2 * j + 1];
                Jk[j] = fma(t2, ss, t1 * cc);
                t1 = fma(t1, ss, -(t2 * cc));
            }
            Jk[iq] = t1;
        }
    }
    bsync();
}

// Builds the normal np of equality row i (CE.row(i)) in LDS; returns its support and ce0(i). No barrier.
__device__ __forceinline__ void build_eq_row(Ctx& c, int i, int& k0, int& k1, double& ce0)
{
    const int nv = c.nv, k = c.k, nu = c.nu, tid = c.tid;
    if (i < nu) {
        // base dynamics [M_u | -J_u'] x = -h_u
        if (tid < nv) c.np[tid] = c.M[i * c.ldm + tid];
        else if (tid < nv + k) c.np[tid] = -c.Jc[(tid - nv) * c.ldc + i];
        k0 = 0;
        k1 = c.n;
        ce0 = c.h[i];
    }
    else {
        const int rr = i - nu; // contact*6 + row
        if (tid < nv) c.np[tid] = c.Ac[rr * nv + tid];
        k0 = 0;
        k1 = nv;
        ce0 = -c.bc[rr];
    }
}

// What one thread keeps about the (at most two) rows of s it owns: rows tid and tid + 256
struct OwnRows {
    int meta[2];
    double ci0[2];
    double coef[2][12]; // friction rows only
};
__device__ __forceinline__ void own_rows_init(Ctx& c, OwnRows& o, const double* fmat, const double* flb, const double* fub)
{
#pragma unroll
    for (int z2 = 0; z2 < 2; ++z2) {
        const int i = c.tid + z2 * kThreads;
        o.meta[z2] = -1;
        o.ci0[z2] = 0.0;
#pragma unroll
        for (int m = 0; m < 12; ++m) o.coef[z2][m] = 0.0;
        if (i < c.nin2) {
            const int mt = c.meta[i];
            const int kind = mt & 3, rr = (mt >> 3) & 255, ct = (mt >> 11) & 15;
            const bool neg = (mt >> 2) & 1;
            o.meta[z2] = mt;
            if (kind == INEQ_BOUNDS) o.ci0[z2] = neg ? c.bub[rr] : -c.blb[rr];
            else if (kind == INEQ_ACTUATION) o.ci0[z2] = neg ? c.tu[rr] : -c.tl[rr];
            else {
                o.ci0[z2] = neg ? fub[ct * 17 + rr] : -flb[ct * 17 + rr];
                const double* B = fmat + (ct * 17 + rr) * 12;
#pragma unroll
                for (int m = 0; m < 12; ++m) o.coef[z2][m] = neg ? -B[m] : B[m];
            }
        }
    }
}

// tau' = M_a xn - J_a' fn with xn = x + t z formed on the fly (t = 0: xn = x exactly), four lanes per actuated row,
// partial sums meet by DPP inside the quad.  Every call sums in the same order, so the value for x + t z here is bitwise
// the value a later call on the stored x would give.  No barrier inside; out[rr] is written by the quad's first lane.
__device__ __forceinline__ void act_rows(Ctx& c, double* out, double t)
{
    const int nv = c.nv, nu = c.nu, k = c.k, na = c.na;
    const int rr = c.tid >> 2, q4 = c.tid & 3;
    const int row = nu + min(rr, na - 1);
    const double* Mr = c.M + row * c.ldm;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // the lane's terms j = q4 + 4 i, eight in flight; past the end: index clamped, weight zero
    for (int j = q4; j < nv; j += 32) {
        double mv[8], zv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = min(j + 4 * u, nv - 1);
            mv[u] = (j + 4 * u < nv) ? Mr[jj] : 0.0;
            zv[u] = c.z[jj];
            xv[u] = c.x[jj];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
            a0 = fma(mv[u], fma(t, zv[u], xv[u]), a0);
            a1 = fma(mv[u + 1], fma(t, zv[u + 1], xv[u + 1]), a1);
            a2 = fma(mv[u + 2], fma(t, zv[u + 2], xv[u + 2]), a2);
            a3 = fma(mv[u + 3], fma(t, zv[u + 3], xv[u + 3]), a3);
        }
    }
    for (int m = q4; m < k; m += 32) {
        double jv[8], zv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int mm = min(m + 4 * u, k - 1);
            jv[u] = (m + 4 * u < k) ? c.Jc[mm * c.ldc + row] : 0.0;
            zv[u] = c.z[nv + mm];
            xv[u] = c.x[nv + mm];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
            a0 = fma(-jv[u], fma(t, zv[u], xv[u]), a0);
            a1 = fma(-jv[u + 1], fma(t, zv[u + 1], xv[u + 1]), a1);
            a2 = fma(-jv[u + 2], fma(t, zv[u + 2], xv[u + 2]), a2);
            a3 = fma(-jv[u + 3], fma(t, zv[u + 3], xv[u + 3]), a3);
        }
    }
    double acc = (a0 + a1) + (a2 + a3);
    acc += dpp_get<0xB1>(acc);
    acc += dpp_get<0x4E>(acc);
    if (q4 == 0 && rr < na) out[rr] = acc;
}

// s = CI x + ci0 for the (at most two) rows this thread owns; tact = tau' of act_rows.  Stores s and returns
// sum min(s, 0) and the most violated eligible row (first index on ties).
__device__ __forceinline__ void own_rows_eval(Ctx& c, const OwnRows& o, const double* tact, double& psi, ValIdx& best)
{
    psi = 0.0;
    best = ValIdx{0.0, 0x7fffffff};
#pragma unroll
    for (int z2 = 0; z2 < 2; ++z2) {
        const int mt = o.meta[z2];
        if (mt >= 0) {
            const int kind = mt & 3, rr = (mt >> 3) & 255, ct = (mt >> 11) & 15, col = (mt >> 15) & 255;
            const bool neg = (mt >> 2) & 1;
            double v;
            if (kind == INEQ_BOUNDS) v = neg ? -c.x[col] : c.x[col];
            else if (kind == INEQ_ACTUATION) v = neg ? -tact[rr] : tact[rr];
            else {
                const double* f = c.x + c.nv + 12 * ct;
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < 12; ++m) a = fma(o.coef[z2][m], f[m], a);
                v = a;
            }
            v += o.ci0[z2];
            const int i = c.tid + z2 * kThreads;
            c.s[i] = v;
            c.iaexcl[i] = 1;
            psi += fmin(0.0, v);
            if (v < 0.0 && c.iai[i] != -1) best = vi_min(best, ValIdx{v, i});
        }
    }
}

// R rounds of 256 elements into registers; indices are clamped instead of masked so that the loads stay unconditional
// (a predicated load becomes an exec-mask branch and splits the block the scheduler works in).  len >= 1.
template <typename TI, int R>
__device__ __forceinline__ void ld_regs(const TI* __restrict__ src, int len, int tid, TI (&v)[R])
{
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] = src[min(tid + u * kThreads, len - 1)];
}
template <typename TI, int R>
__device__ __forceinline__ void st_regs(double* dst, int len, int tid, const TI (&v)[R])
{
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int e = tid + u * kThreads;
        if (e < len) dst[e] = (double)v[u];
    }
}

// sum over the 8 lanes of an aligned lane group (every lane of the group gets the total)
__device__ __forceinline__ double grp8_sum(double v)
{
    v += dpp_get<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(v); // row_half_mirror: the other quad of the same 8 lanes
    return v;
}

// sum over the 4 lanes of a quad (every lane gets the total)
__device__ __forceinline__ double quad_sum(double v)
{
    v += dpp_get<0xB1>(v); // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v); // quad_perm [2,3,0,1]
    return v;
}

// ------------------------------------------------------------------------------------------------
// Householder QR of B (n x m, n <= 80, m <= 22: 4 m + 2 n <= 256) with J <- J Q in its shadow.
// QR: columns resident in registers, 4 lanes per column (the first 4 m lanes), lane kc of a column keeps the row pairs
// (2 kc + 8 t, + 1), t < 10.  Per step only the reflector travels: the owner of column j leaves v_j (zeros above row j,
// v0 on it) and (tau_j, alpha_j) in LDS, every later column reads it once (10 x 16 bytes per lane), reduces its dot
// product over its quad by DPP and updates its registers; the lanes of column j + 1 go on to the next reflector.  One
// barrier per column, no reloads or stores of the trailing matrix.
// J Q: the last 2 n lanes are not part of the QR.  A lane pair keeps ROW r of J (40 + 40 doubles) in registers and
// applies every reflector as it appears: row <- row - tau (row . v_j) v_j' -- row-local, the two halves of the dot
// product meet by DPP, no barrier of its own, and it fits in the time the QR needs for its step.  This replaces the
// compact-WY route (W = J V, W T, J - W T V': three LDS GEMM phases, 20 k cycles) by work nobody waits for.
// (One lane per row needs 160 VGPRs for the row: the allocator then parks it in AGPRs, 4 k cycles per step.)
// On return: J = J0 Q in LDS, the packed R and 1/R(j,j).  Returns false when a column is (numerically) dependent.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool qr_resident(Ctx& c, const double* Bm, double* vbuf, double* sc)
{
    const int n = c.n, m = c.neq, ldb = c.ldb, ldj = c.ldj, tid = c.tid;
    const int e = tid >> 2, kc = tid & 3;
    const bool colv = e < m;
    const int es = colv ? e : 0;
    const int jl = tid - (kThreads - 2 * n); // lane pair of a row of J (the last 2 n lanes), < 0: none
    const int jr = jl >> 1, jh = jl & 1;
    double b[10][2];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 2 * kc + 8 * t + i;
            const double v = Bm[min(row, n - 1) * ldb + es];
            b[t][i] = (row < n) ? v : 0.0;
        }
    double jrow[40];
    if (jl >= 0) {
        const double* Jr = c.J + jr * ldj;
#pragma unroll
        for (int q = 0; q < 40; ++q) {
            const int cc = 40 * jh + q;
            const double v = Jr[min(cc, n - 1)];
            jrow[q] = (cc < n) ? v : 0.0;
        }
    }
    double my_alpha = 1.0;
    // reflector of column jn from the registers of its 4 lanes (call under e == jn).  T0 = jn >> 3 is a compile-time
    // constant per instance: row pairs below T0 lie above the diagonal, pairs past it below -- only pair T0 needs masks
    auto prepare_t = [&](auto T0c, int jn) __attribute__((always_inline)) {
        constexpr int T0 = decltype(T0c)::value;
        const int row0 = 2 * kc + 8 * T0;
        const double e0 = (row0 >= jn) ? b[T0][0] : 0.0, e1 = (row0 + 1 >= jn) ? b[T0][1] : 0.0;
        double sq0 = e0 * e0, sq1 = e1 * e1, sq2 = 0.0, sq3 = 0.0;
#pragma unroll
        for (int t = T0 + 1; t + 1 < 10; t += 2) {
            sq0 = fma(b[t][0], b[t][0], sq0);
            sq1 = fma(b[t][1], b[t][1], sq1);
            sq2 = fma(b[t + 1][0], b[t + 1][0], sq2);
            sq3 = fma(b[t + 1][1], b[t + 1][1], sq3);
        }
        if constexpr (((10 - (T0 + 1)) & 1) != 0) {
            sq0 = fma(b[9][0], b[9][0], sq0);
            sq1 = fma(b[9][1], b[9][1], sq1);
        }
        double x0 = (row0 == jn) ? b[T0][0] : ((row0 + 1 == jn) ? b[T0][1] : 0.0);
        const double nrm = quad_sum((sq0 + sq1) + (sq2 + sq3));
        x0 = quad_sum(x0);
        const double inx = rsqrt(nrm);
        const double nx = (nrm > 0.0) ? nrm * inx : 0.0; // exactly dependent column: alpha = 0 -> reported as redundant
        const double alpha = (x0 >= 0.0) ? -nx : nx;
        const double v0 = x0 - alpha;
        const double tj = fast_rcp(fma(nx, fabs(x0), nrm)); // 2 / v'v
        my_alpha = alpha;
        double* vb = vbuf + (jn & 1) * 80 + 2 * kc;
        if (row0 == jn) b[T0][0] = v0;
        if (row0 + 1 == jn) b[T0][1] = v0;
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            double2v o;
            if (t < T0) {
                o.x = 0.0;
                o.y = 0.0;
            }
            else if (t == T0) {
                o.x = (row0 >= jn) ? b[T0][0] : 0.0;
                o.y = (row0 + 1 >= jn) ? b[T0][1] : 0.0;
            }
            else {
                o.x = b[t][0];
                o.y = b[t][1];
            }
            *reinterpret_cast<double2v*>(__builtin_assume_aligned(vb + 8 * t, 16)) = o;
        }
        if (kc == 0) {
            sc[(jn & 1) * 2] = tj;
            sc[(jn & 1) * 2 + 1] = alpha;
        }
    };
    auto prepare = [&](int jn) __attribute__((always_inline)) {
        switch (jn >> 3) { // jn < 32
        case 0: prepare_t(std::integral_constant<int, 0>{}, jn); break;
        case 1: prepare_t(std::integral_constant<int, 1>{}, jn); break;
        case 2: prepare_t(std::integral_constant<int, 2>{}, jn); break;
        default: prepare_t(std::integral_constant<int, 3>{}, jn); break;
        }
    };
    if (e == 0) prepare(0);
    for (int j = 0; j < m; ++j) {
        bsync();
        const double tj = sc[(j & 1) * 2], alpha = sc[(j & 1) * 2 + 1];
        if (!(fabs(alpha) > 2.220446049250313e-16 * c.R_norm)) return false; // also catches a NaN pivot
        c.R_norm = fmax(c.R_norm, fabs(alpha));
        const double* vbj = vbuf + (j & 1) * 80;
        if (colv && e > j) {
            const double* vb = vbj + 2 * kc;
            double2v v[10];
#pragma unroll
            for (int t = 0; t < 10; ++t) v[t] = ld2(vb + 8 * t);
            double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
            for (int t = 0; t < 10; t += 2) {
                d0 = fma(v[t].x, b[t][0], d0);
                d1 = fma(v[t].y, b[t][1], d1);
                d2 = fma(v[t + 1].x, b[t + 1][0], d2);
                d3 = fma(v[t + 1].y, b[t + 1][1], d3);
            }
            const double coef = quad_sum((d0 + d1) + (d2 + d3)) * tj;
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                b[t][0] = fma(-coef, v[t].x, b[t][0]);
                b[t][1] = fma(-coef, v[t].y, b[t][1]);
            }
            if (e == j + 1) prepare(j + 1);
        }
        else if (jl >= 0) {
            const double* vh = vbj + 40 * jh;
            double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
            // Left alone the scheduler hoists all forty 16-byte reads of the unrolled loops (160 VGPRs on top of the 120 the
            // rows and columns hold) and the allocator then parks live values in AGPRs (see build.py).  Explicit software
            // pipeline instead: groups of five reads, the next group in flight while this one multiplies.
            double2v g0[5], g1[5];
            auto ldg = [&](int grp, double2v (&g)[5]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 5; ++i) g[i] = ld2(vh + 10 * grp + 2 * i);
            };
            auto dotg = [&](int grp, const double2v (&g)[5]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (i & 1) {
                        d2 = fma(g[i].x, jrow[10 * grp + 2 * i], d2);
                        d3 = fma(g[i].y, jrow[10 * grp + 2 * i + 1], d3);
                    }
                    else {
                        d0 = fma(g[i].x, jrow[10 * grp + 2 * i], d0);
                        d1 = fma(g[i].y, jrow[10 * grp + 2 * i + 1], d1);
                    }
                }
            };
            ldg(0, g0);
            __builtin_amdgcn_sched_barrier(0);
            ldg(1, g1);
            dotg(0, g0);
            __builtin_amdgcn_sched_barrier(0);
            ldg(2, g0);
            dotg(1, g1);
            __builtin_amdgcn_sched_barrier(0);
            ldg(3, g1);
            dotg(2, g0);
            __builtin_amdgcn_sched_barrier(0);
            ldg(0, g0); // first group of the update pass
            dotg(3, g1);
            double dot = (d0 + d1) + (d2 + d3);
            dot += dpp_get<0xB1>(dot); // the other half of the row
            const double coef = dot * tj;
            auto updg = [&](int grp, const double2v (&g)[5]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    jrow[10 * grp + 2 * i] = fma(-coef, g[i].x, jrow[10 * grp + 2 * i]);
                    jrow[10 * grp + 2 * i + 1] = fma(-coef, g[i].y, jrow[10 * grp + 2 * i + 1]);
                }
            };
            __builtin_amdgcn_sched_barrier(0);
            ldg(1, g1);
            updg(0, g0);
            __builtin_amdgcn_sched_barrier(0);
            ldg(2, g0);
            updg(1, g1);
            __builtin_amdgcn_sched_barrier(0);
            ldg(3, g1);
            updg(2, g0);
            __builtin_amdgcn_sched_barrier(0);
            updg(3, g1);
        }
    }
    // R packed, 1/R(j,j); J rows back to LDS
    if (colv) {
        double* Rc = c.R + roff(e);
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 2 * kc + 8 * t + i;
                if (row < e) Rc[row] = b[t][i];
            }
        if (kc == 0) {
            Rc[e] = my_alpha;
            c.rdinv[e] = 1.0 / my_alpha;
        }
    }
    if (jl >= 0) {
        double* Jr = c.J + jr * ldj + 40 * jh;
#pragma unroll
        for (int q = 0; q < 40; ++q)
            if (40 * jh + q < n) Jr[q] = jrow[q];
    }
    return true;
}

// y = R'^-1 rhs (forward substitution) on one wave: lane = index.  Column `lane` of the packed R sits in registers
// (clamped loads, all in flight at once); lanes past m carry zeros, so the loop runs to the compile-time bound MM >= m
// without guards.  The multipliers u = R^-1 y of the equality rows are not formed: no later decision reads them (the
// step-length test runs over the inequality rows only) and they are not an output.
template <int MM>
__device__ __forceinline__ void solve_y(Ctx& c, double* rhs)
{
    const int m = c.neq, lane = c.lane;
    const bool live = lane < m;
    const int ls = live ? lane : 0;
    double rc[MM];
#pragma unroll
    for (int i = 0; i < MM; ++i) rc[i] = c.R[roff(ls) + min(i, ls)]; // R(i, lane), used for i < lane
    const double rinv = live ? c.rdinv[ls] : 0.0;
    double yv = live ? rhs[ls] : 0.0;
#pragma unroll
    for (int i = 0; i < MM; ++i) {
        const double yi = bcast_lane(yv * rinv, i);
        if (lane == i) yv = yi;
        if (lane > i) yv = fma(-yi, rc[i], yv);
    }
    if (live) {
        rhs[lane] = yv; // y
        c.u[lane] = 0.0;
        c.A[lane] = -lane - 1;
    }
}

// ------------------------------------------------------------------------------------------------
// Equality phase, blocked.  eiquadprog adds the neq equalities one by one (d = J'n, Givens sweep over J, ...): 18
// full passes over J for Talos.  The same state (J, R, x, u, f) is reached in one go: with N = CE' (n x m) and
// B = J0' N, a Householder QR  Q' B = [R; 0]  gives J = J0 Q (applied as one rank-m update through the compact WY
// form Q = I - V T V'), and the equality-constrained minimiser follows from R' y = -(CE x0 + ce0):
// x = x0 + J[:, :m] y,  u = R^-1 y,  f = f0 + y'y / 2.  J' H J = I and J' N = [R; 0] hold exactly as after m
// add_constraint calls (R's diagonal signs and the null-space basis differ, which the later steps never see).
// Returns false on (numerically) redundant equalities -- upstream's REDUNDANT_EQUALITIES.
// Requires n <= 80, 1 <= m <= 22.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool equality_phase_blocked(Ctx& c, double& f_value)
{
    const int n = c.n, m = c.neq, nv = c.nv, nu = c.nu, ldj = c.ldj, ldb = c.ldb, tid = c.tid;
    double* Nm = c.eqw;        // N = CE' (n x m), later W = J0 V
    double* Bm = c.R + 256;    // B -> V (lower trapezoid) / R (strict upper), in the unused tail of the R region
    double* Tm = c.eqt;        // T (m x (m+1))
    double* tau = Tm + m * (m + 1);
    double* rhs = tau + 2 * m;  // later y

    // ---- N = CE': base dynamics rows [M_u | -J_u'], then the contact motion rows [A_c | 0].  Thread = (equality e,
    //      every 8th row): no index division, the ten loads of a thread are in flight together (m <= 22, n <= 80)
    {
        const int e = tid & 31, k8 = tid >> 5;
        if (e < m) {
            // both candidate sources are read unconditionally (clamped addresses) and selected: a load behind a per-lane
            // branch waits for its own round trip
            double v[10];
            if (e < nu) {
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int kk = min(k8 + 8 * i, n - 1);
                    const double mv = c.M[min(kk, nv - 1) * c.ldm + e];
                    const double jv = (n > nv) ? c.Jc[max(kk - nv, 0) * c.ldc + e] : 0.0;
                    v[i] = (kk < nv) ? mv : -jv;
                }
            }
            else {
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int kk = min(k8 + 8 * i, n - 1);
                    const double av = c.Ac[(e - nu) * nv + min(kk, nv - 1)];
                    v[i] = (kk < nv) ? av : 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < 10; ++i)
                if (k8 + 8 * i < n) Nm[(k8 + 8 * i) * ldb + e] = v[i];
        }
    }
    bsync();
    STAMP(21)
    // ---- rhs_e = -(N(:,e)'x0 + ce0_e): 8 lanes per equality, ten terms each in flight
    {
        const int e = tid >> 3, kc = tid & 7;
        const int es = min(e, m - 1);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < 10; i += 2) {
            const int k0 = min(kc + 8 * i, n - 1), k1 = min(kc + 8 * i + 8, n - 1);
            const double x0 = (kc + 8 * i < n) ? c.x[k0] : 0.0, x1 = (kc + 8 * i + 8 < n) ? c.x[k1] : 0.0;
            a0 = fma(Nm[k0 * ldb + es], x0, a0);
            a1 = fma(Nm[k1 * ldb + es], x1, a1);
        }
        const double acc = grp8_sum(a0 + a1);
        if (e < m && kc == 0) {
            const double ce0 = (e < nu) ? c.h[e] : -c.bc[e - nu];
            rhs[e] = -(acc + ce0);
        }
    }
    STAMP(22)
    // ---- B = J0' N: item (pair of columns of J0, 4 equalities).  The k range is the same for the whole wave (J0 is upper
    //      triangular and block diagonal: whatever lies outside a lane's own range is an exact zero), so every J0 read is
    //      a stride-1 row segment and every N read a broadcast.
    {
        const int ncg = (m + 3) >> 2;
        const int cp = tid / ncg, cg = tid - cp * ncg;
        const int c0 = 2 * cp, c1 = min(c0 + 1, n - 1);
        const bool act = c0 < n;
        int kmin = act ? blk_begin(c0, nv) : n, kmax = act ? c1 + 1 : 0;
        kmin = wave_min_int(kmin);
        kmax = wave_max_int(kmax);
        double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        const int c0s = act ? c0 : 0, c1s = act ? c1 : 0;
        tile2x4(c.J, c0s, c1s, ldj, Nm + 4 * cg, ldb, kmin, kmax, acc);
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * cg + q < m) {
                    Bm[c0 * ldb + 4 * cg + q] = acc[0][q];
                    if (c0 + 1 < n) Bm[(c0 + 1) * ldb + 4 * cg + q] = acc[1][q];
                }
        }
    }
    bsync();
    STAMP(5)
    // ---- Householder QR of B (columns in registers, one barrier per column) and J <- J Q in its shadow (rows in registers)
    if (!qr_resident(c, Bm, c.s, c.s + 160)) return false; // redundant equalities
    bsync();
    STAMP(6)
    // ---- y = R'^-1 rhs on one wave
    if (c.wave == 0) {
        if (m <= 12) solve_y<12>(c, rhs);
        else if (m <= 20) solve_y<20>(c, rhs);
        else solve_y<24>(c, rhs);
    }
    bsync();
    STAMP(19)
    // ---- x = x0 + J[:, :m] y ; f += y'y / 2
    {
        double yy = 0.0;
        if (tid < m) yy = rhs[tid] * rhs[tid];
        if (tid >= 128 && tid - 128 < n) {
            const int kk = tid - 128;
            const double* Jr = c.J + kk * ldj;
            double acc = 0.0;
            for (int e = 0; e < m; ++e) acc = fma(Jr[e], rhs[e], acc);
            c.x[kk] += acc;
        }
        yy = block_sum(c, yy);
        f_value += 0.5 * yy;
    }
    c.iq = m;
    bsync();
    return true;
}


// ------------------------------------------------------------------------------------------------
// Blocked elimination H -> (pivots, Y) with J = U^-1 = Y diag(1/sqrt(pivot)), four pivots per synchronisation.
// Ownership: a G x G thread grid (G = 1 << LG); thread (ta, te) keeps positions (ta + G u, te + G w), u <= w < NU, of H
// (h) and of Y (y, from zero) in registers for the whole factorisation.  S = [H; Y] (Y from the identity) is reduced by
// column operations: with P the four pivot columns, S(:,k) -= S(:,P) H_PP^-1 H_Pk for every later column k.  Only the
// four pivot rows of H (rb[column][p]) and the four pivot columns of Y (yb[row][p]) travel through LDS, RAW, once per
// panel; every thread factors the 4 x 4 pivot block itself (H_PP = U~' D U~, U~ unit upper triangular, four chained
// reciprocals) and brings its own slices to the state a pivot-by-pivot elimination would have published
// (x' = U~^-T x).  JB = panel start / G is a compile-time constant: only h[u >= JB][.] and y[u <= JB][w >= JB] change.
// WLOCAL: the grid is one wavefront -- LDS operations of a wave execute in order, no workgroup barrier is needed and
// one buffer suffices; otherwise one barrier per panel and two buffers.  Positions past the matrix must hold the identity.
// ------------------------------------------------------------------------------------------------
template <int LG, int NU, bool WLOCAL, int UU>
__device__ __forceinline__ void publish_panel(Ctx& c, double (&h)[NU][NU], double (&y)[NU][NU], int ta, int te, int j0n,
                                              double* RB, double* YB)
{
    constexpr int G = 1 << LG, PS = NU * G * 4;
    const int par = WLOCAL ? 0 : ((j0n >> 2) & 1);
    const int grp = (j0n & (G - 1)) >> 2;
    if ((ta >> 2) == grp) { // rows j0n + p, p = ta & 3
        double* dst = RB + par * PS + (ta & 3);
#pragma unroll
        for (int w = UU; w < NU; ++w) dst[(te + G * w) * 4] = h[UU][w];
    }
    if ((te >> 2) == grp) { // columns j0n + p of Y, p = te & 3
        const int pp = te & 3;
        double* dst = YB + par * PS + pp;
#pragma unroll
        for (int u = 0; u < UU; ++u) dst[(ta + G * u) * 4] = y[u][UU];
        const int r = ta + G * UU;
        dst[r * 4] = (r < j0n) ? y[UU][UU] : ((r == j0n + pp) ? 1.0 : 0.0);
    }
}

template <int LG, int NU, bool WLOCAL, int JB>
__device__ __forceinline__ void eliminate_block(Ctx& c, double (&h)[NU][NU], double (&y)[NU][NU], int ta, int te, int npad,
                                                double* RB, double* YB, double* dinv, bool dwriter, int dp)
{
    constexpr int G = 1 << LG, PS = NU * G * 4;
    constexpr int JN = (JB + 1 < NU) ? JB + 1 : JB;
    const int jend = min(G * JB + G, npad);
    for (int j0 = G * JB; j0 < jend; j0 += 4) {
        if (WLOCAL) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
        const int par = WLOCAL ? 0 : ((j0 >> 2) & 1);
        const double* rb = RB + par * PS;
        const double* yb = YB + par * PS;
        // operands: pivot block, this thread's row-role and column-role slices, its rows of Y
        double2v hq[4][2], fa[NU][2], fe[NU][2], fr[NU][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            hq[q][0] = ld2(rb + (j0 + q) * 4);
            hq[q][1] = ld2(rb + (j0 + q) * 4 + 2);
        }
#pragma unroll
        for (int u = JB; u < NU; ++u) {
            fe[u][0] = ld2(rb + (te + G * u) * 4);
            fe[u][1] = ld2(rb + (te + G * u) * 4 + 2);
            fa[u][0] = ld2(rb + (ta + G * u) * 4);
            fa[u][1] = ld2(rb + (ta + G * u) * 4 + 2);
        }
#pragma unroll
        for (int u = 0; u <= JB; ++u) {
            fr[u][0] = ld2(yb + (ta + G * u) * 4);
            fr[u][1] = ld2(yb + (ta + G * u) * 4 + 2);
        }
        // H_PP = U~' D U~ : H(p,q) = hq[q][p>>1][p&1] for p <= q
        const double a0 = hq[0][0].x, i0 = fast_rcp(a0);
        const double u01 = hq[1][0].x * i0, u02 = hq[2][0].x * i0, u03 = hq[3][0].x * i0;
        const double a1 = fma(-u01, hq[1][0].x, hq[1][0].y), i1 = fast_rcp(a1);
        const double t12 = fma(-u01, hq[2][0].x, hq[2][0].y), t13 = fma(-u01, hq[3][0].x, hq[3][0].y);
        const double u12 = t12 * i1, u13 = t13 * i1;
        const double a2 = fma(-u12, t12, fma(-u02, hq[2][0].x, hq[2][1].x)), i2 = fast_rcp(a2);
        const double t23 = fma(-u12, t13, fma(-u02, hq[3][0].x, hq[3][1].x));
        const double u23 = t23 * i2;
        const double a3 = fma(-u23, t23, fma(-u13, t13, fma(-u03, hq[3][0].x, hq[3][1].y))), i3 = fast_rcp(a3);
        auto xform = [&](double2v (&x)[2]) __attribute__((always_inline)) { // x' = U~^-T x (also g' = g U~^-1)
            x[0].y = fma(-u01, x[0].x, x[0].y);
            x[1].x = fma(-u12, x[0].y, fma(-u02, x[0].x, x[1].x));
            x[1].y = fma(-u23, x[1].x, fma(-u13, x[0].y, fma(-u03, x[0].x, x[1].y)));
        };
#pragma unroll
        for (int u = JB; u < NU; ++u) {
            xform(fa[u]);
            xform(fe[u]);
            fe[u][0].x *= i0; fe[u][0].y *= i1; fe[u][1].x *= i2; fe[u][1].y *= i3;
        }
        if (te + G * JB < j0 + 4) { // columns up to the end of the panel take no update
            fe[JB][0].x = 0.0; fe[JB][0].y = 0.0; fe[JB][1].x = 0.0; fe[JB][1].y = 0.0;
        }
#pragma unroll
        for (int u = 0; u <= JB; ++u) xform(fr[u]);
#pragma unroll
        for (int u = JB; u < NU; ++u)
#pragma unroll
            for (int w = u; w < NU; ++w)
                h[u][w] = fma(-fa[u][1].y, fe[w][1].y, fma(-fa[u][1].x, fe[w][1].x,
                          fma(-fa[u][0].y, fe[w][0].y, fma(-fa[u][0].x, fe[w][0].x, h[u][w]))));
#pragma unroll
        for (int u = 0; u <= JB; ++u)
#pragma unroll
            for (int w = JB; w < NU; ++w)
                y[u][w] = fma(-fr[u][1].y, fe[w][1].y, fma(-fr[u][1].x, fe[w][1].x,
                          fma(-fr[u][0].y, fe[w][0].y, fma(-fr[u][0].x, fe[w][0].x, y[u][w]))));
        // the pivot columns of Y themselves are final now
        if ((te >> 2) == ((j0 & (G - 1)) >> 2)) {
            const int pp = te & 3;
#pragma unroll
            for (int u = 0; u <= JB; ++u) {
                const double lo = (pp & 1) ? fr[u][0].y : fr[u][0].x;
                const double hi = (pp & 1) ? fr[u][1].y : fr[u][1].x;
                y[u][JB] = (pp & 2) ? hi : lo;
            }
        }
        if (dwriter) { // 1/sqrt(pivot)
            const double lo = (dp & 1) ? a1 : a0, hi = (dp & 1) ? a3 : a2;
            dinv[j0 + dp] = rsqrt((dp & 2) ? hi : lo);
        }
        if (j0 + 4 < npad) {
            if (j0 + 4 < G * JB + G) publish_panel<LG, NU, WLOCAL, JB>(c, h, y, ta, te, j0 + 4, RB, YB);
            else publish_panel<LG, NU, WLOCAL, JN>(c, h, y, ta, te, j0 + 4, RB, YB);
        }
    }
    if constexpr (JB + 1 < NU) {
        if (npad > G * (JB + 1)) eliminate_block<LG, NU, WLOCAL, JB + 1>(c, h, y, ta, te, npad, RB, YB, dinv, dwriter, dp);
    }
}

// ------------------------------------------------------------------------------------------------
// one QP on one workgroup of 256 threads
// ------------------------------------------------------------------------------------------------
template <typename TI>
__device__ __forceinline__ void solve_one(const GroupArgs<TI>& ga, const DevStruct& S, const int b, double* lds)
{
    const int tid = threadIdx.x;
    Ctx c;
    c.S = &S;
    c.tid = tid;
    c.lane = tid & (kWave - 1);
    c.wave = uni(tid >> 6);
    c.rslot = 0;
    c.nv = S.nv; c.na = S.na; c.nc = S.nc; c.k = S.k; c.n = S.n; c.nu = S.nu;
    c.neq = S.neq; c.nin2 = S.nin2; c.ldj = S.ldj; c.ldm = S.ldm; c.ldc = S.ldc;
    c.J = lds + S.o_J; c.R = lds + S.o_R; c.M = lds + S.o_M; c.Jc = lds + S.o_Jc; c.Ac = lds + S.o_Ac;
    {
        double* vec = lds + S.o_vec;
        c.h = vec + V_H * kSlot; c.x = vec + V_X * kSlot; c.np = vec + V_NP * kSlot; c.d = vec + V_D * kSlot;
        c.z = vec + V_Z * kSlot; c.xold = vec + V_XOLD * kSlot; c.r = vec + V_R * kSlot; c.u = vec + V_U * kSlot;
        c.uold = vec + V_UOLD * kSlot; c.q = vec + V_Q * kSlot; c.g = vec + V_G * kSlot; c.w = vec + V_W * kSlot;
        c.wrow = vec + V_WROW * kSlot; c.blb = vec + V_BLB * kSlot; c.bub = vec + V_BUB * kSlot; c.tl = vec + V_TL * kSlot;
        c.tu = vec + V_TU * kSlot; c.bc = vec + V_BC * kSlot; c.rdinv = vec + V_RDINV * kSlot; c.dinv = vec + V_DINV * kSlot;
        c.red = vec + V_RED * kSlot; c.prm = vec + V_PRM * kSlot; c.b1 = vec + V_B1 * kSlot; c.s = vec + V_S * kSlot;
        c.stash = vec + V_STASH * kSlot; c.part = vec + V_PART * kSlot;
    }
    c.eqw = lds + S.o_eqw; c.eqt = lds + S.o_eqt; c.ldb = S.ldb;
    int* ia = reinterpret_cast<int*>(lds + S.o_int);
    const int n = c.n, nv = c.nv, na = c.na, nc = c.nc, k = c.k, nu = c.nu, neq = c.neq, nin2 = c.nin2;
    const int ldj = c.ldj, ldm = c.ldm, ldc = c.ldc;
    c.A = ia + kIntA; c.Aold = ia + kIntAold; c.gskip = ia + kIntGskip; c.iai = ia + kIntIai; c.iaexcl = ia + kIntIaexcl;
    c.meta = ia + kIntMeta;
    c.iq = 0;
    c.R_norm = 1.0;

    const int n_dense = S.n_dense, n_sel = S.n_sel, n_bound = S.n_bound, r1 = S.r1, n_tasks = S.n_tasks;
    const size_t qp = (size_t)b;
    double* As = c.R;  // dense task rows are staged in the (not yet used) R region

    STAMP_DECL
    // ---------------- phase 0: one pass over the QP's HBM record, every load in flight before the first use -------
    // (measured: the copy loop per array cost one HBM latency per array and per 4 elements -- 16 exposed round trips)
    const int lenM = nv * (nv + 1) / 2, lenA = n_dense * nv, lenAc = nc * 6 * nv, lenT = nc * 72;
    {
        constexpr int RM = 9, RA = 9, RC = 4, RT = 2; // rounds of 256 covered by registers; longer arrays finish in tail loops
        const TI* pM = ga.M + qp * lenM;
        const TI* pA = ga.A + qp * (size_t)lenA;
        const TI* pAc = ga.Ac + qp * (size_t)lenAc;
        TI vM[RM], vA[RA], vC[RC];
        double vT[RT];
        unsigned vP[RM];
        ld_regs<TI, RM>(pM, lenM, tid, vM);
        ld_regs<unsigned, RM>(S.mpack, lenM, tid, vP);
        unsigned vQ[RA];
        if (lenA > 0) {
            ld_regs<TI, RA>(pA, lenA, tid, vA);
            ld_regs<unsigned, RA>(S.apack, lenA, tid, vQ);
        }
        if (nc > 0) {
            ld_regs<TI, RC>(pAc, lenAc, tid, vC);
            ld_regs<double, RT>(S.force_gen, lenT, tid, vT);
        }
        // the short vectors: one (clamped) element per thread each
        const TI vh = ga.h[qp * nv + min(tid, nv - 1)];
        const TI vb1 = ga.b1[qp * r1 + min(tid, r1 - 1)];
        const TI vw = ga.w[qp * n_tasks + min(tid, n_tasks - 1)];
        TI vbc = TI(0), vbl = TI(0), vbu = TI(0), vtl = TI(0), vtu = TI(0), vha = TI(0);
        if (nc > 0) vbc = ga.bc[qp * (nc * 6) + min(tid, nc * 6 - 1)];
        if (n_bound > 0) {
            vbl = ga.blb[qp * n_bound + min(tid, n_bound - 1)];
            vbu = ga.bub[qp * n_bound + min(tid, n_bound - 1)];
        }
        if (S.act_bounds) {
            vtl = ga.tlb[qp * na + min(tid, na - 1)];
            vtu = ga.tub[qp * na + min(tid, na - 1)];
            vha = ga.h[qp * nv + nu + min(tid, na - 1)];
        }
        const int meta0 = (nin2 > 0) ? S.rowmeta[min(tid, nin2 - 1)] : 0;
        const int meta1 = (nin2 > 0) ? S.rowmeta[min(tid + kThreads, nin2 - 1)] : 0;
        const int drt = (n_dense > 0) ? S.dense_row_task[min(tid, n_dense - 1)] : 0;
        // selection rows (posture) and force-regularisation right-hand sides: constants now, arithmetic after the barrier
        int selc = 0, selt = 0, frt = 0;
        double ftc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (n_sel > 0) {
            selc = S.sel_col[min(tid, n_sel - 1)];
            selt = S.sel_task[min(tid, n_sel - 1)];
        }
        if (nc > 0) {
            const int fm = min(tid, k - 1);
            frt = S.forcereg_task[fm / 12];
#pragma unroll
            for (int qd = 0; qd < 6; ++qd) ftc[qd] = S.ft[(fm / 12) * 72 + (fm % 12) * 6 + qd];
        }
        // while the record is on its way: J starts from zero (only the factorisation's final writes touch it)
        for (int e = tid; e < n * ldj; e += kThreads) c.J[e] = 0.0;
        // ---- land: packed M goes straight to both triangles of the full matrix (offsets from the structure's table)
#pragma unroll
        for (int u = 0; u < RM; ++u) {
            const int e = tid + u * kThreads;
            if (e < lenM) {
                const double v = (double)vM[u];
                c.M[vP[u] & 0xffffu] = v;
                c.M[vP[u] >> 16] = v;
            }
        }
        if (lenA > 0) {
            // task rows land transposed inside each row: the four columns ta, ta + 16, .. a thread of the 16 x 16 grid
            // needs are one 32-byte group (two 16-byte reads instead of four 8-byte ones).  Columns past nv are never
            // written: what is read there only reaches positions that the identity padding below overwrites
#pragma unroll
            for (int u = 0; u < RA; ++u) {
                const int e = tid + u * kThreads;
                if (e < lenA) As[vQ[u]] = (double)vA[u];
            }
        }
        if (nc > 0) {
            st_regs<TI, RC>(c.Ac, lenAc, tid, vC);
            st_regs<double, RT>(c.eqw, lenT, tid, vT); // force generators staged in the (still unused) equality scratch
            if (tid < nc * 6) c.bc[tid] = (double)vbc;
        }
        if (tid < nv) c.h[tid] = (double)vh;
        if (tid < r1) c.b1[tid] = (double)vb1;
        if (tid < n_tasks) c.w[tid] = (double)vw;
        if (tid < n_bound) {
            c.blb[tid] = (double)vbl;
            c.bub[tid] = (double)vbu;
        }
        if (S.act_bounds && tid < na) { // lb - h_a, ub - h_a (computeProblemData, actuation tasks)
            c.tl[tid] = (double)vtl - (double)vha;
            c.tu[tid] = (double)vtu - (double)vha;
        }
        if (tid < nin2) c.meta[tid] = meta0;
        if (tid + kThreads < nin2) c.meta[tid + kThreads] = meta1;
        c.iai[tid] = drt; // parked until w has landed (iai is initialised in phase 4)
        // tails of arrays longer than the register rounds (none for the humanoid stacks)
        for (int e = tid + RM * kThreads; e < lenM; e += kThreads) {
            const unsigned pk = S.mpack[e];
            const double v = (double)pM[e];
            c.M[pk & 0xffffu] = v;
            c.M[pk >> 16] = v;
        }
        for (int e = tid + RA * kThreads; e < lenA; e += kThreads) As[S.apack[e]] = (double)pA[e];
        for (int e = tid + RC * kThreads; e < lenAc; e += kThreads) c.Ac[e] = (double)pAc[e];
        for (int e = tid + RT * kThreads; e < lenT; e += kThreads) c.eqw[e] = S.force_gen[e];
        for (int i = tid + 2 * kThreads; i < nin2; i += kThreads) c.meta[i] = S.rowmeta[i];
        if (tid < nv) { // diagonal additions / right-hand sides of the selection rows
            c.z[tid] = 0.0;
            c.d[tid] = 0.0;
        }
        bsync();
        if (tid < n_dense) { // (row weight, right-hand side) pairs behind the staged rows: one 16-byte read per row
            As[n_dense * 64 + 2 * tid] = c.w[c.iai[tid]];
            As[n_dense * 64 + 2 * tid + 1] = c.b1[tid];
        }
        // selection rows (posture): H(c,c) += w, g(c) -= w b  (distinct columns)
        for (int sidx = tid; sidx < n_sel; sidx += kThreads) {
            const int col = (sidx == tid) ? selc : S.sel_col[sidx];
            const double wt = c.w[(sidx == tid) ? selt : S.sel_task[sidx]];
            c.z[col] = wt;
            c.d[col] = wt * c.b1[n_dense + sidx];
        }
        // force regularisation: g_f = -w F' b
        if (tid < k) {
            const double* bb = c.b1 + n_dense + n_sel + 6 * (tid / 12);
            double sacc = 0.0;
#pragma unroll
            for (int qd = 0; qd < 6; ++qd) sacc = fma(ftc[qd], bb[qd], sacc);
            c.g[nv + tid] = -c.w[frt] * sacc;
        }
        // Jc = T' A_c  (12 x nv per contact): thread = (row m of Jc, every G-th column), its six T coefficients in registers
        if (k > 0) {
            const int G = kThreads / k;
            const int m = tid % k, jg = tid / k;
            if (jg < G) {
                const int ct = m / 12, mm = m - 12 * ct;
                const double* T = c.eqw + ct * 72 + mm;
                const double* Acc = c.Ac + ct * 6 * nv;
                double tc[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) tc[r] = T[r * 12];
#pragma unroll 2
                for (int j = jg; j < nv; j += G) {
                    double sacc = 0.0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) sacc = fma(tc[r], Acc[r * nv + j], sacc);
                    c.Jc[m * ldc + j] = sacc;
                }
            }
        }
    }
    bsync();
    STAMP(0)

    // ---------------- phases 1-2b in registers: H assembly, Cholesky H = U'U, J = U^-1 ----------------
    // Thread (ta, te) of a 16 x 16 grid OWNS the strided positions (ta + 16u, te + 16w) of the upper triangle of the
    // dv block for the whole pipeline: H is accumulated, factorised and inverted in its registers (eliminate_block).
    double c1, c2;
    { // nv <= 64 is checked on the host
        const int ta = tid >> 4, te = tid & 15;
        double h[4][4];
        double trace = 0.0;
        {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) h[u][w] = 0.0;
            // rows of the next task line are in flight while this one multiplies; g_j = -sum_r w_r A(r,j) b(r) rides along
            double gacc[4] = {0.0, 0.0, 0.0, 0.0};
            const double* Ai = As + ta * 4;
            const double* Aj = As + te * 4;
            const double* WB = As + n_dense * 64;
            auto ldrow = [&](int r, double2v (&ai)[2], double2v (&aj)[2], double2v& wb) __attribute__((always_inline)) {
                ai[0] = ld2(Ai + r * 64);
                ai[1] = ld2(Ai + r * 64 + 2);
                aj[0] = ld2(Aj + r * 64);
                aj[1] = ld2(Aj + r * 64 + 2);
                wb = ld2(WB + 2 * r);
            };
            auto macrow = [&](const double2v (&ai)[2], const double2v (&aj)[2], const double2v& wb) __attribute__((always_inline)) {
                const double a[4] = {ai[0].x, ai[0].y, ai[1].x, ai[1].y};
                const double ajw[4] = {aj[0].x * wb.x, aj[0].y * wb.x, aj[1].x * wb.x, aj[1].y * wb.x};
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int w = u; w < 4; ++w) h[u][w] = fma(a[u], ajw[w], h[u][w]);
#pragma unroll
                for (int w = 0; w < 4; ++w) gacc[w] = fma(ajw[w], wb.y, gacc[w]);
            };
            if (n_dense > 0) {
                double2v ai0[2], aj0[2], ai1[2], aj1[2], wb0, wb1;
                ldrow(0, ai0, aj0, wb0);
                int r = 0;
                for (; r + 2 <= n_dense; r += 2) {
                    ldrow(r + 1, ai1, aj1, wb1);
                    macrow(ai0, aj0, wb0);
                    ldrow(min(r + 2, n_dense - 1), ai0, aj0, wb0);
                    macrow(ai1, aj1, wb1);
                }
                if (r < n_dense) macrow(ai0, aj0, wb0);
            }
            if (ta == 0) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int col = te + 16 * w;
                    if (col < nv) c.g[col] = -gacc[w] - c.d[col];
                }
            }
            if (ta == te) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = ta + 16 * u;
                    if (i < nv) {
                        h[u][u] += c.z[i] + S.hessian_reg;
                        trace += h[u][u];
                    }
                }
            }
        }
        STAMP(1)
        // positions past nv: identity, so that the pivots of the padded last panel are inert
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = u; w < 4; ++w) {
                const int r = ta + 16 * u, q = te + 16 * w;
                if (r >= nv || q >= nv) h[u][w] = (r == q) ? 1.0 : 0.0;
            }
        // ---- H -> J = U^-1 by blocked elimination (eliminate_block), four pivots per barrier
        double y[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) y[u][w] = 0.0;
        // the force-regularisation blocks H_ff = w F'F + reg I (12 x 12 per contact) go one per wave afterwards; their
        // constants are fetched now so that the loads overlap the dv block
        const int la = c.lane >> 3, le = c.lane & 7;
        double hF[2][2], yF[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        {
            const int cs = (c.wave < nc) ? c.wave : 0;
            const double* ftf = S.ftf + cs * 144;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int w = 0; w < 2; ++w) hF[u][w] = (nc > 0) ? ftf[min(la + 8 * u, 11) * 12 + min(le + 8 * w, 11)] : 0.0;
        }
        publish_panel<4, 4, false, 0>(c, h, y, ta, te, 0, c.s, c.stash);
        eliminate_block<4, 4, false, 0>(c, h, y, ta, te, (nv + 3) & ~3, c.s, c.stash, c.dinv, tid >= 128 && tid < 132, tid & 3);
        STAMP(2)
        bsync();
        // final: J(r,q) = Y(r,q) dinv[q], J(r,r) = dinv[r]
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = u; w < 4; ++w) {
                const int r = ta + 16 * u, q = te + 16 * w;
                if (q < nv && r < q) c.J[r * ldj + q] = y[u][w] * c.dinv[q];
                else if (r == q && r < nv) c.J[r * ldj + r] = c.dinv[r];
            }
        // ---- force blocks: wave-local (8 x 8 lane grid, 2 x 2 positions per lane), no workgroup barrier inside
        for (int ct = c.wave; ct < nc; ct += kWaves) {
            const int fb = nv + 12 * ct;
            const double wt = c.w[S.forcereg_task[ct]];
            if (ct != c.wave) { // contacts beyond the first four: fetch now
                const double* ftf = S.ftf + ct * 144;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int w = 0; w < 2; ++w) hF[u][w] = ftf[min(la + 8 * u, 11) * 12 + min(le + 8 * w, 11)];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int r = la + 8 * u, q = le + 8 * w;
                    if (r < 12 && q < 12) {
                        hF[u][w] = wt * hF[u][w] + ((r == q) ? S.hessian_reg : 0.0);
                        if (r == q) trace += hF[u][w];
                    }
                    else hF[u][w] = (r == q) ? 1.0 : 0.0;
                    yF[u][w] = 0.0;
                }
            double* RBf = c.s + c.wave * 128;
            double* YBf = RBf + 64;
            publish_panel<3, 2, true, 0>(c, hF, yF, la, le, 0, RBf, YBf);
            eliminate_block<3, 2, true, 0>(c, hF, yF, la, le, 12, RBf, YBf, c.dinv + fb, c.lane < 4, c.lane & 3);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int w = u; w < 2; ++w) {
                    const int r = la + 8 * u, q = le + 8 * w;
                    if (q < 12 && r < q) c.J[(fb + r) * ldj + fb + q] = yF[u][w] * c.dinv[fb + q];
                    else if (r == q && r < 12) c.J[(fb + r) * ldj + fb + r] = c.dinv[fb + r];
                }
        }
        bsync();
        c1 = block_sum(c, trace);
        double tr2 = 0.0;
        for (int i = tid; i < n; i += kThreads) tr2 += c.dinv[i];
        c2 = block_sum(c, tr2);
    }
    STAMP(3)

    // ---------------- x = -H^-1 g = -J (J' g); f = 0.5 g'x ----------------
    // J = U^-1 is still upper triangular and block diagonal here: both products run over the structural range only, two
    // lanes per column / row whose halves meet by DPP (2 n <= 256).
    double f_value;
    {
        const int idx = tid >> 1, hf = tid & 1;
        const int ic = min(idx, n - 1);
        {
            const int kb0 = blk_begin(ic, nv), len = ic + 1 - kb0, hl = (len + 1) >> 1;
            const int ka = kb0 + hf * hl, kb = hf ? ic + 1 : kb0 + hl;
            double dv = dot8(c.J + ic, ldj, c.g, 1, ka, kb);
            dv += dpp_get<0xB1>(dv);
            if (hf == 0 && idx < n) c.d[idx] = dv;
        }
        for (int i = tid; i < n + 2; i += kThreads) {
            c.u[i] = 0.0;
            c.A[i] = 0;
        }
        c.iq = 0;
        bsync();
        double part = 0.0;
        {
            const int ce = blk_end(ic, nv), len = ce - ic, hl = (len + 1) >> 1;
            const int ca = ic + hf * hl, cb = hf ? ce : ic + hl;
            double zv = dot8(c.J + ic * ldj, 1, c.d, 1, ca, cb);
            zv += dpp_get<0xB1>(zv);
            if (hf == 0 && idx < n) {
                c.z[idx] = zv;
                c.x[idx] = -zv;
                part = 0.5 * c.g[idx] * (-zv);
            }
        }
        f_value = block_sum(c, part);
    }
    STAMP(4)

    const double eps = 2.220446049250313e-16;
    const double inf = __builtin_huge_val();
    int status = -2; // running
    int iter = 0;

    // ---------------- phase 3: equality constraints ----------------
    const bool blocked_eq = (neq >= 1 && neq <= 22 && n <= 80);
    if (blocked_eq) {
        if (!equality_phase_blocked(c, f_value)) status = HQP_ERROR; // redundant equalities
        STAMP(8)
    }
    for (int i = 0; i < neq && status == -2 && !blocked_eq; ++i) {
        int k0, k1;
        double ce0;
        build_eq_row(c, i, k0, k1, ce0);
        bsync();
        compute_d(c, k0, k1);
        STAMP(5)
        update_z_r(c, 0);
        STAMP(6)
        double zz = 0.0, znp = 0.0, npx = 0.0, dn2 = 0.0;
        if (tid < n) {
            const double zv = c.z[tid];
            zz = zv * zv;
            if (tid >= c.iq) dn2 = c.d[tid] * c.d[tid];
            if (tid >= k0 && tid < k1) {
                const double nv_ = c.np[tid];
                znp = zv * nv_;
                npx = nv_ * c.x[tid];
            }
        }
        block_sum4(c, zz, znp, npx, dn2);
        double t2 = 0.0;
        if (fabs(zz) > eps) t2 = (-npx - ce0) / znp;
        const int iq = c.iq;
        if (tid < n) c.x[tid] = fma(t2, c.z[tid], c.x[tid]);
        if (tid >= 128 && tid - 128 < iq) c.u[tid - 128] = fma(-t2, c.r[tid - 128], c.u[tid - 128]);
        if (tid == kThreads - 1) {
            c.u[iq] = t2;
            c.A[i] = -i - 1;
        }
        f_value += 0.5 * (t2 * t2) * znp;
        STAMP(7)
        if (!add_constraint_hh(c, dn2)) status = HQP_ERROR; // redundant equalities
        STAMP(8)
    }

    // ---------------- phase 4: inequality loop (GI steps 1, 2, 2a-2c) ----------------
    // The common iteration (full step, constraint added) takes five barriers:
    //   P1  every thread evaluates the rows of s it owns, psi and the most violated row meet in one reduction     | B1
    //   P2  d = J'n straight from the row's sources (no staging of n), lanes of one column meet by DPP            | B2
    //   P3  waves 0-2: z = J2 d2 (lanes of one row meet by DPP) + z'z, z'n, |d2|^2; wave 3: r = R^-1 d, t1        | B3
    //   P4  step lengths from the slots; w = tau (z - alpha J(:,iq)); tau' = A_act (x + t z) for the next P1        | B4
    //   P5  J -= w v', new column of R, x, u, A, iai                                                               | B5
    // Partial steps, dual-only steps and rejected (dependent) constraints keep the plain barrier-per-phase code.
    bool tau_stale = true; // tau' = A_act x (act_rows) not current
    if (status == -2) {
        for (int i = tid; i < nin2; i += kThreads) c.iai[i] = i;
        // friction tables: LDS copy in the (now free) equality scratch when they fit
        const double *fmat = S.fric_mat, *flb = S.fric_lb, *fub = S.fric_ub;
        if (S.fric_lds && nc > 0) {
            double* tb = c.eqw;
            // (fetching these early into registers and parking them across the equality phase was tried: a diagnostic
            //  build then went wrong non-deterministically under the added register pressure -- not kept)
            for (int e = tid; e < nc * 204; e += kThreads) tb[e] = S.fric_mat[e];
            for (int e = tid; e < nc * 17; e += kThreads) {
                tb[nc * 204 + e] = S.fric_lb[e];
                tb[nc * 221 + e] = S.fric_ub[e];
            }
            fmat = tb;
            flb = tb + nc * 204;
            fub = tb + nc * 221;
            bsync();
        }
        OwnRows own;
        own_rows_init(c, own, fmat, flb, fub);
        double* tact = c.part + 512;
        if (na > 0) act_rows(c, tact, 0.0);
        bsync();
        const double psi_tol = (double)nin2 * eps * c1 * c2 * 100.0;
        bool redo_l2 = false;
        tau_stale = false;
        while (status == -2) {
            ValIdx best;
            if (!redo_l2) {
                // l1
                ++iter;
                if (iter >= S.max_iter) {
                    status = HQP_MAX_ITER;
                    break;
                }
                if (tau_stale) {
                    if (na > 0) act_rows(c, tact, 0.0);
                    bsync();
                    tau_stale = false;
                }
                for (int i = tid; i < c.iq; i += kThreads) {
                    c.uold[i] = c.u[i];
                    c.Aold[i] = c.A[i];
                }
                for (int i = tid; i < n; i += kThreads) c.xold[i] = c.x[i];
                double psi;
                own_rows_eval(c, own, tact, psi, best);
                psi = wave_sum(psi);
                best = wave_argmin(best);
                double* slot = c.red + c.rslot * 16;
                if (c.lane == 0) {
                    slot[c.wave] = psi;
                    slot[4 + c.wave] = best.v;
                    slot[8 + c.wave] = __hiloint2double(0, best.i);
                }
                bsync(); // B1
                psi = (slot[0] + slot[1]) + (slot[2] + slot[3]);
                best = ValIdx{slot[4], __double2loint(slot[8])};
#pragma unroll
                for (int w = 1; w < kWaves; ++w) best = vi_min(best, ValIdx{slot[4 + w], __double2loint(slot[8 + w])});
                c.rslot ^= 1;
                if (fabs(psi) <= psi_tol) {
                    status = HQP_OPTIMAL;
                    break;
                }
                STAMP(9)
            }
            else {
                // l2 again after a rejected constraint: s is still valid, the rejected row is excluded
                best = ValIdx{0.0, 0x7fffffff};
                for (int i = tid; i < nin2; i += kThreads) {
                    const double sv = c.s[i];
                    if (sv < 0.0 && c.iai[i] != -1 && c.iaexcl[i]) best = vi_min(best, ValIdx{sv, i});
                }
                best = block_argmin(c, best);
                redo_l2 = false;
            }
            if (best.v >= 0.0) {
                status = HQP_OPTIMAL;
                break;
            }
            const int ip = best.i;
            // the row n of constraint ip: kind, support [k0, k1), sign; n itself is staged for z'n and the slow paths
            const int mt = c.meta[ip];
            const int kind = mt & 3, rr = (mt >> 3) & 255, ct = (mt >> 11) & 15, col = (mt >> 15) & 255;
            const double sg = ((mt >> 2) & 1) ? -1.0 : 1.0;
            int k0, k1;
            if (kind == INEQ_BOUNDS) {
                k0 = col;
                k1 = col + 1;
                if (tid == 0) c.np[col] = sg;
            }
            else if (kind == INEQ_ACTUATION) {
                k0 = 0;
                k1 = n;
                const int row = nu + rr;
                if (tid < nv) c.np[tid] = sg * c.M[row * ldm + tid];
                else if (tid < n) c.np[tid] = -sg * c.Jc[(tid - nv) * ldc + row];
            }
            else {
                k0 = nv + 12 * ct;
                k1 = k0 + 12;
                if (tid < 12) c.np[k0 + tid] = sg * fmat[(ct * 17 + rr) * 12 + tid];
            }
            if (tid == kThreads - 1) {
                c.u[c.iq] = 0.0;
                c.A[c.iq] = ip;
            }
            STAMP(10)

            // l2a
            while (true) {
                const int iq = c.iq;
                // ---- P2: d = J' n
                if (kind == INEQ_BOUNDS) {
                    if (tid < n) c.d[tid] = sg * c.J[col * ldj + tid];
                }
                else if (kind == INEQ_FORCE) {
                    if (tid < n) {
                        const double* F = fmat + (ct * 17 + rr) * 12;
                        const double* Jb = c.J + k0 * ldj + tid;
                        double a0 = 0.0, a1 = 0.0;
#pragma unroll
                        for (int m = 0; m < 12; m += 2) {
                            a0 = fma(F[m], Jb[m * ldj], a0);
                            a1 = fma(F[m + 1], Jb[(m + 1) * ldj], a1);
                        }
                        c.d[tid] = sg * (a0 + a1);
                    }
                }
                else { // actuation row: two lanes per column, halves of the support
                    const int idx = tid >> 1, hf = tid & 1;
                    const int ic = min(idx, n - 1);
                    const int row = nu + rr;
                    const int mid = (n + 1) >> 1;
                    const int ka = hf ? mid : 0, kb = hf ? n : mid;
                    const double* Mr = c.M + row * ldm;
                    const double* Jcol = c.J + ic;
                    const int kv = min(kb, nv), kf = max(ka, nv);
                    double acc = 0.0;
                    if (ka < kv) acc = dot8(Mr, 1, Jcol, ldj, ka, kv);
                    if (kf < kb) acc -= dot8(c.Jc + row, ldc, Jcol + nv * ldj, ldj, kf - nv, kb - nv);
                    acc += dpp_get<0xB1>(acc);
                    if (hf == 0 && idx < n) c.d[idx] = sg * acc;
                }
                bsync(); // B2
                STAMP(11)
                // ---- P3: z, r and the reductions of step 2b
                double* slot = c.red + c.rslot * 16;
                if (c.wave < 3) {
                    const int lpr = (2 * n <= 3 * kWave) ? 2 : 1; // lanes per row
                    const int idx = (lpr == 2) ? (tid >> 1) : tid, hf = (lpr == 2) ? (tid & 1) : 0;
                    const int ir = min(idx, n - 1);
                    const int span = n - iq, hlen = (lpr == 2) ? ((span + 1) >> 1) : span;
                    const int ca = iq + hf * hlen, cb = min(n, ca + hlen);
                    const double* Jr = c.J + ir * ldj;
                    double zv = dot8(Jr, 1, c.d, 1, ca, cb);
                    if (lpr == 2) zv += dpp_get<0xB1>(zv);
                    double zz = 0.0, znp = 0.0, dn2 = 0.0;
                    if (hf == 0 && idx < n) {
                        c.z[idx] = zv;
                        zz = zv * zv;
                        if (idx >= iq) {
                            const double dv = c.d[idx];
                            dn2 = dv * dv;
                        }
                        if (idx >= k0 && idx < k1) znp = zv * c.np[idx];
                    }
                    zz = wave_sum(zz);
                    znp = wave_sum(znp);
                    dn2 = wave_sum(dn2);
                    if (c.lane == 0) {
                        slot[c.wave] = zz;
                        slot[4 + c.wave] = znp;
                        slot[8 + c.wave] = dn2;
                    }
                }
                else {
                    update_r_wave(c, neq);
                    // step 2b, partial step length t1 (dual feasibility): this wave just wrote r, LDS keeps its order
                    ValIdx bt{inf, 0x7fffffff};
                    for (int kk = neq + c.lane; kk < iq; kk += kWave) {
                        const double rk = c.r[kk];
                        if (rk > 0.0) bt = vi_min(bt, ValIdx{c.u[kk] / rk, kk});
                    }
                    bt = wave_argmin(bt);
                    if (c.lane == 0) {
                        slot[12] = bt.v;
                        slot[13] = __hiloint2double(0, bt.i);
                    }
                }
                bsync(); // B3
                STAMP(12)
                // ---- P4: step lengths
                const double zz = (slot[0] + slot[1]) + slot[2], znp = (slot[4] + slot[5]) + slot[6];
                const double dn2 = (slot[8] + slot[9]) + slot[10];
                const double t1 = slot[12];
                const int lpos = __double2loint(slot[13]);
                c.rslot ^= 1;
                const int l = (t1 < inf) ? c.A[lpos] : 0;
                const double sip = c.s[ip];
                const double uiq = c.u[iq];
                const double t2 = (fabs(zz) > eps) ? (-sip / znp) : inf;
                const double t = fmin(t1, t2);
                if (t >= inf) {
                    status = HQP_INFEASIBLE; // eiquadprog UNBOUNDED (dual) -> tsid INFEASIBLE
                    break;
                }
                if (t2 >= inf) {
                    // (ii) dual step only, drop l
                    bsync(); // everyone has read s[ip], u[iq], A[lpos] before they change
                    for (int j = neq + tid; j < iq; j += kThreads) c.u[j] = fma(-t, c.r[j], c.u[j]);
                    if (tid == kThreads - 1) {
                        c.u[iq] = uiq + t;
                        c.iai[l] = l;
                    }
                    bsync();
                    STAMP(13)
                    delete_constraint(c, l);
                    STAMP(15)
                    continue;
                }
                f_value += t * znp * (0.5 * t + uiq);
                if (t == t2) {
                    // (iii) full step: add ip to the active set.  One reflector H = I - tau v v' (v = d[iq:] - alpha e_0)
                    // zeroes d[iq+1:]; J[:, iq:] v = z - alpha J[:, iq] needs no new matvec.
                    const double diq = c.d[iq];
                    double alpha = diq, v0 = 0.0, tau = 0.0;
                    const bool reflect = (iq + 1 < n && dn2 > 0.0);
                    if (reflect) {
                        const double inx = rsqrt(dn2);
                        const double nx = dn2 * inx;
                        alpha = (diq >= 0.0) ? -nx : nx;
                        v0 = diq - alpha;
                        tau = fast_rcp(fma(nx, fabs(diq), dn2));
                    }
                    const bool accepted = fabs(alpha) > eps * c.R_norm;
                    if (reflect && tid < n) c.part[tid] = tau * (c.z[tid] - alpha * c.J[tid * ldj + iq]);
                    if (na > 0) act_rows(c, tact, t); // tau' of the next iterate, x + t z formed on the fly
                    bsync(); // B4
                    STAMP(13)
                    // ---- P5
                    if (reflect) {
                        const int kr = tid & 127, half = tid >> 7;
                        if (kr < n) {
                            const int span = n - iq;
                            const int ca = iq + half * ((span + 1) >> 1), cb = half ? n : iq + ((span + 1) >> 1);
                            double* Jk = c.J + kr * ldj;
                            const double wk = c.part[kr];
                            int cc = ca;
                            if (cc == iq && cc < cb) {
                                Jk[cc] = fma(-wk, v0, Jk[cc]);
                                ++cc;
                            }
                            for (; cc + 8 <= cb; cc += 8) {
                                double dd[8], jj[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    dd[u] = c.d[cc + u];
                                    jj[u] = Jk[cc + u];
                                }
#pragma unroll
                                for (int u = 0; u < 8; ++u) Jk[cc + u] = fma(-wk, dd[u], jj[u]);
                            }
                            for (; cc < cb; ++cc) Jk[cc] = fma(-wk, c.d[cc], Jk[cc]);
                        }
                    }
                    {
                        double* Rc = c.R + roff(iq);
                        for (int i = tid; i < iq; i += kThreads) Rc[i] = c.d[i];
                        if (tid < n) c.x[tid] = fma(t, c.z[tid], c.x[tid]);
                        if (tid >= 128 + neq && tid - 128 < iq) c.u[tid - 128] = fma(-t, c.r[tid - 128], c.u[tid - 128]);
                        if (tid == kThreads - 1) {
                            Rc[iq] = alpha;
                            c.rdinv[iq] = 1.0 / alpha;
                            c.u[iq] = uiq + t;
                            if (accepted) c.iai[ip] = -1;
                        }
                    }
                    c.iq = iq + 1;
                    bsync(); // B5
                    STAMP(14)
                    if (accepted) c.R_norm = fmax(c.R_norm, fabs(alpha));
                    else {
                        // numerically dependent: take the constraint out again, back to the saved iterate, pick another
                        if (tid == 0) c.iaexcl[ip] = 0;
                        bsync();
                        delete_constraint(c, ip);
                        for (int i = tid; i < nin2; i += kThreads) c.iai[i] = i;
                        bsync();
                        for (int i = tid; i < c.iq; i += kThreads) {
                            const int av = c.Aold[i];
                            c.A[i] = av;
                            if (av >= 0) c.iai[av] = -1;
                            c.u[i] = c.uold[i];
                        }
                        for (int i = tid; i < n; i += kThreads) c.x[i] = c.xold[i];
                        bsync();
                        redo_l2 = true;
                        tau_stale = true;
                    }
                    break; // -> l1 (or l2 again)
                }
                // (iii) partial step: primal + dual step, drop l, refresh s(ip)
                bsync(); // everyone has read s[ip], u[iq], A[lpos] before they change
                if (tid < n) c.x[tid] = fma(t, c.z[tid], c.x[tid]);
                if (tid >= 128 + neq && tid - 128 < iq) c.u[tid - 128] = fma(-t, c.r[tid - 128], c.u[tid - 128]);
                if (tid == kThreads - 1) c.u[iq] = uiq + t;
                if (tid == 0) c.iai[l] = l;
                bsync();
                STAMP(13)
                tau_stale = true;
                delete_constraint(c, l);
                STAMP(15)
                {
                    double ci0;
                    if (kind == INEQ_BOUNDS) ci0 = (sg < 0.0) ? c.bub[rr] : -c.blb[rr];
                    else if (kind == INEQ_ACTUATION) ci0 = (sg < 0.0) ? c.tu[rr] : -c.tl[rr];
                    else ci0 = (sg < 0.0) ? fub[ct * 17 + rr] : -flb[ct * 17 + rr];
                    double part = 0.0;
                    for (int j = k0 + tid; j < k1; j += kThreads) part = fma(c.np[j], c.x[j], part);
                    part = block_sum(c, part);
                    if (tid == 0) c.s[ip] = part + ci0;
                    bsync();
                }
            }
        }
    }

    STAMP(16)
    // ---------------- phase 5: decode + write-out ----------------
    // tau = h_a + M_a dv - J_a' f   (getActuatorForces)
    bsync();
    TI* xo = ga.x + qp * n;
    for (int i = tid; i < n; i += kThreads) xo[i] = (TI)c.x[i];
    if (na > 0) {
        TI* to = ga.tau + qp * na;
        // tau' = M_a dv - J_a' f of the final iterate: the optimality test of the last P1 ran on exactly this vector
        double* tact = c.part + 512;
        if (status != HQP_OPTIMAL || tau_stale) {
            act_rows(c, tact, 0.0);
            bsync();
        }
        for (int rr = tid; rr < na; rr += kThreads) to[rr] = (TI)(c.h[nu + rr] + tact[rr]);
    }
    if (tid == 0) {
        ga.status[qp] = status;
        ga.iters[qp] = iter;
        if (ga.objective) ga.objective[qp] = (TI)f_value;
        if (ga.n_active) ga.n_active[qp] = c.iq;
    }
#ifdef WBCQP_STAMPS
    STAMP(17)
    if (tid == 0 && ga.dbg)
        for (int i = 0; i < kStamps; ++i) ga.dbg[qp * kStamps + i] = c.st_acc_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// the kernel: grid = total QPs, block = 256 threads = four wavefronts = one QP
// ------------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(kThreads) void solve_kernel(const GroupTable<TI> tab)
{
    extern __shared__ __align__(16) double lds[];
    int b = tab.order ? tab.order[blockIdx.x] : (int)blockIdx.x, gi = 0;
    while (gi + 1 < tab.n && b >= tab.g[gi].count) {
        b -= tab.g[gi].count;
        ++gi;
    }
    const GroupArgs<TI>& ga = tab.g[gi];
    const DevStruct& S = ga.st;
    solve_one<TI>(ga, S, b, lds);
}

// ------------------------------------------------------------------------------------------------
// Longest-first launch order.  A CU holds one QP at a time and the hardware hands out workgroups in index order, so a
// QP with many active-set iterations that starts late ends the launch late (measured on 1024 Talos QPs: 1.15 M cycles
// against 0.76 M for a perfect split; longest-first: 0.89 M).  The cost of a QP is setup + iterations x constant, and
// iteration counts change little from one control tick to the next: the counts of the launch just finished order the
// next launch of the same shape.  One workgroup, counting sort by min(iters, 63), descending.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void schedule_kernel(const ScheduleArgs sa, int* order, int total)
{
    __shared__ int hist[64];
    __shared__ int start[64];
    const int tid = threadIdx.x;
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    auto key_of = [&](int i) {
        int gi = 0, b = i;
        while (gi + 1 < sa.n && b >= sa.count[gi]) {
            b -= sa.count[gi];
            ++gi;
        }
        const int it = sa.iters[gi][b];
        return 63 - min(max(it, 0), 63);
    };
    for (int i = tid; i < total; i += 1024) atomicAdd(&hist[key_of(i)], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < 64; ++k) {
            start[k] = acc;
            acc += hist[k];
        }
    }
    __syncthreads();
    for (int i = tid; i < total; i += 1024) order[atomicAdd(&start[key_of(i)], 1)] = i;
}

// ------------------------------------------------------------------------------------------------
// After the path: Controller::_solve's use of an optimal solution, controller.cpp:250-272:
// v = dq + dt dv, q = pinocchio::integrate(model, q, dt v) for a free-flyer root + revolute joints (or revolute joints
// only), base orientation repacked from quaternion to angle * axis.  One wavefront per instance: lane j integrates
// joint j, lane 0 the SE(3) part (exp6, M0 * exp6, rotation -> quaternion, sign continuity, first-order normalisation:
// pinocchio's free-flyer integrate; Eigen's AngleAxis(quaternion)).  HBM-bound: (3 nq + 3 nv) words per instance.
// ------------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void integrate_kernel(int batch, int nv, int floating_base, double dt, const TI* __restrict__ q,
                                                        const TI* __restrict__ dq, const TI* __restrict__ x, int ldx,
                                                        const int* __restrict__ status, TI* __restrict__ q_next,
                                                        TI* __restrict__ v_next, TI* __restrict__ q_solver)
{
    const int inst = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (inst >= batch) return;
    const int nq = floating_base ? nv + 1 : nv, nqs = floating_base ? nv : nv;
    const TI* qi = q + (size_t)inst * nq;
    const TI* dqi = dq + (size_t)inst * nv;
    const TI* dvi = x + (size_t)inst * ldx;
    TI* qo = q_next + (size_t)inst * nq;
    TI* vo = v_next + (size_t)inst * nv;
    TI* so = q_solver ? q_solver + (size_t)inst * nqs : nullptr;
    const bool ok = !status || status[inst] == HQP_OPTIMAL;
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
            int i = 0;
            if (R1[4] > R1[0]) i = 1;
            if (R1[8] > R1[4 * i]) i = 2;
            const int jx = (i + 1) % 3, kx = (jx + 1) % 3;
            tt = sqrt(R1[4 * i] - R1[4 * jx] - R1[4 * kx] + 1.0);
            double qv[3];
            qv[i] = 0.5 * tt;
            tt = 0.5 / tt;
            qt[3] = (R1[3 * kx + jx] - R1[3 * jx + kx]) * tt;
            qv[jx] = (R1[3 * jx + i] + R1[3 * i + jx]) * tt;
            qv[kx] = (R1[3 * kx + i] + R1[3 * i + kx]) * tt;
            qt[0] = qv[0]; qt[1] = qv[1]; qt[2] = qv[2];
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

#endif // __HIPCC__
} // namespace wbcqp
