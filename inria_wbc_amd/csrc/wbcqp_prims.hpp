// wbcqp_prims.hpp -- wave64 primitives (DPP reductions, readlane broadcasts), the per-workgroup context, workgroup
// reductions, LDS read helpers and the two latency-tolerant inner products (tile2x4, dot8) every phase builds on.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include "wbcqp_types.hpp"

namespace wbcqp {
#ifdef __HIPCC__

// ------------------------------------------------------------------------------------------------
// wave64 primitives (DPP row operations + readlane)
// ------------------------------------------------------------------------------------------------
// The workgroup barrier of every kernel here.  __syncthreads() is release fence + s_barrier + acquire fence, and the fence's
// `s_waitcnt lgkmcnt(0)` is a "soft" wait that ROCm 7.2's waitcnt pass may drop: it did at the header of qr_unified's loop (and in
// some sixty other places), leaving ds_write_b128s of one wave in flight across the barrier behind which another wave reads them.
// On MI355X that is a real race: about one Talos QP in 50 000 came back with a stale reflector applied by one wave (found in round 4,
// when the specialised kernels shifted the timing; the results differed from run to run).  The explicit wait below (vmcnt and expcnt
// left alone: 0xC07F = lgkmcnt(0) only) cannot be dropped; tools/check_barriers.py reads the build's assembly and refuses a barrier
// without it.
__device__ __forceinline__ void bsync()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __syncthreads();
}

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
__device__ __forceinline__ double op_min(double a, double b) { return fmin(a, b); }
// two passes: the minimum value (v_min_f64 through the DPP stages), then the smallest index among the lanes that hold it --
// 40 instructions where the lexicographic (value, index) reduction in one pass took 76 (a compare / select group per stage)
__device__ __forceinline__ ValIdx wave_argmin(ValIdx a)
{
    double v = a.v;
    WBCQP_ROW_REDUCE(v, op_min)
    const double m = fmin(fmin(bcast_lane(v, 0), bcast_lane(v, 16)), fmin(bcast_lane(v, 32), bcast_lane(v, 48)));
    int i = (a.v == m) ? a.i : 0x7fffffff;
    i = min(i, dpp_movi<0xB1>(i));
    i = min(i, dpp_movi<0x4E>(i));
    i = min(i, dpp_movi<0x141>(i));
    i = min(i, dpp_movi<0x140>(i));
    const int r0 = __builtin_amdgcn_readlane(i, 0), r1 = __builtin_amdgcn_readlane(i, 16);
    const int r2 = __builtin_amdgcn_readlane(i, 32), r3 = __builtin_amdgcn_readlane(i, 48);
    return ValIdx{m, min(min(r0, r1), min(r2, r3))};
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
#ifndef WBCQP_STAMP_TID
#define WBCQP_STAMP_TID 0
#endif
#ifdef WBCQP_STAMP_DROP // (tools/drop_profile.py: finer stamps inside a drop's two phases, a drop counter)
constexpr int kStamps = 48;
#define DSTAMP(i) STAMP(i)
#define DCOUNT(i) { c.st_acc_[i] += 1; }
#else
constexpr int kStamps = 32;
#define DSTAMP(i)
#define DCOUNT(i)
#endif
#define STAMP_DECL c.st_prev_ = stamp_now(); for (int i_ = 0; i_ < kStamps; ++i_) c.st_acc_[i_] = 0;
// (the counter is read by an asm volatile with a memory clobber: the compiler may not move it across LDS operations, barriers or the loop's other asm
//  statements -- clock64() was seen hoisted above the code it was meant to time)
__device__ __forceinline__ long long stamp_now()
{
    long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#define STAMP(i) { long long now_ = stamp_now(); c.st_acc_[i] += now_ - c.st_prev_; c.st_prev_ = now_; }
#else
#define STAMP_DECL
#define STAMP(i)
#define DSTAMP(i)
#define DCOUNT(i)
#endif

// ------------------------------------------------------------------------------------------------
// per-workgroup context: LDS pointers + sizes (all uniform across the 256 threads)
// ------------------------------------------------------------------------------------------------
struct Ctx {
    const DevStruct* S;
    int tid, lane, wave;
    int nv, na, nc, k, n, nu, neq, nin2, ldj, ldm, ldc, ldb;
    int nblk; // J0 = U^-1 is block diagonal [dv | 12 | 12 ..] with the first block nblk wide (blk_begin / blk_end); n where H is dense
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
typedef int int4v __attribute__((ext_vector_type(4)));
// 1/x to full precision without the IEEE division's scaling and fix-up: v_rcp_f64 and two Newton steps
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
// u / r of the step-length ratio tests (r > 0).  The reciprocal form, except where 1 / r leaves the double range: v_rcp_f64 of an r
// below 2^-1022 overflows and the Newton steps turn the infinity into NaN, which the argmin would silently drop -- eiquadprog's u / r
// is 0 there for u = 0 and huge otherwise.  That (never observed) case takes the IEEE division, under a branch no lane enters.
__device__ __forceinline__ double ratio_pos(double u, double r) { return (r >= 0x1p-1020) ? u * fast_rcp(r) : u / r; }
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// the same for a value that is the same in every lane, kept in a scalar register: a loop bound made opaque() would turn the loop's branch
// into a per-lane one (EXEC-masked loop around barriers)
__device__ __forceinline__ int opaque_uniform(int v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ double2v ld2(const double* p) { return *reinterpret_cast<const double2v*>(__builtin_assume_aligned(p, 16)); }
__device__ __forceinline__ int wave_min_int(int v) { return -wave_max_int(-v); }

// LDS atomics as cross-lane reductions where only a few lanes take part (tools/ubench/lds_atomics.hip: about 50 cycles + 10 per participating lane,
// against 500-700 for a wave-wide argmin by DPP).  No return value; the issuing wave sees the result with its next LDS read (one wave's LDS
// operations execute in order), other waves behind a barrier.  (The low half of a generic pointer into LDS is the LDS address.)
__device__ __forceinline__ void lds_min_f64(double* p, double v)
{
    asm volatile("ds_min_f64 %0, %1" : : "v"((unsigned)(unsigned long long)p), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_min_u32(unsigned* p, unsigned v)
{
    asm volatile("ds_min_u32 %0, %1" : : "v"((unsigned)(unsigned long long)p), "v"(v) : "memory");
}

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

// inclusive prefix sum over the 64 lanes: row_shr 1/2/4/8 inside each row of 16, then the two gfx9 row broadcasts
// (lane 15 of rows 0 / 2 into rows 1 / 3, lane 31 into rows 2 and 3).  Lanes without a source add the `old` operand, 0.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_add0(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_scan_incl(double v)
{
    v = dpp_add0<0x111, 0xf>(v); // row_shr:1
    v = dpp_add0<0x112, 0xf>(v); // row_shr:2
    v = dpp_add0<0x114, 0xf>(v); // row_shr:4
    v = dpp_add0<0x118, 0xf>(v); // row_shr:8
    v = dpp_add0<0x142, 0xa>(v); // row_bcast:15 -> rows 1, 3
    v = dpp_add0<0x143, 0xc>(v); // row_bcast:31 -> rows 2, 3
    return v;
}

// sum over the 4 lanes of a quad (every lane gets the total)
__device__ __forceinline__ double quad_sum(double v)
{
    v += dpp_get<0xB1>(v); // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v); // quad_perm [2,3,0,1]
    return v;
}

#endif // __HIPCC__
} // namespace wbcqp
