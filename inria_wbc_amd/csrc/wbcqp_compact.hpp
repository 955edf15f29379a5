// wbcqp_compact.hpp -- the same QP on one workgroup with HALF the LDS: two QPs resident per CU.
//
// solve_one (wbcqp_device.hpp) keeps everything a QP ever touches in LDS (Talos: 160 KiB = one QP per CU, one wave per SIMD,
// 44 % of every wave's cycles waiting with nothing else to issue -- profiles/r01/v15_pmc_sq.csv).  This variant holds only
// what is touched MORE THAN ONCE PER PHASE:
//   * J (n x ldj) and the packed R: the two arrays the active-set loop lives on;
//   * their regions double as staging before they exist: the dense task rows sit in the J region until H is assembled,
//     the elimination's panels behind them; N = CE' and then B = J0'N sit in the tail of the R region (R has only its neq
//     first columns until the inequality loop starts);
//   * M, Jc = T'A_c, A_c and the force generators never enter LDS: the six base-dynamics rows of M and J_u = Jc(:, :6) go
//     from the record's registers straight into N; the actuation rows [M_a | -J_a'] (44 x 74 for Talos, constant for the
//     whole loop) are re-read from the L2-resident record once and then live in REGISTERS, four lanes per row, where
//     act_rows() used to read them from LDS in every iteration;
//   * vectors in 80-entry slots (64 where the stack fits them: cp::vec_map) with the dead ones aliased; iai / iaexcl as bytes.
// Talos: 81.2 KB, iCub 60 KB.  Eligibility (host, derive_compact): n <= 80, neq <= 22, nv <= 52, nc <= 2, nu <= 8,
// nin2 <= 256, r1 <= 128, n_tasks <= 64 -- every stack the reference ships.  Anything else runs solve_one.
//
// Same algorithm, same phases, same reference contract as solve_one (controller.cpp:244-251); the row of an actuation or
// friction constraint is published by the lanes that own it (one more barrier in those iterations).
#pragma once

#include "wbcqp_prims.hpp"
#include "wbcqp_factor.hpp"
#include "wbcqp_equality.hpp"
#include "wbcqp_activeset.hpp"

#if defined(WBCQP_X_STOP) && !defined(WBCQP_STAMPS)
// instruction accounting (tools/phase_insts.sh): the QP ends at stamp WBCQP_X_STOP, so the SQ counters of two such builds differ by one phase
#undef STAMP
#define STAMP(i) { if ((i) == WBCQP_X_STOP) return; }
#endif

namespace wbcqp {
#ifdef __HIPCC__

namespace cp {
// vector region (doubles from one base; the offsets are literals in a shipped stack's instantiation, a few scalar operations in the generic kernel).
// Slots of VS entries: 80, or 64 for n <= 62; U has n + 2 entries (VS + 8).  act_dot reads x, x_old and z up to entry nv + 23 unconditionally (the
// coefficients past the stack's own are zero): with 64-entry slots and nv > 40 that runs up to ten entries into the NEXT slot -- NP behind X, the
// R slot (weights, then r from entry neq on) behind XOLD, XOLD behind Z -- which therefore hold finite numbers from the first phase on (NP and the
// weights' tail are zeroed with the padding of x and z).  The two slots only a stack WITH actuation bounds uses come last and are left out without them.
struct VecMap {
    int X, NP, D, Z, XOLD, R, U, UOLD, RDINV, PART, TACT /* 64 */, S /* 256 */, BLB, BUB, RED /* 32 */, CE0 /* 24 */, TL, TU, COUNT, VS;
    // dead-time aliases: g and 1/sqrt(pivot) die with x0, the weights with the force blocks, b1 with the assembly.  The S slot (b1, the force
    // blocks' panels, the equality QR's reflector) is free in the inequality loop: the pending update's w, d of the active positions and the
    // block-reduction scratch of the loop's rare paths live there; the loop's own slots (lp::) take RED and CE0 (adjacent).
    int G, DINV, W, B1, PRM, PW, DI /* 128 */, LRED /* 32 */, EL /* 8 */;
};
// where the friction table lives in the loop (wbcqp_compact.hpp, phase 4).  1: in J's dead equality columns 2..13 of rows 0 .. 17 nc - 1 (fourteen
// equalities or more); 2: in columns 2..7 of TWO rows per entry, rows 0 .. 34 nc - 1 (eight to thirteen equalities); 0: behind the rotation table in the R region
__host__ __device__ constexpr int fric_in_j(int n, int neq, int nc) { return (neq >= 14 && n >= 17 * nc) ? 1 : ((neq >= 8 && n >= 34 * nc) ? 2 : 0); }
__host__ __device__ constexpr int vec_stride(int n, int /* nv */) { return n <= 62 ? 64 : 80; }
__host__ __device__ constexpr VecMap vec_map(int n, int nv, int act_bounds)
{
    VecMap m{};
    const int VS = vec_stride(n, nv);
    m.VS = VS;
    m.X = 0; m.NP = VS; m.D = 2 * VS; m.Z = 3 * VS; m.XOLD = 4 * VS; m.R = 5 * VS; m.U = 6 * VS; m.UOLD = 7 * VS + 8; m.RDINV = 8 * VS + 8;
    m.PART = 9 * VS + 8; m.TACT = 10 * VS + 8; m.S = m.TACT + 64; m.BLB = m.S + 256; m.BUB = m.BLB + 64; m.RED = m.BUB + 64; m.CE0 = m.RED + 32;
    m.TL = m.CE0 + 24; m.TU = m.TL + 64;
    m.COUNT = act_bounds ? m.TU + 64 : m.TL;
    m.G = m.XOLD; m.DINV = m.UOLD; m.W = m.R; m.B1 = m.S; m.PRM = m.D; m.PW = m.S; m.DI = m.S + 80; m.LRED = m.S + 208; m.EL = m.S + 240;
    return m;
}
// int region (ints)
constexpr int IA = 0 /* n + 2 <= 84 */, IAOLD = 84, IACT = 164 /* 256 bytes */, IEXCL = 228, ICOUNT = 292;
constexpr int NVQ = 13; // ceil(52 / 4): M_a coefficients per lane of a row's quad
constexpr int KQ = 6;   // 24 / 4:  J_a' coefficients per lane
} // namespace cp

// the actuation rows [M_a | -J_a'] in registers: lane q4 of row rr's quad keeps columns q4 + 4u
struct ActRegs {
    double am[cp::NVQ];
    double aj[cp::KQ];
};

// One actuation row of tau' = M_a xn - J_a' fn with xn = xp + t zp formed on the fly (t = 0: xn = xp exactly): four lanes per
// row (act_rows()' layout), every lane of the quad returns the row's total.  Every read is base + immediate from ONE
// per-lane address: coefficients past nv / k are zero and what they multiply is finite -- the x and z slots are zero from
// n to their end (set once per QP); past the slot's end they meet finite numbers (cp::VecMap).
__device__ __forceinline__ double act_dot(const Ctx& c, const ActRegs& a, const double* xp, const double* zp, double t)
{
    const int nv = c.nv, q4 = c.tid & 3;
    const double* zq = zp + q4;
    const double* xq = xp + q4;
    const double* zfq = zq + nv;
    const double* xfq = xq + nv;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    double zv[cp::NVQ], xv[cp::NVQ];
#pragma unroll
    for (int u = 0; u < cp::NVQ; ++u) {
        zv[u] = zq[4 * u];
        xv[u] = xq[4 * u];
    }
    double zf[cp::KQ], xf[cp::KQ];
#pragma unroll
    for (int u = 0; u < cp::KQ; ++u) {
        zf[u] = zfq[4 * u];
        xf[u] = xfq[4 * u];
    }
#pragma unroll
    for (int u = 0; u < cp::NVQ; ++u) {
        const double xn = fma(t, zv[u], xv[u]);
        if ((u & 3) == 0) a0 = fma(a.am[u], xn, a0);
        else if ((u & 3) == 1) a1 = fma(a.am[u], xn, a1);
        else if ((u & 3) == 2) a2 = fma(a.am[u], xn, a2);
        else a3 = fma(a.am[u], xn, a3);
    }
#pragma unroll
    for (int u = 0; u < cp::KQ; ++u) {
        const double xn = fma(t, zf[u], xf[u]);
        if ((u & 3) == 0) a0 = fma(-a.aj[u], xn, a0);
        else if ((u & 3) == 1) a1 = fma(-a.aj[u], xn, a1);
        else if ((u & 3) == 2) a2 = fma(-a.aj[u], xn, a2);
        else a3 = fma(-a.aj[u], xn, a3);
    }
    double acc = (a0 + a1) + (a2 + a3);
    acc += dpp_get<0xB1>(acc);
    acc += dpp_get<0x4E>(acc);
    return acc;
}

// One actuation row of A_act z (the increment of tau' along a step): act_dot with the z slots only
__device__ __forceinline__ double act_dot1(const Ctx& c, const ActRegs& a, const double* zp)
{
    const int nv = c.nv, q4 = c.tid & 3;
    const double* zq = zp + q4;
    const double* zfq = zq + nv;
    double zv[cp::NVQ], zf[cp::KQ];
#pragma unroll
    for (int u = 0; u < cp::NVQ; ++u) zv[u] = zq[4 * u];
#pragma unroll
    for (int u = 0; u < cp::KQ; ++u) zf[u] = zfq[4 * u];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int u = 0; u < cp::NVQ; ++u) {
        if ((u & 3) == 0) a0 = fma(a.am[u], zv[u], a0);
        else if ((u & 3) == 1) a1 = fma(a.am[u], zv[u], a1);
        else if ((u & 3) == 2) a2 = fma(a.am[u], zv[u], a2);
        else a3 = fma(a.am[u], zv[u], a3);
    }
#pragma unroll
    for (int u = 0; u < cp::KQ; ++u) {
        if ((u & 3) == 0) a0 = fma(-a.aj[u], zf[u], a0);
        else if ((u & 3) == 1) a1 = fma(-a.aj[u], zf[u], a1);
        else if ((u & 3) == 2) a2 = fma(-a.aj[u], zf[u], a2);
        else a3 = fma(-a.aj[u], zf[u], a3);
    }
    double acc = (a0 + a1) + (a2 + a3);
    acc += dpp_get<0xB1>(acc);
    acc += dpp_get<0x4E>(acc);
    return acc;
}

// ---- the inequality block of R is never stored: its INVERSE Ri is (packed columns like R: (i, j), i <= j, at roff(j) + i,
// positions counted from the first inequality, mi = iq - neq of them).  r = R^-1 d restricted to the inequality rows is
// Ri d_I: a triangular matvec with no dependent chain (the back substitution took one dependent step per active
// constraint on one wave); a new column [d; alpha] of R is the column [-r / alpha; 1 / alpha] of the inverse, and r is at
// hand when a constraint is added; dropping a constraint rotates Ri's columns with the SAME rotations as J's (below).
// One lane per row (mi <= 64, host check), eight columns' operands in flight.  d_I(j) = dsg * dsrc[neq + j].  r lands in
// c.r[neq + i]; the lane's own r is returned (0 past mi).
__device__ __forceinline__ double ri_matvec(Ctx& c, const double* Ri, int mi, const double* dsrc, double dsg)
{
    const int lane = c.lane;
    const double* dI = dsrc + c.neq;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int j = 0; j < mi; j += 8) {
        double rv[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = min(j + u, mi - 1);
            rv[u] = Ri[roff(jj) + min(lane, jj)];
            dv[u] = dI[jj];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double dm = (lane <= j + u && j + u < mi) ? dv[u] : 0.0;
            if ((u & 3) == 0) a0 = fma(rv[u], dm, a0);
            else if ((u & 3) == 1) a1 = fma(rv[u], dm, a1);
            else if ((u & 3) == 2) a2 = fma(rv[u], dm, a2);
            else a3 = fma(rv[u], dm, a3);
        }
    }
    const double r = dsg * ((a0 + a1) + (a2 + a3));
    if (lane < mi) c.r[c.neq + lane] = r;
    return (lane < mi) ? r : 0.0;
}

// Dropping position p of the active inequalities (eiquadprog delete_constraint): R without column p is re-triangularised
// by rotations G of the row pairs (j, j + 1), j = p .. mi - 2; J takes J G' and the inverse takes Ri G' with row p and the
// last column dropped.  Row p of Ri G' must vanish left of its last entry, so rotation l is fixed by the running norm of
// rho(l) = Ri(p, p + l): with S(l) = sum_{i <= l} rho(i)^2 the pair in front of rotation l is (a, b) = (-sqrt S(l), rho(l + 1))
// ((rho(0), rho(1)) for l = 0) and (cc, ss) = (b, -a) / sqrt S(l + 1) in eiquadprog's symmetric form
//   new(j) = cc t1 + ss t2,  new(j + 1) = ss t1 - cc t2.
// ONE prefix sum gives every coefficient at once -- eiquadprog's chain (and round 2's) paid a division, a square root and an
// LDS round trip per step, sequentially.  S(0) = Ri(p, p)^2 > 0, so no step is ever skipped.  Call on one wave;
// prm[2 l], prm[2 l + 1] = cc, ss of rotation l, l < mi - p - 1.
__device__ __forceinline__ void drop_coefficients(Ctx& c, const double* Ri, int mi, int p, double* prm)
{
    const int l = c.lane, L1 = mi - p;
    const double rho0 = (l < L1) ? Ri[roff(p + min(l, L1 - 1)) + p] : 0.0;
    const double rho1 = (l + 1 < L1) ? Ri[roff(p + min(l + 1, L1 - 1)) + p] : 0.0;
    const double S = wave_scan_incl(rho0 * rho0);
    const double S1 = fma(rho1, rho1, S);
    if (l + 1 < L1) {
        const double rs1 = rsqrt(S1);
        const double na = (l == 0) ? -rho0 : S * rsqrt(S); // -a
        double2v o;
        o.x = rho1 * rs1;
        o.y = na * rs1;
        *reinterpret_cast<double2v*>(__builtin_assume_aligned(prm + 2 * l, 16)) = o;
    }
}

// One row (J's or Ri's) through the rotations of a drop, together with the entry of d that travels with it: t1 enters as the
// row's element in the first rotated column, dch as d's; on return they are the elements that LEAVE the active block.
// get(u, jj) loads the row's element of column index jj (the caller masks what does not exist), put(jj, v) stores the new
// element of column jj; e(jj) is d's element.  Eight steps' operands are in flight before the dependent chain starts.
template <typename Get, typename Put, typename PutD>
__device__ __forceinline__ void rotate_row(const double* prm, const double* dcur, int c0, int L, double& t1, double& dch, Get get, Put put, PutD putd)
{
    for (int l0 = 0; l0 < L; l0 += 8) {
        double t2[8], e2[8];
        double2v cs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { // (reads past L stay inside LDS and are not used)
            t2[u] = get(l0 + u + 1);
            e2[u] = dcur[c0 + l0 + u + 1];
            cs[u] = ld2(prm + 2 * (l0 + u));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (l0 + u < L) {
                put(l0 + u, fma(cs[u].y, t2[u], cs[u].x * t1));
                t1 = fma(cs[u].y, t1, -(cs[u].x * t2[u]));
                putd(l0 + u, fma(cs[u].y, e2[u], cs[u].x * dch));
                dch = fma(cs[u].y, dch, -(cs[u].x * e2[u]));
            }
    }
}

// ---- the compact kernel's inequality loop (solve_one_compact, phase 4): its slots and the helpers of its row-packed inverse
namespace lp {
// slots in the RED + CE0 area (doubles from VecMap::RED; `I` entries are int indices into the same area).  Every field has ONE writing phase, and
// at least one barrier lies between its readers and the next write -- no double buffering.  The phase-B / drop fields sit where the set-up's
// block reductions had their slots (first written two barriers into the loop); the pick's fields, which are written before the loop's first
// barrier, in CE0 (y of the equality phase: read for the last time before the set-up's final barrier).
constexpr int BZF = 0 /* 3 ints (doubles 0, 1): a wave's rows hold a z_k^2 > eps */, BDN2 = 2, BALPHA = 3, BV0 = 4, BTAU = 5, BT2C = 6, NPW = 7, DDELTA = 8, DCL = 9,
              DRHO0 = 10;
// election slots (doubles from VecMap::EL; `u` entries index the same area as unsigned): two generations of the pick's (minimum, word), the same for
// the warm start's hinted rows, one of the step length's (minimum, position)
constexpr int EMIN = 0 /* 2 */, EKEY = 2 * 2 /* 2 u */, EWMIN = 3 /* 2 */, EWKEY = 2 * 5 /* 2 u */, ET1 = 6, ET1POS = 2 * 7 /* u */;
constexpr double kNone = 1e300; // "nobody stood": any candidate (s < 0) is below it
// row i of the row-packed inverse: elements (i, j), i <= j <= MM (one spare, always zero), at rio(i, MM) + j - i
__device__ __forceinline__ int rio(int i, int MM) { return i * (MM + 1) - ((i * (i - 1)) >> 1); }
// One row (J's or Ri's) through the L rotations of a drop in their closed form: with P_l = sum_{i <= l} rho_i x_i the element of the new
// column l is a_l P_l + b_l x_{l+1} (prm[4 l], [4 l + 1]; rho_{l+1} at [4 l + 2]).  P enters as rho_0 x_0 and returns as P_L, whose multiple
// -P_L / sqrt(S_L) is the element that leaves the active block.  get(jj) loads x_jj (the caller masks what does not exist), put(jj, v) stores
// the new element of column jj.  Eight steps' operands are in flight; the dependent chain is one FMA per step.
template <typename Get, typename Put>
__device__ __forceinline__ void rotate_row_ps(const double* prm, int L, double& P, Get get, Put put)
{
    for (int l0 = 0; l0 < L; l0 += 8) {
        double x1[8], rh[8];
        double2v ab[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { // (reads past L stay inside LDS and are not used)
            x1[u] = get(l0 + u + 1);
            ab[u] = ld2(prm + 4 * (l0 + u));
            rh[u] = prm[4 * (l0 + u) + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (l0 + u < L) {
                put(l0 + u, fma(ab[u].x, P, ab[u].y * x1[u]));
                P = fma(rh[u], x1[u], P);
            }
    }
}
} // namespace lp

// what a thread keeps about the one row of s it owns (nin2 <= 256: row tid)
struct OwnRow {
    int meta;        // -1: none
    double ci0;
    const double* coef; // friction rows only: the 12 coefficients of the row (of its positive twin) in the LDS table, else null
};

template <int SPEC> __device__ __forceinline__ Dims dims_of(const DevStruct& S)
{
    if constexpr (SPEC > 0) {
        Dims d = kSpecDims[SPEC - 1]; // literals (wbcqp_types.hpp); the host has checked that they are this structure's
        d.max_iter = S.max_iter;      // ... all but the iteration bound: a caller that bounds its tick time keeps its stack's instantiation
        return d;
    }
    else return dims_from(S);
}

// One contact's force block on one wave (8 x 8 lane grid, 2 x 2 positions per lane): H_ff = wt F'F + reg I eliminated to (1 / sqrt(pivot), Y) with
// J_ff = Y diag(1 / sqrt(pivot)).  hF enters as the lane's tile of F'F; tF returns the lane's (at most two) diagonal elements of H_ff, in the order the
// kernel adds them to tr(H).  The solve kernels and ffcache_kernel (which makes DevStruct::ffc) both run THIS function: a cached factor is the computed
// one bit for bit.
__device__ __forceinline__ void force_block_factor(Ctx& c, double (&hF)[2][2], double (&yF)[2][2], double (&tF)[2], double wt, double reg, int la, int le,
                                                   double* RBf, double* YBf, double* dinv_out, int lane)
{
    tF[0] = 0.0;
    tF[1] = 0.0;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int r = la + 8 * u, q = le + 8 * w;
            if (r < 12 && q < 12) {
                hF[u][w] = wt * hF[u][w] + ((r == q) ? reg : 0.0);
                if (r == q) tF[u] = hF[u][w]; // (r == q needs u == w: la, le < 8)
            }
            else hF[u][w] = (r == q) ? 1.0 : 0.0;
            yF[u][w] = 0.0;
        }
    publish_panel<3, 2, true, 0>(c, hF, yF, la, le, 0, RBf, YBf);
    eliminate_block<3, 2, true, 0>(c, hF, yF, la, le, 12, RBf, YBf, dinv_out, lane < 4, lane & 3);
}

// makes DevStruct::ffc from the force-regularisation weights of ONE QP (wbcqp_api.hip launches it once per slot, ahead of the slot's first solve, on that
// launch's stream): one wave per contact, the solve kernels' own code
template <typename TI>
__global__ __launch_bounds__(128) void ffcache_kernel(const DevStruct S, const TI* w, double* out)
{
    __shared__ __align__(16) double scr[2 * 128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= S.nc) return;
    const int la = lane >> 3, le = lane & 7;
    double hF[2][2], yF[2][2], tF[2];
    const double* ftf = S.ftf + wave * 144;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ww = 0; ww < 2; ++ww) hF[u][ww] = ftf[min(la + 8 * u, 11) * 12 + min(le + 8 * ww, 11)];
    const double wt = (double)w[S.forcereg_task[wave]];
    Ctx c;
    double* o = out + wave * kFfcStride;
    force_block_factor(c, hF, yF, tF, wt, S.hessian_reg, la, le, scr + wave * 128, scr + wave * 128 + 64, o + 2, lane);
    double* ol = o + 16 + 6 * lane;
    ol[0] = yF[0][0]; ol[1] = yF[0][1]; ol[2] = yF[1][0]; ol[3] = yF[1][1]; ol[4] = tF[0]; ol[5] = tF[1];
    __builtin_amdgcn_s_waitcnt(0); // (every store of the entry before the weight that validates it; readers are later launches in stream order anyway)
    if (lane == 0) o[0] = wt;
}

// SPEC > 0: the instantiation for the shipped stack kSpecDims[SPEC - 1] -- every size and LDS offset below is a literal
template <typename TI, int SPEC = 0, bool WARM = false>
__device__ __forceinline__ void solve_one_compact(const GroupArgs<TI>& ga, const DevStruct& S, const int b, double* lds, const int tid, const int brec = -1)
{
    const Dims D = dims_of<SPEC>(S);
#ifdef WBCQP_POISON_LDS
    // debugging aid: every LDS word starts as a signalling pattern, so a read of a word nobody wrote shows in the results
    for (int i = tid; i < S.lds_doubles; i += kThreads) lds[i] = __longlong_as_double(WBCQP_POISON_LDS);
    bsync();
#endif
    const cp::VecMap vm = cp::vec_map(D.n, D.nv, D.act_bounds);
    const int VS = vm.VS;
    Ctx c;
    c.S = &S;
    c.tid = tid;
    c.lane = tid & (kWave - 1);
    c.wave = uni(tid >> 6);
    c.rslot = 0;
    c.nblk = D.nv;
    c.nv = D.nv; c.na = D.na; c.nc = D.nc; c.k = D.k; c.n = D.n; c.nu = D.nu;
    c.neq = D.neq; c.nin2 = D.nin2; c.ldj = D.ldj; c.ldm = 0; c.ldc = 0; c.ldb = D.ldb;
    c.J = lds + D.o_J; c.R = lds + D.o_R;
    c.M = nullptr; c.Jc = nullptr; c.Ac = nullptr; c.h = nullptr; c.q = nullptr; c.wrow = nullptr; c.stash = nullptr;
    c.eqw = nullptr; c.eqt = nullptr; c.bc = nullptr; c.iai = nullptr; c.iaexcl = nullptr;
    {
        double* vec = lds + D.o_vec;
        c.x = vec + vm.X; c.np = vec + vm.NP; c.d = vec + vm.D; c.z = vec + vm.Z; c.xold = vec + vm.XOLD;
        c.r = vec + vm.R; c.u = vec + vm.U; c.uold = vec + vm.UOLD; c.rdinv = vec + vm.RDINV; c.part = vec + vm.PART;
        c.s = vec + vm.S; c.blb = vec + vm.BLB; c.bub = vec + vm.BUB; c.tl = vec + vm.TL; c.tu = vec + vm.TU; // (tl, tu: only with actuation bounds)
        c.red = vec + vm.RED; c.g = vec + vm.G; c.dinv = vec + vm.DINV; c.w = vec + vm.W; c.b1 = vec + vm.B1;
        c.prm = vec + vm.PRM;
    }
    double* const tact = lds + D.o_vec + vm.TACT;
    double* const ce0v = lds + D.o_vec + vm.CE0; // ce0 of the equalities, then rhs, then y
    int* ia = reinterpret_cast<int*>(lds + D.o_int);
    c.A = ia + cp::IA; c.Aold = ia + cp::IAOLD; c.gskip = nullptr; c.meta = nullptr;
    signed char* const act = reinterpret_cast<signed char*>(ia + cp::IACT);   // 1: row is in the active set (iai == -1)
    signed char* const excl = reinterpret_cast<signed char*>(ia + cp::IEXCL); // 1: row may be picked (iaexcl)
    const int n = c.n, nv = c.nv, na = c.na, nc = c.nc, k = c.k, nu = c.nu, neq = c.neq, nin2 = c.nin2;
    const int ldj = c.ldj, ldb = c.ldb;
    c.iq = 0;
    c.R_norm = 1.0;

    const int n_dense = D.n_dense, n_sel = D.n_sel, n_bound = D.n_bound, r1 = D.r1, n_tasks = D.n_tasks;
    const size_t qp = (size_t)b;                       // torque limits, weights and every output: the QP's index in the batch
    const size_t qr = (size_t)(brec >= 0 ? brec : b);  // the record (M .. bub): wbcqp_rollout's workgroups keep one record slot each
    double* const As = c.J;                  // dense task rows are staged in the J region (J appears after the elimination)
    double* const RB = c.J + D.o_pan;        // the elimination's panels, behind the staged rows
    double* const YB = RB + 512;
    double* const Nm = c.R;                  // N = CE' (n x ldb), then B = J0'N: the R region (the packed R replaces it when the QR has B in registers)
    const int lenM = nv * (nv + 1) / 2, lenA = n_dense * nv, lenAc = nc * 6 * nv;
    const TI* const pM = ga.M + qr * lenM;
    const TI* const pAc = ga.Ac + qr * (size_t)lenAc;

    STAMP_DECL
    int meta0 = 0; // descriptor of one-sided row tid (the thread that owns the row keeps it: the loop needs no table of them in LDS)
    // ---------------- phase 0: the record's loads all in flight, then land where they are used ----------------
    {
        constexpr int RA = 9, RC = 3; // rounds of 256 covered by registers; longer arrays finish in tail loops
        const TI* pA = ga.A + qr * (size_t)lenA;
        TI vA[RA], vC[RC];
        unsigned vQ[RA], vCq[RC];
        if (lenA > 0) {
            ld_regs<TI, RA>(pA, lenA, tid, vA);
            ld_regs<unsigned, RA>(S.apack, lenA, tid, vQ);
        }
        if (nc > 0) {
            ld_regs<TI, RC>(pAc, lenAc, tid, vC);
            ld_regs<unsigned, RC>(S.acpack, lenAc, tid, vCq);
        }
        // base dynamics rows of N: N(kk, e) = M(kk, e), e < nu -- eight lanes per row, two rounds (nv <= 52)
        TI vMu[2] = {TI(0), TI(0)};
        if (nu > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = tid + u * kThreads, kk = t >> 3, e = t & 7;
                const int hi = max(kk, e), lo = min(kk, e);
                vMu[u] = pM[min(hi * (hi + 1) / 2 + lo, lenM - 1)];
            }
        }
        // N(nv + m, e) = -Jc(m, e) = -sum_r T(r, m) A_c(r, e): eight lanes per force column
        TI acv[6];
        double tcv[6];
        if (nc > 0 && nu > 0) {
            const int m = min(tid >> 3, k - 1), e = min(tid & 7, nu - 1);
            const int ct = (m >= 12) ? 1 : 0, mm = m - 12 * ct;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                acv[r] = pAc[(ct * 6 + r) * nv + e];
                tcv[r] = S.force_gen[ct * 72 + r * 12 + mm];
            }
        }
        // the short vectors: one (clamped) element per thread each
        const TI vb1 = ga.b1[qr * r1 + min(tid, r1 - 1)];
        const TI vw = ga.w[qp * n_tasks + min(tid, n_tasks - 1)];
        TI vce = TI(0), vbl = TI(0), vbu = TI(0), vtl = TI(0), vtu = TI(0), vha = TI(0);
        if (neq > 0) {
            if (tid < nu) vce = ga.h[qr * nv + tid];
            else if (nc > 0) vce = ga.bc[qr * (nc * 6) + min(tid - nu, nc * 6 - 1)];
        }
        if (n_bound > 0) {
            vbl = ga.blb[qr * n_bound + min(tid, n_bound - 1)];
            vbu = ga.bub[qr * n_bound + min(tid, n_bound - 1)];
        }
        if (D.act_bounds) {
            vtl = ga.tlb[qp * na + min(tid, na - 1)];
            vtu = ga.tub[qp * na + min(tid, na - 1)];
            vha = ga.h[qr * nv + nu + min(tid, na - 1)];
        }
        meta0 = (nin2 > 0) ? S.rowmeta[min(tid, nin2 - 1)] : 0;
        const int drt = (n_dense > 0) ? S.dense_row_task[min(tid, n_dense - 1)] : 0;
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
        // ---- land
        if (lenA > 0) {
#pragma unroll
            for (int u = 0; u < RA; ++u) {
                const int e = tid + u * kThreads;
                if (e < lenA) As[vQ[u]] = (double)vA[u];
            }
        }
        if (nc > 0) {
#pragma unroll
            for (int u = 0; u < RC; ++u) {
                const int e = tid + u * kThreads;
                if (e < lenAc) Nm[vCq[u]] = (double)vC[u];
            }
            // rows of the force variables: -J_u' in the base-dynamics columns, zero in the contact-motion columns
            if (nu > 0 && tid < 8 * k && (tid & 7) < nu) {
                double sacc = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) sacc = fma(tcv[r], (double)acv[r], sacc);
                Nm[(nv + (tid >> 3)) * ldb + (tid & 7)] = -sacc;
            }
            for (int t = tid; t < 16 * k; t += kThreads) {
                const int e2 = nu + (t & 15);
                if (e2 < neq) Nm[(nv + (t >> 4)) * ldb + e2] = 0.0;
            }
        }
        if (nu > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = tid + u * kThreads, kk = t >> 3, e = t & 7;
                if (kk < nv && e < nu) Nm[kk * ldb + e] = (double)vMu[u];
            }
        }
        if (tid < neq) ce0v[tid] = (tid < nu) ? (double)vce : -(double)vce; // ce0 = h_u | -bc
        if (tid < r1) c.b1[tid] = (double)vb1;
        if (tid < n_tasks) c.w[tid] = (double)vw;
        if (tid < n_bound) {
            c.blb[tid] = (double)vbl;
            c.bub[tid] = (double)vbu;
        }
        if (D.act_bounds && tid < na) { // lb - h_a, ub - h_a (computeProblemData, actuation tasks)
            c.tl[tid] = (double)vtl - (double)vha;
            c.tu[tid] = (double)vtu - (double)vha;
        }
        // tails of arrays longer than the register rounds (none for the reference's stacks)
        for (int e = tid + RA * kThreads; e < lenA; e += kThreads) As[S.apack[e]] = (double)pA[e];
        for (int e = tid + RC * kThreads; e < lenAc; e += kThreads) Nm[S.acpack[e]] = (double)pAc[e];
        if (tid < nv) { // diagonal additions / right-hand sides of the selection rows
            c.z[tid] = 0.0;
            c.d[tid] = 0.0;
        }
        if (tid >= n && tid < VS) { // finite padding for act_dot's unconditional reads; never written again
            c.z[tid] = 0.0;
            c.x[tid] = 0.0;
        }
        if (tid < VS) c.np[tid] = 0.0;                    // ... which may run into the slot behind x (cp::VecMap)
        if (tid >= n_tasks && tid < VS) c.w[tid] = 0.0;   // ... and into the one behind x_old: the weights' tail (r replaces the slot from entry neq on)
        bsync();
        if (tid < n_dense) { // (row weight, right-hand side) pairs behind the staged rows: one 16-byte read per row
            As[n_dense * 64 + 2 * tid] = c.w[drt];
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
    }
    bsync();
    STAMP(0)

    // ---------------- phases 1-2b in registers: H assembly, Cholesky H = U'U, J = U^-1 (as solve_one) ----------------
    // A thread of the 16 x 16 grid owns positions (ta + 16 u, te + 16 w), u <= w < NU, of the dv block: NU = 4 covers nv <= 64; a shipped stack
    // with nv <= 48 (iCub: 38) is instantiated with NU = 3 -- nine tile elements instead of sixteen in the assembly, the elimination and J = U^-1,
    // the same operations on every element that exists (the positions left out are the identity padding past nv): same bits, 28 registers fewer.
    constexpr int NU = (SPEC > 0 && kSpecDims[SPEC > 0 ? SPEC - 1 : 0].nv <= 48) ? 3 : 4;
    double c1, c2;
    {
        const int ta = tid >> 4, te = tid & 15;
        double h[NU][NU];
        double trace = 0.0;
        {
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int w = 0; w < NU; ++w) h[u][w] = 0.0;
            double gacc[4] = {0.0, 0.0, 0.0, 0.0};
            const double* Ai = As + ta * 2; // (layout of a staged row: wbcqp_types.hpp, apack)
            const double* Aj = As + te * 2;
            const double* WB = As + n_dense * 64;
            auto ldrow = [&](int r, double2v (&ai)[2], double2v (&aj)[2], double2v& wb) __attribute__((always_inline)) {
                ai[0] = ld2(Ai + r * 64);
                ai[1] = ld2(Ai + r * 64 + 32);
                aj[0] = ld2(Aj + r * 64);
                aj[1] = ld2(Aj + r * 64 + 32);
                wb = ld2(WB + 2 * r);
            };
            auto macrow = [&](const double2v (&ai)[2], const double2v (&aj)[2], const double2v& wb) __attribute__((always_inline)) {
                const double a[4] = {ai[0].x, ai[0].y, ai[1].x, ai[1].y};
                const double ajw[4] = {aj[0].x * wb.x, aj[0].y * wb.x, aj[1].x * wb.x, aj[1].y * wb.x};
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int w = u; w < NU; ++w) h[u][w] = fma(a[u], ajw[w], h[u][w]);
#pragma unroll
                for (int w = 0; w < NU; ++w) gacc[w] = fma(ajw[w], wb.y, gacc[w]);
            };
            if (n_dense > 0) {
                const int nd = opaque_uniform(n_dense); // (a literal in the specialised builds: as a loop bound it would unroll forty row bodies)
                double2v ai0[2], aj0[2], ai1[2], aj1[2], wb0, wb1;
                ldrow(0, ai0, aj0, wb0);
                int r = 0;
                for (; r + 2 <= nd; r += 2) {
                    ldrow(r + 1, ai1, aj1, wb1);
                    macrow(ai0, aj0, wb0);
                    ldrow(min(r + 2, nd - 1), ai0, aj0, wb0);
                    macrow(ai1, aj1, wb1);
                }
                if (r < nd) macrow(ai0, aj0, wb0);
            }
            if (ta == 0) {
#pragma unroll
                for (int w = 0; w < NU; ++w) {
                    const int col = te + 16 * w;
                    if (col < nv) c.g[col] = -gacc[w] - c.d[col];
                }
            }
            if (ta == te) {
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int i = ta + 16 * u;
                    if (i < nv) {
                        h[u][u] += c.z[i] + S.hessian_reg;
                        trace += h[u][u];
                    }
                }
            }
        }
        STAMP(1)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int w = u; w < NU; ++w) {
                const int r = ta + 16 * u, q = te + 16 * w;
                if (r >= nv || q >= nv) h[u][w] = (r == q) ? 1.0 : 0.0;
            }
        double y[NU][NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int w = 0; w < NU; ++w) y[u][w] = 0.0;
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
        // the panels lie behind the staged rows: a fast thread may publish while a slow one still reads its last task row
        publish_panel<4, NU, false, 0>(c, h, y, ta, te, 0, RB, YB);
        eliminate_block<4, NU, false, 0>(c, h, y, ta, te, opaque_uniform((nv + 3) & ~3) /* not a constant for the unroller (a specialised build would lay out thirteen panel bodies: 256 VGPRs + 256 AGPRs + scratch) */, RB, YB, c.dinv, tid >= 128 && tid < 132, tid & 3);
        STAMP(2)
        bsync(); // staged rows and panels are dead: the region becomes J
        // ---- force blocks: wave-local (8 x 8 lane grid, 2 x 2 positions per lane), one contact per wave (nc <= 2), panels in the (idle) s slot.
        //      The factor depends on the record through ONE number, the contact's force-regularisation weight: a QP that carries the weight DevStruct::ffc
        //      was made for takes the factor from there (its loads fly while J is zeroed) instead of eliminating the block again
        const int fb = nv + 12 * c.wave;
        double wtF = 0.0, cw = 0.0, cy[6], cd = 0.0;
        const bool probe = c.wave < nc && S.ffc != nullptr;
        if (c.wave < nc) wtF = c.w[S.forcereg_task[c.wave]];
        if (probe) {
            const double* fc = S.ffc + c.wave * kFfcStride;
            cw = fc[0];
            cd = fc[2 + min(c.lane, 11)];
#pragma unroll
            for (int u = 0; u < 6; ++u) cy[u] = fc[16 + 6 * c.lane + u];
        }
        for (int e = tid; e < n * ldj + 2; e += kThreads) c.J[e] = 0.0; // (+ 2: the last row's pad pair when the rows are exactly n long, derive_compact)
        if (c.wave < nc) {
            if (probe && wtF == cw) { // (wave-uniform: one weight per contact; NaN -- no entry yet -- equals nothing)
                yF[0][0] = cy[0]; yF[0][1] = cy[1]; yF[1][0] = cy[2]; yF[1][1] = cy[3];
                trace += cy[4];
                trace += cy[5];
                if (c.lane < 12) c.dinv[fb + c.lane] = cd;
            }
            else {
                double tF[2];
                force_block_factor(c, hF, yF, tF, wtF, S.hessian_reg, la, le, c.s + c.wave * 128, c.s + c.wave * 128 + 64, c.dinv + fb, c.lane);
                trace += tF[0];
                trace += tF[1];
            }
        }
        bsync(); // J is zero, every 1/sqrt(pivot) is published
        // tr(H) and sum 1/sqrt(pivot) (the stopping rule's c1, c2): the waves' partial sums ride on the barrier that ends the J writes below
        // (block_sum()'s arithmetic, without its two barriers)
        {
            double tr2 = 0.0;
            for (int i = tid; i < n; i += kThreads) tr2 += c.dinv[i];
            const double tw = wave_sum(trace), dw = wave_sum(tr2);
            double* slot = c.red + c.rslot * 16;
            if (c.lane == 0) {
                slot[c.wave] = tw;
                slot[4 + c.wave] = dw;
            }
        }
        // final: J(r,q) = Y(r,q) dinv[q], J(r,r) = dinv[r]
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int w = u; w < NU; ++w) {
                const int r = ta + 16 * u, q = te + 16 * w;
                if (q < nv && r < q) c.J[r * ldj + q] = y[u][w] * c.dinv[q];
                else if (r == q && r < nv) c.J[r * ldj + r] = c.dinv[r];
            }
        if (c.wave < nc) {
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
        {
            const double* slot = c.red + c.rslot * 16;
            c1 = (slot[0] + slot[1]) + (slot[2] + slot[3]);
            c2 = (slot[4] + slot[5]) + (slot[6] + slot[7]);
            c.rslot ^= 1;
        }
    }
    STAMP(3)

    // ---------------- x = -H^-1 g = -J (J' g); f = 0.5 g'x ----------------
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
        f_value = block_sum(c, part); // (its barrier also publishes x0 to the equality phase)
    }
    STAMP(4)

    const double eps = 2.220446049250313e-16;
    const double inf = __builtin_huge_val();
    int status = -2; // running
    int left = D.max_iter; // outer iterations left before status 3: counted down, so that the loop keeps one register and never the bound itself
    // the loop's flags, set here so that the equality phase's barriers make them visible (other threads read them in the first evaluation
    // of s): with equalities the loop then needs no barrier of its own before it starts
    if (nin2 > 0) {
        if (tid < nin2) {
            act[tid] = 0;
            excl[tid] = 1;
        }
        if (tid >= n && tid < VS) c.xold[tid] = 0.0; // the second x buffer gets the same finite padding as the first (g, its tenant so far, ends at n)
        if (tid == 0) { // the loop's election slots (lp::E*), both generations armed
            double* EL0 = lds + D.o_vec + vm.EL;
            unsigned* EL0u = reinterpret_cast<unsigned*>(EL0);
            EL0[lp::EMIN] = lp::kNone; EL0[lp::EMIN + 1] = lp::kNone; EL0[lp::EWMIN] = lp::kNone; EL0[lp::EWMIN + 1] = lp::kNone;
            EL0u[lp::EKEY] = 0xffffffffu; EL0u[lp::EKEY + 1] = 0xffffffffu; EL0u[lp::EWKEY] = 0xffffffffu; EL0u[lp::EWKEY + 1] = 0xffffffffu;
        }
    }

    // The actuation rows' global loads (rows nu .. nv - 1 of M, the contact Jacobians' columns: from the record, L2 or HBM by now) are ISSUED as soon as the
    // equality QR has released its registers and CONSUMED behind the phases that follow it (y, x, u): their latency -- most of the 3.1 k cycles the phase
    // "actuation rows -> registers" took -- runs under those phases' barriers.
    TI act_mv[cp::NVQ];
    TI act_av[2][6];
    double act_tv[cp::KQ][6]; // the force generators' coefficients of the lane's columns (a constant table of the structure: L2)
    bool act_issued = false;
    auto issue_act_loads_into = [&](TI (&act_mv)[cp::NVQ], TI (&act_av)[2][6], double (&act_tv)[cp::KQ][6]) __attribute__((always_inline)) {
        if (na > 0) {
            const int rr = min(tid >> 2, na - 1), q4 = tid & 3;
            const int row = nu + rr;
#pragma unroll
            for (int u = 0; u < cp::NVQ; ++u) {
                const int j = min(q4 + 4 * u, nv - 1);
                const int hi = max(row, j), lo = min(row, j);
                act_mv[u] = pM[hi * (hi + 1) / 2 + lo];
            }
            if (nc > 0) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 6; ++r) act_av[ct][r] = pAc[(min(ct, nc - 1) * 6 + r) * nv + row];
#pragma unroll
                for (int u = 0; u < cp::KQ; ++u) {
                    const int mcol = min(q4 + 4 * u, k - 1);
#pragma unroll
                    for (int r = 0; r < 6; ++r) act_tv[u][r] = S.force_gen[(u / 3) * 72 * ((nc > 1) ? 1 : 0) + r * 12 + (mcol - 12 * (u / 3) * ((nc > 1) ? 1 : 0))];
                }
            }
        }
    };
    auto issue_act_loads = [&]() __attribute__((always_inline)) {
        issue_act_loads_into(act_mv, act_av, act_tv);
        act_issued = true;
    };

    // ---------------- phase 3: equality constraints, blocked (equality_phase_blocked with N already in place) ----------------
    if (neq > 0) {
        const int m = neq;
        double* rhs = ce0v;
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
            if (e < m && kc == 0) rhs[e] = -(acc + rhs[e]);
        }
        STAMP(22)
        // ---- B = J0' N over N itself: every product is in registers before the first element is replaced
        {
            const int ncg = (m + 3) >> 2;
            const int cpi = tid / ncg, cg = tid - cpi * ncg;
            const int c0 = 2 * cpi, c1i = min(c0 + 1, n - 1);
            const bool actv = c0 < n;
            int kmin = actv ? blk_begin(c0, nv) : n, kmax = actv ? c1i + 1 : 0;
            kmin = wave_min_int(kmin);
            kmax = wave_max_int(kmax);
            double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
            const int c0s = actv ? c0 : 0, c1s = actv ? c1i : 0;
            tile2x4(c.J, c0s, c1s, ldj, Nm + 4 * cg, ldb, kmin, kmax, acc);
            bsync();
            if (actv) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * cg + q < m) {
                        Nm[c0 * ldb + 4 * cg + q] = acc[0][q];
                        if (c0 + 1 < n) Nm[(c0 + 1) * ldb + 4 * cg + q] = acc[1][q];
                    }
            }
        }
        bsync();
        STAMP(5)
        constexpr int NTQ = (SPEC > 0 && kSpecDims[SPEC > 0 ? SPEC - 1 : 0].n <= 64) ? 8 : 10; // row pairs per lane of the QR (iCub, Talos on one foot: n 62)
        bool ok = qr_unified<NTQ>(c, Nm, c.s, c.s + 160);
        if (!ok) status = HQP_ERROR; // redundant equalities
        else {
            // (the specialised kernels only: in the generic one, whose dimensions are run-time values, the loaded values' longer life pushed the allocator
            // past 256 VGPRs -- AGPR copies, one workgroup per CU; tools/kernel_regs.py, profiles/r06/not_kept.txt item 9)
            if constexpr (SPEC != 0) {
                if (D.act_bounds) issue_act_loads();
            }
            bsync();
            STAMP(6)
            if (c.wave == 0) {
                if (m <= 12) solve_y<12>(c, rhs);
                else if (m <= 20) solve_y<20>(c, rhs);
                else solve_y<24>(c, rhs);
            }
            bsync();
            STAMP(19)
            double yy = 0.0;
            if (tid < m) yy = rhs[tid] * rhs[tid];
            if (tid >= 128 && tid - 128 < n) {
                const int kk = tid - 128;
                const double* Jr = c.J + kk * ldj;
                double acc = 0.0;
                for (int e = 0; e < m; ++e) acc = fma(Jr[e], rhs[e], acc);
                c.x[kk] += acc;
            }
            yy = block_sum(c, yy); // (its barrier is also the one the loop's set-up waits for: every read of J, R and y is behind it)
            f_value += 0.5 * yy;
            c.iq = m;
        }
        STAMP(8)
    }

    // ---------------- the actuation rows into registers (from the record: L2) ----------------
    // With actuation bounds they are inequality rows of the loop and are loaded here.  Without (iCub: etc/icub/tasks.yaml has no actuation-bounds task) only the
    // decode's tau = h_a + M_a dv - J_a' f reads them: they are loaded BEHIND the loop then, and the loop does not carry their 38 registers -- which is what the
    // three-per-CU twin of such a stack (168 VGPRs) kept in scratch (round 5: 148 B/lane, a third of the scratch instructions inside the loop).
    ActRegs ar;
    auto load_act_rows = [&]() __attribute__((always_inline)) {
    if (na > 0) {
        const int q4 = tid & 3;
        auto consume = [&](TI (&mv)[cp::NVQ], TI (&av)[2][6], double (&tv)[cp::KQ][6]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < cp::NVQ; ++u) ar.am[u] = (q4 + 4 * u < nv) ? (double)mv[u] : 0.0;
#pragma unroll
            for (int u = 0; u < cp::KQ; ++u) {
                double sacc = 0.0;
                if (nc > 0) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) sacc = fma(tv[u][r], (double)av[u / 3][r], sacc);
                }
                ar.aj[u] = (q4 + 4 * u < k) ? sacc : 0.0;
            }
        };
        if constexpr (SPEC == 0) { // (the generic kernel: loads and use in one place, the values live nowhere else -- see the early issue behind the QR)
            TI mv[cp::NVQ];
            TI av[2][6];
            double tv[cp::KQ][6];
            issue_act_loads_into(mv, av, tv);
            consume(mv, av, tv);
        }
        else {
            if (!act_issued) issue_act_loads();
            consume(act_mv, act_av, act_tv);
        }
    }
    else {
#pragma unroll
        for (int u = 0; u < cp::NVQ; ++u) ar.am[u] = 0.0;
#pragma unroll
        for (int u = 0; u < cp::KQ; ++u) ar.aj[u] = 0.0;
    }
    };
    if (D.act_bounds) load_act_rows();

    STAMP(7)

    // ---------------- phase 4: inequality loop (GI steps 1, 2, 2a-2c) ----------------
    // What the loop is built from costs, measured on one wave of this kernel's shape (tools/ubench/chain_lat.hip, lds_atomics.hip): a dependent
    // f64 FMA 4 cycles; one DPP reduction step on a double 24-38 (a wave-wide sum about 250, a two-pass argmin 500-700); a vector compare
    // feeding a scalar branch 33-44; an LDS round trip 50-65; an LDS atomic to one address 50 + about 10 per participating lane.  So the
    // loop avoids wave-wide reductions and uniform branches, not arithmetic:
    //   * the most violated row is ELECTED through LDS atomics: the few lanes whose row is violated (at most some twenty, usually under
    //     ten) do ds_min_f64 on one slot | barrier | the lanes that hold the minimum do ds_min_u32 with (row << 23 | row descriptor) --
    //     smallest row first, eiquadprog's rule -- | barrier | every thread reads the pick and its descriptor in one word.  The step
    //     length t1 is elected the same way inside wave 3 (no barrier: one wave's LDS operations execute in order);
    //   * z'z is only ever compared with eps: a wave whose rows hold one z_k^2 > eps says so by ballot, and the sum is formed only when no
    //     wave does (never observed outside degenerate picks);
    //   * the common iteration -- full step, constraint accepted -- is decided by ONE branch.
    // One pick with a full step:
    //   P  decode the elected word                                                                                              (E2)
    //   A  d = J'n for the columns from neq on -- from J as it stands plus the PENDING rank-one update of the last accepted
    //      constraint, d = J_old'n - v (w'n): a bound's d is a corrected row of J (one wave, one round trip), a friction row's a
    //      twelve-term sum per column on a quad, an actuation row's a quad per column over the row its owners publish (one more
    //      barrier).  V = d from iq on (zero below), dI = d of the active inequality positions (zero from mi on)                   | bar A
    //   B  waves 0-2, a lane pair per row of J: the pending update J <- J - w v' and z = J2 d2 in ONE pass over the row (16-byte
    //      accesses); wave 3: r = Ri d_I (Ri row-packed, zero from mi on: no masks), t1 elected, |d2|^2, the reflector, t2           | bar B
    //   C  per row k: w_k = tau (z_k - alpha J(k, iq)) becomes the next pending update, x_next into the other x buffer, u and the
    //      new column of Ri on wave 3's lanes (which hold r), s of the next iterate by increments on the lanes that own the rows,
    //      then the election of the next pick                                                                                    | bar E1 | bar E2
    // Partial steps, dual steps and rejected (dependent) constraints leave this path (by then the pending update has been applied:
    // phase B comes first).  A drop is two barriers: D1 move x and u, wave 1 turns row p of Ri into the L rotations' coefficients --
    // rotation l of the pair (p + l, p + l + 1) follows from the running sums S_l = sum_{i<=l} rho_i^2 alone, so a row's new
    // elements are new_l = a_l P_l + b_l x_{l+1} with P_l = sum_{i<=l} rho_i x_i: ONE running sum per row instead of a chain of
    // rotations, and d's own entries by one more scan on that wave | bar | D2 waves 0-1 the rows of J, wave 3 the rows of Ri | bar.
    // Row ownership: thread i owns row i of s when it is a bound or friction row; the actuation rows +-[M_a | -J_a'] of
    // joint rr belong to lanes 0 (+) and 1 (-) of quad rr, which hold tau' in a register anyway.
    const bool act_ineq = D.act_bounds && na > 0; // actuation rows are inequality rows (otherwise tau' is only decoded)
    if (status == -2 && nin2 > 0) {
        const int MM = n - neq;                       // the most inequality constraints that can be active (<= 64, host check)
        const int ne = (n + 1) & ~1;                  // first pad pair of a row of J (ldj >= ne + 2, or the next row's dead columns 0-1), of V and of the pending v: zero for good
        double* const Ri = c.R + 2;                   // inverse of R's inequality block, ROW-packed: (i, j), i <= j, at rio(i) + j - i (R is dead);
                                                      // one zero in front of it: what a row's rotation reads left of its diagonal
        const int ri_size = lp::rio(MM, MM);
        double* const prm = c.R + ((2 + ri_size + 1) & ~1);  // drop: 4 doubles per rotation (a_l, b_l, rho_{l+1}, -) right behind Ri: the matvec's reads past Ri's
                                                              // last row (they meet zeros of d) land here, on numbers that are finite from the loop's first phase on
        // friction rows: the 12 coefficients of row pair e = 17 ct + rr (one sign kept: the other is folded into the use -- negation commutes with every
        // rounding) at fct + e fst.  With enough equalities they sit in J itself (cp::fric_in_j): columns 2..13 of row e, or 2..7 of rows 2 e and 2 e + 1 --
        // equality columns the loop never reads (columns 0-1 may be the pad pair of the row before); otherwise behind the rotation table in the R region.
        const int fric_mode = cp::fric_in_j(n, neq, nc);
        // (R region: two doubles behind the table's end -- the zeroing loop below runs that far, in the same barrier-free phase as the owners' stores
        // of the coefficients, from other waves; derive_compact's `need` holds the + 2)
        double* const fct = fric_mode ? c.J + 2 : prm + 4 * (MM + 2) + 2;
        const int fst = fric_mode == 1 ? ldj : (fric_mode == 2 ? 2 * ldj : 12); // entry to entry
        const int fh = fric_mode == 2 ? ldj : 6;                                 // coefficient m of an entry at (m / 6) fh + m % 6
        double* const Wp = lds + D.o_vec + vm.PW;    // pending update: J(:, pc:) <- J(:, pc:) - w v(pc:)'; w = 0 when there is none
        double* const dI = lds + D.o_vec + vm.DI;    // d at the active inequality positions, zero from mi on (read up to 2 x 64)
        double* const LS = lds + D.o_vec + vm.RED;   // the loop's slots (lp::)
        int* const LSi = reinterpret_cast<int*>(LS);
        double* const EL = lds + D.o_vec + vm.EL;    // election slots (lp::E*; initialised ahead of the equality phase)
        unsigned* const ELu = reinterpret_cast<unsigned*>(EL);
        double* const dbuf0 = lds + D.o_vec + vm.D;      // V of the current pick in one, the pending v in the other
        double* const dbuf1 = lds + D.o_vec + vm.RDINV;  // (1/R(j,j) of the equality phase is dead by now)
        c.red = lds + D.o_vec + vm.LRED;             // block_sum of the rare paths (the set-up's slot is part of LS now)
        c.rslot = 0;
        // ---- loop-time arrays start from zero (every region here was last read behind the equality phase's final barrier)
        // Rows of J exactly n long (derive_compact: n = 2 mod 4, neq >= 2): the zero pad pair the 16-byte row passes need behind column n is the
        // NEXT row's first two columns -- equality columns, which the loop never reads (x = x0 + J1 y was their last use) -- zeroed here.
        if (ldj == n && tid < n) {
            double2v zz;
            zz.x = 0.0;
            zz.y = 0.0;
            *reinterpret_cast<double2v*>(__builtin_assume_aligned(c.J + tid * ldj, 16)) = zz;
        }
        for (int e = tid; e < ((2 + ri_size + 1) & ~1) + 4 * (MM + 2) + 2; e += kThreads) c.R[e] = 0.0; // Ri and the rotation table
        if (tid < 128) dI[tid] = 0.0;
        if (tid < 80) Wp[tid] = 0.0;
        if (tid < VS) {
            dbuf0[tid] = 0.0;
            dbuf1[tid] = 0.0;
        }
        OwnRow own;
        {
            own.meta = -1;
            own.ci0 = 0.0;
            own.coef = nullptr;
            if (tid < nin2) {
                const int mt = meta0;
                const int kind = mt & 3, rr = (mt >> 3) & 255, ct = (mt >> 11) & 15;
                const bool neg = (mt >> 2) & 1;
                if (kind == INEQ_BOUNDS) {
                    own.meta = mt;
                    own.ci0 = neg ? c.bub[rr] : -c.blb[rr];
                }
                else if (kind == INEQ_FORCE) {
                    own.meta = mt;
                    own.ci0 = neg ? S.fric_ub[ct * 17 + rr] : -S.fric_lb[ct * 17 + rr];
                    const double* B = S.fric_mat + (ct * 17 + rr) * 12;
                    double* dst = fct + (ct * 17 + rr) * fst;
                    // (every friction row comes with its negation, [A; -A]: both twins store the same twelve values, so each reads what it wrote
                    // itself in the first evaluation below -- no barrier needed between here and there)
                    double bv[12];
#pragma unroll
                    for (int m = 0; m < 12; ++m) bv[m] = B[m];
#pragma unroll
                    for (int m = 0; m < 12; ++m) dst[(m / 6) * fh + m % 6] = bv[m];
                    own.coef = dst;
                }
            }
        }
#ifdef WBCQP_STAMP_FIRST_PICK
        STAMP(22)
#endif
        const int arr = tid >> 2, aq = tid & 3;
        const bool act_owner = act_ineq && arr < na && aq < 2;
        const int arow = act_owner ? D.act_off + (aq ? na : 0) + arr : -1;
        // the word a row is elected with: row << 23 | descriptor (23 bits: row_meta_pack) -- the smallest word is the smallest row
        const unsigned okey = (own.meta >= 0) ? ((unsigned)tid << 23) | (unsigned)own.meta : 0xffffffffu;
        const unsigned akey = act_owner ? ((unsigned)arow << 23) | (unsigned)row_meta_pack(INEQ_ACTUATION, aq ? 1 : 0, arr, 0, 0) : 0xffffffffu;
        const double aci0 = act_owner ? (aq ? c.tu[arr] : -c.tl[arr]) : 0.0;
        const double asg = aq ? -1.0 : 1.0;
        double s_own = 0.0, s_act = 0.0; // s of the owned rows at the iterate of the last evaluation
        // WBCQP_FLAG_WARM_START (opt-in, include/wbcqp.h): rows that were active at the previous tick's solution are picked first
        // (among the violated ones; the most violated of them first).  Any violated row is a legal Goldfarb-Idnani pick, so only
        // the order of the picks changes -- and with it the add / drop churn a cold start goes through.
        // (WARM: the instantiations behind WBCQP_FLAG_WARM_START.  The default kernels are compiled without the hint's code: carrying it unused cost
        //  them 4 % per pick -- four registers, a second pair of election slots, their masks; profiles/r05/not_kept.txt's KEPT list)
        const bool use_warm = WARM && ga.warm != 0 && ga.amask != nullptr;
        bool warm_own = false, warm_act = false;
        if (use_warm) {
            const unsigned* am = ga.amask + qp * 8;
            if (tid < nin2) warm_own = (am[tid >> 5] >> (tid & 31)) & 1u;
            if (act_owner) warm_act = (am[arow >> 5] >> (arow & 31)) & 1u;
        }

        // a thread's candidate for the next pick: the more violated of its (at most two) eligible rows, as (s, word); {0, ~0}: none
        struct Cand {
            double v, wv;      // most violated eligible row; the same among the warm start's hinted rows
            unsigned key, wkey;
        };
        auto cand_of = [&](bool oka, bool oko) __attribute__((always_inline)) {
            Cand k{0.0, 0.0, 0xffffffffu, 0xffffffffu};
            if (oka) {
                k.v = s_act;
                k.key = akey;
                if (warm_act) {
                    k.wv = s_act;
                    k.wkey = akey;
                }
            }
            if (oko) {
                if (s_own < k.v || (s_own == k.v && okey < k.key)) {
                    k.v = s_own;
                    k.key = okey;
                }
                if (warm_own && (s_own < k.wv || (s_own == k.wv && okey < k.wkey))) {
                    k.wv = s_own;
                    k.wkey = okey;
                }
            }
            return k;
        };
        // s = CI (xp + t zp) + ci0 for the rows this thread owns (kept in s_own / s_act); returns its candidate (row ipx counts as active: its
        // flag is being set while this runs).  tau' of that iterate lands in tact[] and tau_q.
        double tau_q = 0.0; // tau' of the quad's actuation row at the iterate of the last evaluation (all four lanes)
        auto eval_rows = [&](const double* xp, const double* zp, double t, int ipx) __attribute__((always_inline)) {
            bool oka = false, oko = false;
            if (act_ineq && arr < na) { // (quads past the last actuated joint have no row: the fourth wave skips the dot product altogether)
                const double acc = act_dot(c, ar, xp, zp, t);
                tau_q = acc;
                if (aq == 0 && arr < na) tact[arr] = acc;
                if (act_owner) {
                    const double v = fma(asg, acc, aci0);
                    s_act = v;
                    oka = v < 0.0 && !act[arow] && arow != ipx;
                }
            }
            const int mt = own.meta;
            if (mt >= 0) {
                const int kind = mt & 3, ct = (mt >> 11) & 15, col = (mt >> 15) & 255;
                const bool neg = (mt >> 2) & 1;
                double v;
                if (kind == INEQ_BOUNDS) {
                    const double xv = fma(t, zp[col], xp[col]);
                    v = neg ? -xv : xv;
                }
                else {
                    const double* f = xp + nv + 12 * ct;
                    const double* zf = zp + nv + 12 * ct;
                    double a = 0.0;
#pragma unroll
                    for (int m = 0; m < 12; ++m) a = fma(own.coef[(m / 6) * fh + m % 6], fma(t, zf[m], f[m]), a); // (coef: LDS)
                    v = neg ? -a : a;
                }
                v += own.ci0;
                s_own = v;
                oko = v < 0.0 && !act[tid] && tid != ipx;
            }
            return cand_of(oka, oko);
        };
        // the same for the iterate x + t z when s_own / s_act / tau_q hold the values AT x (the common, fused step): only the
        // increments t (CI z) are formed -- half the LDS reads and a third of the arithmetic of the evaluation from scratch.
        // tau' of the final iterate is evaluated from scratch in the decode, so the output never carries the accumulated sum.
        auto eval_rows_inc = [&](const double* zp, double t, int ipx) __attribute__((always_inline)) {
            bool oka = false, oko = false;
            if (act_ineq && arr < na) {
                const double acc = fma(t, act_dot1(c, ar, zp), tau_q);
                tau_q = acc;
                if (act_owner) {
                    const double v = fma(asg, acc, aci0);
                    s_act = v;
                    oka = v < 0.0 && !act[arow] && arow != ipx;
                }
            }
            const int mt = own.meta;
            if (mt >= 0) {
                const int kind = mt & 3, ct = (mt >> 11) & 15, col = (mt >> 15) & 255;
                const bool neg = (mt >> 2) & 1;
                double dz;
                if (kind == INEQ_BOUNDS) {
                    const double zc = zp[col];
                    dz = neg ? -zc : zc;
                }
                else {
                    const double* zf = zp + nv + 12 * ct;
                    double a = 0.0, b2 = 0.0;
#pragma unroll
                    for (int m = 0; m < 12; m += 2) {
                        a = fma(own.coef[(m / 6) * fh + m % 6], zf[m], a);
                        b2 = fma(own.coef[((m + 1) / 6) * fh + (m + 1) % 6], zf[m + 1], b2);
                    }
                    dz = neg ? -(a + b2) : (a + b2);
                }
                const double v = fma(t, dz, s_own);
                s_own = v;
                oko = v < 0.0 && !act[tid] && tid != ipx;
            }
            return cand_of(oka, oko);
        };
        // The election (two barriers): on return every thread holds the most violated candidate's (s, word) -- word ~0: nobody stood --
        // and the same for the hinted rows.  The slots alternate; the one not in use is re-armed in the next pick's phase A.
        int par = 0;
        double el_v = 0.0, el_wv = 0.0;
        unsigned el_key = 0xffffffffu, el_wkey = 0xffffffffu;
        auto elect = [&](const Cand& k) __attribute__((always_inline)) {
            if (k.v < 0.0) lds_min_f64(EL + lp::EMIN + par, k.v);
            if (use_warm && k.wv < 0.0) lds_min_f64(EL + lp::EWMIN + par, k.wv);
            bsync(); // E1
            STAMP(29)
            const double m = EL[lp::EMIN + par];
            if (k.v == m && k.v < 0.0) lds_min_u32(ELu + lp::EKEY + par, k.key);
            double mw = 0.0;
            if (use_warm) {
                mw = EL[lp::EWMIN + par];
                if (k.wv == mw && k.wv < 0.0) lds_min_u32(ELu + lp::EWKEY + par, k.wkey);
            }
            bsync(); // E2
            STAMP(30)
            el_v = m;
            el_key = ELu[lp::EKEY + par];
            if (use_warm) {
                el_wv = mw;
                el_wkey = ELu[lp::EWKEY + par];
            }
            par ^= 1; // (the other slot is re-armed where the pick is set up: phase A, below)
        };
        if (neq == 0) bsync(); // (with equalities: the barrier of the equality phase's last reduction)
        const double psi_tol = (double)nin2 * eps * c1 * c2 * 100.0;
        bool redo = false;       // the election in hand follows a rejected constraint (eiquadprog's l2 again): same outer iteration
        bool excl_dirty = false; // some excl[] entry is 0
        bool slow = false;       // this pick has left the common path: x is updated in place, its snapshot is in c.xold
        double sip = 0.0;        // s(ip) of the constraint being added
        int pc = neq;            // first column of the pending update (w = 0: none)
        double* Vp = dbuf0;      // the pending v (zero below pc, v0 at pc)
        double* Vn = dbuf1;      // V of the pick in hand
        // ---- l1 of the first iteration: s from scratch
        for (int i = tid; i < c.iq; i += kThreads) {
            c.uold[i] = c.u[i];
            c.Aold[i] = c.A[i];
        }
        {
#ifdef WBCQP_STAMP_FIRST_PICK // (tools/phase_profile.py --lib: the way to the first pick split, under the names of phases 5, 19 and 6)
            STAMP(5)
            const Cand k0 = eval_rows(c.x, c.z, 0.0, -1);
            STAMP(19)
            elect(k0);
            STAMP(6)
#else
            elect(eval_rows(c.x, c.z, 0.0, -1));
#endif
        }
        while (true) {
            if (!redo) {
                // l1: a new outer iteration
                slow = false;
                if (--left <= 0) { // (eiquadprog: ++iter >= maxIter)
                    status = HQP_MAX_ITER;
                    break;
                }
                if (excl_dirty) { // (its readers -- a rejection's second election -- lie behind the rejection's barriers)
                    if (tid < nin2) excl[tid] = 1;
                    excl_dirty = false;
                }
                if (!(el_v < 0.0)) { // nothing violated (psi = 0)
                    status = HQP_OPTIMAL;
                    break;
                }
                // psi = sum min(s, 0) decides the termination only when it is small: |psi| >= |s(most violated row)|, so the sum is
                // formed (one more reduction) only when that row alone does not already exceed the tolerance
                if (-el_v <= psi_tol) { // rare: the sum itself decides
                    const double psi = block_sum(c, fmin(0.0, s_own) + fmin(0.0, s_act));
                    if (fabs(psi) <= psi_tol) {
                        status = HQP_OPTIMAL;
                        break;
                    }
                }
            }
            else {
                redo = false;
                if (!(el_v < 0.0)) {
                    status = HQP_OPTIMAL;
                    break;
                }
            }
            unsigned key = el_key;
            sip = el_v;
            if (use_warm && el_wv < 0.0) { // a hinted row is violated: it goes first
                key = el_wkey;
                sip = el_wv;
            }
            STAMP(9)
            // the row n of constraint ip: kind, first column of its support, sign
            const int ip = (int)(key >> 23), mt = (int)(key & 0x7fffffu);
            const int kind = mt & 3, rr = (mt >> 3) & 255, ct = (mt >> 11) & 15, col = (mt >> 15) & 255;
            const bool negrow = (mt >> 2) & 1;
            const double sg = negrow ? -1.0 : 1.0;
            const int iq0 = c.iq;
            if (tid == kThreads - 1) {
                c.u[iq0] = 0.0;
                c.A[iq0] = ip;
                // the slot of the NEXT election is re-armed here: its last readers passed the barrier of the election just held, its next atomics lie
                // behind this pick's barriers
                EL[lp::EMIN + par] = lp::kNone;
                ELu[lp::EKEY + par] = 0xffffffffu;
                if (use_warm) {
                    EL[lp::EWMIN + par] = lp::kNone;
                    ELu[lp::EWKEY + par] = 0xffffffffu;
                }
            }
            STAMP(10)

            // l2a.  d, z, r and the reductions of step 2b are formed ONCE per pick; a partial or dual step then carries them
            // through the drop (rank-one updates) instead of recomputing them.
            // ---- A: d = J'n over the columns from neq on (the equality block's d feeds r of the equality rows, which nothing reads), with
            //      the pending update folded in: d = J_old'n - v (w'n)
            if (kind == INEQ_BOUNDS) {
                if (c.wave == 0) {
                    const int j = neq + min(c.lane, MM - 1);
                    const double dj = sg * fma(-Wp[col], Vp[j], c.J[col * ldj + j]);
                    if (c.lane < MM) {
                        Vn[j] = (j >= iq0) ? dj : 0.0;
                        dI[c.lane] = (j < iq0) ? dj : 0.0;
                    }
                }
            }
            else if (kind == INEQ_FORCE) {
                // a quad per column: three of the row's twelve coefficients per lane
                const int jq = tid >> 2, q4 = tid & 3;
                const int j = neq + min(jq, MM - 1);
                const int k0 = nv + 12 * ct + 3 * q4;
                const double* F = fct + (17 * ct + rr) * fst + (q4 >> 1) * fh + 3 * (q4 & 1); // coefficients 3 q4 .. 3 q4 + 2
                const double* Jb = c.J + k0 * ldj + j;
                const double* wb = Wp + k0;
                const double f0 = F[0], f1 = F[1], f2 = F[2];
                const double j0 = Jb[0], j1 = Jb[ldj], j2 = Jb[2 * ldj];
                const double w0 = wb[0], w1 = wb[1], w2 = wb[2];
                const double vj = Vp[j];
                double a = fma(f2, j2, fma(f1, j1, f0 * j0));
                double fw = fma(f2, w2, fma(f1, w1, f0 * w0));
                a = quad_sum(a);
                fw = quad_sum(fw);
                const double dj = sg * fma(-fw, vj, a); // (the table holds the positive row)
                if (q4 == 0 && jq < MM) {
                    Vn[j] = (j >= iq0) ? dj : 0.0;
                    dI[jq] = (j < iq0) ? dj : 0.0;
                }
            }
            else {
                // the row is in the registers of quad rr: published with n'w, then a quad per column, a quarter of the rows per lane
                if ((tid >> 2) == rr) {
                    // unconditional stores from one address: the M part first (zeros past nv), then the force part on top of
                    // it -- the four lanes are one wave, whose LDS operations execute in program order
                    double* npq = c.np + (tid & 3);
                    const double* wq = Wp + (tid & 3);
                    double wm[cp::NVQ], wf[cp::KQ];
#pragma unroll
                    for (int u = 0; u < cp::NVQ; ++u) wm[u] = wq[4 * u];
#pragma unroll
                    for (int u = 0; u < cp::KQ; ++u) wf[u] = wq[nv + 4 * u];
#pragma unroll
                    for (int u = 0; u < cp::NVQ; ++u) npq[4 * u] = sg * ar.am[u];
                    double* npf = npq + nv;
#pragma unroll
                    for (int u = 0; u < cp::KQ; ++u) npf[4 * u] = -sg * ar.aj[u];
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int u = 0; u < cp::NVQ; ++u) s0 = fma(ar.am[u], wm[u], s0);
#pragma unroll
                    for (int u = 0; u < cp::KQ; ++u) s1 = fma(ar.aj[u], wf[u], s1);
                    const double nw = quad_sum(sg * (s0 - s1));
                    if ((tid & 3) == 0) LS[lp::NPW] = nw;
                }
                bsync(); // the published row
                const int jq = tid >> 2, q4 = tid & 3;
                const int j = neq + min(jq, MM - 1);
                const int qlen = (n + 3) >> 2;
                const int ka = q4 * qlen, kb = min(n, ka + qlen);
                const double npw = LS[lp::NPW], vj = Vp[j];
                double acc = dot8(c.np, 1, c.J + j, ldj, ka, kb);
                acc = quad_sum(acc);
                const double dj = fma(-npw, vj, acc);
                if (q4 == 0 && jq < MM) {
                    Vn[j] = (j >= iq0) ? dj : 0.0;
                    dI[jq] = (j < iq0) ? dj : 0.0;
                }
            }
            STAMP(24)
            bsync(); // A
            STAMP(11)
            // ---- B: pending update + z on waves 0-2; r, t1 and the step's scalars on wave 3
            {
                const int mi = iq0 - neq;
                if (c.wave < 3) {
                    const int idx = tid >> 1, hf = tid & 1; // lane pair per row
                    const int cs = pc & ~1;
                    const int P = (ne - cs) >> 1;           // 16-byte pairs of a row from cs on
                    const int T = max(3, (P + 1) >> 1);     // per lane
                    bool zbig = false;
                    if (idx < n) {
                        double* Jk = c.J + idx * ldj;
                        const double wk = Wp[idx];
                        const int p0 = hf * T, pe = min(P, p0 + T);
                        const int off = iq0 - cs; // 0, 1 or 2: where column iq sits in the first pairs
                        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, stash = 0.0;
                        for (int s0 = 0; s0 < T; s0 += 4) {
                            double2v jv[4], vv[4], dv[4];
                            int cc[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int p = p0 + s0 + u;
                                cc[u] = (p < pe) ? cs + 2 * p : ne; // past the lane's share: the row's pad pair (zeros, written back as zeros)
                                jv[u] = ld2(Jk + cc[u]);
                                vv[u] = ld2(Vp + cc[u]);
                                dv[u] = ld2(Vn + cc[u]);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                jv[u].x = fma(-wk, vv[u].x, jv[u].x);
                                jv[u].y = fma(-wk, vv[u].y, jv[u].y);
                                *reinterpret_cast<double2v*>(__builtin_assume_aligned(Jk + cc[u], 16)) = jv[u];
                            }
                            if (s0 == 0) stash = (off == 0) ? jv[0].x : ((off == 1) ? jv[0].y : jv[1].x);
#pragma unroll
                            for (int u = 0; u < 4; u += 2) {
                                a0 = fma(jv[u].x, dv[u].x, a0);
                                a1 = fma(jv[u].y, dv[u].y, a1);
                                a2 = fma(jv[u + 1].x, dv[u + 1].x, a2);
                                a3 = fma(jv[u + 1].y, dv[u + 1].y, a3);
                            }
                        }
                        double zv = (a0 + a1) + (a2 + a3);
                        zv += dpp_get<0xB1>(zv);
                        if (hf == 0) {
                            c.z[idx] = zv;
                            c.part[idx] = stash; // column iq of J, for the w of phase C
                        }
                        zbig = zv * zv > eps;
                    }
                    STAMP(18)
                    const unsigned long long any = __ballot(zbig);
                    if (c.lane == 0) LSi[lp::BZF + c.wave] = (any != 0ull) ? 1 : 0;
                }
                else {
                    const int i = c.lane;
                    if (i == 0) { // the step length's election slots (their last readers are two barriers back)
                        EL[lp::ET1] = inf;
                        ELu[lp::ET1POS] = 0x7fffffffu;
                    }
                    const double* Rr = Ri + lp::rio(min(i, MM - 1), MM);
                    const double* dq = dI + i;
                    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
                    for (int t0 = 0; t0 < mi; t0 += 8) { // (past mi: zeros on one side or the other)
                        double rv[8], dv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            rv[u] = Rr[t0 + u];
                            dv[u] = dq[t0 + u];
                        }
                        r0 = fma(rv[0], dv[0], r0); r1 = fma(rv[1], dv[1], r1); r2 = fma(rv[2], dv[2], r2); r3 = fma(rv[3], dv[3], r3);
                        r0 = fma(rv[4], dv[4], r0); r1 = fma(rv[5], dv[5], r1); r2 = fma(rv[6], dv[6], r2); r3 = fma(rv[7], dv[7], r3);
                    }
                    const double rl = (i < mi) ? (r0 + r1) + (r2 + r3) : 0.0;
                    const double vq = Vn[neq + min(i, MM - 1)];
                    const double diq = Vn[iq0 < n ? iq0 : n - 1];
                    if (i < mi) c.r[neq + i] = rl;
                    // step 2b's partial step length t1 (dual feasibility): elected among the lanes with r > 0, first position on ties
                    double ratio = inf;
                    if (rl > 0.0) {
                        ratio = ratio_pos(c.u[neq + i], rl);
                        lds_min_f64(EL + lp::ET1, ratio);
                    }
                    const double dn2 = wave_sum((i < MM) ? vq * vq : 0.0); // |d2|^2 = z'n (V is zero below iq)
                    if (rl > 0.0 && ratio == EL[lp::ET1]) lds_min_u32(ELu + lp::ET1POS, (unsigned)(neq + i));
                    // the reflector of a full step, H = I - tau v v' with v = d[iq:] - alpha e_0, and the step length t2
                    double alpha = (iq0 < n) ? diq : 0.0, v0 = 0.0, tau = 0.0;
                    if (iq0 + 1 < n && dn2 > 0.0) {
                        const double inx = rsqrt(dn2);
                        const double nx = dn2 * inx;
                        alpha = (diq >= 0.0) ? -nx : nx;
                        v0 = diq - alpha;
                        tau = fast_rcp(fma(nx, fabs(diq), dn2));
                    }
                    if (c.lane == 0) { // (|d2|^2, alpha) and (v0, tau) are read as pairs: written as pairs
                        double2v o;
                        o.x = dn2;
                        o.y = alpha;
                        *reinterpret_cast<double2v*>(__builtin_assume_aligned(LS + lp::BDN2, 16)) = o;
                        o.x = v0;
                        o.y = tau;
                        *reinterpret_cast<double2v*>(__builtin_assume_aligned(LS + lp::BV0, 16)) = o;
                        LS[lp::BT2C] = ratio_pos(-sip, dn2);
                    }
                }
                STAMP(25)
                bsync(); // B
            }
            STAMP(12)
            double znp, dn2, t1, alpha, v0, tau, t2;
            int lpos;
            bool zok;
            {
                const double2 da = *reinterpret_cast<const double2*>(LS + lp::BDN2); // |d2|^2, alpha
                const double2 vt = *reinterpret_cast<const double2*>(LS + lp::BV0);  // v0, tau
                const double t2c = LS[lp::BT2C];
                const int zf = LSi[lp::BZF] | LSi[lp::BZF + 1] | LSi[lp::BZF + 2];
                t1 = EL[lp::ET1];
                lpos = (int)ELu[lp::ET1POS];
                dn2 = da.x;
                alpha = da.y;
                v0 = vt.x;
                tau = vt.y;
                znp = dn2; // z'n = (J2 d2)'n = d2'(J2'n) = |d2|^2: the second reduction eiquadprog spends on it is the first one again
                zok = zf != 0;
                if (!zok) { // no row's z_k^2 alone exceeds eps: the sum decides (degenerate picks only)
                    const double zk = (tid < n) ? c.z[tid] : 0.0;
                    zok = fabs(block_sum(c, zk * zk)) > eps;
                }
                t2 = zok ? t2c : inf;
            }
            double uiq = 0.0; // u[iq] of the candidate: every thread carries it, LDS sees it when the constraint is added
            int drops = 0;
            while (true) {
                const int iq = c.iq;
                // ---- the common case in one branch: a full step (t2 <= t1, finite) with a constraint that is not (numerically) dependent
                const bool full = (t2 <= t1) & (t2 < inf);
                if (full & (fabs(alpha) > eps * c.R_norm)) {
                    // (iii) full step: add ip to the active set with one reflector H = I - tau v v' (v = d[iq:] - alpha e_0)
                    const double t = t2;
                    const bool reflect = (iq + 1 < n) & (dn2 > 0.0);
                    f_value += t * znp * (0.5 * t + uiq);
                    STAMP(20)
                    // ---- C: the step, the next pending update and the next iterate's s in one phase
                    const double ralpha = fast_rcp(alpha);
                    const int mi = iq - neq;
                    if (tid < n) {
                        const double zk = c.z[tid];
                        // w_k = tau (z_k - alpha J(k, iq)); J(:, iq) was stashed in phase B (or by the last drop)
                        Wp[tid] = reflect ? tau * (zk - alpha * c.part[tid]) : 0.0;
                        c.xold[tid] = fma(t, zk, c.x[tid]); // x_next: the buffers swap below
                    }
                    if (c.wave == 3) {
                        // the lanes that formed r: u, and the new column of the inverse [-r / alpha; 1 / alpha] -- entry (i, mi) of row i
                        const int i = c.lane;
                        if (i < mi) {
                            const double rk = c.r[neq + i];
                            const double un = fma(-t, rk, c.u[neq + i]);
                            c.u[neq + i] = un;
                            c.uold[neq + i] = un;
                            Ri[lp::rio(i, MM) + mi - i] = -rk * ralpha;
                        }
                        if (i == mi) {
                            Ri[lp::rio(mi < MM ? mi : MM - 1, MM)] = ralpha;
                            c.u[iq] = uiq + t;
                            c.uold[iq] = uiq + t;
                            c.Aold[iq] = ip;
                            act[ip] = 1;
                            if (reflect) Vn[iq] = v0; // V becomes the pending v
                        }
                    }
                    if (slow && tid < iq) { // drops moved entries of A and u: the snapshot of the next pick is taken whole
                        c.Aold[tid] = c.A[tid];
                        if (tid < neq) c.uold[tid] = c.u[tid];
                    }
                    Cand nk;
                    if (slow) nk = eval_rows(c.x, c.z, t, ip); // drops moved x since the last evaluation: from scratch
                    else nk = eval_rows_inc(c.z, t, ip);
                    STAMP(23)
                    c.iq = iq + 1;
                    c.R_norm = fmax(c.R_norm, fabs(alpha));
                    pc = iq;
                    {
                        double* vt = Vp;
                        Vp = Vn;
                        Vn = vt;
                    }
                    STAMP(26)
                    elect(nk); // its first barrier ends phase C
                    double* xt = c.x;
                    c.x = c.xold;
                    c.xold = xt;
                    STAMP(14)
                    break; // -> l1
                }
                const double t = fmin(t1, t2);
                if (++drops > n + 2) { // a pick drops at most its active set: anything beyond is a NaN's doing -- give up on the QP instead of spinning
                    status = HQP_ERROR;
                    break;
                }
                if (t >= inf) {
                    status = HQP_INFEASIBLE; // eiquadprog UNBOUNDED (dual) -> tsid INFEASIBLE
                    break;
                }
                if (t2 < inf) f_value += t * znp * (0.5 * t + uiq);
                if (full) {
                    // the constraint is numerically dependent on the active set: eiquadprog adds it, takes it out again (the last position:
                    // no rotation), returns to the saved iterate and picks another.  Nothing has been written yet, so the
                    // reflector is not applied at all (it would only turn the basis of the null space).
                    if (tid == 0) excl[ip] = 0;
                    excl_dirty = true;
                    if (tid < nin2) act[tid] = 0;
                    if (tid < 80) Wp[tid] = 0.0; // no pending update either
                    bsync();
                    for (int i = tid; i < iq; i += kThreads) {
                        const int av = c.Aold[i];
                        c.A[i] = av;
                        if (av >= 0) act[av] = 1;
                        c.u[i] = c.uold[i];
                    }
                    if (slow && tid < n) c.x[tid] = c.xold[tid];
                    pc = iq;
                    bsync();
                    // l2 again: the owners still hold s of the (restored) iterate; the rejected row is excluded
                    elect(cand_of(act_owner && s_act < 0.0 && !act[arow] && excl[arow], own.meta >= 0 && s_own < 0.0 && !act[tid] && excl[tid]));
                    redo = true;
                    break;
                }
                // ---- (ii) dual step / (iii) partial step: move, then drop l.  Two barriers; d, z, r, z'n, |d2|^2, s(ip) follow
                //      the drop by rank-one updates (delta = the entry of d that leaves the active block):
                //      z += delta J(:, iq'), r_i -= delta Z(i, last), z'n += delta^2, |d2|^2 += delta^2, s(ip) += t z'n.
                DSTAMP(36)
                const bool primal = t2 < inf;
                const int l = c.A[lpos];
                const int qq = lpos, p = lpos - neq, mi = iq - neq;
                const int L = iq - 1 - qq; // rotations
                double qP = 0.0; // wave 2, lane i: P_L of row i of Ri (formed in D1, used in D2)
#ifdef WBCQP_STAMP_DROP
                c.st_acc_[45] += L;
                c.st_acc_[46] += mi;
#endif
                {
                    if (tid < n) {
                        const double xv = c.x[tid];
                        if (!slow) c.xold[tid] = xv; // leaving the common path: snapshot of x for a later rejection
                        if (primal) c.x[tid] = fma(t, c.z[tid], xv);
                    }
                    if (c.wave == 3 && c.lane < mi) c.u[neq + c.lane] = fma(-t, c.r[neq + c.lane], c.u[neq + c.lane]);
                    if (tid == kThreads - 1) act[l] = 0;
                    DSTAMP(37)
                    if (c.wave == 1) {
                        // row p of Ri -> the coefficients of the L rotations, and d's own entries through them
                        const int ll = c.lane;
                        const double* Rp = Ri + lp::rio(p, MM);
                        const double rho = (ll <= L) ? Rp[min(ll, L)] : 0.0;
                        const double rho1 = (ll < L) ? Rp[min(ll + 1, L)] : 0.0;
                        const double dl = (ll <= L) ? dI[p + min(ll, L)] : 0.0;
                        const double dl1 = (ll < L) ? dI[p + min(ll + 1, L)] : 0.0;
                        DSTAMP(32)
                        const double Sl = wave_scan_incl(rho * rho);
                        const double Pd = wave_scan_incl(rho * dl);
                        DSTAMP(33)
                        const double S1 = fma(rho1, rho1, Sl);
                        const double rs = rsqrt(Sl), rs1 = rsqrt(S1);
                        DSTAMP(34)
                        if (ll < L) {
                            const double al = -rho1 * (rs * rs1);      // -rho_{l+1} / sqrt(S_l S_{l+1})
                            const double bl = (Sl * rs) * rs1;          //  sqrt(S_l / S_{l+1})
                            double2v o;
                            o.x = al;
                            o.y = bl;
                            *reinterpret_cast<double2v*>(__builtin_assume_aligned(prm + 4 * ll, 16)) = o;
                            prm[4 * ll + 2] = rho1;
                            dI[p + ll] = fma(al, Pd, bl * dl1);
                        }
                        if (ll == L) {
                            const double cl = -rs;                       // the element that leaves: -P_L / sqrt(S_L)
                            const double delta = cl * Pd;
                            dI[p + L] = 0.0;                             // the active block is one shorter
                            Vn[iq - 1] = delta;                          // ... and d's leaving entry heads the null-space part
                            LS[lp::DDELTA] = delta;
                            LS[lp::DCL] = cl;
                        }
                        if (ll == 0) LS[lp::DRHO0] = rho;
                        DSTAMP(35)
                    }
                    else if (c.wave == 2) {
                        // (idle until round 6) P_L = sum_{l <= L} rho_l Ri(i, p + l) of every row i of Ri, by the very sequence of FMAs the rotation's
                        // running sum takes in D2 (rotate_row_ps) -- the same bits: what leaves row i, Z(i, last) = cl P_L, then needs no rotation, and
                        // this wave moves r, u, A and elects t1 in D2 BESIDE wave 3's rotation of the rows instead of behind it on the same wave
                        // (the drop's longest chain: 2.7 k of D2's 2.85 k cycles, profiles/r06/drop_profile_before.txt).  Nobody writes Ri in D1.
                        const int i = c.lane;
                        const int ic = min(i, MM - 1);
                        const double* Rrow = Ri + lp::rio(ic, MM);
                        const double* Rp = Ri + lp::rio(p, MM);
                        const int sh = p - ic;
                        double P = Rp[0] * Rrow[max(sh, -1)];
                        for (int l0 = 0; l0 < L; l0 += 8) {
                            double x1[8], rh[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) { // (reads past L stay inside LDS and are not used)
                                x1[u] = Rrow[max(sh + l0 + u + 1, -1)];
                                rh[u] = Rp[min(l0 + u + 1, L)];
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (l0 + u < L) P = fma(rh[u], x1[u], P);
                        }
                        qP = P;
                    }
                    DCOUNT(44)
                    if (primal) sip = fma(t, znp, sip);
                    uiq += t;
                    slow = true;
                }
                STAMP(27)
                bsync();
                STAMP(13)
                // (read once, here: the next drop's D1 may rewrite these fields as soon as D2's barrier has passed)
                const double delta = LS[lp::DDELTA], cl = LS[lp::DCL], rho0 = LS[lp::DRHO0];
                {
                    if (c.wave < 2) {
                        // rows of J: columns qq .. iq - 1
                        const int kr = min(tid, n - 1);
                        const bool live = tid < n;
                        double* Jk = c.J + kr * ldj + qq;
                        double Pk = rho0 * Jk[0];
                        lp::rotate_row_ps(prm, L, Pk, [&](int jj) { return Jk[jj]; }, [&](int jj, double v) { if (live) Jk[jj] = v; });
                        DSTAMP(38)
                        bool zbig = false;
                        if (live) {
                            const double tj = cl * Pk;
                            Jk[L] = tj;
                            const double zn = fma(delta, tj, c.z[kr]);
                            c.z[kr] = zn;
                            c.part[kr] = tj; // column iq' of J, for the w of phase C
                            zbig = zn * zn > eps;
                        }
                        const unsigned long long any = __ballot(zbig);
                        if (c.lane == 0) LSi[lp::BZF + c.wave] = (any != 0ull) ? 1 : 0;
                    }
                    else if (c.wave == 3) {
                        // rows of Ri.  Row i <= p has all its elements from column p on; row i > p begins at column i (left of it
                        // the clamped address reads a zero: the spare element that ends the row before), moves up one row, and its first new
                        // element (column i - 1) is the fill-in of rotation i - p - 1.  What a row's rotation produces left of its new diagonal is an
                        // exact zero (no element of the row has entered the running sum yet): those stores land on the same spare element.
                        const int i = c.lane;
                        const bool row = i < mi;
                        const bool has = row && i != p;
                        const int i2 = i - ((i > p) ? 1 : 0);
                        const int ic = min(i, MM - 1), i2c = min(i2, MM - 1);
                        const double* Rrow = Ri + lp::rio(ic, MM);              // (i, i): the row's first element; Rrow[-1] reads zero
                        double* Rnew = Ri + lp::rio(i2c, MM);                   // (i2, i2)
                        const int sh = p - ic, sh2 = p - i2c;                   // element (i, p + jj) at Rrow[sh + jj]
                        double Pk = rho0 * Rrow[max(sh, -1)];
                        lp::rotate_row_ps(prm, L, Pk,
                                          [&](int jj) { return Rrow[max(sh + jj, -1)]; },
                                          [&](int jj, double v) { if (has) Rnew[max(sh2 + jj, -1)] = v; });
                        DSTAMP(39)
                        if (has) Rnew[sh2 + L] = 0.0;          // the last column is gone (row i2 keeps its zeros from the new mi on)
                        if (i == mi - 1) Ri[lp::rio(i, MM)] = 0.0; // ... and so is row mi - 1 (it moved up, or it was row p): its storage reads zero again
                    }
                    else {
                        // wave 2: r, u, A, t1 from the P_L it formed in D1 (Z(i, last) = cl P_L leaves the matrix)
                        const int i = c.lane;
                        const bool row = i < mi;
                        const bool has = row && i != p;
                        const int i2 = i - ((i > p) ? 1 : 0);
                        if (i == 0) {
                            EL[lp::ET1] = inf;
                            ELu[lp::ET1POS] = 0x7fffffffu;
                        }
                        const double tj = row ? cl * qP : 0.0;
                        const double rn = row ? fma(-delta, tj, c.r[neq + min(i, mi - 1)]) : 0.0;
                        const double uu = c.u[neq + min(i, mi - 1)];
                        const int aa = c.A[neq + min(i, mi - 1)];
                        if (has) {
                            c.r[neq + i2] = rn;
                            c.u[neq + i2] = uu;
                            c.A[neq + i2] = aa;
                        }
                        if (i == mi) c.A[iq - 1] = ip; // the candidate moves with its position
                        DSTAMP(40)
                        double ratio = inf;
                        if (has && rn > 0.0) {
                            ratio = ratio_pos(uu, rn);
                            lds_min_f64(EL + lp::ET1, ratio);
                        }
                        if (has && rn > 0.0 && ratio == EL[lp::ET1]) lds_min_u32(ELu + lp::ET1POS, (unsigned)(neq + i2));
                        DSTAMP(41)
                    }
                }
                STAMP(28)
                bsync();
                STAMP(15)
                {
                    const int zf = LSi[lp::BZF] | LSi[lp::BZF + 1];
                    t1 = EL[lp::ET1];
                    lpos = (int)ELu[lp::ET1POS];
                    znp = fma(delta, delta, znp);
                    dn2 = fma(delta, delta, dn2);
                    c.iq = iq - 1;
                    zok = zf != 0;
                    if (!zok) {
                        const double zk = (tid < n) ? c.z[tid] : 0.0;
                        zok = fabs(block_sum(c, zk * zk)) > eps;
                    }
                    t2 = zok ? ratio_pos(-sip, znp) : inf;
                    // the reflector of the shorter active set: d[iq'] = delta heads the null-space part now
                    alpha = delta;
                    v0 = 0.0;
                    tau = 0.0;
                    if (iq < n && dn2 > 0.0) { // (iq' + 1 < n)
                        const double inx = rsqrt(dn2);
                        const double nx = dn2 * inx;
                        alpha = (delta >= 0.0) ? -nx : nx;
                        v0 = delta - alpha;
                        tau = fast_rcp(fma(nx, fabs(delta), dn2));
                    }
                    DSTAMP(42)
                }
            }
            if (status != -2) break;
        }
    }
    else if (status == -2) {
        // no inequality rows at all: the equality-constrained minimiser is the solution (eiquadprog: one pass of l1)
        --left; // (iter = 1)
        status = HQP_OPTIMAL;
    }

    STAMP(16)
    // ---------------- phase 5: decode + write-out ----------------
    // tau = h_a + M_a dv - J_a' f   (getActuatorForces)
    if (!D.act_bounds) load_act_rows(); // (no actuation inequality rows: the decode is their only reader; the loads fly under the barrier and the x stores)
    bsync();
    TI* xo = ga.x + qp * n;
    for (int i = tid; i < n; i += kThreads) xo[i] = (TI)c.x[i];
    if (na > 0) {
        TI* to = ga.tau + qp * na;
        const TI hav = ga.h[qr * nv + nu + min(tid, na - 1)];
        // tau' of the final iterate, from scratch (inside the loop it is carried by increments)
        {
            const double acc = act_dot(c, ar, c.x, c.z, 0.0);
            if ((tid & 3) == 0 && (tid >> 2) < na) tact[tid >> 2] = acc;
            bsync();
        }
        if (tid < na) to[tid] = (TI)((double)hav + tact[tid]);
    }
    if (ga.amask) { // bit r = one-sided row r is active at the solution (the next tick's hint under WBCQP_FLAG_WARM_START)
        const signed char* actf = reinterpret_cast<const signed char*>(reinterpret_cast<int*>(lds + D.o_int) + cp::IACT);
        const bool on = (status == HQP_OPTIMAL) && nin2 > 0 && tid < nin2 && actf[tid] != 0;
        const unsigned long long m = __ballot(on);
        if (c.lane == 0) {
            ga.amask[qp * 8 + 2 * c.wave] = (unsigned)(m & 0xffffffffull);
            ga.amask[qp * 8 + 2 * c.wave + 1] = (unsigned)(m >> 32);
        }
    }
    if (tid == 0) {
        ga.status[qp] = status;
        ga.iters[qp] = D.max_iter - left;
        if (ga.objective) ga.objective[qp] = (TI)f_value;
        if (ga.n_active) ga.n_active[qp] = c.iq;
    }
#ifdef WBCQP_STAMPS
    STAMP(17)
    if (tid == WBCQP_STAMP_TID && ga.dbg) // (the stamps are per wave: -DWBCQP_STAMP_TID=192 shows wave 3's view of the phases)
        for (int i = 0; i < kStamps; ++i) ga.dbg[qp * kStamps + i] = c.st_acc_[i];
#endif
}

#endif // __HIPCC__
} // namespace wbcqp

#if defined(WBCQP_X_STOP) && !defined(WBCQP_STAMPS)
#undef STAMP
#define STAMP(i)
#endif
