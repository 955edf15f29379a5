// wbcqp_factor.hpp -- H -> J = U^-1 by blocked elimination in registers (four pivots per synchronisation).
#pragma once

#include "wbcqp_prims.hpp"

namespace wbcqp {
#ifdef __HIPCC__

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
    // A panel buffer holds four values per index k (the four pivot rows of column k of H; the four pivot columns of row k of Y) in two planes:
    // value p of index k at (p >> 1) HP + 2 k + (p & 1).  A thread reads them as two 16-byte pairs; sixteen lanes with consecutive k then
    // cover 256 contiguous bytes per read, and the 8-byte stores of a publishing row land in distinct banks.  (Until round 4: 4 k + p, the
    // two pairs side by side -- lanes k and k + 8 in the same banks.)
    constexpr int G = 1 << LG, PS = NU * G * 4, HP = PS / 2;
    const int par = WLOCAL ? 0 : ((j0n >> 2) & 1);
    const int grp = (j0n & (G - 1)) >> 2;
    if ((ta >> 2) == grp) { // rows j0n + p, p = ta & 3
        double* dst = RB + par * PS + ((ta >> 1) & 1) * HP + (ta & 1);
#pragma unroll
        for (int w = UU; w < NU; ++w) dst[(te + G * w) * 2] = h[UU][w];
    }
    if ((te >> 2) == grp) { // columns j0n + p of Y, p = te & 3
        const int pp = te & 3;
        double* dst = YB + par * PS + ((te >> 1) & 1) * HP + (te & 1);
#pragma unroll
        for (int u = 0; u < UU; ++u) dst[(ta + G * u) * 2] = y[u][UU];
        const int r = ta + G * UU;
        dst[r * 2] = (r < j0n) ? y[UU][UU] : ((r == j0n + pp) ? 1.0 : 0.0);
    }
}

template <int LG, int NU, bool WLOCAL, int JB>
__device__ __forceinline__ void eliminate_block(Ctx& c, double (&h)[NU][NU], double (&y)[NU][NU], int ta, int te, int npad,
                                                double* RB, double* YB, double* dinv, bool dwriter, int dp)
{
    constexpr int G = 1 << LG, PS = NU * G * 4, HP = PS / 2;
    constexpr int JN = (JB + 1 < NU) ? JB + 1 : JB;
    const int jend = min(G * JB + G, npad);
    for (int j0 = G * JB; j0 < jend; j0 += 4) {
        if (WLOCAL) __builtin_amdgcn_wave_barrier();
        else bsync();
        const int par = WLOCAL ? 0 : ((j0 >> 2) & 1);
        const double* rb = RB + par * PS;
        const double* yb = YB + par * PS;
        // operands: pivot block, this thread's row-role and column-role slices, its rows of Y
        // the pivot block: only its upper triangle is read (H(p,q) = hq[q][p>>1][p&1], p <= q), each piece with a load of exactly its width.
        // (Whole 16-byte pairs for all four columns left half-dead destination registers, which the allocator reused for the next pair:
        // a write-after-write hazard the compiler guards with `s_waitcnt lgkmcnt(0)` right behind the first load of every panel step.)
        double2v hq[4][2], fa[NU][2], fe[NU][2], fr[NU][2];
        hq[0][0].x = rb[j0 * 2];
        hq[1][0] = ld2(rb + (j0 + 1) * 2);
        hq[2][0] = ld2(rb + (j0 + 2) * 2);
        hq[2][1].x = rb[HP + (j0 + 2) * 2];
        hq[3][0] = ld2(rb + (j0 + 3) * 2);
        hq[3][1] = ld2(rb + HP + (j0 + 3) * 2);
#pragma unroll
        for (int u = JB; u < NU; ++u) {
            fe[u][0] = ld2(rb + (te + G * u) * 2);
            fe[u][1] = ld2(rb + HP + (te + G * u) * 2);
            fa[u][0] = ld2(rb + (ta + G * u) * 2);
            fa[u][1] = ld2(rb + HP + (ta + G * u) * 2);
        }
#pragma unroll
        for (int u = 0; u <= JB; ++u) {
            fr[u][0] = ld2(yb + (ta + G * u) * 2);
            fr[u][1] = ld2(yb + HP + (ta + G * u) * 2);
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

#endif // __HIPCC__
} // namespace wbcqp
