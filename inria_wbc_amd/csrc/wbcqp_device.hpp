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
//
// Files: wbcqp_types.hpp (shared structures), wbcqp_prims.hpp (wave primitives, context, inner products),
// wbcqp_factor.hpp (H -> J), wbcqp_equality.hpp (equality phase), wbcqp_activeset.hpp (active-set pieces),
// wbcqp_integrate.hpp (after the path); this file: one QP on one workgroup (solve_one), the kernels.
#pragma once

#include "wbcqp_types.hpp"
#include "wbcqp_prims.hpp"
#include "wbcqp_factor.hpp"
#include "wbcqp_equality.hpp"
#include "wbcqp_activeset.hpp"
#include "wbcqp_integrate.hpp"
#include "wbcqp_compact.hpp"

namespace wbcqp {
#ifdef __HIPCC__


// ------------------------------------------------------------------------------------------------
// Phases 1-2 for a stack with a "torque" or a "cop" task (tasks.cpp:227-271, :156-178): those level-1 rows couple dv with f and
// one contact's forces with another's, so H is ONE n x n matrix (n <= 80, checked on the host) -- a 16 x 16 thread grid keeps the
// positions (ta + 16 u, te + 16 w), u <= w < 5, in registers, accumulates every level-1 row into them and eliminates with the
// same eliminate_block the dense seam uses (wbcqp_dense.hpp).  Sources of the rows: the staged motion rows (columns < nv), the
// actuation rows scale_j [M_a(joint_j, :) | -J_a(:, joint_j)'] read in place from M and Jc in LDS, the cop rows from the record
// (c.xold), the selection rows' diagonal (c.z), the force-regularisation blocks w F'F.  g: dv part formed here, force part = what
// phase 0 left (force regularisation) minus these rows' share.  Leaves J = U^-1 (upper triangular, dense), c1 = tr H, c2 = tr J.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void assemble_eliminate_dense(Ctx& c, const DevStruct& S, const double* As, double& c1, double& c2)
{
    const int tid = c.tid, n = c.n, nv = c.nv, nu = c.nu, nc = c.nc, k = c.k, ldj = c.ldj, ldm = c.ldm, ldc = c.ldc;
    const int n_dense = S.n_dense, n_sel = S.n_sel;
    const int ta = tid >> 4, te = tid & 15;
    double h[5][5], y[5][5], gacc[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        gacc[u] = 0.0;
#pragma unroll
        for (int w = 0; w < 5; ++w) {
            h[u][w] = 0.0;
            y[u][w] = 0.0;
        }
    }
    // (a) motion rows: columns < nv <= 64 live in the tile's first four positions (what is staged past nv was never written)
    {
        const double* WB = As + n_dense * 64;
        for (int r = 0; r < n_dense; ++r) {
            const double2v ai0 = ld2(As + r * 64 + ta * 2), ai1 = ld2(As + r * 64 + ta * 2 + 32);
            const double2v aj0 = ld2(As + r * 64 + te * 2), aj1 = ld2(As + r * 64 + te * 2 + 32);
            const double2v wb = ld2(WB + 2 * r);
            const double a[4] = {ai0.x, ai0.y, ai1.x, ai1.y};
            const double b[4] = {aj0.x, aj0.y, aj1.x, aj1.y};
            double am[4], bw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                am[u] = (ta + 16 * u < nv) ? a[u] : 0.0;
                bw[u] = (te + 16 * u < nv) ? b[u] * wb.x : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = u; w < 4; ++w) h[u][w] = fma(am[u], bw[w], h[u][w]);
#pragma unroll
            for (int w = 0; w < 4; ++w) gacc[w] = fma(bw[w], wb.y, gacc[w]);
        }
    }
    // (b) torque task: row j = scale_j [M(nu + joint_j, :) | -Jc(:, nu + joint_j)'], rhs scale_j tau_ref_j - scale_j h_a(joint_j)
    if (S.n_acteq > 0) {
        const double wt = c.w[S.acteq_task];
        const double* bt = c.b1 + n_dense + n_sel + 6 * nc;
        for (int j = 0; j < S.n_acteq; ++j) {
            const int ax = nu + S.acteq_joint[j];
            const double sc = S.acteq_scale[j];
            const double bj = bt[j] - sc * c.h[ax];
            const double ws = wt * sc * sc, wg = wt * sc * bj;
            double ai[5], aj[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int ci = ta + 16 * u, cj = te + 16 * u;
                const double mi = c.M[ax * ldm + min(ci, nv - 1)], mj = c.M[ax * ldm + min(cj, nv - 1)];
                const double ji = (k > 0) ? c.Jc[min(max(ci - nv, 0), max(k - 1, 0)) * ldc + ax] : 0.0;
                const double jj = (k > 0) ? c.Jc[min(max(cj - nv, 0), max(k - 1, 0)) * ldc + ax] : 0.0;
                ai[u] = (ci < nv) ? mi : ((ci < n) ? -ji : 0.0);
                aj[u] = (cj < nv) ? mj : ((cj < n) ? -jj : 0.0);
            }
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int w = u; w < 5; ++w) h[u][w] = fma(ws * ai[u], aj[w], h[u][w]);
#pragma unroll
            for (int w = 0; w < 5; ++w) gacc[w] = fma(wg, aj[w], gacc[w]);
        }
    }
    // (c) cop task: three rows over the force columns, staged in c.xold by phase 0
    if (S.cop_task >= 0) {
        const double wt = c.w[S.cop_task];
        const double* bcop = c.b1 + n_dense + n_sel + 6 * nc + S.n_acteq;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double* row = c.xold + r * k;
            const double wb = wt * bcop[r];
            double ai[5], aj[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int ci = ta + 16 * u, cj = te + 16 * u;
                const double vi = row[min(max(ci - nv, 0), k - 1)], vj = row[min(max(cj - nv, 0), k - 1)];
                ai[u] = (ci >= nv && ci < n) ? vi : 0.0;
                aj[u] = (cj >= nv && cj < n) ? vj : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int w = u; w < 5; ++w) h[u][w] = fma(wt * ai[u], aj[w], h[u][w]);
#pragma unroll
            for (int w = 0; w < 5; ++w) gacc[w] = fma(wb, aj[w], gacc[w]);
        }
    }
    // (d) g: the dv part is formed here (selection rows' share in c.d), the force part already holds the force regularisation's
    if (ta == 0) {
#pragma unroll
        for (int w = 0; w < 5; ++w) {
            const int col = te + 16 * w;
            if (col < nv) c.g[col] = -gacc[w] - c.d[col];
            else if (col < n) c.g[col] -= gacc[w];
        }
    }
    // (e) force regularisation w F'F on each contact's 12 x 12 block, the selection rows and the regulariser on the diagonal
    double trace = 0.0;
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int w = u; w < 5; ++w) {
            const int r = ta + 16 * u, q = te + 16 * w;
            if (nc > 0) {
                const int cr = max(r - nv, 0) / 12, cq = max(q - nv, 0) / 12;
                const bool in = r >= nv && q >= nv && r < n && q < n && cr == cq;
                const int cs = min(cr, nc - 1);
                const double fv = S.ftf[cs * 144 + min(max(r - nv - 12 * cs, 0), 11) * 12 + min(max(q - nv - 12 * cs, 0), 11)];
                const double wf = c.w[S.forcereg_task[cs]];
                if (in) h[u][w] = fma(wf, fv, h[u][w]);
            }
            if (r == q && r < n) {
                h[u][w] += ((r < nv) ? c.z[min(r, nv - 1)] : 0.0) + S.hessian_reg;
                trace += h[u][w];
            }
            if (r >= n || q >= n) h[u][w] = (r == q) ? 1.0 : 0.0; // positions past the matrix: identity (inert pivots)
        }
    c1 = block_sum(c, trace);
    // ---- H -> J = U^-1 (wbcqp_factor.hpp), panels in the s / stash / part slots (11 x 128 doubles, contiguous)
    double* RB = c.s;
    double* YB = c.s + 2 * 5 * 16 * 4;
    publish_panel<4, 5, false, 0>(c, h, y, ta, te, 0, RB, YB);
    eliminate_block<4, 5, false, 0>(c, h, y, ta, te, (n + 3) & ~3, RB, YB, c.dinv, tid >= 128 && tid < 132, tid & 3);
    bsync();
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int w = u; w < 5; ++w) {
            const int r = ta + 16 * u, q = te + 16 * w;
            if (q < n && r < q) c.J[r * ldj + q] = y[u][w] * c.dinv[q];
            else if (r == q && r < n) c.J[r * ldj + r] = c.dinv[r];
        }
    double tr2 = 0.0;
    for (int i = tid; i < n; i += kThreads) tr2 += c.dinv[i];
    c2 = block_sum(c, tr2);
}

// ------------------------------------------------------------------------------------------------
// one QP on one workgroup of 256 threads
// ------------------------------------------------------------------------------------------------
template <typename TI>
__device__ __forceinline__ void solve_one(const GroupArgs<TI>& ga, const DevStruct& S, const int b, double* lds, const int tid)
{
    Ctx c;
    c.S = &S;
    c.tid = tid;
    c.lane = tid & (kWave - 1);
    c.wave = uni(tid >> 6);
    c.rslot = 0;
    c.nv = S.nv; c.na = S.na; c.nc = S.nc; c.k = S.k; c.n = S.n; c.nu = S.nu;
    c.nblk = S.dense_h ? S.n : S.nv;
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
        if (S.cop_task >= 0 && tid < 3 * k) c.xold[tid] = (double)ga.Acop[qp * (size_t)(3 * k) + tid]; // cop rows (slot idle until the loop)
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
    if (S.dense_h) assemble_eliminate_dense(c, S, As, c1, c2);
    else { // nv <= 64 is checked on the host
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
            const int kb0 = blk_begin(ic, c.nblk), len = ic + 1 - kb0, hl = (len + 1) >> 1;
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
            const int ce = blk_end(ic, c.nblk), len = ce - ic, hl = (len + 1) >> 1;
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
    if (ga.amask) { // bit r = one-sided row r (r < 256) is active at the solution; this layout does not take the mask as a hint
        const bool on = (status == HQP_OPTIMAL) && nin2 > 0 && tid < nin2 && c.iai[tid] == -1;
        const unsigned long long m = __ballot(on);
        if ((tid & 63) == 0) {
            ga.amask[qp * 8 + 2 * (tid >> 6)] = (unsigned)(m & 0xffffffffull);
            ga.amask[qp * 8 + 2 * (tid >> 6) + 1] = (unsigned)(m >> 32);
        }
    }
    if (tid == 0) {
        ga.status[qp] = status;
        ga.iters[qp] = iter;
        if (ga.objective) ga.objective[qp] = (TI)f_value;
        if (ga.n_active) ga.n_active[qp] = c.iq;
    }
#ifdef WBCQP_STAMPS
    STAMP(17)
    if (tid == WBCQP_STAMP_TID && ga.dbg) // (the stamps are per wave: -DWBCQP_STAMP_TID=192 shows wave 3's view of the phases)
        for (int i = 0; i < kStamps; ++i) ga.dbg[qp * kStamps + i] = c.st_acc_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// the kernel: grid = total QPs, block = 256 threads = four wavefronts = one QP
// ------------------------------------------------------------------------------------------------
template <typename TI, bool CP, int SPEC = 0>
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
    if constexpr (CP) solve_one_compact<TI, SPEC>(ga, S, b, lds, threadIdx.x);
    else solve_one<TI>(ga, S, b, lds, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// the same, with the workgroups handing out the QPs themselves: grid = as many workgroups as the chip holds at once, each
// takes the next entry of the launch order from a counter in HBM until the order is exhausted.
// Why: measured (tools/ubench/dispatch_order.hip), the hardware deals workgroup i to XCD i % 8 and, inside the XCD, to
// shader engine (i / 8) % 4, strictly in index order -- a workgroup waits for a CU of *its* engine (8 CUs) while CUs of
// the other engines stand free, and everything behind it in that XCD waits with it.  A queue is list scheduling over
// all 256 CUs.  *queue: positions handed out beyond the first one of each workgroup; the last fetch of a launch zeroes it
// for the next launch (stream order makes that visible; no memset on the path, and a captured launch replays as it is).
// ------------------------------------------------------------------------------------------------
template <typename TI, bool CP, int SPEC = 0>
__global__ __launch_bounds__(kThreads) void solve_queue_kernel(const GroupTable<TI> tab, int* queue, const int total)
{
    extern __shared__ __align__(16) double lds[];
    __shared__ int next_qp;
    for (bool first = true;; first = false) {
        if (threadIdx.x == 0) {
            // a workgroup's first position is its own index: 256 atomics on one address at the start of the launch go
            // through one at a time (measured: 6 us per launch).  Every workgroup ends on exactly one fetch past the end,
            // so the counter stops at `total`; whoever draws total - 1 made the last fetch of the launch and zeroes it.
            int pos = (int)blockIdx.x;
            if (!first) {
                const int c = atomicAdd(queue, 1);
                pos = (int)gridDim.x + c;
                if (c == total - 1) *queue = 0;
            }
            next_qp = pos < total ? (tab.order ? tab.order[pos] : pos) : -1;
        }
        bsync();
        int b = uni(next_qp), gi = 0;
        if (b < 0) break;
        while (gi + 1 < tab.n && b >= tab.g[gi].count) {
            b -= tab.g[gi].count;
            ++gi;
        }
        const GroupArgs<TI>& ga = tab.g[gi];
        // the thread index is made opaque once per QP: otherwise every per-thread offset of solve_one is hoisted out of
        // this loop and stays live across it (measured: 256 VGPRs + 146 AGPRs instead of 240 + 0; build.py refuses that)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if constexpr (CP) solve_one_compact<TI, SPEC>(ga, ga.st, b, lds, tid);
        else solve_one<TI>(ga, ga.st, b, lds, tid);
        bsync(); // the next QP reuses every byte of LDS, next_qp included
    }
}

// ------------------------------------------------------------------------------------------------
// The two compact kernels with the warm start's pick hint compiled IN (WBCQP_FLAG_WARM_START, opt-in: include/wbcqp.h): the generic one and Talos's.  The default
// kernels above are compiled without it -- measured (tools/cmp_variants.sh, twice in one call): the stream's longest QP 203.4 -> 196.9 us, 3.52 -> 3.38 us per pick,
// B = 8192 7.58 -> 7.71 M QP/s with the hint's code gone from a kernel that never takes the hint.  Same bodies.
// ------------------------------------------------------------------------------------------------
template <typename TI, int SPEC = 0>
__global__ __launch_bounds__(kThreads) void solve_kernel_warm(const GroupTable<TI> tab)
{
    extern __shared__ __align__(16) double lds[];
    int b = tab.order ? tab.order[blockIdx.x] : (int)blockIdx.x, gi = 0;
    while (gi + 1 < tab.n && b >= tab.g[gi].count) {
        b -= tab.g[gi].count;
        ++gi;
    }
    const GroupArgs<TI>& ga = tab.g[gi];
    solve_one_compact<TI, SPEC, true>(ga, ga.st, b, lds, threadIdx.x);
}

template <typename TI, int SPEC = 0>
__global__ __launch_bounds__(kThreads) void solve_queue_kernel_warm(const GroupTable<TI> tab, int* queue, const int total)
{
    extern __shared__ __align__(16) double lds[];
    __shared__ int next_qp;
    for (bool first = true;; first = false) {
        if (threadIdx.x == 0) {
            int pos = (int)blockIdx.x;
            if (!first) {
                const int c = atomicAdd(queue, 1);
                pos = (int)gridDim.x + c;
                if (c == total - 1) *queue = 0;
            }
            next_qp = pos < total ? (tab.order ? tab.order[pos] : pos) : -1;
        }
        bsync();
        int b = uni(next_qp), gi = 0;
        if (b < 0) break;
        while (gi + 1 < tab.n && b >= tab.g[gi].count) {
            b -= tab.g[gi].count;
            ++gi;
        }
        const GroupArgs<TI>& ga = tab.g[gi];
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        solve_one_compact<TI, SPEC, true>(ga, ga.st, b, lds, tid);
        bsync();
    }
}

// ------------------------------------------------------------------------------------------------
// The queue kernel for stacks whose workgroup leaves room for a THIRD one on a CU (compact layout, LDS <= a third of the CU's 160 KB:
// n <= ~52 today).  Same body, compiled for three waves per SIMD: 168 VGPRs, what does not fit goes to scratch (208-232 bytes per lane;
// build.py admits scratch for this kernel only).  Measured before it was built in (tools/occ3_probe.py, profiles/r05/occ3_probe.txt:
// iCub on one foot, n 50, B = 8192): two per CU without scratch 10.55 M QP/s; this build held at two per CU 9.59 M (the spills cost 9 %);
// this build at three per CU 12.24 M (+16 %, same bits) -- the chains of a third QP fill the issue slots two leave idle.
// Taken only for stacks WITHOUT actuation bounds: with them the loop's 38 registers of actuation rows go to scratch and three per CU loses (Talos on one
// foot: 8.03 M QP/s against 9.32 M at two; wbcqp_api.hip, kThree).
// ------------------------------------------------------------------------------------------------
template <typename TI, int SPEC = 0>
__global__ __launch_bounds__(kThreads, 3) void solve_queue3_kernel(const GroupTable<TI> tab, int* queue, const int total)
{
    extern __shared__ __align__(16) double lds[];
    __shared__ int next_qp;
    for (bool first = true;; first = false) {
        if (threadIdx.x == 0) {
            int pos = (int)blockIdx.x;
            if (!first) {
                const int c = atomicAdd(queue, 1);
                pos = (int)gridDim.x + c;
                if (c == total - 1) *queue = 0;
            }
            next_qp = pos < total ? (tab.order ? tab.order[pos] : pos) : -1;
        }
        bsync();
        int b = uni(next_qp), gi = 0;
        if (b < 0) break;
        while (gi + 1 < tab.n && b >= tab.g[gi].count) {
            b -= tab.g[gi].count;
            ++gi;
        }
        const GroupArgs<TI>& ga = tab.g[gi];
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        solve_one_compact<TI, SPEC>(ga, ga.st, b, lds, tid);
        bsync();
    }
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
    bsync();
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
    bsync();
    if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < 64; ++k) {
            start[k] = acc;
            acc += hist[k];
        }
    }
    bsync();
    for (int i = tid; i < total; i += 1024) order[atomicAdd(&start[key_of(i)], 1)] = i;
}

// ------------------------------------------------------------------------------------------------
// Packed launch order for the queue (host model, same integer arithmetic: inria_wbc_amd/launch_order.py).
// Longest-first is list scheduling; with about four QPs per CU its makespan sits up to one short QP above the mean
// (measured on the bench batch: 1.15 x the perfect split).  A QP costs about kSetupIters + iterations units, so the batch
// is bin-packed instead: 32 sub-problems (the longest-first order dealt out boustrophedon, so that they are alike),
// each packing its share into its 32nd of the resident workgroups for a capacity C -- a bin takes the largest QP left,
// then QPs nearest to room / (number of QPs the room holds at the mean size left), the last two chosen to fill the room
// exactly when such a pair exists.  Four capacities are tried at once (one wave each, all scalar: the class counts live
// in the lanes of one register, the classes in play in a 64-bit mask), the smallest that fits wins, and the QPs are
// emitted by start time, the sub-problems interleaved.  The queue then reproduces the packing, or does better where
// the real costs differ from the predicted ones.
// ------------------------------------------------------------------------------------------------
constexpr int kSetupIters = 7;   // setup of a Talos QP in units of one active-set iteration (115.8 k against 14.7 k cycles)
constexpr int kPackSubs = 32, kPackTrials = 4, kPackMaxItems = 128;

struct PackArgs {
    const int* iters;     // iteration counts of the launch just finished
    const int* order_in;  // longest-first order (schedule_kernel)
    int* order_out;       // packed order
    int total;            // multiple of kPackSubs, total / kPackSubs <= kPackMaxItems
    int bins;             // resident workgroups / kPackSubs
};

namespace pack {
using u64 = unsigned long long;
__device__ __forceinline__ int hi(u64 m) { return m ? 63 - __builtin_clzll(m) : -1; }
__device__ __forceinline__ int lo(u64 m) { return m ? __builtin_ctzll(m) : -1; }
__device__ __forceinline__ u64 below(int k) { return k < 0 ? 0ull : (k >= 63 ? ~0ull : ((1ull << (k + 1)) - 1ull)); }
// the set bit nearest to t2 / 2, the larger one on a tie; -1 if none
__device__ __forceinline__ int nearest(u64 m, int t2)
{
    const int fl = min(t2 >> 1, 63);
    const int kl = hi(m & below(fl)), kh = lo(m & ~below(fl));
    if (kl < 0) return kh;
    if (kh < 0) return kl;
    return (2 * kh - t2 <= t2 - 2 * kl) ? kh : kl;
}
// x / j for 0 <= x < 8192, 1 <= j <= 8 without the divider sequence (exact in that range)
__device__ __forceinline__ int div_small(int x, int j)
{
    const int r = j == 1 ? 65536 : j == 2 ? 32768 : j == 3 ? 21846 : j == 4 ? 16384 : j == 5 ? 13108 : j == 6 ? 10923 : j == 7 ? 9363 : 8192;
    return (x * r) >> 16;
}
} // namespace pack

// grid = kPackSubs, block = 256: one sub-problem per workgroup, one capacity per wave.  One wave per SIMD matters: the
// packing is scalar code, a CU has one scalar unit, and with sixteen such waves on a CU (the first version) each of them
// issued once in sixteen cycles -- 63 us for what takes a fraction of that spread over 32 CUs.
__global__ __launch_bounds__(256) void pack_order_kernel(const PackArgs pa)
{
    using namespace pack;
    __shared__ int s_job[kPackMaxItems];
    __shared__ int s_cls[kPackMaxItems];
    __shared__ unsigned s_ev[kPackTrials][kPackMaxItems];
    __shared__ int s_ok[kPackTrials];
    constexpr int A = kSetupIters;
    const int tid = threadIdx.x, lane = tid & 63;
    const int trial = uni(tid >> 6);
    const int sub = blockIdx.x, nw = pa.total / kPackSubs;
    if (tid < nw) {
        // entry e of sub-problem `sub`: rank 32 e + sub on even rounds, 32 e + 31 - sub on odd ones
        const int job = pa.order_in[kPackSubs * tid + ((tid & 1) ? kPackSubs - 1 - sub : sub)];
        s_job[tid] = job;
        s_cls[tid] = min(max(pa.iters[job], 0), 63);
    }
    bsync();
    // class counts in the lanes (lane k: class k), gend(k) = number of entries of class >= k (the entries are sorted)
    int cnt = 0;
    for (int e = 0; e < nw; ++e) cnt += (s_cls[e] == lane);
    int gend = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_down(gend, d);
        if (lane + d < 64) gend += v;
    }
    int tot = cnt * (lane + A);
    for (int d = 32; d; d >>= 1) tot += __shfl_xor(tot, d);
    tot = uni(tot);
    u64 avail = __ballot(cnt > 0);
    const int c0 = max((tot + pa.bins - 1) / pa.bins, hi(avail) + A);
    const int cap = c0 + trial;
    int n = nw, ne = 0;
    for (int bin = 0; bin < pa.bins && n > 0; ++bin) {
        int room = cap;
        bool first = true;
        while (n > 0) {
            const int kmin = lo(avail);
            if (room < kmin + A) break;
            const int kmax = min(63, room - A);
            int k;
            if (first) {
                k = hi(avail & below(kmax));
                first = false;
            }
            else {
                int j = 1;
                while (j < 8 && (2 * j + 1) * tot <= 2 * room * n) ++j;
                while (j > 1 && j * (kmin + A) > room) --j;
                if (j == 1) k = hi(avail & below(kmax));
                else {
                    const int lim = min(kmax, room - (j - 1) * (kmin + A) - A);
                    const int t2 = div_small(2 * room, j) - 2 * A;
                    k = nearest(avail & below(lim), t2);
                    if (j == 2) {
                        // a pair that fills the room exactly: k2 + k3 = need, both in play (twice if k2 == k3)
                        const int need = room - 2 * A;
                        if (need >= 0 && need <= 126) {
                            const u64 rev = __builtin_bitreverse64(avail); // bit (63 - k) of rev = bit k of avail
                            u64 cand = avail & (need <= 63 ? rev >> (63 - need) : rev << (need - 63));
                            if (!(need & 1) && (need >> 1) <= 63 && __builtin_amdgcn_readlane(cnt, need >> 1) < 2)
                                cand &= ~(1ull << (need >> 1));
                            const int bp = nearest(cand, t2);
                            if (bp >= 0) k = max(bp, need - bp);
                        }
                    }
                }
            }
            if (k < 0) break;
            const int left = __builtin_amdgcn_readlane(cnt, k);
            const int li = __builtin_amdgcn_readlane(gend, k) - left;
            if (lane == 0) s_ev[trial][ne] = ((unsigned)(cap - room) << 16) | ((unsigned)ne << 8) | (unsigned)li;
            ++ne;
            cnt -= (lane == k);
            if (left == 1) avail &= ~(1ull << k);
            --n;
            tot -= k + A;
            room -= k + A;
        }
    }
    if (lane == 0) s_ok[trial] = (n == 0);
    bsync();
    int win = -1;
    for (int t = kPackTrials - 1; t >= 0; --t)
        if (s_ok[t]) win = t;
    if (tid < nw) {
        if (win < 0) pa.order_out[kPackSubs * tid + sub] = s_job[tid]; // no capacity fitted: longest-first, dealt out
        else {
            const unsigned key = s_ev[win][tid];
            int rank = 0;
            for (int e = 0; e < nw; ++e) rank += (s_ev[win][e] < key);
            pa.order_out[kPackSubs * rank + sub] = s_job[key & 255u];
        }
    }
}

#endif // __HIPCC__
} // namespace wbcqp
