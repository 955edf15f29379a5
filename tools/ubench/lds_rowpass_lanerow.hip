// Microbenchmark: the inequality loop's ROW PASS (J <- J - w v' and z = J2 d2 in one pass over the trailing columns of J) in three dealings of the same
// arithmetic to a workgroup's waves (csrc/wbcqp_compact.hpp phase B; tools/ubench/lds_rowpass.hip is the pass's present form on 1-4 waves):
//   mode 0  a lane PAIR per row, a lane takes half of the row's 16-byte pairs; v and d come from LDS per pair (the product today: waves 0-2, 148 lanes)
//   mode 1  a LANE per row on waves 0 and 1 (rows 0-63), wave h takes half h of the pairs; rows 64-73 stay a lane pair per row on wave 2; v and d from LDS
//           (all lanes of a wave read ONE address: a broadcast), the halves of z meet by ds_add_rtn_f64
//   mode 2  as 1, with v and d of the wave's half held one element per lane and handed to the FMAs as SGPR operands by v_readlane: a pair costs one load and
//           one store per lane instead of three loads and one store
// The summation order of a row's z is the same in all three: per half (a0 + a1) + (a2 + a3) over blocks of four pairs, then half 0 + half 1.
//   make -C tools/ubench && gpurun -- tools/ubench/_build/lds_rowpass_lanerow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef double double2v __attribute__((ext_vector_type(2)));
constexpr int N = 74, LDJ = 74, REPS = 64;

__device__ __forceinline__ long long now()
{
    long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
__device__ __forceinline__ double2v ld2(const double* p) { return *reinterpret_cast<const double2v*>(__builtin_assume_aligned(p, 16)); }
__device__ __forceinline__ void st2(double* p, double2v v) { *reinterpret_cast<double2v*>(__builtin_assume_aligned(p, 16)) = v; }
__device__ __forceinline__ double rl(double v, int lane) // the value lane `lane` holds, as a wave-uniform (SGPR) operand
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lds_add_rtn(double* p, double v)
{
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// a lane's share of one row: pairs p0 .. pe - 1 from column cs on, v and d from LDS; returns the half's partial of z
__device__ __forceinline__ double pass_lds(double* Jk, const double* Vp, const double* Vn, double wk, int cs, int ne, int T, int p0, int pe)
{
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int s0 = 0; s0 < T; s0 += 4) {
        double2v jv[4], vv[4], dv[4];
        int cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + s0 + u;
            cc[u] = (p < pe) ? cs + 2 * p : ne;
            jv[u] = ld2(Jk + cc[u]);
            vv[u] = ld2(Vp + cc[u]);
            dv[u] = ld2(Vn + cc[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            jv[u].x = fma(-wk, vv[u].x, jv[u].x);
            jv[u].y = fma(-wk, vv[u].y, jv[u].y);
            st2(Jk + cc[u], jv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            a0 = fma(jv[u].x, dv[u].x, a0);
            a1 = fma(jv[u].y, dv[u].y, a1);
            a2 = fma(jv[u + 1].x, dv[u + 1].x, a2);
            a3 = fma(jv[u + 1].y, dv[u + 1].y, a3);
        }
    }
    return (a0 + a1) + (a2 + a3);
}

// the same share with v and d of the half in registers, one element per lane (lane l: column cs + 2 p0 + l), read by v_readlane
__device__ __forceinline__ double pass_sgpr(double* Jk, double vl, double dl, double wk, int cs, int ne, int T, int p0, int pe)
{
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int s0 = 0; s0 < T; s0 += 4) {
        double2v jv[4];
        int cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + s0 + u;
            cc[u] = (p < pe) ? cs + 2 * p : ne;
            jv[u] = ld2(Jk + cc[u]);
        }
        double dx[4], dy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = 2 * (s0 + u); // (past the share: lanes that hold zeros -- the caller loads zeros there)
            const double vx = rl(vl, l), vy = rl(vl, l + 1);
            dx[u] = rl(dl, l);
            dy[u] = rl(dl, l + 1);
            jv[u].x = fma(-wk, vx, jv[u].x);
            jv[u].y = fma(-wk, vy, jv[u].y);
            st2(Jk + cc[u], jv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            a0 = fma(jv[u].x, dx[u], a0);
            a1 = fma(jv[u].y, dy[u], a1);
            a2 = fma(jv[u + 1].x, dx[u + 1], a2);
            a3 = fma(jv[u + 1].y, dy[u + 1], a3);
        }
    }
    return (a0 + a1) + (a2 + a3);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(long long* cyc, double* zout, int pc)
{
    extern __shared__ __align__(16) double lds[];
    double* J = lds;
    double* Vp = J + N * LDJ + 4;
    double* Vn = Vp + 80;
    double* W = Vn + 80;
    double* Z = W + 80;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < N * LDJ + 4; i += 256) lds[i] = 1.0 + 1e-3 * ((i * 37) % 101);
    if (tid < 80) {
        Vp[tid] = (tid < N) ? 0.01 * ((tid * 13) % 17) - 0.05 : 0.0;
        Vn[tid] = (tid < N) ? 0.02 * ((tid * 7) % 19) - 0.1 : 0.0;
        W[tid] = (tid < N) ? 1e-4 * (tid + 1) : 0.0;
        Z[tid] = 0.0;
    }
    __syncthreads();
    const int ne = (N + 1) & ~1, cs = pc & ~1, P = (ne - cs) >> 1, T = (3 > ((P + 1) >> 1)) ? 3 : ((P + 1) >> 1);
    long long t = 0;
    __builtin_amdgcn_s_barrier();
    const long long t0 = now();
    for (int rep = 0; rep < REPS; ++rep) {
        asm volatile("" ::: "memory");
        if (MODE == 0) {
            if (wave < 3) {
                const int idx = tid >> 1, hf = tid & 1;
                if (idx < N) {
                    const int p0 = hf * T, pe = (P < p0 + T) ? P : p0 + T;
                    double zv = pass_lds(J + idx * LDJ, Vp, Vn, W[idx], cs, ne, T, p0, pe);
                    zv += __shfl_xor(zv, 1);
                    if (hf == 0) Z[idx] = zv;
                }
            }
        }
        else {
            if (wave < 2) {
                const int idx = lane, hf = wave;
                const int p0 = hf * T, pe = (P < p0 + T) ? P : p0 + T;
                double zp;
                if (MODE == 1) zp = pass_lds(J + idx * LDJ, Vp, Vn, W[idx], cs, ne, T, p0, pe);
                else {
                    const int col = cs + 2 * p0 + lane;
                    const bool in = col < cs + 2 * pe;
                    const double vl = in ? Vp[col] : 0.0, dl = in ? Vn[col] : 0.0;
                    zp = pass_sgpr(J + idx * LDJ, vl, dl, W[idx], cs, ne, T, p0, pe);
                }
                const double old = lds_add_rtn(Z + idx, zp);
                if (old != 0.0 && (old + zp) * (old + zp) > 1e300) t = 1; // (the product's flag: the second to arrive knows the row's z)
            }
            else if (wave == 2) {
                const int idx = 64 + (lane >> 1), hf = lane & 1;
                if (idx < N) {
                    const int p0 = hf * T, pe = (P < p0 + T) ? P : p0 + T;
                    double zv = pass_lds(J + idx * LDJ, Vp, Vn, W[idx], cs, ne, T, p0, pe);
                    zv += __shfl_xor(zv, 1);
                    if (hf == 0) Z[idx] = zv;
                }
            }
        }
        if (rep + 1 < REPS) { // the next pass starts from z = 0 (in the product: zeroed a phase earlier)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            if (tid < 80) Z[tid] = 0.0;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = now();
    if (lane == 0) cyc[wave] = t1 - t0 + t;
    __syncthreads();
    if (tid < N) zout[tid] = Z[tid];
    if (tid < N) zout[N + tid] = J[tid * LDJ + N - 1 - (tid % 7)];
}

// the barriers' own cost in the loop above (two per pass), to be subtracted
__global__ __launch_bounds__(256) void kbar(long long* cyc)
{
    __shared__ double Z[80];
    const int tid = threadIdx.x;
    __builtin_amdgcn_s_barrier();
    const long long t0 = now();
    for (int rep = 0; rep < REPS; ++rep) {
        asm volatile("" ::: "memory");
        if (rep + 1 < REPS) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            if (tid < 80) Z[tid] = 0.0;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = now();
    if ((tid & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, long long* dc, double* dz, double* ref, long long bar)
{
    const size_t lds = (N * LDJ + 4 + 4 * 80) * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int pc : {18, 30, 40, 50}) {
        long long c[4];
        double z[2 * N];
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), lds, 0, dc, dz, pc);
        hipDeviceSynchronize();
        hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
        hipMemcpy(z, dz, sizeof(z), hipMemcpyDeviceToHost);
        if (MODE == 0 && pc == 18) memcpy(ref, z, sizeof(z));
        int same = (pc == 18) ? !memcmp(ref, z, sizeof(z)) : -1;
        printf("%-44s pc %2d: %6.0f ticks per pass (waves: %5.0f %5.0f %5.0f %5.0f; two barriers %4.0f subtracted)%s\n", name, pc,
               (double)(c[0] - bar) / REPS, (double)(c[0] - bar) / REPS, (double)(c[1] - bar) / REPS, (double)(c[2] - bar) / REPS, (double)(c[3] - bar) / REPS,
               (double)bar / REPS, same < 0 ? "" : (same ? "  z, J: the same bits as mode 0" : "  z, J: DIFFER from mode 0"));
    }
}

int main()
{
    long long* dc;
    double* dz;
    hipMalloc(&dc, 4 * sizeof(long long));
    hipMalloc(&dz, 2 * N * sizeof(double));
    long long bar[4];
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kbar, dim3(1), dim3(256), 0, 0, dc);
    hipDeviceSynchronize();
    hipMemcpy(bar, dc, sizeof(bar), hipMemcpyDeviceToHost);
    double ref[2 * N];
    run<0>("0 lane pair per row, waves 0-2 (today)", dc, dz, ref, bar[0]);
    run<1>("1 lane per row x half per wave, v d from LDS", dc, dz, ref, bar[0]);
    run<2>("2 lane per row x half per wave, v d by readlane", dc, dz, ref, bar[0]);
    return 0;
}
