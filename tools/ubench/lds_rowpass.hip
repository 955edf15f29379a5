// Microbenchmark: what the inequality loop's ROW PASS (phase B of a pick: J <- J - w v' and z = J2 d2 in one pass, a lane pair per row, 16-byte accesses;
// csrc/wbcqp_compact.hpp) can get out of a CU's LDS, as a function of how many waves of the workgroup run it at once.  Same address pattern as the kernel
// (rows 74 doubles apart, a lane takes 14 of a row's 28 pairs in blocks of four: three 16-byte loads -- the row, the pending v, the pick's d -- and one
// 16-byte store per pair), the arithmetic of the pass, and three reduced forms: loads only, stores only, loads + stores without arithmetic.
// Per form and wave count: cycles per pass on the slowest wave, bytes moved, bytes per cycle over the workgroup.
//   make -C tools/ubench && gpurun -- tools/ubench/_build/lds_rowpass
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double2v __attribute__((ext_vector_type(2)));
constexpr int N = 74, LDJ = 74, PC = 18, REPS = 64;

__device__ __forceinline__ long long now()
{
    long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
__device__ __forceinline__ double2v ld2(const double* p) { return *reinterpret_cast<const double2v*>(__builtin_assume_aligned(p, 16)); }
__device__ __forceinline__ void st2(double* p, double2v v) { *reinterpret_cast<double2v*>(__builtin_assume_aligned(p, 16)) = v; }

// mode 0: the pass; 1: loads only; 2: stores only; 3: loads + stores, no arithmetic
template <int MODE>
__global__ __launch_bounds__(256) void k(long long* cyc, double* sink, int nwaves)
{
    extern __shared__ __align__(16) double lds[];
    double* J = lds;
    double* Vp = J + N * LDJ + 4;
    double* Vn = Vp + 80;
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < N * LDJ + 4 + 160; i += 256) lds[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    const int ne = 74, cs = PC & ~1, P = (ne - cs) >> 1, T = (P + 1) >> 1;
    // the kernel runs the pass on waves 0-2 (148 lanes, 74 rows); here `nwaves` waves run it, each on its own 32 rows (rows wrap: the pattern, not the result, matters)
    long long best = 0;
    double acc = 0.0;
    if (wave < nwaves) {
        const int idx = (tid >> 1) % N, hf = tid & 1;
        double* Jk = J + idx * LDJ;
        const double wk = 1e-9 * (idx + 1);
        const int p0 = hf * T, pe = (P < p0 + T) ? P : p0 + T;
        __builtin_amdgcn_s_barrier();
        const long long t0 = now();
        for (int rep = 0; rep < REPS; ++rep) {
            asm volatile("" ::: "memory"); // (no hoisting of the loads, no merging of the stores across passes)
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (int s0 = 0; s0 < T; s0 += 4) {
                double2v jv[4], vv[4], dv[4];
                int cc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + s0 + u;
                    cc[u] = (p < pe) ? cs + 2 * p : ne;
                    if (MODE != 2) {
                        jv[u] = ld2(Jk + cc[u]);
                        vv[u] = ld2(Vp + cc[u]);
                        dv[u] = ld2(Vn + cc[u]);
                    }
                    else {
                        jv[u].x = wk; jv[u].y = wk; vv[u] = jv[u]; dv[u] = jv[u];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (MODE == 0) {
                        jv[u].x = fma(-wk, vv[u].x, jv[u].x);
                        jv[u].y = fma(-wk, vv[u].y, jv[u].y);
                    }
                    if (MODE != 1) st2(Jk + cc[u], jv[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    if (MODE == 0) {
                        a0 = fma(jv[u].x, dv[u].x, a0);
                        a1 = fma(jv[u].y, dv[u].y, a1);
                        a2 = fma(jv[u + 1].x, dv[u + 1].x, a2);
                        a3 = fma(jv[u + 1].y, dv[u + 1].y, a3);
                    }
                    else {
                        a0 += jv[u].x + vv[u].y + dv[u].x;
                        a1 += jv[u + 1].y + vv[u + 1].x + dv[u + 1].y;
                    }
                }
            }
            acc += (a0 + a1) + (a2 + a3);
        }
        best = now() - t0;
    }
    if ((tid & 63) == 0) cyc[wave] = best;
    sink[tid] = acc;
}

template <int MODE>
void run(const char* name, long long* dc, double* ds)
{
    const size_t lds = (N * LDJ + 4 + 160) * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int nw = 1; nw <= 4; ++nw) {
        long long c[4] = {0, 0, 0, 0};
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), lds, 0, dc, ds, nw);
        hipDeviceSynchronize();
        hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
        long long worst = 0;
        for (int w = 0; w < nw; ++w) worst = c[w] > worst ? c[w] : worst;
        // per lane and pass: 14 pairs (16 slots, the two past the share hit the pad pair) x (3 loads, 1 store) x 16 B
        const double per_pass = (double)worst / REPS;
        const double rd = (MODE == 2) ? 0.0 : 64.0 * nw * 16 * 3 * 16, wr = (MODE == 1) ? 0.0 : 64.0 * nw * 16 * 16;
        printf("%-34s waves %d: %7.0f cycles per pass  (loads %6.1f KB, stores %5.1f KB: %6.1f B/clk over the workgroup)\n", name, nw, per_pass, rd / 1024, wr / 1024,
               (rd + wr) / per_pass);
    }
}

// calibration of the clock: 4096 dependent v_fma_f64 on one wave (4.3 cycles each, tools/ubench/chain_lat.hip)
__global__ __launch_bounds__(64) void kcal(long long* cyc, double* sink)
{
    double a = 1.0 + threadIdx.x * 1e-9;
    const double x = 1.0000001, b = 0.5;
    const long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < 64; ++i)
        asm volatile("v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\n" : "+v"(a) : "v"(x), "v"(b));
    const long long t1 = now();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    sink[threadIdx.x] = a;
}

int main()
{
    long long* dc;
    double* ds;
    hipMalloc(&dc, 4 * sizeof(long long));
    hipMalloc(&ds, 256 * sizeof(double));
    {
        long long c = 0;
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kcal, dim3(1), dim3(64), 0, 0, dc, ds);
        hipDeviceSynchronize();
        hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost);
        printf("clock: 1024 dependent v_fma_f64 = %lld ticks of s_memtime (%.2f per FMA; 4.3 core cycles each by tools/ubench/chain_lat.hip)\n", c, c / 1024.0);
    }
    run<0>("the row pass (update + z)", dc, ds);
    run<1>("its loads only", dc, ds);
    run<2>("its stores only", dc, ds);
    run<3>("loads + stores, no arithmetic", dc, ds);
    return 0;
}
