// Microbenchmark + layout check for v_mfma_f64_16x16x4_f64 on gfx950.
//  (1) layout: D = A(16x4) * B(4x16) with asymmetric integer data; lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
//      result register r of lane l is expected at row (l>>4) + 4 r, column l&15.
//  (2) cycles per MFMA on one SIMD with 1 / 2 / 4 independent accumulators (one wave per SIMD, 4 waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void layout_k(double* out)
{
    const int l = threadIdx.x;
    const double a = (double)((l & 15) * 10 + (l >> 4) + 1);     // A[i][k] = 10 i + k + 1
    const double b = (double)(((l >> 4) + 1) * 100 + (l & 15));  // B[k][j] = 100 (k+1) + j
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void time_k(double* out, long long* cyc, double a, double b, int iters)
{
    double4_t c[NACC];
    for (int q = 0; q < NACC; ++q) c[q] = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < NACC; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// MFMA with VALU f64 FMAs from the same wave in between: does the matrix pipe run beside the vector pipe?
__global__ __launch_bounds__(256) void mix_k(double* out, long long* cyc, double a, double b, int iters, int nf)
{
    double4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        for (int f = 0; f < nf; ++f) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); }
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    double* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * 256 * 256); (void)hipMalloc(&cyc, sizeof(long long) * 256);
    hipLaunchKernelGGL(layout_k, dim3(1), dim3(64), 0, 0, out);
    std::vector<double> h(256);
    (void)hipMemcpy(h.data(), out, sizeof(double) * 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = (l >> 4) + 4 * r, col = l & 15;
            double ref = 0;
            for (int k = 0; k < 4; ++k) ref += (double)(row * 10 + k + 1) * (double)((k + 1) * 100 + col);
            if (h[l * 4 + r] != ref) ++bad;
        }
    printf("layout row=(lane>>4)+4*reg, col=lane&15 : %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    const int iters = 1024;
    auto report = [&](const char* name, int nmfma) {
        (void)hipDeviceSynchronize();
        std::vector<long long> hc(256);
        (void)hipMemcpy(hc.data(), cyc, sizeof(long long) * 256, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : hc) avg += v; avg /= 256;
        printf("%-44s : %7.1f cycles per MFMA\n", name, avg / nmfma);
    };
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(time_k<1>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0, 1e-3, iters);
    report("1 accumulator (dependent chain)", iters);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(time_k<2>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0, 1e-3, iters);
    report("2 independent accumulators", 2 * iters);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(time_k<4>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0, 1e-3, iters);
    report("4 independent accumulators", 4 * iters);
    for (int nf : {0, 1, 2, 3, 4}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(mix_k, dim3(256), dim3(256), 0, 0, out, cyc, 0.999, 1e-3, iters, nf);
        char nm[96];
        snprintf(nm, sizeof nm, "2 MFMA + %d v_fma_f64 per iteration (per iter/2)", 4 * nf);
        report(nm, 2 * iters);
    }
    return 0;
}
