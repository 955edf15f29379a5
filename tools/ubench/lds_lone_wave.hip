// Microbenchmark: what ONE wavefront alone on a CU gets from LDS and the fp64 VALU (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_lone_wave.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_IT 256

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int stride)
{
    extern __shared__ __align__(16) double lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) lds[i] = 1.0 + i * 1e-3;
    __syncthreads();
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const double* p = lds + lane * stride;
    long long t0 = clock64();
    if (MODE == 0) { // ds_read_b64 own-row, 8 per iteration
        for (int it = 0; it < N_IT; ++it) {
            const double* q = p + (it & 7) * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) a0 += q[u * 2 * 0 + u];
        }
    }
    else if (MODE == 1) { // ds_read_b128
        for (int it = 0; it < N_IT; ++it) {
            const double2* q = reinterpret_cast<const double2*>(p + (it & 7) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) { double2 v = q[u]; a0 += v.x; a1 += v.y; }
        }
    }
    else if (MODE == 2) { // broadcast b64
        for (int it = 0; it < N_IT; ++it) {
            const double* q = lds + (it & 7) * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) a0 += q[u];
        }
    }
    else if (MODE == 3) { // dependent fma chain
        double x = p[0];
        for (int it = 0; it < N_IT; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a0 = fma(a0, x, 1.0);
        }
    }
    else if (MODE == 4) { // 4 independent fma chains
        double x = p[0];
        for (int it = 0; it < N_IT; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u) { a0 = fma(a0, x, 1.0); a1 = fma(a1, x, 1.0); a2 = fma(a2, x, 1.0); a3 = fma(a3, x, 1.0); }
        }
    }
    else if (MODE == 5) { // ds_write_b64 own-row
        double* w = lds + lane * stride;
        for (int it = 0; it < N_IT; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) w[(it & 7) * 8 + u] = a0 + u;
        }
    }
    else if (MODE == 6) { // fma + b64 read mix: 1 own read + 1 bcast read per 2 fma (TWO-like)
        for (int it = 0; it < N_IT; ++it) {
            const double* q = p + (it & 7) * 8;
            const double* b = lds + 4096 + (it & 7) * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 = fma(q[u], b[u], a0); a1 = fma(q[u], b[u], a1); }
        }
    }
    else if (MODE == 7) { // same with b128
        for (int it = 0; it < N_IT; ++it) {
            const double2* q = reinterpret_cast<const double2*>(p + (it & 7) * 8);
            const double2* b = reinterpret_cast<const double2*>(lds + 4096 + (it & 7) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) { double2 v = q[u], w = b[u]; a0 = fma(v.x, w.x, a0); a1 = fma(v.y, w.y, a1); a2 = fma(v.x, w.x, a2); a3 = fma(v.y, w.y, a3); }
        }
    }
    else if (MODE == 8) { // sqrt+div chain
        double x = p[0] + 2.0;
        for (int it = 0; it < N_IT / 8; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { x = sqrt(x) + 1.5; }
        }
        a0 = x;
    }
    else if (MODE == 9) { // division chain
        double x = p[0] + 2.0;
        for (int it = 0; it < N_IT / 8; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { x = 3.0 / x + 1.5; }
        }
        a0 = x;
    }
    else if (MODE == 10) { // rsqrt chain
        double x = p[0] + 2.0;
        for (int it = 0; it < N_IT / 8; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { x = rsqrt(x) + 1.5; }
        }
        a0 = x;
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int stride, double per, int grid)
{
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * 64 * grid); hipMalloc(&cyc, sizeof(long long) * grid);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 8192 * 8 + 70000, 0, out, cyc, stride);
    hipDeviceSynchronize();
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    printf("%-44s stride %3d : %8.0f cycles total, %6.1f cycles per %s\n", name, stride, avg, avg / per, "op");
    hipFree(out); hipFree(cyc);
}

int main()
{
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int G = 256; // one block per CU (LDS > 80 KiB keeps a second block off the CU)
    run<0>("ds_read_b64 own row (8/iter)", 75, N_IT * 8, G);
    run<0>("ds_read_b64 own row (8/iter)", 78, N_IT * 8, G);
    run<1>("ds_read_b128 own row (4/iter)", 78, N_IT * 4, G);
    run<1>("ds_read_b128 own row (4/iter)", 76, N_IT * 4, G);
    run<2>("ds_read_b64 broadcast (8/iter)", 0, N_IT * 8, G);
    run<3>("v_fma_f64 dependent chain", 75, N_IT * 8, G);
    run<4>("v_fma_f64 4 chains", 75, N_IT * 8, G);
    run<5>("ds_write_b64 own row", 75, N_IT * 8, G);
    run<6>("2 fma + own b64 + bcast b64", 75, N_IT * 8, G);
    run<7>("4 fma + own b128 + bcast b128", 78, N_IT * 4, G);
    run<8>("sqrt chain", 75, N_IT, G);
    run<9>("div chain", 75, N_IT, G);
    run<10>("rsqrt chain", 75, N_IT, G);
    return 0;
}
