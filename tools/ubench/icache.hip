// Microbenchmark: what straight-line code costs on gfx950 when it does not fit / is not yet in the instruction cache.
// A body of N x 4 independent v_fma_f64 (8 bytes each => 32 N bytes of code) is executed REPS times inside one launch;
// the first pass is cold, later passes are warm if the body fits.  Run with 1 workgroup (a lone CU) and with 512
// (every CU busy, two workgroups per CU back to back).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int REPS = 4;

template <int N>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, double a, double b, int reps)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int r = 0; r < reps; ++r) {
        const long long t0 = clock64();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            x0 = fma(x0, a, b);
            x1 = fma(x1, a, b);
            x2 = fma(x2, a, b);
            x3 = fma(x3, a, b);
        }
        const long long t1 = clock64();
        if (threadIdx.x == 0) cyc[blockIdx.x * REPS + r] = t1 - t0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <int N>
void run(int grid)
{
    double* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * 256 * grid); (void)hipMalloc(&cyc, sizeof(long long) * grid * REPS);
    hipLaunchKernelGGL(k<N>, dim3(grid), dim3(256), 0, 0, out, cyc, 0.999, 1e-3, REPS);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(grid * REPS);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * grid * REPS, hipMemcpyDeviceToHost);
    printf("code %4d KB, grid %3d :", 32 * N / 1024, grid);
    for (int r = 0; r < REPS; ++r) {
        double avg = 0; for (int g = 0; g < grid; ++g) avg += h[g * REPS + r]; avg /= grid;
        printf("  pass %d %6.2f cyc/fma", r, avg / (4.0 * N));
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    for (int grid : {1, 256, 512}) {
        run<256>(grid);
        run<1024>(grid);
        run<2048>(grid);
        run<3072>(grid);
        run<4096>(grid);
    }
    return 0;
}
