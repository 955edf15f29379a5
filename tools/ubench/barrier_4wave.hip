// Microbenchmark: cost of __syncthreads() and LDS traffic for a 256-thread workgroup alone on a CU (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc)
{
    extern __shared__ __align__(16) double lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) lds[i] = 1.0 + i * 1e-3;
    __syncthreads();
    double a0 = 0, a1 = 0;
    long long t0 = clock64();
    if (MODE == 0) { // bare barriers
        for (int it = 0; it < 1024; ++it) __syncthreads();
    }
    else if (MODE == 1) { // barrier + one LDS write/read exchange (typical phase boundary)
        for (int it = 0; it < 1024; ++it) {
            lds[tid] = a0 + it;
            __syncthreads();
            a0 += lds[(tid + 64) & 255];
        }
    }
    else if (MODE == 2) { // 4 waves streaming ds_read_b64, own rows (stride 75), 8 per iter, no barrier
        const double* p = lds + (tid & 63) * 75 + (tid >> 6) * 16;
        for (int it = 0; it < 256; ++it) {
            const double* q = p + (it & 1) * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) a0 += q[u];
        }
    }
    else if (MODE == 3) { // dependent single read latency
        int idx = tid;
        for (int it = 0; it < 256; ++it) { double v = lds[idx & 8191]; idx = (int)v + idx + 1; a0 += v; }
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + tid] = a0 + a1;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, double per)
{
    const int grid = 256;
    double* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * 256 * grid); (void)hipMalloc(&cyc, sizeof(long long) * grid);
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 100 * 1024, 0, out, cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    printf("%-52s : %8.0f cycles total, %7.1f cycles per op\n", name, avg, avg / per);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("__syncthreads() alone, 4 waves", 1024);
    run<1>("lds write + __syncthreads() + lds read", 1024);
    run<2>("4 waves x ds_read_b64 own row (per wave-instr)", 256 * 8);
    run<3>("dependent ds_read_b64 latency", 256);
    return 0;
}
